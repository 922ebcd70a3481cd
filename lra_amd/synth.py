"""Seeded synthetic inputs (only this repo reaches the GPU box, so genomes and
reads are generated there): a random genome with repeat families, a long-read simulator with
truth, and a simple global minimizer index builder.

The index builder is a stand-in for `lra index` (StoreIndex, MMIndex.h:286-400), which is row
(f)1 "next" of the scope table: it yields a valid `.mms`-shaped payload (tuples sorted by masked
key, strand flag in bit 63, over-frequent keys dropped) but does not reproduce StoreIndex's
window thinning; every parity test feeds the SAME payload to the oracle and to the HIP path.
"""
import numpy as np

COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    COMP[a] = b
CODE = np.zeros(256, dtype=np.int64)
for i, c in enumerate(b"ACGT"):
    CODE[c] = i
    CODE[c + 32] = i
BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def make_genome(n, seed=1, repeat_frac=0.25, n_families=4, fam_len=(300, 1000, 6000, 300), divergence=0.10):
    """Uniform ACGT of length n with `repeat_frac` of it overwritten by diverged copies of a few families."""
    rng = np.random.default_rng(seed)
    g = BASES[rng.integers(0, 4, size=n)]
    fams = [BASES[rng.integers(0, 4, size=fam_len[i % len(fam_len)])] for i in range(n_families)]
    target = int(n * repeat_frac)
    placed = 0
    while placed < target:
        f = fams[int(rng.integers(0, n_families))]
        if len(f) >= n:
            break
        c = f.copy()
        mut = rng.random(len(c)) < divergence
        c[mut] = BASES[rng.integers(0, 4, size=int(mut.sum()))]
        p = int(rng.integers(0, n - len(c)))
        g[p:p + len(c)] = c
        placed += len(c)
    return g


def revcomp(a):
    return COMP[a[::-1]]


def simulate_read(rng, genome, length, err, mix, rev):
    """One read: genome[start:start+length] with errors (mix = sub:ins:del), optional reverse strand.
    Returns (read uint8, start, ref_len_used, strand)."""
    n = len(genome)
    length = int(max(50, min(length, n - 1)))
    start = int(rng.integers(0, n - length))
    src = genome[start:start + length]
    r = rng.random(length)
    ps, pi, pd = (err * m / sum(mix) for m in mix)
    sub = r < ps
    ins = (r >= ps) & (r < ps + pi)
    dele = (r >= ps + pi) & (r < ps + pi + pd)
    base = src.copy()
    base[sub] = BASES[(CODE[src[sub]] + rng.integers(1, 4, size=int(sub.sum()))) % 4]
    counts = np.ones(length, dtype=np.int64)
    counts[ins] = 2
    counts[dele] = 0
    out = np.repeat(base, counts)
    # second copy of an inserted base becomes a random base
    ends = np.cumsum(counts)
    ins_pos = ends[ins] - 1
    out[ins_pos] = BASES[rng.integers(0, 4, size=len(ins_pos))]
    if rev:
        out = revcomp(out)
    return out, start, length, int(rev)


def simulate_read_with_blocks(rng, genome, length, err, mix):
    """Forward-strand read plus its TRUE alignment as gapless blocks (qPos, tPos, len) in absolute
    read / genome coordinates (substitutions stay inside blocks; an insertion puts one extra read
    base after its reference base; a deletion skips the reference base)."""
    n = len(genome)
    length = int(max(50, min(length, n - 1)))
    start = int(rng.integers(0, n - length))
    src = genome[start:start + length]
    r = rng.random(length)
    ps, pi, pd = (err * m / sum(mix) for m in mix)
    sub = r < ps
    ins = (r >= ps) & (r < ps + pi)
    dele = (r >= ps + pi) & (r < ps + pi + pd)
    dele[0] = dele[-1] = False
    ins[-1] = False
    base = src.copy()
    base[sub] = BASES[(CODE[src[sub]] + rng.integers(1, 4, size=int(sub.sum()))) % 4]
    counts = np.ones(length, dtype=np.int64)
    counts[ins] = 2
    counts[dele] = 0
    out = np.repeat(base, counts)
    ends = np.cumsum(counts)
    out[ends[ins] - 1] = BASES[rng.integers(0, 4, size=int(ins.sum()))]
    qpos = ends - counts                      # read index of each reference base (if kept)
    keep = ~dele
    # a block breaks after an insertion and around deletions
    idx = np.nonzero(keep)[0]
    brk = np.ones(len(idx), dtype=bool)
    brk[1:] = (np.diff(idx) != 1) | ins[idx[:-1]]
    starts = np.nonzero(brk)[0]
    lens = np.diff(np.append(starts, len(idx)))
    blocks = np.stack([qpos[idx[starts]], idx[starts] + start, lens], axis=1).astype(np.int32)
    return out, blocks


def simulate_reads(genome, n_reads, mean_len, sd_len, err, mix=(30, 35, 35), seed=3, rev_frac=0.5):
    rng = np.random.default_rng(seed)
    reads, truth = [], []
    for _ in range(n_reads):
        L = int(rng.normal(mean_len, sd_len))
        r, s, l, st = simulate_read(rng, genome, L, err, mix, rng.random() < rev_frac)
        reads.append(r)
        truth.append((s, l, st))
    return reads, truth


def canonical_keys(seq, k):
    """Per position p (0..n-k): (masked key, strand) of the canonical k-mer, as the reference defines it
    (MinCount.h:60-61: forward if fwd < rc else reverse-complement with bit 63 set)."""
    c = CODE[seq]
    n = len(seq) - k + 1
    fwd = np.zeros(n, dtype=np.uint64)
    rc = np.zeros(n, dtype=np.uint64)
    for i in range(k):
        fwd = (fwd << np.uint64(2)) | c[i:i + n].astype(np.uint64)
        rc |= (np.uint64(3) - c[i:i + n].astype(np.uint64)) << np.uint64(2 * i)
    use_f = fwd < rc
    key = np.where(use_f, fwd, rc)
    return key, ~use_f


def build_global_index(genome, k, w, max_freq, chunk=1 << 22):
    """(w,k)-window minimizers of the genome (leftmost minimum of every window), keys with more than
    max_freq occurrences dropped, sorted by masked key.  Returns (key uint64 with strand in bit 63, pos uint32)."""
    n = len(genome) - k + 1
    sel_pos = []
    for s in range(0, max(n - w + 1, 1), chunk):
        e = min(n, s + chunk + w - 1)
        key, _ = canonical_keys(genome[s:e + k - 1], k)
        m = len(key) - w + 1
        if m <= 0:
            continue
        best = key[:m].copy()
        arg = np.zeros(m, dtype=np.int64)
        for j in range(1, w):
            kj = key[j:j + m]
            lt = kj < best
            best[lt] = kj[lt]
            arg[lt] = j
        sel_pos.append(np.unique(arg + np.arange(m)) + s)
    pos = np.unique(np.concatenate(sel_pos)) if sel_pos else np.zeros(0, dtype=np.int64)
    # keys of the selected positions
    keys = np.zeros(len(pos), dtype=np.uint64)
    strand = np.zeros(len(pos), dtype=bool)
    c = CODE[genome]
    fwd = np.zeros(len(pos), dtype=np.uint64)
    rc = np.zeros(len(pos), dtype=np.uint64)
    for i in range(k):
        ci = c[pos + i].astype(np.uint64)
        fwd = (fwd << np.uint64(2)) | ci
        rc |= (np.uint64(3) - ci) << np.uint64(2 * i)
    use_f = fwd < rc
    keys = np.where(use_f, fwd, rc)
    strand = ~use_f
    order = np.argsort(keys, kind="stable")
    keys, pos, strand = keys[order], pos[order], strand[order]
    # drop over-frequent keys (StoreIndex: globalMaxFreq, MMIndex.h:286-400)
    if len(keys):
        _, inv, cnt = np.unique(keys, return_inverse=True, return_counts=True)
        keep = cnt[inv] <= max_freq
        keys, pos, strand = keys[keep], pos[keep], strand[keep]
    raw = keys | (strand.astype(np.uint64) << np.uint64(63))
    return raw, pos.astype(np.uint32)
