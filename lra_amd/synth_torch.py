"""Vectorised (torch, CPU or GPU) versions of the synthetic-input generators in synth.py, used by
bench.py to build large batches in seconds: a whole batch of simulated reads with their true
alignment blocks, the between-anchor gap problems for a12, perturbed block lists for a14, and the
stand-in global minimizer index (see synth.py for what the stand-in does and does not reproduce).
torch is plumbing here (device memory + bulk tensor ops); none of this is on the timed path."""
import numpy as np
import torch

_B = torch.tensor(list(b"ACGT"), dtype=torch.uint8)
_COMP = torch.zeros(256, dtype=torch.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b
_CODE = torch.zeros(256, dtype=torch.int64)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i


def make_genome(n, seed, device, repeat_frac=0.25, fam_len=(300, 1000, 6000, 300), divergence=0.10):
    g = torch.Generator(device=device).manual_seed(seed)
    B = _B.to(device)
    gen = B[torch.randint(0, 4, (n,), generator=g, device=device)]
    fams = [B[torch.randint(0, 4, (L,), generator=g, device=device)] for L in fam_len]
    target, placed = int(n * repeat_frac), 0
    cpu = np.random.default_rng(seed)
    while placed < target:
        f = fams[int(cpu.integers(0, len(fams)))]
        c = f.clone()
        mut = torch.rand(len(c), generator=g, device=device) < divergence
        c[mut] = B[torch.randint(0, 4, (int(mut.sum()),), generator=g, device=device)]
        p = int(cpu.integers(0, n - len(c)))
        gen[p:p + len(c)] = c
        placed += len(c)
    return gen


def simulate_batch(genome, n_reads, mean_len, sd_len, err, mix, seed):
    """Forward-strand reads of the batch with truth.  Returns dict of tensors on genome.device:
    seq (uint8, concatenated), off (int64 [R+1]), blocks (int32 [nb,3]: read-relative qPos, genome tPos, len),
    block_off (int64 [R+1])."""
    dev = genome.device
    g = torch.Generator(device=dev).manual_seed(seed)
    B = _B.to(dev)
    CODE = _CODE.to(dev)
    n = genome.numel()
    lens = torch.clamp((torch.randn(n_reads, generator=g, device=dev) * sd_len + mean_len).long(), 1000, n - 2)
    starts = (torch.rand(n_reads, generator=g, device=dev, dtype=torch.float64) * (n - lens).double()).long()
    roff = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
    roff[1:] = torch.cumsum(lens, 0)
    total = int(roff[-1])
    rid = torch.repeat_interleave(torch.arange(n_reads, device=dev), lens)
    rel = torch.arange(total, device=dev) - roff[rid]
    gpos = starts[rid] + rel
    src = genome[gpos]
    r = torch.rand(total, generator=g, device=dev)
    ps, pi, pd = (err * m / sum(mix) for m in mix)
    sub = r < ps
    ins = (r >= ps) & (r < ps + pi)
    dele = (r >= ps + pi) & (r < ps + pi + pd)
    first = rel == 0
    last = rel == lens[rid] - 1
    dele &= ~(first | last)
    ins &= ~last
    base = src.clone()
    nsub = int(sub.sum())
    base[sub] = B[(CODE[src[sub].long()] + torch.randint(1, 4, (nsub,), generator=g, device=dev)) % 4]
    counts = torch.ones(total, dtype=torch.int64, device=dev)
    counts[ins] = 2
    counts[dele] = 0
    out = torch.repeat_interleave(base, counts)
    ends = torch.cumsum(counts, 0)
    nins = int(ins.sum())
    out[ends[ins] - 1] = B[torch.randint(0, 4, (nins,), generator=g, device=dev)]
    # read offsets in the output
    ooff = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
    ooff[1:] = ends[roff[1:] - 1]
    qabs = ends - counts                       # output index of every reference base
    keep = ~dele
    idx = torch.nonzero(keep).squeeze(1)
    brk = torch.ones(idx.numel(), dtype=torch.bool, device=dev)
    brk[1:] = (idx[1:] - idx[:-1] != 1) | ins[idx[:-1]] | (rid[idx[1:]] != rid[idx[:-1]])
    st = torch.nonzero(brk).squeeze(1)
    blen = torch.diff(torch.cat([st, torch.tensor([idx.numel()], device=dev)]))
    bidx = idx[st]
    brid = rid[bidx]
    blocks = torch.stack([qabs[bidx] - ooff[brid], gpos[bidx], blen], 1).to(torch.int32)
    bcnt = torch.bincount(brid, minlength=n_reads)
    boff = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
    boff[1:] = torch.cumsum(bcnt, 0)
    return {"seq": out, "off": ooff, "blocks": blocks, "block_off": boff, "block_rid": brid}


def revcomp_some(seq, off, rev_mask):
    """Copy of the concatenated reads with the reads selected by rev_mask reverse-complemented in place."""
    dev = seq.device
    lens = off[1:] - off[:-1]
    rid = torch.repeat_interleave(torch.arange(lens.numel(), device=dev), lens)
    pos = torch.arange(seq.numel(), device=dev)
    rel = pos - off[rid]
    srcpos = torch.where(rev_mask[rid], off[rid] + lens[rid] - 1 - rel, pos)
    out = seq[srcpos]
    comp = _COMP.to(dev)[out.long()]
    return torch.where(rev_mask[rid], comp, out)


def gap_problems(sim, min_anchor=12, local_band=15):
    """a12 inputs: gaps between consecutive anchors (true blocks >= min_anchor) of every read.
    Offsets are into the read buffer (q) and the genome (t); k as LocalRefineAlignment.h:101-115."""
    b, rid = sim["blocks"].long(), sim["block_rid"]
    isanc = b[:, 2] >= min_anchor
    a, ar = b[isanc], rid[isanc]
    same = ar[1:] == ar[:-1]
    qs = a[:-1, 0] + a[:-1, 2]; qe = a[1:, 0]
    ts = a[:-1, 1] + a[:-1, 2]; te = a[1:, 1]
    ok = same & ((qe - qs > 0) | (te - ts > 0))
    qs, qe, ts, te, r = qs[ok], qe[ok], ts[ok], te[ok], ar[:-1][ok]
    k = torch.clamp((qe - qs - (te - ts)).abs() * 2 + 1, max=local_band)
    return {"q_off": sim["off"][r] + qs, "q_len": (qe - qs).to(torch.int32), "t_off": ts, "t_len": (te - ts).to(torch.int32),
            "k": k.to(torch.int32), "rid": r}


def perturbed_blocks(sim, seed, drop=0.15, trim=0.3):
    """a14 inputs: truth blocks with some removed and some ends trimmed (what seed extension hands over)."""
    dev = sim["blocks"].device
    g = torch.Generator(device=dev).manual_seed(seed)
    b, rid, boff = sim["blocks"].clone(), sim["block_rid"], sim["block_off"]
    nb = b.shape[0]
    keep = torch.rand(nb, generator=g, device=dev) > drop
    keep[boff[:-1][boff[:-1] < nb]] = True
    keep[boff[1:] - 1] = True
    tr = (b[:, 2] > 6) & (torch.rand(nb, generator=g, device=dev) < trim)
    a = torch.randint(0, 3, (nb,), generator=g, device=dev, dtype=torch.int32) * tr
    z = torch.randint(0, 3, (nb,), generator=g, device=dev, dtype=torch.int32) * tr
    b[:, 0] += a; b[:, 1] += a; b[:, 2] -= (a + z)
    keep &= b[:, 2] > 0
    b, rid = b[keep], rid[keep]
    cnt = torch.bincount(rid, minlength=boff.numel() - 1)
    off = torch.zeros(boff.numel(), dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(cnt, 0)
    return b.contiguous(), off


def build_global_index(genome, k, w, max_freq, chunk=1 << 26):
    """Stand-in for StoreIndex (see synth.py): (w,k) window minimizers (leftmost minimum), keys with more
    than max_freq occurrences dropped, sorted by masked key.  Returns numpy (key uint64, pos uint32)."""
    assert k <= 31
    dev = genome.device
    CODE = _CODE.to(dev)
    n = genome.numel() - k + 1
    sel = torch.zeros(n, dtype=torch.bool, device=dev)
    for s in range(0, max(n - w + 1, 1), chunk):
        e = min(n, s + chunk + w - 1)
        c = CODE[genome[s:e + k - 1].long()]
        m = e - s
        fwd = torch.zeros(m, dtype=torch.int64, device=dev)
        rc = torch.zeros(m, dtype=torch.int64, device=dev)
        for i in range(k):
            ci = c[i:i + m]
            fwd = (fwd << 2) | ci
            rc |= (3 - ci) << (2 * i)
        key = torch.minimum(fwd, rc)
        mm = m - w + 1
        if mm <= 0:
            continue
        best = key[:mm].clone()
        arg = torch.zeros(mm, dtype=torch.int64, device=dev)
        for j in range(1, w):
            kj = key[j:j + mm]
            lt = kj < best
            best = torch.where(lt, kj, best)
            arg = torch.where(lt, torch.full_like(arg, j), arg)
        sel[s + arg + torch.arange(mm, device=dev)] = True
        del c, fwd, rc, key, best, arg
    pos = torch.nonzero(sel).squeeze(1)
    del sel
    CODEg = CODE[genome.long()] if genome.numel() < (1 << 28) else None
    fwd = torch.zeros(pos.numel(), dtype=torch.int64, device=dev)
    rc = torch.zeros(pos.numel(), dtype=torch.int64, device=dev)
    for i in range(k):
        ci = CODEg[pos + i] if CODEg is not None else CODE[genome[pos + i].long()]
        fwd = (fwd << 2) | ci
        rc |= (3 - ci) << (2 * i)
    usef = fwd < rc
    key = torch.where(usef, fwd, rc)
    order = torch.argsort(key, stable=True)
    key, pos, usef = key[order], pos[order], usef[order]
    _, inv, cnt = torch.unique_consecutive(key, return_inverse=True, return_counts=True)
    keepm = cnt[inv] <= max_freq
    key, pos, usef = key[keepm], pos[keepm], usef[keepm]
    raw = key.cpu().numpy().astype(np.uint64) | ((~usef).cpu().numpy().astype(np.uint64) << np.uint64(63))
    return raw, pos.cpu().numpy().astype(np.uint32)
