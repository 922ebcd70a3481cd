"""Seeded synthetic stand-in for GRCh38 (only /root/repo travels to the GPU box, so the reference genome is generated
there): 24 chromosomes with the GRCh38 length table, about half of the bases in interspersed repeats (families of 300 bp / 1 kb / 6 kb
consensus sequences, every copy 5-20 % diverged, half of them reverse-complemented, long ones truncated), alpha-satellite-like arrays of a
171-base monomer, and runs of N at the chromosome ends and the centromere.  `scale` shrinks every length (tests use 1/1000).  Reads come
from simulate_reads_sv: N(mean, sd) bases, a given error mix, half of them from the reverse strand, a fraction carrying one planted
structural variant (deletion, novel insertion, inversion, tandem duplication, translocation).
torch is plumbing here (device memory + bulk tensor ops); nothing in this file is on the timed path."""
import numpy as np
import torch

# GRCh38 primary assembly, chr1..22, X, Y
GRCH38_LEN = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622, 133275309,
              114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]
GRCH38_NAMES = [b"chr%d" % i for i in range(1, 23)] + [b"chrX", b"chrY"]

_B = torch.tensor(list(b"ACGT"), dtype=torch.uint8)
_COMP = torch.zeros(256, dtype=torch.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b


def _rand_bases(n, g, dev):
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    B = _B.to(dev)
    step = 1 << 28
    for s in range(0, n, step):
        e = min(n, s + step)
        out[s:e] = B[torch.randint(0, 4, (e - s,), generator=g, device=dev, dtype=torch.int32).long()]
    return out


def make_grch38_like(device, scale=1.0, seed=3, repeat_frac=0.47, satellite_frac=0.03, n_gaps=True):
    """-> (genome uint8 tensor on device, chrom_pos list [n+1], names list).  Deterministic in (scale, seed)."""
    dev = torch.device(device)
    lens = [max(2000, int(round(L * scale))) for L in GRCH38_LEN]
    chrom_pos = [0]
    for L in lens:
        chrom_pos.append(chrom_pos[-1] + L)
    G = chrom_pos[-1]
    g = torch.Generator(device=dev).manual_seed(seed)
    rng = np.random.default_rng(seed)
    genome = _rand_bases(G, g, dev)
    B = _B.to(dev)
    # ---- repeat library: families back to back in one tensor
    fam_len = [300] * 40 + [1000] * 40 + [6000] * 20
    fam_w = np.array([3.0] * 40 + [1.0] * 40 + [0.6] * 20)               # relative copy numbers (Alu-like families are the most numerous)
    fam_off = np.concatenate([[0], np.cumsum(fam_len)])
    lib = _rand_bases(int(fam_off[-1]), g, dev)
    # ---- instances: family, sub-interval (long families are mostly truncated), divergence, strand, position
    target = int(G * repeat_frac)
    mean_len = float(np.sum(fam_w / fam_w.sum() * np.array(fam_len) * np.where(np.array(fam_len) >= 6000, 0.4, 1.0)))
    n_inst = max(1, int(target / mean_len))
    fam = rng.choice(len(fam_len), size=n_inst, p=fam_w / fam_w.sum())
    full = np.array(fam_len)[fam]
    frac = np.where(full >= 6000, rng.uniform(0.05, 0.75, n_inst), 1.0)
    ilen = np.maximum(100, (full * frac).astype(np.int64))
    ilen = np.minimum(ilen, full)
    isub = (rng.uniform(0, 1, n_inst) * (full - ilen + 1)).astype(np.int64)
    idiv = rng.uniform(0.05, 0.20, n_inst)
    irev = rng.uniform(0, 1, n_inst) < 0.5
    ipos = (rng.uniform(0, 1, n_inst) * (G - 6001)).astype(np.int64)
    order = np.argsort(ipos, kind="stable")                                # later instances overwrite earlier ones where they overlap, left to right
    fam, ilen, isub, idiv, irev, ipos = fam[order], ilen[order], isub[order], idiv[order], irev[order], ipos[order]
    chunk = 1 << 20                                                        # instances per bulk operation
    for s in range(0, n_inst, chunk):
        e = min(n_inst, s + chunk)
        L = torch.from_numpy(ilen[s:e]).to(dev)
        tot = int(L.sum())
        iid = torch.repeat_interleave(torch.arange(e - s, device=dev), L)
        off = torch.zeros(e - s + 1, dtype=torch.int64, device=dev); off[1:] = torch.cumsum(L, 0)
        j = torch.arange(tot, device=dev) - off[iid]
        rev = torch.from_numpy(irev[s:e]).to(dev)[iid]
        src0 = torch.from_numpy(fam_off[fam[s:e]] + isub[s:e]).to(dev)[iid]
        jj = torch.where(rev, L[iid] - 1 - j, j)
        b = lib[src0 + jj]
        b = torch.where(rev, _COMP.to(dev)[b.long()], b)
        mut = torch.rand(tot, generator=g, device=dev) < torch.from_numpy(idiv[s:e]).to(dev).to(torch.float32)[iid]
        nm = int(mut.sum())
        b[mut] = B[torch.randint(0, 4, (nm,), generator=g, device=dev)]
        genome[torch.from_numpy(ipos[s:e]).to(dev)[iid] + j] = b
        del iid, j, rev, src0, jj, b, mut
    # ---- satellite arrays: a 171-base monomer repeated for 0.5-3 Mb (x scale), 2 % divergence between monomers, one array per chromosome
    mono = _rand_bases(171, g, dev)
    for c in range(len(lens)):
        alen = int(lens[c] * satellite_frac)
        if alen < 1000:
            continue
        a0 = chrom_pos[c] + int(lens[c] * 0.40)
        arr = mono.repeat((alen + 170) // 171)[:alen].clone()
        mut = torch.rand(alen, generator=g, device=dev) < 0.02
        arr[mut] = B[torch.randint(0, 4, (int(mut.sum()),), generator=g, device=dev)]
        genome[a0:a0 + alen] = arr
    # ---- N runs: 10 kb (x scale) at both ends of every chromosome, a centromere gap of 1 % of its length
    if n_gaps:
        for c in range(len(lens)):
            t = max(20, int(10_000 * scale))
            genome[chrom_pos[c]:chrom_pos[c] + t] = ord("N")
            genome[chrom_pos[c + 1] - t:chrom_pos[c + 1]] = ord("N")
            cg = int(lens[c] * 0.01)
            c0 = chrom_pos[c] + int(lens[c] * 0.44)
            genome[c0:c0 + cg] = ord("N")
    return genome, chrom_pos, list(GRCH38_NAMES)


def simulate_reads_sv(genome, chrom_pos, n_reads, mean_len, sd_len, err, mix, seed, sv_frac=0.05):
    """Reads with truth-free simulation for the throughput workload: every read is up to three reference segments (start, length, strand)
    plus an optional novel insertion between the first two, then substitutions / insertions / deletions at rate err split as mix.
    -> dict(seq uint8 (concatenated, 64 bytes of zero padding behind), off int64 [R+1], sv int8 [R]: 0 none, 1 deletion, 2 insertion,
    3 inversion, 4 tandem duplication, 5 translocation, start int64 [R] (genome offset of the first segment), rev bool [R])."""
    dev = genome.device
    g = torch.Generator(device=dev).manual_seed(seed)
    rng = np.random.default_rng(seed)
    B = _B.to(dev)
    G = int(genome.numel())
    cp = np.asarray(chrom_pos, dtype=np.int64)
    lens = np.clip((rng.normal(mean_len, sd_len, n_reads)).astype(np.int64), 1000, None)
    # the read's locus: inside one chromosome, away from its ends
    chrom = rng.choice(len(cp) - 1, size=n_reads, p=(cp[1:] - cp[:-1]) / (cp[-1] - cp[0]))
    clen = cp[chrom + 1] - cp[chrom]
    lens = np.minimum(lens, np.maximum(1000, clen // 2))
    room = np.maximum(1, clen - lens - 20_000)
    start = cp[chrom] + (rng.uniform(0, 1, n_reads) * room).astype(np.int64)
    sv = np.where(rng.uniform(0, 1, n_reads) < sv_frac, rng.integers(1, 6, n_reads), 0).astype(np.int8)
    cut = (lens * rng.uniform(0.3, 0.7, n_reads)).astype(np.int64)          # where the variant sits in the read
    svlen = np.exp(rng.uniform(np.log(50), np.log(10_000), n_reads)).astype(np.int64)
    # segments: (gstart, len, rev); a novel insertion is a segment with gstart < 0
    seg_s = np.zeros((n_reads, 3), np.int64); seg_l = np.zeros((n_reads, 3), np.int64); seg_r = np.zeros((n_reads, 3), bool)
    seg_s[:, 0] = start; seg_l[:, 0] = lens
    for kind in (1, 2, 3, 4, 5):
        m = sv == kind
        if not m.any():
            continue
        seg_l[m, 0] = cut[m]
        rest = lens[m] - cut[m]
        if kind == 1:                                                      # deletion: skip svlen reference bases
            seg_s[m, 1] = start[m] + cut[m] + svlen[m]; seg_l[m, 1] = rest
        elif kind == 2:                                                    # novel insertion of svlen bases
            seg_s[m, 1] = -1; seg_l[m, 1] = np.minimum(svlen[m], 5000)
            seg_s[m, 2] = start[m] + cut[m]; seg_l[m, 2] = rest
        elif kind == 3:                                                    # inversion of min(svlen, rest / 2) bases
            il = np.maximum(50, np.minimum(svlen[m], rest // 2))
            seg_s[m, 1] = start[m] + cut[m]; seg_l[m, 1] = il; seg_r[m, 1] = True
            seg_s[m, 2] = start[m] + cut[m] + il; seg_l[m, 2] = rest - il
        elif kind == 4:                                                    # tandem duplication: the last min(svlen, cut) bases again
            dl = np.minimum(svlen[m], cut[m])
            seg_s[m, 1] = start[m] + cut[m] - dl; seg_l[m, 1] = dl + rest
        else:                                                              # translocation: the rest comes from another locus, either strand
            o = (rng.uniform(0, 1, int(m.sum())) * (G - rest - 1)).astype(np.int64)
            seg_s[m, 1] = o; seg_l[m, 1] = rest; seg_r[m, 1] = rng.uniform(0, 1, int(m.sum())) < 0.5
    seg_s = np.minimum(seg_s, G - 1 - seg_l)                                # keep every segment inside the genome
    seg_s = np.where(seg_l > 0, seg_s, 0)
    rlen = seg_l.sum(1)
    S = torch.from_numpy(seg_s.reshape(-1)).to(dev); Ls = torch.from_numpy(seg_l.reshape(-1)).to(dev); Rv = torch.from_numpy(seg_r.reshape(-1)).to(dev)
    tot = int(Ls.sum())
    sid = torch.repeat_interleave(torch.arange(3 * n_reads, device=dev), Ls)
    soff = torch.zeros(3 * n_reads + 1, dtype=torch.int64, device=dev); soff[1:] = torch.cumsum(Ls, 0)
    j = torch.arange(tot, device=dev) - soff[sid]
    rv = Rv[sid]
    novel = S[sid] < 0
    gp = torch.where(rv, S[sid] + Ls[sid] - 1 - j, S[sid] + j).clamp_(0, G - 1)
    src = genome[gp]
    src = torch.where(rv, _COMP.to(dev)[src.long()], src)
    nn = int(novel.sum())
    if nn:
        src[novel] = B[torch.randint(0, 4, (nn,), generator=g, device=dev)]
    isN = src == ord("N")
    nN = int(isN.sum())
    if nN:                                                                  # a sequencer never reports N for a reference gap: random bases
        src[isN] = B[torch.randint(0, 4, (nN,), generator=g, device=dev)]
    del gp, rv, novel, j, sid
    # errors
    rid = torch.repeat_interleave(torch.arange(n_reads, device=dev), torch.from_numpy(rlen).to(dev))
    roff = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev); roff[1:] = torch.cumsum(torch.from_numpy(rlen).to(dev), 0)
    r = torch.rand(tot, generator=g, device=dev)
    ps, pi, pd = (err * m / sum(mix) for m in mix)
    sub = r < ps
    ins = (r >= ps) & (r < ps + pi)
    dele = (r >= ps + pi) & (r < ps + pi + pd)
    rel = torch.arange(tot, device=dev) - roff[rid]
    first = rel == 0
    last = rel == torch.from_numpy(rlen).to(dev)[rid] - 1
    dele &= ~(first | last); ins &= ~last
    code = torch.zeros(256, dtype=torch.int64, device=dev)
    for i, c in enumerate(b"ACGT"):
        code[c] = i
    base = src.clone()
    nsub = int(sub.sum())
    base[sub] = B[(code[src[sub].long()] + torch.randint(1, 4, (nsub,), generator=g, device=dev)) % 4]
    counts = torch.ones(tot, dtype=torch.int64, device=dev)
    counts[ins] = 2; counts[dele] = 0
    out = torch.repeat_interleave(base, counts)
    ends = torch.cumsum(counts, 0)
    nins = int(ins.sum())
    out[ends[ins] - 1] = B[torch.randint(0, 4, (nins,), generator=g, device=dev)]
    ooff = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
    ooff[1:] = ends[roff[1:] - 1]
    del r, sub, ins, dele, rel, first, last, base, counts, ends, src, rid
    # half of the reads come off the sequencer reverse-complemented
    rev = torch.from_numpy(rng.uniform(0, 1, n_reads) < 0.5).to(dev)
    olen = ooff[1:] - ooff[:-1]
    rid2 = torch.repeat_interleave(torch.arange(n_reads, device=dev), olen)
    pos = torch.arange(out.numel(), device=dev)
    rel2 = pos - ooff[rid2]
    srcpos = torch.where(rev[rid2], ooff[rid2] + olen[rid2] - 1 - rel2, pos)
    o2 = out[srcpos]
    o2 = torch.where(rev[rid2], _COMP.to(dev)[o2.long()], o2)
    seq = torch.cat([o2, torch.zeros(64, dtype=torch.uint8, device=dev)])
    return dict(seq=seq, off=ooff, sv=torch.from_numpy(sv).to(dev), start=torch.from_numpy(start).to(dev), rev=rev, chrom=torch.from_numpy(chrom).to(dev))
