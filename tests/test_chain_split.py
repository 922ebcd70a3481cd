"""a9 (low-accuracy path): RemoveSpuriousJump + SPLITChain + MergeSplitchainINS + RemoveSpuriousSplitChain.
Oracle sanity on CPU (parity unpinned: Chain.h / Mapping_ultility.h need htslib); HIP vs oracle on the GPU."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

CHROM = [0, 3_000_000, 6_000_000]


def _random_chain(rng, n, kind):
    """anchors in trace-back order (q descending)"""
    q = 200_000; t = int(rng.integers(1_000_000, 2_000_000)); strand = int(rng.random() < 0.3)
    t_home = t
    Q = []; T = []; L = []; S = []; CL = []; LK = []
    cl = 0
    for i in range(n):
        ln = int(rng.choice([17, 20, 30, 45, 60, 120]))
        Q.append(q); T.append(t); L.append(ln); S.append(strand); CL.append(cl)
        if i < n - 1: LK.append(int(rng.random() < 0.1))
        u = rng.random()
        q -= ln + int(rng.integers(0, 150))
        step = ln + int(rng.integers(0, 150))
        if kind >= 1 and u < 0.08:                       # diagonal jump > 100 (an SV), often paired with the opposite one right after
            step += int(rng.choice([-400, -150, 150, 400, 900]))
        if kind >= 2 and u > 0.93:                       # far jump and maybe back (translocation-like), or a strand flip
            v = rng.random()
            if v < 0.4: t = t + int(rng.choice([-1, 1])) * int(rng.integers(60_000, 400_000)); cl += 1
            elif v < 0.7: t = t_home - (200_000 - q) + int(rng.integers(-800, 800)); cl += 1
            else: strand ^= 1; cl += 1
        if kind >= 3 and u > 0.985: t = 3_000_000 - int(rng.integers(0, 60)); cl += 1   # straddle the chromosome boundary
        if kind >= 1 and 0.5 < u < 0.53: q -= int(rng.integers(1000, 5000)); step += int(rng.integers(1000, 5000))   # long collinear gap
        t = t - step if strand == 0 else t + step
        t = max(1, min(t, 5_999_000))
        if rng.random() < 0.2: cl += 1
    return (np.array(Q, np.uint32), np.array(T, np.uint32), np.array(L, np.int32), np.array(S, np.uint8), np.array(CL, np.int32), np.array(LK, np.uint8))


def test_oracle_split_sanity():
    # one forward chain on a diagonal, a 80 kb jump, five more anchors: a 'T' piece and an 'N' piece, forward pieces listed first-to-last
    q = []; t = []
    qq = 5000; tt = 1_000_000
    for i in range(10): q.append(qq); t.append(tt); qq -= 100; tt -= 100
    tt -= 80000
    for i in range(5): q.append(qq); t.append(tt); qq -= 100; tt -= 100
    n = len(q)
    r = O.split_chain(q, t, [30] * n, [0] * n, [0] * 10 + [1] * 5, [0] * (n - 1), [0, 50_000_000])
    assert r["n_kept"] == n and [s["type"] for s in r["splits"]] == ["T", "N"]
    assert r["splits"][0]["idx"].tolist() == list(range(9, -1, -1)) and r["splits"][1]["clusters"].tolist() == [1]
    assert r["split_link"].tolist() == [0]
    # paired opposite jumps one anchor apart: the short anchor between them goes (RemoveSpuriousJump)
    q = [1000, 900, 800, 700, 600]; t = [5000, 4900, 4500, 4700, 4600]
    r = O.split_chain(q, t, [20] * 5, [0] * 5, [0] * 5, [0] * 4, [0, 50_000_000])
    assert r["keep"].tolist() == [1, 1, 0, 1, 1]


@pytest.mark.gpu
def test_hip_split_chains_oracle(ctx):
    import torch
    from lra_amd import chain
    rng = np.random.default_rng(3)
    num_aln = 2
    chains = []
    for k in range(400):
        chains.append(_random_chain(rng, int(rng.integers(1, 120)), kind=k % 4))
    n_reads = len(chains) // num_aln
    # fake lra_chain_result: slot s = chains[s]
    start = [0]
    for c in chains: start.append(start[-1] + len(c[0]))
    cat = lambda j, dt: np.concatenate([c[j] for c in chains]).astype(dt)
    link = np.concatenate([np.concatenate([c[5], [0]]) for c in chains]).astype(np.uint8)
    dev = ctx.device
    tt = lambda a: torch.tensor(a, device=dev)
    bufs = dict(n_chains=tt(np.full(n_reads, num_aln, np.int32)), start=tt(np.array(start[:-1], np.int64)), length=tt(np.array([len(c[0]) for c in chains], np.int32)),
                q=tt(cat(0, np.int64)).to(torch.int32), t=tt(cat(1, np.int64)).to(torch.int32), ln=tt(cat(2, np.int32)), st=tt(cat(3, np.uint8)), cl=tt(cat(4, np.int32)), lk=tt(link))
    res = chain.ChainResult()
    res.n_reads = n_reads; res.num_aln = num_aln; res.n_frags = start[-1]
    res.d_n_chains = bufs["n_chains"].data_ptr(); res.d_chain_start = bufs["start"].data_ptr(); res.d_chain_len = bufs["length"].data_ptr()
    res.d_chain_q = bufs["q"].data_ptr(); res.d_chain_t = bufs["t"].data_ptr(); res.d_chain_alen = bufs["ln"].data_ptr(); res.d_chain_strand = bufs["st"].data_ptr()
    res.d_chain_cluster = bufs["cl"].data_ptr(); res.d_chain_link = bufs["lk"].data_ptr()
    for splitdist, bypass in ((50000, 1), (50000, 0)):
        sres = chain.split_chains_batch(ctx, res, CHROM, splitdist, bypass)
        out = chain.fetch_split(ctx, sres)
        n_t = n_i = n_merge = 0
        for s, c in enumerate(chains):
            exp = O.split_chain(c[0], c[1], c[2], c[3], c[4], c[5], CHROM, splitdist, bypass)
            b = start[s]
            if exp is None:
                assert out["status"][s] != 0
                continue
            assert out["status"][s] == 0, s
            n = len(c[0])
            assert out["keep"][b:b + n].tolist() == exp["keep"].tolist(), s
            assert out["n_kept"][s] == exp["n_kept"]
            assert out["link"][b:b + max(exp["n_kept"] - 1, 0)].tolist() == exp["link"].tolist(), s
            assert out["n_split"][s] == len(exp["splits"]), (s, out["n_split"][s], len(exp["splits"]))
            for k, e in enumerate(exp["splits"]):
                x = b + k
                a0, m = b + int(out["sp_beg"][x]), int(out["sp_len"][x])
                assert out["sp_idx"][a0:a0 + m].tolist() == e["idx"].tolist(), (s, k)
                assert out["sp_link"][a0:a0 + m - 1].tolist() == e["link"].tolist(), (s, k)
                assert chr(out["sp_type"][x]) == e["type"] and out["sp_strand"][x] == e["strand"] and out["sp_chrom"][x] == e["chrom"], (s, k)
                assert out["sp_box"][x].tolist() == e["box"].tolist(), (s, k)
                c0, cm = b + int(out["ci_beg"][x]), int(out["ci_len"][x])
                assert out["ci_idx"][c0:c0 + cm].tolist() == e["clusters"].tolist(), (s, k)
                n_t += e["type"] == "T"; n_i += e["type"] == "I"
                n_merge += int(np.any(np.abs(np.diff(e["idx"].astype(np.int64))) > 1))
            assert out["n_split_link"][s] == len(exp["split_link"])
            assert out["split_link"][b:b + len(exp["split_link"])].tolist() == exp["split_link"].tolist(), s
        assert n_t > 10 and n_i > 10 and n_merge > 0, (n_t, n_i, n_merge)


def _indel_chain(rng, n):
    """a chain with small / large paired diagonal jumps, short and long anchors, the odd strand flip and far end anchors"""
    q = 300_000; t = int(rng.integers(1_000_000, 2_000_000)); strand = int(rng.random() < 0.3)
    Q = []; T = []; L = []; S = []; LK = []
    pend = 0
    for i in range(n):
        ln = int(rng.choice([17, 20, 30, 45, 60, 99, 100, 150]))
        Q.append(q); T.append(t); L.append(ln); S.append(strand)
        if i < n - 1: LK.append(int(rng.random() < 0.15))
        q -= ln + int(rng.integers(0, 120))
        step = ln + int(rng.integers(0, 120))
        u = rng.random()
        if pend and rng.random() < 0.7: step -= pend + int(rng.integers(-25, 25)); pend = 0      # the opposite jump shortly after
        elif u < 0.12: pend = int(rng.choice([8, 20, 45, 60, 90, 320, 600, 900])) * int(rng.choice([-1, 1])); step += pend
        if u > 0.985: strand ^= 1
        if i in (0, 1, n - 3, n - 2) and rng.random() < 0.3: step += int(rng.integers(20_000, 90_000))   # far-away chain ends (refineEnds)
        t = t - step if strand == 0 else t + step
        t = max(1000, min(t, 5_000_000))
    return (np.array(Q, np.uint32), np.array(T, np.uint32), np.array(L, np.int32), np.array(S, np.uint8), np.array(LK, np.uint8))


def test_oracle_filter_chain_sanity():
    # +400 / -400 diagonal jumps two anchors apart: RemovePairedIndels drops the short anchors in between; no link -> no link out
    q = [1000, 900, 800, 700, 600, 500]; t = [5000, 4900, 5200, 5100, 4600, 4500]
    keep, lk = O.filter_chain(q, t, [20] * 6, [0] * 6, None, [3])
    assert keep.tolist() == [1, 1, 0, 0, 1, 1] and len(lk) == 0
    keep, lk = O.filter_chain(q, t, [20, 20, 150, 20, 20, 20], [0] * 6, [0, 1, 0, 1, 0], [3])
    assert keep.tolist() == [1, 1, 1, 0, 1, 1] and lk.tolist() == [0, 1, 1, 0]


@pytest.mark.gpu
def test_hip_filter_chains_oracle(ctx):
    import torch
    from lra_amd import chain
    rng = np.random.default_rng(12)
    chains = [_indel_chain(rng, int(rng.integers(1, 90))) for _ in range(300)]
    off = np.cumsum([0] + [len(c[0]) for c in chains])
    cat = lambda j, dt: np.concatenate([c[j] for c in chains]).astype(dt)
    link = np.concatenate([np.concatenate([c[4], [0]]) for c in chains]).astype(np.uint8)
    dev = ctx.device
    tt = lambda a: torch.tensor(a, device=dev)
    d = dict(off=tt(off.astype(np.int64)), q=tt(cat(0, np.int64)).to(torch.int32), t=tt(cat(1, np.int64)).to(torch.int32), ln=tt(cat(2, np.int32)),
             st=tt(cat(3, np.uint8)), lk=tt(link))
    removed = 0
    # FinalChain's qEnd (merged entries: the length added to the LAST anchor's read position): q + len + an extra for some anchors
    qend = cat(0, np.int64) + cat(2, np.int64) + np.where(rng.random(int(off[-1])) < 0.3, rng.integers(0, 300, int(off[-1])), 0)
    d["qe"] = tt(qend).to(torch.int32)
    for ops, with_link, ex in (([2, 4], True, False), ([1, 2, 4], False, False), ([1, 3, 4], True, False), ([8], True, False), ([4, 8, 1], True, False), ([2], True, False),
                               ([5], False, False), ([5, 4], True, False), ([1, 3, 4], False, True), ([1, 2, 4], True, True)):
        res = chain.filter_chains_batch(ctx, len(chains), d["off"], int(off[-1]), d["q"], d["t"], d["ln"], d["st"], d["lk"] if with_link else None, ops,
                                        qend=d["qe"] if ex else None)
        out = chain.fetch_filter(ctx, res)
        for i, c in enumerate(chains):
            keep, lk = O.filter_chain(c[0], c[1], c[2], c[3], c[4] if with_link else None, ops, qend=qend[int(off[i]):int(off[i + 1])] if ex else None)
            a, b = int(off[i]), int(off[i + 1])
            assert out["keep"][a:b].tolist() == keep.tolist(), (ops, i)
            assert out["n_kept"][i] == keep.sum()
            assert out["n_link"][i] == len(lk), (ops, i, out["n_link"][i], len(lk))
            assert out["link"][a:a + len(lk)].tolist() == lk.tolist(), (ops, i)
            removed += int((keep == 0).sum())
    assert removed > 300


@pytest.mark.gpu
def test_hip_trim_anchor_pairs_oracle(ctx):
    """TrimOverlappedAnchors(GenomePairs&, lengths) LinearExtend.h:722: lists of anchors, some long ones overlapping by up to 30 on either axis,
    equal (q, t) keys with different lengths included (the tie the exact std::sort decides)"""
    import ctypes as C
    import torch
    rng = np.random.default_rng(4)
    lists = []
    for _ in range(200):
        n = int(rng.integers(0, 60))
        q = 100; t = 5000; Q = []; T = []; L = []
        for i in range(n):
            ln = int(rng.choice([12, 30, 49, 50, 51, 80, 140]))
            Q.append(q); T.append(t); L.append(ln)
            if rng.random() < 0.1: Q.append(q); T.append(t); L.append(int(rng.choice([50, 75])))          # same corner, another length
            q += ln + int(rng.choice([-25, -10, -1, 0, 3, 40])); t += ln + int(rng.choice([-30, -12, 0, 5, 60]))
        p = rng.permutation(len(Q))
        lists.append((np.array(Q, np.uint32)[p], np.array(T, np.uint32)[p], np.array(L, np.int32)[p]))
    off = np.cumsum([0] + [len(l[0]) for l in lists]).astype(np.int64)
    dev = ctx.device
    cat = lambda j, dt: torch.from_numpy(np.concatenate([l[j] for l in lists]).astype(dt)).to(dev)
    dq = cat(0, np.int64).to(torch.int32); dt_ = cat(1, np.int64).to(torch.int32); dl = cat(2, np.int32)
    doff = torch.from_numpy(off).to(dev)
    ctx.check(ctx.lib.lra_trim_anchor_pairs_batch(ctx.h, C.c_uint64(len(lists)), C.c_void_p(doff.data_ptr()), C.c_uint64(int(off[-1])), C.c_void_p(dq.data_ptr()),
                                                  C.c_void_p(dt_.data_ptr()), C.c_void_p(dl.data_ptr())))
    got = dl.cpu().numpy()
    L_ = O.lib()
    changed = 0
    for i, (Q, T, Ln) in enumerate(lists):
        e = Ln.copy()
        Qc = np.ascontiguousarray(Q); Tc = np.ascontiguousarray(T)
        L_.oracle_trim_anchor_pairs(C.c_int(len(Q)), Qc.ctypes.data_as(C.POINTER(C.c_uint32)), Tc.ctypes.data_as(C.POINTER(C.c_uint32)), e.ctypes.data_as(C.POINTER(C.c_int)))
        assert np.array_equal(got[off[i]:off[i + 1]], e), i
        changed += int((e != Ln).sum())
    assert changed > 100


@pytest.mark.gpu
def test_hip_trim_overlapped_anchors_clusters(ctx, oracle):
    """the cluster version of TrimOverlappedAnchors (LinearExtend.h:574): both strands, long anchors overlapping by 1..30 on the read and / or
    the genome, short anchors in between, ties in the sort key"""
    import torch
    rng = np.random.default_rng(17)
    Q, T, L, off, ST = [], [], [], [0], []
    for c in range(400):
        strand = int(rng.integers(0, 2)); n = int(rng.integers(0, 60))
        q = int(rng.integers(0, 500)); t = int(rng.integers(1000, 50000))
        qs, ts, ls = [], [], []
        for i in range(n):
            ln = int(rng.integers(10, 120))
            qs.append(q); ts.append(t); ls.append(ln)
            step_q = ln + int(rng.integers(-30, 40)); step_t = ln + int(rng.integers(-30, 40))
            q += max(step_q, 1); t += max(step_t, 1)
        if strand:                                                        # reverse clusters run right to left on the read
            top = q + 200
            qs = [top - a - b for a, b in zip(qs, ls)]
        Q += qs; T += ts; L += ls; ST.append(strand); off.append(len(Q))
    dev = ctx.device
    dq = torch.tensor(np.array(Q, np.uint32).astype(np.int64), dtype=torch.int64, device=dev).to(torch.int32)
    dq = torch.from_numpy(np.array(Q, np.uint32)).to(dev); dt = torch.from_numpy(np.array(T, np.uint32)).to(dev); dl = torch.from_numpy(np.array(L, np.int32)).to(dev)
    doff = torch.from_numpy(np.array(off, np.int64)).to(dev); dst = torch.from_numpy(np.array(ST, np.int32)).to(dev)
    ctx.check(ctx.lib.lra_trim_overlapped_anchors_batch(ctx.h, len(ST), doff.data_ptr(), len(Q), dst.data_ptr(), dq.data_ptr(), dt.data_ptr(), dl.data_ptr()))
    gq = dq.cpu().numpy(); gl = dl.cpu().numpy()
    n_trim = 0
    for c in range(len(ST)):
        a, b = off[c], off[c + 1]
        eq, el = oracle.trim_overlapped_anchors(Q[a:b], T[a:b], L[a:b], ST[c])
        assert np.array_equal(gq[a:b], eq) and np.array_equal(gl[a:b], el), c
        n_trim += int(np.sum(el != np.array(L[a:b], np.int32)))
    assert n_trim > 300, n_trim
