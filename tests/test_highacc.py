"""a6 + the high-accuracy SparseDP overload (SparseDP.h:1956): SplitClusters / DecideSplitClustersValue and the sparse DP over the
resulting boxes.  Oracle sanity on CPU (parity unpinned: SplitClusters.h needs Cluster from Clustering.h -> htslib; the SDP engine's
components are pinned in test_sdp.py); HIP vs oracle on the GPU."""
import numpy as np
import pytest

import oracle_lib as O

K = 17


def _random_read(rng, n, read_len, tight):
    """n cluster boxes on a read; tight => many shared / near coordinates so cuts coincide and ties appear in the cut order"""
    qs = []; qe = []; ts = []; te = []; st = []; af = []; moff = [0]; mq = []
    grid = 50 if tight else 1
    for _ in range(n):
        lq = int(rng.integers(20, min(6000, read_len - 10)))
        a = int(rng.integers(0, read_len - lq)) // grid * grid
        b = a + max(4, lq // grid * grid)
        slope = rng.choice([1.0, 1.0, 0.97, 1.04, 0.5, 2.0])
        lt = max(4, int((b - a) * slope) // grid * grid)
        c = int(rng.integers(1000, 200_000)) // grid * grid
        qs.append(a); qe.append(b); ts.append(c); te.append(c + lt); st.append(int(rng.random() < 0.4)); af.append(float(rng.choice([1.0, 2.5, 3.0, 4.0, 5.0, 6.5])))
        m = int(rng.integers(0, 40))
        pos = np.sort(rng.integers(a, max(a + 1, b - K), size=m)) if m else np.zeros(0, np.int64)
        mq.extend(pos.tolist()); moff.append(len(mq))
    return (np.array(qs, np.uint32), np.array(qe, np.uint32), np.array(ts, np.uint32), np.array(te, np.uint32), np.array(st, np.uint8),
            np.array(af, np.float32), np.array(moff, np.int32), np.array(mq, np.uint32))


def test_oracle_split_clusters_sanity():
    # one forward cluster alone: no cut coordinate strictly inside it -> itself; Val = covered read bases, NumofAnchors0 = all matches
    r = O.split_clusters([0], [100], [1000], [1100], [0], [1.0], [0, 3], [0, 10, 50])
    assert r["qs"].tolist() == [0] and r["qe"].tolist() == [100] and r["ts"].tolist() == [1000] and r["te"].tolist() == [1100]
    assert r["cluster_val"].tolist() == [44] and r["val"].tolist() == [44] and r["num"].tolist() == [3]
    # two overlapping forward clusters cut each other on q
    r = O.split_clusters([0, 50], [100, 150], [1000, 2000], [1100, 2100], [0, 0], [1.0, 1.0], [0, 2, 4], [10, 60, 70, 120])
    assert list(zip(r["qs"].tolist(), r["qe"].tolist(), r["ts"].tolist(), r["te"].tolist())) == [
        (0, 50, 1000, 1050), (50, 100, 1050, 1100), (50, 100, 2000, 2050), (100, 150, 2050, 2100)]
    assert r["coarse"].tolist() == [0, 0, 1, 1] and r["num"].tolist() == [1, 1, 1, 1]
    assert r["cluster_val"].tolist() == [34, 34] and r["val"].tolist() == [17, 17, 17, 17]
    # a reverse cluster: pieces run down the anti-diagonal
    r = O.split_clusters([0, 40], [100, 60], [1000, 5000], [1100, 5020], [1, 0], [1.0, 1.0], [0, 0, 0], [])
    got = list(zip(r["qs"].tolist(), r["qe"].tolist(), r["ts"].tolist(), r["te"].tolist(), r["strand"].tolist()))
    assert got[:3] == [(0, 40, 1060, 1100, 1), (40, 60, 1040, 1060, 1), (60, 100, 1000, 1040, 1)]
    # contig reads keep frequent clusters whole, and list them first
    r = O.split_clusters([0, 50], [100, 150], [1000, 2000], [1100, 2100], [0, 0], [2.0, 6.5], [0, 0, 0], [], contig=True)
    assert r["cluster_split"].tolist() == [1, 0] and r["coarse"].tolist()[0] == 1 and (r["qs"][0], r["qe"][0]) == (50, 150)


def test_oracle_sdp_boxes_sanity():
    # three collinear forward boxes chain into one; the value is the sum of Val*rate minus nothing (diagonal gaps <= 2 are free)
    qs = [0, 100, 200]; qe = [100, 200, 300]; ts = [1000, 1100, 1200]; te = [1100, 1200, 1300]
    r = O.sdp_chain_boxes(qs, qe, ts, te, [0, 0, 0], [50, 60, 70], [5, 6, 7], O.sdp_opts(10000, rate=2.0))
    assert r["status"] == 1 and r["chains"][0]["frags"].tolist() == [2, 1, 0] and r["chains"][0]["num_anchors"] == 18
    assert r["chains"][0]["value"] == 2.0 * 180 and r["chains"][0]["box"].tolist() == [0, 300, 1000, 1300]


def _hip_split(ctx, reads, contig):
    import torch
    from lra_amd import chain
    dev = ctx.device
    coff = np.concatenate([[0], np.cumsum([len(r[0]) for r in reads])]).astype(np.int64)
    cat = lambda i, dt: np.concatenate([r[i] for r in reads]).astype(dt) if reads else np.zeros(0, dt)
    moff = [0]
    for r in reads:
        base = moff[-1]
        moff.extend((base + r[6][1:].astype(np.int64)).tolist())
    tt = lambda a: torch.tensor(a, device=dev)
    pad = lambda a: a if len(a) else np.zeros(1, a.dtype)
    d = dict(coff=tt(coff), qs=tt(pad(cat(0, np.int64)).astype(np.int32)), qe=tt(pad(cat(1, np.int64)).astype(np.int32)),
             ts=tt(pad(cat(2, np.int64)).astype(np.int32)), te=tt(pad(cat(3, np.int64)).astype(np.int32)), st=tt(pad(cat(4, np.int32))),
             af=tt(pad(cat(5, np.float32))), moff=tt(np.array(moff, np.int64)), mq=tt(pad(cat(7, np.int64)).astype(np.int32)))
    res = chain.split_clusters_batch(ctx, len(reads), d["coff"], d["qs"], d["qe"], d["ts"], d["te"], d["st"], d["af"], d["moff"], d["mq"], contig=contig, K=K)
    return res, chain.fetch_split_clusters(ctx, res), coff, d


@pytest.mark.gpu
@pytest.mark.parametrize("contig", [False, True])
def test_hip_split_clusters_and_boxes_sdp_oracle(ctx, contig):
    import torch
    from lra_amd import chain
    rng = np.random.default_rng(77 + int(contig))
    reads = []; lens = []
    for i in range(160):
        L = int(rng.integers(3000, 40000))
        n = int(rng.choice([0, 1, 2, 5, 12, 30, 70])) if i % 7 else 0
        reads.append(_random_read(rng, n, L, tight=(i % 3 == 0)) if n else tuple(np.zeros(0, dt) for dt in (np.uint32,) * 4 + (np.uint8, np.float32)) + (np.zeros(1, np.int32), np.zeros(0, np.uint32)))
        lens.append(L)
    res, out, coff, d = _hip_split(ctx, reads, contig)
    n_pieces = 0
    exp_all = []
    for r, rd in enumerate(reads):
        exp = O.split_clusters(*rd, contig=contig, K=K)
        exp_all.append(exp)
        a, b = int(out["split_off"][r]), int(out["split_off"][r + 1])
        assert b - a == len(exp["qs"]), (r, b - a, len(exp["qs"]))
        for k in ("qs", "qe", "ts", "te", "strand", "coarse", "val"):
            assert np.array_equal(out[k][a:b].astype(np.int64), exp[k].astype(np.int64)), (r, k)
        assert np.array_equal(out["num_anchors"][a:b], exp["num"]), r
        assert np.all(out["read"][a:b] == r)
        c0, c1 = int(coff[r]), int(coff[r + 1])
        assert np.array_equal(out["cluster_val"][c0:c1], exp["cluster_val"]) and np.array_equal(out["cluster_split"][c0:c1], exp["cluster_split"]), r
        n_pieces += b - a
    assert n_pieces > 3000
    # the sparse DP over the boxes, straight from the device arrays of the split
    roff = torch.tensor(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64), device=ctx.device)
    for kw in (dict(rate=1.0, NumAln=3), dict(rate=0.5, NumAln=1, alnthres=0.3)):
        opts = chain.sdp_opts(globalK=K, **kw)
        cres = chain.sparse_dp_boxes_batch(ctx, len(reads), res.d_split_off, res.d_qs, res.d_qe, res.d_ts, res.d_te, res.d_strand, res.d_val,
                                           res.d_num_anchors, roff, opts)
        co = chain.fetch(ctx, cres)
        na = opts.NumAln
        n_chains = 0
        for r, exp_s in enumerate(exp_all):
            if len(exp_s["qs"]) == 0:
                assert co["n_chains"][r] == 0
                continue
            exp = O.sdp_chain_boxes(exp_s["qs"], exp_s["qe"], exp_s["ts"], exp_s["te"], exp_s["strand"], exp_s["val"], exp_s["num"],
                                    O.sdp_opts(lens[r], globalK=K, **kw))
            if exp["status"] < 0:
                assert co["status"][r] != 0, r
                continue
            assert co["status"][r] == 0, (r, co["status"][r])
            f0, f1 = int(co["frag_off"][r]), int(co["frag_off"][r + 1])
            assert f1 - f0 == len(exp_s["qs"])
            assert np.array_equal(co["frag_val"][f0:f1].view(np.uint32), exp["val"].view(np.uint32)), r
            assert int(co["n_chains"][r]) == len(exp["chains"]), (r, co["n_chains"][r], len(exp["chains"]))
            for c, ch in enumerate(exp["chains"]):
                s = r * na + c
                a = int(co["chain_start"][s]); ln = int(co["chain_len"][s])
                assert ln == len(ch["frags"]), (r, c)
                assert np.array_equal(co["chain_cluster"][a:a + ln], ch["frags"]), (r, c)
                assert np.array_equal(co["chain_link"][a:a + ln - 1], ch["link"]), (r, c)
                assert np.array_equal(co["chain_box"][s], ch["box"]), (r, c)
                assert np.float32(co["chain_value"][s]).view(np.uint32) == np.float32(ch["value"]).view(np.uint32)
                assert int(co["chain_num_anchors"][s]) == ch["num_anchors"]
                n_chains += 1
        assert n_chains > 100


def _same_diag_clusters(rng, n_clusters):
    """extended clusters as LinearExtend leaves them: runs of anchors on one diagonal (some adjacent, some far apart), diagonal jumps,
    overlap flags, both strands; an empty cluster"""
    Q, T, L, OV, ST, off = [], [], [], [], [], [0]
    for c in range(n_clusters):
        strand = int(rng.integers(0, 2))
        n = int(rng.integers(1, 200)) if c != 3 else 0
        q = int(rng.integers(0, 1000)); t = int(rng.integers(5000, 100000))
        for i in range(n):
            ln = int(rng.integers(10, 60))
            Q.append(q); L.append(ln); OV.append(int(rng.random() < 0.08))
            T.append(t if strand == 0 else t - ln)                    # reverse strand: q + t + len constant along a diagonal
            kind = rng.random()
            gap = int(rng.integers(1, 30)) if kind < 0.6 else int(rng.integers(90, 130)) if kind < 0.75 else 0 if kind < 0.8 else int(rng.integers(1, 30))
            q += ln + gap
            t = t + (ln + gap) if strand == 0 else t - (ln + gap)
            if kind >= 0.8:
                t += int(rng.integers(-3, 4))                          # leave the diagonal
        ST.append(strand); off.append(len(Q))
    return (np.array(Q, np.uint32), np.array(T, np.int64).astype(np.uint32), np.array(L, np.int32), np.array(OV, np.uint8), np.array(ST, np.int32),
            np.array(off, np.int64))


def test_oracle_merge_same_diag_sanity(oracle):
    # three anchors on one forward diagonal 20 apart, then a jump: two entries; an overlap flag splits; a far anchor (> merge_dist) splits
    q = [0, 40, 80, 200]; t = [1000, 1040, 1080, 1300]; ln = [20, 20, 20, 20]
    assert [x.tolist() for x in oracle.merge_same_diag(q, t, ln, [0, 0, 0, 0], 0)] == [[0, 3], [3, 4]]
    assert [x.tolist() for x in oracle.merge_same_diag(q, t, ln, [0, 1, 0, 0], 0)] == [[0, 1, 2, 3], [1, 2, 3, 4]]
    assert [x.tolist() for x in oracle.merge_same_diag([0, 200], [1000, 1200], [20, 20], [0, 0], 0)] == [[0, 1], [1, 2]]
    assert [x.tolist() for x in oracle.merge_same_diag([0, 40], [1040, 1000], [20, 20], [0, 0], 1)] == [[0], [2]]      # q + t + len equal
    assert oracle.merge_same_diag([], [], [], [], 0) is None


@pytest.mark.gpu
def test_hip_merge_same_diag_oracle(ctx, oracle):
    import torch
    from lra_amd import chain
    rng = np.random.default_rng(12)
    Q, T, L, OV, ST, off = _same_diag_clusters(rng, 300)
    dev = ctx.device
    tt = lambda a: torch.from_numpy(a).to(dev)
    res = chain.merge_same_diag_batch(ctx, tt(off), tt(Q), tt(T), tt(L), tt(OV), tt(ST), 100)
    out = chain.fetch_same_diag(ctx, res)
    n_merged = n_groups = 0
    for c in range(len(ST)):
        a, b = int(off[c]), int(off[c + 1])
        exp = oracle.merge_same_diag(Q[a:b], T[a:b], L[a:b], OV[a:b], int(ST[c]), 100)
        g0, g1 = int(out["group_off"][c]), int(out["group_off"][c + 1])
        if exp is None:
            assert out["status"][c] != 0 and g1 == g0
            continue
        assert out["status"][c] == 0
        assert out["start"][g0:g1].tolist() == exp[0].tolist() and out["end"][g0:g1].tolist() == exp[1].tolist(), c
        n_groups += g1 - g0; n_merged += int(np.sum(exp[1] - exp[0] > 1))
    assert n_groups > 5000 and n_merged > 2000, (n_groups, n_merged)
    # SwitchToOriginalAnchors (LocalRefineAlignment.h:187): chains over those entries back to the original anchors
    chains = []
    for _ in range(200):
        m = int(rng.integers(0, 40))
        cl = rng.integers(0, len(ST), m)
        cl = cl[np.array([off[c + 1] > off[c] for c in cl], bool)] if m else cl
        ent = np.array([int(rng.integers(0, int(out["group_off"][c + 1] - out["group_off"][c]))) for c in cl], np.uint32)
        chains.append((cl.astype(np.int32), ent))
    coff = np.cumsum([0] + [len(c[0]) for c in chains]).astype(np.int64)
    ecl = np.concatenate([c[0] for c in chains]).astype(np.int32); een = np.concatenate([c[1] for c in chains]).astype(np.uint32)
    coarse = rng.integers(0, 50, len(ST)).astype(np.int32)
    r2 = chain.switch_to_original_anchors_batch(ctx, tt(coff), tt(ecl), tt(een), res, tt(coarse))
    o_off = ctx.to_host(r2.d_chain_off, len(chains) + 1, np.uint64); o_a = ctx.to_host(r2.d_anchor, int(r2.n_anchors), np.uint32)
    o_c = ctx.to_host(r2.d_cluster, int(r2.n_anchors), np.int32)
    for i, (cl, ent) in enumerate(chains):
        ea, ec = oracle.switch_to_original_anchors(cl, ent, out["group_off"], out["start"], out["end"], coarse)
        a0, a1 = int(o_off[i]), int(o_off[i + 1])
        assert np.array_equal(o_a[a0:a1], ea) and np.array_equal(o_c[a0:a1], ec), i
    assert int(r2.n_anchors) > 3000


def test_oracle_switchindex_sanity(oracle):
    co = np.arange(200); qs = np.arange(200) * 100; qe = qs + 50
    ch, lk = oracle.switchindex([22, 125, 19, 125, 16, 17, 125, 57, 125], [0] * 8, co, qs, qe)          # the reference's own example (:82)
    assert ch.tolist() == [22, 125] and lk.tolist() == [0]
    ch, lk = oracle.switchindex([5, 5, 6, 6, 7], [1, 0, 1, 0], co, qs, qe)                               # links inside one cluster go
    assert ch.tolist() == [5, 6, 7] and lk.tolist() == [0, 0]
    qe2 = qe.copy(); qe2[3] = qs[4] + 60                                                                 # cluster 4 inside cluster 3 on the read
    qs2 = qs.copy(); qs2[4] = qs2[3] + 10
    ch, lk = oracle.switchindex([3, 4, 9], [1, 0], co, qs2, qe2)
    assert ch.tolist() == [3, 9] and lk.tolist() == [0]


@pytest.mark.gpu
def test_hip_switchindex_oracle(ctx, oracle):
    import torch
    from lra_amd import chain
    rng = np.random.default_rng(21)
    n_reads = 60
    ch_all, lk_all, off, nlk, spb, clb, coarse_all, qs_all, qe_all = [], [], [0], [], [], [], [], [], []
    per_chain = []
    for r in range(n_reads):
        ncl = int(rng.integers(2, 12)); nsp = int(rng.integers(ncl, 5 * ncl))
        coarse = np.sort(rng.integers(0, ncl, nsp)).astype(np.int32)                      # split clusters in cluster order
        s = np.sort(rng.integers(0, 20000, ncl)); e = s + rng.integers(50, 3000, ncl)
        for j in range(1, ncl):
            if rng.random() < 0.3:
                s[j] = s[j - 1] + 5; e[j] = e[j - 1] - 1                                  # nested on the read
        sp0, cl0 = len(coarse_all), len(qs_all)
        coarse_all.extend(coarse.tolist()); qs_all.extend(s.tolist()); qe_all.extend(e.tolist())
        for h in range(int(rng.integers(1, 4))):
            n = int(rng.integers(1, 40))
            ch = rng.integers(0, nsp, n)
            if rng.random() < 0.5:
                ch = np.sort(ch)[::-1]                                                     # colinear-looking chain: adjacent repeats
            has_link = rng.random() < 0.85
            lk = rng.integers(0, 2, n - 1) if has_link else np.zeros(0, np.int64)
            ch_all.extend(ch.tolist()); lk_all.extend(lk.tolist() + [0] * (n - len(lk))); off.append(len(ch_all)); nlk.append(len(lk))
            spb.append(sp0); clb.append(cl0)
            per_chain.append((ch.copy(), lk.copy(), coarse, s, e))
    dev = ctx.device
    T = lambda a, dt: torch.from_numpy(np.asarray(a, dt)).to(dev)
    res = chain.switchindex_batch(ctx, T(off, np.int64), T(ch_all, np.uint32), T(lk_all, np.uint8), T(nlk, np.uint32), T(spb, np.int64), T(clb, np.int64),
                                  T(coarse_all, np.int32), T(qs_all, np.uint32), T(qe_all, np.uint32))
    out = chain.fetch_switchindex(ctx, res, len(ch_all))
    n_short = n_ub = 0
    for c, (ch, lk, coarse, s, e) in enumerate(per_chain):
        exp = oracle.switchindex(ch, lk, coarse, s, e)
        if exp is None:
            assert out["status"][c] != 0, c
            n_ub += 1
            continue
        assert out["status"][c] == 0, c
        b = off[c]
        assert out["ch"][b:b + int(out["n"][c])].tolist() == exp[0].tolist(), c
        assert out["link"][b:b + int(out["n_link"][c])].tolist() == exp[1].tolist(), c
        n_short += len(exp[0]) < len(ch)
    assert n_short > 40 and len(per_chain) > 100, (n_short, n_ub, len(per_chain))


@pytest.mark.gpu
def test_hip_refine_btwn_space_oracle(ctx, oracle):
    """RefineBtwnSpace (ClusterRefine.h:331): spaces at the true locus (dense on the read's strand), at the locus of the reverse complement (dense
    only on the other strand -> RevBtwnCluster), at unrelated loci (nothing / sparse either way), two-block spaces, short (< 1 kb: the
    AffineOneGapAlign branch of RefineSpace) and long ones, both presets of refineSpaceDiag"""
    import ctypes as C
    import torch
    from lra_amd import seed, synth
    rng = np.random.default_rng(33)
    genome = synth.make_genome(300_000, seed=14, repeat_frac=0.1, n_families=2)
    CH = [0, 120_000, 300_000]
    reads, locus = [], []
    for i in range(6):
        a = int(rng.integers(5_000, 100_000)) if i % 2 == 0 else int(rng.integers(130_000, 280_000))
        rev = bool(i & 1)
        reads.append(synth.simulate_read(rng, genome[a:a + 9001], 9000, 0.08, (30, 35, 35), rev)[0]); locus.append((a, rev))
    batch = seed.ReadBatch(ctx, [r.tobytes() for r in reads])
    tot = int(batch.off[-1])
    rcb = seed.create_rc(ctx, batch)
    both = torch.cat([batch.seq[:tot], rcb[:tot], torch.zeros(64, dtype=torch.uint8, device=ctx.device)])
    gdev = torch.from_numpy(np.concatenate([genome, np.zeros(64, np.uint8)])).to(ctx.device)
    P = []
    for k in range(240):
        r = int(rng.integers(0, len(reads))); a, rev = locus[r]; L = len(reads[r])
        ci = 0 if a < CH[1] else 1
        span = int(rng.integers(300, 900)) if k % 3 == 0 else int(rng.integers(1100, 4000))
        q0 = int(rng.integers(0, L - span))
        st = int(rev) if k % 4 else 1 - int(rev)                          # mostly the strand the read lies on; sometimes the wrong one
        # the space in the orientation the caller passes it: forward read coordinates for st 0, and the flip of that for st 1 is done inside
        qs, qe = q0, q0 + span
        # genome interval under that read interval (read drawn forward from a, or reverse-complemented)
        g0 = a + (q0 if not rev else L - (q0 + span))
        kind = k % 5
        t0 = g0 - CH[ci] + int(rng.integers(-20, 20)) if kind else int(rng.integers(1000, CH[ci + 1] - CH[ci] - 5000))
        t0 = max(0, min(t0, CH[ci + 1] - CH[ci] - span - 64))
        tlen = span + int(rng.integers(-40, 40))
        P.append((r, ci, qs, qe, t0, t0 + tlen, st, int(k % 7 == 0)))
    dev = ctx.device
    T = lambda a, dt: torch.from_numpy(np.asarray(a, dt)).to(dev)
    cols = list(zip(*P))
    for read_type, sparse in ((0, 0.005), (3, 0.05)):
        res = (C.c_uint64 * 3)()
        class R(C.Structure):
            _fields_ = [("n", C.c_uint64), ("n_pairs", C.c_uint64), ("n_rev", C.c_uint64)] + [(x, C.c_void_p) for x in ("off", "q", "t", "dec", "eff", "reff")]
        out = R()
        cp = (C.c_uint64 * len(CH))(*CH)
        args = [T(cols[2], np.uint32), T(cols[3], np.uint32), T(cols[4], np.uint32), T(cols[5], np.uint32), T(cols[6], np.int32), T(cols[7], np.uint8), T(cols[0], np.uint32),
                T(cols[1], np.int32)]
        ctx.check(ctx.lib.lra_refine_btwn_space_batch(ctx.h, len(P), *[x.data_ptr() for x in args], None, None, batch.off.data_ptr(), both.data_ptr(), C.c_uint64(tot),
                                                      gdev.data_ptr(), cp, len(CH) - 1, 10, 5, read_type, C.c_float(sparse), 4, -1, -2, 15, C.byref(out)))
        n = len(P)
        off = ctx.to_host(out.off, n + 1, np.uint64); pq = ctx.to_host(out.q, int(out.n_pairs), np.uint32); pt = ctx.to_host(out.t, int(out.n_pairs), np.uint32)
        dec = ctx.to_host(out.dec, n, np.int32); eff = ctx.to_host(out.eff, n, np.float32); reff = ctx.to_host(out.reff, n, np.float32)
        seen = set()
        for i, (r, ci, qs, qe, ts, te, st, two) in enumerate(P):
            fwd = reads[r].tobytes(); rc = synth.revcomp(reads[r]).tobytes()
            chrom = genome[CH[ci]:CH[ci + 1]].tobytes() + b"\0" * 64
            ed, eq, et, eeff, ereff = oracle.refine_btwn_space(fwd, rc, chrom, qe, qs, te, ts, st, two, K=10, W=5, read_type=read_type, anchorstoosparse=sparse)
            assert dec[i] == ed, (i, dec[i], ed)
            assert eff[i].view(np.uint32) == eeff.view(np.uint32) and reff[i].view(np.uint32) == ereff.view(np.uint32), i
            a0, a1 = int(off[i]), int(off[i + 1])
            assert np.array_equal(pq[a0:a1], eq) and np.array_equal(pt[a0:a1], et), (i, ed, a1 - a0, len(eq))
            seen.add(int(ed))
        assert seen == {0, 1, 2, 3}, seen
