"""The high-accuracy path (MapRead_highacc, Map_highacc.h:37-798; -CCS / -CONTIG): the leaves that were missing after round 1 and the path end to end.
a5: MatchesToFineClusters (Clustering.h:1555; SplitRoughClustersWithGaps :1358, StoreFineClusters :892).
Oracle restatements are PARITY UNPINNED (Clustering.h & co need htslib headers); CPU tests check their properties, GPU tests compare HIP with them."""
import re

import numpy as np
import pytest

import oracle_lib as O
from lra_amd import synth

CLEAN = {"CCS": dict(globalK=17, cleanMaxDiag=150, minDiagCluster=10, bypassClustering=0, cleanClustersize=100, SecondCleanMinDiagCluster=30, SecondCleanMaxDiag=100,
                     punish_anchorfreq=10, anchorPerlength=10),
         "CONTIG": dict(globalK=19, cleanMaxDiag=150, minDiagCluster=30, bypassClustering=0, cleanClustersize=100, SecondCleanMinDiagCluster=30, SecondCleanMaxDiag=100,
                        punish_anchorfreq=10, anchorPerlength=10)}
FINE = {"CCS": dict(globalK=17, RoughClustermaxGap=500, maxDiag=500, maxGap=400, minClusterSize=10, minUniqueStretchNum=1, minUniqueStretchDist=50),
        "CONTIG": dict(globalK=19, RoughClustermaxGap=500, maxDiag=100, maxGap=500, minClusterSize=10, minUniqueStretchNum=1, minUniqueStretchDist=50),
        "LOOSE": dict(globalK=17, RoughClustermaxGap=200, maxDiag=60, maxGap=150, minClusterSize=2, minUniqueStretchNum=1, minUniqueStretchDist=20)}


def _genome_with_repeats(seed, n=600_000):
    """random sequence + interspersed repeat families + tandem arrays (so that clusters with anchorfreq > 1 and non-unique stretches appear) + a
    segmental duplication"""
    g = synth.make_genome(n, seed=seed, repeat_frac=0.25, n_families=3).copy()
    rng = np.random.default_rng(seed)
    unit = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 400)]
    for a0 in (150_000, 420_000):
        arr = np.tile(unit, 20)
        mut = rng.random(len(arr)) < 0.03
        arr[mut] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(mut.sum()))]
        g[a0:a0 + len(arr)] = arr
    g[500_000:512_000] = g[80_000:92_000]
    return g


def _reads(genome, rng, n=14, err=0.01):
    reads, _ = synth.simulate_reads(genome, n, 9000, 2500, err, (34, 33, 33), seed=int(rng.integers(1 << 30)))
    sim = lambda a, ln, rev=False: synth.simulate_read(rng, genome[a:a + ln + 1], ln, err, (34, 33, 33), rev)[0]
    reads.append(sim(146_000, 14_000))                                       # across a tandem array
    reads.append(sim(418_000, 12_000, True))
    reads.append(sim(78_000, 16_000))                                        # inside the duplication
    reads.append(np.concatenate([sim(30_000, 5000), sim(300_000, 5000, True)]))   # chimera, second half reversed
    reads.append(np.concatenate([sim(200_000, 4000), sim(204_000 + 3000, 4000)]))  # 3 kb deletion
    reads.append(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 3000)].copy())
    return reads


def _seed_cpu(read, ik, ip, genome_bytes, K, W, max_freq):
    """tier-1 matches of one read with the forward-strand ones first: (qpos, tpos, qkey, n_forward)"""
    keys, pos = O.store_minimizers(read, K, W)
    sk, sp = O.sort_minimizers(keys, pos)
    qi, ti = O.compare_lists(sk, sp, ik, ip, max_freq)
    st = O.separate_strand(read, genome_bytes, K, sp[qi], ip[ti])
    f = st == 0
    return (np.concatenate([sp[qi][f], sp[qi][~f]]), np.concatenate([ip[ti][f], ip[ti][~f]]), np.concatenate([sk[qi][f], sk[qi][~f]]), int(f.sum()))


def test_oracle_fine_clusters_sanity(oracle):
    g = _genome_with_repeats(7)
    CH = [0, 250_000, len(g)]
    ik, ip, _ = O.store_index(g.tobytes(), CH, 17, 10, 150, 15, 1)
    rng = np.random.default_rng(1)
    reads = _reads(g, rng, n=6)
    co = O.CleanOpts(**CLEAN["CCS"]); fo = O.FineOpts(**FINE["CCS"])
    gb = g.tobytes() + b"\0" * 64
    n_cl = 0
    for rd in reads[:-1]:
        q, t, k, nf = _seed_cpu(rd.tobytes(), ik, ip, gb, 17, 10, 150)
        fc, st = O.matches_to_fine_clusters(q, t, k, nf, co, fo, CH)
        assert st == 0
        raw = set(zip(q.tolist(), t.tolist()))
        for c in range(len(fc["strand"])):
            a, b = int(fc["off"][c]), int(fc["off"][c + 1])
            cq, ct = fc["q"][a:b], fc["t"][a:b]
            assert b - a >= 10 and set(zip(cq.tolist(), ct.tolist())) <= raw                # every match is a tier-1 match; clusters reach minClusterSize
            assert np.all(np.diff(cq.astype(np.int64)) >= 0)                                # Cartesian order survives
            assert fc["box"][c].tolist() == [cq.min(), cq.max() + 17, ct.min(), ct.max() + 17]
            ci = int(fc["chrom"][c]); assert CH[ci] <= ct.min() and ct.max() + 17 <= CH[ci + 1]   # inside one chromosome
            d = (ct.astype(np.int64) - cq) if fc["strand"][c] == 0 else (ct.astype(np.int64) + cq)
            assert np.abs(np.diff(d)).max() < 500 + 400                                     # neighbours stay near one diagonal
            n_cl += 1
    assert n_cl >= 6
    q, t, k, nf = _seed_cpu(reads[-1].tobytes(), ik, ip, gb, 17, 10, 150)
    assert len(O.matches_to_fine_clusters(q, t, k, nf, co, fo, CH)[0]["strand"]) == 0           # junk: nothing


@pytest.mark.gpu
@pytest.mark.parametrize("preset", ["CCS", "CONTIG", "LOOSE"])
def test_hip_fine_clusters_oracle(ctx, oracle, preset):
    from lra_amd import seed, cluster, index as I
    g = _genome_with_repeats(11)
    CH = [0, 250_000, len(g)]
    K = FINE[preset]["globalK"]
    I.load_genome(ctx, g)
    I.build_global_index(ctx, CH, K, 10, 150, 15, 1)
    ik, ip = I.global_index(ctx)
    rng = np.random.default_rng(5)
    reads = _reads(g, rng, err=0.01 if preset != "LOOSE" else 0.06)
    batch = seed.ReadBatch(ctx, [r.tobytes() for r in reads])
    seed.seed_batch(ctx, batch, K, 10, 150)
    cl = dict(CLEAN["CONTIG" if preset == "CONTIG" else "CCS"], globalK=K)
    if preset == "LOOSE":
        cl.update(minDiagCluster=3, SecondCleanMinDiagCluster=10)
    rough = cluster.clean_matches_batch(ctx, cluster.CleanOpts(**cl), CH)
    res = cluster.fine_clusters_batch(ctx, rough, cluster.FineOpts(**FINE[preset]), CH)
    out = cluster.fetch_fine(ctx, res)
    co = O.CleanOpts(**cl); fo = O.FineOpts(**FINE[preset])
    gb = g.tobytes() + b"\0" * 64
    n_cl = n_rev = n_multi = 0
    for r, rd in enumerate(reads):
        q, t, k, nf = _seed_cpu(rd.tobytes(), ik, ip, gb, K, 10, 150)
        exp, st = O.matches_to_fine_clusters(q, t, k, nf, co, fo, CH)
        assert (out["status"][r] != 0) == (st != 0), r
        if st:
            continue
        c0, c1 = int(out["cluster_off"][r]), int(out["cluster_off"][r + 1])
        assert c1 - c0 == len(exp["strand"]), (r, c1 - c0, len(exp["strand"]))
        for c in range(c1 - c0):
            a, b = int(out["match_off"][c0 + c]), int(out["match_off"][c0 + c + 1])
            ea, eb = int(exp["off"][c]), int(exp["off"][c + 1])
            assert np.array_equal(out["q"][a:b], exp["q"][ea:eb]) and np.array_equal(out["t"][a:b], exp["t"][ea:eb]), (r, c)
            assert out["box"][c0 + c].tolist() == exp["box"][c].tolist() and out["strand"][c0 + c] == exp["strand"][c] and out["chrom"][c0 + c] == exp["chrom"][c], (r, c)
            assert np.float32(out["freq"][c0 + c]).view(np.uint32) == np.float32(exp["freq"][c]).view(np.uint32), (r, c)
            n_cl += 1; n_rev += int(exp["strand"][c])
        n_multi += (c1 - c0) > 1
    assert n_cl >= len(reads) - 2 and n_rev >= 2 and n_multi >= 2, (n_cl, n_rev, n_multi)


def _fine_clusters_on_gpu(ctx, preset="CCS", seed_=11, err=0.01):
    """The a5 stage on the GPU for a set of reads: (genome, CH, reads, batch, fetched fine clusters)"""
    from lra_amd import seed, cluster, index as I
    g = _genome_with_repeats(seed_)
    CH = [0, 250_000, len(g)]
    K = FINE[preset]["globalK"]
    I.load_genome(ctx, g)
    I.build_global_index(ctx, CH, K, 10, 150, 15, 1)
    rng = np.random.default_rng(5)
    reads = _reads(g, rng, err=err)
    batch = seed.ReadBatch(ctx, [r.tobytes() for r in reads])
    seed.seed_batch(ctx, batch, K, 10, 150)
    cl = dict(CLEAN["CONTIG" if preset == "CONTIG" else "CCS"], globalK=K)
    rough = cluster.clean_matches_batch(ctx, cluster.CleanOpts(**cl), CH)
    res = cluster.fine_clusters_batch(ctx, rough, cluster.FineOpts(**FINE[preset]), CH)
    return g, CH, reads, batch, cluster.fetch_fine(ctx, res), K


@pytest.mark.gpu
@pytest.mark.parametrize("skip", [True, False])
def test_hip_linear_extend_clusters_oracle(ctx, oracle, skip):
    """a7, cluster version: every fine cluster of every read as an element of a chain (the read's clusters in order of their read start), plus
    clusters cut into two overlapping halves so that the neighbours' box coordinates fall inside a cluster (the CheckOverlap branches)."""
    import torch
    from lra_amd import cluster
    g, CH, reads, batch, fc, K = _fine_clusters_on_gpu(ctx)
    dev = ctx.device
    # refined clusters = the fine clusters with t relative to their chromosome, and halves of the large ones
    cl = []   # (read, q, t, strand, chrom, freq)
    for r in range(len(reads)):
        for c in range(int(fc["cluster_off"][r]), int(fc["cluster_off"][r + 1])):
            a, b = int(fc["match_off"][c]), int(fc["match_off"][c + 1])
            q = fc["q"][a:b]; t = fc["t"][a:b] - np.uint32(CH[int(fc["chrom"][c])])
            cl.append((r, q, t, int(fc["strand"][c]), int(fc["chrom"][c]), float(fc["freq"][c])))
            if b - a >= 40:
                o = np.argsort(q, kind="stable"); h = len(o) * 6 // 10
                cl.append((r, q[o[:h]], t[o[:h]], int(fc["strand"][c]), int(fc["chrom"][c]), 1.0))
                cl.append((r, q[o[-h:]], t[o[-h:]], int(fc["strand"][c]), int(fc["chrom"][c]), 1.05))
    box = np.array([[q.min(), q.max() + K, t.min(), t.max() + K] for _, q, t, *_ in cl], np.uint32)
    # chains: per read, its clusters ordered by qStart
    items = []
    for r in range(len(reads)):
        ids = [i for i, c in enumerate(cl) if c[0] == r]
        ids.sort(key=lambda i: (int(box[i][0]), i))
        for k, i in enumerate(ids):
            items.append((i, ids[k - 1] if k > 0 else -1, ids[k + 1] if k + 1 < len(ids) else -1, r))
    moff = np.concatenate([[0], np.cumsum([len(c[1]) for c in cl])]).astype(np.int64)
    tt = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    d_mq = tt(np.concatenate([c[1] for c in cl]), np.uint32).view(torch.int32) if False else tt(np.concatenate([c[1] for c in cl]).astype(np.int64), np.int64).to(torch.int32)
    d_mt = tt(np.concatenate([c[2] for c in cl]).astype(np.int64), np.int64).to(torch.int32)
    gdev = torch.from_numpy(np.concatenate([g, np.zeros(64, np.uint8)])).to(dev)
    res = cluster.linear_extend_clusters_batch(ctx, tt([x[0] for x in items], np.int32), tt([x[1] for x in items], np.int32), tt([x[2] for x in items], np.int32),
                                               tt([x[3] for x in items], np.int32), tt(moff, np.int64), d_mq, d_mt, tt(box.astype(np.int64), np.int64).to(torch.int32).view(-1),
                                               tt([c[3] for c in cl], np.int32), tt([c[4] for c in cl], np.int32), tt([c[5] for c in cl], np.float32), batch, gdev, CH,
                                               skiprepetitive=skip, K=K, trim=True)
    out = cluster.fetch_ext_clusters(ctx, res)
    sq = d_mq.cpu().numpy().view(np.uint32); stt = d_mt.cpu().numpy().view(np.uint32)
    n_ovl = n_rev = 0
    for k, (i, pv, nx, r) in enumerate(items):
        _, q, t, st, ci, fr = cl[i]
        exp = O.linear_extend_cluster(q, t, st, box[i], box[pv] if pv >= 0 else None, box[nx] if nx >= 0 else None, fr, reads[r].tobytes(), g[CH[ci]:CH[ci + 1]].tobytes(),
                                      K=K, skiprepetitive=skip, trim=True)
        a, b = int(out["off"][k]), int(out["off"][k + 1])
        assert np.array_equal(sq[moff[i]:moff[i + 1]], exp["sorted_q"]) and np.array_equal(stt[moff[i]:moff[i + 1]], exp["sorted_t"]), k
        assert np.array_equal(out["q"][a:b], exp["q"]) and np.array_equal(out["t"][a:b], exp["t"]) and np.array_equal(out["len"][a:b], exp["len"]), k
        assert np.array_equal(out["overlap"][a:b], exp["overlap"]), k
        assert out["box"][k].tolist() == exp["box"].tolist() and out["strand"][k] == st and out["chrom"][k] == ci, k
        n_ovl += int(exp["overlap"].sum()); n_rev += st
    assert len(items) >= 25 and n_rev >= 3 and (n_ovl >= 5) == skip, (len(items), n_rev, n_ovl)


def _both_strands(ctx, batch):
    """the reads forward, then reverse complemented, in one device buffer (what the drivers build once per batch)"""
    import ctypes as C_
    import torch
    tot = int(batch.total_bases)
    both = torch.zeros(2 * tot + 64, dtype=torch.uint8, device=ctx.device)
    both[:tot] = batch.seq[:tot]
    ctx.check(ctx.lib.lra_create_rc_batch(ctx.h, batch.n, C_.c_void_p(batch.seq.data_ptr()), C_.c_void_p(batch.off.data_ptr()), C_.c_void_p(both.data_ptr() + tot)))
    return both, tot


@pytest.mark.gpu
@pytest.mark.parametrize("read_type", [2, 3, 0])
def test_hip_refine_btwn_clusters_oracle(ctx, oracle, read_type):
    """a11 caller, high-accuracy path: RefineBtwnClusters_chain over chains made of the GPU's own fine clusters (a read's clusters by descending read
    start; reads with several clusters also get a second chain that shares a cluster with the first): the appended pairs, the boxes as the
    serial walk leaves them, refinespace and anchorfreq, for -CCS, -CONTIG and a low-accuracy read type (other refineSpaceDiag)."""
    import torch
    from lra_amd import cluster
    g, CH, reads, batch, fc0, K = _fine_clusters_on_gpu(ctx, err=0.02)
    dev = ctx.device
    # the clusters of the test: the fine clusters, the large ones cut into a head and a tail with the middle third removed (a space RefineBtwnSpace refills)
    pieces = []
    for r in range(len(reads)):
        for c in range(int(fc0["cluster_off"][r]), int(fc0["cluster_off"][r + 1])):
            a, b = int(fc0["match_off"][c]), int(fc0["match_off"][c + 1])
            q = fc0["q"][a:b]; t = fc0["t"][a:b] - np.uint32(CH[int(fc0["chrom"][c])])
            meta = (r, int(fc0["strand"][c]), int(fc0["chrom"][c]), float(fc0["freq"][c]))
            if b - a >= 60:
                o = np.argsort(q, kind="stable"); h = len(o) // 3
                pieces.append((q[o[:h]], t[o[:h]]) + meta); pieces.append((q[o[-h:]], t[o[-h:]]) + meta)
            else:
                pieces.append((q, t) + meta)
    nC = len(pieces)
    fc = dict(q=np.concatenate([x[0] for x in pieces]), strand=np.array([x[3] for x in pieces], np.int32), chrom=np.array([x[4] for x in pieces], np.int32),
              freq=np.array([x[5] for x in pieces], np.float32), match_off=np.concatenate([[0], np.cumsum([len(x[0]) for x in pieces])]).astype(np.int64),
              cluster_off=np.searchsorted(np.array([x[2] for x in pieces]), np.arange(len(reads) + 1)).astype(np.int64))
    rel_t = np.concatenate([x[1] for x in pieces])
    box = np.array([[x[0].min(), x[0].max() + K, x[1].min(), x[1].max() + K] for x in pieces], np.int64)
    chain_off = [0]; ch = []; read_chain_off = [0]
    for r in range(len(reads)):
        ids = list(range(int(fc["cluster_off"][r]), int(fc["cluster_off"][r + 1])))
        ids.sort(key=lambda i: -int(box[i][0]))
        if ids:
            ch.extend(ids); chain_off.append(len(ch))
            if len(ids) >= 2:
                ch.extend(ids[1:]); chain_off.append(len(ch))
        read_chain_off.append(len(chain_off) - 1)
    tt = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    both, tot = _both_strands(ctx, batch)
    gdev = torch.from_numpy(np.concatenate([g, np.zeros(64, np.uint8)])).to(dev)
    d_box = tt(box, np.int64).to(torch.int32).view(-1).contiguous(); d_freq = tt(fc["freq"], np.float32)
    res = cluster.refine_btwn_clusters_batch(ctx, tt(read_chain_off, np.int64), tt(chain_off, np.int64), tt(ch if ch else [0], np.int32), tt(fc["match_off"].astype(np.int64), np.int64),
                                             tt(fc["q"].astype(np.int64), np.int64).to(torch.int32), tt(rel_t.astype(np.int64), np.int64).to(torch.int32), d_box,
                                             tt(fc["strand"], np.int32), tt(fc["chrom"], np.int32), d_freq, batch.off, both, tot, gdev, CH, K=K, W=10, read_type=read_type)
    moff = ctx.to_host(res.d_match_off, nC + 1, np.uint64).astype(np.int64)
    oq = ctx.to_host(res.d_q, int(res.n_matches), np.uint32); ot = ctx.to_host(res.d_t, int(res.n_matches), np.uint32)
    rs = ctx.to_host(res.d_refinespace, nC, np.uint8)
    nbox = d_box.cpu().numpy().view(np.uint32).reshape(-1, 4); nfreq = d_freq.cpu().numpy()
    gb = g.tobytes() + b"\0" * 64
    n_added = n_ref = 0
    for r in range(len(reads)):
        c0, c1 = int(fc["cluster_off"][r]), int(fc["cluster_off"][r + 1])
        if c1 == c0:
            continue
        m0 = int(fc["match_off"][c0])
        x0, x1 = read_chain_off[r], read_chain_off[r + 1]
        co = [chain_off[x] - chain_off[x0] for x in range(x0, x1 + 1)]
        chn = [i - c0 for i in ch[chain_off[x0]:chain_off[x1]]]
        rd = reads[r].tobytes()
        exp = O.refine_btwn_clusters_chains(fc["match_off"][c0:c1 + 1].astype(np.int64) - m0, fc["q"][m0:int(fc["match_off"][c1])], rel_t[m0:int(fc["match_off"][c1])], box[c0:c1],
                                            fc["strand"][c0:c1], fc["chrom"][c0:c1], fc["freq"][c0:c1], co, chn, rd, synth.revcomp(reads[r]).tobytes(), gb, CH, K=K, W=10,
                                            read_type=read_type)
        for c in range(c1 - c0):
            a, b = int(moff[c0 + c]), int(moff[c0 + c + 1]); ea, eb = int(exp["off"][c]), int(exp["off"][c + 1])
            assert np.array_equal(oq[a:b], exp["q"][ea:eb]) and np.array_equal(ot[a:b], exp["t"][ea:eb]), (r, c, b - a, eb - ea)
            assert nbox[c0 + c].tolist() == exp["box"][c].tolist() and rs[c0 + c] == exp["refinespace"][c], (r, c)
            assert np.float32(nfreq[c0 + c]).view(np.uint32) == np.float32(exp["freq"][c]).view(np.uint32), (r, c)
            n_added += (eb - ea) - int(fc["match_off"][c0 + c + 1] - fc["match_off"][c0 + c]); n_ref += int(exp["refinespace"][c])
    assert int(res.n_pairs_added) == n_added and n_ref >= 3 and n_added >= 100 and int(res.n_rounds) >= 2, (int(res.n_pairs_added), n_added, n_ref, int(res.n_rounds))


def _random_cluster_chains(rng, n_jobs):
    """chains of merged clusters with every SPLITChain branch: far jumps, other chromosomes, inversions, duplicated target ranges, and the
    A / insert / A' pattern MergeSplitchainINS joins."""
    jobs = []
    for j in range(n_jobs):
        n = int(rng.integers(1, 14)) if j % 7 else int(rng.integers(1, 3))
        strand = []; chrom = []; box = []; link = []
        t = int(rng.integers(200000, 400000)); q = 100; st = int(rng.integers(0, 2)); ci = int(rng.integers(0, 3))
        for v in range(n):
            ln = int(rng.integers(200, 3000))
            mode = rng.integers(0, 10)
            if v and mode == 0: t += int(rng.integers(150000, 300000))
            elif v and mode == 1: ci = (ci + 1) % 3
            elif v and mode == 2: st ^= 1
            elif v and mode == 3: t -= int(ln * rng.uniform(0.7, 1.0))             # repeated target range
            elif v and mode == 4 and len(box) >= 2: t = box[-2][2] - ln - int(rng.integers(0, 1400)); ci = chrom[-2]; st = strand[-2]   # back next to the piece before the insert
            strand.append(st); chrom.append(ci); box.append([q, q + ln, max(0, t), max(0, t) + ln])
            q += ln + int(rng.integers(0, 50)); t = max(0, t) + ln + int(rng.integers(0, 60))
            if v: link.append(int(rng.integers(0, 2)))
        jobs.append((strand, chrom, box, link))
    return jobs


@pytest.mark.gpu
def test_hip_split_chains_highacc_oracle(ctx, oracle):
    """a9, high-accuracy SPLITChain: pieces, their order after MergeSplitchainINS, types, boxes and LSC for 400 random chains of merged clusters"""
    import torch
    from lra_amd import chain
    rng = np.random.default_rng(23)
    jobs = _random_cluster_chains(rng, 400)
    dev = ctx.device
    job_off = np.concatenate([[0], np.cumsum([len(j[0]) for j in jobs])]).astype(np.uint64)
    link_off = np.concatenate([[0], np.cumsum([len(j[3]) for j in jobs])]).astype(np.uint64)
    tt = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    cat = lambda k, dt: np.concatenate([np.asarray(j[k], dt).reshape(-1) for j in jobs] + [np.zeros(4, dt)])
    res = chain.split_chains_highacc_batch(ctx, tt(job_off.astype(np.int64), np.int64), tt(cat(0, np.int32)[:-4], np.int32), tt(cat(1, np.int32), np.int32),
                                           tt(cat(2, np.int64), np.int64).to(torch.int32), tt(link_off.astype(np.int64), np.int64), tt(cat(3, np.uint8), np.uint8))
    out = chain.fetch_hsplit(ctx, res)
    kinds = set(); merged = 0
    for j, (strand, chrom, box, link) in enumerate(jobs):
        exp = O.split_chain_highacc(strand, chrom, box, link)
        p0, p1 = int(out["job_piece_off"][j]), int(out["job_piece_off"][j + 1])
        assert p1 - p0 == len(exp["type"]), j
        assert int(out["lsc"][j]) == exp["lsc"], j
        for k in range(p1 - p0):
            a, b = int(out["piece_off"][p0 + k]) - int(job_off[j]), int(out["piece_off"][p0 + k + 1]) - int(job_off[j])
            got = out["sptc"][int(job_off[j]) + a:int(job_off[j]) + b].tolist(); want = exp["idx"][exp["off"][k]:exp["off"][k + 1]].tolist()
            assert got == want, (j, k)
            assert out["type"][p0 + k] == exp["type"][k] and out["strand"][p0 + k] == exp["strand"][k] and out["box"][p0 + k].tolist() == exp["box"][k].tolist(), (j, k)
            assert int(out["job"][p0 + k]) == j
            kinds.add(chr(int(exp["type"][k])))
            merged += int(any(want[i + 1] != want[i] + 1 for i in range(len(want) - 1)))
    assert kinds >= {"T", "D", "I", "N"} and merged >= 3, (kinds, merged)


def test_oracle_split_chain_highacc_sanity(oracle):
    """hand-made chains: an insert from another chromosome between two collinear pieces is bridged only when the far piece is not the chain's last one"""
    # chains run from the read's end to its start: A, a far insert X, A' (continues A on the target), B far away
    box = [[3000, 4000, 52000, 53000], [2500, 3000, 400000, 400500], [1500, 2500, 51000, 52000]]
    r = O.split_chain_highacc([0, 0, 0], [0, 0, 0], box, [0, 0])
    assert [chr(int(x)) for x in r["type"]] == ["T", "T", "N"] and r["idx"].tolist() == [0, 1, 2]     # A' is the last piece: chromIndex indeterminate, never bridged
    box2 = box + [[0, 1000, 900000, 901000]]
    r = O.split_chain_highacc([0, 0, 0, 0], [0, 0, 0, 0], box2, [0, 0, 0])
    assert r["idx"].tolist() == [0, 2, 1, 3] and r["off"].tolist() == [0, 2, 3, 4] and r["lsc"] == 0 and [chr(int(x)) for x in r["type"]] == ["T", "T", "N"], r
    r = O.split_chain_highacc([0, 1, 0], [0, 0, 0], [[0, 500, 1000, 1500], [500, 900, 1500, 1900], [900, 2000, 1900, 3000]], [0, 0])
    assert [chr(int(x)) for x in r["type"]] == ["I", "I", "N"] and r["lsc"] == 2


def _sv_reads(genome, rng, err, n_plain=10):
    mix = (34, 33, 33)
    reads, _ = synth.simulate_reads(genome, n_plain, 9000, 2500, err, mix, seed=int(rng.integers(1 << 30)))
    sim = lambda a, n, rev=False: synth.simulate_read(rng, genome[a:a + n + 1], n, err, mix, rev)[0]
    reads.append(np.concatenate([sim(50_000, 4000), sim(60_000, 4000)]))                                  # 6 kb deletion
    reads.append(np.concatenate([sim(160_000, 4000), sim(164_000, 2500, True), sim(166_500, 4000)]))      # inversion
    reads.append(np.concatenate([sim(250_000, 4500), sim(400_000, 4500, True)]))                          # translocation, second half reversed
    reads.append(synth.revcomp(np.concatenate([sim(300_000, 3000), sim(303_200, 3000)])))                 # 200 bp deletion, read on the reverse strand
    reads.append(np.concatenate([sim(20_000, 3000), sim(520_000, 2500), sim(23_000, 3000)]))              # insert from far away between two collinear parts
    reads.append(sim(146_000, 14_000))                                                                    # across a tandem array
    reads.append(sim(78_000, 16_000))                                                                     # inside the segmental duplication
    reads.append(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 2500)].copy())                        # junk
    return reads


@pytest.mark.gpu
@pytest.mark.parametrize("preset", ["ccs", "ccs-bp", "ccs-k17", "contig", "ccs-sparse", "ccs-gli", "contig-gli", "ccs-sparse-gli"])
def test_map_reads_highacc_match_oracle_pipeline(ctx, oracle, preset):
    """lra_map_reads_highacc_batch against MapRead_highacc composed from the oracle's stage functions (tests/oracle_pipeline.map_read_highacc): every
    SegAlignment of every chain -- strand, Supplymentary, ISsecondary, NumOfAnchors0/1, the chain's value, the refined blocks, the counters the two
    CalculateStatistics calls leave, NV bits and CIGAR runs -- on plain reads and reads with a deletion / an inversion / a translocation / an insert,
    reads in repeats, and a read that cannot align.  ccs-bp: --refineBreakpoints (which on this path turns RefineBreakpoint OFF, Map_highacc.h:723)."""
    import oracle_pipeline as OP
    from lra_amd import seed, mapread, index as I
    g = _genome_with_repeats(19)
    CH = [0, 250_000, len(g)]
    rng = np.random.default_rng(8)
    reads = _sv_reads(g, rng, 0.01)
    over = {}
    oo = dict(OP.CCS)
    ip = (17, 10, 150, 15, 1)                                             # `lra index -CCS`
    gli = preset.endswith("-gli")                                         # glIndex as glIndex.Read leaves it after `lra index`: k = 10, w = 5, windows of 2048 bases (LocalIndex(0),
    preset = preset.replace("-gli", "")                                   # MMIndex.h:110-127; RunStoreLocal keeps k = 10 under -CCS / -CONTIG, lra.cpp:785-804); without a .gli file: opts.localK = 7, 256
    if preset == "contig":                                                # -CONTIG: refineBand 50 (rows of more than 64 cells), K 19, other gap costs, contig thresholds
        oo = dict(OP.CONTIG); ip = (19, 10, 30, 20, 1)
        sim = lambda a, n, rev=False: synth.simulate_read(rng, g[a:a + n + 1], n, 0.003, (34, 33, 33), rev)[0]
        reads = reads[:6] + [sim(20_000, 60_000), sim(300_000, 45_000, True), np.concatenate([sim(260_000, 20_000), sim(284_000, 25_000)]),
                             np.concatenate([sim(30_000, 15_000), sim(350_000, 12_000, True), sim(45_000, 15_000)])] + reads[10:]
    if preset == "ccs-bp":
        over["refineBreakpoint"] = 1; oo["refineBreakpoint"] = True
    if preset == "ccs-k17":                                               # denser seeds: more clusters per read, more second chains
        over.update({"globalK": 17, "globalW": 10, "clean.globalK": 17, "sdp.globalK": 17, "fine.globalK": 17}); oo.update(globalK=17, globalW=10); ip = (17, 10, 150, 15, 1)
    if preset == "ccs-sparse":                                            # a slightly thinner global index (one minimizer per 18 bases instead of 15; the reads are sketched with W = 20): clusters at ~0.01 anchors per base, so some
        ip = (17, 10, 150, 18, 1)                                         # reads take the REFINEclusters branch (Map_highacc.h:413-447) and some do not
    if gli:
        over.update(localK=10, localIndexWindow=2048); oo.update(localK=10, localIndexWindow=2048)
    mapper = mapread.HighAccMapper(ctx, g, None, None, [b"chrA", b"chrB"], CH, "contig" if preset == "contig" else "ccs", index_params=ip, **over)
    ik, ipos = I.global_index(ctx)
    g_index = mapper.fetch_local_index()
    res = mapper.align(seed.ReadBatch(ctx, [r.tobytes() for r in reads]))
    out = mapper.fetch(res)
    na = int(res.num_aln)
    gb = g.tobytes()
    n_seg = n_supp = n_rev = n_multi = n_bp = n_sec = n_unsup = n_acc = 0
    for r, rd in enumerate(reads):
        exp, unaligned, note = OP.map_read_highacc(rd.tobytes(), gb, ik, ipos, oo, chrom_pos=CH, g_index=g_index)
        n_unsup += int(bool(OP.TRACE.get("sparse")) and exp is not None and len(exp) > 0)        # (name kept: reads through the REFINEclusters branch)
        if exp and OP.TRACE.get("sparse"):
            assert all(out["job_reached"][r * na + G["h"]] == 3 for G in exp), r
        assert note is None and out["read_status"][r] == 0, (r, note, out["read_status"][r])
        by_h = {G["h"]: G["segs"] for G in exp}
        for h in range(na):
            a0, a1 = int(out["job_aln_off"][r * na + h]), int(out["job_aln_off"][r * na + h + 1])
            assert bool(out["job_reached"][r * na + h]) == (h in by_h), (r, h, note)
            e = by_h.get(h, [])
            assert a1 - a0 == len(e), (r, h, a1 - a0, len(e))
            for a, s in zip(range(a0, a1), e):
                assert (out["strand"][a], out["supp"][a], out["secondary"][a], out["n0"][a], out["n1"][a], out["chrom"][a]) == \
                       (s["strand"], s["supp"], s["secondary"], s["n0"], s["n1"], s["chrom"]), (r, h, a)
                assert np.float32(out["first_sdp_value"][a]).view(np.uint32) == np.float32(s["value"]).view(np.uint32), (r, h, a)
                assert out["refine_status"][a] == s["refine_status"] == 0, (r, h, a)
                b = out["blocks"][int(out["block_off"][a]):int(out["block_off"][a + 1])]
                assert np.array_equal(b, s["blocks"]), (r, h, a, len(b), len(s["blocks"]))
                ec, ev, eruns, _ = s["stats"]
                assert out["counts"][a].tolist() == [ec[k] for k in O.STAT_NAMES], (r, h, a, out["counts"][a].tolist(), [ec[k] for k in O.STAT_NAMES])
                assert np.float32(out["value"][a]).view(np.uint32) == np.float32(ev).view(np.uint32), (r, h, a)
                assert np.array_equal(out["runs"][int(out["run_off"][a]):int(out["run_off"][a + 1])], eruns), (r, h, a)
                n_seg += 1; n_supp += int(s["supp"]); n_rev += int(s["strand"]); n_bp += int(s.get("breakpoint", 0) == 1); n_sec += int(s["secondary"])
                n_acc += int(ec["tdel"] + ec["tins"] > 0)
            n_multi += len(e) > 1
        if unaligned:
            assert not any(out["job_reached"][r * na:(r + 1) * na]), r
    assert n_seg >= len(reads) - 3 and n_supp >= 3 and n_rev >= 3 and n_multi >= 2 and n_acc >= 3, (n_seg, n_supp, n_rev, n_multi, n_acc, n_unsup)
    assert (n_bp >= 1) == (preset != "ccs-bp"), n_bp
    if preset == "ccs-sparse":
        assert 3 <= n_unsup <= len(reads) - 3, n_unsup                       # both branches in one batch
    # the records: every read gets its lines (or none), supplementary segments carry SA tags, flagged reads are left out
    names = [b"r%d" % i for i in range(len(reads))]
    texts = mapper.records(res, names, [r.tobytes() for r in reads])
    assert len(texts) == len(reads)
    for r, t in enumerate(texts):
        if out["read_status"][r]:
            assert t == b"", r
    n_rev_lines = 0
    for r, t in enumerate(texts):                                         # SEQ = strands[str] (Map_highacc.h:704): the reverse complement for a reverse-strand record
        for l in t.decode().split("\n"):
            if not l or int(l.split("\t")[1]) & 4:
                continue
            ff = l.split("\t")
            sread = mapread.create_rc(reads[r].tobytes()) if int(ff[1]) & 16 else reads[r].tobytes()
            m0 = re.match(r"^(\d+)H", ff[5]); m1 = re.search(r"(\d+)H$", ff[5])
            h0 = int(m0.group(1)) if m0 else 0; h1 = int(m1.group(1)) if m1 else 0
            assert ff[9].encode() == sread[h0:len(sread) - h1], (r, ff[1])
            n_rev_lines += bool(int(ff[1]) & 16)
    assert n_rev_lines >= 3
    assert texts[-1].split(b"\t")[1] == b"4"                              # the junk read: one unaligned record
    assert sum(1 for t in texts if t.count(b"\n") >= 2) >= 2               # split reads: several lines


@pytest.mark.gpu
def test_config0_shape_ccs_ecoli_sized(ctx, oracle):
    """BASELINE configs[0] at its own size: a 4.6 Mb single-chromosome reference (E. coli K-12 sized) + 1000 simulated 10 kb CCS reads, -CCS: every read's
    SegAlignments from lra_map_reads_highacc_batch against the oracle composition (the reference's CPU-runnable case)."""
    import oracle_pipeline as OP
    from lra_amd import seed, mapread, index as I
    g = synth.make_genome(4_600_000, seed=101, repeat_frac=0.03, n_families=4)
    CH = [0, len(g)]
    reads, truth = synth.simulate_reads(g, 1000, 10000, 1500, 0.01, (34, 33, 33), seed=77)
    mapper = mapread.HighAccMapper(ctx, g, None, None, [b"U00096.3"], CH, "ccs")
    ik, ipos = I.global_index(ctx)
    g_index = mapper.fetch_local_index()
    res = mapper.align(seed.ReadBatch(ctx, [r.tobytes() for r in reads]))
    out = mapper.fetch(res)
    na = int(res.num_aln)
    gb = g.tobytes()
    n_aln = n_right = 0
    for r, rd in enumerate(reads):
        exp, unaligned, note = OP.map_read_highacc(rd.tobytes(), gb, ik, ipos, OP.CCS, chrom_pos=CH, g_index=g_index, stats=(r % 10 == 0))
        assert note is None and out["read_status"][r] == 0, (r, note, out["read_status"][r])
        by_h = {G["h"]: G["segs"] for G in exp}
        for h in range(na):
            a0, a1 = int(out["job_aln_off"][r * na + h]), int(out["job_aln_off"][r * na + h + 1])
            e = by_h.get(h, [])
            assert a1 - a0 == len(e) and bool(out["job_reached"][r * na + h]) == (h in by_h), (r, h, a1 - a0, len(e))
            for a, s in zip(range(a0, a1), e):
                assert (out["strand"][a], out["supp"][a], out["secondary"][a], out["n0"][a], out["n1"][a]) == (s["strand"], s["supp"], s["secondary"], s["n0"], s["n1"]), (r, h)
                b = out["blocks"][int(out["block_off"][a]):int(out["block_off"][a + 1])]
                assert np.array_equal(b, s["blocks"]), (r, h, len(b), len(s["blocks"]))
                if "stats" in s:
                    ec, ev, eruns, _ = s["stats"]
                    assert out["counts"][a].tolist() == [ec[k] for k in O.STAT_NAMES] and np.array_equal(out["runs"][int(out["run_off"][a]):int(out["run_off"][a + 1])], eruns), (r, h)
                n_aln += 1
        if by_h.get(0):                                                     # the primary alignment sits where the read was drawn from
            s0 = by_h[0][0]
            st, ln, rev = truth[r]
            n_right += abs(int(s0["blocks"][0][1]) - st) < 200 and s0["strand"] == int(rev)
    assert n_aln >= 990 and n_right >= 980, (n_aln, n_right)
