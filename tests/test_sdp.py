"""a8 sparse DP (SDP#A, SparseDP.h:2139): oracle vs. the reference components' golden file (CPU) and HIP vs. oracle (GPU)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(__file__), "golden", "sdp_parts_golden.json")


@pytest.fixture(scope="module")
def gold():
    return json.load(open(GOLD))


def test_oracle_divide_matches_reference(gold):
    """sorts (incl. libstdc++'s permutation of tied points), row/col tables, SS lists, all four decompositions"""
    for c in gold["divide"]:
        p = np.array(c["pts"], dtype=np.int64).reshape(-1, 4)
        if "text" in c:
            n, h, text = O.sdp_divide_dump(p[:, 0], p[:, 1], p[:, 2], p[:, 3], want_text=True)
            assert text == c["text"]
        else:
            n, h = O.sdp_divide_dump(p[:, 0], p[:, 1], p[:, 2], p[:, 3])
        assert n == c["len"] and "%016x" % h == c["fnv1a"]


def test_oracle_pwl_matches_reference(gold):
    xs = gold["xs"]
    for c in gold["pwl"]:
        a, b, s, i = O.sdp_pwl(c["params"], xs)
        assert s.tolist() == c["slope_bits"] and i.tolist() == c["inter_bits"]
        assert a.tolist() == c["pwl_bits"]
        assert b.tolist() == c["w_bits"]


def test_oracle_maximization_matches_reference(gold):
    for c in gold["maxim"]:
        ops = np.array(c["ops"], dtype=np.int64).reshape(-1, 3)
        out, blk = O.sdp_maximization_script(c["params"], c["Di"], c["Ei"], [tuple(o) for o in ops])
        ref = np.array(c["out"], dtype=np.int64)
        # the reference prints i2 as unsigned; -2 marks inputs where it reads outside its arrays
        ok = out != -2
        assert out.shape == ref.shape
        assert (out[ok] == ref[ok]).all()
        assert ok.mean() > 0.99
        if ok.all():
            assert blk.tolist() == c["block"]


# ---------------------------------------------------------------- GPU: HIP SDP#A vs. oracle ---------------------------
def _random_clusters(rng, n_clusters, per_cluster, span, ties):
    offs = [0]; strands = []; Q = []; T = []; L = []
    for c in range(n_clusters):
        strand = int(rng.random() < 0.4)
        q = int(rng.integers(0, span)); t = int(rng.integers(0, span)) + (span if strand else 0)
        m = int(rng.integers(1, per_cluster + 1))
        for i in range(m):
            ln = int(rng.choice([1, 2, 3])) if ties else int(rng.choice([17, 17, 20, 30, 60, 150]))
            Q.append(q); T.append(t); L.append(ln)
            step = int(rng.integers(0, 3)) if ties else int(rng.integers(0, 200))
            q += ln + step
            if strand:
                t = max(0, t - ln - (int(rng.integers(0, 3)) if ties else int(rng.integers(0, 200))))
            else:
                t += ln + (int(rng.integers(0, 3)) if ties else int(rng.integers(0, 200)))
        strands.append(strand); offs.append(len(Q))
    return (np.array(offs, np.int32), np.array(strands, np.uint8), np.array(Q, np.uint32), np.array(T, np.uint32), np.array(L, np.int32))


def _run_hip(ctx, reads_in, read_lens, opts_kw):
    """reads_in: list of (cluster_off, strands, q, t, len) per read -> fetched result dict"""
    import torch
    from lra_amd import chain
    dev = ctx.device
    coff = [0]; cstart = []; ccount = []; cstrand = []; Q = []; T = []; L = []
    pad = 0
    for (offs, st, q, t, ln) in reads_in:
        for c in range(len(st)):
            a, b = int(offs[c]), int(offs[c + 1])
            Q.extend([7] * pad); T.extend([9] * pad); L.extend([1] * pad)          # gaps between clusters, as lra_linear_extend_batch leaves them
            cstart.append(len(Q)); ccount.append(b - a); cstrand.append(int(st[c]))
            Q.extend(q[a:b].tolist()); T.extend(t[a:b].tolist()); L.extend(ln[a:b].tolist())
            pad = (pad + 1) % 3
        coff.append(len(cstart))
    roff = np.concatenate([[0], np.cumsum(read_lens)]).astype(np.int64)
    tt = lambda a, dt: torch.tensor(np.asarray(a, dtype=dt), device=dev)
    d = dict(cluster_off=tt(coff, np.int64), c_start=tt(cstart if cstart else [0], np.int64), c_count=tt(ccount if ccount else [0], np.int32),
             c_strand=tt(cstrand if cstrand else [0], np.int32), q=tt(Q if Q else [0], np.int64).to(torch.int32),
             t=tt(T if T else [0], np.int64).to(torch.int32), ln=tt(L if L else [0], np.int32), roff=tt(roff, np.int64))
    res = chain.sparse_dp_batch(ctx, len(reads_in), d["cluster_off"], d["c_start"], d["c_count"], d["c_strand"], d["q"], d["t"], d["ln"], d["roff"],
                                chain.sdp_opts(**opts_kw))
    return res, chain.fetch(ctx, res)


def _compare(out, num_aln, reads_in, read_lens, opts_kw):
    n_ok = 0
    for r, (offs, st, q, t, ln) in enumerate(reads_in):
        exp = O.sdp_chain(offs, st, q, t, ln, O.sdp_opts(read_lens[r], **opts_kw))
        if exp["status"] < 0:
            assert out["status"][r] != 0, r
            continue
        assert out["status"][r] == 0, (r, out["status"][r])
        f0, f1 = int(out["frag_off"][r]), int(out["frag_off"][r + 1])
        assert f1 - f0 == len(q)
        assert np.array_equal(out["frag_val"][f0:f1].view(np.uint32), exp["val"].view(np.uint32)), r
        assert int(out["n_chains"][r]) == len(exp["chains"]), (r, out["n_chains"][r], len(exp["chains"]))
        for c, ch in enumerate(exp["chains"]):
            s = r * num_aln + c
            a = int(out["chain_start"][s]); m = int(out["chain_len"][s])
            assert m == len(ch["frags"]), (r, c)
            cl = np.searchsorted(offs, ch["frags"], side="right") - 1
            assert np.array_equal(out["chain_cluster"][a:a + m], cl), (r, c)
            assert np.array_equal(out["chain_anchor"][a:a + m], ch["frags"] - offs[cl]), (r, c)
            assert np.array_equal(out["chain_link"][a:a + m - 1], ch["link"]), (r, c)
            assert out["chain_box"][s].tolist() == ch["box"].tolist(), (r, c)
            assert np.float32(out["chain_value"][s]) == np.float32(ch["value"]), (r, c)
            n_ok += 1
    return n_ok


@pytest.mark.gpu
@pytest.mark.parametrize("sizes", ["estimated", "outgrown", "counted"])
def test_hip_sdp_random_clusters(ctx, sizes, monkeypatch):
    """`sizes`: how a read's blocks are laid out -- from the per-point estimate (the default: no count pass), from estimates so low that most reads outgrow them and
    are counted and built again, and from the count pass for every read (LRA_SDP_ONEPASS=0)."""
    if sizes == "outgrown":
        monkeypatch.setenv("LRA_SDP_ESTIMATE", "2.0,0.5")
    if sizes == "counted":
        monkeypatch.setenv("LRA_SDP_ONEPASS", "0")
    rng = np.random.default_rng(77)
    reads_in = []
    for k in range(160):
        if k < 40: reads_in.append(_random_clusters(rng, int(rng.integers(1, 4)), 3, 12, True))        # heavy coordinate ties
        elif k < 80: reads_in.append(_random_clusters(rng, int(rng.integers(1, 5)), 6, 40, True))
        elif k < 140: reads_in.append(_random_clusters(rng, int(rng.integers(1, 8)), 12, 3000, False))
        else: reads_in.append(_random_clusters(rng, int(rng.integers(4, 30)), 40, 30000, False))
    reads_in.insert(5, (np.array([0], np.int32), np.zeros(0, np.uint8), np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.int32)))   # read without clusters
    reads_in.insert(9, (np.array([0, 1], np.int32), np.array([1], np.uint8), np.array([5], np.uint32), np.array([100], np.uint32), np.array([17], np.int32)))
    read_lens = [max(300, int(q.max()) + 200) if len(q) else 300 for (_, _, q, _, _) in reads_in]
    for kw in (dict(), dict(NumAln=3, alnthres=0.3, rate=3.0), dict(gapopen=4.0, gapextend=20.0, gapCeiling1=3000, gapCeiling2=5000, rate=1.0)):
        res, out = _run_hip(ctx, reads_in, read_lens, kw)
        n = _compare(out, res.num_aln, reads_in, read_lens, kw)
        assert n > 40


@pytest.mark.gpu
def test_hip_sdp_single_cluster_mode(ctx):
    """SparseDP(ClusterIndex, ...) (SparseDP.h:2287, Map_lowacc.h:535): one cluster per job, first maximum, plain trace back"""
    rng = np.random.default_rng(5)
    jobs = []
    for k in range(120):
        ties = k < 40
        offs, st, q, t, ln = _random_clusters(rng, 1, int(rng.integers(1, 60)), 12 if ties else 5000, ties)
        jobs.append((offs, st, q, t, ln))
    read_lens = [1000] * len(jobs)
    # SparseDP(SplitChain&, vector<Cluster_SameDiag*>&, FinalChain&, ...) (SparseDP.h:1766, LocalRefineAlignment.h:563) is the same
    # engine over the several clusters of a split chain: one point pair per anchor by its cluster's strand, first maximum, plain trace back
    for k in range(60):
        jobs.append(_random_clusters(rng, int(rng.integers(2, 7)), 25, 4000, False))
    read_lens = [1000] * len(jobs)
    kw = dict(mode=1, NumAln=1, rate=6.0)                                 # second_anchorbonus of the -ONT / -CLR presets
    res, out = _run_hip(ctx, jobs, read_lens, kw)
    assert _compare(out, res.num_aln, jobs, read_lens, kw) == len(jobs)


@pytest.mark.gpu
def test_hip_sdp_on_oracle_pipeline_reads(ctx):
    """30 kb ONT-like reads through the oracle's a1-a7, then SDP#A on the GPU vs. the oracle"""
    from lra_amd import synth
    from sdp_inputs import oracle_ext_clusters
    genome = synth.make_genome(1_500_000, seed=12, repeat_frac=0.3)
    ik, ip = synth.build_global_index(genome, 17, 10, 150)
    reads, truth = synth.simulate_reads(genome, 10, 20000, 6000, 0.10, seed=4)
    reads_in = [oracle_ext_clusters(O, r.tobytes(), genome, ik, ip) for r in reads]
    read_lens = [len(r) for r in reads]
    res, out = _run_hip(ctx, reads_in, read_lens, {})
    assert _compare(out, res.num_aln, reads_in, read_lens, {}) >= 8
    # the primary chain lies on the simulated locus
    for r, (s0, L, strand) in enumerate(truth):
        if out["n_chains"][r]:
            box = out["chain_box"][r * res.num_aln]
            assert box[2] >= s0 - 200 and box[2] <= s0 + L + 200


@pytest.mark.gpu
def test_hip_sdp_chained_after_hip_extend(ctx, oracle):
    """seed -> clean -> extend -> SDP#A all on the GPU; the oracle's SDP on the GPU's own extended anchors must agree"""
    from lra_amd import synth, seed, cluster, chain
    genome = synth.make_genome(600_000, seed=31, repeat_frac=0.4, n_families=3)
    ik, ip = synth.build_global_index(genome, 17, 10, 100)
    reads, truth = synth.simulate_reads(genome, 40, 12000, 4000, 0.10, seed=9)
    reads += [np.frombuffer(b"ACGT" * 5, dtype=np.uint8)]
    seed.load_reference(ctx, genome, ik, ip)
    batch = seed.ReadBatch(ctx, [r.tobytes() for r in reads])
    seed.seed_batch(ctx, batch, 17, 10, 150)
    po = dict(oracle.CLEAN_PRESETS["ONT"]); po["globalK"] = 17
    cres = cluster.clean_matches_batch(ctx, cluster.CleanOpts(**po), [0, len(genome)])
    eres = cluster.linear_extend_batch(ctx, 17, batch)
    co = cluster.fetch(ctx, cres); eo = cluster.fetch_extend(ctx, eres)
    res = chain.sparse_dp_batch(ctx, len(reads), cres.d_cluster_off, eres.d_e_start, eres.d_e_count, cres.d_c_strand, eres.d_e_qpos, eres.d_e_tpos,
                                eres.d_e_len, batch.off, chain.sdp_opts())
    out = chain.fetch(ctx, res)
    reads_in = []
    for r in range(len(reads)):
        offs = [0]; st = []; Q = []; T = []; L = []
        for x in range(int(co["cluster_off"][r]), int(co["cluster_off"][r + 1])):
            a, n = int(eo["e_start"][x]), int(eo["e_count"][x])
            Q.extend(eo["e_qpos"][a:a + n].tolist()); T.extend(eo["e_tpos"][a:a + n].tolist()); L.extend(eo["e_len"][a:a + n].tolist())
            st.append(int(co["strand"][x])); offs.append(len(Q))
        reads_in.append((np.array(offs, np.int32), np.array(st, np.uint8), np.array(Q, np.uint32), np.array(T, np.uint32), np.array(L, np.int32)))
    read_lens = [len(r) for r in reads]
    assert _compare(out, res.num_aln, reads_in, read_lens, {}) >= 30
    hit = 0
    for r, (s0, L, strand) in enumerate(truth):
        if out["n_chains"][r]:
            box = out["chain_box"][r * res.num_aln]
            hit += int(box[2] >= s0 - 300 and box[2] <= s0 + L + 300)
    assert hit >= 35


@pytest.mark.gpu
def test_hip_sdp_bench_workload_flags_and_sample(ctx, oracle):
    """bench-like reads (30 kb, 10 % error, repeats in a 16 Mb genome): every read the kernel flags must be one where the reference itself
    reads outside its arrays (oracle status -1), never a work-buffer bound; a sample of the others must match the oracle."""
    import torch
    from lra_amd import synth_torch as st, seed, cluster, chain
    dev = ctx.device
    genome = st.make_genome(16_000_000, 1, dev)
    ik, ip = st.build_global_index(genome, 17, 10, 150)
    sim = st.simulate_batch(genome, 1024, 30000, 3000, 0.10, (30, 35, 35), 77)
    pad = torch.zeros(64, dtype=torch.uint8, device=dev)
    reads = torch.cat([sim["seq"], pad])
    seed.load_reference(ctx, genome.cpu().numpy(), ik, ip)
    rb = seed.read_batch_from_device(ctx, reads, sim["off"])
    seed.seed_batch(ctx, rb, 17, 10, 150)
    po = dict(oracle.CLEAN_PRESETS["ONT"]); po["globalK"] = 17
    cres = cluster.clean_matches_batch(ctx, cluster.CleanOpts(**po), [0, int(genome.numel())])
    eres = cluster.linear_extend_batch(ctx, 17, rb)
    res = chain.sparse_dp_batch(ctx, 1024, cres.d_cluster_off, eres.d_e_start, eres.d_e_count, cres.d_c_strand, eres.d_e_qpos, eres.d_e_tpos, eres.d_e_len,
                                rb.off, chain.sdp_opts())
    out = chain.fetch(ctx, res)
    co = cluster.fetch(ctx, cres); eo = cluster.fetch_extend(ctx, eres)
    off = sim["off"].cpu().numpy()
    flagged = np.nonzero(out["status"])[0].tolist()
    sample = sorted(set(flagged + list(range(0, 1024, 37))))
    reads_in = {}
    for r in sample:
        offs = [0]; stv = []; Q = []; T = []; L = []
        for x in range(int(co["cluster_off"][r]), int(co["cluster_off"][r + 1])):
            a, n = int(eo["e_start"][x]), int(eo["e_count"][x])
            Q.extend(eo["e_qpos"][a:a + n].tolist()); T.extend(eo["e_tpos"][a:a + n].tolist()); L.extend(eo["e_len"][a:a + n].tolist())
            stv.append(int(co["strand"][x])); offs.append(len(Q))
        reads_in[r] = (np.array(offs, np.int32), np.array(stv, np.uint8), np.array(Q, np.uint32), np.array(T, np.uint32), np.array(L, np.int32))
    n_ok = 0
    for r in sample:
        offs, stv, q, t, ln = reads_in[r]
        exp = O.sdp_chain(offs, stv, q, t, ln, O.sdp_opts(int(off[r + 1] - off[r])))
        if out["status"][r] != 0:
            assert out["status"][r] & 8 == 0, (r, "work-buffer bound hit", out["status"][r])
            assert exp["status"] < 0, (r, out["status"][r])
            continue
        assert exp["status"] >= 0, r
        f0, f1 = int(out["frag_off"][r]), int(out["frag_off"][r + 1])
        assert np.array_equal(out["frag_val"][f0:f1].view(np.uint32), exp["val"].view(np.uint32)), r
        assert int(out["n_chains"][r]) == len(exp["chains"]), r
        for c, ch in enumerate(exp["chains"]):
            s = r * res.num_aln + c
            a = int(out["chain_start"][s]); m = int(out["chain_len"][s])
            cl = np.searchsorted(offs, ch["frags"], side="right") - 1
            assert m == len(ch["frags"]) and np.array_equal(out["chain_cluster"][a:a + m], cl) and np.array_equal(out["chain_anchor"][a:a + m], ch["frags"] - offs[cl]), (r, c)
        n_ok += 1
    assert n_ok >= 20


@pytest.mark.gpu
@pytest.mark.parametrize("window", ["window", "l2", "window-old-order"])
@pytest.mark.parametrize("which", ["clusters", "single"])
def test_hip_sdp_workgroup_kernel(ctx, which, window, monkeypatch):
    """The workgroup-per-read ProcessPoint (sdp_process_wg, for reads with many points: the slots spread over 16 waves) against the oracle: the
    threshold is lowered so that the ordinary test reads go through it, plus one read of 9000 anchors on a lattice of tied rows / columns /
    diagonals (what a read from a satellite array looks like).  Both ways the waves exchange an anchor's value: the LDS window of anchors in
    progress, and the words at L2 that reads with long anchor spans fall back to (forced here with LRA_SDP_WG_RING=0)."""
    monkeypatch.setenv("LRA_SDP_BIG_POINTS", "40")
    if window == "l2":
        monkeypatch.setenv("LRA_SDP_WG_RING", "0")
    if window == "window-old-order":
        # the large reads' build and workgroup launch beside the small reads' (the order before round 5) instead of in front of them: both orders stay tested
        monkeypatch.setenv("LRA_SDP_BIG_FIRST", "0")
    rng = np.random.default_rng(23)
    if which == "clusters":
        reads_in = [_random_clusters(rng, nc, per, span, ties) for nc, per, span, ties in [(6, 80, 30000, False), (3, 150, 20000, True), (10, 30, 30000, True), (1, 5, 1000, False)]]
        kw = {}
    else:
        reads_in = [_random_clusters(rng, 1, per, span, ties) for per, span, ties in [(300, 20000, False), (600, 30000, True), (40, 3000, True)]]
        # a lattice: anchors at (171 a + 7 b, 171 c + 7 b) -- thousands of tied coordinates and diagonals
        a_ = rng.integers(0, 60, 9000); c_ = rng.integers(0, 60, 9000); b_ = rng.integers(0, 24, 9000)
        q = (171 * a_ + 7 * b_).astype(np.uint32); t = (171 * c_ + 7 * b_ + 5000).astype(np.uint32)
        _, u = np.unique(q.astype(np.int64) * 100000 + t, return_index=True)
        u = np.sort(u)
        o = np.lexsort((q[u], q[u].astype(np.int64) - t[u].astype(np.int64)))
        reads_in.append((np.array([0, len(u)], np.int32), np.array([0], np.uint8), q[u][o], t[u][o], np.full(len(u), 12, np.int32)))
        # long anchors one base apart: ~1200 points between an anchor's start and its end -- more than the LDS window serves, the read takes the L2 words by itself
        i_ = np.arange(2500); q = i_.astype(np.uint32); t = (i_ + 7000 + (i_ % 3) * 50).astype(np.uint32)
        o = np.lexsort((q, q.astype(np.int64) - t.astype(np.int64)))
        reads_in.append((np.array([0, 2500], np.int32), np.array([0], np.uint8), q[o], t[o], np.full(2500, 600, np.int32)))
        kw = dict(mode=1, rate=2.0)
    read_lens = [40000] * len(reads_in)
    res, out = _run_hip(ctx, reads_in, read_lens, kw)
    assert _compare(out, int(res.num_aln), reads_in, read_lens, kw) >= len(reads_in) - 1


def _load_jobs(path):
    """jobs written by LRA_SDP_DUMP (lra_amd/csrc/sdp.hip): header (mode, clusters, anchors, read length), rate, cluster offsets, strands, q, t, len"""
    b = open(path, "rb").read(); at = 0; jobs = []
    while at < len(b):
        mode, nc, total, rl = (int(x) for x in np.frombuffer(b, np.int32, 4, at)); at += 16
        rate = float(np.frombuffer(b, np.float32, 1, at)[0]); at += 4
        off = np.frombuffer(b, np.int32, nc + 1, at).copy(); at += 4 * (nc + 1)
        st = np.frombuffer(b, np.uint8, nc, at).copy(); at += nc
        q = np.frombuffer(b, np.uint32, total, at).copy(); at += 4 * total
        t = np.frombuffer(b, np.uint32, total, at).copy(); at += 4 * total
        ln = np.frombuffer(b, np.int32, total, at).copy(); at += 4 * total
        jobs.append((mode, rate, (off, st, q, t, ln)))
    return jobs


@pytest.mark.gpu
@pytest.mark.parametrize("window", ["window", "l2"])
def test_hip_sdp_heavy_jobs_of_the_bench(ctx, window, monkeypatch):
    """Jobs captured from the bench batch (tests/golden/sdp_heavy_jobs.bin, inputs only): the largest merged cluster of a read from a satellite array
    (23390 anchors on the reverse strand: 46780 points, Block lists of 17 k pairs, `last` moving backwards in four queries of five), another one of
    19812 anchors, the largest SDP#A read and the largest job of a13's inner sparse DP -- the workgroup kernel on real input, against the oracle."""
    if window == "l2":
        monkeypatch.setenv("LRA_SDP_WG_RING", "0")
    jobs = _load_jobs(os.path.join(os.path.dirname(__file__), "golden", "sdp_heavy_jobs.bin"))
    assert len(jobs) == 4
    for mode in (1, 0):
        sel = [j for j in jobs if j[0] == mode]
        reads_in = [j[2] for j in sel]
        kw = dict(mode=1, rate=sel[0][1]) if mode == 1 else dict(rate=sel[0][1])
        read_lens = [40000] * len(reads_in)
        res, out = _run_hip(ctx, reads_in, read_lens, kw)
        assert _compare(out, int(res.num_aln), reads_in, read_lens, kw) >= len(reads_in)


def test_oracle_stack_pairs_all_have_the_boundary_n():
    """What both HIP ProcessPoint kernels are built on: in the literal Maximization every pair pushed on S_1 besides the dummy is (i, Ei.size()) -- FindBoundary
    searches an empty range or returns Ei.size() (SubRountine.h:339-354, :388-434).  Checked on the literal restatement over the pinned Maximization scripts, random
    clusters with heavy ties, and the heavy jobs captured from the bench batch."""
    L = O.lib()
    L.oracle_sdp_off_boundary_pushes.restype = __import__("ctypes").c_long
    L.oracle_sdp_off_boundary_pushes(1)
    gold = json.load(open(GOLD))
    for c in gold["maxim"]:
        ops = np.array(c["ops"], dtype=np.int64).reshape(-1, 3)
        O.sdp_maximization_script(c["params"], c["Di"], c["Ei"], [tuple(o) for o in ops])
    rng = np.random.default_rng(5)
    n_anchors = 0
    for k in range(60):
        offs, st, q, t, ln = _random_clusters(rng, int(rng.integers(1, 8)), 120, 20000, bool(k & 1))
        O.sdp_chain(offs, st, q, t, ln, O.sdp_opts(40000))
        n_anchors += len(q)
    for mode, rate, job in _load_jobs(os.path.join(os.path.dirname(__file__), "golden", "sdp_heavy_jobs.bin")):
        O.sdp_chain(*job, O.sdp_opts(40000, mode=mode, rate=rate))
        n_anchors += len(job[2])
    assert n_anchors > 40000 and L.oracle_sdp_off_boundary_pushes(0) == 0
