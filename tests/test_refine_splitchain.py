"""a10 glue: Refine_splitchain (ChainRefine.h:384-576).  The whole low-accuracy front end runs on the GPU (seed -> clean -> extend ->
SDP#A -> SPLITChain), then lra_refine_splitchain_batch; the oracle restates Refine_splitchain per split chain on the GPU's own chains.
Parity unpinned (ChainRefine.h needs htslib headers); with limitrefine the reference reads an uninitialised upper bound: implemented as
the binary's measured behaviour, no upper bound (see the oracle file)."""
import numpy as np
import pytest

import oracle_lib as O

K = 17


def _seq_offsets(starts, total, window):
    """LocalIndex::seqOffsets of sequences starting at `starts` (MMIndex.h:200-245): window ends, restarting at each sequence"""
    out = [0]
    ends = list(starts[1:]) + [total]
    for s, e in zip(starts, ends):
        p = s
        while p < e:
            p = min(p + window, e)
            out.append(p)
    return np.array(out, np.uint64)


def test_oracle_refine_splitchain_sanity():
    # one forward anchor chain on the diagonal t = q + 1000 of a random sequence: every refined match must lie no more than 100 below
    # that diagonal (there is no upper bound with limitrefine: ChainRefine.h:468) and inside the split chain's box, in
    # chromosome coordinates
    rng = np.random.default_rng(5)
    genome = rng.integers(0, 4, 6000).astype(np.uint8)
    g = np.frombuffer(b"ACGT", np.uint8)[genome]
    read = g[1000:3000].copy()
    gt, gb = O.local_index_seq(g.tobytes(), 10, 5, 256, 15)
    qt, qb = O.local_index_seq(read.tobytes(), 10, 5, 256, 15)
    gso = _seq_offsets([0], len(g), 256); qso = _seq_offsets([0], len(read), 256)
    q = np.arange(1900, 0, -100, dtype=np.uint32); t = q + 1000                       # trace-back order
    n = len(q)
    sptc = np.arange(n - 1, -1, -1)                                                   # forward split chains list anchors first-to-last
    box = [int(q.min()), int(q.max()) + 20, int(t.min()), int(t.max()) + 20]
    r = O.refine_splitchain(q, t, [20] * n, [0] * n, [0] * n, sptc, box, 0, 0, [0], [0, len(g)], len(read), (qso, qb, qt), (gso, gb, gt))
    assert r is not None and len(r["q"]) > 50
    d = r["t"].astype(np.int64) - r["q"].astype(np.int64)
    assert np.all(d - 1000 >= -100) and np.all(r["q"] >= box[0]) and np.all(r["q"] < box[1]) and np.all(r["t"] >= box[2]) and np.all(r["t"] < box[3])
    assert np.sum(d == 1000) > 40 and r["box"][0] == r["q"].min() and r["box"][1] == r["q"].max() + 10
    r2 = O.refine_splitchain(q, t, [20] * n, [0] * n, [0] * n, sptc, box, 0, 0, [0], [0, len(g)], len(read), (qso, qb, qt), (gso, gb, gt), limitrefine=False)
    assert np.all(np.abs(r2["t"].astype(np.int64) - r2["q"].astype(np.int64) - 1000) <= 50) and len(r2["q"]) <= len(r["q"])
    # a stretch of the read copied from 140 bases further along the genome (diagonal 1140, inside the same genome / read windows as the
    # anchors around it): kept with limitrefine (no upper bound), dropped without it (maxDiagNum = 1050), and a stretch 140 bases
    # further back (diagonal 860) is dropped by both (lower bounds 900 / 950)
    read2 = read.copy(); read2[560:700] = g[1700:1840]; read2[1100:1240] = g[1960:2100]
    qt2, qb2 = O.local_index_seq(read2.tobytes(), 10, 5, 256, 15)
    r3 = O.refine_splitchain(q, t, [20] * n, [0] * n, [0] * n, sptc, box, 0, 0, [0], [0, len(g)], len(read), (qso, qb2, qt2), (gso, gb, gt))
    d3 = r3["t"].astype(np.int64) - r3["q"].astype(np.int64)
    assert np.sum(d3 == 1140) > 5 and np.sum(d3 == 860) == 0 and np.all(d3 >= 900)
    r4 = O.refine_splitchain(q, t, [20] * n, [0] * n, [0] * n, sptc, box, 0, 0, [0], [0, len(g)], len(read), (qso, qb2, qt2), (gso, gb, gt), limitrefine=False)
    d4 = r4["t"].astype(np.int64) - r4["q"].astype(np.int64)
    assert np.sum(d4 == 1140) == 0 and np.sum(d4 == 860) == 0


def _front_end(ctx, oracle):
    import torch
    from lra_amd import synth, seed, cluster, chain, local
    dev = ctx.device
    genome = synth.make_genome(600_000, seed=41, repeat_frac=0.3, n_families=3)
    CH = [0, 299_900, 600_000]                                                        # two chromosomes; the second does not start on a window edge
    ik, ip = synth.build_global_index(genome, K, 10, 100)
    reads, truth = synth.simulate_reads(genome, 36, 9000, 3000, 0.10, seed=19)
    # chimeric reads: two loci far apart, the second half possibly reverse-complemented -> several split chains, 'T' / 'I' pieces
    rng = np.random.default_rng(8)
    for j in range(8):
        a = int(rng.integers(10_000, 250_000)); b = int(rng.integers(330_000, 560_000))
        ra = synth.simulate_read(rng, genome[a:a + 5000], 4000, 0.08, (30, 35, 35), False)[0]
        rb = synth.simulate_read(rng, genome[b:b + 5000], 4000, 0.08, (30, 35, 35), bool(j & 1))[0]
        reads.append(np.concatenate([ra, rb]))
    # an inversion inside a read (forward, reverse-complemented middle, forward) and a 6 kb deletion: neighbouring split chains with a
    # space between them that Refine_Btwnsplitchain seeds (the INV / two-block and the plain branches)
    for j in range(4):
        a = int(rng.integers(20_000, 250_000))
        A_ = synth.simulate_read(rng, genome[a:a + 4001], 4000, 0.06, (30, 35, 35), False)[0]
        B_ = synth.simulate_read(rng, genome[a + 4000:a + 6501], 2500, 0.06, (30, 35, 35), True)[0]
        C_ = synth.simulate_read(rng, genome[a + 6500:a + 10501], 4000, 0.06, (30, 35, 35), False)[0]
        reads.append(np.concatenate([A_, B_, C_]))
        b = int(rng.integers(320_000, 560_000))
        D_ = synth.simulate_read(rng, genome[b:b + 4001], 4000, 0.06, (30, 35, 35), False)[0]
        E_ = synth.simulate_read(rng, genome[b + 10_000:b + 14_001], 4000, 0.06, (30, 35, 35), False)[0]
        reads.append(np.concatenate([D_, E_]) if j & 1 else synth.revcomp(np.concatenate([D_, E_])))
    # reads hugging the chromosome ends (the walk then runs into the last windows)
    reads.append(genome[599_000:600_000].copy()); reads.append(genome[299_000:299_900].copy()); reads.append(genome[0:1500].copy())
    reads.append(np.frombuffer(b"ACGT" * 5, dtype=np.uint8))
    n = len(reads)
    seed.load_reference(ctx, genome, ik, ip)
    batch = seed.ReadBatch(ctx, [r.tobytes() for r in reads])
    seed.seed_batch(ctx, batch, K, 10, 150)
    po = dict(oracle.CLEAN_PRESETS["ONT"]); po["globalK"] = K
    cres = cluster.clean_matches_batch(ctx, cluster.CleanOpts(**po), CH)
    eres = cluster.linear_extend_batch(ctx, K, batch)
    chres = chain.sparse_dp_batch(ctx, n, cres.d_cluster_off, eres.d_e_start, eres.d_e_count, cres.d_c_strand, eres.d_e_qpos, eres.d_e_tpos,
                                  eres.d_e_len, batch.off, chain.sdp_opts())
    co = chain.fetch(ctx, chres)
    # the chain arrays are reused by later SDP calls: keep our own copies alive for the split / refine stages
    spres = chain.split_chains_batch(ctx, chres, CH, 5000, 1)                       # splitdist 5000: the 6 kb deletions split their chains
    so = chain.fetch_split(ctx, spres)
    fidx = ctx.to_host(spres.d_fidx, spres.n_frags, np.uint32)
    # local indexes: genome (two sequences), reads forward + reverse complement in one index
    gdev = torch.from_numpy(np.concatenate([genome, np.zeros(64, np.uint8)])).to(dev)
    gli = local.LocalIndex(ctx, gdev, torch.tensor(CH, dtype=torch.int64, device=dev), 10, 5, 256, 15)
    rc = seed.create_rc(ctx, batch)
    tot = int(batch.off[-1])
    both = torch.cat([batch.seq[:tot], rc[:tot], torch.zeros(64, dtype=torch.uint8, device=dev)])
    off2 = torch.cat([batch.off, batch.off[1:] + tot])
    rli = local.LocalIndex(ctx, both, off2, 10, 5, 256, 15)
    gso = _seq_offsets(CH[:-1], CH[-1], 256)
    assert len(gso) == gli.n_windows + 1
    gso_d = torch.from_numpy(gso.astype(np.int64)).to(dev)
    g_win, g_bnd, g_tup = gli.fetch()
    r_win, r_bnd, r_tup = rli.fetch()
    lens = [len(r) for r in reads]
    na = chres.num_aln
    return dict(**{k: v for k, v in locals().items() if k not in ("ctx", "oracle")})


@pytest.mark.gpu
def test_hip_refine_splitchain_oracle(ctx, oracle):
    from lra_amd import chain
    P = _front_end(ctx, oracle)
    chres, spres, batch, CH, rli, gso_d, gli, co, so, fidx, n, na, lens = (P[k] for k in ("chres", "spres", "batch", "CH", "rli", "gso_d", "gli", "co", "so", "fidx", "n", "na", "lens"))
    r_win, r_bnd, r_tup, gso, g_bnd, g_tup = (P[k] for k in ("r_win", "r_bnd", "r_tup", "gso", "g_bnd", "g_tup"))
    for limit in (True, False):
        res = chain.refine_splitchain_batch(ctx, chres, spres, batch.off, CH, rli, gso_d, gli, window=100, smallK=10, K=K, limitrefine=limit, max_freq=15)
        out = chain.fetch_refined(ctx, res)
        n_checked = n_rev = n_matches = n_multi = 0
        for r in range(n):
            for c in range(int(co["n_chains"][r])):
                s = r * na + c
                if so["status"][s]:
                    continue
                b = int(co["chain_start"][s]); ln = int(co["chain_len"][s])
                keep = so["keep"][b:b + ln].astype(bool)
                sel = np.nonzero(keep)[0]
                assert np.array_equal(fidx[b:b + len(sel)], sel)
                q = co["chain_q"][b:b + ln][sel]; t = co["chain_t"][b:b + ln][sel]; al = co["chain_alen"][b:b + ln][sel]
                cl = co["chain_cluster"][b:b + ln][sel]; cst = co["chain_strand"][b:b + ln][sel]
                nsp = int(so["n_split"][s])
                n_multi += nsp > 1
                for k in range(nsp):
                    x = b + k
                    a0, m = b + int(so["sp_beg"][x]), int(so["sp_len"][x])
                    sptc = so["sp_idx"][a0:a0 + m]
                    c0, cm = b + int(so["ci_beg"][x]), int(so["ci_len"][x])
                    strand = int(so["sp_strand"][x])
                    w0, w1 = int(r_win[strand * n + r]), int(r_win[strand * n + r + 1])
                    q_index = (_seq_offsets([0], lens[r], 256), r_bnd[w0:w1 + 1] - r_bnd[w0], r_tup[int(r_bnd[w0]):int(r_bnd[w1])])
                    exp = O.refine_splitchain(q, t, al, cl, cst, sptc, so["sp_box"][x], strand, int(so["sp_chrom"][x]), so["ci_idx"][c0:c0 + cm], CH, lens[r],
                                              q_index, (gso, g_bnd, g_tup), window=100, smallK=10, K=K, limitrefine=limit, max_freq=15)
                    m0, m1 = int(out["match_off"][x]), int(out["match_off"][x + 1])
                    if exp is None:
                        assert out["status"][x] != 0 and m1 == m0, (r, c, k)
                        continue
                    assert out["status"][x] == 0, (r, c, k, out["status"][x])
                    assert m1 - m0 == len(exp["q"]), (r, c, k, m1 - m0, len(exp["q"]))
                    assert np.array_equal(out["match_q"][m0:m1], exp["q"]) and np.array_equal(out["match_t"][m0:m1], exp["t"]), (r, c, k)
                    if m1 > m0:
                        assert np.array_equal(out["box"][x], exp["box"]), (r, c, k)
                        assert out["eff"][x].view(np.uint32) == exp["eff"].view(np.uint32), (r, c, k)
                    n_checked += 1; n_rev += strand; n_matches += m1 - m0
        assert n_checked >= 40 and n_rev >= 10 and n_matches > 20000 and n_multi >= 4, (n_checked, n_rev, n_matches, n_multi)


@pytest.mark.gpu
def test_hip_refine_btwn_splitchain_oracle(ctx, oracle):
    """a11 callers: Refine_Btwnsplitchain on the refined clusters the GPU produced, against the oracle chain by chain"""
    from lra_amd import chain, synth
    P = _front_end(ctx, oracle)
    chres, spres, batch, CH, rli, gso_d, gli, co, so, n, na, reads, genome, both, tot, gdev = (P[k] for k in (
        "chres", "spres", "batch", "CH", "rli", "gso_d", "gli", "co", "so", "n", "na", "reads", "genome", "both", "tot", "gdev"))
    rres = chain.refine_splitchain_batch(ctx, chres, spres, batch.off, CH, rli, gso_d, gli, window=100, smallK=10, K=K, limitrefine=True, max_freq=15)
    ro = chain.fetch_refined(ctx, rres)
    gbytes = genome.tobytes()
    for rsd, sparse in ((10000, 0.01), (10000, 0.2)):                    # the second threshold pushes results through the sparse (grouping) branch
        bres = chain.refine_btwn_splitchain_batch(ctx, chres, spres, rres, batch.off, both, tot, gdev, CH, K=10, W=5, refineSpaceDist=rsd,
                                                  anchorstoosparse=sparse, match=4, mismatch=-1, indel=-2, max_freq=15)
        bo = chain.fetch_btwn(ctx, bres)
        assert bres.n_problems > 40 and bres.n_rounds >= 4, (bres.n_problems, bres.n_rounds)
        n_chains = n_grown = n_multi = n_added = 0
        for r in range(n):
            fwd = reads[r].tobytes(); rc = synth.revcomp(reads[r]).tobytes()
            for c in range(int(co["n_chains"][r])):
                s = r * na + c
                if so["status"][s]:
                    continue
                b = int(co["chain_start"][s]); nsp = int(so["n_split"][s])
                if nsp == 0:
                    continue
                offs = [0]; mq = []; mt = []
                for k in range(nsp):
                    m0, m1 = int(ro["match_off"][b + k]), int(ro["match_off"][b + k + 1])
                    mq.extend(ro["match_q"][m0:m1].tolist()); mt.extend(ro["match_t"][m0:m1].tolist()); offs.append(len(mq))
                exp = O.refine_btwn_splitchain(offs, mq, mt, ro["box"][b:b + nsp], so["sp_strand"][b:b + nsp], so["sp_chrom"][b:b + nsp],
                                               so["split_link"][b:b + max(nsp - 1, 0)], fwd, rc, gbytes, CH, K=10, W=5, refineSpaceDist=rsd,
                                               anchorstoosparse=sparse, match=4, mismatch=-1, indel=-2, max_freq=15)
                assert exp is not None
                for k in range(nsp):
                    x = b + k
                    m0, m1 = int(bo["match_off"][x]), int(bo["match_off"][x + 1])
                    e0, e1 = int(exp["off"][k]), int(exp["off"][k + 1])
                    assert m1 - m0 == e1 - e0, (r, c, k, m1 - m0, e1 - e0)
                    assert np.array_equal(bo["match_q"][m0:m1], exp["q"][e0:e1]) and np.array_equal(bo["match_t"][m0:m1], exp["t"][e0:e1]), (r, c, k)
                    assert np.array_equal(bo["box"][x], exp["box"][k]), (r, c, k)
                    assert bo["refinespace"][x] == exp["refinespace"][k], (r, c, k)
                    grown = (m1 - m0) - (offs[k + 1] - offs[k])
                    n_grown += grown > 0; n_added += grown
                n_chains += 1; n_multi += nsp > 1
        assert n_chains >= 40 and n_multi >= 8 and n_grown >= 20 and n_added > 130, (n_chains, n_multi, n_grown, n_added)


@pytest.mark.gpu
def test_hip_merge_extend_and_second_sdp_oracle(ctx, oracle):
    """MergeChain + LinearExtend + DecideCoordinates + TrimOverlappedAnchors on the GPU's refined clusters, then the second sparse DP
    (single-cluster mode) straight from the device arrays, both against the oracle"""
    from lra_amd import chain
    P = _front_end(ctx, oracle)
    chres, spres, batch, CH, rli, gso_d, gli, co, so, n, na, reads, genome, both, tot, gdev = (P[k] for k in (
        "chres", "spres", "batch", "CH", "rli", "gso_d", "gli", "co", "so", "n", "na", "reads", "genome", "both", "tot", "gdev"))
    rres = chain.refine_splitchain_batch(ctx, chres, spres, batch.off, CH, rli, gso_d, gli, window=100, smallK=10, K=K, limitrefine=True, max_freq=15)
    bres = chain.refine_btwn_splitchain_batch(ctx, chres, spres, rres, batch.off, both, tot, gdev, CH, K=10, W=5)
    bo = chain.fetch_btwn(ctx, bres)
    mres = chain.merge_extend_batch(ctx, chres, spres, bres, batch.seq, batch.off, gdev, CH, K=10)
    mo = chain.fetch_merge(ctx, mres)
    gbytes = genome.tobytes()
    n_groups = n_merged = n_trim = n_anchors = 0
    exp_groups = {}
    for r in range(n):
        for c in range(int(co["n_chains"][r])):
            s = r * na + c
            g0, g1 = int(mo["slot_group_off"][s]), int(mo["slot_group_off"][s + 1])
            if so["status"][s] or int(so["n_split"][s]) == 0:
                assert g1 == g0
                continue
            b = int(co["chain_start"][s]); nsp = int(so["n_split"][s])
            offs = [0]; mq = []; mt = []
            for k in range(nsp):
                m0, m1 = int(bo["match_off"][b + k]), int(bo["match_off"][b + k + 1])
                mq.extend(bo["match_q"][m0:m1].tolist()); mt.extend(bo["match_t"][m0:m1].tolist()); offs.append(len(mq))
            exp = O.merge_extend(offs, mq, mt, bo["box"][b:b + nsp], so["sp_strand"][b:b + nsp], so["sp_chrom"][b:b + nsp], reads[r].tobytes(), gbytes, CH, K=10)
            ng = len(exp["member"]) - 1
            assert g1 - g0 == ng, (r, c, g1 - g0, ng)
            cb = int(mo["cluster_base"][s])
            for g in range(ng):
                G = g0 + g
                assert int(mo["group_first"][G]) - cb == (exp["member"][g] if g else 0) and int(mo["group_last"][G]) - cb == exp["member"][g + 1] - 1
                a0, cnt = int(mo["anchor_off"][G]), int(mo["count"][G])
                e0, e1 = int(exp["anchor_off"][g]), int(exp["anchor_off"][g + 1])
                assert cnt == e1 - e0, (r, c, g)
                assert np.array_equal(mo["q"][a0:a0 + cnt], exp["q"][e0:e1]) and np.array_equal(mo["t"][a0:a0 + cnt], exp["t"][e0:e1]), (r, c, g)
                assert np.array_equal(mo["len"][a0:a0 + cnt], exp["len"][e0:e1]), (r, c, g)
                assert np.array_equal(mo["box"][G], exp["box"][g]) and mo["strand"][G] == exp["strand"][g] and mo["chrom"][G] == exp["chrom"][g], (r, c, g)
                exp_groups[G] = (exp["q"][e0:e1], exp["t"][e0:e1], exp["len"][e0:e1], int(exp["strand"][g]))
                n_groups += 1; n_merged += int(exp["member"][g + 1] - (exp["member"][g] if g else 0) > 1); n_anchors += cnt
    assert n_groups >= 40 and n_anchors > 5000, (n_groups, n_merged, n_anchors)
    # the second sparse DP (Map_lowacc.h:535) on the merged clusters, fed from the device arrays
    opts = chain.sdp_opts(mode=1, rate=1.0)
    cres2 = chain.sparse_dp_batch(ctx, int(mres.n_groups), mres.d_iota, mres.d_anchor_off, mres.d_count, mres.d_strand, mres.d_q, mres.d_t, mres.d_len,
                                  mres.d_iota, opts)
    c2 = chain.fetch(ctx, cres2)
    n2 = 0
    for G, (q, t, ln, st) in exp_groups.items():
        if len(q) == 0:
            continue
        exp = O.sdp_chain([0, len(q)], [st], q, t, ln, O.sdp_opts(1000, mode=1, rate=1.0))
        if exp["status"] < 0:
            assert c2["status"][G] != 0
            continue
        assert c2["status"][G] == 0
        f0 = int(c2["frag_off"][G])
        assert np.array_equal(c2["frag_val"][f0:f0 + len(q)].view(np.uint32), exp["val"].view(np.uint32)), G
        a = int(c2["chain_start"][G * cres2.num_aln]); m = int(c2["chain_len"][G * cres2.num_aln])
        assert m == len(exp["chains"][0]["frags"]) and np.array_equal(c2["chain_anchor"][a:a + m], exp["chains"][0]["frags"]), G
        n2 += 1
    assert n2 >= 40


@pytest.mark.gpu
def test_hip_lowacc_chain_bench_like_sample(ctx, oracle):
    """bench-like reads (30 kb, 10 % error, 16 Mb genome with repeats) through the whole chained low-accuracy front end on the GPU
    (seed -> ... -> second sparse DP's inputs); a sample of reads is checked stage by stage against the oracle, every read against
    size-independent properties (matches inside their boxes, anchors inside merged boxes, CSR consistency)."""
    import torch
    from lra_amd import synth_torch as st, seed, cluster, chain, local
    dev = ctx.device
    NR = 384
    genome = st.make_genome(16_000_000, 1, dev)
    ik, ip = st.build_global_index(genome, K, 10, 150)
    sim = st.simulate_batch(genome, NR, 30000, 3000, 0.10, (30, 35, 35), 99)
    pad = torch.zeros(64, dtype=torch.uint8, device=dev)
    g2 = torch.Generator(device=dev).manual_seed(5)
    rev = torch.rand(NR, generator=g2, device=dev) < 0.5
    reads = torch.cat([st.revcomp_some(sim["seq"], sim["off"], rev), pad])
    G = int(genome.numel())
    CH = [0, G]
    seed.load_reference(ctx, genome.cpu().numpy(), ik, ip)
    rb = seed.read_batch_from_device(ctx, reads, sim["off"])
    seed.seed_batch(ctx, rb, K, 10, 150)
    po = dict(oracle.CLEAN_PRESETS["ONT"]); po["globalK"] = K
    cres = cluster.clean_matches_batch(ctx, cluster.CleanOpts(**po), CH)
    eres = cluster.linear_extend_batch(ctx, K, rb)
    chres = chain.sparse_dp_batch(ctx, NR, cres.d_cluster_off, eres.d_e_start, eres.d_e_count, cres.d_c_strand, eres.d_e_qpos, eres.d_e_tpos, eres.d_e_len,
                                  rb.off, chain.sdp_opts())
    co = chain.fetch(ctx, chres)
    spres = chain.split_chains_batch(ctx, chres, CH)
    so = chain.fetch_split(ctx, spres)
    gdev = torch.cat([genome, pad])
    gli = local.LocalIndex(ctx, gdev, torch.tensor(CH, dtype=torch.int64, device=dev), 10, 5, 256, 15)
    tot = int(rb.off[-1])
    both = torch.zeros(2 * tot + 64, dtype=torch.uint8, device=dev)
    both[:tot] = rb.seq[:tot]
    both[tot:2 * tot] = seed.create_rc(ctx, rb)[:tot]
    off2 = torch.cat([rb.off, rb.off[1:] + tot]).contiguous()
    rli = local.LocalIndex(ctx, both, off2, 10, 5, 256, 15)
    gso = _seq_offsets([0], G, 256)
    gso_d = torch.from_numpy(gso.astype(np.int64)).to(dev)
    rres = chain.refine_splitchain_batch(ctx, chres, spres, rb.off, CH, rli, gso_d, gli, window=100, smallK=10, K=K, limitrefine=True, max_freq=15)
    ro = chain.fetch_refined(ctx, rres)
    bres = chain.refine_btwn_splitchain_batch(ctx, chres, spres, rres, rb.off, both, tot, gdev, CH, K=10, W=5, refineSpaceDist=30000, anchorstoosparse=0.005)
    bo = chain.fetch_btwn(ctx, bres)
    mres = chain.merge_extend_batch(ctx, chres, spres, bres, rb.seq, rb.off, gdev, CH, K=10)
    mo = chain.fetch_merge(ctx, mres)
    ch2 = chain.sparse_dp_batch(ctx, int(mres.n_groups), mres.d_iota, mres.d_anchor_off, mres.d_count, mres.d_strand, mres.d_q, mres.d_t, mres.d_len,
                                mres.d_iota, chain.sdp_opts(mode=1, rate=2.0))
    c2 = chain.fetch(ctx, ch2)
    na = chres.num_aln
    off = sim["off"].cpu().numpy()
    # ---- properties on every read
    assert int(ro["match_off"][-1]) == rres.n_matches and np.all(np.diff(ro["match_off"].astype(np.int64)) >= 0)
    assert int(bo["match_off"][-1]) == bres.n_matches and np.all(bo["match_off"].astype(np.int64) >= ro["match_off"].astype(np.int64))
    assert int(mo["anchor_off"][-1]) == mres.n_anchors
    n_with = 0
    for r in range(NR):
        L = int(off[r + 1] - off[r])
        for c in range(int(co["n_chains"][r])):
            s = r * na + c
            if so["status"][s]:
                continue
            b = int(co["chain_start"][s])
            for k in range(int(so["n_split"][s])):
                x = b + k
                m0, m1 = int(bo["match_off"][x]), int(bo["match_off"][x + 1])
                if m1 == m0:
                    continue
                q = bo["match_q"][m0:m1]; t = bo["match_t"][m0:m1]; bx = bo["box"][x]
                assert q.min() == bx[0] and q.max() + 10 == bx[1] and t.min() == bx[2] and t.max() + 10 == bx[3], (r, c, k)
                assert bx[1] <= L and bx[3] <= G
                n_with += 1
            for g in range(int(mo["slot_group_off"][s]), int(mo["slot_group_off"][s + 1])):
                a0, cnt = int(mo["anchor_off"][g]), int(mo["count"][g])
                if cnt:
                    assert int(c2["status"][g]) == 0 and int(c2["n_chains"][g]) == 1
    assert n_with >= NR // 2
    # ---- stage-by-stage parity on a sample of reads
    g_win, g_bnd, g_tup = gli.fetch()
    r_win, r_bnd, r_tup = rli.fetch()
    gbytes = genome.cpu().numpy().tobytes()
    reads_h = reads.cpu().numpy()
    n_checked = 0
    for r in range(0, NR, 24):
        L = int(off[r + 1] - off[r])
        rd = reads_h[int(off[r]):int(off[r + 1])]
        from lra_amd import synth
        fwd = rd.tobytes(); rc = synth.revcomp(rd).tobytes()
        for c in range(int(co["n_chains"][r])):
            s = r * na + c
            if so["status"][s] or int(so["n_split"][s]) == 0:
                continue
            b = int(co["chain_start"][s]); ln = int(co["chain_len"][s]); nsp = int(so["n_split"][s])
            sel = np.nonzero(so["keep"][b:b + ln])[0]
            q = co["chain_q"][b:b + ln][sel]; t = co["chain_t"][b:b + ln][sel]; al = co["chain_alen"][b:b + ln][sel]
            cl = co["chain_cluster"][b:b + ln][sel]; cst = co["chain_strand"][b:b + ln][sel]
            offs = [0]; mq = []; mt = []
            for k in range(nsp):
                x = b + k
                a0, m = b + int(so["sp_beg"][x]), int(so["sp_len"][x])
                c0, cm = b + int(so["ci_beg"][x]), int(so["ci_len"][x])
                strand = int(so["sp_strand"][x])
                w0, w1 = int(r_win[strand * NR + r]), int(r_win[strand * NR + r + 1])
                q_index = (_seq_offsets([0], L, 256), r_bnd[w0:w1 + 1] - r_bnd[w0], r_tup[int(r_bnd[w0]):int(r_bnd[w1])])
                exp = O.refine_splitchain(q, t, al, cl, cst, so["sp_idx"][a0:a0 + m], so["sp_box"][x], strand, int(so["sp_chrom"][x]), so["ci_idx"][c0:c0 + cm], CH, L,
                                          q_index, (gso, g_bnd, g_tup), window=100, smallK=10, K=K, limitrefine=True, max_freq=15)
                m0, m1 = int(ro["match_off"][x]), int(ro["match_off"][x + 1])
                assert exp is not None and np.array_equal(ro["match_q"][m0:m1], exp["q"]) and np.array_equal(ro["match_t"][m0:m1], exp["t"]), (r, c, k)
                mq.extend(exp["q"].tolist()); mt.extend(exp["t"].tolist()); offs.append(len(mq))
            expb = O.refine_btwn_splitchain(offs, mq, mt, ro["box"][b:b + nsp], so["sp_strand"][b:b + nsp], so["sp_chrom"][b:b + nsp],
                                            so["split_link"][b:b + max(nsp - 1, 0)], fwd, rc, gbytes, CH, K=10, W=5, refineSpaceDist=30000, anchorstoosparse=0.005)
            assert expb is not None
            for k in range(nsp):
                x = b + k
                m0, m1 = int(bo["match_off"][x]), int(bo["match_off"][x + 1]); e0, e1 = int(expb["off"][k]), int(expb["off"][k + 1])
                assert np.array_equal(bo["match_q"][m0:m1], expb["q"][e0:e1]) and np.array_equal(bo["match_t"][m0:m1], expb["t"][e0:e1]), (r, c, k)
                assert np.array_equal(bo["box"][x], expb["box"][k])
            expm = O.merge_extend(expb["off"], expb["q"], expb["t"], expb["box"], so["sp_strand"][b:b + nsp], so["sp_chrom"][b:b + nsp], fwd, gbytes, CH, K=10)
            g0 = int(mo["slot_group_off"][s])
            assert int(mo["slot_group_off"][s + 1]) - g0 == len(expm["member"]) - 1
            for g in range(len(expm["member"]) - 1):
                a0, cnt = int(mo["anchor_off"][g0 + g]), int(mo["count"][g0 + g]); e0, e1 = int(expm["anchor_off"][g]), int(expm["anchor_off"][g + 1])
                assert cnt == e1 - e0 and np.array_equal(mo["q"][a0:a0 + cnt], expm["q"][e0:e1]) and np.array_equal(mo["t"][a0:a0 + cnt], expm["t"][e0:e1])
                assert np.array_equal(mo["len"][a0:a0 + cnt], expm["len"][e0:e1]) and np.array_equal(mo["box"][g0 + g], expm["box"][g])
            n_checked += 1
    assert n_checked >= 12


@pytest.mark.gpu
def test_hip_refine_clusters_oracle(ctx, oracle):
    """REFINEclusters (high-accuracy path) on the clusters the GPU's CleanMatches produced (two chromosomes; a few clusters are made to span
    the chromosome boundary so that CHROMIndex rejects them), against the oracle cluster by cluster"""
    import torch
    from lra_amd import chain, cluster
    P = _front_end(ctx, oracle)
    cres, batch, CH, rli, gso_d, gli, n, lens = (P[k] for k in ("cres", "batch", "CH", "rli", "gso_d", "gli", "n", "lens"))
    r_win, r_bnd, r_tup, gso, g_bnd, g_tup = (P[k] for k in ("r_win", "r_bnd", "r_tup", "gso", "g_bnd", "g_tup"))
    co = cluster.fetch(ctx, cres)
    nc = int(cres.n_clusters)
    dev = ctx.device
    cnt = (co["end"] - co["start"]).astype(np.int32)
    ts = co["tStart"].copy(); te = co["tEnd"].copy()
    rej = [c for c in range(nc) if ts[c] < CH[1] < te[c] + 4000][:3]                # stretch a few boxes across the chromosome boundary
    for c in rej: te[c] = max(te[c], CH[1] + 50)
    tt = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    d_cnt = tt(cnt, np.int32); d_ts = tt(ts.astype(np.int32), np.int32); d_te = tt(te.astype(np.int32), np.int32)
    res = chain.refine_clusters_batch(ctx, n, cres.d_cluster_off, cres.d_c_start, d_cnt, cres.d_c_strand, cres.d_c_qStart, cres.d_c_qEnd, d_ts, d_te,
                                      cres.d_cl_qpos, cres.d_cl_tpos, int(cres.n_matches), batch.off, CH, rli, gso_d, gli, window=100, smallK=10, K=K, max_freq=15)
    out = chain.fetch_refined_clusters(ctx, res)
    coff = co["cluster_off"]
    n_ok = n_rev = n_rej = n_matches = 0
    for r in range(n):
        for c in range(int(coff[r]), int(coff[r + 1])):
            a, b = int(co["start"][c]), int(co["end"][c])
            strand = int(co["strand"][c])
            w0, w1 = int(r_win[strand * n + r]), int(r_win[strand * n + r + 1])
            q_index = (_seq_offsets([0], lens[r], 256), r_bnd[w0:w1 + 1] - r_bnd[w0], r_tup[int(r_bnd[w0]):int(r_bnd[w1])])
            exp = O.refine_cluster(co["cl_qpos"][a:b], co["cl_tpos"][a:b], [co["qStart"][c], co["qEnd"][c], ts[c], te[c]], strand, CH, lens[r], q_index,
                                   (gso, g_bnd, g_tup), window=100, smallK=10, K=K, max_freq=15)
            m0, m1 = int(out["match_off"][c]), int(out["match_off"][c + 1])
            if exp == "rejected":
                assert out["status"][c] == 16 and m1 == m0, c
                n_rej += 1
                continue
            if exp is None:
                assert out["status"][c] == 1 and m1 == m0, c
                continue
            assert out["status"][c] == 0, (c, out["status"][c])
            assert m1 - m0 == len(exp["q"]), (c, m1 - m0, len(exp["q"]))
            assert np.array_equal(out["match_q"][m0:m1], exp["q"]) and np.array_equal(out["match_t"][m0:m1], exp["t"]), c
            assert out["chrom"][c] == exp["chrom"]
            if m1 > m0:
                assert np.array_equal(out["box"][c], exp["box"]) and out["eff"][c].view(np.uint32) == exp["eff"].view(np.uint32), c
            n_ok += 1; n_rev += strand; n_matches += m1 - m0
    assert n_ok >= 100 and n_rev >= 20 and n_matches > 20000, (n_ok, n_rev, n_rej, n_matches)


def _second_sdp_chains(ctx, oracle, P):
    """front end -> ... -> second sparse DP + its filters; returns everything the a13 test needs (jobs = chain slots)"""
    import torch
    from lra_amd import chain
    chres, spres, batch, CH, rli, gso_d, gli, co, so, n, na, both, tot, gdev = (P[k] for k in (
        "chres", "spres", "batch", "CH", "rli", "gso_d", "gli", "co", "so", "n", "na", "both", "tot", "gdev"))
    rres = chain.refine_splitchain_batch(ctx, chres, spres, batch.off, CH, rli, gso_d, gli, window=100, smallK=10, K=K, limitrefine=True, max_freq=15)
    bres = chain.refine_btwn_splitchain_batch(ctx, chres, spres, rres, batch.off, both, tot, gdev, CH, K=10, W=5)
    mres = chain.merge_extend_batch(ctx, chres, spres, bres, batch.seq, batch.off, gdev, CH, K=10)
    mo = chain.fetch_merge(ctx, mres)
    cres2 = chain.sparse_dp_batch(ctx, int(mres.n_groups), mres.d_iota, mres.d_anchor_off, mres.d_count, mres.d_strand, mres.d_q, mres.d_t, mres.d_len,
                                  mres.d_iota, chain.sdp_opts(mode=1, rate=2.0))
    c2 = chain.fetch(ctx, cres2)
    jobs = []                                                              # (read, h, [chains]) with chain = (q, t, len, strand, chrom, value)
    for r in range(n):
        for c in range(int(co["n_chains"][r])):
            s = r * na + c
            chains = []
            for G in range(int(mo["slot_group_off"][s]), int(mo["slot_group_off"][s + 1])):
                cnt = int(mo["count"][G])
                if cnt == 0 or c2["status"][G] or int(c2["n_chains"][G]) == 0:
                    continue
                a0 = int(mo["anchor_off"][G])
                cs_ = int(c2["chain_start"][G * cres2.num_aln]); m = int(c2["chain_len"][G * cres2.num_aln])
                idx = c2["chain_anchor"][cs_:cs_ + m].astype(np.int64)
                q = mo["q"][a0 + idx]; t = mo["t"][a0 + idx]; ln = mo["len"][a0 + idx]
                st = int(mo["strand"][G])
                keep, _ = O.filter_chain(q, t, ln, [st] * m, None, [2, 4])    # RemovePairedIndels + RemoveSpuriousAnchors (Map_lowacc.h:538-539)
                kb = keep.astype(bool)
                chains.append((q[kb], t[kb], ln[kb], st, int(mo["chrom"][G]), float(c2["chain_value"][G * cres2.num_aln]), m))
            if chains:
                jobs.append((r, c, chains))
    P["_mres"] = mres; P["_cres2"] = cres2; P["_mo"] = mo
    return jobs


@pytest.mark.gpu
def test_hip_local_refine_alignment_oracle(ctx, oracle):
    """a13: LocalRefineAlignment + RefinedAlignmentbtwnAnchors on the chains of the second sparse DP (reads with inversions, deletions, junk
    insertions so that large spaces, breaks and inverted seeds occur), against the oracle job by job"""
    import torch
    from lra_amd import chain, synth
    P = _front_end(ctx, oracle)
    jobs = _second_sdp_chains(ctx, oracle, P)
    reads, genome, batch, both, tot, gdev, CH = (P[k] for k in ("reads", "genome", "batch", "both", "tot", "gdev", "CH"))
    dev = ctx.device
    jco = [0]; jr = []; jh = []; cao = [0]; cs = []; cc = []; cv = []; c0 = []; c1 = []; Q = []; T = []; Ln = []
    for (r, h, chains) in jobs:
        for (q, t, ln, st, ch, val, _m) in chains:
            Q.extend(q.tolist()); T.extend(t.tolist()); Ln.extend(ln.tolist()); cao.append(len(Q)); cs.append(st); cc.append(ch); cv.append(val); c0.append(len(q)); c1.append(7)
        jco.append(len(cs)); jr.append(r); jh.append(h)
    tt = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    res = chain.local_refine_batch(ctx, tt(jco, np.int64), tt(jr, np.int32), tt(jh, np.int32), tt(cao, np.int64), tt(cs, np.int32), tt(cc, np.int32), tt(cv, np.float32),
                                   tt(c0, np.int32), tt(c1, np.int32), tt(Q, np.int64).to(torch.int32), tt(T, np.int64).to(torch.int32), tt(Ln, np.int32), batch.off, both, tot,
                                   gdev, CH)
    out = chain.fetch_alignments(ctx, res)
    gbytes = genome.tobytes()
    n_jobs = n_aln = n_events = n_blocks = 0
    ci = 0
    for ji, (r, h, chains) in enumerate(jobs):
        fwd = reads[r].tobytes(); rc = synth.revcomp(reads[r]).tobytes()
        off = [0]; aq = []; at = []; al = []
        for (q, t, ln, st, ch, val, _m) in chains:
            aq.extend(q.tolist()); at.extend(t.tolist()); al.extend(ln.tolist()); off.append(len(aq))
        nch = len(chains)
        exp = O.local_refine_alignment(off, aq, at, al, [x[3] for x in chains], [x[4] for x in chains], [x[5] for x in chains], c0[ci:ci + nch], c1[ci:ci + nch], h, fwd, rc,
                                       gbytes, CH)
        ci += nch
        a0, a1 = int(out["job_aln_off"][ji]), int(out["job_aln_off"][ji + 1])
        if exp is None:
            assert out["status"][ji] != 0, ji
            continue
        assert out["status"][ji] == 0, (ji, out["status"][ji])
        assert a1 - a0 == len(exp), (ji, a1 - a0, len(exp))
        for k, e in enumerate(exp):
            x = a0 + k
            for f in ("strand", "supp", "secondary", "n0", "n1", "chrom"):
                assert int(out[f][x]) == e[f], (ji, k, f, int(out[f][x]), e[f])
            assert np.float32(out["value"][x]).view(np.uint32) == np.float32(e["value"]).view(np.uint32), (ji, k)
            b0, b1 = int(out["block_off"][x]), int(out["block_off"][x + 1])
            assert b1 - b0 == len(e["blocks"]), (ji, k, b1 - b0, len(e["blocks"]))
            assert np.array_equal(out["blocks"][b0:b1], e["blocks"]), (ji, k)
            n_blocks += b1 - b0
        n_jobs += 1; n_aln += len(exp); n_events += len(exp) - sum(1 for c_ in chains if len(c_[0]) > 1)
    print("a13 stats: jobs %d alignments %d events %d blocks %d large spaces %d seed-set jobs %d" % (n_jobs, n_aln, n_events, n_blocks, res.n_big, res.n_inner_jobs))
    assert n_jobs >= 40 and n_blocks > 10000 and res.n_big >= 5, (n_jobs, n_aln, n_events, n_blocks, res.n_big, res.n_inner_jobs)


@pytest.mark.gpu
def test_hip_local_refine_events_oracle(ctx):
    """a13 on crafted chains: an inverted stretch between two anchors (inverted seeds -> an inversion alignment), junk of 800 / 400 bases on both
    sides (no seed on either strand -> the alignment breaks; the K = 9 / maxFreq 50 branch), mutated stretches (forward seeds -> the inner
    chain), on both strands, near the start of a chain (fewer than 5 blocks: no inversion is tried) and later"""
    import torch
    from lra_amd import chain, synth, seed
    rng = np.random.default_rng(123)
    genome = synth.make_genome(400_000, seed=77, repeat_frac=0.1, n_families=2)
    ALPH = np.frombuffer(b"ACGT", np.uint8)
    CH = [0, 180_000, 400_000]
    reads = []; jobs = []
    for j in range(40):
        kind = j % 5; rev = (j // 5) % 2; early = (j // 10) % 2
        chrom = int(rng.integers(0, 2))
        a = CH[chrom] + int(rng.integers(5_000, 150_000))
        Lr = 5000
        seg = genome[a:a + Lr].copy()
        e0 = 150 if early else 2200                                       # where the event sits (read-strand coordinate of the alignment)
        span = {0: 1500, 1: 1500, 2: 400, 3: 900, 4: 0}[kind]                  # >= 1000 on both sides: the minimizer path of RefineSpace, no chance seeds
        if kind == 0: seg[e0:e0 + span] = synth.revcomp(genome[a + e0:a + e0 + span])                 # inversion
        elif kind in (1, 2): seg[e0:e0 + span] = ALPH[rng.integers(0, 4, span)]                          # junk on both sides
        elif kind == 3:                                                                                  # 12 % substitutions: seeds survive
            m_ = rng.random(span) < 0.12
            seg[e0:e0 + span][m_] = ALPH[rng.integers(0, 4, int(m_.sum()))]
        # anchors: exact 40-mers every 55 bases outside the event, in the alignment's own (read-strand) coordinates
        anchors = [(x, x, 40) for x in range(20, Lr - 60, 55) if x + 40 <= e0 - 45 or x >= e0 + span + 45]
        if span: anchors += [(e0 - 40, e0 - 40, 40), (e0 + span, e0 + span, 40)]     # anchors flush with the event: the space is the event itself
        read = synth.revcomp(seg) if rev else seg
        reads.append(read)
        L = len(read)
        chrom_off = CH[chrom]
        ch = []
        for (qr, tr, ln) in sorted(anchors, reverse=True):               # chain order: highest forward q first
            t = a + tr - chrom_off
            q = L - qr - ln if rev else qr
            ch.append((q, t, ln))
        if rev: ch = ch[::-1]                                              # strand 1: descending forward q = ascending strand coordinate
        ch = sorted(ch, key=lambda x: -x[0])
        jobs.append((j, int(early), [(np.array([c[0] for c in ch], np.uint32), np.array([c[1] for c in ch], np.uint32), np.array([c[2] for c in ch], np.int32), rev, chrom,
                                      float(100 + j))]))
    batch = seed.ReadBatch(ctx, [r.tobytes() for r in reads])
    dev = ctx.device
    tot = int(batch.off[-1])
    both = torch.zeros(2 * tot + 64, dtype=torch.uint8, device=dev)
    both[:tot] = batch.seq[:tot]; both[tot:2 * tot] = seed.create_rc(ctx, batch)[:tot]
    gdev = torch.from_numpy(np.concatenate([genome, np.zeros(64, np.uint8)])).to(dev)
    jco = [0]; jr = []; jh = []; cao = [0]; cs = []; cc = []; cv = []; c0 = []; c1 = []; Q = []; T = []; Ln = []
    for (r, h, chains) in jobs:
        for (q, t, ln, st, chm, val) in chains:
            Q.extend(q.tolist()); T.extend(t.tolist()); Ln.extend(ln.tolist()); cao.append(len(Q)); cs.append(st); cc.append(chm); cv.append(val); c0.append(len(q)); c1.append(3)
        jco.append(len(cs)); jr.append(r); jh.append(h)
    tt = lambda a_, dt: torch.from_numpy(np.ascontiguousarray(a_, dtype=dt)).to(dev)
    res = chain.local_refine_batch(ctx, tt(jco, np.int64), tt(jr, np.int32), tt(jh, np.int32), tt(cao, np.int64), tt(cs, np.int32), tt(cc, np.int32), tt(cv, np.float32),
                                   tt(c0, np.int32), tt(c1, np.int32), tt(Q, np.int64).to(torch.int32), tt(T, np.int64).to(torch.int32), tt(Ln, np.int32), batch.off, both, tot,
                                   gdev, CH)
    out = chain.fetch_alignments(ctx, res)
    gbytes = genome.tobytes()
    n_inv = n_multi = 0
    for ji, (r, h, chains) in enumerate(jobs):
        q, t, ln, st, chm, val = chains[0]
        exp = O.local_refine_alignment([0, len(q)], q, t, ln, [st], [chm], [val], [len(q)], [3], h, reads[r].tobytes(), synth.revcomp(reads[r]).tobytes(), gbytes, CH)
        a0, a1 = int(out["job_aln_off"][ji]), int(out["job_aln_off"][ji + 1])
        assert exp is not None and out["status"][ji] == 0, ji
        assert a1 - a0 == len(exp), (ji, ji % 5, a1 - a0, len(exp))
        for k, e in enumerate(exp):
            x = a0 + k
            for f in ("strand", "supp", "secondary", "n0", "n1", "chrom"):
                assert int(out[f][x]) == e[f], (ji, k, f, int(out[f][x]), e[f])
            assert np.float32(out["value"][x]).view(np.uint32) == np.float32(e["value"]).view(np.uint32), (ji, k)
            b0, b1 = int(out["block_off"][x]), int(out["block_off"][x + 1])
            assert b1 - b0 == len(e["blocks"]) and np.array_equal(out["blocks"][b0:b1], e["blocks"]), (ji, k)
            n_inv += e["strand"] != st
        n_multi += len(exp) > 1
    print("a13 events: jobs with several alignments %d, inverted alignments %d, large spaces %d, seed-set jobs %d" % (n_multi, n_inv, res.n_big, res.n_inner_jobs))
    assert n_multi >= 8 and n_inv >= 2 and res.n_big >= 24


@pytest.mark.gpu
def test_hip_local_refine_from_sdp_on_device(ctx, oracle):
    """the same walk fed on the device: second sparse DP -> filters {2, 4} -> job / chain arrays (lra_local_refine_inputs_batch) -> a13, no host
    round trip; checked against the oracle with the host-built chains"""
    import torch
    from lra_amd import chain, synth
    P = _front_end(ctx, oracle)
    co, na, n = P["co"], P["na"], P["n"]
    slot_n0 = torch.from_numpy(co["chain_len"].astype(np.int32)).to(ctx.device)          # kept before the second sparse DP reuses the result buffers
    jobs = _second_sdp_chains(ctx, oracle, P)
    reads, genome, batch, both, tot, gdev, CH = (P[k] for k in ("reads", "genome", "batch", "both", "tot", "gdev", "CH"))
    inp, res = chain.local_refine_from_sdp(ctx, na, slot_n0, P["_mres"], P["_cres2"], batch.off, both, tot, gdev, CH)
    out = chain.fetch_alignments(ctx, res)
    assert int(res.n_jobs) == n * na
    gbytes = genome.tobytes()
    by_slot = {r * na + h: chains for (r, h, chains) in jobs}
    n_checked = 0
    for s in range(n * na):
        a0, a1 = int(out["job_aln_off"][s]), int(out["job_aln_off"][s + 1])
        chains = by_slot.get(s)
        if chains is None:
            assert a1 == a0, s
            continue
        r, h = s // na, s % na
        # the device keeps empty / unchained merged clusters as empty chains: they only shift LSC's index, never its choice (size 0 never wins)
        off = [0]; aq = []; at = []; al = []
        for (q, t, ln, st, ch, val, m) in chains:
            aq.extend(q.tolist()); at.extend(t.tolist()); al.extend(ln.tolist()); off.append(len(aq))
        exp = O.local_refine_alignment(off, aq, at, al, [x[3] for x in chains], [x[4] for x in chains], [x[5] for x in chains], [int(co["chain_len"][s])] * len(chains),
                                       [x[6] for x in chains], h, reads[r].tobytes(), synth.revcomp(reads[r]).tobytes(), gbytes, CH)
        assert exp is not None and out["status"][s] == 0 and a1 - a0 == len(exp), (s, a1 - a0, None if exp is None else len(exp))
        for k, e in enumerate(exp):
            x = a0 + k
            for f in ("strand", "secondary", "n0", "n1", "chrom"):
                assert int(out[f][x]) == e[f], (s, k, f, int(out[f][x]), e[f])
            b0, b1 = int(out["block_off"][x]), int(out["block_off"][x + 1])
            assert b1 - b0 == len(e["blocks"]) and np.array_equal(out["blocks"][b0:b1], e["blocks"]), (s, k)
        n_checked += 1
    assert n_checked >= 40
