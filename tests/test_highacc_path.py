"""The high-accuracy path (MapRead_highacc, Map_highacc.h:37-798; -CCS / -CONTIG): the leaves that were missing after round 1 and the path end to end.
a5: MatchesToFineClusters (Clustering.h:1555; SplitRoughClustersWithGaps :1358, StoreFineClusters :892).
Oracle restatements are PARITY UNPINNED (Clustering.h & co need htslib headers); CPU tests check their properties, GPU tests compare HIP with them."""
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
