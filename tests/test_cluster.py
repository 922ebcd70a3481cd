"""a5 CleanMatches: oracle sanity (CPU; parity unpinned) and HIP vs oracle (GPU)."""
import numpy as np
import pytest

from lra_amd import synth


def _oracle_seed(oracle, rb, g, ik, ip, k, w, mf):
    keys, pos = oracle.store_minimizers(rb, k, w)
    sk, sp = oracle.sort_minimizers(keys, pos)
    qi, ti = oracle.compare_lists(sk, sp, ik, ip, mf)
    st = oracle.separate_strand(rb, g, k, sp[qi], ip[ti])
    return sp[qi], ip[ti], sk[qi], st


def test_oracle_clean_sanity(oracle):
    genome = synth.make_genome(300000, seed=12, repeat_frac=0.3)
    ik, ip = synth.build_global_index(genome, 17, 10, 50)
    reads, truth = synth.simulate_reads(genome, 5, 15000, 1000, 0.10, seed=4)
    g = genome.tobytes() + b"\0" * 64
    opts = oracle.CleanOpts(**oracle.CLEAN_PRESETS["ONT"])
    for r, (s0, L, strand) in zip(reads, truth):
        q, t, key, st = _oracle_seed(oracle, r.tobytes(), g, ik, ip, 17, 10, 150)
        sel = st == strand
        oq, ot, cl = oracle.clean_matches(q[sel], t[sel], key[sel], strand, opts, [0, len(genome)])
        assert len(cl["start"]) >= 1
        big = np.argmax(cl["end"] - cl["start"])
        assert cl["tStart"][big] >= s0 - 50 and cl["tEnd"][big] <= s0 + L + 50          # the main cluster is the true locus
        d = (ot.astype(np.int64) - oq) if strand == 0 else (ot.astype(np.int64) + oq)
        for a, b in zip(cl["start"], cl["end"]):
            assert np.all(np.abs(np.diff(d[a:b])) < 200)                                 # cleanMaxDiag inside a cluster
        assert np.all(cl["chrom"] == 0)
    assert oracle.clean_matches([], [], [], 0, opts, [0, 10])[2]["start"].size == 0
    assert oracle.clean_matches([5], [100], [1], 0, opts, [0, 1000])[2]["start"].size == 0   # a single match has no neighbour


@pytest.mark.gpu
@pytest.mark.parametrize("preset,k,w,mf,err", [("ONT", 17, 10, 150, 0.10), ("CLR", 15, 10, 250, 0.15), ("CCS", 17, 10, 150, 0.01)])
def test_hip_clean_matches_oracle(ctx, oracle, preset, k, w, mf, err):
    from lra_amd import seed, cluster
    genome = synth.make_genome(500000, seed=31, repeat_frac=0.5, n_families=3)
    ik, ip = synth.build_global_index(genome, k, w, 60)
    reads, _ = synth.simulate_reads(genome, 60, 9000, 3000, err, seed=k + 1)
    reads += [np.frombuffer(b"ACGT" * 5, dtype=np.uint8), genome[100:100 + 3000].copy()]
    seed.load_reference(ctx, genome, ik, ip)
    batch = seed.ReadBatch(ctx, [r.tobytes() for r in reads])
    sres = seed.seed_batch(ctx, batch, k, w, mf)
    so = seed.fetch(ctx, sres)
    po = dict(oracle.CLEAN_PRESETS[preset]); po["globalK"] = k
    chrom_pos = [0, 200000, len(genome)]                      # two "chromosomes"
    res = cluster.clean_matches_batch(ctx, cluster.CleanOpts(**po), chrom_pos)
    out = cluster.fetch(ctx, res)
    g = genome.tobytes() + b"\0" * 64
    oopts = oracle.CleanOpts(**po)
    total = 0
    for r, read in enumerate(reads):
        q, t, key, st = _oracle_seed(oracle, read.tobytes(), g, ik, ip, k, w, mf)
        m0 = int(so["match_off"][r]); nf = int(so["n_forward"][r]); m1 = int(so["match_off"][r + 1])
        exp = []
        for strand, base in ((0, m0), (1, m0 + nf)):
            sel = st == strand
            oq, ot, cl = oracle.clean_matches(q[sel], t[sel], key[sel], strand, oopts, chrom_pos)
            for i in range(len(cl["start"])):
                a, b = int(cl["start"][i]), int(cl["end"][i])
                exp.append((strand, int(cl["qStart"][i]), int(cl["qEnd"][i]), int(cl["tStart"][i]), int(cl["tEnd"][i]), int(cl["chrom"][i]),
                            float(cl["freq"][i]), oq[a:b].tolist(), ot[a:b].tolist()))
        c0, c1 = int(out["cluster_off"][r]), int(out["cluster_off"][r + 1])
        assert c1 - c0 == len(exp), (r, c1 - c0, len(exp))
        for x, e in zip(range(c0, c1), exp):
            a, b = int(out["start"][x]), int(out["end"][x])
            got = (int(out["strand"][x]), int(out["qStart"][x]), int(out["qEnd"][x]), int(out["tStart"][x]), int(out["tEnd"][x]),
                   int(out["chrom"][x]), float(out["freq"][x]), out["cl_qpos"][a:b].tolist(), out["cl_tpos"][a:b].tolist())
            assert got == e, (r, x)
        total += len(exp)
    assert total > 50
