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
    # ---- a7: LinearExtend + DecideCoordinates on those clusters
    eres = cluster.linear_extend_batch(ctx, k, batch)
    eo = cluster.fetch_extend(ctx, eres)
    cp = np.asarray(chrom_pos, dtype=np.int64)
    n_ext = 0
    for r, read in enumerate(reads):
        rb = read.tobytes()
        for x in range(int(out["cluster_off"][r]), int(out["cluster_off"][r + 1])):
            a, b = int(out["start"][x]), int(out["end"][x])
            ch = int(out["chrom"][x]); off = int(cp[ch])
            chrom = genome[off:int(cp[ch + 1])].tobytes()
            eq, et, el, box = oracle.linear_extend(out["cl_qpos"][a:b], out["cl_tpos"][a:b] - np.uint32(off), int(out["strand"][x]), k, rb, chrom)
            es, ec = int(eo["e_start"][x]), int(eo["e_count"][x])
            assert ec == len(eq), (r, x, ec, len(eq))
            assert np.array_equal(eo["e_qpos"][es:es + ec], eq) and np.array_equal(eo["e_tpos"][es:es + ec], et + np.uint32(off))
            assert np.array_equal(eo["e_len"][es:es + ec], el)
            assert eo["box"][x].tolist() == [int(box[0]), int(box[1]), int(box[2]) + off, int(box[3]) + off]
            n_ext += ec
    assert n_ext > 50


def test_oracle_linear_extend_sanity(oracle):
    g = synth.make_genome(5000, seed=3).tobytes()
    read = g[1000:1400]
    # three 17-mers on one diagonal, exact matches in between -> one anchor covering all; then a diagonal change
    q = [0, 40, 120, 200]; t = [1000, 1040, 1120, 1210]
    eq, et, el, box = oracle.linear_extend(q, t, 0, 17, read, g)
    assert eq.tolist() == [0, 200] and et.tolist() == [1000, 1210] and el.tolist() == [137, 17]
    assert box.tolist() == [0, 217, 1000, 1227]
    assert len(oracle.linear_extend([], [], 0, 17, read, g)[0]) == 0
    assert oracle.linear_extend([5], [1005], 0, 17, read, g)[2].tolist() == [17]
