"""a1-a4 tier-1 seeding: oracle vs reference golden (CPU) and HIP vs oracle (GPU)."""
import json
import os

import numpy as np
import pytest

from lra_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
M63 = np.uint64((1 << 63) - 1)


def _golden():
    return json.load(open(os.path.join(HERE, "golden", "comparelists_golden.json")))["cases"]


def test_oracle_comparelists_and_sort_match_reference_golden(oracle):
    """Restated CompareLists + std::sort reproduce what the reference's own CompareLists.h /
    std::sort(readmm) (compiled in place, oracle/_ref/comparelists_ref) emitted."""
    for c in _golden():
        q = np.array(c["q"], dtype=np.uint64).reshape(-1, 2)
        t = np.array(c["t"], dtype=np.uint64).reshape(-1, 2)
        qk, qp = oracle.sort_minimizers(q[:, 0], q[:, 1].astype(np.uint32))
        assert [int((p - 1) // 3) for p in qp] == c["sorted_perm"]
        oq, ot = oracle.compare_lists(qk, qp, t[:, 0], t[:, 1].astype(np.uint32), c["maxFreq"])
        assert np.stack([oq, ot], 1).flatten().tolist() == c["pairs"]


def test_oracle_minimizers_basic_properties(oracle):
    """(parity unpinned for a1) sanity: every emitted tuple is the canonical k-mer at its position and
    is a minimum (by masked key) of some w-window containing it; short / N-only inputs give nothing."""
    rng = np.random.default_rng(5)
    seq = synth.BASES[rng.integers(0, 4, size=5000)].tobytes()
    k, w = 17, 10
    keys, pos = oracle.store_minimizers(seq, k, w)
    ck, cs = synth.canonical_keys(np.frombuffer(seq, dtype=np.uint8), k)
    assert len(keys) > 0
    assert np.all((keys & M63) == ck[pos])
    assert np.all(((keys >> np.uint64(63)) == 1) == cs[pos])
    assert np.all(np.diff(pos.astype(np.int64)) > 0)
    for p in pos[1:]:
        lo, hi = max(0, int(p) - w + 1), int(p)
        assert any(ck[s:s + w].min() == ck[p] for s in range(lo, hi + 1) if s + w <= len(ck))
    assert len(oracle.store_minimizers(b"ACGT" * 6, k, w)[0]) == 0     # shorter than a window span... 24 < 26
    assert len(oracle.store_minimizers(b"N" * 500, k, w)[0]) == 0
    # sequence whose only valid stretch starts after an N run
    s2 = b"N" * 40 + seq[:300]
    k2, p2 = oracle.store_minimizers(s2, k, w)
    assert len(k2) > 0 and p2.min() >= 40


def _oracle_pipeline(oracle, read, genome, ik, ip, k, w, max_freq):
    keys, pos = oracle.store_minimizers(read, k, w)
    sk, sp = oracle.sort_minimizers(keys, pos)
    qi, ti = oracle.compare_lists(sk, sp, ik, ip, max_freq)
    strand = oracle.separate_strand(read, genome, k, sp[qi], ip[ti])
    return sk, sp, qi, ti, strand


def _check_batch(ctx, oracle, genome, reads, k, w, max_freq, index_max_freq=50):
    from lra_amd import seed
    ik, ip = synth.build_global_index(genome, k, w, index_max_freq)
    seed.load_reference(ctx, genome, ik, ip)
    batch = seed.ReadBatch(ctx, [r.tobytes() if hasattr(r, "tobytes") else r for r in reads])
    res = seed.seed_batch(ctx, batch, k, w, max_freq)
    out = seed.fetch(ctx, res)
    gbytes = genome.tobytes() + b"\0" * 64
    n_match = 0
    for r, read in enumerate(reads):
        rb = read.tobytes() if hasattr(read, "tobytes") else read
        sk, sp, qi, ti, strand = _oracle_pipeline(oracle, rb, gbytes, ik, ip, k, w, max_freq)
        a, b = int(out["mm_off"][r]), int(out["mm_off"][r + 1])
        assert b - a == len(sk), (r, b - a, len(sk))
        assert np.array_equal(out["mm_key"][a:b], sk), r
        assert np.array_equal(out["mm_pos"][a:b], sp), r
        m0, m1 = int(out["match_off"][r]), int(out["match_off"][r + 1])
        assert m1 - m0 == len(qi), (r, m1 - m0, len(qi))
        assert np.array_equal(out["match_qi"][m0:m1], qi), r
        assert np.array_equal(out["match_ti"][m0:m1], ti), r
        nf = int((strand == 0).sum())
        assert int(out["n_forward"][r]) == nf
        eq, et = sp[qi], ip[ti]
        assert np.array_equal(out["sep_qpos"][m0:m0 + nf], eq[strand == 0])
        assert np.array_equal(out["sep_tpos"][m0:m0 + nf], et[strand == 0])
        assert np.array_equal(out["sep_qpos"][m0 + nf:m1], eq[strand == 1])
        assert np.array_equal(out["sep_tpos"][m0 + nf:m1], et[strand == 1])
        n_match += len(qi)
    return n_match


@pytest.mark.gpu
@pytest.mark.parametrize("k,w,max_freq,err", [(17, 10, 150, 0.10), (15, 10, 250, 0.15), (25, 20, 150, 0.01)])
def test_hip_seed_matches_oracle(ctx, oracle, k, w, max_freq, err):
    genome = synth.make_genome(400000, seed=2, repeat_frac=0.4)
    reads, _ = synth.simulate_reads(genome, 70, 6000, 2500, err, seed=k)
    reads += [np.frombuffer(b"", dtype=np.uint8), np.frombuffer(b"ACGTACGTAC", dtype=np.uint8),
              np.frombuffer(b"N" * 300, dtype=np.uint8), genome[1000:1000 + w + k - 1].copy(),
              np.concatenate([genome[5000:5400], np.frombuffer(b"NNNNN", dtype=np.uint8), genome[5400:6000]])]
    n = _check_batch(ctx, oracle, genome, reads, k, w, max_freq)
    assert n > 1000


@pytest.mark.gpu
def test_hip_seed_repeats_and_ties(ctx, oracle):
    """Tandem repeats / low complexity: many equal minimizers with both strand flags, which is where
    the introsort permutation and CompareLists' raw-key run skipping become visible."""
    rng = np.random.default_rng(9)
    unit = synth.BASES[rng.integers(0, 4, size=37)]
    pal = np.concatenate([unit, synth.revcomp(unit)])
    genome = np.concatenate([synth.make_genome(60000, seed=4, repeat_frac=0.1), np.tile(pal, 60), synth.make_genome(60000, seed=5)])
    reads = []
    for i in range(40):
        s = int(rng.integers(55000, 66000))
        r = genome[s:s + int(rng.integers(800, 5000))].copy()
        mut = rng.random(len(r)) < 0.03
        r[mut] = synth.BASES[rng.integers(0, 4, size=int(mut.sum()))]
        reads.append(synth.revcomp(r) if i % 2 else r)
    reads.append(np.tile(np.frombuffer(b"A", dtype=np.uint8), 500))
    reads.append(np.tile(np.frombuffer(b"AC", dtype=np.uint8), 300))
    n = _check_batch(ctx, oracle, genome, reads, 11, 5, 8, index_max_freq=400)
    assert n > 100


@pytest.mark.gpu
def test_hip_create_rc(ctx):
    from lra_amd import seed
    reads = [b"ACGTNacgtnXY", b"", b"A", b"GATTACA" * 50]
    b = seed.ReadBatch(ctx, reads)
    rc = seed.create_rc(ctx, b).cpu().numpy()
    comp = {ord(a): ord(c) for a, c in zip("ACGTacgtn", "TGCAtgcan")}
    for i, r in enumerate(reads):
        exp = bytes(comp.get(c, ord("N")) for c in reversed(r))
        assert rc[int(b.off_h[i]):int(b.off_h[i + 1])].tobytes() == exp
