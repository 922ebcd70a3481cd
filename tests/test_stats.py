"""a16 CalculateStatistics: oracle vs reference golden (CPU) and HIP vs golden / oracle (GPU)."""
import json
import os
import struct

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "stats_golden.json")))["cases"]


def _bits(v):
    return struct.unpack("I", struct.pack("f", float(v)))[0]


def test_oracle_matches_reference_golden(oracle):
    for c in GOLD:
        cnt, val, runs, cigar = oracle.calculate_statistics(c["blocks"], c["read"].encode(), c["genome"].encode())
        assert cigar == c["cigar"]
        assert _bits(val) == c["out"]["value_bits"]
        for k, v in cnt.items():
            assert v == c["out"][k], k


@pytest.mark.gpu
def test_hip_matches_reference_golden(ctx, oracle):
    import torch
    from lra_amd import refine
    reads = [np.frombuffer(c["read"].encode(), dtype=np.uint8) for c in GOLD]
    gens = [np.frombuffer(c["genome"].encode(), dtype=np.uint8) for c in GOLD]
    rl = np.array([len(r) for r in reads], dtype=np.int64); gl = np.array([len(g) for g in gens], dtype=np.int64)
    qoff = np.concatenate([[0], np.cumsum(rl)[:-1]]); toff = np.concatenate([[0], np.cumsum(gl)[:-1]])
    qdev = torch.from_numpy(np.concatenate(reads + [np.zeros(64, np.uint8)])).to(ctx.device)
    tdev = torch.from_numpy(np.concatenate(gens + [np.zeros(64, np.uint8)])).to(ctx.device)
    blocks = [np.asarray(c["blocks"], dtype=np.int32).reshape(-1, 3) for c in GOLD]
    b = refine.RefineBatch(ctx, blocks, qdev, qoff, rl.astype(np.int32), tdev, toff, gl)
    res = refine.calculate_statistics_batch(ctx, b, oracle.log_lookup_table())
    counts, value, cigars = refine.fetch_stats(ctx, res)
    for i, c in enumerate(GOLD):
        assert cigars[i] == c["cigar"], i
        assert _bits(value[i]) == c["out"]["value_bits"], i
        for j, k in enumerate(refine.STAT_NAMES):
            assert int(counts[i, j]) == c["out"][k], (i, k)


@pytest.mark.gpu
def test_hip_stats_on_refined_long_reads(ctx, oracle):
    """30 kb alignments straight out of the indel-refinement stage."""
    import torch
    from lra_amd import refine, synth
    from test_refine import make_cases
    genome = synth.make_genome(300000, seed=41)
    reads, blocks = make_cases(9, 12, 25000, 0.10, (30, 35, 35), genome)
    gdev = torch.from_numpy(np.concatenate([genome, np.zeros(64, np.uint8)])).to(ctx.device)
    lens = np.array([len(r) for r in reads], dtype=np.int64)
    qoff = np.concatenate([[0], np.cumsum(lens)[:-1]])
    qdev = torch.from_numpy(np.concatenate(list(reads) + [np.zeros(64, np.uint8)])).to(ctx.device)
    b = refine.RefineBatch(ctx, blocks, qdev, qoff, lens.astype(np.int32), gdev, np.zeros(len(reads), np.int64), np.full(len(reads), len(genome), np.int64))
    rres = refine.indel_refine_batch(ctx, b, 7, 4, -1, -2)
    rblocks, status = refine.fetch(ctx, rres)
    assert not status.any()
    b2 = refine.RefineBatch(ctx, rblocks, qdev, qoff, lens.astype(np.int32), gdev, np.zeros(len(reads), np.int64), np.full(len(reads), len(genome), np.int64))
    res = refine.calculate_statistics_batch(ctx, b2, oracle.log_lookup_table())
    counts, value, cigars = refine.fetch_stats(ctx, res)
    g = genome.tobytes()
    for i, (r, bl) in enumerate(zip(reads, rblocks)):
        cnt, val, runs, cigar = oracle.calculate_statistics(bl, r.tobytes(), g)
        assert cigars[i] == cigar
        assert _bits(value[i]) == _bits(val)
        assert [int(x) for x in counts[i]] == [cnt[k] for k in refine.STAT_NAMES]
