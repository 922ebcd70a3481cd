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


@pytest.mark.gpu
def test_hip_stats_overlapping_and_empty_blocks(ctx, oracle):
    """Blocks that overlap their successor in the read or the genome (negative gaps), blocks of length 0, gaps longer than 20 / 50 (the float chain of `value`):
    the walk's own q / t then differ from the blocks' coordinates; the parallel form takes them as prefix sums of the advances."""
    import torch
    from lra_amd import refine
    rng = np.random.default_rng(17)
    n_aln = 48
    L = 6000
    reads = [rng.integers(0, 4, L).astype(np.uint8) for _ in range(n_aln)]
    gens = []
    blocks = []
    for a in range(n_aln):
        g = reads[a].copy()
        mut = rng.random(L) < 0.08
        g[mut] = (g[mut] + 1 + rng.integers(0, 3, int(mut.sum()))) % 4
        gens.append(g)
        q, t = int(rng.integers(0, 50)), int(rng.integers(0, 50))
        bl = []
        nb = int(rng.integers(1, 160))
        for b in range(nb):
            ln = int(rng.integers(0, 3)) if rng.random() < 0.06 else int(rng.integers(1, 90))
            if q + ln >= L - 200 or t + ln >= L - 200:
                break
            bl.append((q, t, ln))
            u = rng.random()
            if u < 0.12:
                dq, dt = -int(rng.integers(1, max(1, min(ln, 6)) + 1)), int(rng.integers(0, 8))          # the next block starts inside this one (read side)
            elif u < 0.24:
                dq, dt = int(rng.integers(0, 8)), -int(rng.integers(1, max(1, min(ln, 6)) + 1))          # ... genome side
            elif u < 0.30:
                dq, dt = -int(rng.integers(1, max(1, min(ln, 4)) + 1)), -int(rng.integers(1, max(1, min(ln, 4)) + 1))
            elif u < 0.40:
                dq, dt = int(rng.integers(21, 80)), int(rng.integers(0, 3))                        # long insertion
            elif u < 0.50:
                dq, dt = int(rng.integers(0, 3)), int(rng.integers(21, 80))                        # long deletion
            else:
                dq, dt = int(rng.integers(0, 6)), int(rng.integers(0, 6))
            q, t = max(q + ln + dq, 0), max(t + ln + dt, 0)
        if not bl:
            bl = [(5, 5, 10)]
        blocks.append(np.asarray(bl, dtype=np.int32).reshape(-1, 3))
    to_ascii = np.frombuffer(b"ACGT", dtype=np.uint8)
    reads = [to_ascii[r] for r in reads]; gens = [to_ascii[g] for g in gens]
    rl = np.array([len(r) for r in reads], dtype=np.int64); gl = np.array([len(g) for g in gens], dtype=np.int64)
    qoff = np.concatenate([[0], np.cumsum(rl)[:-1]]); toff = np.concatenate([[0], np.cumsum(gl)[:-1]])
    qdev = torch.from_numpy(np.concatenate(reads + [np.zeros(64, np.uint8)])).to(ctx.device)
    tdev = torch.from_numpy(np.concatenate(gens + [np.zeros(64, np.uint8)])).to(ctx.device)
    b = refine.RefineBatch(ctx, blocks, qdev, qoff, rl.astype(np.int32), tdev, toff, gl)
    res = refine.calculate_statistics_batch(ctx, b, oracle.log_lookup_table())
    counts, value, cigars = refine.fetch_stats(ctx, res)
    for i in range(n_aln):
        cnt, val, runs, cigar = oracle.calculate_statistics(blocks[i], reads[i].tobytes(), gens[i].tobytes())
        assert cigars[i] == cigar, i
        assert _bits(value[i]) == _bits(val), i
        assert [int(x) for x in counts[i]] == [cnt[k] for k in refine.STAT_NAMES], i
