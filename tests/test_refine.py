"""a14 IndelRefineAlignment: oracle sanity (CPU; parity unpinned) and HIP vs oracle (GPU)."""
import numpy as np
import pytest

from lra_amd import synth


def perturb(rng, blocks, drop=0.15, trim=0.3):
    """Make truth blocks look like what the chaining / seed-extension stages hand over: some blocks
    missing, some ends trimmed (gaps on both sequences), everything still colinear."""
    b = blocks.copy()
    keep = rng.random(len(b)) > drop
    keep[0] = keep[-1] = True
    b = b[keep]
    for i in range(len(b)):
        if b[i, 2] > 6 and rng.random() < trim:
            a = int(rng.integers(0, 3)); z = int(rng.integers(0, 3))
            b[i, 0] += a; b[i, 1] += a; b[i, 2] -= a + z
    return b[b[:, 2] > 0]


def make_cases(seed, n, mean_len, err, mix, genome, drop=0.15, trim=0.3):
    rng = np.random.default_rng(seed)
    reads, blocks = [], []
    for i in range(n):
        L = int(max(200, rng.normal(mean_len, mean_len / 4)))
        r, b = synth.simulate_read_with_blocks(rng, genome, L, err, mix)
        reads.append(r)
        blocks.append(perturb(rng, b, drop, trim))
    return reads, blocks


def check_consistent(b, inp):
    b = np.asarray(b, dtype=np.int64)
    assert np.all(b[:-1, 0] + b[:-1, 2] <= b[1:, 0]) and np.all(b[:-1, 1] + b[:-1, 2] <= b[1:, 1])
    assert b[0, 0] == inp[0, 0] and b[0, 1] == inp[0, 1]
    assert b[-1, 0] + b[-1, 2] == inp[-1, 0] + inp[-1, 2] and b[-1, 1] + b[-1, 2] == inp[-1, 1] + inp[-1, 2]


def test_oracle_refine_sanity(oracle):
    genome = synth.make_genome(200000, seed=8)
    g = genome.tobytes()
    for (band, par, err, mix) in [(7, (4, -1, -2), 0.10, (30, 35, 35)), (20, (4, -1, -2), 0.15, (20, 30, 50)), (7, (4, -3, -4), 0.01, (34, 33, 33))]:
        reads, blocks = make_cases(band, 12, 3000, err, mix, genome)
        for r, b in zip(reads, blocks):
            out, st = oracle.indel_refine(b, r.tobytes(), g, band, *par)
            assert st == 0
            check_consistent(out, b)
            # refinement never aligns fewer bases than it was given inside small-gap runs by much
            assert out[:, 2].sum() >= 0.9 * b[:, 2].sum()
    # identity: a perfect single-gap alignment comes back unchanged in extent
    seq = genome[1000:1400]
    b = np.array([[0, 1000, 150], [152, 1152, 248]], dtype=np.int32)
    out, st = oracle.indel_refine(b, seq.tobytes(), g, 7, 4, -3, -4)
    assert st == 0 and out[:, 2].sum() == 400      # head pass-through + refined run + tail: contiguous, nothing lost
    assert np.all(out[:-1, 0] + out[:-1, 2] == out[1:, 0]) and np.all(out[:-1, 1] + out[:-1, 2] == out[1:, 1])
    # 0 / 1 block: untouched
    out, st = oracle.indel_refine(b[:1], seq.tobytes(), g, 7, 4, -3, -4)
    assert np.array_equal(out, b[:1])


def _run_gpu(ctx, genome, reads, blocks, band, par, end_align=False):
    import torch
    from lra_amd import refine
    gdev = torch.from_numpy(np.concatenate([genome, np.zeros(64, np.uint8)])).to(ctx.device)
    lens = np.array([len(r) for r in reads], dtype=np.int64)
    qoff = np.zeros(len(reads), dtype=np.int64)
    qoff[1:] = np.cumsum(lens[:-1])
    qdev = torch.from_numpy(np.concatenate(list(reads) + [np.zeros(64, np.uint8)])).to(ctx.device)
    batch = refine.RefineBatch(ctx, blocks, qdev, qoff, lens.astype(np.int32), gdev, np.zeros(len(reads), np.int64),
                               np.full(len(reads), len(genome), np.int64))
    res = refine.indel_refine_batch(ctx, batch, band, *par, end_align=end_align)
    return refine.fetch(ctx, res), res


@pytest.mark.gpu
@pytest.mark.parametrize("band,par,err,mix,mean_len,n,end_align", [
    (7, (4, -1, -2), 0.10, (30, 35, 35), 4000, 48, False),
    (20, (4, -1, -2), 0.15, (20, 30, 50), 3000, 32, False),
    (7, (4, -3, -4), 0.01, (34, 33, 33), 5000, 40, True),
    (7, (4, -1, -2), 0.10, (30, 35, 35), 30000, 6, False),
    (50, (4, -3, -4), 0.02, (34, 33, 33), 6000, 24, True),       # -CONTIG: refineBand 50, rows of more than 64 cells (ir_fill_wide)
    (50, (4, -3, -4), 0.08, (30, 35, 35), 4000, 16, False),
    (33, (4, -1, -2), 0.10, (30, 35, 35), 4000, 16, False),
])
def test_hip_refine_matches_oracle(ctx, oracle, band, par, err, mix, mean_len, n, end_align):
    genome = synth.make_genome(300000, seed=21)
    reads, blocks = make_cases(100 + band, n, mean_len, err, mix, genome)
    # edge cases: empty, single block, two touching blocks, a run that is too short for the DP
    reads += [genome[50:90].copy(), genome[500:900].copy(), genome[2000:2040].copy()]
    blocks += [np.zeros((0, 3), np.int32), np.array([[0, 500, 400]], np.int32),
               np.array([[0, 2000, 3], [4, 2004, 2], [7, 2007, 30]], np.int32)]
    (got, status), res = _run_gpu(ctx, genome, reads, blocks, band, par, end_align)
    g = genome.tobytes()
    assert res.n_cells > 0
    for i, (r, b) in enumerate(zip(reads, blocks)):
        exp, st = oracle.indel_refine(b, r.tobytes(), g, band, *par, end_align=end_align)
        assert st == 0
        assert status[i] == 0, (i, status[i])
        assert np.array_equal(got[i], exp), (i, len(got[i]), len(exp))


@pytest.mark.gpu
@pytest.mark.parametrize("band,err,mean_len,n,odd", [(7, 0.10, 6000, 24, 0.0), (7, 0.12, 20000, 6, 0.02), (20, 0.10, 8000, 8, 0.01)])
def test_hip_refine_long_segments(ctx, oracle, band, err, mean_len, n, odd):
    """What LocalRefineAlignment hands over: every gap is small, so one segment spans hundreds of
    blocks (the chunked window construction).  `odd` > 0 also pulls some blocks' query start back
    (overlapping query coordinates, a drifting query cursor in IndelRefine.h:287-314) and some
    target starts back (the row count no longer adds up -> the alignment is rejected)."""
    genome = synth.make_genome(300000, seed=22)
    reads, blocks = make_cases(7 + band, n, mean_len, err, (30, 35, 35), genome, drop=0.0, trim=0.5)
    rng = np.random.default_rng(5)
    if odd:
        # the drifting cursor widens the last rows past the segment's query end: keep the alignment
        # clear of the read end so that those cells are still inside the read (both sides then see
        # the same bases; past the read the reference reads whatever follows in memory)
        blocks = [b[b[:, 0] + b[:, 2] < len(r) - 64].copy() for r, b in zip(reads, blocks)]
        for b in blocks:
            m = rng.random(len(b)) < odd
            m[0] = False
            b[m, 0] -= rng.integers(1, 3, size=int(m.sum())).astype(b.dtype)
        b = blocks[-1]
        b[len(b) // 2, 1] -= 8
    (got, status), res = _run_gpu(ctx, genome, reads, blocks, band, (4, -1, -2))
    assert res.n_rows / max(res.n_segments, 1) > 500
    g = genome.tobytes()
    n_ok = 0
    for i, (r, b) in enumerate(zip(reads, blocks)):
        exp, st = oracle.indel_refine(b, r.tobytes(), g, band, 4, -1, -2)
        assert (status[i] != 0) == (st != 0), (i, status[i], st)
        if st == 0:
            n_ok += 1
            assert np.array_equal(got[i], exp), (i, len(got[i]), len(exp))
    assert n_ok >= n // 2
