"""a2: the device std::sort emulation must reproduce libstdc++'s permutation exactly (ties included)."""
import numpy as np
import pytest


def _cases():
    rng = np.random.default_rng(17)
    cs = []
    for n in [0, 1, 2, 3, 15, 16, 17, 18, 33, 64, 65, 100, 257, 1000, 4097, 6000, 8999, 9000, 9001, 12000]:
        for ks in [2, 5, 50, 1 << 40]:
            k = rng.integers(0, ks, size=n).astype(np.uint64)
            k |= (rng.integers(0, 2, size=n).astype(np.uint64) << np.uint64(63))     # strand flag: ignored by the comparison
            cs.append(k)
    for n in [500, 6000]:
        cs.append(np.arange(n, dtype=np.uint64))                      # sorted
        cs.append(np.arange(n, dtype=np.uint64)[::-1].copy())         # reversed
        cs.append(np.zeros(n, dtype=np.uint64))                       # all equal
        cs.append(np.concatenate([np.arange(n // 2), np.arange(n // 2)[::-1]]).astype(np.uint64))   # organ pipe
    return cs


def test_oracle_sort_is_std_sort_and_adversary_hits_depth_limit(oracle):
    k = oracle.antiqsort_keys(5000)
    assert k.max() <= 5000 and len(np.unique(k)) > 4000
    sk, sp = oracle.sort_minimizers(k, np.arange(5000, dtype=np.uint32))
    assert np.all(np.diff(sk.astype(np.int64)) >= 0) and sorted(sp.tolist()) == list(range(5000))


@pytest.mark.gpu
def test_hip_sort_matches_std_sort(ctx, oracle):
    from lra_amd import seed
    cs = _cases()
    for n in [300, 3000, 7000]:
        cs.append(oracle.antiqsort_keys(n))                           # forces the heap-sort fall-back
    pos = [np.arange(len(k), dtype=np.uint32) for k in cs]
    got = seed.sort_minimizers_batch(ctx, cs, pos)
    for i, (k, p) in enumerate(zip(cs, pos)):
        ek, ep = oracle.sort_minimizers(k, p)
        assert np.array_equal(got[i][0], ek), (i, len(k))
        assert np.array_equal(got[i][1], ep), (i, len(k))


@pytest.mark.gpu
def test_hip_sort_big_lists(ctx, oracle):
    """Lists beyond the LDS capacity (9000 tuples): the global-memory workgroup sort up to 65534 tuples, the serial kernel beyond; heavy ties,
    the depth-limit adversary, sizes around both limits."""
    from lra_amd import seed
    rng = np.random.default_rng(3)
    cs = [rng.integers(0, 1 << 34, 9001).astype(np.uint64), rng.integers(0, 50, 20000).astype(np.uint64), rng.integers(0, 1 << 20, 65534).astype(np.uint64),
          rng.integers(0, 4000, 65535).astype(np.uint64), rng.integers(0, 1 << 30, 70000).astype(np.uint64), oracle.antiqsort_keys(12000),
          np.arange(30000, 0, -1).astype(np.uint64), np.zeros(15000, np.uint64), rng.integers(0, 1 << 34, 100).astype(np.uint64)]
    cs[1] |= (rng.integers(0, 2, 20000).astype(np.uint64) << np.uint64(63))      # strand bits: not part of the order
    pos = [np.arange(len(k), dtype=np.uint32) for k in cs]
    got = seed.sort_minimizers_batch(ctx, cs, pos)
    for i, (k, p) in enumerate(zip(cs, pos)):
        ek, ep = oracle.sort_minimizers(k, p)
        assert np.array_equal(got[i][0], ek), (i, len(k))
        assert np.array_equal(got[i][1], ep), (i, len(k))


@pytest.mark.gpu
def test_hip_sort_huge_lists(ctx, oracle):
    """Lists beyond 65534 tuples (the minimizers of an assembly contig: ~180 k per Mb) take the workgroup sort with 32-bit indices and its tables in
    global memory: heavy ties, few ties, the depth-limit adversary, already sorted / reversed input, two such lists beside small ones in one batch."""
    from lra_amd import seed
    rng = np.random.default_rng(11)
    cs = [rng.integers(0, 1 << 34, 180000).astype(np.uint64), rng.integers(0, 3000, 131072).astype(np.uint64), oracle.antiqsort_keys(70000),
          np.arange(90000, 0, -1).astype(np.uint64), np.arange(66000).astype(np.uint64), rng.integers(0, 1 << 34, 500).astype(np.uint64),
          rng.integers(0, 1 << 20, 65535).astype(np.uint64), np.zeros(70001, np.uint64), rng.integers(0, 7, 300000).astype(np.uint64)]
    cs[0] |= (rng.integers(0, 2, 180000).astype(np.uint64) << np.uint64(63))     # strand bits: not part of the order
    pos = [np.arange(len(k), dtype=np.uint32) for k in cs]
    got = seed.sort_minimizers_batch(ctx, cs, pos)
    for i, (k, p) in enumerate(zip(cs, pos)):
        ek, ep = oracle.sort_minimizers(k, p)
        assert np.array_equal(got[i][0], ek), (i, len(k))
        assert np.array_equal(got[i][1], ep), (i, len(k))


@pytest.mark.gpu
@pytest.mark.parametrize("bits", [(0, 64), (0, 34), (1, 42), (17, 63)])
def test_hip_sort_pairs_segments_are_stably_sorted(ctx, bits):
    """lra_sort_pairs_batch (segsort.hip) against numpy's stable sort on the selected key bits: segment lengths around the boundaries between its size classes (one
    workgroup's LDS sort for 257 .. 8192 pairs: 1024 / 2048 / 4096 / 8192) and rocprim's share (<= 256, > 8192), empty segments, gaps between segments, heavy ties
    (stability is what the callers rely on), keys that differ only outside the compared bits."""
    from lra_amd import seed
    rng = np.random.default_rng(5)
    b0, b1 = bits
    lens = [0, 1, 2, 255, 256, 257, 258, 1000, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097, 6000, 8191, 8192, 8193, 9000, 20000, 0, 300, 700]
    lens += [int(x) for x in rng.integers(1, 3000, size=40)]
    begin, end, at = [], [], 5
    for n in lens:
        begin.append(at); end.append(at + n); at += n + int(rng.integers(0, 4))    # (a few unused positions between segments)
    total = at + 3
    keys = rng.integers(0, 1 << 62, size=total).astype(np.uint64)
    for i, (b, e) in enumerate(zip(begin, end)):                                   # every third segment: few distinct keys -> long runs of ties
        if i % 3 == 0 and e > b:
            keys[b:e] = rng.integers(0, 7, size=e - b).astype(np.uint64) << np.uint64(b0 + 3)
            keys[b:e] |= rng.integers(0, 1 << max(b0, 1), size=e - b).astype(np.uint64) & np.uint64((1 << b0) - 1)   # noise below begin_bit: must not matter
    vals = np.arange(total, dtype=np.uint32)
    ok, ov = seed.sort_pairs_batch(ctx, keys, vals, begin, end, b0, b1)
    mask = np.uint64(((1 << b1) - 1) & ~((1 << b0) - 1)) if b1 < 64 else np.uint64((0xFFFFFFFFFFFFFFFF >> b0) << b0)
    for b, e in zip(begin, end):
        if e == b:
            continue
        o = np.argsort(keys[b:e] & mask, kind="stable")
        assert np.array_equal(ok[b:e], keys[b:e][o]), (b, e)
        assert np.array_equal(ov[b:e], vals[b:e][o]), (b, e)
