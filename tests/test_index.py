"""`lra index` for the global minimizer index (StoreIndex, MMIndex.h:286-400) and the .mms / .gli files.
CPU: the oracle's StoreIndex properties, the file layouts against a struct-level restatement of WriteIndex / LocalIndex::Write.
GPU: lra_ctx_build_global_index against the oracle bit for bit (equal keys in emission order on both sides), on genomes with N runs, lower
case, low-complexity stretches (the serial back-off path), chromosomes around the chunk size, chromosomes shorter than a window."""
import struct

import numpy as np
import pytest

import oracle_lib as O
from lra_amd import synth

M63 = np.uint64((1 << 63) - 1)


def _genome(seed, n, repeat_frac=0.2):
    return synth.make_genome(n, seed=seed, repeat_frac=repeat_frac, n_families=2)


def test_oracle_store_index_properties(oracle):
    g = _genome(3, 300_000)
    CH = [0, 120_000, 300_000]
    key, pos, st = O.store_index(g.tobytes(), CH, 17, 10, 150, 15, 1)
    ks, ps, st2 = O.store_index(g.tobytes(), CH, 17, 10, 150, 15, 1, stable=True)
    assert st == 0 and st2 == 0
    # sorted by masked key; at most one entry per 15-base window; about one entry per window
    mk = key & M63
    assert np.all(mk[:-1] <= mk[1:])
    win = pos // 15
    assert len(np.unique(win)) == len(win)
    assert 0.75 * 300_000 / 15 < len(key) <= 300_000 / 15 + 1
    # the std::sort and the stable order give the same entries except where one window holds two candidates with the same key
    a = set(zip(key.tolist(), pos.tolist())); b = set(zip(ks.tolist(), ps.tolist()))
    assert len(a ^ b) <= 0.002 * len(a)
    # every entry is a minimizer of its chromosome with that key
    allk, allp, _ = O.store_index(g.tobytes(), CH, 17, 10, 1 << 20, 1, 1 << 30)      # (max_freq sizes CountSort's table: 2^20 keeps every key of a 300 kb genome)
    full = dict(zip(allp.tolist(), allk.tolist()))
    assert all(full.get(p) == k for k, p in zip(key.tolist(), pos.tolist()))
    # frequency filter: with max_freq 1 every surviving key is unique in the full minimizer list
    k1, p1, _ = O.store_index(g.tobytes(), CH, 17, 10, 1, 15, 1)
    cnt = {}
    for k in (allk & M63).tolist():
        cnt[k] = cnt.get(k, 0) + 1
    assert all(cnt[k] == 1 for k in (k1 & M63).tolist()) and 0 < len(k1) < len(key)
    # n_per_window 2 keeps at most two per window and a superset of the one-per-window entries' windows
    k2, p2, _ = O.store_index(g.tobytes(), CH, 17, 10, 150, 15, 2)
    _, c2 = np.unique(p2 // 15, return_counts=True)
    assert c2.max() == 2 and len(k2) > len(key)


def test_mms_gli_files(tmp_path, oracle):
    """lra_write_mms / lra_read_mms / lra_write_gli / lra_read_gli against the byte layout of WriteIndex (MMIndex.h:416-424, Genome.h:59-68) and
    LocalIndex::Write (MMIndex.h:138-151) restated with struct."""
    from lra_amd import index as I
    g = _genome(5, 40_000)
    CH = [0, 15_000, 40_000]
    names = [b"chr1", b"chrUn_long_name"]
    key, pos, _ = O.store_index(g.tobytes(), CH, 17, 10, 150, 15, 1)
    p = tmp_path / "ref.fa.mms"
    I.write_mms(p, 17, names, CH, key, pos)
    exp = struct.pack("<qi", len(key), 17) + struct.pack("<i", 2)
    for n in names:
        exp += struct.pack("<i", len(n)) + n
    exp += np.asarray(CH, np.uint64).tobytes()
    rec = np.zeros(len(key), dtype=[("t", "<u8"), ("pos", "<u4"), ("pad", "<u4")])
    rec["t"] = key; rec["pos"] = pos
    exp += rec.tobytes()
    assert p.read_bytes() == exp
    r = I.read_mms(p)
    assert r["globalK"] == 17 and r["names"] == names and r["chrom_pos"].tolist() == CH and np.array_equal(r["key"], key) and np.array_equal(r["pos"], pos)
    # .gli
    tup, bnd = [], [0]
    so = [0]
    for c in range(2):
        t, b = O.local_index_seq(g[CH[c]:CH[c + 1]].tobytes(), 10, 5, 256, 15)
        tup.append(t); bnd.extend((b[1:] + bnd[-1]).tolist())
        x = CH[c]
        while x < CH[c + 1]:
            x = min(x + 256, CH[c + 1]); so.append(x)
    tup = np.concatenate(tup)
    q = tmp_path / "ref.fa.gli"
    I.write_gli(q, 10, 5, 256, so, bnd, tup)
    exp = struct.pack("<iiii", 10, 5, 256, len(so)) + np.asarray(so, np.uint64).tobytes() + np.asarray(bnd, np.uint64).tobytes() + struct.pack("<Q", len(tup)) + tup.astype("<u4").tobytes()
    assert q.read_bytes() == exp
    r = I.read_gli(q)
    assert (r["k"], r["w"], r["window"]) == (10, 5, 256) and r["seq_offsets"].tolist() == so and r["tuple_bnd"].tolist() == bnd and np.array_equal(r["tuples"], tup)
    # empty index
    I.write_mms(tmp_path / "e.mms", 15, [b"c"], [0, 5], np.zeros(0, np.uint64), np.zeros(0, np.uint32))
    r = I.read_mms(tmp_path / "e.mms")
    assert len(r["key"]) == 0 and r["names"] == [b"c"]


def _hard_genome(rng):
    """Chromosomes that exercise the builder: N runs at the start / in the middle / at the end (one ending exactly one window before the end),
    lower case, poly-A and dinucleotide stretches longer than the warm-up, lengths around the 4096-position chunks, a chromosome shorter
    than a window and one of exactly window + 1 bases."""
    B = np.frombuffer(b"ACGT", np.uint8)
    def rnd(n): return B[rng.integers(0, 4, n)].copy()
    chroms = []
    c = rnd(30_000); c[:700] = ord("N"); c[9_000:9_003] = ord("N"); c[15_000:21_000] = ord("N"); c[-400:] = ord("N"); chroms.append(c)
    c = rnd(4096 + 16); chroms.append(c)                                     # nk = 4096: exactly one chunk
    c = rnd(4096 + 17); chroms.append(c)                                     # one position into the second chunk
    c = rnd(2 * 4096 + 16 + 190); chroms.append(c)
    c = rnd(25); chroms.append(c)                                            # shorter than w + k - 1
    c = rnd(26); chroms.append(c)                                            # == span: still nothing (MinCount.h:27 `<`)
    c = rnd(27); chroms.append(c)
    c = rnd(20_000); c[3_000:9_000] = ord("A"); c[12_000:16_500] = np.tile(np.frombuffer(b"AC", np.uint8), 2250); chroms.append(c)   # ties for thousands of windows
    c = rnd(12_000); c[5_000:7_000] = np.frombuffer(c[5_000:7_000].tobytes().lower(), np.uint8); chroms.append(c)
    c = rnd(9_000); c[9_000 - 27:9_000 - 26] = ord("N"); chroms.append(c)   # the last N leaves exactly span bases: the last window is NOT emitted (:117)
    c = rnd(9_000); c[9_000 - 28:9_000 - 27] = ord("N"); chroms.append(c)   # one more base: it is
    c = np.tile(np.frombuffer(b"ACGGTCA", np.uint8), 3000); chroms.append(c)  # tandem repeat, period 7: every window ties
    c = rnd(50_000); c[20_000:24_096] = c[10_000:14_096]; chroms.append(c)   # a duplication (key runs of 2)
    pos = np.concatenate([[0], np.cumsum([len(x) for x in chroms])])
    return np.concatenate(chroms), [int(x) for x in pos]


@pytest.mark.gpu
@pytest.mark.parametrize("params", [(17, 10, 150, 15, 1), (15, 10, 3, 12, 1), (19, 10, 30, 20, 2), (17, 10, 1 << 20, 1, 1 << 20), (9, 4, 50, 15, 1), (21, 32, 100, 15, 1)])
def test_hip_build_global_index_oracle(ctx, oracle, params):
    from lra_amd import index as I
    k, w, mf, ws, npw = params
    g, CH = _hard_genome(np.random.default_rng(11))
    I.load_genome(ctx, g)
    r = I.build_global_index(ctx, CH, k, w, mf, ws, npw)
    key, pos = I.global_index(ctx)
    ek, ep, est = O.store_index(g.tobytes(), CH, k, w, mf, ws, npw, stable=True)
    assert r["n_index"] == len(ek) == len(key), (r, len(ek))
    assert np.array_equal(key, ek) and np.array_equal(pos, ep)
    assert (r["status"] != 0) == (est != 0)
    if mf >= 1 << 20:                                                        # no filter, no thinning: the index IS the sorted minimizer list
        assert r["n_minimizers"] == len(ek)
    # the serial back-off path ran (the poly-A / tandem chromosomes) and the fast path did most of the work
    if k == 17:
        assert ctx.timing_get("gsketch_serial")[1] >= 0


@pytest.mark.gpu
def test_hip_index_feeds_the_path(ctx, oracle):
    """An index built on the device is what lra_seed_batch looks up: matches of simulated reads against it equal the oracle's CompareLists on
    the oracle's StoreIndex (same order of equal keys)."""
    from lra_amd import index as I, seed
    g = _genome(21, 400_000)
    CH = [0, 400_000]
    I.load_genome(ctx, g)
    I.build_global_index(ctx, CH, 17, 10, 150, 15, 1)
    ek, ep, _ = O.store_index(g.tobytes(), CH, 17, 10, 150, 15, 1, stable=True)
    reads, _ = synth.simulate_reads(g, 6, 6000, 1500, 0.08, seed=3)
    batch = seed.ReadBatch(ctx, [r.tobytes() for r in reads])
    sres = seed.seed_batch(ctx, batch, 17, 10, 150)
    out = seed.fetch(ctx, sres) if hasattr(seed, "fetch") else None
    n_tot = 0
    for i, rd in enumerate(reads):
        keys, p_ = O.store_minimizers(rd.tobytes(), 17, 10)
        sk, sp = O.sort_minimizers(keys, p_)
        qi, ti = O.compare_lists(sk, sp, ek, ep, 150)
        n_tot += len(qi)
    assert int(sres.n_matches) == n_tot and n_tot > 100


@pytest.mark.gpu
@pytest.mark.parametrize("preset", ["ont", "ccs"])
def test_local_index_from_a_gli_file(ctx, oracle, tmp_path, preset):
    """glIndex handed over as LocalIndex::Read leaves it (lra_ctx_load_local_index: the .gli payload, k / w / window from the file) instead of built on the device: an
    index built with `lra index`'s values (k 10, w 5, windows of 2048 bases), written with lra_write_gli, read back, loaded into a second context -- the same alignments
    as on the context that built it; the options' own local values are refused until lra_map_opts_apply_local_index has overridden them; an index of another chromosome
    table is refused."""
    import ctypes as C
    from lra_amd import index as I, seed, mapread
    from lra_amd.context import Context
    from lra_amd._lib import LraError
    g = _genome(33, 500_000)
    CH = [0, 230_000, 500_000]
    reads, _ = synth.simulate_reads(g, 8, 7000, 1500, 0.10 if preset == "ont" else 0.01, seed=5)
    names = [b"c1", b"c2"]
    if preset == "ont":
        a = mapread.LowAccMapper(ctx, g, None, None, names, CH, mapread.with_gli(mapread.LowAccOptions()), staged=False)
    else:
        a = mapread.HighAccMapper(ctx, g, None, None, names, CH, "ccs", gli=True, index_params=(17, 10, 150, 18, 1))   # (a thin index: some reads take the REFINEclusters branch)
    so, tb, tu = a.fetch_local_index()
    k_, w_, win_ = C.c_int(0), C.c_int(0), C.c_int(0)
    ctx.check(ctx.lib.lra_ctx_local_index_params(ctx.h, C.byref(k_), C.byref(w_), C.byref(win_)))
    assert (k_.value, w_.value, win_.value) == (10, 5, 2048)
    path = tmp_path / "ref.fa.gli"
    I.write_gli(path, 10, 5, 2048, so, tb, tu)
    f = I.read_gli(path)
    assert (f["k"], f["w"], f["window"]) == (10, 5, 2048) and np.array_equal(f["tuples"], tu)
    batch = seed.ReadBatch(ctx, [r.tobytes() for r in reads])
    ra = a.fetch(a.align(batch))
    ik, ip = I.global_index(ctx)
    c2 = Context(ctx.device.index or 0)
    b = (mapread.LowAccMapper(c2, g, ik, ip, names, CH, mapread.LowAccOptions(), staged=False) if preset == "ont"
         else mapread.HighAccMapper(c2, g, ik, ip, names, CH, "ccs"))          # the options' own values: the context builds a 256-base-window index first
    def load(cp):
        so_, tb_, tu_ = (np.ascontiguousarray(f[n]) for n in ("seq_offsets", "tuple_bnd", "tuples"))
        return c2.lib.lra_ctx_load_local_index(c2.h, f["k"], f["w"], f["window"], C.c_uint64(len(so_) - 1), C.c_void_p(so_.ctypes.data), C.c_void_p(tb_.ctypes.data),
                                               C.c_uint64(len(tu_)), C.c_void_p(tu_.ctypes.data))
    c2.check(load(CH))
    b2 = seed.ReadBatch(c2, [r.tobytes() for r in reads])
    with pytest.raises(LraError):                                              # the options still say k / window of the preset: refused
        b.align(b2)
    c2.lib.lra_map_opts_apply_local_index(C.byref(b.copts), f["k"], f["w"], f["window"])
    rb = b.fetch(b.align(b2))
    assert int(ra["job_aln_off"][-1]) >= 6
    for kk in ("job_aln_off", "strand", "chrom", "block_off", "blocks", "counts", "runs", "read_status"):
        assert np.array_equal(ra[kk], rb[kk]), kk
    # an index of another genome (here: the same tuples against a chromosome table cut elsewhere) is refused
    cp = (C.c_uint64 * 3)(0, 200_000, 500_000)
    c2.check(c2.lib.lra_ctx_load_chromosomes(c2.h, cp, 2))
    assert load(None) != 0
