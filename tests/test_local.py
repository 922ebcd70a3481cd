"""a10 tier-2 primitives: oracle vs reference golden (CPU) and HIP vs oracle (GPU)."""
import json
import os

import numpy as np
import pytest

from lra_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "local_compare_golden.json")))["cases"]


def test_oracle_local_compare_matches_reference_golden(oracle):
    for c in GOLD:
        q = np.array(c["q"], dtype=np.uint32).reshape(-1, 2); t = np.array(c["t"], dtype=np.uint32).reshape(-1, 2)
        oq, ot = oracle.compare_lists_local(oracle.pack_local(q[:, 0], q[:, 1]), oracle.pack_local(t[:, 0], t[:, 1]), c["maxFreq"], c["maxDiag"], c["minDiag"])
        got = []
        for a, b in zip(oq, ot):
            got += [int(q[a, 0]), int(q[a, 1]), int(t[b, 0]), int(t[b, 1])]
        assert got == c["pairs"]


def test_oracle_local_index_sanity(oracle):
    g = synth.make_genome(3000, seed=6)
    tup, bnd = oracle.local_index_seq(g.tobytes(), 10, 5, 256, 15)
    assert len(bnd) == 13 and bnd[-1] == len(tup) and len(tup) > 300
    ck = np.zeros(len(g) - 9, dtype=np.int64)
    for i in range(10):
        ck = (ck << 2) | synth.CODE[g[i:i + len(ck)]]
    for wi in range(12):
        seg = tup[int(bnd[wi]):int(bnd[wi + 1])]
        t = seg & 0xFFFFF; p = seg >> 20
        assert np.all(np.diff(t.astype(np.int64)) >= 0)                      # sorted by k-mer
        assert np.all(ck[wi * 256 + p] == t)                                 # the k-mer at that window position
    assert len(oracle.local_index_seq(b"ACGTACGTAC", 10, 5, 256, 15)[0]) == 0
    poly = oracle.local_index_seq(b"A" * 600, 10, 5, 256, 15)[0]
    assert len(poly) < 40                                                    # RemoveFrequent drops the repeated k-mer


def _seqs():
    rng = np.random.default_rng(77)
    g = synth.make_genome(40000, seed=8, repeat_frac=0.3)
    seqs = [g[i * 3000:i * 3000 + int(rng.integers(200, 3000))].copy() for i in range(12)]
    seqs += [np.frombuffer(b"", np.uint8), np.frombuffer(b"ACGTACGTACGTAC", np.uint8), np.frombuffer(b"N" * 700, np.uint8),
             np.concatenate([g[100:400], np.frombuffer(b"NNN", np.uint8), g[400:900]]), np.tile(np.frombuffer(b"ACG", np.uint8), 300),
             g[5000:5256].copy(), g[6000:6257].copy()]
    return g, seqs


def _device_seqs(ctx, seqs):
    import torch
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    off = np.zeros(len(seqs) + 1, dtype=np.int64); off[1:] = np.cumsum(lens)
    buf = np.concatenate(list(seqs) + [np.zeros(64, np.uint8)])
    return torch.from_numpy(buf).to(ctx.device), torch.from_numpy(off).to(ctx.device)


@pytest.mark.gpu
@pytest.mark.parametrize("k,w,window,mf", [(10, 5, 256, 15), (7, 5, 256, 30), (10, 5, 128, 5), (10, 5, 2048, 15), (10, 5, 2048, 3), (7, 5, 1000, 15), (10, 5, 4096, 15), (8, 16, 2048, 15)])
def test_hip_local_index_matches_oracle(ctx, oracle, k, w, window, mf):
    """window 2048 = the .gli file `lra index` writes (LocalIndex(0): 1 << (LOCAL_POS_BITS - 1), MMIndex.h:110-127) and the read indexes copied from it: ~700 tuples a
    window, a third of the windows with a k-mer twice (the exact sort), tandem repeats whose k-mers RemoveFrequent drops; 4096 = the widest window a LocalTuple's position holds."""
    from lra_amd import local
    g, seqs = _seqs()
    if window > 256:
        rng = np.random.default_rng(5)
        gl = synth.make_genome(400_000, seed=9, repeat_frac=0.4)
        seqs = seqs + [gl[a:a + n].copy() for a, n in ((0, 30_000), (50_000, 2048), (60_000, 2049), (70_000, 4096 * 3 + 17), (100_000, 25_000), (200_000, 12_345))]
        seqs.append(np.tile(gl[1000:1037], 200))                                                        # a tandem array: every k-mer 55 times in a 2048-base window
        seqs.append(np.concatenate([gl[3000:4000], np.tile(np.frombuffer(b"AC", np.uint8), 700), gl[4000:5500]]))
        seqs.append(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 9000)].copy())
        seqs.append(np.concatenate([gl[8000:9000], np.frombuffer(b"N" * 40, np.uint8), gl[9000:12000], gl[8000:9000], gl[8200:8900]]))   # repeats inside one window: keys twice
    sd, od = _device_seqs(ctx, seqs)
    li = local.LocalIndex(ctx, sd, od, k, w, window, mf)
    win_off, bnd, tup = li.fetch()
    total = 0
    for i, s in enumerate(seqs):
        et, eb = oracle.local_index_seq(s.tobytes(), k, w, window, mf)
        w0, w1 = int(win_off[i]), int(win_off[i + 1])
        assert w1 - w0 == len(eb) - 1, i
        got_b = bnd[w0:w1 + 1] - bnd[w0]
        assert np.array_equal(got_b, eb), i
        assert np.array_equal(tup[int(bnd[w0]):int(bnd[w1])], et), i
        total += len(et)
    assert total > 1000


@pytest.mark.gpu
def test_hip_local_compare_matches_golden_and_oracle(ctx, oracle):
    import torch
    from lra_amd import local
    # (i) the reference golden lists, through a fake index object holding raw tuple arrays
    class Raw:
        pass
    qs, ts, ql, qh, tl, th, mxd, mnd = [], [], [], [], [], [], [], []
    qo = to = 0
    by_mf = {}
    for c in GOLD:
        by_mf.setdefault(c["maxFreq"], []).append(c)
    for mf, cs in by_mf.items():
        qs, ts, ql, qh, tl, th, mxd, mnd = [], [], [], [], [], [], [], []
        qo = to = 0
        for c in cs:
            q = np.array(c["q"], dtype=np.uint32).reshape(-1, 2); t = np.array(c["t"], dtype=np.uint32).reshape(-1, 2)
            qs.append(oracle.pack_local(q[:, 0], q[:, 1])); ts.append(oracle.pack_local(t[:, 0], t[:, 1]))
            ql.append(qo); qo += len(q); qh.append(qo); tl.append(to); to += len(t); th.append(to)
            mxd.append(c["maxDiag"]); mnd.append(c["minDiag"])
        qa = np.concatenate(qs + [np.zeros(1, np.uint32)]); ta = np.concatenate(ts + [np.zeros(1, np.uint32)])
        A, B = Raw(), Raw()
        A.t_ = torch.from_numpy(qa.view(np.int32)).to(ctx.device); B.t_ = torch.from_numpy(ta.view(np.int32)).to(ctx.device)
        A.res = local.LocalIndexResult(); B.res = local.LocalIndexResult()
        A.res.d_tuples = A.t_.data_ptr(); B.res.d_tuples = B.t_.data_ptr()
        off, pqi, pti = local.local_compare_batch(ctx, A, ql, qh, B, tl, th, mf, mxd, mnd)
        for i, c in enumerate(cs):
            got = []
            for a, b in zip(pqi[int(off[i]):int(off[i + 1])], pti[int(off[i]):int(off[i + 1])]):
                got += [int(qa[a] & 0xFFFFF), int(qa[a] >> 20), int(ta[b] & 0xFFFFF), int(ta[b] >> 20)]
            assert got == c["pairs"], (mf, i)
    # (ii) real windows: read windows against the genome windows they overlap
    g = synth.make_genome(60000, seed=9, repeat_frac=0.3)
    rng = np.random.default_rng(3)
    reads = []
    for i in range(10):
        s = int(rng.integers(0, 50000))
        r = g[s:s + 5000].copy()
        mut = rng.random(len(r)) < 0.08
        r[mut] = synth.BASES[rng.integers(0, 4, size=int(mut.sum()))]
        reads.append((s, r))
    gd, god = _device_seqs(ctx, [g])
    rd, rod = _device_seqs(ctx, [r for _, r in reads])
    gi = local.LocalIndex(ctx, gd, god); ri = local.LocalIndex(ctx, rd, rod)
    gw, gb, gt = gi.fetch(); rw, rb, rt = ri.fetch()
    ql, qh, tl, th = [], [], [], []
    for i, (s, r) in enumerate(reads):
        for wloc in range(int(rw[i + 1] - rw[i])):
            wq = int(rw[i]) + wloc
            gwin = (s + wloc * 256) // 256
            for gwx in (gwin, gwin + 1):
                if gwx < len(gb) - 1:
                    ql.append(int(rb[wq])); qh.append(int(rb[wq + 1])); tl.append(int(gb[gwx])); th.append(int(gb[gwx + 1]))
    off, pqi, pti = local.local_compare_batch(ctx, ri, ql, qh, gi, tl, th, 15)
    npairs = 0
    for x in range(len(ql)):
        eq, et = oracle.compare_lists_local(rt[ql[x]:qh[x]], gt[tl[x]:th[x]], 15)
        a, b = int(off[x]), int(off[x + 1])
        assert np.array_equal(pqi[a:b], eq + np.uint32(ql[x])) and np.array_equal(pti[a:b], et + np.uint32(tl[x])), x
        npairs += b - a
    assert npairs > 500


@pytest.mark.gpu
@pytest.mark.parametrize("banded", [False, True])
def test_hip_local_compare_large_tasks(ctx, oracle, banded):
    """A batch whose tasks are lists of hundreds of tuples (windows of 2048 bases): the lane-per-task form of the walk (its bounds found from where the walk stands),
    against the oracle's CompareLists -- shared keys, keys several times on either side (beyond localMaxFreq too), lists of one tuple, with and without the diagonal band."""
    import torch
    from lra_amd import local
    class Raw:
        pass
    rng = np.random.default_rng(21 + banded)
    qs, ts, ql, qh, tl, th, mx, mn = [], [], [], [], [], [], [], []
    qo = to = 0
    for c in range(300):
        if c % 17 == 5:
            nq, nt, nkeys = int(rng.integers(1, 4)), int(rng.integers(200, 700)), 500
        elif c % 17 == 9:
            nq, nt, nkeys = int(rng.integers(300, 900)), int(rng.integers(1, 3)), 500
        elif c % 17 == 12:                                          # few distinct k-mers: long runs on both sides
            nq, nt, nkeys = 400, 350, 30
        else:
            nq, nt, nkeys = int(rng.integers(250, 1000)), int(rng.integers(250, 1000)), int(rng.integers(600, 3000))
        keys = np.sort(rng.choice(1 << 20, nkeys, replace=False)).astype(np.uint32)
        qk = np.sort(rng.choice(keys, nq)); tk = np.sort(rng.choice(keys, nt))
        qp = rng.integers(0, 2048, nq).astype(np.uint32); tp = rng.integers(0, 2048, nt).astype(np.uint32)
        qs.append(oracle.pack_local(qk, qp)); ts.append(oracle.pack_local(tk, tp))
        ql.append(qo); qo += nq; qh.append(qo); tl.append(to); to += nt; th.append(to)
        d0 = int(rng.integers(-1500, 1500))
        mx.append(d0 + int(rng.integers(50, 800)) if c % 5 else 0); mn.append(d0 - int(rng.integers(50, 800)) if c % 7 else 0)
    qa = np.concatenate(qs + [np.zeros(1, np.uint32)]); ta = np.concatenate(ts + [np.zeros(1, np.uint32)])
    A, B = Raw(), Raw()
    A.t_ = torch.from_numpy(qa.view(np.int32)).to(ctx.device); B.t_ = torch.from_numpy(ta.view(np.int32)).to(ctx.device)
    A.res = local.LocalIndexResult(); B.res = local.LocalIndexResult()
    A.res.d_tuples = A.t_.data_ptr(); B.res.d_tuples = B.t_.data_ptr()
    kw = dict(max_diag=mx, min_diag=mn) if banded else {}
    off, pqi, pti = local.local_compare_batch(ctx, A, ql, qh, B, tl, th, 15, **kw)
    n_pairs = 0
    for x in range(len(ql)):
        eq, et = oracle.compare_lists_local(qa[ql[x]:qh[x]], ta[tl[x]:th[x]], 15, *((mx[x], mn[x]) if banded else ()))
        a, b = int(off[x]), int(off[x + 1])
        assert np.array_equal(pqi[a:b], eq + np.uint32(ql[x])) and np.array_equal(pti[a:b], et + np.uint32(tl[x])), x
        n_pairs += b - a
    assert n_pairs > 5000


@pytest.mark.gpu
def test_hip_local_compare_oversized_tasks(ctx, oracle):
    """Tasks that do not fit the one-pass form's row (more than 128 pairs, a list of more than 255 tuples) beside ordinary ones: the batch falls back to count + emit."""
    import torch
    from lra_amd import local
    class Raw:
        pass
    rng = np.random.default_rng(11)
    qs, ts, ql, qh, tl, th = [], [], [], [], [], []
    qo = to = 0
    for c in range(40):
        if c % 10 == 3:                                             # few distinct k-mers, many copies: hundreds of pairs
            nq, nt, nkeys = 60, 60, 6
        elif c % 10 == 7:                                           # long lists
            nq, nt, nkeys = 300, 280, 400
        else:
            nq, nt, nkeys = int(rng.integers(1, 40)), int(rng.integers(1, 40)), 60
        keys = np.sort(rng.choice(1 << 20, nkeys, replace=False)).astype(np.uint32)
        qk = np.sort(rng.choice(keys, nq)); tk = np.sort(rng.choice(keys, nt))
        qp = rng.integers(0, 256, nq).astype(np.uint32); tp = rng.integers(0, 256, nt).astype(np.uint32)
        qs.append(oracle.pack_local(qk, qp)); ts.append(oracle.pack_local(tk, tp))
        ql.append(qo); qo += nq; qh.append(qo); tl.append(to); to += nt; th.append(to)
    qa = np.concatenate(qs + [np.zeros(1, np.uint32)]); ta = np.concatenate(ts + [np.zeros(1, np.uint32)])
    A, B = Raw(), Raw()
    A.t_ = torch.from_numpy(qa.view(np.int32)).to(ctx.device); B.t_ = torch.from_numpy(ta.view(np.int32)).to(ctx.device)
    A.res = local.LocalIndexResult(); B.res = local.LocalIndexResult()
    A.res.d_tuples = A.t_.data_ptr(); B.res.d_tuples = B.t_.data_ptr()
    off, pqi, pti = local.local_compare_batch(ctx, A, ql, qh, B, tl, th, 15)
    big = 0
    for x in range(len(ql)):
        eq, et = oracle.compare_lists_local(qa[ql[x]:qh[x]], ta[tl[x]:th[x]], 15)
        a, b = int(off[x]), int(off[x + 1])
        assert np.array_equal(pqi[a:b], eq + np.uint32(ql[x])) and np.array_equal(pti[a:b], et + np.uint32(tl[x])), x
        big += (b - a) > 128
    assert big >= 2
