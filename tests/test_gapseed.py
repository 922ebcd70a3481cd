"""a11 RefineSpace (ClusterRefine.h:242-325): oracle sanity on CPU (its pieces are pinned elsewhere; the glue is unpinned), HIP vs oracle on the GPU."""
import numpy as np
import pytest

import oracle_lib as O
from lra_amd import synth


def _gap(rng, genome, qlen, err, with_n=False):
    s0 = int(rng.integers(1000, len(genome) - 40000))
    tlen = int(qlen * rng.uniform(0.85, 1.15)) + int(rng.integers(0, 40))
    t = genome[s0:s0 + tlen].copy()
    q = synth.mutate(genome[s0:s0 + qlen], err, rng) if hasattr(synth, "mutate") else None
    if q is None:
        q = genome[s0:s0 + qlen].copy()
        m = rng.random(len(q)) < err
        q[m] = rng.choice(np.frombuffer(b"ACGT", np.uint8), int(m.sum()))
    if with_n and len(q) > 50:
        q[10:13] = ord("N")
    return q.tobytes(), t.tobytes()


def test_oracle_refine_space_sanity():
    rng = np.random.default_rng(2)
    genome = synth.make_genome(200000, seed=5)
    q, t = _gap(rng, genome, 400, 0.0)
    oq, ot, ident = O.refine_space(q, t, len(t), 10, 5, 30)
    assert ident > 0.99 and len(oq) > 20 and np.all(oq == ot)                   # identical sequences: K-mers on the main diagonal
    q, t = _gap(rng, genome, 3000, 0.05)
    oq, ot, ident = O.refine_space(q, t, len(t), 9, 5, 60, q_add=100, t_add=7000)
    assert ident == -1 and len(oq) > 30
    d = ot.astype(np.int64) - 7000 - (oq.astype(np.int64) - 100)
    assert np.all(np.abs(d) <= abs(len(t) - len(q)) + 60)                        # inside the diagonal band
    k1, p1 = O.store_minimizers_noncanonical64(b"ACGT" * 3, 6, 5)
    assert len(k1) >= 1
    assert len(O.store_minimizers_noncanonical64(b"ACG", 6, 5)[0]) == 0


@pytest.mark.gpu
def test_hip_refine_space_oracle(ctx):
    import torch
    from lra_amd import gapseed
    rng = np.random.default_rng(11)
    genome = synth.make_genome(400000, seed=8, repeat_frac=0.3)
    probs = []
    for i in range(300):
        kind = i % 6
        qlen = int(rng.choice([8, 31, 60, 200, 600, 990])) if kind < 4 else int(rng.choice([1000, 1500, 4000, 12000]))
        q, t = _gap(rng, genome, qlen, float(rng.choice([0.0, 0.03, 0.12])), with_n=(i % 17 == 0))
        if kind == 3: t = t[: max(5, len(t) // 3)]                                   # very unequal spans
        K = int(rng.choice([6, 9, 12])); W = int(rng.choice([3, 5]))
        flip = int(rng.choice([0, 0, 50000]))
        probs.append(dict(q=q, t=t, t_span=len(t) - int(rng.integers(0, 3)), K=K, W=W, diag=int(rng.choice([30, 60, 200])), q_add=int(rng.integers(0, 30000)),
                          t_add=int(rng.integers(0, 1 << 30)), flip=flip))
    qcat = b"".join(p["q"] for p in probs); tcat = b"".join(p["t"] for p in probs)
    qoff = np.cumsum([0] + [len(p["q"]) for p in probs])[:-1]; toff = np.cumsum([0] + [len(p["t"]) for p in probs])[:-1]
    dev = ctx.device
    T = lambda a, dt: torch.tensor(np.asarray(a, dtype=dt), device=dev)
    dq = torch.tensor(np.frombuffer(qcat + b"\0" * 64, np.uint8).copy(), device=dev); dt_ = torch.tensor(np.frombuffer(tcat + b"\0" * 64, np.uint8).copy(), device=dev)
    args = [T(qoff, np.int64), T([len(p["q"]) for p in probs], np.int32)]
    res = gapseed.refine_space_batch(ctx, len(probs), dq, args[0], args[1], dt_, T(toff, np.int64), T([len(p["t"]) for p in probs], np.int32),
                                     T([p["t_span"] for p in probs], np.int64).to(torch.int32), T([p["K"] for p in probs], np.int32),
                                     T([p["W"] for p in probs], np.int32), T([p["diag"] for p in probs], np.int32),
                                     T([p["q_add"] for p in probs], np.int64).to(torch.int32), T([p["t_add"] for p in probs], np.int64).to(torch.int32),
                                     T([p["flip"] for p in probs], np.int64).to(torch.int32), 4, -1, -2, 15)
    out = gapseed.fetch(ctx, res)
    assert 0 < res.n_small < len(probs)
    total = 0
    for i, p in enumerate(probs):
        eq, et, ident = O.refine_space(p["q"], p["t"], p["t_span"], p["K"], p["W"], p["diag"], 4, -1, -2, 15, p["q_add"], p["t_add"], p["flip"])
        a, b = int(out["pair_off"][i]), int(out["pair_off"][i + 1])
        assert b - a == len(eq), (i, b - a, len(eq))
        assert np.array_equal(out["pair_q"][a:b], eq) and np.array_equal(out["pair_t"][a:b], et), i
        assert np.float32(out["identity"][i]).view(np.uint32) == np.float32(ident).view(np.uint32), (i, out["identity"][i], ident)
        total += len(eq)
    assert total > 1000


@pytest.mark.gpu
def test_hip_refine_space_sketch_edges(ctx):
    """The long-gap branch's minimizer sketch (StoreMinimizers_noncanonical, MinCount.h:182-338) seen through RefineSpace on identical spans: every tuple of
    the query list meets its twin in the target list, so the pairs spell out the lists.  Ties (homopolymer and short tandem stretches), N runs that leave clean
    stretches of exactly / one less / one more than w + k - 1 bases, N at either end, windows of 2 .. 20 k-mers."""
    import torch
    from lra_amd import gapseed
    rng = np.random.default_rng(23)
    probs = []
    for i in range(400):
        K = int(rng.choice([6, 9, 12, 15, 19])); W = int(rng.choice([2, 3, 5, 8, 9, 10, 13, 16, 17, 20]))   # (w <= 8: 32-bit offset maps; 9 .. 16: 64-bit ones; beyond: the serial machine)
        span = W + K - 1
        L = int(rng.integers(1000, 1500))
        q = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)].copy()
        for _ in range(int(rng.integers(0, 6))):                                        # ties: homopolymers, dinucleotide and 7-mer tandem stretches
            a = int(rng.integers(0, L - 80)); n = int(rng.integers(5, 70)); unit = q[a:a + int(rng.choice([1, 2, 7]))].copy()
            q[a:a + n] = np.resize(unit, n)
        kind = i % 5
        if kind == 1:                                                                    # scattered N
            q[rng.integers(0, L, int(rng.integers(1, 9)))] = ord("N")
        elif kind == 2:                                                                  # clean stretches of span - 1, span, span + 1 between N
            at = int(rng.integers(0, 200))
            for d in (span - 1, span, span + 1, span + 2, 2 * span):
                q[at] = ord("N"); at += d + 1
            q[at] = ord("N")
        elif kind == 3:                                                                  # N at the ends; a clean tail of exactly span / span + 1
            q[0] = ord("N"); q[L - 1 - int(rng.choice([0, span, span + 1]))] = ord("N")
            if i % 2: q[1:int(rng.integers(2, 40))] = ord("N")
        elif kind == 4 and i % 10 == 4:
            q[:] = ord("A"); q[int(rng.integers(100, 900))] = ord("C")
        qb = q.tobytes()
        probs.append(dict(q=qb, t=qb, t_span=L, K=K, W=W, diag=int(rng.choice([5, 30])), q_add=0, t_add=0, flip=0))
    qcat = b"".join(p["q"] for p in probs); tcat = b"".join(p["t"] for p in probs)
    qoff = np.cumsum([0] + [len(p["q"]) for p in probs])[:-1]; toff = np.cumsum([0] + [len(p["t"]) for p in probs])[:-1]
    dev = ctx.device
    T = lambda a, dt: torch.tensor(np.asarray(a, dtype=dt), device=dev)
    dq = torch.tensor(np.frombuffer(qcat + b"\0" * 64, np.uint8).copy(), device=dev); dt_ = torch.tensor(np.frombuffer(tcat + b"\0" * 64, np.uint8).copy(), device=dev)
    res = gapseed.refine_space_batch(ctx, len(probs), dq, T(qoff, np.int64), T([len(p["q"]) for p in probs], np.int32), dt_, T(toff, np.int64),
                                     T([len(p["t"]) for p in probs], np.int32), T([p["t_span"] for p in probs], np.int64).to(torch.int32),
                                     T([p["K"] for p in probs], np.int32), T([p["W"] for p in probs], np.int32), T([p["diag"] for p in probs], np.int32),
                                     T([0] * len(probs), np.int32), T([0] * len(probs), np.int32), T([0] * len(probs), np.int32), 4, -1, -2, 200)
    out = gapseed.fetch(ctx, res)
    total = 0
    for i, p in enumerate(probs):
        eq, et, ident = O.refine_space(p["q"], p["t"], p["t_span"], p["K"], p["W"], p["diag"], 4, -1, -2, 200, 0, 0, 0)
        a, b = int(out["pair_off"][i]), int(out["pair_off"][i + 1])
        assert b - a == len(eq), (i, p["K"], p["W"], b - a, len(eq))
        assert np.array_equal(out["pair_q"][a:b], eq) and np.array_equal(out["pair_t"][a:b], et), (i, p["K"], p["W"])
        total += len(eq)
    assert total > 20000


@pytest.mark.gpu
def test_hip_between_anchors_oracle(ctx):
    """a13 DP leaf: RefineByLinearAlignment (LocalRefineAlignment.h:141-185) for consecutive anchor pairs"""
    import torch
    from lra_amd import gapseed
    rng = np.random.default_rng(4)
    genome = synth.make_genome(300000, seed=6)
    reads, truth = synth.simulate_reads(genome, 6, 8000, 500, 0.10, seed=2)
    g = genome.tobytes()
    roff = np.cumsum([0] + [len(r) for r in reads])
    rcat = b"".join(r.tobytes() for r in reads)
    Q = []; cases = []
    for i in range(600):
        r = int(rng.integers(0, len(reads))); s0 = truth[r][0]; L = len(reads[r])
        qs = int(rng.integers(0, L - 400)); ql = int(rng.choice([0, 0, 1, 5, 30, 120, 350]))
        ts = s0 + qs + int(rng.integers(-20, 20)); tl = max(0, ql + int(rng.choice([0, 0, -3, 4, 25, -ql, 60])))
        qe, te = qs + ql, ts + tl
        if i % 41 == 0: qe = qs - 1                     # m == 0: nothing aligned
        if i % 53 == 0: qe = max(0, qs - 5)             # negative span: the reference builds a string of negative length
        cases.append((r, qs, qe, ts, te))
    dev = ctx.device
    T = lambda a, dt: torch.tensor(np.asarray(a, dtype=dt), device=dev)
    dq = torch.tensor(np.frombuffer(rcat + b"\0" * 64, np.uint8).copy(), device=dev); dg = torch.tensor(np.frombuffer(g + b"\0" * 64, np.uint8).copy(), device=dev)
    for refine_dp in (1, 0):
        res = gapseed.between_anchors_batch(ctx, len(cases), dq, T([roff[c[0]] for c in cases], np.int64), T([c[1] for c in cases], np.int64).to(torch.int32),
                                            T([c[2] for c in cases], np.int64).to(torch.int32), dg, T([0] * len(cases), np.int64),
                                            T([c[3] for c in cases], np.int64).to(torch.int32), T([c[4] for c in cases], np.int64).to(torch.int32),
                                            4, -1, -2, 15, refine_dp)
        out = gapseed.fetch_between(ctx, res)
        nb = 0
        for i, (r, qs, qe, ts, te) in enumerate(cases):
            exp = O.between_anchors(reads[r].tobytes(), g, qs, qe, ts, te, 4, -1, -2, 15, refine_dp)
            a, b = int(out["block_off"][i]), int(out["block_off"][i + 1])
            if exp is None:
                assert out["status"][i] != 0 and b == a
                continue
            assert out["status"][i] == 0, i
            assert np.array_equal(out["blocks"][a:b], exp[0]), (i, qs, qe, ts, te)
            if len(exp[0]): assert out["score"][i] == exp[1], i
            nb += b - a
        assert (nb > 500) == bool(refine_dp)


def _junction(rng, read_len, chrom_len, seq_f, seq_r):
    """two segments with a gap of `span` forward read bases between them; any strand combination; a few blocks each"""
    span = int(rng.choice([1, 2, 7, 40, 180, 350, 499, 500, 700]))
    flqe = int(rng.integers(700, 1200)); frqs = flqe + span
    ls, rs = int(rng.integers(0, 2)), int(rng.integers(0, 2))

    def blocks(q0, t0, n, back):
        out = []; q, t = q0, t0
        for _ in range(n):
            ln = int(rng.integers(5, 60))
            if back: q -= ln; t -= ln
            out.append([q, t, ln])
            if not back: q += ln; t += ln
            g = int(rng.integers(0, 4))
            if back: q -= g; t -= int(rng.integers(0, 4))
            else: q += g; t += int(rng.integers(0, 4))
        return sorted(out)
    # left segment ends (forward coordinates) at flqe
    lt = int(rng.choice([rng.integers(2000, chrom_len - 2000), chrom_len - int(rng.integers(0, 300)), int(rng.integers(100, 400))]))
    if ls == 0: L = blocks(flqe, lt, int(rng.integers(1, 4)), True)                       # its last block ends at q = flqe
    else: L = blocks(read_len - flqe, lt, int(rng.integers(1, 4)), False)                 # reverse strand: GetQStart = readLen - flqe
    rt = int(rng.choice([rng.integers(2000, chrom_len - 2000), int(rng.integers(0, 300)), chrom_len - int(rng.integers(100, 400))]))
    if rs == 0: R = blocks(frqs, rt, int(rng.integers(1, 4)), False)                      # starts at q = frqs
    else: R = blocks(read_len - frqs, rt, int(rng.integers(1, 4)), True)                  # reverse strand: GetQEnd = readLen - frqs
    ok = all(b[0] >= 0 and b[1] >= 0 and b[0] + b[2] <= read_len and b[1] + b[2] <= chrom_len for b in L + R)
    return (L, ls, R, rs) if ok else None


@pytest.mark.gpu
def test_hip_refine_breakpoint_oracle(ctx):
    import torch
    from lra_amd import gapseed
    rng = np.random.default_rng(21)
    read_len, chrom_len = 3000, 20000
    alpha = np.frombuffer(b"AACCGT", np.uint8)                               # skewed alphabet: chance matches make the DP paths non-trivial
    seq_f = rng.choice(alpha, read_len).astype(np.uint8).tobytes(); seq_r = rng.choice(alpha, read_len).astype(np.uint8).tobytes()
    chrom = rng.choice(alpha, chrom_len).astype(np.uint8).tobytes()
    juncs = []
    while len(juncs) < 120:
        j = _junction(rng, read_len, chrom_len, seq_f, seq_r)
        if j: juncs.append(j)
    dev = ctx.device
    T = lambda a, dt: torch.tensor(np.asarray(a, dtype=dt), device=dev)
    seq = torch.tensor(np.frombuffer(seq_f + seq_r + b"\0" * 64, np.uint8).copy(), device=dev)
    gen = torch.tensor(np.frombuffer(chrom + b"\0" * 64, np.uint8).copy(), device=dev)
    lcat = np.array([b for j in juncs for b in j[0]], np.int32); rcat = np.array([b for j in juncs for b in j[2]], np.int32)
    loff = np.cumsum([0] + [len(j[0]) for j in juncs]); roff = np.cumsum([0] + [len(j[2]) for j in juncs])
    n = len(juncs)
    res = gapseed.refine_breakpoint_batch(ctx, n, T([read_len] * n, np.int32), seq, gen, T(lcat.reshape(-1), np.int32), T(loff, np.int64),
                                          T([j[1] for j in juncs], np.int32), T([j[1] * read_len for j in juncs], np.int64), T([0] * n, np.int64),
                                          T([chrom_len] * n, np.int32), T(rcat.reshape(-1), np.int32), T(roff, np.int64), T([j[3] for j in juncs], np.int32),
                                          T([j[3] * read_len for j in juncs], np.int64), T([0] * n, np.int64), T([chrom_len] * n, np.int32))
    out = gapseed.fetch_breakpoint(ctx, res, int(loff[-1]) + 502 * n, int(roff[-1]) + 502 * n)
    refined = 0
    for i, (L, ls, R, rs) in enumerate(juncs):
        ret, el, er = O.refine_breakpoint(read_len, L, ls, seq_r if ls else seq_f, chrom, R, rs, seq_r if rs else seq_f, chrom)
        a, b = int(out["l_off"][i]), int(out["r_off"][i])
        gl = out["l_blocks"][a:a + int(out["l_n"][i])]; gr = out["r_blocks"][b:b + int(out["r_n"][i])]
        if ret < 0:
            assert out["status"][i] & 1, i
            continue
        assert np.array_equal(gl, el), (i, ls, rs, gl.tolist(), el.tolist())
        assert np.array_equal(gr, er), (i, ls, rs)
        assert bool(out["status"][i] & 0x10000) == (ret == 1), i
        refined += ret == 1
    assert refined > 40
