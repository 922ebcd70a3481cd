"""Host-side mirror of the tier-2 (local) minimizer machinery: LocalIndex::IndexSeq (MMIndex.h:200) and
CompareLists<LocalTuple,SmallTuple> (CompareLists.h:9) for batches."""
import ctypes as C

import numpy as np
import torch

from .context import Context, ptr


class LocalIndexResult(C.Structure):
    _fields_ = [("n_seqs", C.c_int32), ("n_windows", C.c_uint64), ("n_tuples", C.c_uint64), ("bytes", C.c_uint64), ("d_base", C.c_void_p),
                ("d_win_off", C.c_void_p), ("d_tuple_bnd", C.c_void_p), ("d_tuples", C.c_void_p)]


class LocalPairsResult(C.Structure):
    _fields_ = [("n_tasks", C.c_uint64), ("n_pairs", C.c_uint64), ("d_pair_off", C.c_void_p), ("d_pair_qi", C.c_void_p), ("d_pair_ti", C.c_void_p)]


class LocalIndex:
    """Device-resident local index of a set of sequences (owns its buffer)."""

    def __init__(self, ctx: Context, seq_dev, off_dev, k=10, w=5, window=256, max_freq=15):
        self.ctx, self.k, self.w, self.window, self.max_freq = ctx, k, w, window, max_freq
        n = int(off_dev.numel()) - 1
        res = LocalIndexResult()
        ctx.check(ctx.lib.lra_local_index_batch(ctx.h, n, ptr(seq_dev), ptr(off_dev), k, w, window, max_freq, C.byref(res)))
        # keep it: copy out of the context-owned buffer and re-base the three pointers
        self.buf = ctx.to_tensor(res.d_base, int(res.bytes), torch.uint8)
        base_old, base_new = int(res.d_base), self.buf.data_ptr()
        res.d_win_off = base_new + (int(res.d_win_off) - base_old)
        res.d_tuple_bnd = base_new + (int(res.d_tuple_bnd) - base_old)
        res.d_tuples = base_new + (int(res.d_tuples) - base_old)
        res.d_base = base_new
        self.res = res
        self.n_seqs, self.n_windows, self.n_tuples = n, int(res.n_windows), int(res.n_tuples)

    def bnd_tensor(self):
        """tupleBoundaries as a device int64 tensor view into the index buffer (n_windows + 1 entries)."""
        a = lambda n: (n + 255) & ~255
        start = a((self.n_seqs + 1) * 8)
        return self.buf[start:start + (self.n_windows + 1) * 8].view(torch.int64)

    def fetch(self):
        c, r = self.ctx, self.res
        return (c.to_host(r.d_win_off, self.n_seqs + 1, np.uint64), c.to_host(r.d_tuple_bnd, self.n_windows + 1, np.uint64),
                c.to_host(r.d_tuples, self.n_tuples, np.uint32))


def local_compare_batch(ctx: Context, q_index: LocalIndex, q_lo, q_hi, t_index: LocalIndex, t_lo, t_hi, max_freq, max_diag=None, min_diag=None, fetch=True):
    """Tasks: tuple index ranges [q_lo,q_hi) of q_index vs [t_lo,t_hi) of t_index.  Returns (pair_off, qi, ti) on the host."""
    dev = ctx.device
    tt = lambda a, dt: a.to(torch.int64).contiguous() if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    ql, qh, tl, th = tt(q_lo, np.int64), tt(q_hi, np.int64), tt(t_lo, np.int64), tt(t_hi, np.int64)
    mx = tt(max_diag, np.int64) if max_diag is not None else None
    mn = tt(min_diag, np.int64) if min_diag is not None else None
    res = LocalPairsResult()
    ctx.check(ctx.lib.lra_local_compare_batch(ctx.h, C.c_uint64(len(q_lo)), C.c_void_p(q_index.res.d_tuples), ptr(ql), ptr(qh),
                                              C.c_void_p(t_index.res.d_tuples), ptr(tl), ptr(th), int(max_freq), ptr(mx), ptr(mn), C.byref(res)))
    n = len(q_lo)
    if not fetch:
        return res
    return (ctx.to_host(res.d_pair_off, n + 1, np.uint64), ctx.to_host(res.d_pair_qi, res.n_pairs, np.uint32),
            ctx.to_host(res.d_pair_ti, res.n_pairs, np.uint32))
