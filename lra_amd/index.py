"""`lra index` on the device and the index files: ctypes mirror of lra_ctx_build_global_index / lra_write_mms / lra_read_mms / lra_write_gli /
lra_read_gli (include/lra_hip.h; reference MMIndex.h:286-424, :138-173).  No algorithmic code here."""
import ctypes as C

import numpy as np
import torch

from ._lib import load_library
from .context import Context

# index-time presets (lra.cpp:884-911): K, W, globalMaxFreq, globalWinsize, NumOfminimizersPerWindow
INDEX_PRESETS = {"ont": (17, 10, 150, 15, 1), "ccs": (17, 10, 150, 15, 1), "clr": (15, 10, 250, 12, 1), "contig": (19, 10, 30, 20, 1)}


def load_genome(ctx: Context, genome):
    """genome: uint8 bases of all sequences back to back, numpy array or device tensor."""
    if torch.is_tensor(genome):
        g = genome.contiguous()
        if g.is_cuda:
            ctx.check(ctx.lib.lra_ctx_load_genome_device(ctx.h, C.c_void_p(g.data_ptr()), C.c_uint64(g.numel())))
            return
        genome = g.numpy()
    g = np.ascontiguousarray(genome, dtype=np.uint8)
    ctx.check(ctx.lib.lra_ctx_load_genome(ctx.h, C.c_void_p(g.ctypes.data), C.c_uint64(len(g))))


def build_global_index(ctx: Context, chrom_pos, k=17, w=10, max_freq=150, winsize=15, n_per_window=1):
    """StoreIndex on the genome loaded into ctx; installs the result as the context's global index.  -> dict(n_minimizers, n_index, status)."""
    cp = (C.c_uint64 * len(chrom_pos))(*[int(x) for x in chrom_pos])
    nm, ni, st = C.c_uint64(0), C.c_uint64(0), C.c_int(0)
    ctx.check(ctx.lib.lra_ctx_build_global_index(ctx.h, cp, len(chrom_pos) - 1, int(k), int(w), int(max_freq), int(winsize), int(n_per_window),
                                                 C.byref(nm), C.byref(ni), C.byref(st)))
    return dict(n_minimizers=nm.value, n_index=ni.value, status=st.value)


def global_index(ctx: Context):
    """The context's global index as host arrays (key uint64, pos uint32)."""
    dk, dp, n = C.c_void_p(), C.c_void_p(), C.c_uint64(0)
    ctx.check(ctx.lib.lra_ctx_global_index(ctx.h, C.byref(dk), C.byref(dp), C.byref(n)))
    if n.value == 0:
        return np.zeros(0, np.uint64), np.zeros(0, np.uint32)
    return ctx.to_host(dk.value, n.value, np.uint64), ctx.to_host(dp.value, n.value, np.uint32)


def _chk(rc, what):
    if rc != 0:
        raise IOError("%s failed (%d)" % (what, rc))


def write_mms(path, globalK, chrom_names, chrom_pos, key, pos):
    lib = load_library()
    names = [n if isinstance(n, bytes) else str(n).encode() for n in chrom_names]
    a_names = (C.c_char_p * len(names))(*names)
    cp = np.ascontiguousarray(chrom_pos, dtype=np.uint64)
    k = np.ascontiguousarray(key, dtype=np.uint64); p = np.ascontiguousarray(pos, dtype=np.uint32)
    _chk(lib.lra_write_mms(str(path).encode(), int(globalK), a_names, C.c_void_p(cp.ctypes.data), len(names), C.c_void_p(k.ctypes.data),
                           C.c_void_p(p.ctypes.data), C.c_uint64(len(k))), "lra_write_mms")


def read_mms(path):
    """-> dict(globalK, names, chrom_pos, key, pos)"""
    lib = load_library()
    K, n, nc, nl = C.c_int(0), C.c_uint64(0), C.c_int(0), C.c_uint64(0)
    _chk(lib.lra_read_mms(str(path).encode(), C.byref(K), C.byref(n), C.byref(nc), C.byref(nl), None, None, None, None), "lra_read_mms")
    names = C.create_string_buffer(max(1, nl.value))
    cp = np.zeros(nc.value + 1, np.uint64); key = np.zeros(max(1, n.value), np.uint64); pos = np.zeros(max(1, n.value), np.uint32)
    _chk(lib.lra_read_mms(str(path).encode(), C.byref(K), C.byref(n), C.byref(nc), C.byref(nl), names, C.c_void_p(cp.ctypes.data), C.c_void_p(key.ctypes.data),
                          C.c_void_p(pos.ctypes.data)), "lra_read_mms")
    key = key[:n.value]; pos = pos[:n.value]
    return dict(globalK=K.value, names=names.raw[:nl.value].split(b"\0")[:nc.value], chrom_pos=cp, key=key, pos=pos)


def write_gli(path, k, w, window, seq_offsets, tuple_bnd, tuples):
    lib = load_library()
    so = np.ascontiguousarray(seq_offsets, dtype=np.uint64); tb = np.ascontiguousarray(tuple_bnd, dtype=np.uint64)
    tu = np.ascontiguousarray(tuples, dtype=np.uint32)
    assert len(so) == len(tb) and int(tb[-1]) == len(tu)
    _chk(lib.lra_write_gli(str(path).encode(), int(k), int(w), int(window), C.c_uint64(len(so) - 1), C.c_void_p(so.ctypes.data), C.c_void_p(tb.ctypes.data),
                           C.c_void_p(tu.ctypes.data)), "lra_write_gli")


def read_gli(path):
    """-> dict(k, w, window, seq_offsets, tuple_bnd, tuples)"""
    lib = load_library()
    k, w, win, nw, nt = C.c_int(0), C.c_int(0), C.c_int(0), C.c_uint64(0), C.c_uint64(0)
    _chk(lib.lra_read_gli(str(path).encode(), C.byref(k), C.byref(w), C.byref(win), C.byref(nw), C.byref(nt), None, None, None), "lra_read_gli")
    so = np.zeros(nw.value + 1, np.uint64); tb = np.zeros(nw.value + 1, np.uint64); tu = np.zeros(max(1, nt.value), np.uint32)
    _chk(lib.lra_read_gli(str(path).encode(), C.byref(k), C.byref(w), C.byref(win), C.byref(nw), C.byref(nt), C.c_void_p(so.ctypes.data),
                          C.c_void_p(tb.ctypes.data), C.c_void_p(tu.ctypes.data)), "lra_read_gli")
    return dict(k=k.value, w=w.value, window=win.value, seq_offsets=so, tuple_bnd=tb, tuples=tu[:nt.value])
