"""Host-side mirror of SparseDP (SDP#A; reference: SparseDP.h:2139, called at Map_lowacc.h:188) for a batch of reads."""
import ctypes as C

import numpy as np

from .context import Context, ptr


class SdpOpts(C.Structure):
    _fields_ = [("rate", C.c_float), ("NumAln", C.c_int32), ("alnthres", C.c_float), ("gapopen", C.c_float), ("gapextend", C.c_float),
                ("gaproot", C.c_float), ("gapCeiling1", C.c_int32), ("gapCeiling2", C.c_int32)]


# -ONT preset (lra.cpp:388-420; alnthres: Options.h:198)
ONT = dict(rate=20.0, NumAln=2, alnthres=0.7, gapopen=7.0, gapextend=10.0, gaproot=1.5, gapCeiling1=1500, gapCeiling2=3000)


def sdp_opts(**kw):
    d = dict(ONT); d.update(kw)
    return SdpOpts(d["rate"], d["NumAln"], d["alnthres"], d["gapopen"], d["gapextend"], d["gaproot"], d["gapCeiling1"], d["gapCeiling2"])


class ChainResult(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("num_aln", C.c_int32), ("n_frags", C.c_uint64), ("n_points", C.c_uint64),
                ("n_subproblem_entries", C.c_uint64), ("d_n_chains", C.c_void_p), ("d_chain_start", C.c_void_p), ("d_chain_len", C.c_void_p),
                ("d_chain_box", C.c_void_p), ("d_chain_value", C.c_void_p), ("d_chain_cluster", C.c_void_p), ("d_chain_anchor", C.c_void_p),
                ("d_chain_link", C.c_void_p), ("d_frag_off", C.c_void_p), ("d_frag_val", C.c_void_p), ("d_status", C.c_void_p)]


def sparse_dp_batch(ctx: Context, n_reads, cluster_off, c_start, c_count, c_strand, q, t, length, read_off, opts: SdpOpts, rate=None):
    """All array arguments are device tensors or raw device addresses (see include/lra_hip.h)."""
    res = ChainResult()
    ctx.check(ctx.lib.lra_sparse_dp_batch(ctx.h, int(n_reads), ptr(cluster_off), ptr(c_start), ptr(c_count), ptr(c_strand), ptr(q), ptr(t),
                                          ptr(length), ptr(read_off), ptr(rate) if rate is not None else None, C.byref(opts), C.byref(res)))
    return res


def fetch(ctx: Context, res: ChainResult):
    n, na, nf = res.n_reads, res.num_aln, res.n_frags
    return {"n_chains": ctx.to_host(res.d_n_chains, n, np.uint32), "chain_start": ctx.to_host(res.d_chain_start, n * na, np.uint64),
            "chain_len": ctx.to_host(res.d_chain_len, n * na, np.uint32), "chain_box": ctx.to_host(res.d_chain_box, 4 * n * na, np.uint32).reshape(-1, 4),
            "chain_value": ctx.to_host(res.d_chain_value, n * na, np.float32), "chain_cluster": ctx.to_host(res.d_chain_cluster, nf, np.uint32),
            "chain_anchor": ctx.to_host(res.d_chain_anchor, nf, np.uint32), "chain_link": ctx.to_host(res.d_chain_link, nf, np.uint8),
            "frag_off": ctx.to_host(res.d_frag_off, n + 1, np.uint64), "frag_val": ctx.to_host(res.d_frag_val, nf, np.float32),
            "status": ctx.to_host(res.d_status, n, np.uint32)}
