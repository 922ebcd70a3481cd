"""Host-side mirror of RefineSpace (reference: ClusterRefine.h:242-325) for a batch of gaps."""
import ctypes as C

import numpy as np

from .context import Context, ptr


class RefineSpaceResult(C.Structure):
    _fields_ = [("n_problems", C.c_uint64), ("n_pairs", C.c_uint64), ("n_small", C.c_uint64), ("d_pair_off", C.c_void_p), ("d_pair_q", C.c_void_p),
                ("d_pair_t", C.c_void_p), ("d_identity", C.c_void_p), ("d_status", C.c_void_p)]


def refine_space_batch(ctx: Context, n, qseq, q_off, q_len, tseq, t_off, t_len, t_span, K, W, diag, q_add, t_add, flip_len, match=4, mismatch=-1,
                       indel=-2, max_freq=15):
    """All array arguments are device tensors (see include/lra_hip.h)."""
    res = RefineSpaceResult()
    ctx.check(ctx.lib.lra_refine_space_batch(ctx.h, int(n), ptr(qseq), ptr(q_off), ptr(q_len), ptr(tseq), ptr(t_off), ptr(t_len), ptr(t_span), ptr(K),
                                             ptr(W), ptr(diag), ptr(q_add), ptr(t_add), ptr(flip_len), int(match), int(mismatch), int(indel),
                                             int(max_freq), C.byref(res)))
    return res


def fetch(ctx: Context, res: RefineSpaceResult):
    n, m = res.n_problems, res.n_pairs
    return {"pair_off": ctx.to_host(res.d_pair_off, n + 1, np.uint64), "pair_q": ctx.to_host(res.d_pair_q, m, np.uint32),
            "pair_t": ctx.to_host(res.d_pair_t, m, np.uint32), "identity": ctx.to_host(res.d_identity, n, np.float32),
            "status": ctx.to_host(res.d_status, n, np.uint32)}


class BetweenResult(C.Structure):
    _fields_ = [("n_gaps", C.c_uint64), ("n_blocks", C.c_uint64), ("d_block_off", C.c_void_p), ("d_blocks", C.c_void_p), ("d_score", C.c_void_p),
                ("d_status", C.c_void_p)]


def between_anchors_batch(ctx: Context, n, qseq, q_base, cur_read_end, next_read_start, tseq, t_base, cur_genome_end, next_genome_start, match=4, mismatch=-1,
                          indel=-2, local_band=15, refine_dp=1):
    """RefineByLinearAlignment (LocalRefineAlignment.h:141) for n anchor pairs; array arguments are device tensors."""
    res = BetweenResult()
    ctx.check(ctx.lib.lra_between_anchors_batch(ctx.h, int(n), ptr(qseq), ptr(q_base), ptr(cur_read_end), ptr(next_read_start), ptr(tseq), ptr(t_base),
                                                ptr(cur_genome_end), ptr(next_genome_start), int(match), int(mismatch), int(indel), int(local_band),
                                                int(refine_dp), C.byref(res)))
    return res


def fetch_between(ctx: Context, res: BetweenResult):
    n, m = res.n_gaps, res.n_blocks
    return {"block_off": ctx.to_host(res.d_block_off, n + 1, np.uint64), "blocks": ctx.to_host(res.d_blocks, 3 * m, np.int32).reshape(-1, 3),
            "score": ctx.to_host(res.d_score, n, np.int32), "status": ctx.to_host(res.d_status, n, np.uint32)}


class BreakpointResult(C.Structure):
    _fields_ = [("n_junctions", C.c_uint64), ("d_l_blocks", C.c_void_p), ("d_l_off", C.c_void_p), ("d_l_n", C.c_void_p), ("d_r_blocks", C.c_void_p),
                ("d_r_off", C.c_void_p), ("d_r_n", C.c_void_p), ("d_status", C.c_void_p)]


def refine_breakpoint_batch(ctx: Context, n, read_len, seq, genome, l_blocks, l_off, l_strand, l_read_off, l_chrom_off, l_chrom_len, r_blocks, r_off,
                            r_strand, r_read_off, r_chrom_off, r_chrom_len):
    """RefineBreakpoint (RefineBreakpoint.h:210) for n junctions; array arguments are device tensors."""
    res = BreakpointResult()
    ctx.check(ctx.lib.lra_refine_breakpoint_batch(ctx.h, int(n), ptr(read_len), ptr(seq), ptr(genome), ptr(l_blocks), ptr(l_off), ptr(l_strand),
                                                  ptr(l_read_off), ptr(l_chrom_off), ptr(l_chrom_len), ptr(r_blocks), ptr(r_off), ptr(r_strand),
                                                  ptr(r_read_off), ptr(r_chrom_off), ptr(r_chrom_len), C.byref(res)))
    return res


def fetch_breakpoint(ctx: Context, res: BreakpointResult, l_total_cap, r_total_cap):
    n = res.n_junctions
    return {"l_off": ctx.to_host(res.d_l_off, n + 1, np.uint64), "l_n": ctx.to_host(res.d_l_n, n, np.int32),
            "l_blocks": ctx.to_host(res.d_l_blocks, 3 * l_total_cap, np.int32).reshape(-1, 3),
            "r_off": ctx.to_host(res.d_r_off, n + 1, np.uint64), "r_n": ctx.to_host(res.d_r_n, n, np.int32),
            "r_blocks": ctx.to_host(res.d_r_blocks, 3 * r_total_cap, np.int32).reshape(-1, 3), "status": ctx.to_host(res.d_status, n, np.uint32)}
