"""Host-side mirror of IndelRefineAlignment (reference: IndelRefine.h:53) for a batch of alignments."""
import ctypes as C

import numpy as np
import torch

from .context import Context, ptr


class RefineResult(C.Structure):
    _fields_ = [("n_aln", C.c_int32), ("n_blocks", C.c_uint64), ("n_segments", C.c_uint64), ("n_rows", C.c_uint64),
                ("n_cells", C.c_uint64), ("n_aog", C.c_uint64), ("d_block_off", C.c_void_p), ("d_blocks", C.c_void_p),
                ("d_status", C.c_void_p)]


class RefineBatch:
    """Device-resident inputs: per alignment its blocks, the read strand it lies on and its chromosome."""

    def __init__(self, ctx: Context, blocks_list, q_seq_dev, q_off, q_len, t_seq_dev, t_off, t_len):
        self.ctx = ctx
        self.n = len(blocks_list)
        nb = np.fromiter((len(b) for b in blocks_list), dtype=np.int64, count=self.n)
        boff = np.zeros(self.n + 1, dtype=np.int64)
        boff[1:] = np.cumsum(nb)
        flat = (np.concatenate([np.asarray(b, dtype=np.int32).reshape(-1, 3) for b in blocks_list]) if boff[-1]
                else np.zeros((0, 3), np.int32))
        dev = ctx.device
        self.n_blocks_in = int(boff[-1])
        self.blocks = torch.from_numpy(np.ascontiguousarray(flat).reshape(-1)).to(dev) if boff[-1] else torch.zeros(3, dtype=torch.int32, device=dev)
        self.block_off = torch.from_numpy(boff).to(dev)
        self.q_seq, self.t_seq = q_seq_dev, t_seq_dev
        self.q_off = torch.from_numpy(np.asarray(q_off, dtype=np.int64)).to(dev)
        self.q_len = torch.from_numpy(np.asarray(q_len, dtype=np.int32)).to(dev)
        self.t_off = torch.from_numpy(np.asarray(t_off, dtype=np.int64)).to(dev)
        self.t_len = torch.from_numpy(np.asarray(t_len, dtype=np.int64)).to(dev)


def refine_batch_from_device(ctx, blocks, block_off, q_seq, q_off, q_len, t_seq, t_off, t_len):
    """RefineBatch whose inputs already are device tensors (blocks int32 [nb,3], CSR offsets int64)."""
    b = RefineBatch.__new__(RefineBatch)
    b.ctx, b.n = ctx, int(block_off.numel()) - 1
    b.n_blocks_in = int(blocks.shape[0])
    b.blocks = blocks.to(torch.int32).contiguous().reshape(-1)
    b.block_off = block_off.to(torch.int64).contiguous()
    b.q_seq, b.t_seq = q_seq, t_seq
    b.q_off, b.q_len = q_off.to(torch.int64).contiguous(), q_len.to(torch.int32).contiguous()
    b.t_off, b.t_len = t_off.to(torch.int64).contiguous(), t_len.to(torch.int64).contiguous()
    return b


def indel_refine_batch(ctx: Context, b: RefineBatch, refine_band, match, mismatch, indel, end_align=False):
    res = RefineResult()
    ctx.check(ctx.lib.lra_indel_refine_batch(ctx.h, b.n, ptr(b.blocks), ptr(b.block_off), C.c_uint64(b.n_blocks_in), ptr(b.q_seq),
                                             ptr(b.q_off), ptr(b.q_len), ptr(b.t_seq), ptr(b.t_off), ptr(b.t_len), refine_band,
                                             match, mismatch, indel, 1 if end_align else 0, C.byref(res)))
    return res


def fetch(ctx: Context, res: RefineResult):
    off = ctx.to_host(res.d_block_off, res.n_aln + 1, np.uint64)
    blocks = ctx.to_host(res.d_blocks, 3 * res.n_blocks, np.int32).reshape(-1, 3)
    status = ctx.to_host(res.d_status, res.n_aln, np.int32)
    return [blocks[int(off[a]):int(off[a + 1])] for a in range(res.n_aln)], status


class StatsResult(C.Structure):
    _fields_ = [("n_aln", C.c_int32), ("n_runs", C.c_uint64), ("d_counts", C.c_void_p), ("d_value", C.c_void_p),
                ("d_run_off", C.c_void_p), ("d_runs", C.c_void_p)]


STAT_NAMES = ["nm", "nmm", "nins", "ndel", "tdel", "tins", "nSmallDel", "nMedDel", "nLargeDel", "nSmallIns", "nMedIns", "nLargeIns",
              "preClip", "sufClip", "qStart", "qEnd", "tStart", "tEnd"]


def calculate_statistics_batch(ctx: Context, b: RefineBatch, lookup_table):
    """Alignment::CalculateStatistics over the alignments of a RefineBatch-shaped input (blocks + sequences).
    lookup_table: float32[2001] = logf(1), logf(6), ... from the host libm."""
    lut = np.ascontiguousarray(lookup_table, dtype=np.float32)
    res = StatsResult()
    ctx.check(ctx.lib.lra_calculate_statistics_batch(ctx.h, b.n, ptr(b.blocks), ptr(b.block_off), ptr(b.q_seq), ptr(b.q_off), ptr(b.q_len),
                                                     ptr(b.t_seq), ptr(b.t_off), C.c_void_p(lut.ctypes.data), len(lut), C.byref(res)))
    return res


def fetch_stats(ctx: Context, res: StatsResult):
    counts = ctx.to_host(res.d_counts, 18 * res.n_aln, np.int32).reshape(-1, 18)
    value = ctx.to_host(res.d_value, res.n_aln, np.float32)
    off = ctx.to_host(res.d_run_off, res.n_aln + 1, np.uint64)
    runs = ctx.to_host(res.d_runs, res.n_runs, np.uint32)
    cigars = ["".join("%d%s" % (r >> 4, "=XID"[r & 15]) for r in runs[int(off[a]):int(off[a + 1])]) for a in range(res.n_aln)]
    return counts, value, cigars


def stats_of_refined(ctx: Context, b: RefineBatch, rres: RefineResult, lookup_table):
    """CalculateStatistics straight on the (context-owned) output of indel_refine_batch, no copies."""
    v = RefineBatch.__new__(RefineBatch)
    v.ctx, v.n, v.n_blocks_in = ctx, b.n, int(rres.n_blocks)
    v.blocks, v.block_off = int(rres.d_blocks), int(rres.d_block_off)
    v.q_seq, v.q_off, v.q_len, v.t_seq, v.t_off, v.t_len = b.q_seq, b.q_off, b.q_len, b.t_seq, b.t_off, b.t_len
    return calculate_statistics_batch(ctx, v, lookup_table)
