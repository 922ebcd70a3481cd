"""Host-side mirror of the reference's base-level DP entry points.

AffineOneGapAlign (reference: AffineOneGapAlign.h:157) -> affine_one_gap_align_batch.
"""
import numpy as np
import torch

from .context import Context, ptr


def pack_strings(strings):
    """Concatenate byte strings; return (uint8 array, offsets uint64, lengths int32)."""
    lens = np.fromiter((len(s) for s in strings), dtype=np.int64, count=len(strings))
    off = np.zeros(len(strings), dtype=np.uint64)
    if len(strings) > 1:
        off[1:] = np.cumsum(lens[:-1]).astype(np.uint64)
    buf = np.frombuffer(b"".join(strings), dtype=np.uint8) if lens.sum() else np.zeros(0, dtype=np.uint8)
    return buf, off, lens.astype(np.int32)


class AogBatch:
    """Device-resident batch of AffineOneGapAlign problems (inputs + output buffers)."""

    def __init__(self, ctx: Context, q_list, t_list, k_list, m, mm, indel):
        n = len(q_list)
        assert len(t_list) == n and len(k_list) == n
        self.ctx, self.n, self.m, self.mm, self.indel = ctx, n, m, mm, indel
        buf, off, lens = pack_strings(list(q_list) + list(t_list))
        dev = ctx.device
        self.seq = torch.from_numpy(np.concatenate([buf, np.zeros(16, np.uint8)])).to(dev)
        self.q_off = torch.from_numpy(off[:n].astype(np.int64)).to(dev)
        self.t_off = torch.from_numpy(off[n:].astype(np.int64)).to(dev)
        self.q_len = torch.from_numpy(lens[:n].copy()).to(dev)
        self.t_len = torch.from_numpy(lens[n:].copy()).to(dev)
        self.k = torch.tensor(np.asarray(k_list, dtype=np.int32)).to(dev)
        cap = np.minimum(lens[:n], lens[n:]).astype(np.int64) + 1
        boff = np.zeros(n + 1, dtype=np.int64)
        boff[1:] = np.cumsum(cap)
        self.block_off_h = boff
        self.block_off = torch.from_numpy(boff).to(dev)
        self.score = torch.empty(n, dtype=torch.int32, device=dev)
        self.nblocks = torch.empty(n, dtype=torch.int32, device=dev)
        self.status = torch.empty(n, dtype=torch.int32, device=dev)
        self.blocks = torch.empty(max(1, int(boff[-1])) * 3, dtype=torch.int32, device=dev)

    @classmethod
    def from_device(cls, ctx, qseq, tseq, q_off, q_len, t_off, t_len, k, m, mm, indel):
        """Problems described by device tensors: offsets into the device buffers qseq / tseq."""
        self = cls.__new__(cls)
        self.ctx, self.n, self.m, self.mm, self.indel = ctx, int(q_off.numel()), m, mm, indel
        self.seq, self.tseq = qseq, tseq
        self.q_off, self.t_off = q_off.to(torch.int64).contiguous(), t_off.to(torch.int64).contiguous()
        self.q_len, self.t_len = q_len.to(torch.int32).contiguous(), t_len.to(torch.int32).contiguous()
        self.k = k.to(torch.int32).contiguous()
        cap = torch.minimum(self.q_len, self.t_len).to(torch.int64) + 1
        boff = torch.zeros(self.n + 1, dtype=torch.int64, device=ctx.device)
        boff[1:] = torch.cumsum(cap, 0)
        self.block_off = boff
        self.block_off_h = None
        self.score = torch.empty(self.n, dtype=torch.int32, device=ctx.device)
        self.nblocks = torch.empty(self.n, dtype=torch.int32, device=ctx.device)
        self.status = torch.empty(self.n, dtype=torch.int32, device=ctx.device)
        self.blocks = torch.empty(max(1, int(boff[-1])) * 3, dtype=torch.int32, device=ctx.device)
        return self

    def run(self):
        """Launch the batch on the context's stream (asynchronous)."""
        c = self.ctx
        c.check(c.lib.lra_affine_one_gap_align_batch(
            c.h, self.n, ptr(self.seq), ptr(getattr(self, "tseq", self.seq)), ptr(self.q_off), ptr(self.q_len), ptr(self.t_off), ptr(self.t_len),
            ptr(self.k), self.m, self.mm, self.indel, ptr(self.score), ptr(self.nblocks), ptr(self.blocks),
            ptr(self.block_off), ptr(self.status)))

    def results(self):
        """Synchronise and return (scores, list of (nb,3) block arrays, status) on the host."""
        torch.cuda.synchronize(self.ctx.device)
        if self.block_off_h is None:
            self.block_off_h = self.block_off.cpu().numpy()
        score = self.score.cpu().numpy()
        nb = self.nblocks.cpu().numpy()
        st = self.status.cpu().numpy()
        blocks = self.blocks.cpu().numpy().reshape(-1, 3)
        out = []
        for p in range(self.n):
            o = int(self.block_off_h[p])
            cap = int(self.block_off_h[p + 1] - self.block_off_h[p])
            out.append(blocks[o:o + min(int(nb[p]), cap)].copy())
        return score, out, st


def affine_one_gap_align_batch(ctx, q_list, t_list, k_list, m, mm, indel):
    """Batched AffineOneGapAlign: returns (scores int32[n], blocks list, status int32[n])."""
    b = AogBatch(ctx, q_list, t_list, k_list, m, mm, indel)
    b.run()
    return b.results()
