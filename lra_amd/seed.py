"""Host-side mirror of the tier-1 seeding stages of MapRead (reference: MapRead.h:169-203):
StoreMinimizers -> sort -> CompareLists -> SeparateMatchesByStrand, for a batch of reads."""
import ctypes as C

import numpy as np
import torch

from .context import Context, ptr


class SeedResult(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("n_minimizers", C.c_uint64), ("n_matches", C.c_uint64),
                ("d_mm_off", C.c_void_p), ("d_mm_key", C.c_void_p), ("d_mm_pos", C.c_void_p),
                ("d_match_off", C.c_void_p), ("d_match_qi", C.c_void_p), ("d_match_ti", C.c_void_p),
                ("d_n_forward", C.c_void_p), ("d_sep_qpos", C.c_void_p), ("d_sep_tpos", C.c_void_p)]


def load_reference(ctx: Context, genome_u8, idx_key, idx_pos):
    """Replicate genome bytes and the global minimizer index (`.mms` payload) into this GPU's HBM."""
    g = np.ascontiguousarray(genome_u8, dtype=np.uint8)
    k = np.ascontiguousarray(idx_key).view(np.uint64)
    p = np.ascontiguousarray(idx_pos, dtype=np.uint32)
    assert len(k) == len(p)
    ctx.check(ctx.lib.lra_ctx_load_genome(ctx.h, C.c_void_p(g.ctypes.data), C.c_uint64(len(g))))
    ctx.check(ctx.lib.lra_ctx_load_global_index(ctx.h, C.c_void_p(k.ctypes.data), C.c_void_p(p.ctypes.data), C.c_uint64(len(k))))


class ReadBatch:
    """Reads resident in HBM: concatenated upper-case bytes + CSR offsets."""

    def __init__(self, ctx: Context, reads):
        lens = np.fromiter((len(r) for r in reads), dtype=np.int64, count=len(reads))
        off = np.zeros(len(reads) + 1, dtype=np.int64)
        off[1:] = np.cumsum(lens)
        buf = np.frombuffer(b"".join(bytes(r) for r in reads), dtype=np.uint8) if off[-1] else np.zeros(0, np.uint8)
        self.n = len(reads)
        self.total_bases = int(off[-1])
        self.seq = torch.from_numpy(np.concatenate([buf, np.zeros(64, np.uint8)])).to(ctx.device)
        self.off = torch.from_numpy(off).to(ctx.device)
        self.off_h = off


def read_batch_from_device(ctx, seq, off):
    """ReadBatch over device tensors (seq uint8 with >= 64 bytes of padding after the last read, off int64 [n+1])."""
    b = ReadBatch.__new__(ReadBatch)
    b.n = int(off.numel()) - 1
    b.seq, b.off = seq, off.to(torch.int64).contiguous()
    b.off_h = None
    b.total_bases = int(off[-1])
    return b


def seed_batch(ctx: Context, batch: ReadBatch, k, w, max_freq):
    """Run a1-a4 on the batch; returns the SeedResult struct (device pointers owned by ctx)."""
    res = SeedResult()
    ctx.check(ctx.lib.lra_seed_batch(ctx.h, batch.n, ptr(batch.seq), ptr(batch.off), k, w, max_freq, C.byref(res)))
    return res


def seed_prefetch(side: Context, batch: ReadBatch, k, w, max_freq):
    """a1-a4 of a batch AHEAD of its mapping call, on a side context (own host thread, own stream; shares the mapping context's reference data); the mapping
    context takes the result with adopt_seed and its next lra_map_reads_*_batch on the same batch starts from it (include/lra_hip.h: lra_seed_prefetch)."""
    side.check(side.lib.lra_seed_prefetch(side.h, batch.n, ptr(batch.seq), ptr(batch.off), k, w, max_freq))


def adopt_seed(ctx: Context, side: Context):
    ctx.check(ctx.lib.lra_ctx_adopt_seed(ctx.h, side.h))


def fetch(ctx: Context, res: SeedResult):
    """Copy a SeedResult to host numpy arrays (dict)."""
    n = res.n_reads
    return {
        "mm_off": ctx.to_host(res.d_mm_off, n + 1, np.uint64),
        "mm_key": ctx.to_host(res.d_mm_key, res.n_minimizers, np.uint64),
        "mm_pos": ctx.to_host(res.d_mm_pos, res.n_minimizers, np.uint32),
        "match_off": ctx.to_host(res.d_match_off, n + 1, np.uint64),
        "match_qi": ctx.to_host(res.d_match_qi, res.n_matches, np.uint32),
        "match_ti": ctx.to_host(res.d_match_ti, res.n_matches, np.uint32),
        "n_forward": ctx.to_host(res.d_n_forward, n, np.uint32),
        "sep_qpos": ctx.to_host(res.d_sep_qpos, res.n_matches, np.uint32),
        "sep_tpos": ctx.to_host(res.d_sep_tpos, res.n_matches, np.uint32),
    }


def sort_minimizers_batch(ctx: Context, keys_list, pos_list):
    """a2 alone (std::sort emulation) on a list of (keys uint64, pos uint32) arrays; returns sorted copies."""
    n = len(keys_list)
    lens = np.array([len(k) for k in keys_list], dtype=np.int64)
    off = np.zeros(n + 1, dtype=np.int64)
    off[1:] = np.cumsum(lens)
    K = np.concatenate([np.asarray(k, dtype=np.uint64) for k in keys_list] + [np.zeros(1, np.uint64)])
    P = np.concatenate([np.asarray(p, dtype=np.uint32) for p in pos_list] + [np.zeros(1, np.uint32)])
    dk = torch.from_numpy(K.view(np.int64)).to(ctx.device)
    dp = torch.from_numpy(P.view(np.int32)).to(ctx.device)
    do = torch.from_numpy(off).to(ctx.device)
    ctx.check(ctx.lib.lra_sort_minimizers_batch(ctx.h, n, ptr(do), ptr(dk), ptr(dp)))
    torch.cuda.synchronize(ctx.device)
    K2 = dk.cpu().numpy().view(np.uint64)
    P2 = dp.cpu().numpy().view(np.uint32)
    return [(K2[off[i]:off[i + 1]].copy(), P2[off[i]:off[i + 1]].copy()) for i in range(n)]


def sort_pairs_batch(ctx: Context, keys, vals, begin, end, begin_bit=0, end_bit=64):
    """lra_sort_pairs_batch: every segment [begin[i], end[i]) of (uint64 key, uint32 value) pairs sorted stably by the key bits [begin_bit, end_bit); returns
    (keys_out, vals_out) -- positions outside every segment are left as the output buffers were (zero here)."""
    K = np.ascontiguousarray(np.asarray(keys, dtype=np.uint64)); V = np.ascontiguousarray(np.asarray(vals, dtype=np.uint32))
    B = np.ascontiguousarray(np.asarray(begin, dtype=np.uint64)); E = np.ascontiguousarray(np.asarray(end, dtype=np.uint64))
    pad = lambda a, dt: np.concatenate([a, np.zeros(1, dt)])
    dk = torch.from_numpy(pad(K, np.uint64).view(np.int64)).to(ctx.device); dv = torch.from_numpy(pad(V, np.uint32).view(np.int32)).to(ctx.device)
    ok = torch.zeros_like(dk); ov = torch.zeros_like(dv)
    db = torch.from_numpy(pad(B, np.uint64).view(np.int64)).to(ctx.device); de = torch.from_numpy(pad(E, np.uint64).view(np.int64)).to(ctx.device)
    ctx.check(ctx.lib.lra_sort_pairs_batch(ctx.h, len(K), len(B), ptr(db), ptr(de), ptr(dk), ptr(ok), ptr(dv), ptr(ov), int(begin_bit), int(end_bit)))
    torch.cuda.synchronize(ctx.device)
    return ok.cpu().numpy().view(np.uint64)[:len(K)].copy(), ov.cpu().numpy().view(np.uint32)[:len(V)].copy()


def create_rc(ctx: Context, batch: ReadBatch):
    """CreateRC for every read of the batch: returns a device tensor laid out like batch.seq."""
    rc = torch.zeros_like(batch.seq)
    ctx.check(ctx.lib.lra_create_rc_batch(ctx.h, batch.n, ptr(batch.seq), ptr(batch.off), ptr(rc)))
    return rc
