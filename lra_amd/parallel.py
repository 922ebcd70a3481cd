"""Multi-GPU plumbing: reads are hash-partitioned by ordinal (no data-path collective); the only exchange step is gathering the per-rank record
buffers (lra_map_pack: per-alignment fields, counters, CIGAR runs -- variable length) to rank 0, which turns them into text and emits the reads
in input order (the reference's ordered output, lra.cpp:145-166): all_gather of the per-rank sizes, then one padded
gather.  Backend "nccl" is RCCL over xGMI on the GPU box; the same code runs on "gloo" in the CPU tests."""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

_M = (1 << 64) - 1


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M
    return x ^ (x >> 31)


def shard_of(ordinal, world_size):
    """Rank that owns read `ordinal`: a hash of the ordinal (static, deterministic, independent of read length or file order)."""
    return _splitmix64(int(ordinal)) % world_size


def shard_ordinals(n_total, rank, world_size):
    return [i for i in range(n_total) if shard_of(i, world_size) == rank]


def gather_records(local: torch.Tensor, dst=0):
    """Gather a 1-D tensor of per-rank variable length to `dst`.  Returns the list of per-rank tensors on
    dst (None elsewhere).  Works for world_size 1 without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [local]
    ws, rank = dist.get_world_size(), dist.get_rank()
    size = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(size) for _ in range(ws)]
    dist.all_gather(sizes, size)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    padded = torch.zeros(mx, dtype=local.dtype, device=local.device)
    padded[:local.numel()] = local
    if rank == dst:
        bufs = [torch.empty(mx, dtype=local.dtype, device=local.device) for _ in range(ws)]
        dist.gather(padded, bufs, dst=dst)
        return [b[:n] for b, n in zip(bufs, sizes)]
    dist.gather(padded, None, dst=dst)
    return None


def records_from_packed(lib, copts, packed: np.ndarray, names, reads, chrom_names, quals=None, passthrough=None, n_threads=0):
    """One rank's record buffer (host bytes, the layout of lra_map_pack) -> one text record per read of that rank, in the rank's read order.
    Host only: lra_map_unpack_host + lra_map_records_host."""
    n = len(names)
    buf = np.ascontiguousarray(packed, dtype=np.uint8)
    snap = C.c_void_p()
    rc = lib.lra_map_unpack_host(C.c_void_p(buf.ctypes.data), C.c_uint64(buf.nbytes), C.byref(snap))
    if rc != 0:
        raise RuntimeError("lra_map_unpack_host failed (%d)" % rc)
    nm = [x if isinstance(x, bytes) else str(x).encode() for x in names]; rd = [bytes(x) for x in reads]
    a_names = (C.c_char_p * n)(*nm); a_reads = (C.c_char_p * n)(*rd)
    a_quals = (C.c_char_p * n)(*[None if q is None else bytes(q) for q in quals]) if quals is not None else None
    a_len = (C.c_int32 * n)(*[len(x) for x in rd])
    cn = [x if isinstance(x, bytes) else str(x).encode() for x in chrom_names]
    a_chr = (C.c_char_p * len(cn))(*cn)
    text = C.c_char_p(); ln = C.c_uint64(0); roff = C.POINTER(C.c_uint64)()
    rc = lib.lra_map_records_host(snap, C.byref(copts), a_names, a_reads, a_quals, a_len, a_chr, passthrough, int(n_threads), C.byref(text), C.byref(ln), C.byref(roff))
    if rc != 0:
        lib.lra_map_host_free(snap)
        raise RuntimeError("lra_map_records_host failed (%d)" % rc)
    raw = C.string_at(text, ln.value)
    out = [raw[roff[i]:roff[i + 1]] for i in range(n)]
    lib.lra_map_host_free(snap)
    return out


def merge_by_ordinal(per_rank_texts, per_rank_ordinals, n_total):
    """Rank 0's ordered emission: per_rank_texts[r][j] is the record of read per_rank_ordinals[r][j]; -> the records in input order."""
    out = [None] * n_total
    for texts, ords in zip(per_rank_texts, per_rank_ordinals):
        assert len(texts) == len(ords)
        for t, o in zip(texts, ords):
            out[o] = t
    assert all(x is not None for x in out)
    return out
