"""N>1 path on CPU: world_size-2 gloo processes own hash-partitioned read ordinals, pack their record buffers (the layout of lra_map_pack,
restated here), gather them to rank 0, which unpacks every rank's buffer (lra_map_unpack_host), formats the records (lra_map_records_host) and
emits them in input order: the merged SAM must equal the SAM of the same reads handled by one rank."""
import os
import socket
import struct

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

PACK_MAGIC = 0x4c52414d41503031
N_READS = 23
CHROM_POS = [0, 50_000, 120_000]
CHROM_NAMES = [b"chrA", b"chrB"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _read(i):
    """Fabricated result of read i (deterministic): bases, and per primary chain p a list of SegAlignments."""
    rng = np.random.default_rng(100 + i)
    L = int(rng.integers(300, 900))
    seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)].tobytes()
    kind = i % 6                       # 0, 1: one alignment; 2: unaligned; 3: two segments (supplementary); 4: a secondary chain too; 5: flagged
    def aln(q0, q1, chrom, strand, supp, n0):
        t0 = int(rng.integers(1000, 40_000))
        nm = q1 - q0 - 7
        runs = [((q1 - q0 - 7) << 4) | 0, (4 << 4) | 1, (3 << 4) | 0, (5 << 4) | 3]            # = X = D
        counts = [nm + 3 - 3, 4, 0, 1, 5, 0, 1, 0, 0, 0, 0, 0, q0, L - q1, q0, q1, t0, t0 + (q1 - q0) + 5]
        return dict(strand=strand, supp=supp, sec=0, n0=n0, n1=n0 // 2, chrom=chrom, fval=float(10 * n0), counts=counts, runs=runs, ends=[q0, q1], nblocks=3)
    jobs = [[], []]
    if kind in (0, 1):
        jobs[0] = [aln(5, L - 3, i % 2, kind, 0, 40 + i)]
    elif kind == 3:
        jobs[0] = [aln(L // 2, L - 2, 1, 0, 0, 30), aln(4, L // 2 - 10, 0, 1, 1, 30)]
    elif kind == 4:
        jobs[0] = [aln(0, L, 0, 0, 0, 50)]
        jobs[1] = [aln(10, L - 10, 1, 1, 0, 12)]
    elif kind == 5:
        jobs[0] = [aln(0, L, 0, 0, 0, 50)]
    return dict(seq=seq, jobs=jobs, reached=[1 if jobs[0] else 0, 1 if jobs[1] else 0], status=4 if kind == 5 else 0)


def pack(ordinals):
    """The record buffer of the reads `ordinals` (include/lra_hip.h lra_map_pack): header, then the arrays, each padded to 8 bytes."""
    reads = [_read(i) for i in ordinals]
    nR = len(reads); na = 2; nJ = nR * na
    alns = [a for r in reads for j in r["jobs"] for a in j]
    nA = len(alns)
    jo = np.zeros(nJ + 1, np.uint64)
    jo[1:] = np.cumsum([len(j) for r in reads for j in r["jobs"]])
    runs = np.array([x for a in alns for x in a["runs"]], np.uint32)
    roff = np.zeros(nA + 1, np.uint64); roff[1:] = np.cumsum([len(a["runs"]) for a in alns])
    boff = np.zeros(nA + 1, np.uint64); boff[1:] = np.cumsum([a["nblocks"] for a in alns])
    def i32(k): return np.array([a[k] for a in alns], np.int32)
    parts = [np.array([PACK_MAGIC, nR, na, nJ, nA, 0, len(runs), len(CHROM_POS) - 1, 1, 1, 0, 0, 0, 0, 0, 0], np.int64),
             np.array(CHROM_POS, np.uint64), np.array([x for r in reads for x in r["reached"]], np.uint8), np.array([r["status"] for r in reads], np.uint32),
             jo, i32("strand"), i32("supp"), i32("sec"), i32("n0"), i32("n1"), i32("chrom"), np.array([a["fval"] for a in alns], np.float32),
             np.array([a["counts"] for a in alns], np.int32).reshape(-1), boff, np.array([a["ends"] for a in alns], np.uint32).reshape(-1), roff, runs]
    out = b""
    for p in parts:
        b = p.tobytes()
        out += b + b"\0" * ((-len(b)) % 8)
    return np.frombuffer(out, np.uint8).copy(), reads


def _opts():
    from lra_amd import mapread
    from lra_amd._lib import load_library
    lib = load_library()
    m = mapread.MapOpts()
    lib.lra_map_opts_preset_ont(__import__("ctypes").byref(m))
    return lib, m


def _texts(lib, m, ordinals, packed, reads):
    from lra_amd import parallel
    return parallel.records_from_packed(lib, m, packed, [b"read%d" % i for i in ordinals], [r["seq"] for r in reads], CHROM_NAMES, n_threads=2)


def _worker(rank, ws, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from lra_amd import parallel
    mine = parallel.shard_ordinals(N_READS, rank, ws)
    packed, _ = pack(mine)
    got = parallel.gather_records(torch.from_numpy(packed), dst=0)
    if rank == 0:
        lib, m = _opts()
        per_rank, ords = [], []
        for r in range(ws):
            o = parallel.shard_ordinals(N_READS, r, ws)
            per_rank.append(_texts(lib, m, o, got[r].numpy(), [_read(i) for i in o]))
            ords.append(o)
        q.put(parallel.merge_by_ordinal(per_rank, ords, N_READS))
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_emit_the_same_sam_as_one():
    from lra_amd import parallel
    ws, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, ws, port, q)) for r in range(ws)]
    for p in ps:
        p.start()
    merged = q.get(timeout=180)
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    lib, m = _opts()
    allo = list(range(N_READS))
    packed, reads = pack(allo)
    single = _texts(lib, m, allo, packed, reads)
    assert merged == single
    # the partition covers every ordinal once, is not the identity on length-sorted input, and both ranks got work
    o0, o1 = parallel.shard_ordinals(N_READS, 0, 2), parallel.shard_ordinals(N_READS, 1, 2)
    assert sorted(o0 + o1) == allo and min(len(o0), len(o1)) >= 5 and o0 != list(range(0, N_READS, 2))
    # what the records look like: flags, the supplementary read's two lines, the flagged read's empty record
    for i, t in enumerate(single):
        lines = [l for l in t.decode().split("\n") if l]
        kind = i % 6
        if kind == 5:
            assert t == b""
            continue
        f = lines[0].split("\t")
        assert f[0] == "read%d" % i
        if kind == 2:
            assert int(f[1]) & 4 and len(lines) == 1
        elif kind == 3:
            assert len(lines) == 2 and sum(int(l.split("\t")[1]) & 2048 > 0 for l in lines) == 1
        else:
            assert len(lines) == 1 and f[2] in ("chrA", "chrB") and "4X3=5D" in f[5]


def test_gather_single_process():
    from lra_amd import parallel
    t = torch.arange(5)
    assert parallel.gather_records(t)[0] is t
