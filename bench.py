#!/usr/bin/env python
"""bench.py -- throughput of the MI355X hot path on BASELINE.json configs[2]: 30 kb ONT-like reads (10 % error) against a GRCh38-sized
reference, -ONT preset.

Reference (generated on the box, seeded: lra_amd/synth_genome.py): 24 chromosomes with the GRCh38 length table (3.09 Gb), ~50 % interspersed
repeats + satellite arrays + N gaps; global index = StoreIndex on the device (lra_ctx_build_global_index, -ONT index preset 17/10/150/15/1),
local index = LocalIndex::IndexSeq on the device; both resident in HBM with the genome (one replica per GPU).
One "step" = one batch of reads already in HBM through MapRead_lowacc (lra_map_reads_lowacc_batch: a1-a5, a7-a11, a13 incl. a12, a14, a16), its
record buffer packed (lra_map_pack) and gathered to rank 0 (RCCL; the single exchange step), and on rank 0 the host tail (MAPQ, ordering, SAM
text: lra_map_records_host) of every rank's reads in input order -- the host tail of batch i runs beside the device side of batch i + 1.

Contract: python bench.py --gpus N --steps K --warmup W  -> rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import contextlib
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def host_cpus():
    """(CPUs this process may use at once, hardware threads): a container's CPU bandwidth quota (cgroup cpu.max / cfs_quota_us) caps the first -- threads beyond it
    only use the period's budget up sooner, after which the kernel throttles the whole process (the thread that drives the device included)."""
    hw = os.cpu_count() or 1
    try:
        q, p_ = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, min(hw, int(q) // int(p_))), hw
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p_ > 0:
                return max(1, min(hw, q // p_)), hw
        except Exception:
            pass
    return hw, hw


def cpu_baseline(mapper, reads_h, off_h, args, last_res=None, opts=None, sample=None, res_mapper=None):
    """The oracle (CPU restatement) on the host's cores, on a bounded sample of the same batch: MapRead_lowacc read by read
    (oracle_map_reads_lowacc_mt, oracle/pipeline.cpp: the same stages in the same order as the GPU step; tests/test_mapread.py compares its
    alignments with the GPU's bit for bit) on all hardware threads."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    import oracle_pipeline as OP
    from lra_amd import index as I
    O.lib()
    ctx = mapper.ctx
    t0 = time.time()
    key, pos = I.global_index(ctx)
    g = ctx.to_host(ctx.lib.lra_ctx_genome_ptr(ctx.h), mapper.G, np.uint8).tobytes() + b"\0" * 64
    g_index = mapper.fetch_local_index()
    fetch_s = time.time() - t0
    if opts is None:                                                        # (tools/bench_presets.py hands over another preset's options)
        opts = dict(globalK=args.k, globalW=args.w, globalMaxFreq=args.max_freq, refineBand=args.refine_band, localIndexWindow=args.local_window)
    n_threads, hw_threads = host_cpus()                                       # the CPUs the box gives this process (its cgroup quota), not the threads it lists
    # a bounded sample: about 10-30 s of wall time
    S = int(min(len(off_h) - 1, 4096, max(512, 8 * n_threads)))
    if os.environ.get("LRA_BENCH_CPU_SAMPLE"):                               # (a wider parity sweep: the whole batch takes ~2 minutes on 256 threads)
        S = int(min(len(off_h) - 1, int(os.environ["LRA_BENCH_CPU_SAMPLE"])))
    if sample:
        S = int(min(len(off_h) - 1, int(sample)))
    res = OP.map_reads_lowacc_mt(reads_h, off_h, 0, S, g, key, pos, g_index, opts, mapper.chrom_pos, n_threads=n_threads)
    # The same reads' alignments as the last GPU step left them (refined blocks + the 18 counters of every SegAlignment), folded the way the oracle folds its own
    # (oracle/pipeline.cpp: oracle_map_reads_lowacc_mt): the sample is also a parity check at the benchmark's scale.
    parity = None
    if last_res is not None:
        out = (res_mapper or mapper).fetch(last_res)                        # (two-stage batches: the result lives on the back context)
        na = int(last_res.num_aln)
        P = np.uint64(1099511628211)
        total = np.uint64(0)
        with np.errstate(over="ignore"):
            reached = out.get("job_reached")
            for r in range(S):
                # the loop over the primary chains p as Map_lowacc.h:259-267, :486-491 run it (and lra_map_records_host reads the result): a chain that does not reach its
                # SegAlignmentGroup ends the loop -- at p = 0 the read is unaligned whatever the later chains hold
                a0 = int(out["job_aln_off"][r * na]); a1 = a0
                for p_ in range(na):
                    j = r * na + p_
                    if reached is not None and len(reached) and not reached[j]:
                        break
                    a1 = int(out["job_aln_off"][j + 1])
                if out["read_status"][r] or a1 == a0 or int(out["job_aln_off"][r * na + 1]) == a0:                       # (p = 0 without a SegAlignment: unaligned, :578-581)
                    continue
                parts = []
                for a in range(a0, a1):
                    b = out["blocks"][int(out["block_off"][a]):int(out["block_off"][a + 1])]
                    parts.append(b.reshape(-1).astype(np.uint32).astype(np.uint64)); parts.append(out["counts"][a].astype(np.int64).astype(np.uint64))
                x = np.concatenate(parts)
                pw = np.ones(len(x), np.uint64)
                if len(x) > 1:
                    pw[1:] = P
                    pw = np.multiply.accumulate(pw)                        # P^0 .. P^(n-1), wrapping
                h = np.sum(x * pw[::-1], dtype=np.uint64)                   # sum x_i P^(n-1-i)
                total = total + h * np.uint64(r + 1)
        parity = bool(int(total) == int(res["checksum"]))
    return {"value": res["bases"] / res["seconds"] / 1e9, "unit": "Gbp/s", "cores": n_threads, "kind": "port", "sample_equals_gpu": parity,
            "sample": "first %d reads (%d bp, %d alignments) of the same batch through the oracle's MapRead_lowacc (a1-a5, a7-a11, a13 incl. a12, a14, a16: the stages of "
                      "the GPU step, oracle/pipeline.cpp) on %d host threads = the CPUs this box gives the process (cgroup CPU quota; %d hardware threads listed) in %.1f s (reference data fetched from the device in %.1f s, not timed)"
                      % (res["n_reads"], res["bases"], res["n_alignments"], n_threads, hw_threads, res["seconds"], fetch_s)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)        # (the pipeline of three contexts takes ~3 steps to fill: fewer warm-up steps time its ramp)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--genome-scale", type=float, default=float(os.environ.get("LRA_BENCH_GENOME_SCALE", 1.0)), help="1.0 = GRCh38-sized (3.09 Gb)")
    ap.add_argument("--reads", type=int, default=int(os.environ.get("LRA_BENCH_READS", 28672)),
                    help="reads per GPU per step (a batch: the pipeline's granularity, not the job's size).  28672: the largest batch beside which the seed stage of the front half "
                         "after next still has the memory for a context of its own at the .gli parameters (282 GB in use; 32768 reads: 286 GB without it, 4.6 %% slower per read)")
    ap.add_argument("--read-len", type=int, default=30000)
    ap.add_argument("--err", type=float, default=0.10)
    ap.add_argument("--sv-frac", type=float, default=0.05, help="fraction of reads carrying one planted structural variant")
    ap.add_argument("--k", type=int, default=17)          # -ONT: globalK 17, globalW 10 (lra.cpp:386-431)
    ap.add_argument("--w", type=int, default=10)
    ap.add_argument("--max-freq", type=int, default=150)
    ap.add_argument("--refine-band", type=int, default=7)
    ap.add_argument("--local-window", type=int, default=int(os.environ.get("LRA_BENCH_LOCAL_WINDOW", 2048)),
                    help="glIndex.localIndexWindow: 2048 = the .gli file `lra index` writes (LocalIndex(0), MMIndex.h:110-127), 256 = `lra align` without a .gli file (opts.localIndexWindow)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-records", action="store_true", help="leave the host tail (SAM text) out of the step")
    ap.add_argument("--satellite-frac", type=float, default=0.03, help="fraction of every chromosome in its satellite array")
    ap.add_argument("--genome-seed", type=int, default=3, help="seed of the synthetic reference (3 = the one every recorded figure is on)")
    ap.add_argument("--defer-seed", type=int, default=int(os.environ.get("LRA_BENCH_DEFER_SEED", 0)),
                    help="lra_map_opts.defer_seed_matches: reads with more tier-1 matches are handed back by the batch they arrive in, pooled, and mapped as batches of "
                         "their own inside the timed region (cost-ordered batching: every read is mapped exactly once per step either way); 0 = off")
    ap.add_argument("--heavy-lane", type=int, default=int(os.environ.get("LRA_BENCH_HEAVY_LANE", 1)),
                    help="1 = the batches of handed-back reads run on a context of their own (low-priority streams) beside the next steps; 0 = on lane 0, between steps")
    ap.add_argument("--heavy-pool", type=int, default=int(os.environ.get("LRA_BENCH_HEAVY_POOL", 4096)), help="handed-back reads per batch of their own")
    ap.add_argument("--two-stage", type=int, default=int(os.environ.get("LRA_BENCH_TWO_STAGE", 1)),
                    help="1 = two-stage batches (lra_map_reads_lowacc_front / _back): the front half of step i + 1 on one host thread beside the back half of step i on another")
    ap.add_argument("--front-priority", default=os.environ.get("LRA_BENCH_FRONT_PRIORITY", "low"), choices=["low", "normal"],
                    help="with --two-stage: the stream priority of the front halves (the back halves: LRA_BACK_PRIORITY, default the highest)")
    ap.add_argument("--seed-ahead", type=int, default=int(os.environ.get("LRA_BENCH_SEED_AHEAD", -1)),
                    help="1 = a step's seed stage (a1-a4) runs beside the step before it, on a side context (lra_seed_prefetch / lra_ctx_adopt_seed); 0 = every step seeds itself; "
                         "2 = also with two-stage batches (the seed stage of the front half after next on a third context: ~30 GB more); -1 (default) = 2 when, after the first "
                         "warm-up step, at least LRA_BENCH_SEED_AHEAD_FREE_GB (40) of HBM are free, else 1")
    ap.add_argument("--seed-ahead-delay-ms", type=float, default=float(os.environ.get("LRA_BENCH_SEED_AHEAD_DELAY_MS", 150)),
                    help="how long into a step the seeding of the next one starts (measured on one box, 10 steps each: 0 ms 790 ms per step, 75: 757, 150: 747 / 756 / 752, "
                         "225: 752, 300: 775 / 765 / 761 / 765, 450: 819)")
    ap.add_argument("--lane-priority", type=int, default=int(os.environ.get("LRA_BENCH_LANE_PRIORITY", 1)),
                    help="with --lanes > 1: 1 = lane 0 on a high-priority stream, the others below it (they fill what it leaves idle); 0 = all lanes alike")
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("LRA_BENCH_LANES", 1)),
                    help="the batch is cut into this many sub-batches, each driven by its own context and HIP stream from its own host thread (ONE shared replica "
                         "of the reference), so that the serial tails of one sub-batch's kernels overlap the other's work")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # LRA_BENCH_ONE_DEVICE=1 (a dry run of the multi-rank control flow on a one-GPU box): every rank on device 0, gloo instead of RCCL (which refuses two ranks per GPU)
    one_dev = world > 1 and os.environ.get("LRA_BENCH_ONE_DEVICE") == "1"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_dev:
            local_rank = 0
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev_index = local_rank if world > 1 else 0
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    from lra_amd.context import Context
    from lra_amd import seed, parallel, mapread, synth_genome as sg
    mopts = mapread.LowAccOptions(globalK=args.k, globalW=args.w, globalMaxFreq=args.max_freq, refineBand=args.refine_band, localIndexWindow=args.local_window)   # -ONT

    # ---- reference side, once per process: genome, StoreIndex, LocalIndex (all on the device)
    t0 = time.time()
    genome, chrom_pos, chrom_names = sg.make_grch38_like(dev, scale=args.genome_scale, seed=args.genome_seed, satellite_frac=args.satellite_frac)
    torch.cuda.synchronize()
    gen_s = time.time() - t0
    ctx = Context(dev_index)
    t0 = time.time()
    mapper = mapread.LowAccMapper(ctx, genome, None, None, chrom_names, chrom_pos, mopts, index_params=(args.k, args.w, args.max_freq, 15, 1), staged=False)
    torch.cuda.synchronize()
    index_s = time.time() - t0
    G = mapper.G
    # ---- this rank's reads (hash partition of the job's ordinals; weak scaling: args.reads per GPU), cut into args.lanes sub-batches
    t0 = time.time()
    sim = sg.simulate_reads_sv(genome, chrom_pos, args.reads, args.read_len, args.read_len / 10, args.err, (30, 35, 35), 1000 + rank, sv_frac=args.sv_frac)
    off_h = sim["off"].cpu().numpy()
    reads_h = sim["seq"][:int(off_h[-1])].cpu().numpy()
    n_sv = int((sim["sv"] > 0).sum())
    total_bases = int(off_h[-1])
    del genome
    rb = reads_h.tobytes()
    # the record text on half of the host's hardware threads (per rank): the thread that drives the device (sizing round trips between the stages) needs a core of its own
    # the record text on the host threads the library allows itself (lra_host_thread_budget: the hardware threads, capped by the container's CPU quota less two -- a
    # burst of more runnable threads than the quota gets the whole process throttled for the rest of the scheduler period, the thread that drives the device included)
    from lra_amd._lib import load_library
    n_threads_rec = max(2, load_library().lra_host_thread_budget() // world)
    # A step = every sub-batch once.  With one lane that is one call on the whole batch.  With several, the sub-batches of all timed steps form one work list and
    # every lane (a context of its own: HIP streams, work buffers; the reference data shared) takes the next item when it is free -- so a lane on a lower-priority
    # stream, which only gets what the lane above it leaves idle, simply takes fewer of them.
    lanes, subs = [], []
    prio_lo, prio_hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    for li in range(args.lanes):
        r0, r1 = args.reads * li // args.lanes, args.reads * (li + 1) // args.lanes
        stream = None
        if li == 0:
            lctx, lmap = ctx, mapper
        else:
            lctx = Context(dev_index)
            lmap = mapread.LowAccMapper.sharing(lctx, mapper)
        if args.lanes > 1:
            pr = (prio_hi if li == 0 else min(prio_hi + li, prio_lo)) if args.lane_priority else 0
            stream = torch.cuda.Stream(device=dev_index, priority=pr)
            lctx.bind_stream(stream)
        b0, b1 = int(off_h[r0]), int(off_h[r1])
        lseq = torch.cat([sim["seq"][b0:b1], torch.zeros(64, dtype=torch.uint8, device=dev)])
        loff = (sim["off"][r0:r1 + 1] - b0).contiguous()
        rbatch = seed.read_batch_from_device(lctx, lseq, loff)
        names = [b"read%d" % i for i in range(r0, r1)]
        reads_b = [rb[int(off_h[i]):int(off_h[i + 1])] for i in range(r0, r1)]
        subs.append(dict(rbatch=rbatch, rargs=lmap.record_args(names, reads_b), bases=b1 - b0, names=names, reads=reads_b))
        lanes.append(dict(ctx=lctx, mapper=lmap, stream=stream, packed=None, n_items=0))
    del sim
    # Reads handed back by their batch (opts.defer_seed_matches) are pooled and mapped as batches of their own, by a second view of lane 0's mapper with the field off
    defer_T = args.defer_seed if args.lanes == 1 else 0
    heavy = dict(pool=[], n=0, batches=0, reads=0)
    if defer_T:
        import copy
        mapper.copts.defer_seed_matches = defer_T
        if args.heavy_lane:
            hctx = Context(dev_index)
            hmap = mapread.LowAccMapper.sharing(hctx, mapper)
            hstream = torch.cuda.Stream(device=dev_index, priority=prio_lo)
            hctx.bind_stream(hstream)
        else:
            hctx, hstream = ctx, None
            hmap = copy.copy(mapper)
        hmap.copts = type(mapper.copts).from_buffer_copy(mapper.copts)
        hmap.copts.defer_seed_matches = 0
        hmap.stats = {}
        heavy["lane"] = dict(ctx=hctx, mapper=hmap, stream=hstream, packed=None, n_items=0)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    sim_s = time.time() - t0

    worker = [None]
    text_bytes = [0]
    err = []

    def host_tail(items):
        try:
            for lane, sub, snap in items:
                text_bytes[0] += lane["mapper"].records_host(snap, sub["rargs"], n_threads=n_threads_rec, as_list=False)
        except BaseException as e:
            err.append(e)

    two_stage = bool(args.two_stage) and args.lanes == 1 and not defer_T
    sa_auto = args.seed_ahead < 0
    def ahead_wanted(sa):
        return bool(sa) and args.lanes == 1 and not defer_T and (not two_stage or sa == 2)   # (2: also with two-stage batches)
    ahead_on = ahead_wanted(1 if sa_auto else args.seed_ahead)           # (auto: decided behind the first warm-up step, below)
    if two_stage:                                                          # the front halves are the work done ahead: below the back halves' priority
        fstream_ = torch.cuda.Stream(device=dev_index, priority=prio_lo if args.front_priority == "low" else 0)
        ctx.bind_stream(fstream_)
    ahead = {"on": ahead_on}
    def make_ahead():
        ahead["ctx"] = Context(dev_index)
        mapread.LowAccMapper.sharing(ahead["ctx"], mapper)
        ahead["stream"] = torch.cuda.Stream(device=dev_index, priority=prio_lo)
        ahead["ctx"].bind_stream(ahead["stream"])
    if ahead["on"]:
        make_ahead()

    def seed_ahead(sub):
        try:
            torch.cuda.set_device(dev_index)
            if args.seed_ahead_delay_ms > 0:
                time.sleep(args.seed_ahead_delay_ms * 1e-3)
            t0_ = time.perf_counter()
            seed.seed_prefetch(ahead["ctx"], sub["rbatch"], args.k, args.w, args.max_freq)
            ahead["ok"] = True
            if os.environ.get("LRA_BENCH_DBG"):
                sys.stderr.write("[bench] seeding ahead %.0f ms\n" % ((time.perf_counter() - t0_) * 1e3))
        except BaseException as e:
            err.append(e)

    def lane_device_side(lane, sub):
        try:
            torch.cuda.set_device(dev_index)                               # the current device is per host thread
            lc = lane["ctx"]
            def run():
                tA_ = time.perf_counter()
                if ahead["on"] and not two_stage:
                    # the seeding of the NEXT step's batch beside this step (lra_seed_prefetch on a side context, low-priority stream, a host thread of its own), and this
                    # step's own seed result -- made beside the previous step -- adopted instead of seeding: every timed step runs one alignment pass and one seeding
                    th_ = ahead.pop("thread", None)
                    if th_ is not None:
                        th_.join()
                        if ahead.pop("ok", False):
                            seed.adopt_seed(lc, ahead["ctx"])
                    ahead["thread"] = threading.Thread(target=seed_ahead, args=(sub,))
                    ahead["thread"].start()
                if two_stage:
                    # this thread runs the back halves (the front halves: front_loop, a thread of its own); the result's arrays belong to the back context
                    try:
                        res, bc = lane["mapper"].back()
                        lane["last_res"] = res
                        if not args.no_records and not err:
                            d_buf, nb = C.c_void_p(), C.c_uint64(0)
                            bc.check(bc.lib.lra_map_pack(bc.h, C.byref(res), 0, C.byref(d_buf), C.byref(nb)))
                            lane["packed"] = bc.to_tensor(d_buf.value, nb.value, torch.uint8)
                            bc.to_host(d_buf.value, 1, np.uint8)           # (a synchronous copy: the back context's own stream is through the pack and the copy)
                    finally:
                        try:
                            lane["mapper"].release()                       # (whatever happened: the front thread waits for this)
                        except BaseException:
                            pass
                    return
                res = lane["mapper"].align(sub["rbatch"])
                lane["last_res"] = res
                if args.no_records:
                    return
                d_buf, nb = C.c_void_p(), C.c_uint64(0)
                tP = time.perf_counter()
                lc.check(lc.lib.lra_map_pack(lc.h, C.byref(res), 0, C.byref(d_buf), C.byref(nb)))
                lane["packed"] = lc.to_tensor(d_buf.value, nb.value, torch.uint8)
                if os.environ.get("LRA_BENCH_DBG"):
                    torch.cuda.synchronize()
                    sys.stderr.write("[bench] align %.0f ms, pack + copy %.0f ms\n" % ((tP - tA_) * 1e3, (time.perf_counter() - tP) * 1e3))
            if lane["stream"] is not None:
                with torch.cuda.stream(lane["stream"]):
                    run()
                lane["stream"].synchronize()
            else:
                run()
        except BaseException as e:                                         # surfaced by the caller: a thread's exception would vanish otherwise
            err.append(e)

    def pool_handed_back(lane, j):
        """the reads of sub-batch j that the last call handed back (LRA_ST_DEFERRED) join the pool"""
        res = lane["last_res"]
        st_ = lane["ctx"].to_host(res.d_read_status, int(res.n_reads), np.uint32)
        idx = np.nonzero(st_ & 64)[0]
        if len(idx):
            heavy["pool"].append((j, idx)); heavy["n"] += len(idx)

    def heavy_sub():
        """the pooled reads as a batch: their bases gathered on the device, their names / sequences for the records"""
        seqs, lens, names, rds = [], [], [], []
        for j, idx in heavy["pool"]:
            rb_ = subs[j]["rbatch"]
            it = torch.from_numpy(idx.astype(np.int64)).to(dev)
            s0 = rb_.off[it]; ln = rb_.off[it + 1] - s0
            dst0 = torch.cumsum(ln, 0) - ln
            src = torch.arange(int(ln.sum()), device=dev) + torch.repeat_interleave(s0 - dst0, ln)
            seqs.append(rb_.seq[src]); lens.append(ln)
            names += [subs[j]["names"][i] for i in idx]; rds += [subs[j]["reads"][i] for i in idx]
        ln = torch.cat(lens)
        off = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(ln, 0)])
        seq = torch.cat(seqs + [torch.zeros(64, dtype=torch.uint8, device=dev)])
        hl = heavy["lane"]
        sub = dict(rbatch=seed.read_batch_from_device(hl["ctx"], seq, off), rargs=hl["mapper"].record_args(names, rds), bases=int(off[-1]), names=names, reads=rds)
        heavy["batches"] += 1; heavy["reads"] += len(names)
        heavy["pool"] = []; heavy["n"] = 0
        return sub

    # Every lane runs its own sequence of steps (no barrier between the lanes inside the timed region): lane i starts i / lanes of a step late, so
    # that the long serial tail of one sub-batch's sparse DP runs beside the other sub-batches' wide kernels.
    copy_stream = torch.cuda.Stream(device=dev_index)

    def tail_thread(lane, sub, held, ev):
        try:
            torch.cuda.set_device(dev_index)
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(ev)
                # Every rank turns its own reads' records into text (its shard of the output, in ordinal order) with its share of the host's cores: the
                # path has no exchange step -- reads are independent, and so are their records.  (lra_amd.parallel.gather_records / merge_by_ordinal
                # bring the record buffers of all ranks to rank 0 when one process has to write one stream: tests/test_parallel.py.)
                tA = time.perf_counter()
                # into a page-locked buffer that is kept (one per lane: a lane's tails run one after the other).  A copy into fresh pageable memory makes the runtime pin and
                # unpin 0.7 GB of host pages per step, and the unpin stalls every queue of the process for ~50 ms (a kernel trace shows the stall on whatever kernel runs then)
                nb_ = held.numel()
                if lane.get("pin") is None or lane["pin"].numel() < nb_:
                    lane["pin"] = torch.empty(int(nb_ * 1.25) + 4096, dtype=torch.uint8, pin_memory=True)
                lane["pin"][:nb_].copy_(held, non_blocking=True)
                copy_stream.synchronize()
                hb = lane["pin"][:nb_].numpy()
                tB = time.perf_counter()
                snap = C.c_void_p()
                rc = ctx.lib.lra_map_unpack_host(C.c_void_p(hb.ctypes.data), C.c_uint64(hb.nbytes), C.byref(snap))
                assert rc == 0, rc
                items = [(lane, sub, snap)]
            tC = time.perf_counter()
            host_tail(items)
            if os.environ.get("LRA_BENCH_DBG"):
                sys.stderr.write("[bench] host tail: copy %.0f ms (%.2f GB), unpack %.0f ms, records %.0f ms\n" % ((tB - tA) * 1e3, hb.nbytes / 1e9, (tC - tB) * 1e3, (time.perf_counter() - tC) * 1e3))
        except BaseException as e:
            err.append(e)

    work = {"items": [], "next": 0}
    work_lock = threading.Lock()

    def take_item():
        with work_lock:
            if work["next"] >= len(work["items"]):
                return None
            it = work["items"][work["next"]]
            work["next"] += 1
            return it

    def process_item(lane, sub, prev, tag):
        """device side + pack of one batch on `lane`, then its host tail on a thread of its own (after the previous one of this caller) -> that thread"""
        dbg = os.environ.get("LRA_BENCH_DBG")
        tA = time.perf_counter()
        lane_device_side(lane, sub)
        lane["n_items"] += 1
        if dbg:
            sys.stderr.write("[bench] %s (%d reads) device side + pack %.0f ms\n" % (tag, sub["rbatch"].n, (time.perf_counter() - tA) * 1e3))
        if err or args.no_records:
            return prev
        # The exchange step and everything behind it -- gather to rank 0, the copy to the host, unpacking, the record text -- belong to the step's
        # host tail and run beside the next step's device side (the reference interleaves its output with the next reads the same way, lra.cpp:117-158).
        # The packed buffer is context-owned (the next lra_map_pack reuses it), so the tail works on a device copy.
        with torch.cuda.stream(lane["stream"]) if lane["stream"] is not None else contextlib.nullcontext():
            held = lane["packed"].clone()
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
        if prev is not None:
            tC = time.perf_counter()
            prev.join()
            if dbg:
                sys.stderr.write("[bench] %s waited for the previous host tail %.0f ms\n" % (tag, (time.perf_counter() - tC) * 1e3))
        th = threading.Thread(target=tail_thread, args=(lane, sub, held, ev))
        th.start()
        return th

    import queue
    heavy_q = queue.Queue()

    def heavy_worker():
        """the batches of handed-back reads, one after the other on their own context, beside the steps that follow the ones they came from"""
        prev = None
        try:
            torch.cuda.set_device(dev_index)
            while True:
                sub = heavy_q.get()
                if sub is None:
                    break
                if not err:
                    prev = process_item(heavy["lane"], sub, prev, "handed-back reads")
            if prev is not None:
                prev.join()
        except BaseException as e:
            err.append(e)

    def lane_loop(li, stagger_s):
        try:
            if stagger_s:
                time.sleep(stagger_s)
            prev = None
            while True:
                it = take_item()
                if it is None or (err and not two_stage):                  # (two-stage batches: the front thread hands over a batch per item whatever happens; each is taken)
                    break
                s_, j = it
                if j < 0:                                                  # (--heavy-lane 0) a batch of handed-back reads, between two steps
                    prev = process_item(heavy["lane"], heavy_sub(), prev, "handed-back reads")
                    continue
                prev = process_item(lanes[li], subs[j], prev, "lane %d: step %d sub-batch %d" % (li, s_, j))
                if defer_T and not err:
                    pool_handed_back(lanes[li], j)
                    with work_lock:
                        last_main = work["next"] >= len(work["items"])
                    if heavy["n"] and (heavy["n"] >= args.heavy_pool or last_main):   # the pool as a batch of its own (the last step flushes what is left)
                        if args.heavy_lane:
                            hs_ = heavy_sub()
                            torch.cuda.current_stream().synchronize()      # (the gather ran on this thread's stream)
                            heavy_q.put(hs_)
                        else:
                            with work_lock:
                                work["items"].insert(work["next"], (s_, -1))
            if prev is not None:
                prev.join()
        except BaseException as e:
            err.append(e)

    def front_loop(n_steps):
        try:
            torch.cuda.set_device(dev_index)
            done = 0
            for _ in range(n_steps):
                for sub in subs:
                    handed = False
                    if not err:
                        try:
                            if ahead["on"]:                               # (--seed-ahead 2: the seed stage of the front half after this one on a third context)
                                th_ = ahead.pop("thread", None)
                                if th_ is not None:
                                    th_.join()
                                    if ahead.pop("ok", False):
                                        seed.adopt_seed(ctx, ahead["ctx"])
                                ahead["thread"] = threading.Thread(target=seed_ahead, args=(sub,))
                                ahead["thread"].start()
                            sub["_in_front"] = True                        # (set right in front of the library call: adopt_seed above raises LraError as well)
                            lanes[0]["mapper"].front(sub["rbatch"])
                            done += 1
                            continue
                        except BaseException as e:
                            from lra_amd._lib import LraError
                            # a front CALL that fails hands over an error batch (the back call for this item returns its code); an exception on this side of the
                            # call (marshalling, the seed stage's adoption) has handed nothing over: the empty batch below takes the item's turn
                            handed = isinstance(e, LraError) and sub.get("_in_front", False)
                            err.append(e)
                        finally:
                            sub["_in_front"] = False
                    # after an error the back thread still takes a batch per item: an empty one (no device work), so that the run ends and reports the error
                    if not handed:
                        lanes[0]["mapper"].front(seed.read_batch_from_device(ctx, torch.zeros(64, dtype=torch.uint8, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)))
        except BaseException as e:
            err.append(e)

    def run_steps(n_steps, stagger):
        work["items"] = [(s_, j) for s_ in range(n_steps) for j in range(len(subs))]
        work["next"] = 0
        ft = None
        if two_stage and n_steps > 0:
            ft = threading.Thread(target=front_loop, args=(n_steps,))
            ft.start()
        hw = None
        if defer_T and args.heavy_lane:
            hw = threading.Thread(target=heavy_worker)
            hw.start()
        if len(lanes) == 1:
            lane_loop(0, 0.0)
            if hw is not None:                                             # the step is over when the reads it handed back are mapped too
                heavy_q.put(None)
                hw.join()
        else:
            ths = [threading.Thread(target=lane_loop, args=(li, stagger * li / len(lanes))) for li in range(len(lanes))]
            for t in ths: t.start()
            for t in ths: t.join()
        if ft is not None:
            ft.join()
        th_ = ahead.get("thread")
        if th_ is not None:                                                # (the seeding of the batch after the last one: joined here, so that a run of n steps holds n of them)
            th_.join()
        if err:
            raise err[0]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # (the first warm-up step allocates the contexts' work buffers -- seconds -- so the lanes' stagger is taken from the LAST warm-up step alone)
    n_warm = args.warmup
    if sa_auto and two_stage and args.lanes == 1 and not defer_T and not ahead["on"] and n_warm >= 2:
        # the third stage of the pipeline -- the seed stage of the front half after next on a context of its own -- where the device has the memory for it: every work
        # buffer of the two halves exists after the first step, the seed context needs ~30 GB, and the warm-up steps that follow allocate it outside the timed region
        run_steps(1, 0.0); n_warm -= 1
        torch.cuda.synchronize()
        free_b, _ = torch.cuda.mem_get_info(dev_index)
        ok_ = free_b >= float(os.environ.get("LRA_BENCH_SEED_AHEAD_FREE_GB", 40)) * 1e9
        if world > 1:                                                      # (every rank the same pipeline)
            tf_ = torch.tensor([1 if ok_ else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(tf_, op=dist.ReduceOp.MIN)
            ok_ = bool(tf_.item())
        if ok_:
            args.seed_ahead = 2
            make_ahead(); ahead["on"] = True
    run_steps(max(n_warm - 1, 0), 0.0)
    tw = time.perf_counter()
    run_steps(min(n_warm, 1), 0.0)
    step_guess = (time.perf_counter() - tw) if n_warm > 1 else 0.0
    ahead_on = ahead["on"]
    timed = lanes + ([heavy["lane"]] if defer_T and args.heavy_lane else [])
    if ahead_on:
        ahead["ctx"].timing(True); ahead["ctx"].timing_reset()
    for l in timed:
        l["ctx"].timing(True)
        l["ctx"].timing_reset()
        l["n_items"] = 0
    heavy["batches"] = 0; heavy["reads"] = 0
    text_bytes[0] = 0
    sync()
    t0 = time.perf_counter()
    run_steps(args.steps, step_guess if args.lanes > 1 else 0.0)
    sync()
    dt = time.perf_counter() - t0
    t_dev = dt
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        tb = torch.tensor([total_bases], dtype=torch.int64, device=dev)
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        job_bases = int(tb.item())
    else:
        job_bases = total_bases
    nreads = args.reads * world

    kernels = ["sketch_emit", "sketch_serial", "sketch_compact", "sort", "sort_fallback", "index_bounds", "compare", "strand",
               "aog_lds_tiny", "aog_lds_small", "aog_lds_medium", "aog_lds_large", "aog_lane", "aog_reg", "aog_hbm", "ir_segment", "ir_band", "ir_fill", "ir_trace", "ir_gather", "clean_sort", "clean", "linear_extend", "stats", "stats_cigar", "create_rc", "local_sketch", "local_sort_filter", "local_compare",
               "rsc_tasks", "rsc_filter", "refine_space", "rs_long_sketch", "rs_long_compare", "btwn_plan", "btwn_apply", "merge_extend", "between_anchors", "local_refine", "sdp_inner_points", "sdp_inner_sort", "sdp_inner_build_count", "sdp_inner_build", "sdp_inner_process", "sdp_inner_trace", "chain_split", "sdp_points", "sdp_sort", "sdp_sort_fallback", "sdp_build_count", "sdp_build", "sdp_process", "sdp_process_wg", "sdp_inner_process_wg", "sdp_trace"]
    ktimes = {}
    for k in kernels:
        tt_ = [l["ctx"].timing_get(k) for l in timed] + ([ahead["ctx"].timing_get(k)] if ahead_on else [])
        ktimes[k] = (sum(x[0] for x in tt_), sum(x[1] for x in tt_))
    stats = {}
    for l in timed:
        l["ctx"].timing(False)
        for k, v in l["mapper"].stats.items():
            stats[k] = stats.get(k, 0) + v if isinstance(v, (int, float)) else v
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        gbps = job_bases * args.steps / dt / 1e9
        # (the workgroup ProcessPoint launch of the large reads runs on a side stream, from the end of their build: it is part of the sdp_process stage, not a family of its
        # own -- the stage's launch duration is the longer of the two ranges, the main stream's (wave-per-read launch + the wait for the side stream) and the side stream's)
        dom = max((k for k in ktimes if not k.endswith("_wg")), key=lambda k: ktimes[k][0])
        dom_ms, dom_n = ktimes[dom]
        avg_ms = dom_ms / max(dom_n, 1)
        if dom == "sdp_process" and ktimes.get("sdp_process_wg", (0, 0))[1]:
            avg_ms = max(avg_ms, ktimes["sdp_process_wg"][0] / ktimes["sdp_process_wg"][1])
        launches_per_step = max(dom_n, 1) / args.steps
        L = total_bases
        # ALGORITHMIC bytes per step of every kernel family, SURVEY.md section 8(d): 2 L (read + RC) + 12 n_q (minimizers out) + n_q (12 + 64) (sorted
        # minimizers in + one index line per query) + 16 n_m (match pairs) + 24 per anchor per sparse-DP pass (16 in, 8 out) + 2 (q_span + t_span) + 1
        # per DP cell + 12 per block out.  DESIGN.md section 3 states which kernel owns which term.
        n_q, n_m = stats.get("n_mm", 0), stats.get("n_match", 0)
        sdp_anchors = stats.get("n_sdp_anchors", 0) + stats.get("n_sdp2_anchors", 0)
        alg8d = {
            "sketch_emit": L + 12 * n_q, "sketch_compact": 2 * 12 * n_q, "sort": 2 * 12 * n_q, "index_bounds": n_q * (12 + 64), "compare": 12 * n_q + 16 * n_m,
            "strand": 16 * n_m + 2 * args.k * n_m, "clean": 16 * n_m, "clean_sort": 2 * 16 * n_m,
            "sdp_process": 24 * sdp_anchors, "sdp_build": 16 * sdp_anchors, "sdp_build_count": 16 * sdp_anchors, "sdp_sort": 2 * 16 * sdp_anchors, "sdp_trace": 8 * sdp_anchors,
            "ir_fill": 1 * stats.get("n_cells", 0) + 2 * stats.get("n_rows", 0), "ir_band": 12 * stats.get("n_blocks", 0), "ir_trace": 1 * stats.get("n_rows", 0) + 12 * stats.get("n_blocks", 0),
            "stats": 2 * L + 12 * stats.get("n_blocks", 0), "local_sketch": 2 * L + 4 * stats.get("n_local_tuples", 0), "local_sort_filter": 2 * 4 * stats.get("n_local_tuples", 0),
            "local_compare": 4 * stats.get("n_local_task_words", 0) + 8 * stats.get("n_local_pairs", 0),
        }
        # what the implementation's own data structures occupy (the footprint the kernel must at least touch once)
        sdp_entries = stats.get("n_sdp_entries", 0) + stats.get("n_sdp2_entries", 0)
        sdp_points = stats.get("n_sdp_points", 0) + 2 * stats.get("n_sdp2_anchors", 0)
        footprint = {"sdp_process": 44 * sdp_entries + 269 * sdp_points, "sdp_build": 20 * sdp_entries + 269 * sdp_points,
                     "ir_fill": 1 * stats.get("n_cells", 0) + 17 * stats.get("n_rows", 0)}
        alg = alg8d.get(dom, 0) / launches_per_step
        achieved = alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        fp = footprint.get(dom, alg8d.get(dom, 0)) / launches_per_step
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))["kernels"].get(dom)
            if pmc and abs(pmc["reads_per_launch"] - args.reads / launches_per_step) < 1 and "genome_scale" in pmc and abs(pmc["genome_scale"] - args.genome_scale) < 1e-9 and args.lanes == 1:
                traffic = pmc["fetch_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        step_alg = 2 * L + 12 * n_q + n_q * 76 + 16 * n_m + 24 * sdp_anchors + 1 * stats.get("n_cells", 0) + 12 * stats.get("n_blocks", 0)
        out = {
            "metric": "aligned Gbp/s, 30 kb ONT vs GRCh38 (MapRead_lowacc end to end incl. SAM text), per job", "value": gbps, "unit": "Gbp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "reads_per_s": nreads * args.steps / dt,
            "config": {"workload": "BASELINE configs[2]: synthetic GRCh38-like reference (24 chromosomes, GRCh38 length table x %g = %.3f Gb, ~50%% interspersed repeats + "
                                   "satellite arrays + N gaps; seeded generator lra_amd/synth_genome.py) with a StoreIndex-faithful global index built on the device "
                                   "(K=%d W=%d maxFreq=%d winsize=15: %d entries = 1 per %.1f bp) + %d reads/GPU of N(%d, 10%%) bp, %g%% error 30:35:35, half reverse "
                                   "strand, %.0f%% with one planted SV (deletion / insertion / inversion / tandem duplication / translocation, 50 bp-10 kb)"
                                   % (args.genome_scale, G / 1e9, args.k, args.w, args.max_freq, mapper.index_stats["n_index"], G / max(mapper.index_stats["n_index"], 1),
                                      args.reads, args.read_len, args.err * 100, args.sv_frac * 100),
                       "preset": "-ONT (k=%d w=%d maxFreq=%d refineBand=%d match/mismatch/indel=4/-1/-2); local index k=10 w=5 maxFreq=15, windows of %d bases (%s)"
                                 % (args.k, args.w, args.max_freq, args.refine_band, args.local_window,
                                    "the .gli file `lra index -ONT` writes: LocalIndex(0), MMIndex.h:110-127 -- what glIndex.Read hands `lra align`" if args.local_window == 2048 else
                                    "`lra align` without a .gli file: opts.localIndexWindow" if args.local_window == 256 else "--local-window"),
                       "stages": "MapRead_lowacc chained on the reads, every stage on what the previous one produced on the device: a1-a5, a7, a8 (SDP#A), a9, a10, a11, "
                                 "a9 (MergeChain), a7 (second LinearExtend + Trim), a8 (second SDP + filters), a13 (incl. a12), a14, a16; then lra_map_pack, the gather of the "
                                 "record buffers to rank 0 and the host tail a16-a17 (SetFromSegAlignment, AlignmentsOrder, SimpleMapQV, SAM text) of batch i beside the "
                                 "device side of batch i + 1%s; with --seed-ahead the seed stage a1-a4 of batch i + 1 also runs beside batch i (lra_seed_prefetch).  Not in the step: RefineBreakpoint (a15, built; off by default in lra)" % (" -- SKIPPED (--no-records)" if args.no_records else ""),
                       "parallelism": "reads hash-partitioned by ordinal, 1 process/GPU, genome + both indexes replicated per GPU; %d sub-batch(es) per process; no data-path collective: every rank formats the records of its own reads (its shard of the output) beside its next step; RCCL only for the barriers / the final reductions of the timing" % args.lanes,
                       "per_step": {k: int(v) for k, v in stats.items() if not k.startswith("_") and isinstance(v, (int, float))},
                       "reads_with_sv": n_sv},
            "kernel_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in ktimes.items() if v[1]},
            "kernel_ms_note": ("HIP-event time of each kernel family on its own stream; with two-stage batches / the seed stage ahead two contexts run side by side, so the "
                               "families stretch each other and sum to more than the step") if (two_stage or ahead_on) else "HIP-event time of each kernel family on its stream",
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": traffic, "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": alg,
                         "footprint_bytes_per_launch": fp, "frac_footprint": (fp / (avg_ms * 1e-3) / 1e9 / 8000.0) if avg_ms > 0 else 0.0,
                         "step_algorithmic_bytes": step_alg, "step_frac": step_alg / (ms_step * 1e-3) / 1e9 / 8000.0},
            "sam_text_gb_per_step": round(text_bytes[0] / max(args.steps, 1) / 1e9, 3),
            "lane_items": [l["n_items"] for l in lanes],
            "two_stage": {"on": bool(two_stage), "what": "lra_map_reads_lowacc_front / _back: the front half (a1 .. the second LinearExtend) of step i + 1 on the context (low-priority stream, its own host thread) beside the back half (second sparse DP .. statistics) of step i on the companion context; a queue of one handed-over batch between them (two sets of handover buffers), so the back context goes from one batch straight to the next"},
            "seed_ahead": {"on": bool(ahead_on), "delay_ms": args.seed_ahead_delay_ms,
                           "what": "a1-a4 of step i + 1 on a side context (low-priority stream, own host thread) beside step i; every timed step holds one alignment pass and one seeding"},
            "handed_back": {"defer_seed_matches": defer_T, "pool": args.heavy_pool, "reads_per_step": round(heavy["reads"] / max(args.steps, 1), 1), "batches": heavy["batches"]},
            "device_side_ms_per_step": round(t_dev / args.steps * 1e3, 1),
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                if len(lanes) > 1 or defer_T:                              # (untimed: the sample is the head of sub-batch 0, whichever lane mapped it last -- and in ONE batch, nothing handed back)
                    mapper.copts.defer_seed_matches = 0
                    lane_device_side(lanes[0], subs[0])
                    if err:
                        raise err[0]
                for l in timed[1:]:
                    l["ctx"].close()
                out["cpu_baseline"] = cpu_baseline(mapper, reads_h, off_h, args, lanes[0].get("last_res"))
            except Exception as e:                                          # the bench line must survive a baseline problem; say what happened
                out["cpu_baseline"] = {"value": None, "unit": "Gbp/s", "cores": host_cpus()[0], "kind": "port", "sample": "failed: %r" % (e,)}
        out["setup_s"] = {"genome": round(gen_s, 1), "index": round(index_s, 1), "reads": round(sim_s, 1)}
        free_b, total_b = torch.cuda.mem_get_info(dev_index)
        out["hbm_used_gb"] = round((total_b - free_b) / 1e9, 1)               # everything resident at the end of the run: reference, reads, work buffers
        out["hbm_torch_reserved_gb"] = round(torch.cuda.memory_reserved(dev_index) / 1e9, 1)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
