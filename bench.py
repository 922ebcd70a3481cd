#!/usr/bin/env python
"""bench.py -- throughput of the MI355X hot path on synthetic long reads.

One "step" = MapRead_lowacc (the -ONT / -CLR path of `lra align`) over one batch of reads already in HBM, every stage consuming what the previous one
produced on the device:
  a1-a4  tier-1 seeding   (StoreMinimizers -> sort -> CompareLists -> SeparateMatchesByStrand)
  a5     CleanMatches;  a7 LinearExtend + DecideCoordinates;  a8 SparseDP (SDP#A)
  a9     chain filters, SPLITChain;  a10 CreateRC, LocalIndex::IndexSeq of both strands, Refine_splitchain
  a11    Refine_Btwnsplitchain;  a9 MergeChain;  a7 second LinearExtend + TrimOverlappedAnchors;  a8 the per-merged-cluster sparse DP + its filters
  a13    LocalRefineAlignment (incl. a12 AffineOneGapAlign between anchors, RefineSpace + inner sparse DP on large spaces)
  a14    IndelRefineAlignment on those alignments;  a16 CalculateStatistics (CIGAR runs, NM/NX/ND/NI/TD/TI counters, NV)
Not in the step: RefineBreakpoint (built, off by default in lra), the per-read MAPQ / ordering / SAM text (host code, built).

Contract: python bench.py --gpus N --steps K --warmup W  -> rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def build_reference(args, device):
    """Genome + global index, generated on the GPU with bulk tensor ops (seconds)."""
    from lra_amd import synth_torch as st
    genome = st.make_genome(int(args.genome_mb * 1e6), 1, device)
    idx_key, idx_pos = st.build_global_index(genome, args.k, args.w, 150)
    return genome, idx_key, idx_pos


def build_workload(args, rank, device, ref, n_reads, lane):
    """One lane's reads (seeded per rank and lane) and the truth-derived inputs of the stages behind a13."""
    import torch
    from lra_amd import synth_torch as st
    t0 = time.time()
    genome, idx_key, idx_pos = ref
    sim = st.simulate_batch(genome, n_reads, args.read_len, args.read_len / 10, args.err, (30, 35, 35), 1000 + rank + 7919 * lane)
    pad = torch.zeros(64, dtype=torch.uint8, device=device)
    strands = torch.cat([sim["seq"], pad])                                   # the strand every alignment lies on
    g2 = torch.Generator(device=device).manual_seed(77 + rank + 7919 * lane)
    rev = torch.rand(n_reads, generator=g2, device=device) < 0.5
    reads = torch.cat([st.revcomp_some(sim["seq"], sim["off"], rev), pad])   # what the sequencer gave us (half reverse strand)
    gaps = st.gap_problems(sim)
    rblocks, rboff = st.perturbed_blocks(sim, 5 + rank + 7919 * lane)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()                                                 # hand the generator's temporaries back: the stages need the room
    return dict(genome=genome, idx_key=idx_key, idx_pos=idx_pos, sim=sim, strands=strands, reads=reads, gaps=gaps, rev=rev,
                rblocks=rblocks, rboff=rboff, gen_s=time.time() - t0)


def cpu_baseline(wl, args, mapper, budget_s=20.0, max_reads=768):
    """The oracle (CPU restatement) timed single-threaded on a bounded sample of the same workload: MapRead_lowacc read by read through
    tests/oracle_pipeline.py, the same stages in the same order as the GPU step (its alignments equal the GPU's: tests/test_mapread.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    import oracle_pipeline as OP
    O.lib()
    sim = wl["sim"]
    S = min(max_reads, int(sim["off"].numel()) - 1)
    off = sim["off"][:S + 1].cpu().numpy()
    reads = wl["reads"][:int(off[-1])].cpu().numpy()
    g = wl["genome"].cpu().numpy().tobytes() + b"\0" * 64
    # the reference side (read-only, built once in the reference too): the genome's local index
    g_win, g_bnd, g_tup = mapper.gli.fetch()
    g_index = (OP.seq_offsets(len(g) - 64, mapper.opts.localIndexWindow), g_bnd, g_tup)
    opts = dict(globalK=args.k, globalW=args.w, globalMaxFreq=args.max_freq, refineBand=args.refine_band)
    t0 = time.time()
    bases = n = n_aln = 0
    for r in range(S):
        rbytes = reads[off[r]:off[r + 1]].tobytes()
        alns, _ = OP.map_read_lowacc(rbytes, g, wl["idx_key"], wl["idx_pos"], g_index, opts)
        n_aln += sum(len(x) for x in alns)
        bases += len(rbytes)
        n += 1
        if time.time() - t0 > budget_s:
            break
    dt = time.time() - t0
    return {"value": bases / dt / 1e9, "unit": "Gbp/s", "cores": 1, "kind": "port",
            "sample": "first %d reads (%d bp, %d alignments) of the same batch through the oracle's MapRead_lowacc (a1-a5, a7-a11, a13 incl. a12, a14, a16: "
                      "the stages of the GPU step, tests/oracle_pipeline.py) in %.1f s, 1 thread (python ctypes call overhead included)" % (n, bases, n_aln, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-mb", type=float, default=float(os.environ.get("LRA_BENCH_GENOME_MB", 64)))
    ap.add_argument("--reads", type=int, default=int(os.environ.get("LRA_BENCH_READS", 32768)), help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=30000)
    ap.add_argument("--err", type=float, default=0.10)
    ap.add_argument("--k", type=int, default=17)          # -ONT: globalK 17, globalW 10 (lra.cpp:386-431)
    ap.add_argument("--w", type=int, default=10)
    ap.add_argument("--max-freq", type=int, default=150)
    ap.add_argument("--refine-band", type=int, default=7)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("LRA_BENCH_LANES", 1)),
                    help="the batch is cut into this many sub-batches, each driven by its own context and HIP stream from its own host thread, so "
                         "that the serial tails of one sub-batch's kernels overlap the other's work")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev_index = local_rank if world > 1 else 0
    torch.cuda.set_device(dev_index)

    from lra_amd.context import Context
    from lra_amd import seed, parallel, mapread
    mopts = mapread.LowAccOptions(globalK=args.k, globalW=args.w, globalMaxFreq=args.max_freq, refineBand=args.refine_band)   # -ONT

    import threading
    errors = []

    def make_lane(lane, n_reads, ref):
        """One sub-batch: its own context (buffers + HIP stream), mapper and reads; returns (ctx, step, stats, constants)."""
        stream = torch.cuda.Stream(device=dev_index) if args.lanes > 1 else None
        ctx = Context(dev_index)
        if stream is not None:
            ctx.bind_stream(stream)
        wl = build_workload(args, rank, ctx.device, ref, n_reads, lane)
        G = int(wl["genome"].numel())
        # the reference side, built once: global index, genome, the genome's local index (.gli)
        mapper = mapread.LowAccMapper(ctx, wl["genome"], wl["idx_key"], wl["idx_pos"], [b"chr1"], [0, G], mopts)
        sim = wl["sim"]
        rbatch = seed.read_batch_from_device(ctx, wl["reads"], sim["off"])
        gp = wl["gaps"]
        lens = (sim["off"][1:] - sim["off"][:-1])
        total_bases = int(lens.sum())
        n_gap_bytes = int(gp["q_len"].sum() + gp["t_len"].sum())
        n_gaps = int(gp["k"].numel())

        stats = mapper.stats
        out_rec = [None]

        def step():
            # MapRead_lowacc for the whole batch behind the C boundary (lra_map_reads_lowacc_batch): a1-a5, a7-a11, a13, a14, a16
            res = mapper.align(rbatch)
            out_rec[0] = mapper.block_records(res)                       # the one exchange step: refined block records -> rank 0

        def run_step():
            try:
                torch.cuda.set_device(dev_index)                         # the current device is per host thread
                if stream is not None:
                    with torch.cuda.stream(stream):
                        step()
                else:
                    step()
            except BaseException as e:                                   # surfaced by the caller: a thread's exception would vanish otherwise
                errors.append(e)
        return dict(ctx=ctx, step=run_step, stats=stats, wl=wl, mapper=mapper, total_bases=total_bases, n_gap_bytes=n_gap_bytes, n_gaps=n_gaps, out_rec=out_rec)

    ref = build_reference(args, torch.device("cuda", dev_index))
    per_lane = [args.reads // args.lanes + (1 if i < args.reads % args.lanes else 0) for i in range(args.lanes)]
    lanes = [make_lane(i, per_lane[i], ref) for i in range(args.lanes)]
    ctx = lanes[0]["ctx"]
    wl = lanes[0]["wl"]
    torch.cuda.synchronize()
    total_bases = sum(l["total_bases"] for l in lanes)
    n_gap_bytes = sum(l["n_gap_bytes"] for l in lanes); n_gaps = sum(l["n_gaps"] for l in lanes)

    def step():
        if len(lanes) == 1:
            lanes[0]["step"]()
        else:
            ths = [threading.Thread(target=l["step"]) for l in lanes]
            for t in ths: t.start()
            for t in ths: t.join()
        if errors:
            raise errors[0]
        torch.cuda.synchronize()
        # the one exchange step: refined block records -> rank 0
        rec = torch.cat([l["out_rec"][0] for l in lanes]) if len(lanes) > 1 else lanes[0]["out_rec"][0]
        parallel.gather_records(rec, dst=0)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    for l in lanes:
        l["ctx"].timing(True)
        l["ctx"].timing_reset()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=ctx.device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        tb = torch.tensor([total_bases], dtype=torch.int64, device=ctx.device)
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        job_bases = int(tb.item())
        nreads = args.reads * world
    else:
        job_bases = total_bases
        nreads = args.reads

    kernels = ["sketch_count", "sketch_serial", "sketch_emit", "sort", "sort_fallback", "index_bounds", "compare", "strand",
               "aog_lds_tiny", "aog_lds_small", "aog_lds_medium", "aog_lds_large", "aog_hbm", "ir_segment", "ir_band", "ir_fill", "ir_trace", "ir_gather", "clean_sort", "clean", "linear_extend", "stats", "stats_cigar", "create_rc", "local_sketch", "local_sort_filter", "local_compare",
               "rsc_tasks", "rsc_filter", "refine_space", "rs_long_sketch", "rs_long_compare", "btwn_plan", "btwn_apply", "merge_extend", "between_anchors", "local_refine", "sdp_inner_points", "sdp_inner_sort", "sdp_inner_build_count", "sdp_inner_build", "sdp_inner_process", "sdp_inner_trace", "chain_split", "sdp_points", "sdp_sort", "sdp_sort_fallback", "sdp_build_count", "sdp_build", "sdp_process", "sdp_trace"]
    ktimes = {}
    for k in kernels:
        tt_ = [l["ctx"].timing_get(k) for l in lanes]
        ktimes[k] = (sum(x[0] for x in tt_), sum(x[1] for x in tt_))
    for l in lanes:
        l["ctx"].timing(False)
    stats = {}
    for l in lanes:
        for k, v in l["stats"].items():
            stats[k] = stats.get(k, 0) + v if isinstance(v, (int, float)) else v
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        gbps = job_bases * args.steps / dt / 1e9
        dom = max(ktimes, key=lambda k: ktimes[k][0])
        dom_ms, dom_n = ktimes[dom]
        avg_ms = dom_ms / max(dom_n, 1)
        launches_per_step = max(dom_n, 1) / args.steps
        # algorithmic bytes PER STEP of the dominant kernel, all its launches together (DESIGN.md section 3 gives the per-unit figures)
        L = total_bases
        sdp_entries = stats.get("n_sdp_entries", 0) + stats.get("n_sdp2_entries", 0)
        sdp_points = stats.get("n_sdp_points", 0) + 2 * stats.get("n_sdp2_anchors", 0)
        alg_step = {
            "ir_fill": 1 * stats["n_cells"] + 16 * stats["n_rows"] + 1 * stats["n_rows"],        # 1 B arrow/cell + row windows + both sequences
            "ir_band": 16 * stats["n_rows"] + 12 * stats["n_blocks"],
            "ir_trace": 1 * stats["n_rows"] + 16 * stats["n_rows"] + 12 * stats["n_blocks"],   # ~1 arrow + 1 row record per row walked
            "sort": 2 * 12 * stats["n_mm"],
            "index_bounds": stats["n_mm"] * (12 + 64 + 8),
            "compare": stats["n_mm"] * (8 + 8 + 64) + 8 * stats["n_match"],
            "sketch_count": L, "sketch_emit": L + 12 * stats["n_mm"],
            "strand": stats["n_match"] * (8 + 8 + 2 * args.k),
            "local_compare": 2 * 4 * stats.get("n_local_task_words", 0) + 8 * stats.get("n_local_pairs", 0),   # count + emit passes, both strands
            "local_sort_filter": 2 * 4 * stats.get("n_local_tuples", 0),
            "rsc_tasks": 2 * (21 * stats.get("n_sdp_anchors", 0) + 36 * stats.get("n_local_tasks", 0)),
            "rsc_filter": 2 * 8 * stats.get("n_local_pairs", 0) + 8 * stats.get("n_refined_matches", 0),
            "local_sketch": 2 * 2 * L + 4 * stats.get("n_local_tuples", 0),
            "stats": 2 * L + 12 * stats["n_blocks"] + 4 * stats.get("n_cigar_runs", 0),
            "clean": 16 * stats["n_match"] * 3,
            "aog_lds_small": n_gap_bytes + 12 * n_gaps,
            "aog_lds_tiny": n_gap_bytes + 12 * n_gaps,
            "ir_segment": 12 * stats["n_blocks"],
            # a8: 44 B per sub-problem entry (Di/Ei + Db/Eb + value 16, back pointer 4, stack 8, Block 16) + the 256 B visit row and 13 B
            # of coordinates per point
            # (both sparse DPs of the step: SDP#A and the per-merged-cluster one, whose anchors give two points each)
            "sdp_process": 44 * sdp_entries + 269 * sdp_points,
            "sdp_build": 20 * sdp_entries + 269 * sdp_points,
            "sdp_build_count": 13 * sdp_points,
            "sdp_sort": 4 * 2 * 12 * sdp_points,
        }.get(dom, 0)
        alg = alg_step / launches_per_step
        achieved = alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM traffic per launch: from the committed PMC pass (profiles/pmc_latest.json) when it was taken on launches of the same size
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))["kernels"].get(dom)
            if pmc and abs(pmc["reads_per_launch"] - args.reads / launches_per_step) < 1:
                traffic = pmc["fetch_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        out = {
            "metric": "aligned Gbp/s (hot-path stages a1-a5, a7, a8 SDP#A, a9 split + MergeChain, a10 Refine_splitchain, a11 Refine_Btwnsplitchain, a7/a8 second pass, a13 LocalRefineAlignment incl. a12, a14, a16 = MapRead_lowacc), 30 kb ONT-like reads", "value": gbps, "unit": "Gbp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "reads_per_s": nreads * args.steps / dt,
            "config": {"workload": "synthetic %g Mb chromosome (chr20-sized, BASELINE configs[1] reference) + %d reads/GPU of N(%d, 10%%) bp, %g%% "
                                   "error 30:35:35 (BASELINE configs[2] -ONT read profile; full GRCh38 not generated in round 1)"
                                   % (args.genome_mb, args.reads, args.read_len, args.err * 100),
                       "preset": "-ONT (k=%d w=%d maxFreq=%d refineBand=%d match/mismatch/indel=4/-1/-2)" % (args.k, args.w, args.max_freq, args.refine_band),
                       "stages": "MapRead_lowacc chained on the reads, every stage on the alignments the previous one produced: a1-a5, a7, a8 (SDP#A), a9 (chain filters, SPLITChain), "
                                 "a10 (Refine_splitchain), a11 (Refine_Btwnsplitchain), a9 (MergeChain), a7 (second LinearExtend + Trim), a8 (second SDP + filters), a13 "
                                 "(LocalRefineAlignment incl. a12 AffineOneGapAlign between anchors), a14 (IndelRefineAlignment), a16 (CalculateStatistics).  Not in the step: "
                                 "RefineBreakpoint (a15, built; off by default in lra), MAPQ / ordering / SAM text (a16-a17 host code, built)",
                       "parallelism": "reads sharded by ordinal, 1 process/GPU, %d sub-batches per process on their own HIP streams; RCCL gather of block records to rank 0" % args.lanes,
                       "per_step": {k: int(v) for k, v in stats.items() if not k.startswith("_")}},
            "kernel_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in ktimes.items() if v[1]},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": traffic, "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": alg},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, args, lanes[0]["mapper"])
        out["setup_s"] = round(wl["gen_s"], 1)
        free_b, total_b = torch.cuda.mem_get_info(dev_index)
        out["hbm_used_gb"] = round((total_b - free_b) / 1e9, 1)               # everything resident at the end of the run: reference, reads, work buffers
        out["hbm_torch_reserved_gb"] = round(torch.cuda.memory_reserved(dev_index) / 1e9, 1)   # of which torch's allocator (workload + gathered records)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
