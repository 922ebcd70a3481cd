"""Time the host tail (lra_map_records: grouping, ordering, MAPQ, SAM text) next to the device side on a bench-like batch (needs the GPU)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lra_amd import synth_torch as st, seed, mapread
from lra_amd.context import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda", 0)
ctx = Context(0)
genome = st.make_genome(16_000_000, 1, dev)
ik, ip = st.build_global_index(genome, 17, 10, 150)
sim = st.simulate_batch(genome, n, 30000, 3000, 0.10, (30, 35, 35), 5)
reads = torch.cat([sim["seq"], torch.zeros(64, dtype=torch.uint8, device=dev)])
mapper = mapread.LowAccMapper(ctx, genome, ik, ip, [b"chr1"], [0, int(genome.numel())])
rb = seed.read_batch_from_device(ctx, reads, sim["off"])
mapper.align(rb)
torch.cuda.synchronize(); t0 = time.time()
res = mapper.align(rb)
torch.cuda.synchronize(); t1 = time.time()
off = sim["off"].cpu().numpy(); h = reads.cpu().numpy()
rl = [h[off[i]:off[i + 1]].tobytes() for i in range(n)]
t2 = time.time()
txt = mapper.records(res, [b"r%d" % i for i in range(n)], rl)
t3 = time.time()
print("reads %d  bases %d  align %.3f s  records %.3f s  text %.1f MB  -> records %.0f reads/s (1 host thread)" % (
    n, int(off[-1]), t1 - t0, t3 - t2, sum(len(x) for x in txt) / 1e6, n / (t3 - t2)))
if os.environ.get("LRA_TIME_TAGS"):
    ctx.timing(True); ctx.timing_reset()
    mapper.align(rb)
    tags = ["sketch_emit", "sketch_compact", "sort", "index_bounds", "compare", "strand", "clean_sort", "clean", "linear_extend", "sdp_points", "sdp_sort", "sdp_build_count",
            "sdp_build", "sdp_process", "sdp_trace", "chain_split", "create_rc", "local_sketch", "local_sort_filter", "local_compare", "rsc_tasks", "rsc_filter", "refine_space",
            "rs_long_sketch", "rs_long_compare", "btwn_plan", "btwn_apply", "merge_extend", "between_anchors", "local_refine", "aog_lds_tiny", "aog_lds_small", "aog_lds_medium",
            "aog_lds_large", "aog_hbm", "ir_segment", "ir_band", "ir_fill", "ir_trace", "ir_gather", "stats", "sdp_inner_process", "sdp_inner_build", "sdp_sort_fallback", "sort_fallback"]
    tot = 0
    for t in tags:
        try:
            ms, nl = ctx.timing_get(t)
        except Exception:
            continue
        tot += ms
        if ms > 5: print("%-20s %8.1f ms %4d launches" % (t, ms, nl))
    print("sum", tot)
