"""One batch through lra_map_reads_lowacc_batch on the bench workload with debug output (LRA_SDP_DBG=1 etc.); prints the kernel times."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lra_amd import seed, mapread, synth_genome as sg
from lra_amd.context import Context
ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=8192)
ap.add_argument("--genome-scale", type=float, default=1.0)
ap.add_argument("--sat", type=float, default=0.03)
a = ap.parse_args()
dev = torch.device("cuda", 0)
genome, cp, names = sg.make_grch38_like(dev, scale=a.genome_scale, seed=3, satellite_frac=a.sat)
ctx = Context(0)
mapper = mapread.LowAccMapper(ctx, genome, None, None, names, cp, mapread.LowAccOptions(), staged=False)
sim = sg.simulate_reads_sv(genome, cp, a.reads, 30000, 3000, 0.10, (30, 35, 35), 1000)
rb = seed.read_batch_from_device(ctx, sim["seq"], sim["off"])
mapper.align(rb)
ctx.timing(True); ctx.timing_reset()
res = mapper.align(rb)
ks = ["sort", "sort_fallback", "compare", "clean", "sdp_sort", "sdp_sort_fallback", "sdp_build_count", "sdp_build", "sdp_process", "sdp_inner_process", "aog_hbm", "aog_lds_large",
      "aog_lds_medium", "rs_long_sketch", "rs_long_compare", "ir_fill", "local_compare", "stats"]
print({k: round(ctx.timing_get(k)[0], 1) for k in ks})
print(mapper.stats)
