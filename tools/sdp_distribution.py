"""Distribution of the SDP#A problem size per read on the bench workload (run on the GPU box): anchors, points, entries per read."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from lra_amd import seed, cluster, chain
from lra_amd.context import Context

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=8192)
a = ap.parse_args()
args = argparse.Namespace(genome_mb=64, reads=a.reads, read_len=30000, err=0.10, k=17, w=10, max_freq=150, refine_band=7)
dev = torch.device("cuda", 0)
ref = bench.build_reference(args, dev)
wl = bench.build_workload(args, 0, dev, ref, a.reads, 0)
ctx = Context(0)
seed.load_reference(ctx, wl["genome"].cpu().numpy(), wl["idx_key"], wl["idx_pos"])
rb = seed.read_batch_from_device(ctx, wl["reads"], wl["sim"]["off"])
seed.seed_batch(ctx, rb, 17, 10, 150)
copts = cluster.CleanOpts(globalK=17, cleanMaxDiag=200, minDiagCluster=3, bypassClustering=1, cleanClustersize=100, SecondCleanMinDiagCluster=10,
                          SecondCleanMaxDiag=100, punish_anchorfreq=5, anchorPerlength=5)
cres = cluster.clean_matches_batch(ctx, copts, [0, int(wl["genome"].numel())])
eres = cluster.linear_extend_batch(ctx, 17, rb)
ctx.timing(True); ctx.timing_reset()
res = chain.sparse_dp_batch(ctx, a.reads, cres.d_cluster_off, eres.d_e_start, eres.d_e_count, cres.d_c_strand, eres.d_e_qpos, eres.d_e_tpos, eres.d_e_len,
                            rb.off, chain.sdp_opts())
out = chain.fetch(ctx, res)
fr = np.diff(out["frag_off"].astype(np.int64))
print("reads", a.reads, "anchors/read: mean %.0f  p50 %d  p90 %d  p99 %d  p99.9 %d  max %d" % (fr.mean(), *np.percentile(fr, [50, 90, 99, 99.9]).astype(int), fr.max()))
print("share of anchors in the top 1%% reads: %.2f" % (np.sort(fr)[-max(1, len(fr) // 100):].sum() / fr.sum()))
print("status nonzero:", int((out["status"] != 0).sum()), " chains/read: %.2f" % out["n_chains"].mean())
for k in ("sdp_points", "sdp_sort", "sdp_sort_fallback", "sdp_build_count", "sdp_build", "sdp_process", "sdp_trace"):
    print(k, ctx.timing_get(k))
