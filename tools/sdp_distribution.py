"""Distribution of the per-read problem sizes on the bench workload (run on the GPU box): tier-1 matches, clusters, SDP#A anchors per read, which
reads they come from (satellite array / interspersed repeat / unique), and the time of the sparse-DP kernels."""
import argparse, ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lra_amd import seed, cluster, chain, mapread, synth_genome as sg
from lra_amd.context import Context

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=4096)
ap.add_argument("--genome-scale", type=float, default=0.05)
ap.add_argument("--sat", type=float, default=0.03)
a = ap.parse_args()
dev = torch.device("cuda", 0)
genome, cp, names = sg.make_grch38_like(dev, scale=a.genome_scale, seed=3, satellite_frac=a.sat)
ctx = Context(0)
mapper = mapread.LowAccMapper(ctx, genome, None, None, names, cp, mapread.LowAccOptions(), staged=False)
sim = sg.simulate_reads_sv(genome, cp, a.reads, 30000, 3000, 0.10, (30, 35, 35), 1000)
rb = seed.read_batch_from_device(ctx, sim["seq"], sim["off"])
sres = seed.seed_batch(ctx, rb, 17, 10, 150)
mo = ctx.to_host(sres.d_match_off, a.reads + 1, np.uint64).astype(np.int64)
nm = np.diff(mo)
cres = cluster.clean_matches_batch(ctx, mapper.clean_opts, cp)
co = ctx.to_host(cres.d_cluster_off, a.reads + 1, np.uint64).astype(np.int64)
eres = cluster.linear_extend_batch(ctx, 17, rb)
ctx.timing(True); ctx.timing_reset()
res = chain.sparse_dp_batch(ctx, a.reads, cres.d_cluster_off, eres.d_e_start, eres.d_e_count, cres.d_c_strand, eres.d_e_qpos, eres.d_e_tpos, eres.d_e_len,
                            rb.off, chain.sdp_opts())
out = chain.fetch(ctx, res)
fr = np.diff(out["frag_off"].astype(np.int64))
# where the reads come from: inside a satellite array?
start = sim["start"].cpu().numpy(); chrom = sim["chrom"].cpu().numpy()
cpn = np.asarray(cp)
clen = cpn[chrom + 1] - cpn[chrom]
rel = (start - cpn[chrom]) / clen
in_sat = (rel > 0.40 - 30000 / clen) & (rel < 0.40 + a.sat)
def q(x): return "mean %.0f p50 %d p90 %d p99 %d p99.9 %d max %d" % (x.mean(), *np.percentile(x, [50, 90, 99, 99.9]).astype(int), x.max())
print("index entries", mapper.index_stats)
print("matches/read  ", q(nm)); print("clusters/read ", q(np.diff(co))); print("SDP#A anchors ", q(fr))
print("reads in satellite arrays: %d of %d; their matches/read %s" % (in_sat.sum(), a.reads, q(nm[in_sat]) if in_sat.any() else "-"))
print("                              their SDP#A anchors %s" % (q(fr[in_sat]) if in_sat.any() else "-"))
print("other reads: matches %s ; anchors %s" % (q(nm[~in_sat]), q(fr[~in_sat])))
print("share of anchors in the top 1%% reads: %.2f" % (np.sort(fr)[-max(1, len(fr) // 100):].sum() / max(fr.sum(), 1)))
print("status nonzero:", int((out["status"] != 0).sum()), " chains/read: %.2f" % out["n_chains"].mean())
for k in ("sort", "sort_fallback", "compare", "clean", "sdp_points", "sdp_sort", "sdp_sort_fallback", "sdp_build_count", "sdp_build", "sdp_process", "sdp_trace"):
    print(k, ctx.timing_get(k))
