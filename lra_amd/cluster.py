"""Host-side mirror of CleanMatches (reference: Clustering.h:1840) for the current seed result of a context."""
import ctypes as C

import numpy as np

from .context import Context


class CleanOpts(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("globalK", "cleanMaxDiag", "minDiagCluster", "bypassClustering", "cleanClustersize",
                                         "SecondCleanMinDiagCluster", "SecondCleanMaxDiag", "punish_anchorfreq", "anchorPerlength")]


class ClusterResult(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("n_clusters", C.c_uint64), ("n_matches", C.c_uint64), ("d_cluster_off", C.c_void_p),
                ("d_c_start", C.c_void_p), ("d_c_end", C.c_void_p), ("d_c_qStart", C.c_void_p), ("d_c_qEnd", C.c_void_p),
                ("d_c_tStart", C.c_void_p), ("d_c_tEnd", C.c_void_p), ("d_c_strand", C.c_void_p), ("d_c_chrom", C.c_void_p),
                ("d_c_anchorfreq", C.c_void_p), ("d_cl_qpos", C.c_void_p), ("d_cl_tpos", C.c_void_p)]


def clean_matches_batch(ctx: Context, opts: CleanOpts, chrom_pos):
    """chrom_pos = genome.header.pos (cumulative chromosome starts, n_chrom+1 entries)."""
    cp = np.ascontiguousarray(chrom_pos, dtype=np.uint64)
    res = ClusterResult()
    ctx.check(ctx.lib.lra_clean_matches_batch(ctx.h, C.byref(opts), C.c_void_p(cp.ctypes.data), len(cp) - 1, C.byref(res)))
    return res


def fetch(ctx: Context, res: ClusterResult):
    n, nc, nm = res.n_reads, res.n_clusters, res.n_matches
    return {"cluster_off": ctx.to_host(res.d_cluster_off, n + 1, np.uint64),
            "start": ctx.to_host(res.d_c_start, nc, np.uint64), "end": ctx.to_host(res.d_c_end, nc, np.uint64),
            "qStart": ctx.to_host(res.d_c_qStart, nc, np.uint32), "qEnd": ctx.to_host(res.d_c_qEnd, nc, np.uint32),
            "tStart": ctx.to_host(res.d_c_tStart, nc, np.uint32), "tEnd": ctx.to_host(res.d_c_tEnd, nc, np.uint32),
            "strand": ctx.to_host(res.d_c_strand, nc, np.int32), "chrom": ctx.to_host(res.d_c_chrom, nc, np.int32),
            "freq": ctx.to_host(res.d_c_anchorfreq, nc, np.float32),
            "cl_qpos": ctx.to_host(res.d_cl_qpos, nm, np.uint32), "cl_tpos": ctx.to_host(res.d_cl_tpos, nm, np.uint32)}


class ExtendResult(C.Structure):
    _fields_ = [("n_clusters", C.c_uint64), ("n_anchors_cap", C.c_uint64), ("d_e_start", C.c_void_p), ("d_e_count", C.c_void_p),
                ("d_e_qpos", C.c_void_p), ("d_e_tpos", C.c_void_p), ("d_e_len", C.c_void_p), ("d_box", C.c_void_p)]


def linear_extend_batch(ctx: Context, K, read_batch):
    """LinearExtend + DecideCoordinates over the context's current clusters (reads = the seeded ReadBatch)."""
    from .context import ptr
    res = ExtendResult()
    ctx.check(ctx.lib.lra_linear_extend_batch(ctx.h, int(K), ptr(read_batch.seq), ptr(read_batch.off), C.byref(res)))
    return res


def fetch_extend(ctx: Context, res: ExtendResult):
    nc, nm = res.n_clusters, res.n_anchors_cap
    return {"e_start": ctx.to_host(res.d_e_start, nc, np.uint64), "e_count": ctx.to_host(res.d_e_count, nc, np.uint32),
            "e_qpos": ctx.to_host(res.d_e_qpos, nm, np.uint32), "e_tpos": ctx.to_host(res.d_e_tpos, nm, np.uint32),
            "e_len": ctx.to_host(res.d_e_len, nm, np.int32), "box": ctx.to_host(res.d_box, 4 * nc, np.uint32).reshape(-1, 4)}


class FineOpts(C.Structure):
    """lra_fine_opts: Options::globalK, RoughClustermaxGap, maxDiag, maxGap, minClusterSize, minUniqueStretchNum, minUniqueStretchDist"""
    _fields_ = [(n, C.c_int32) for n in ("globalK", "RoughClustermaxGap", "maxDiag", "maxGap", "minClusterSize", "minUniqueStretchNum", "minUniqueStretchDist")]


class FineResult(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("n_clusters", C.c_uint64), ("n_matches", C.c_uint64)] + [
        (n, C.c_void_p) for n in ("d_cluster_off", "d_match_off", "d_q", "d_t", "d_box", "d_strand", "d_chrom", "d_anchorfreq", "d_status")]


# lra.cpp:268-340 over Options.h:123-240
FINE_PRESETS = {"CCS": dict(globalK=17, RoughClustermaxGap=500, maxDiag=500, maxGap=400, minClusterSize=10, minUniqueStretchNum=1, minUniqueStretchDist=50),
                "CONTIG": dict(globalK=19, RoughClustermaxGap=500, maxDiag=100, maxGap=500, minClusterSize=10, minUniqueStretchNum=1, minUniqueStretchDist=50)}


def fine_clusters_batch(ctx: Context, rough: ClusterResult, opts: FineOpts, chrom_pos):
    """MatchesToFineClusters behind clean_matches_batch (run with bypassClustering = 0)."""
    cp = np.ascontiguousarray(chrom_pos, dtype=np.uint64)
    res = FineResult()
    ctx.check(ctx.lib.lra_fine_clusters_batch(ctx.h, C.byref(rough), C.byref(opts), C.c_void_p(cp.ctypes.data), len(cp) - 1, C.byref(res)))
    return res


def fetch_fine(ctx: Context, res: FineResult):
    n, nc, nm = int(res.n_reads), int(res.n_clusters), int(res.n_matches)
    return dict(cluster_off=ctx.to_host(res.d_cluster_off, n + 1, np.uint64), match_off=ctx.to_host(res.d_match_off, nc + 1, np.uint64) if nc else np.zeros(1, np.uint64),
                q=ctx.to_host(res.d_q, nm, np.uint32), t=ctx.to_host(res.d_t, nm, np.uint32), box=ctx.to_host(res.d_box, 4 * nc, np.uint32).reshape(-1, 4),
                strand=ctx.to_host(res.d_strand, nc, np.int32), chrom=ctx.to_host(res.d_chrom, nc, np.int32), freq=ctx.to_host(res.d_anchorfreq, nc, np.float32),
                status=ctx.to_host(res.d_status, n, np.uint32))


class ExtClustersResult(C.Structure):
    _fields_ = [("n_items", C.c_uint64), ("n_anchors", C.c_uint64)] + [
        (n, C.c_void_p) for n in ("d_anchor_off", "d_q", "d_t", "d_len", "d_overlap", "d_box", "d_strand", "d_chrom", "d_anchorfreq")]


def linear_extend_clusters_batch(ctx: Context, item_cluster, item_prev, item_next, item_read, match_off, mq, mt, box, strand, chrom, freq, read_batch, genome_dev, chrom_pos,
                                 skiprepetitive=True, K=17, trim=True):
    """LinearExtend_chain (cluster version of LinearExtend + TrimOverlappedAnchors) for chain elements; array arguments are device tensors."""
    from .context import ptr
    cp = np.ascontiguousarray(chrom_pos, dtype=np.uint64)
    res = ExtClustersResult()
    ctx.check(ctx.lib.lra_linear_extend_clusters_batch(ctx.h, C.c_uint64(int(item_cluster.numel())), ptr(item_cluster), ptr(item_prev), ptr(item_next), ptr(item_read),
                                                       C.c_uint64(int(strand.numel())), ptr(match_off), C.c_uint64(int(mq.numel())), ptr(mq), ptr(mt), ptr(box), ptr(strand),
                                                       ptr(chrom), ptr(freq), ptr(read_batch.seq), ptr(read_batch.off), ptr(genome_dev), C.c_void_p(cp.ctypes.data), len(cp) - 1,
                                                       1 if skiprepetitive else 0, int(K), 1 if trim else 0, C.byref(res)))
    return res


def fetch_ext_clusters(ctx: Context, res: ExtClustersResult):
    n, na = int(res.n_items), int(res.n_anchors)
    return dict(off=ctx.to_host(res.d_anchor_off, n + 1, np.uint64), q=ctx.to_host(res.d_q, na, np.uint32), t=ctx.to_host(res.d_t, na, np.uint32),
                len=ctx.to_host(res.d_len, na, np.int32), overlap=ctx.to_host(res.d_overlap, na, np.uint8), box=ctx.to_host(res.d_box, 4 * n, np.uint32).reshape(-1, 4),
                strand=ctx.to_host(res.d_strand, n, np.int32), chrom=ctx.to_host(res.d_chrom, n, np.int32), freq=ctx.to_host(res.d_anchorfreq, n, np.float32))


class BtwnClustersResult(C.Structure):
    _fields_ = [("n_clusters", C.c_uint64), ("n_matches", C.c_uint64), ("n_problems", C.c_uint64), ("n_pairs_added", C.c_uint64), ("n_rounds", C.c_uint32)] + [
        (n, C.c_void_p) for n in ("d_match_off", "d_q", "d_t", "d_refinespace")]


def refine_btwn_clusters_batch(ctx: Context, read_chain_off, chain_off, ch, match_off, mq, mt, box, strand, chrom, freq, read_off, strands, rc_base, genome_dev, chrom_pos,
                               K=17, W=10, read_type=2, anchorstoosparse=0.005, match=4, mismatch=-3, indel=-4, max_freq=15):
    """RefineBtwnClusters_chain over the chains of a batch (box and freq are updated in place); array arguments are device tensors."""
    from .context import ptr
    cp = np.ascontiguousarray(chrom_pos, dtype=np.uint64)
    res = BtwnClustersResult()
    ctx.check(ctx.lib.lra_refine_btwn_clusters_batch(ctx.h, int(read_chain_off.numel()) - 1, ptr(read_chain_off), C.c_uint64(int(chain_off.numel()) - 1), ptr(chain_off), ptr(ch),
                                                     C.c_uint64(int(strand.numel())), ptr(match_off), C.c_uint64(int(mq.numel())), ptr(mq), ptr(mt), ptr(box), ptr(strand), ptr(chrom),
                                                     ptr(freq), ptr(read_off), ptr(strands), C.c_uint64(int(rc_base)), ptr(genome_dev), C.c_void_p(cp.ctypes.data), len(cp) - 1,
                                                     int(K), int(W), int(read_type), C.c_float(anchorstoosparse), int(match), int(mismatch), int(indel), int(max_freq), C.byref(res)))
    return res
