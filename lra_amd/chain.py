"""Host-side mirror of SparseDP (SDP#A; reference: SparseDP.h:2139, called at Map_lowacc.h:188) for a batch of reads."""
import ctypes as C

import numpy as np

from .context import Context, ptr


class SdpOpts(C.Structure):
    _fields_ = [("rate", C.c_float), ("NumAln", C.c_int32), ("alnthres", C.c_float), ("gapopen", C.c_float), ("gapextend", C.c_float),
                ("gaproot", C.c_float), ("gapCeiling1", C.c_int32), ("gapCeiling2", C.c_int32), ("mode", C.c_int32),
                ("globalK", C.c_int32)]


# -ONT preset (lra.cpp:388-420; alnthres 0.65: lra.cpp:429)
ONT = dict(rate=20.0, NumAln=2, alnthres=0.65, gapopen=7.0, gapextend=10.0, gaproot=1.5, gapCeiling1=1500, gapCeiling2=3000, mode=0, globalK=17)


def sdp_opts(**kw):
    d = dict(ONT); d.update(kw)
    return SdpOpts(d["rate"], d["NumAln"], d["alnthres"], d["gapopen"], d["gapextend"], d["gaproot"], d["gapCeiling1"], d["gapCeiling2"], d["mode"],
                   d["globalK"])


class ChainResult(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("num_aln", C.c_int32), ("n_frags", C.c_uint64), ("n_points", C.c_uint64),
                ("n_subproblem_entries", C.c_uint64), ("d_n_chains", C.c_void_p), ("d_chain_start", C.c_void_p), ("d_chain_len", C.c_void_p),
                ("d_chain_box", C.c_void_p), ("d_chain_value", C.c_void_p), ("d_chain_cluster", C.c_void_p), ("d_chain_anchor", C.c_void_p),
                ("d_chain_link", C.c_void_p), ("d_chain_q", C.c_void_p), ("d_chain_t", C.c_void_p), ("d_chain_alen", C.c_void_p),
                ("d_chain_strand", C.c_void_p), ("d_frag_off", C.c_void_p), ("d_frag_val", C.c_void_p), ("d_status", C.c_void_p),
                ("d_chain_num_anchors", C.c_void_p)]


def sparse_dp_batch(ctx: Context, n_reads, cluster_off, c_start, c_count, c_strand, q, t, length, read_off, opts: SdpOpts, rate=None):
    """All array arguments are device tensors or raw device addresses (see include/lra_hip.h)."""
    res = ChainResult()
    ctx.check(ctx.lib.lra_sparse_dp_batch(ctx.h, int(n_reads), ptr(cluster_off), ptr(c_start), ptr(c_count), ptr(c_strand), ptr(q), ptr(t),
                                          ptr(length), ptr(read_off), ptr(rate) if rate is not None else None, C.byref(opts), C.byref(res)))
    return res


def sparse_dp_boxes_batch(ctx: Context, n_reads, box_off, qs, qe, ts, te, strand, val, num_anchors, read_off, opts: SdpOpts, rate=None):
    """The high-accuracy SparseDP over split-cluster boxes (SparseDP.h:1956); array arguments are device tensors (see include/lra_hip.h)."""
    res = ChainResult()
    ctx.check(ctx.lib.lra_sparse_dp_boxes_batch(ctx.h, int(n_reads), ptr(box_off), ptr(qs), ptr(qe), ptr(ts), ptr(te), ptr(strand), ptr(val),
                                                ptr(num_anchors) if num_anchors is not None else None, ptr(read_off),
                                                ptr(rate) if rate is not None else None, C.byref(opts), C.byref(res)))
    return res


def fetch(ctx: Context, res: ChainResult):
    n, na, nf = res.n_reads, res.num_aln, res.n_frags
    extra = {"chain_num_anchors": ctx.to_host(res.d_chain_num_anchors, n * na, np.int32)} if res.d_chain_num_anchors else {}
    return {**extra, "n_chains": ctx.to_host(res.d_n_chains, n, np.uint32), "chain_start": ctx.to_host(res.d_chain_start, n * na, np.uint64),
            "chain_len": ctx.to_host(res.d_chain_len, n * na, np.uint32), "chain_box": ctx.to_host(res.d_chain_box, 4 * n * na, np.uint32).reshape(-1, 4),
            "chain_value": ctx.to_host(res.d_chain_value, n * na, np.float32), "chain_cluster": ctx.to_host(res.d_chain_cluster, nf, np.uint32),
            "chain_anchor": ctx.to_host(res.d_chain_anchor, nf, np.uint32), "chain_link": ctx.to_host(res.d_chain_link, nf, np.uint8),
            "chain_q": ctx.to_host(res.d_chain_q, nf, np.uint32), "chain_t": ctx.to_host(res.d_chain_t, nf, np.uint32),
            "chain_alen": ctx.to_host(res.d_chain_alen, nf, np.int32), "chain_strand": ctx.to_host(res.d_chain_strand, nf, np.uint8),
            "frag_off": ctx.to_host(res.d_frag_off, n + 1, np.uint64), "frag_val": ctx.to_host(res.d_frag_val, nf, np.float32),
            "status": ctx.to_host(res.d_status, n, np.uint32)}


class SplitResult(C.Structure):
    _fields_ = [("n_slots", C.c_uint64), ("n_frags", C.c_uint64)] + [(n, C.c_void_p) for n in (
        "d_keep", "d_n_kept", "d_link", "d_n_split", "d_sp_beg", "d_sp_len", "d_sp_idx", "d_sp_link", "d_sp_type", "d_sp_strand", "d_sp_chrom",
        "d_sp_box", "d_ci_beg", "d_ci_len", "d_ci_idx", "d_split_link", "d_n_split_link", "d_status", "d_fidx")]


def split_chains_batch(ctx: Context, chains: ChainResult, chrom_pos, splitdist=50000, bypass=1):
    """RemoveSpuriousJump + SPLITChain + RemoveSpuriousSplitChain (Map_lowacc.h:189-256) on every chain of an SDP#A result."""
    cp = np.ascontiguousarray(chrom_pos, dtype=np.uint64)
    res = SplitResult()
    ctx.check(ctx.lib.lra_split_chains_batch(ctx.h, C.byref(chains), C.c_void_p(cp.ctypes.data), len(cp) - 1, int(splitdist), int(bypass), C.byref(res)))
    return res


def fetch_split(ctx: Context, res: SplitResult):
    ns, nf = res.n_slots, res.n_frags
    u8, u32 = np.uint8, np.uint32
    return {"keep": ctx.to_host(res.d_keep, nf, u8), "n_kept": ctx.to_host(res.d_n_kept, ns, u32), "link": ctx.to_host(res.d_link, nf, u8),
            "n_split": ctx.to_host(res.d_n_split, ns, u32), "sp_beg": ctx.to_host(res.d_sp_beg, nf, u32), "sp_len": ctx.to_host(res.d_sp_len, nf, u32),
            "sp_idx": ctx.to_host(res.d_sp_idx, nf, u32), "sp_link": ctx.to_host(res.d_sp_link, nf, u8), "sp_type": ctx.to_host(res.d_sp_type, nf, u8),
            "sp_strand": ctx.to_host(res.d_sp_strand, nf, u8), "sp_chrom": ctx.to_host(res.d_sp_chrom, nf, np.int32),
            "sp_box": ctx.to_host(res.d_sp_box, 4 * nf, u32).reshape(-1, 4), "ci_beg": ctx.to_host(res.d_ci_beg, nf, u32),
            "ci_len": ctx.to_host(res.d_ci_len, nf, u32), "ci_idx": ctx.to_host(res.d_ci_idx, nf, u32),
            "split_link": ctx.to_host(res.d_split_link, nf, u8), "n_split_link": ctx.to_host(res.d_n_split_link, ns, u32),
            "status": ctx.to_host(res.d_status, ns, u32)}


class FilterResult(C.Structure):
    _fields_ = [("n_chains", C.c_uint64), ("n_anchors", C.c_uint64), ("d_keep", C.c_void_p), ("d_n_kept", C.c_void_p), ("d_link", C.c_void_p),
                ("d_n_link", C.c_void_p)]


def filter_chains_batch(ctx: Context, n_chains, off, n_anchors, q, t, length, strand, link, ops, qend=None):
    """Chain.h filters (ops: 1 RemoveSmallPairedIndels, 2/3 RemovePairedIndels with/without refineEnds, 4 RemoveSpuriousAnchors,
    8 RemoveSpuriousJump) on CSR chains; array arguments are device tensors, link may be None."""
    o = np.ascontiguousarray(ops, dtype=np.int32)
    res = FilterResult()
    ctx.check(ctx.lib.lra_filter_chains_ex_batch(ctx.h, C.c_uint64(n_chains), ptr(off), C.c_uint64(n_anchors), ptr(q), ptr(t), ptr(length),
                                                 ptr(qend) if qend is not None else None, ptr(strand), ptr(link) if link is not None else None,
                                                 C.c_void_p(o.ctypes.data), len(o), C.byref(res)))
    return res


def fetch_filter(ctx: Context, res: FilterResult):
    return {"keep": ctx.to_host(res.d_keep, res.n_anchors, np.uint8), "n_kept": ctx.to_host(res.d_n_kept, res.n_chains, np.uint32),
            "link": ctx.to_host(res.d_link, res.n_anchors, np.uint8), "n_link": ctx.to_host(res.d_n_link, res.n_chains, np.uint32)}


class SplitClustersResult(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("n_clusters", C.c_uint64), ("n_split", C.c_uint64)] + [(n, C.c_void_p) for n in (
        "d_split_off", "d_qs", "d_qe", "d_ts", "d_te", "d_strand", "d_coarse", "d_val", "d_num_anchors", "d_read", "d_cluster_val",
        "d_cluster_split")]


def split_clusters_batch(ctx: Context, n_reads, cluster_off, qs, qe, ts, te, strand, anchorfreq, match_off, match_q, contig=False, K=17):
    """SplitClusters + DecideSplitClustersValue (SplitClusters.h:63,176; high-accuracy path) for a batch; device tensors in."""
    res = SplitClustersResult()
    ctx.check(ctx.lib.lra_split_clusters_batch(ctx.h, int(n_reads), ptr(cluster_off), ptr(qs), ptr(qe), ptr(ts), ptr(te), ptr(strand), ptr(anchorfreq),
                                               ptr(match_off), ptr(match_q), 1 if contig else 0, int(K), C.byref(res)))
    return res


def fetch_split_clusters(ctx: Context, res: SplitClustersResult):
    n, ns, nc = res.n_reads, res.n_split, res.n_clusters
    d = {"split_off": ctx.to_host(res.d_split_off, n + 1, np.uint64), "cluster_val": ctx.to_host(res.d_cluster_val, nc, np.int32),
         "cluster_split": ctx.to_host(res.d_cluster_split, nc, np.uint8)}
    for k, dt in (("qs", np.uint32), ("qe", np.uint32), ("ts", np.uint32), ("te", np.uint32), ("strand", np.int32), ("coarse", np.int32),
                  ("val", np.int32), ("num_anchors", np.int32), ("read", np.uint32)):
        d[k] = ctx.to_host(getattr(res, "d_" + k), ns, dt)
    return d


class RscOpts(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("window", "smallK", "K", "limitrefine", "max_freq", "local_window")]


class RefinedResult(C.Structure):
    _fields_ = [("n_frags", C.c_uint64), ("n_tasks", C.c_uint64), ("n_pairs", C.c_uint64), ("n_matches", C.c_uint64)] + [
        (n, C.c_void_p) for n in ("d_match_off", "d_match_q", "d_match_t", "d_box", "d_eff", "d_status", "d_task_q_lo", "d_task_q_hi", "d_task_t_lo",
                                  "d_task_t_hi")]


def refine_splitchain_batch(ctx: Context, chains: ChainResult, split: SplitResult, read_off, chrom_pos, read_index, g_seq_off, g_index,
                            window=100, smallK=10, K=17, limitrefine=True, max_freq=15):
    """Refine_splitchain (ChainRefine.h:384) for every split chain.  read_index: local.LocalIndex over the reads forward then reverse
    complemented; g_index: local.LocalIndex of the genome, g_seq_off its seqOffsets as a device int64 tensor."""
    cp = np.ascontiguousarray(chrom_pos, dtype=np.uint64)
    o = RscOpts(int(window), int(smallK), int(K), 1 if limitrefine else 0, int(max_freq), int(g_index.window))
    res = RefinedResult()
    ctx.check(ctx.lib.lra_refine_splitchain_batch(ctx.h, C.byref(chains), C.byref(split), ptr(read_off), C.c_void_p(cp.ctypes.data), len(cp) - 1,
                                                  C.byref(read_index.res), C.c_uint64(g_index.n_windows), ptr(g_seq_off),
                                                  C.c_void_p(g_index.res.d_tuple_bnd), C.c_void_p(g_index.res.d_tuples), C.byref(o), C.byref(res)))
    return res


def fetch_refined(ctx: Context, res: RefinedResult):
    nf = res.n_frags
    return {"match_off": ctx.to_host(res.d_match_off, nf + 1, np.uint64), "match_q": ctx.to_host(res.d_match_q, res.n_matches, np.uint32),
            "match_t": ctx.to_host(res.d_match_t, res.n_matches, np.uint32), "box": ctx.to_host(res.d_box, 4 * nf, np.uint32).reshape(-1, 4),
            "eff": ctx.to_host(res.d_eff, nf, np.float32), "status": ctx.to_host(res.d_status, nf, np.uint32)}


class BtwnOpts(C.Structure):
    _fields_ = [("K", C.c_int32), ("W", C.c_int32), ("refineSpaceDist", C.c_int32), ("anchorstoosparse", C.c_float), ("match", C.c_int32),
                ("mismatch", C.c_int32), ("indel", C.c_int32), ("max_freq", C.c_int32)]


class BtwnResult(C.Structure):
    _fields_ = [("n_frags", C.c_uint64), ("n_matches", C.c_uint64), ("n_problems", C.c_uint64), ("n_pairs", C.c_uint64), ("n_rounds", C.c_uint32)] + [
        (n, C.c_void_p) for n in ("d_match_off", "d_match_q", "d_match_t", "d_box", "d_eff", "d_refinespace")]


def refine_btwn_splitchain_batch(ctx: Context, chains: ChainResult, split: SplitResult, refined: RefinedResult, read_off, strands, rc_base, genome,
                                 chrom_pos, K=10, W=5, refineSpaceDist=10000, anchorstoosparse=0.01, match=4, mismatch=-1, indel=-2, max_freq=15):
    """Refine_Btwnsplitchain (ChainRefine.h:579) for every chain; strands = reads forward then reverse complemented (device uint8)."""
    cp = np.ascontiguousarray(chrom_pos, dtype=np.uint64)
    o = BtwnOpts(int(K), int(W), int(refineSpaceDist), float(anchorstoosparse), int(match), int(mismatch), int(indel), int(max_freq))
    res = BtwnResult()
    ctx.check(ctx.lib.lra_refine_btwn_splitchain_batch(ctx.h, C.byref(chains), C.byref(split), C.byref(refined), ptr(read_off), ptr(strands),
                                                       C.c_uint64(int(rc_base)), ptr(genome), C.c_void_p(cp.ctypes.data), len(cp) - 1, C.byref(o),
                                                       C.byref(res)))
    return res


def fetch_btwn(ctx: Context, res: BtwnResult):
    nf = res.n_frags
    return {"match_off": ctx.to_host(res.d_match_off, nf + 1, np.uint64), "match_q": ctx.to_host(res.d_match_q, res.n_matches, np.uint32),
            "match_t": ctx.to_host(res.d_match_t, res.n_matches, np.uint32), "box": ctx.to_host(res.d_box, 4 * nf, np.uint32).reshape(-1, 4),
            "eff": ctx.to_host(res.d_eff, nf, np.float32), "refinespace": ctx.to_host(res.d_refinespace, nf, np.uint8)}


class MergeResult(C.Structure):
    _fields_ = [("n_slots", C.c_uint64), ("n_groups", C.c_uint64), ("n_anchors", C.c_uint64)] + [(n, C.c_void_p) for n in (
        "d_slot_group_off", "d_cluster_base", "d_group_slot", "d_group_first", "d_group_last", "d_anchor_off", "d_count", "d_q", "d_t", "d_len", "d_box",
        "d_strand", "d_chrom", "d_iota")]


def merge_extend_batch(ctx: Context, chains: ChainResult, split: SplitResult, refined: BtwnResult, seq, read_off, genome, chrom_pos, K=10):
    """MergeChain + LinearExtend + DecideCoordinates + TrimOverlappedAnchors (Map_lowacc.h:440-476) for every chain."""
    cp = np.ascontiguousarray(chrom_pos, dtype=np.uint64)
    res = MergeResult()
    ctx.check(ctx.lib.lra_merge_extend_batch(ctx.h, C.byref(chains), C.byref(split), C.byref(refined), ptr(seq), ptr(read_off), ptr(genome),
                                             C.c_void_p(cp.ctypes.data), len(cp) - 1, int(K), C.byref(res)))
    return res


def fetch_merge(ctx: Context, res: MergeResult):
    ns, ng, na = res.n_slots, res.n_groups, res.n_anchors
    return {"slot_group_off": ctx.to_host(res.d_slot_group_off, ns + 1, np.uint64), "cluster_base": ctx.to_host(res.d_cluster_base, ns + 1, np.uint64),
            "group_first": ctx.to_host(res.d_group_first, ng, np.uint32), "group_last": ctx.to_host(res.d_group_last, ng, np.uint32),
            "anchor_off": ctx.to_host(res.d_anchor_off, ng + 1, np.uint64), "count": ctx.to_host(res.d_count, ng, np.uint32),
            "q": ctx.to_host(res.d_q, na, np.uint32), "t": ctx.to_host(res.d_t, na, np.uint32), "len": ctx.to_host(res.d_len, na, np.int32),
            "box": ctx.to_host(res.d_box, 4 * ng, np.uint32).reshape(-1, 4), "strand": ctx.to_host(res.d_strand, ng, np.int32),
            "chrom": ctx.to_host(res.d_chrom, ng, np.int32)}


class RefinedClustersResult(C.Structure):
    _fields_ = [("n_clusters", C.c_uint64), ("n_tasks", C.c_uint64), ("n_pairs", C.c_uint64), ("n_matches", C.c_uint64)] + [
        (n, C.c_void_p) for n in ("d_match_off", "d_match_q", "d_match_t", "d_box", "d_eff", "d_status", "d_chrom")]


def refine_clusters_batch(ctx: Context, n_reads, cluster_off, c_start, c_count, c_strand, qs, qe, ts, te, q, t, n_matches_cap, read_off, chrom_pos, read_index,
                          g_seq_off, g_index, window=100, smallK=10, K=17, max_freq=15):
    """REFINEclusters (ClusterRefine.h:50) for every cluster; array arguments are device tensors or raw device addresses."""
    cp = np.ascontiguousarray(chrom_pos, dtype=np.uint64)
    o = RscOpts(int(window), int(smallK), int(K), 0, int(max_freq), int(g_index.window))
    res = RefinedClustersResult()
    ctx.check(ctx.lib.lra_refine_clusters_batch(ctx.h, int(n_reads), ptr(cluster_off), ptr(c_start), ptr(c_count), ptr(c_strand), ptr(qs), ptr(qe), ptr(ts), ptr(te),
                                                ptr(q), ptr(t), C.c_uint64(int(n_matches_cap)), ptr(read_off), C.c_void_p(cp.ctypes.data), len(cp) - 1,
                                                C.byref(read_index.res), C.c_uint64(g_index.n_windows), ptr(g_seq_off), C.c_void_p(g_index.res.d_tuple_bnd),
                                                C.c_void_p(g_index.res.d_tuples), C.byref(o), C.byref(res)))
    return res


def fetch_refined_clusters(ctx: Context, res: RefinedClustersResult):
    nc = res.n_clusters
    return {"match_off": ctx.to_host(res.d_match_off, nc + 1, np.uint64), "match_q": ctx.to_host(res.d_match_q, res.n_matches, np.uint32),
            "match_t": ctx.to_host(res.d_match_t, res.n_matches, np.uint32), "box": ctx.to_host(res.d_box, 4 * nc, np.uint32).reshape(-1, 4),
            "eff": ctx.to_host(res.d_eff, nc, np.float32), "status": ctx.to_host(res.d_status, nc, np.uint32), "chrom": ctx.to_host(res.d_chrom, nc, np.int32)}


class LraOpts(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("localW", "globalW", "localMaxFreq", "match", "mismatch", "indel", "localBand", "refineBySDP", "isOnt")] + [
        ("gapopen", C.c_float), ("gapextend", C.c_float), ("gaproot", C.c_float), ("gapCeiling1", C.c_int32), ("gapCeiling2", C.c_int32)]


ONT_LRA = dict(localW=5, globalW=5, localMaxFreq=15, match=4, mismatch=-1, indel=-2, localBand=15, refineBySDP=1, isOnt=1, gapopen=7.0, gapextend=10.0, gaproot=1.5,
               gapCeiling1=1500, gapCeiling2=3000)


class AlignmentsResult(C.Structure):
    _fields_ = [("n_jobs", C.c_uint64), ("n_alignments", C.c_uint64), ("n_blocks", C.c_uint64), ("n_big", C.c_uint64), ("n_inner_jobs", C.c_uint64)] + [
        (n, C.c_void_p) for n in ("d_job_aln_off", "d_strand", "d_supp", "d_secondary", "d_n0", "d_n1", "d_chrom", "d_value", "d_block_off", "d_blocks", "d_status")]


def local_refine_batch(ctx: Context, job_chain_off, job_read, job_h, chain_anchor_off, chain_strand, chain_chrom, chain_value, chain_n0, chain_n1, q, t, length,
                       read_off, strands, rc_base, genome, chrom_pos, **kw):
    """LocalRefineAlignment (LocalRefineAlignment.h:885) for every primary chain; array arguments are device tensors."""
    cp = np.ascontiguousarray(chrom_pos, dtype=np.uint64)
    d = dict(ONT_LRA); d.update(kw)
    o = LraOpts(*[d[n] for n, _ in LraOpts._fields_])
    res = AlignmentsResult()
    nj, nc, na = int(job_read.numel()), int(chain_strand.numel()), int(q.numel())
    ctx.check(ctx.lib.lra_local_refine_batch(ctx.h, C.c_uint64(nj), ptr(job_chain_off), ptr(job_read), ptr(job_h), C.c_uint64(nc), ptr(chain_anchor_off), ptr(chain_strand),
                                             ptr(chain_chrom), ptr(chain_value), ptr(chain_n0), ptr(chain_n1), C.c_uint64(na), ptr(q), ptr(t), ptr(length), ptr(read_off),
                                             ptr(strands), C.c_uint64(int(rc_base)), ptr(genome), C.c_void_p(cp.ctypes.data), len(cp) - 1, C.byref(o), C.byref(res)))
    return res


def fetch_alignments(ctx: Context, res: AlignmentsResult):
    nj, na, nb = res.n_jobs, res.n_alignments, res.n_blocks
    d = {"job_aln_off": ctx.to_host(res.d_job_aln_off, nj + 1, np.uint64), "block_off": ctx.to_host(res.d_block_off, na + 1, np.uint64),
         "blocks": ctx.to_host(res.d_blocks, 3 * nb, np.int32).reshape(-1, 3), "value": ctx.to_host(res.d_value, na, np.float32),
         "status": ctx.to_host(res.d_status, nj, np.uint32)}
    for k in ("strand", "supp", "secondary", "n0", "n1", "chrom"):
        d[k] = ctx.to_host(getattr(res, "d_" + k), na, np.int32)
    return d


class LocalRefineInputs(C.Structure):
    _fields_ = [("n_jobs", C.c_uint64), ("n_chains", C.c_uint64), ("n_anchors", C.c_uint64)] + [(n, C.c_void_p) for n in (
        "d_job_chain_off", "d_job_read", "d_job_h", "d_chain_anchor_off", "d_chain_strand", "d_chain_chrom", "d_chain_value", "d_chain_n0", "d_chain_n1", "d_q", "d_t", "d_len")]


def local_refine_from_sdp(ctx: Context, num_aln, slot_n0, merge: MergeResult, second: ChainResult, read_off, strands, rc_base, genome, chrom_pos, **kw):
    """Filters of the second sparse DP + LocalRefineAlignment for every primary chain, without leaving the device (Map_lowacc.h:530-576)."""
    inp = LocalRefineInputs()
    ctx.check(ctx.lib.lra_local_refine_inputs_batch(ctx.h, int(num_aln), ptr(slot_n0) if slot_n0 is not None else None, C.byref(merge), C.byref(second), C.byref(inp)))
    cp = np.ascontiguousarray(chrom_pos, dtype=np.uint64)
    d = dict(ONT_LRA); d.update(kw)
    o = LraOpts(*[d[n] for n, _ in LraOpts._fields_])
    res = AlignmentsResult()
    v = C.c_void_p
    ctx.check(ctx.lib.lra_local_refine_batch(ctx.h, C.c_uint64(inp.n_jobs), v(inp.d_job_chain_off), v(inp.d_job_read), v(inp.d_job_h), C.c_uint64(inp.n_chains),
                                             v(inp.d_chain_anchor_off), v(inp.d_chain_strand), v(inp.d_chain_chrom), v(inp.d_chain_value), v(inp.d_chain_n0),
                                             v(inp.d_chain_n1), C.c_uint64(inp.n_anchors), v(inp.d_q), v(inp.d_t), v(inp.d_len), ptr(read_off), ptr(strands),
                                             C.c_uint64(int(rc_base)), ptr(genome), C.c_void_p(cp.ctypes.data), len(cp) - 1, C.byref(o), C.byref(res)))
    return inp, res


class SameDiagResult(C.Structure):
    _fields_ = [("n_clusters", C.c_uint64), ("n_groups", C.c_uint64)] + [(n, C.c_void_p) for n in ("d_group_off", "d_start", "d_end", "d_status")]


def merge_same_diag_batch(ctx: Context, anchor_off, q, t, length, overlap, strand, merge_dist=100):
    """MergeMatchesSameDiag (LinearExtend.h:795, high-accuracy path) for a batch of extended clusters; device tensors in."""
    res = SameDiagResult()
    n = int(strand.numel())
    ctx.check(ctx.lib.lra_merge_same_diag_batch(ctx.h, C.c_uint64(n), ptr(anchor_off), ptr(q), ptr(t), ptr(length), ptr(overlap), ptr(strand), int(merge_dist),
                                                C.byref(res)))
    return res


def fetch_same_diag(ctx: Context, res: SameDiagResult):
    n, ng = int(res.n_clusters), int(res.n_groups)
    return {"group_off": ctx.to_host(res.d_group_off, n + 1, np.uint64), "start": ctx.to_host(res.d_start, ng, np.uint32),
            "end": ctx.to_host(res.d_end, ng, np.uint32), "status": ctx.to_host(res.d_status, n, np.uint32)}


class SwitchIndexResult(C.Structure):
    _fields_ = [("n_chains", C.c_uint64)] + [(n, C.c_void_p) for n in ("d_ch", "d_link", "d_n", "d_n_link", "d_status")]


def switchindex_batch(ctx: Context, chain_off, ch, link, n_link, split_base, cluster_base, coarse, cl_qs, cl_qe):
    """switchindex (Mapping_ultility.h:39, high-accuracy path) for a batch of chains; device tensors in."""
    res = SwitchIndexResult()
    n = int(chain_off.numel()) - 1
    ctx.check(ctx.lib.lra_switchindex_batch(ctx.h, C.c_uint64(n), ptr(chain_off), ptr(ch), ptr(link), ptr(n_link), ptr(split_base), ptr(cluster_base), ptr(coarse),
                                            ptr(cl_qs), ptr(cl_qe), C.c_uint64(int(chain_off[-1])), C.byref(res)))
    return res


def fetch_switchindex(ctx: Context, res: SwitchIndexResult, n_total):
    n = int(res.n_chains)
    return {"ch": ctx.to_host(res.d_ch, n_total, np.uint32), "link": ctx.to_host(res.d_link, n_total, np.uint8), "n": ctx.to_host(res.d_n, n, np.uint32),
            "n_link": ctx.to_host(res.d_n_link, n, np.uint32), "status": ctx.to_host(res.d_status, n, np.uint32)}


class OriginalAnchorsResult(C.Structure):
    _fields_ = [("n_chains", C.c_uint64), ("n_anchors", C.c_uint64)] + [(n, C.c_void_p) for n in ("d_chain_off", "d_anchor", "d_cluster")]


def switch_to_original_anchors_batch(ctx: Context, chain_off, elem_cluster, elem_entry, same_diag: SameDiagResult, coarse):
    """SwitchToOriginalAnchors (LocalRefineAlignment.h:187, high-accuracy path) for a batch of chains over Cluster_SameDiag entries."""
    res = OriginalAnchorsResult()
    ctx.check(ctx.lib.lra_switch_to_original_anchors_batch(ctx.h, C.c_uint64(int(chain_off.numel()) - 1), ptr(chain_off), C.c_uint64(int(elem_cluster.numel())),
                                                           ptr(elem_cluster), ptr(elem_entry), C.byref(same_diag), ptr(coarse), C.byref(res)))
    return res


class HSplitResult(C.Structure):
    _fields_ = [("n_jobs", C.c_uint64), ("n_pieces", C.c_uint64), ("n_elems", C.c_uint64)] + [(n, C.c_void_p) for n in (
        "d_job_piece_off", "d_piece_off", "d_sptc", "d_piece_type", "d_piece_strand", "d_piece_box", "d_piece_job", "d_job_lsc")]


def split_chains_highacc_batch(ctx: Context, job_off, strand, chrom, box, link_off, link, splitdist=100000):
    """High-accuracy SPLITChain + MergeSplitchainINS + LargestSplitChain_dist over chains of merged clusters; array arguments are device tensors."""
    res = HSplitResult()
    ctx.check(ctx.lib.lra_split_chains_highacc_batch(ctx.h, C.c_uint64(int(job_off.numel()) - 1), ptr(job_off), C.c_uint64(int(strand.numel())), ptr(strand), ptr(chrom),
                                                     ptr(box), ptr(link_off), ptr(link), int(splitdist), C.byref(res)))
    return res


def fetch_hsplit(ctx: Context, res: HSplitResult):
    nj, npc, ne = res.n_jobs, res.n_pieces, res.n_elems
    return {"job_piece_off": ctx.to_host(res.d_job_piece_off, nj + 1, np.uint64), "piece_off": ctx.to_host(res.d_piece_off, npc + 1, np.uint64),
            "sptc": ctx.to_host(res.d_sptc, ne, np.uint32), "type": ctx.to_host(res.d_piece_type, npc, np.uint8), "strand": ctx.to_host(res.d_piece_strand, npc, np.uint8),
            "box": ctx.to_host(res.d_piece_box, 4 * npc, np.uint32).reshape(-1, 4), "job": ctx.to_host(res.d_piece_job, npc, np.uint32),
            "lsc": ctx.to_host(res.d_job_lsc, nj, np.uint32)}


class GlobalChainResult(C.Structure):
    _fields_ = [("n_sets", C.c_uint64), ("n_fragments", C.c_uint64)] + [(n, C.c_void_p) for n in ("d_score", "d_prev", "d_chain", "d_chain_len")]


def global_chain_batch(ctx: Context, off, xl, yl, xh, yh, score):
    """GlobalChain (GlobalChain.h:85) over fragment sets in CSR; array arguments are device tensors (off int64, the rest int32)."""
    res = GlobalChainResult()
    ctx.check(ctx.lib.lra_global_chain_batch(ctx.h, C.c_uint64(int(off.numel()) - 1), ptr(off), C.c_uint64(int(xl.numel())), ptr(xl), ptr(yl), ptr(xh), ptr(yh), ptr(score),
                                             C.byref(res)))
    return res
