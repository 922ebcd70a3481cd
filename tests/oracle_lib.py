"""ctypes bindings to oracle/liblra_oracle.so (the CPU restatement; TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "all"], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ORACLE_DIR, "liblra_oracle.so")
        build_oracle()  # make is a no-op when up to date
        _LIB = C.CDLL(path)
    return _LIB


def ref_bin(name):
    """Path of a reference-function driver under oracle/_ref (None if not built)."""
    p = os.path.join(ORACLE_DIR, "_ref", name)
    return p if os.path.exists(p) else None


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def affine_one_gap_align(q: bytes, t: bytes, m, mm, indel, k, cap=4096):
    L = lib()
    blocks = np.zeros(3 * cap, dtype=np.int32)
    nb = C.c_int(0)
    st = C.c_int(0)
    L.oracle_affine_one_gap_align.restype = C.c_int
    score = L.oracle_affine_one_gap_align(C.c_char_p(q), len(q), C.c_char_p(t), len(t), m, mm, indel, k,
                                          _p(blocks, C.c_int), cap, C.byref(nb), C.byref(st))
    n = nb.value
    assert n <= cap
    return score, blocks[:3 * n].reshape(n, 3).copy(), st.value


def store_minimizers(seq: bytes, k, w):
    L = lib()
    cap = max(1, len(seq))
    keys = np.zeros(cap, dtype=np.uint64)
    pos = np.zeros(cap, dtype=np.uint32)
    L.oracle_store_minimizers.restype = C.c_long
    n = L.oracle_store_minimizers(C.c_char_p(seq), C.c_uint32(len(seq)), k, w, _p(keys, C.c_uint64), _p(pos, C.c_uint32), C.c_long(cap))
    assert n <= cap
    return keys[:n].copy(), pos[:n].copy()


def sort_minimizers(keys, pos):
    L = lib()
    keys = np.ascontiguousarray(keys, dtype=np.uint64).copy()
    pos = np.ascontiguousarray(pos, dtype=np.uint32).copy()
    L.oracle_sort_minimizers(_p(keys, C.c_uint64), _p(pos, C.c_uint32), C.c_long(len(keys)))
    return keys, pos


def compare_lists(qk, qp, tk, tp, max_freq, max_diag=0, min_diag=0, cap=None):
    L = lib()
    qk = np.ascontiguousarray(qk, dtype=np.uint64); qp = np.ascontiguousarray(qp, dtype=np.uint32)
    tk = np.ascontiguousarray(tk, dtype=np.uint64); tp = np.ascontiguousarray(tp, dtype=np.uint32)
    L.oracle_compare_lists.restype = C.c_long
    args = lambda oq, ot, c: (_p(qk, C.c_uint64), _p(qp, C.c_uint32), C.c_long(len(qk)), _p(tk, C.c_uint64), _p(tp, C.c_uint32),
                              C.c_long(len(tk)), C.c_long(max_freq), C.c_int64(max_diag), C.c_int64(min_diag), oq, ot, C.c_long(c))
    dummy = np.zeros(1, dtype=np.uint32)
    n = L.oracle_compare_lists(*args(_p(dummy, C.c_uint32), _p(dummy, C.c_uint32), 0))
    oq = np.zeros(max(1, n), dtype=np.uint32); ot = np.zeros(max(1, n), dtype=np.uint32)
    n2 = L.oracle_compare_lists(*args(_p(oq, C.c_uint32), _p(ot, C.c_uint32), n))
    assert n2 == n
    return oq[:n], ot[:n]


def separate_strand(read: bytes, genome, k, qpos, tpos):
    L = lib()
    qpos = np.ascontiguousarray(qpos, dtype=np.uint32); tpos = np.ascontiguousarray(tpos, dtype=np.uint32)
    strand = np.zeros(max(1, len(qpos)), dtype=np.uint8)
    g = genome if isinstance(genome, bytes) else genome.tobytes()
    L.oracle_separate_strand.restype = C.c_long
    L.oracle_separate_strand(C.c_char_p(read), C.c_char_p(g), k, _p(qpos, C.c_uint32), _p(tpos, C.c_uint32), C.c_long(len(qpos)), _p(strand, C.c_uint8))
    return strand[:len(qpos)]


def indel_refine(blocks, q_seq: bytes, t_seq: bytes, refine_band, match, mismatch, indel, end_align=False, read_len=None, chrom_len=None):
    """blocks: (n,3) int32 (qPos,tPos,len).  Returns (refined (m,3) int32, status)."""
    L = lib()
    b = np.ascontiguousarray(np.asarray(blocks, dtype=np.int32).reshape(-1, 3))
    n = len(b)
    cap = int(b[:, 2].sum() + 2 * n + 64 + (b[-1, 1] + b[-1, 2] - b[0, 1] if n else 0) + (b[-1, 0] + b[-1, 2] - b[0, 0] if n else 0))
    out = np.zeros(3 * cap, dtype=np.int32)
    st = C.c_int(0)
    L.oracle_indel_refine.restype = C.c_long
    m = L.oracle_indel_refine(_p(b, C.c_int), C.c_long(n), C.c_char_p(q_seq), C.c_long(len(q_seq) if read_len is None else read_len),
                              C.c_char_p(t_seq), C.c_long(len(t_seq) if chrom_len is None else chrom_len), refine_band, match, mismatch,
                              indel, 1 if end_align else 0, _p(out, C.c_int), C.c_long(cap), C.byref(st))
    assert m <= cap, (m, cap)
    return out[:3 * m].reshape(m, 3).copy(), st.value


def antiqsort_keys(n):
    L = lib()
    out = np.zeros(n, dtype=np.uint64)
    L.oracle_antiqsort(n, _p(out, C.c_uint64))
    return out


class CleanOpts(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("globalK", "cleanMaxDiag", "minDiagCluster", "bypassClustering", "cleanClustersize",
                                         "SecondCleanMinDiagCluster", "SecondCleanMaxDiag", "punish_anchorfreq", "anchorPerlength")]


# (lra.cpp:268-431) the fields CleanMatches reads, per preset
CLEAN_PRESETS = {
    "ONT": dict(globalK=17, cleanMaxDiag=200, minDiagCluster=3, bypassClustering=1, cleanClustersize=100,
                SecondCleanMinDiagCluster=10, SecondCleanMaxDiag=100, punish_anchorfreq=5, anchorPerlength=5),
    "CLR": dict(globalK=15, cleanMaxDiag=200, minDiagCluster=3, bypassClustering=1, cleanClustersize=100,
                SecondCleanMinDiagCluster=10, SecondCleanMaxDiag=120, punish_anchorfreq=5, anchorPerlength=5),
    "CCS": dict(globalK=17, cleanMaxDiag=150, minDiagCluster=10, bypassClustering=0, cleanClustersize=100,
                SecondCleanMinDiagCluster=30, SecondCleanMaxDiag=100, punish_anchorfreq=10, anchorPerlength=10),
}


def clean_matches(qpos, tpos, qkey, strand, opts: "CleanOpts", chrom_pos):
    """One strand of one read.  Returns (out_q, out_t, clusters dict of arrays)."""
    L = lib()
    n = len(qpos)
    qpos = np.ascontiguousarray(qpos, dtype=np.uint32); tpos = np.ascontiguousarray(tpos, dtype=np.uint32)
    qkey = np.ascontiguousarray(qkey, dtype=np.uint64)
    cp = np.ascontiguousarray(chrom_pos, dtype=np.uint64)
    cap = max(1, n)
    oq = np.zeros(cap, np.uint32); ot = np.zeros(cap, np.uint32)
    cs = np.zeros(cap, np.int64); ce = np.zeros(cap, np.int64)
    qs = np.zeros(cap, np.uint32); qe = np.zeros(cap, np.uint32); ts = np.zeros(cap, np.uint32); te = np.zeros(cap, np.uint32)
    ch = np.zeros(cap, np.int32); fr = np.zeros(cap, np.float32)
    nclean = C.c_long(0)
    L.oracle_clean_matches.restype = C.c_long
    ncl = L.oracle_clean_matches(_p(qpos, C.c_uint32), _p(tpos, C.c_uint32), _p(qkey, C.c_uint64), C.c_long(n), int(strand), C.byref(opts),
                                 _p(cp, C.c_uint64), len(cp) - 1, _p(oq, C.c_uint32), _p(ot, C.c_uint32), C.byref(nclean),
                                 _p(cs, C.c_long), _p(ce, C.c_long), _p(qs, C.c_uint32), _p(qe, C.c_uint32), _p(ts, C.c_uint32),
                                 _p(te, C.c_uint32), _p(ch, C.c_int), _p(fr, C.c_float))
    k = nclean.value
    return oq[:k].copy(), ot[:k].copy(), dict(start=cs[:ncl].copy(), end=ce[:ncl].copy(), qStart=qs[:ncl].copy(), qEnd=qe[:ncl].copy(),
                                              tStart=ts[:ncl].copy(), tEnd=te[:ncl].copy(), chrom=ch[:ncl].copy(), freq=fr[:ncl].copy())


def linear_extend_cluster(q, t, strand, box, prev_box, next_box, anchorfreq, read: bytes, chrom: bytes, K=17, skiprepetitive=True, trim=False):
    """One chain element of the cluster version of LinearExtend (LinearExtend.h:136-352); t relative to the chromosome.
    -> dict(q, t, len, overlap, box, sorted_q, sorted_t)"""
    L = lib()
    q = np.ascontiguousarray(q, np.uint32).copy(); t = np.ascontiguousarray(t, np.uint32).copy()
    n = len(q)
    bx = np.ascontiguousarray(box, np.uint32)
    pb = np.ascontiguousarray(prev_box, np.uint32) if prev_box is not None else None
    nb = np.ascontiguousarray(next_box, np.uint32) if next_box is not None else None
    eq = np.zeros(max(1, n), np.uint32); et = np.zeros(max(1, n), np.uint32); el = np.zeros(max(1, n), np.int32); eo = np.zeros(max(1, n), np.uint8)
    ob = np.zeros(4, np.uint32); nov = C.c_int(0)
    L.oracle_linear_extend_cluster.restype = C.c_long
    ne = L.oracle_linear_extend_cluster(C.c_long(n), _p(q, C.c_uint32), _p(t, C.c_uint32), int(strand), _p(bx, C.c_uint32), _p(pb, C.c_uint32) if pb is not None else None,
                                        _p(nb, C.c_uint32) if nb is not None else None, C.c_float(float(anchorfreq)), 1 if skiprepetitive else 0, int(K), C.c_char_p(read),
                                        C.c_uint32(len(read)), C.c_char_p(chrom), C.c_uint32(len(chrom)), _p(eq, C.c_uint32), _p(et, C.c_uint32), _p(el, C.c_int),
                                        _p(eo, C.c_uint8), _p(ob, C.c_uint32), C.byref(nov))
    eq, et, el, eo = eq[:ne].copy(), et[:ne].copy(), el[:ne].copy(), eo[:ne].copy()
    if trim and ne:
        L.oracle_trim_overlapped_anchors(C.c_int(ne), _p(eq, C.c_uint32), _p(et, C.c_uint32), _p(el, C.c_int), C.c_int(int(strand)))
    return dict(q=eq, t=et, len=el, overlap=eo, box=ob, sorted_q=q, sorted_t=t, n_overlap=nov.value)


def refine_btwn_clusters_chains(match_off, mq, mt, box, strand, chrom, freq, chain_off, ch, fwd: bytes, rc: bytes, genome: bytes, chrom_pos, K=17, W=10, read_type=2,
                                anchorstoosparse=0.005, match=4, mismatch=-3, indel=-4, max_freq=15):
    """RefineBtwnClusters_chain over all chains of one read (ClusterRefine.h:433) -> dict(off, q, t, box, freq, refinespace)"""
    L = lib()
    mo = np.ascontiguousarray(match_off, np.int32); mq = np.ascontiguousarray(mq, np.uint32); mt = np.ascontiguousarray(mt, np.uint32)
    bx = np.ascontiguousarray(box, np.uint32).reshape(-1).copy(); st = np.ascontiguousarray(strand, np.uint8); chv = np.ascontiguousarray(chrom, np.int32)
    fr = np.ascontiguousarray(freq, np.float32).copy(); co = np.ascontiguousarray(chain_off, np.int32); chn = np.ascontiguousarray(ch if len(ch) else [0], np.int32)
    pos = np.ascontiguousarray(chrom_pos, np.uint64)
    ncl = len(st)
    rs = np.zeros(max(1, ncl), np.uint8); oo = np.zeros(ncl + 1, np.int32)
    cap = len(mq) + 8 * len(fwd) + 1024
    oq = np.zeros(cap, np.uint32); ot = np.zeros(cap, np.uint32)
    L.oracle_refine_btwn_clusters_chains.restype = C.c_long
    n = L.oracle_refine_btwn_clusters_chains(C.c_int(ncl), _p(mo, C.c_int), _p(mq, C.c_uint32), _p(mt, C.c_uint32), _p(bx, C.c_uint32), _p(st, C.c_uint8), _p(chv, C.c_int),
                                             _p(fr, C.c_float), _p(rs, C.c_uint8), C.c_int(len(co) - 1), _p(co, C.c_int), _p(chn, C.c_int), int(K), int(W), int(read_type),
                                             C.c_float(anchorstoosparse), int(match), int(mismatch), int(indel), C.c_long(max_freq), C.c_char_p(fwd), C.c_char_p(rc),
                                             C.c_uint32(len(fwd)), C.c_char_p(genome), _p(pos, C.c_uint64), C.c_long(cap), _p(oo, C.c_int), _p(oq, C.c_uint32), _p(ot, C.c_uint32))
    assert 0 <= n <= cap
    return dict(off=oo, q=oq[:n].copy(), t=ot[:n].copy(), box=bx.reshape(-1, 4), freq=fr, refinespace=rs[:ncl])


def split_chain_highacc(strand, chrom, box, link, splitdist=100000):
    """High-accuracy SPLITChain + MergeSplitchainINS + LargestSplitChain_dist (Mapping_ultility.h:266-346) -> dict(off, idx, type, strand, box, lsc)"""
    L = lib()
    n = len(strand)
    st = np.ascontiguousarray(strand, np.uint8); ch = np.ascontiguousarray(chrom, np.int32); bx = np.ascontiguousarray(box, np.uint32).reshape(-1)
    lk = np.ascontiguousarray(np.concatenate([np.asarray(link, np.uint8), np.zeros(1, np.uint8)]))
    off = np.zeros(n + 2, np.int32); idx = np.zeros(n + 1, np.int32); ty = np.zeros(n + 1, np.uint8); ss = np.zeros(n + 1, np.uint8); ob = np.zeros(4 * n + 4, np.uint32)
    lsc = C.c_int(0)
    k = L.oracle_split_chain_highacc(n, _p(st, C.c_uint8), _p(ch, C.c_int), _p(bx, C.c_uint32), _p(lk, C.c_uint8), int(splitdist), _p(off, C.c_int), _p(idx, C.c_int),
                                     ty.ctypes.data_as(C.c_char_p), _p(ss, C.c_uint8), _p(ob, C.c_uint32), C.byref(lsc))
    return dict(off=off[:k + 1], idx=idx[:off[k]], type=ty[:k], strand=ss[:k], box=ob[:4 * k].reshape(-1, 4), lsc=lsc.value)


def global_chain(fragments, score=None):
    """GlobalChain (GlobalChain.h:85-189) on one fragment set [[xl, yl, xh, yh], ...] -> (chain, score, prev); score defaults to xh - xl (TestGlobalChain.cpp:14)"""
    L = lib()
    f = np.ascontiguousarray(np.asarray(fragments, np.int32).reshape(-1, 4))
    n = len(f)
    xl, yl, xh, yh = (np.ascontiguousarray(f[:, k]) for k in range(4))
    sc = np.ascontiguousarray(xh - xl if score is None else score, np.int32).copy()
    pv = np.zeros(max(1, n), np.int32); ch = np.zeros(max(1, n), np.int32)
    k = L.oracle_global_chain(C.c_int(n), _p(xl, C.c_int), _p(yl, C.c_int), _p(xh, C.c_int), _p(yh, C.c_int), _p(sc, C.c_int), _p(pv, C.c_int), _p(ch, C.c_int))
    return ch[:k].tolist(), sc[:n].tolist(), pv[:n].tolist()


class FineOpts(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("globalK", "RoughClustermaxGap", "maxDiag", "maxGap", "minClusterSize", "minUniqueStretchNum", "minUniqueStretchDist")]


def matches_to_fine_clusters(qpos, tpos, qkey, n_forward, clean_opts: "CleanOpts", fine_opts: "FineOpts", chrom_pos):
    """Both strands of one read (forward matches first) -> (dict(off, q, t, box, strand, chrom, freq), status)"""
    L = lib()
    qpos = np.ascontiguousarray(qpos, np.uint32); tpos = np.ascontiguousarray(tpos, np.uint32); qkey = np.ascontiguousarray(qkey, np.uint64)
    cp = np.ascontiguousarray(chrom_pos, np.uint64)
    n = len(qpos); cap = max(1, n)
    oq = np.zeros(cap, np.uint32); ot = np.zeros(cap, np.uint32); off = np.zeros(cap + 1, np.int64); box = np.zeros(4 * cap, np.uint32)
    st = np.zeros(cap, np.int32); ch = np.zeros(cap, np.int32); fr = np.zeros(cap, np.float32); nm = C.c_long(0); status = C.c_int(0)
    L.oracle_matches_to_fine_clusters.restype = C.c_long
    nc = L.oracle_matches_to_fine_clusters(_p(qpos, C.c_uint32), _p(tpos, C.c_uint32), _p(qkey, C.c_uint64), C.c_long(n), C.c_long(int(n_forward)), C.byref(clean_opts),
                                           C.byref(fine_opts), _p(cp, C.c_uint64), len(cp) - 1, C.c_long(cap), C.c_long(cap), _p(oq, C.c_uint32), _p(ot, C.c_uint32),
                                           _p(off, C.c_long), _p(box, C.c_uint32), _p(st, C.c_int), _p(ch, C.c_int), _p(fr, C.c_float), C.byref(nm), C.byref(status))
    assert nc >= 0
    m = nm.value
    return dict(off=off[:nc + 1].copy(), q=oq[:m].copy(), t=ot[:m].copy(), box=box.reshape(-1, 4)[:nc].copy(), strand=st[:nc].copy(), chrom=ch[:nc].copy(), freq=fr[:nc].copy()), status.value


def store_index(genome: bytes, chrom_pos, k=17, w=10, max_freq=150, winsize=15, n_per_window=1, stable=False):
    """StoreIndex (MMIndex.h:286-400) -> (key uint64[], pos uint32[], status).  stable: equal keys keep their emission order (the device builder's
    order); False = the reference's std::sort."""
    L = lib()
    cp = np.ascontiguousarray(chrom_pos, dtype=np.uint64)
    L.oracle_store_index.restype = C.c_long
    st = C.c_int(0)
    args = (C.c_char_p(genome), _p(cp, C.c_uint64), len(cp) - 1, int(k), int(w), int(max_freq), int(winsize), int(n_per_window), 1 if stable else 0)
    n = L.oracle_store_index(*args, None, None, C.c_long(0), C.byref(st))
    key = np.zeros(max(1, n), np.uint64); pos = np.zeros(max(1, n), np.uint32)
    n2 = L.oracle_store_index(*args, _p(key, C.c_uint64), _p(pos, C.c_uint32), C.c_long(n), C.byref(st))
    assert n2 == n
    return key[:n].copy(), pos[:n].copy(), st.value


def linear_extend(q, t, strand, K, read: bytes, chrom: bytes):
    """Pair-version LinearExtend + DecideCoordinates for one cluster (t chromosome-relative)."""
    L = lib()
    q = np.ascontiguousarray(q, dtype=np.uint32); t = np.ascontiguousarray(t, dtype=np.uint32)
    n = len(q)
    eq = np.zeros(max(1, n), np.uint32); et = np.zeros(max(1, n), np.uint32); el = np.zeros(max(1, n), np.int32); box = np.zeros(4, np.uint32)
    L.oracle_linear_extend.restype = C.c_long
    ne = L.oracle_linear_extend(_p(q, C.c_uint32), _p(t, C.c_uint32), C.c_long(n), int(strand), int(K), C.c_char_p(read), C.c_uint32(len(read)),
                                C.c_char_p(chrom), C.c_uint32(len(chrom)), _p(eq, C.c_uint32), _p(et, C.c_uint32), _p(el, C.c_int), _p(box, C.c_uint32))
    return eq[:ne].copy(), et[:ne].copy(), el[:ne].copy(), box


_LUT = None


def log_lookup_table():
    """LookUpTable of the reference (LogLookUpTable.h:9-15): logf(i) for i = 1, 6, ..., 10001, float32 (host libm)."""
    global _LUT
    if _LUT is None:
        import math
        import struct
        vals = []
        libm = C.CDLL("libm.so.6")
        libm.logf.restype = C.c_float
        libm.logf.argtypes = [C.c_float]
        for i in range(1, 10002, 5):
            vals.append(libm.logf(float(i)))
        _LUT = np.array(vals, dtype=np.float32)
    return _LUT


STAT_NAMES = ["nm", "nmm", "nins", "ndel", "tdel", "tins", "nSmallDel", "nMedDel", "nLargeDel", "nSmallIns", "nMedIns", "nLargeIns",
              "preClip", "sufClip", "qStart", "qEnd", "tStart", "tEnd"]


def calculate_statistics(blocks, read: bytes, genome: bytes):
    """Returns (counts dict, value float32, runs uint32 array (len<<4|op), cigar string)."""
    L = lib()
    b = np.ascontiguousarray(np.asarray(blocks, dtype=np.int32).reshape(-1, 3))
    lut = log_lookup_table()
    counts = np.zeros(18, dtype=np.int64)
    val = C.c_float(0)
    cap = int(b[:, 2].sum() * 2 + 4 * len(b) + 16) if len(b) else 1
    runs = np.zeros(cap, dtype=np.uint32)
    L.oracle_calculate_statistics.restype = C.c_long
    n = L.oracle_calculate_statistics(_p(b, C.c_int), C.c_long(len(b)), C.c_char_p(read), C.c_long(len(read)), C.c_char_p(genome),
                                      _p(lut, C.c_float), _p(counts, C.c_long), C.byref(val), _p(runs, C.c_uint32), C.c_long(cap))
    assert n <= cap
    runs = runs[:n].copy()
    cigar = "".join("%d%s" % (r >> 4, "=XID"[r & 15]) for r in runs)
    return dict(zip(STAT_NAMES, counts.tolist())), np.float32(val.value), runs, cigar


def pack_local(t, pos):
    """LocalTuple words  t | pos << 20  (TupleOps.h:20-25)."""
    return (np.asarray(t, dtype=np.uint32) & np.uint32(0xFFFFF)) | (np.asarray(pos, dtype=np.uint32) << np.uint32(20))


def store_minimizers_noncanonical(seq: bytes, k, w):
    L = lib()
    cap = max(1, len(seq))
    out = np.zeros(cap, dtype=np.uint32)
    L.oracle_store_minimizers_noncanonical.restype = C.c_long
    n = L.oracle_store_minimizers_noncanonical(C.c_char_p(seq), C.c_uint32(len(seq)), k, w, _p(out, C.c_uint32), C.c_long(cap))
    return out[:n].copy()


def local_index_seq(seq: bytes, k, w, window, max_freq):
    """LocalIndex::IndexSeq: (tuples uint32, boundaries uint64[nWindows+1])."""
    L = lib()
    nwin = (len(seq) + window - 1) // window
    cap = max(1, len(seq))
    tup = np.zeros(cap, dtype=np.uint32)
    bnd = np.zeros(nwin + 1, dtype=np.uint64)
    L.oracle_local_index_seq.restype = C.c_long
    n = L.oracle_local_index_seq(C.c_char_p(seq), C.c_long(len(seq)), k, w, window, max_freq, _p(tup, C.c_uint32), C.c_long(cap), _p(bnd, C.c_uint64))
    assert n == nwin
    return tup[:int(bnd[-1])].copy(), bnd


def compare_lists_local(q, t, max_freq, max_diag=0, min_diag=0):
    L = lib()
    q = np.ascontiguousarray(q, dtype=np.uint32); t = np.ascontiguousarray(t, dtype=np.uint32)
    cap = max(1, 3 * len(q) * max(1, len(t)))
    oq = np.zeros(cap, np.uint32); ot = np.zeros(cap, np.uint32)
    L.oracle_compare_lists_local.restype = C.c_long
    n = L.oracle_compare_lists_local(_p(q, C.c_uint32), C.c_long(len(q)), _p(t, C.c_uint32), C.c_long(len(t)), C.c_long(max_freq),
                                     C.c_int64(max_diag), C.c_int64(min_diag), _p(oq, C.c_uint32), _p(ot, C.c_uint32), C.c_long(cap))
    assert n <= cap
    return oq[:n].copy(), ot[:n].copy()


# ---- sparse DP (a8) -----------------------------------------------------------------------------------------------
class SdpOpts(C.Structure):
    _fields_ = [("rate", C.c_float), ("NumAln", C.c_int), ("alnthres", C.c_float), ("readLen", C.c_int),
                ("gapopen", C.c_float), ("gapextend", C.c_float), ("gaproot", C.c_float),
                ("gapCeiling1", C.c_int), ("gapCeiling2", C.c_int), ("mode", C.c_int), ("globalK", C.c_int)]


# -ONT preset (lra.cpp:388-420) + Options.h defaults
SDP_ONT = dict(rate=20.0, NumAln=2, alnthres=0.65, gapopen=7.0, gapextend=10.0, gaproot=1.5, gapCeiling1=1500, gapCeiling2=3000, mode=0,
               globalK=17)


def sdp_opts(read_len, **kw):
    d = dict(SDP_ONT); d.update(kw)
    return SdpOpts(d["rate"], d["NumAln"], d["alnthres"], int(read_len), d["gapopen"], d["gapextend"], d["gaproot"],
                   d["gapCeiling1"], d["gapCeiling2"], d["mode"], d["globalK"])


def sdp_divide_dump(q, t, ind, inv, want_text=False):
    """(length, fnv1a hash[, text]) of the canonical decomposition text."""
    L = lib()
    q = np.ascontiguousarray(q, np.uint32); t = np.ascontiguousarray(t, np.uint32)
    ind = np.ascontiguousarray(ind, np.uint8); inv = np.ascontiguousarray(inv, np.uint8)
    h = C.c_uint64(0)
    L.oracle_sdp_divide_dump.restype = C.c_long
    n = L.oracle_sdp_divide_dump(C.c_long(len(q)), _p(q, C.c_uint32), _p(t, C.c_uint32), _p(ind, C.c_uint8), _p(inv, C.c_uint8),
                                 C.byref(h), None, C.c_long(0))
    if not want_text:
        return n, h.value
    buf = C.create_string_buffer(n + 1)
    L.oracle_sdp_divide_dump(C.c_long(len(q)), _p(q, C.c_uint32), _p(t, C.c_uint32), _p(ind, C.c_uint8), _p(inv, C.c_uint8),
                             C.byref(h), buf, C.c_long(n + 1))
    return n, h.value, buf.value.decode()


def sdp_pwl(params, xs):
    """params = (intercept, scalar, root, g1, g2) -> (pwl bits, w bits, slope bits[25], inter bits[25])"""
    L = lib()
    xs = np.ascontiguousarray(xs, np.int64)
    a = np.zeros(len(xs), np.float32); b = np.zeros(len(xs), np.float32)
    s = np.zeros(25, np.float32); i = np.zeros(25, np.float32)
    L.oracle_sdp_pwl(C.c_float(params[0]), C.c_float(params[1]), C.c_float(params[2]), C.c_int(params[3]), C.c_int(params[4]),
                     C.c_long(len(xs)), _p(xs, C.c_long), _p(a, C.c_float), _p(b, C.c_float), _p(s, C.c_float), _p(i, C.c_float))
    return a.view(np.uint32), b.view(np.uint32), s.view(np.uint32), i.view(np.uint32)


def sdp_maximization_script(params, Di, Ei, ops):
    """ops: list of (op, a, value_bits).  -> (outputs, final Block pairs)"""
    L = lib()
    Di = np.ascontiguousarray(Di, np.int64); Ei = np.ascontiguousarray(Ei, np.int64)
    op = np.ascontiguousarray([o[0] for o in ops], np.int32)
    a = np.ascontiguousarray([o[1] for o in ops], np.int64)
    v = np.ascontiguousarray([o[2] for o in ops], np.uint32).view(np.float32)
    out = np.zeros(max(1, len(ops)), np.int64)
    blk = np.zeros(2 * (len(Di) + len(ops) + 8) * 2, np.int64)
    nb = C.c_long(0)
    L.oracle_sdp_maximization_script.restype = C.c_long
    n = L.oracle_sdp_maximization_script(C.c_long(len(Di)), _p(Di, C.c_long), C.c_long(len(Ei)), _p(Ei, C.c_long), C.c_long(len(ops)),
                                         _p(op, C.c_int), _p(a, C.c_long), _p(v, C.c_float), C.c_float(params[0]), C.c_float(params[1]),
                                         C.c_float(params[2]), C.c_int(params[3]), C.c_int(params[4]), _p(out, C.c_long), _p(blk, C.c_long),
                                         C.byref(nb))
    return out[:n].copy(), blk[:2 * nb.value].copy()


def sdp_chain(cluster_off, cluster_strand, q, t, length, opts: "SdpOpts"):
    """SDP#A on one read's extended clusters -> dict(val, prev_sub, prev_ind, flags, chains=[dict(frags, link, box, value)])."""
    L = lib()
    off = np.ascontiguousarray(cluster_off, np.int32); st = np.ascontiguousarray(cluster_strand, np.uint8)
    q = np.ascontiguousarray(q, np.uint32); t = np.ascontiguousarray(t, np.uint32); ln = np.ascontiguousarray(length, np.int32)
    n = len(q); nc = len(st)
    assert len(off) == nc + 1 and (nc == 0 or off[-1] == n)
    val = np.zeros(max(1, n), np.float32); ps = np.zeros(max(1, n), np.int64); pi = np.zeros(max(1, n), np.int64)
    fl = np.zeros(max(1, n), np.uint8)
    mc = max(1, opts.NumAln)
    coff = np.zeros(mc + 1, np.int32); cf = np.zeros(max(1, n), np.uint32); cl = np.zeros(max(1, n), np.uint8)
    box = np.zeros(4 * mc, np.uint32); cv = np.zeros(mc, np.float32)
    L.oracle_sdp_chain.restype = C.c_int
    r = L.oracle_sdp_chain(C.c_int(nc), _p(off, C.c_int), _p(st, C.c_uint8), _p(q, C.c_uint32), _p(t, C.c_uint32), _p(ln, C.c_int),
                           C.byref(opts), _p(val, C.c_float), _p(ps, C.c_long), _p(pi, C.c_long), _p(fl, C.c_uint8), C.c_int(mc),
                           _p(coff, C.c_int), _p(cf, C.c_uint32), _p(cl, C.c_uint8), _p(box, C.c_uint32), _p(cv, C.c_float))
    chains = []
    for c in range(max(r, 0)):
        a, b = coff[c], coff[c + 1]
        chains.append(dict(frags=cf[a:b].copy(), link=cl[a:b - 1].copy(), box=box[4 * c:4 * c + 4].copy(), value=float(cv[c])))
    return dict(status=r, val=val[:n], prev_sub=ps[:n], prev_ind=pi[:n], flags=fl[:n], chains=chains)


def sdp_chain_boxes(qs, qe, ts, te, strand, val, num_anchors, opts: "SdpOpts"):
    """The high-accuracy SparseDP (SparseDP.h:1956) on one read's split clusters -> as sdp_chain, chains carry num_anchors."""
    L = lib()
    qs = np.ascontiguousarray(qs, np.uint32); qe = np.ascontiguousarray(qe, np.uint32); ts = np.ascontiguousarray(ts, np.uint32)
    te = np.ascontiguousarray(te, np.uint32); st = np.ascontiguousarray(strand, np.uint8); vl = np.ascontiguousarray(val, np.int32)
    na = np.ascontiguousarray(num_anchors, np.int32)
    n = len(qs)
    val_o = np.zeros(max(1, n), np.float32); ps = np.zeros(max(1, n), np.int64); pi = np.zeros(max(1, n), np.int64)
    fl = np.zeros(max(1, n), np.uint8)
    mc = max(1, opts.NumAln)
    coff = np.zeros(mc + 1, np.int32); cf = np.zeros(max(1, n), np.uint32); cl = np.zeros(max(1, n), np.uint8)
    box = np.zeros(4 * mc, np.uint32); cv = np.zeros(mc, np.float32); cn = np.zeros(mc, np.int32)
    L.oracle_sdp_chain_boxes.restype = C.c_int
    r = L.oracle_sdp_chain_boxes(C.c_int(n), _p(qs, C.c_uint32), _p(qe, C.c_uint32), _p(ts, C.c_uint32), _p(te, C.c_uint32), _p(st, C.c_uint8),
                                 _p(vl, C.c_int), _p(na, C.c_int), C.byref(opts), _p(val_o, C.c_float), _p(ps, C.c_long), _p(pi, C.c_long),
                                 _p(fl, C.c_uint8), C.c_int(mc), _p(coff, C.c_int), _p(cf, C.c_uint32), _p(cl, C.c_uint8), _p(box, C.c_uint32),
                                 _p(cv, C.c_float), _p(cn, C.c_int))
    chains = []
    for c in range(max(r, 0)):
        a, b = coff[c], coff[c + 1]
        chains.append(dict(frags=cf[a:b].copy(), link=cl[a:b - 1].copy(), box=box[4 * c:4 * c + 4].copy(), value=float(cv[c]),
                           num_anchors=int(cn[c])))
    return dict(status=r, val=val_o[:n], prev_sub=ps[:n], prev_ind=pi[:n], flags=fl[:n], chains=chains)


def split_clusters(qs, qe, ts, te, strand, anchorfreq, match_off, match_q, contig=False, K=17):
    """SplitClusters + DecideSplitClustersValue (SplitClusters.h:63,176) on one read -> dict(cluster_val, cluster_split, qs, qe, ts, te,
    strand, coarse, val, num)."""
    L = lib()
    qs = np.ascontiguousarray(qs, np.uint32); qe = np.ascontiguousarray(qe, np.uint32); ts = np.ascontiguousarray(ts, np.uint32)
    te = np.ascontiguousarray(te, np.uint32); st = np.ascontiguousarray(strand, np.uint8); af = np.ascontiguousarray(anchorfreq, np.float32)
    mo = np.ascontiguousarray(match_off, np.int32); mq = np.ascontiguousarray(match_q, np.uint32)
    n = len(qs)
    cap = 8 * n * n + 16
    cv = np.zeros(max(1, n), np.int32); cs = np.zeros(max(1, n), np.uint8)
    o = [np.zeros(cap, np.uint32) for _ in range(4)]
    ost = np.zeros(cap, np.uint8); oc = np.zeros(cap, np.int32); ov = np.zeros(cap, np.int32); on = np.zeros(cap, np.int32)
    L.oracle_split_clusters.restype = C.c_int
    r = L.oracle_split_clusters(C.c_int(n), _p(qs, C.c_uint32), _p(qe, C.c_uint32), _p(ts, C.c_uint32), _p(te, C.c_uint32), _p(st, C.c_uint8),
                                _p(af, C.c_float), _p(mo, C.c_int), _p(mq, C.c_uint32), C.c_int(1 if contig else 0), C.c_int(K),
                                _p(cv, C.c_int), _p(cs, C.c_uint8), C.c_int(cap), _p(o[0], C.c_uint32), _p(o[1], C.c_uint32),
                                _p(o[2], C.c_uint32), _p(o[3], C.c_uint32), _p(ost, C.c_uint8), _p(oc, C.c_int), _p(ov, C.c_int), _p(on, C.c_int))
    assert r >= 0, r
    return dict(cluster_val=cv[:n].copy(), cluster_split=cs[:n].copy(), qs=o[0][:r].copy(), qe=o[1][:r].copy(), ts=o[2][:r].copy(),
                te=o[3][:r].copy(), strand=ost[:r].copy(), coarse=oc[:r].copy(), val=ov[:r].copy(), num=on[:r].copy())


class RscOpts(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("window", "smallK", "K", "limitrefine", "maxFreq")]


def refine_splitchain(q, t, length, cluster, cstrand, sptc, box, strand, chrom, ci, chrom_pos, read_len, q_index, g_index,
                      window=100, smallK=10, K=17, limitrefine=True, max_freq=15):
    """Refine_splitchain (ChainRefine.h:384) for one split chain.  q_index / g_index = (seqOffsets, tupleBoundaries, tuples) of the read's
    index on the split chain's strand / of the genome.  -> dict(q, t, box, eff) or None when the reference reads outside an array."""
    L = lib()
    q = np.ascontiguousarray(q, np.uint32); t = np.ascontiguousarray(t, np.uint32); ln = np.ascontiguousarray(length, np.int32)
    cl = np.ascontiguousarray(cluster, np.int32); cs = np.ascontiguousarray(cstrand, np.uint8); sp = np.ascontiguousarray(sptc, np.int32)
    bx = np.ascontiguousarray(box, np.uint32); cia = np.ascontiguousarray(ci, np.int32); pos = np.ascontiguousarray(chrom_pos, np.uint64)
    qs, qb, qt = (np.ascontiguousarray(q_index[0], np.uint64), np.ascontiguousarray(q_index[1], np.uint64), np.ascontiguousarray(q_index[2], np.uint32))
    gs, gb, gt = (np.ascontiguousarray(g_index[0], np.uint64), np.ascontiguousarray(g_index[1], np.uint64), np.ascontiguousarray(g_index[2], np.uint32))
    o = RscOpts(window, smallK, K, 1 if limitrefine else 0, max_freq)
    cap = 1 << 16
    while True:
        oq = np.zeros(cap, np.uint32); ot = np.zeros(cap, np.uint32); ob = np.zeros(4, np.uint32); eff = C.c_float(0)
        L.oracle_refine_splitchain.restype = C.c_long
        n = L.oracle_refine_splitchain(C.c_int(len(q)), _p(q, C.c_uint32), _p(t, C.c_uint32), _p(ln, C.c_int), _p(cl, C.c_int), _p(cs, C.c_uint8),
                                       C.c_int(len(sp)), _p(sp, C.c_int), _p(bx, C.c_uint32), C.c_int(int(strand)), C.c_int(int(chrom)),
                                       C.c_int(len(cia)), _p(cia, C.c_int), _p(pos, C.c_uint64), C.c_int(len(pos) - 1), C.c_uint32(int(read_len)),
                                       C.c_long(len(qs) - 1), _p(qs, C.c_uint64), _p(qb, C.c_uint64), _p(qt, C.c_uint32), C.c_long(len(gs) - 1),
                                       _p(gs, C.c_uint64), _p(gb, C.c_uint64), _p(gt, C.c_uint32), C.byref(o), C.c_long(cap), _p(oq, C.c_uint32),
                                       _p(ot, C.c_uint32), _p(ob, C.c_uint32), C.byref(eff))
        if n < 0:
            return None
        if n <= cap:
            return dict(q=oq[:n].copy(), t=ot[:n].copy(), box=ob.copy(), eff=np.float32(eff.value))
        cap = int(n)


class BtwnOpts(C.Structure):
    _fields_ = [("K", C.c_int), ("W", C.c_int), ("refineSpaceDist", C.c_int), ("anchorstoosparse", C.c_float), ("match", C.c_int),
                ("mismatch", C.c_int), ("indel", C.c_int), ("maxFreq", C.c_int)]


def refine_btwn_splitchain(match_off, mq, mt, box, strand, chrom, link, fwd: bytes, rc: bytes, genome: bytes, chrom_pos, K=10, W=5,
                           refineSpaceDist=10000, anchorstoosparse=0.01, match=4, mismatch=-1, indel=-2, max_freq=15):
    """Refine_Btwnsplitchain (ChainRefine.h:579) on one chain's refined clusters -> dict(off, q, t, box, refinespace, n_rev) or None (UB)."""
    L = lib()
    mo = np.ascontiguousarray(match_off, np.int32); mq = np.ascontiguousarray(mq, np.uint32); mt = np.ascontiguousarray(mt, np.uint32)
    bx = np.ascontiguousarray(box, np.uint32).reshape(-1); st = np.ascontiguousarray(strand, np.uint8); chv = np.ascontiguousarray(chrom, np.int32)
    lk = np.ascontiguousarray(link, np.uint8); pos = np.ascontiguousarray(chrom_pos, np.uint64)
    nsp = len(st)
    o = BtwnOpts(K, W, refineSpaceDist, anchorstoosparse, match, mismatch, indel, max_freq)
    cap = len(mq) + 4 * len(fwd) + 1024
    oo = np.zeros(nsp + 1, np.int32); oq = np.zeros(cap, np.uint32); ot = np.zeros(cap, np.uint32); ob = np.zeros(4 * max(1, nsp), np.uint32)
    orf = np.zeros(max(1, nsp), np.uint8); nrev = C.c_int(0)
    if len(lk) == 0: lk = np.zeros(1, np.uint8)
    L.oracle_refine_btwn_splitchain.restype = C.c_long
    n = L.oracle_refine_btwn_splitchain(C.c_int(nsp), _p(mo, C.c_int), _p(mq, C.c_uint32), _p(mt, C.c_uint32), _p(bx, C.c_uint32), _p(st, C.c_uint8),
                                        _p(chv, C.c_int), _p(lk, C.c_uint8), C.c_char_p(fwd), C.c_char_p(rc), C.c_uint32(len(fwd)), C.c_char_p(genome),
                                        _p(pos, C.c_uint64), C.c_int(len(pos) - 1), C.byref(o), C.c_long(cap), _p(oo, C.c_int), _p(oq, C.c_uint32),
                                        _p(ot, C.c_uint32), _p(ob, C.c_uint32), _p(orf, C.c_uint8), C.byref(nrev))
    if n < 0:
        return None
    assert n <= cap
    return dict(off=oo.copy(), q=oq[:n].copy(), t=ot[:n].copy(), box=ob.reshape(-1, 4)[:nsp].copy(), refinespace=orf[:nsp].copy(), n_rev=nrev.value)


def merge_extend(match_off, mq, mt, box, strand, chrom, read: bytes, genome: bytes, chrom_pos, K=10):
    """MergeChain + LinearExtend + DecideCoordinates + TrimOverlappedAnchors on one chain's refined clusters ->
    dict(member, anchor_off, q, t, len, box, strand, chrom)."""
    L = lib()
    mo = np.ascontiguousarray(match_off, np.int32); mq = np.ascontiguousarray(mq, np.uint32); mt = np.ascontiguousarray(mt, np.uint32)
    bx = np.ascontiguousarray(box, np.uint32).reshape(-1); st = np.ascontiguousarray(strand, np.uint8); chv = np.ascontiguousarray(chrom, np.int32)
    pos = np.ascontiguousarray(chrom_pos, np.uint64)
    nsp = len(st)
    cap = len(mq) + 8
    gm = np.zeros(nsp + 2, np.int32); ao = np.zeros(nsp + 2, np.int32); aq = np.zeros(cap, np.uint32); at = np.zeros(cap, np.uint32); al = np.zeros(cap, np.int32)
    gb = np.zeros(4 * (nsp + 1), np.uint32); gs = np.zeros(nsp + 1, np.uint8); gc = np.zeros(nsp + 1, np.int32)
    L.oracle_merge_extend.restype = C.c_int
    ng = L.oracle_merge_extend(C.c_int(nsp), _p(mo, C.c_int), _p(mq, C.c_uint32), _p(mt, C.c_uint32), _p(bx, C.c_uint32), _p(st, C.c_uint8), _p(chv, C.c_int),
                               C.c_char_p(read), C.c_uint32(len(read)), C.c_char_p(genome), _p(pos, C.c_uint64), C.c_int(K), C.c_long(cap), _p(gm, C.c_int),
                               _p(ao, C.c_int), _p(aq, C.c_uint32), _p(at, C.c_uint32), _p(al, C.c_int), _p(gb, C.c_uint32), _p(gs, C.c_uint8), _p(gc, C.c_int))
    assert ng >= 0
    n = int(ao[ng])
    return dict(member=gm[:ng + 1].copy(), anchor_off=ao[:ng + 1].copy(), q=aq[:n].copy(), t=at[:n].copy(), len=al[:n].copy(),
                box=gb.reshape(-1, 4)[:ng].copy(), strand=gs[:ng].copy(), chrom=gc[:ng].copy())


class RclOpts(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("window", "smallK", "K", "maxFreq")]


def refine_cluster(mq, mt, box, strand, chrom_pos, read_len, q_index, g_index, window=100, smallK=10, K=17, max_freq=15):
    """REFINEclusters (ClusterRefine.h:50) for one cluster -> dict(q, t, box, eff, chrom), "rejected" (CHROMIndex) or None (UB)."""
    L = lib()
    mq = np.ascontiguousarray(mq, np.uint32); mt = np.ascontiguousarray(mt, np.uint32); bx = np.ascontiguousarray(box, np.uint32)
    pos = np.ascontiguousarray(chrom_pos, np.uint64)
    qs, qb, qt = (np.ascontiguousarray(q_index[0], np.uint64), np.ascontiguousarray(q_index[1], np.uint64), np.ascontiguousarray(q_index[2], np.uint32))
    gs, gb, gt = (np.ascontiguousarray(g_index[0], np.uint64), np.ascontiguousarray(g_index[1], np.uint64), np.ascontiguousarray(g_index[2], np.uint32))
    o = RclOpts(window, smallK, K, max_freq)
    cap = 1 << 16
    while True:
        oq = np.zeros(cap, np.uint32); ot = np.zeros(cap, np.uint32); ob = np.zeros(4, np.uint32); eff = C.c_float(0); ch = C.c_int(0)
        L.oracle_refine_cluster.restype = C.c_long
        n = L.oracle_refine_cluster(C.c_int(len(mq)), _p(mq, C.c_uint32), _p(mt, C.c_uint32), _p(bx, C.c_uint32), C.c_int(int(strand)), _p(pos, C.c_uint64),
                                    C.c_int(len(pos) - 1), C.c_uint32(int(read_len)), C.c_long(len(qs) - 1), _p(qs, C.c_uint64), _p(qb, C.c_uint64),
                                    _p(qt, C.c_uint32), C.c_long(len(gs) - 1), _p(gs, C.c_uint64), _p(gb, C.c_uint64), _p(gt, C.c_uint32), C.byref(o),
                                    C.c_long(cap), C.byref(ch), _p(oq, C.c_uint32), _p(ot, C.c_uint32), _p(ob, C.c_uint32), C.byref(eff))
        if n == -2:
            return "rejected"
        if n < 0:
            return None
        if n <= cap:
            return dict(q=oq[:n].copy(), t=ot[:n].copy(), box=ob.copy(), eff=np.float32(eff.value), chrom=ch.value)
        cap = int(n)


class LraOpts(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("localW", "globalW", "localMaxFreq", "match", "mismatch", "indel", "localBand", "refineBySDP", "isOnt")] + [
        ("gapopen", C.c_float), ("gapextend", C.c_float), ("gaproot", C.c_float), ("gapCeiling1", C.c_int), ("gapCeiling2", C.c_int)]


def local_refine_alignment(chain_off, aq, at, alen, chain_strand, chain_chrom, first_sdp, n0, n1, h, fwd: bytes, rc: bytes, genome: bytes, chrom_pos, lsc=None,
                           min_anchors=2, **kw):
    """LocalRefineAlignment (LocalRefineAlignment.h:885) for one primary chain -> list of dict(strand, supp, secondary, n0, n1, value, chrom, blocks) or None (UB)."""
    L = lib()
    d = dict(localW=5, globalW=5, localMaxFreq=15, match=4, mismatch=-1, indel=-2, localBand=15, refineBySDP=1, isOnt=1, gapopen=7.0, gapextend=10.0, gaproot=1.5,
             gapCeiling1=1500, gapCeiling2=3000)
    d.update(kw)
    o = LraOpts(*[d[n] for n, _ in LraOpts._fields_])
    co = np.ascontiguousarray(chain_off, np.int32); aq = np.ascontiguousarray(aq, np.uint32); at = np.ascontiguousarray(at, np.uint32); al = np.ascontiguousarray(alen, np.int32)
    cs = np.ascontiguousarray(chain_strand, np.uint8); cc = np.ascontiguousarray(chain_chrom, np.int32); fs = np.ascontiguousarray(first_sdp, np.float32)
    a0 = np.ascontiguousarray(n0, np.int32); a1 = np.ascontiguousarray(n1, np.int32); pos = np.ascontiguousarray(chrom_pos, np.uint64)
    nch = len(cs)
    sizes = np.diff(co)
    if lsc is None:
        lsc = int(np.argmax(sizes)) if nch else 0                          # LargestSplitChain: first maximum
    max_seg = 4 * len(aq) + 8; cap = 4 * (len(aq) + len(fwd)) + 64
    seg = [np.zeros(max_seg, np.int32) for _ in range(5)]; sv = np.zeros(max_seg, np.float32); sc = np.zeros(max_seg, np.int32)
    sbo = np.zeros(max_seg + 1, np.int32); blk = np.zeros(3 * cap, np.int32)
    L.oracle_local_refine_alignment_ex.restype = C.c_int
    n = L.oracle_local_refine_alignment_ex(C.c_int(nch), _p(co, C.c_int), _p(aq, C.c_uint32), _p(at, C.c_uint32), _p(al, C.c_int), _p(cs, C.c_uint8), _p(cc, C.c_int),
                                        _p(fs, C.c_float), _p(a0, C.c_int), _p(a1, C.c_int), C.c_int(lsc), C.c_int(int(h)), C.c_int(int(min_anchors)), C.c_char_p(fwd), C.c_char_p(rc),
                                        C.c_uint32(len(fwd)), C.c_char_p(genome), _p(pos, C.c_uint64), C.byref(o), C.c_int(max_seg), _p(seg[0], C.c_int),
                                        _p(seg[1], C.c_int), _p(seg[2], C.c_int), _p(seg[3], C.c_int), _p(seg[4], C.c_int), _p(sv, C.c_float), _p(sc, C.c_int),
                                        _p(sbo, C.c_int), _p(blk, C.c_int), C.c_long(cap))
    if n == -1:
        return None
    assert n >= 0, n
    B = blk.reshape(-1, 3)
    return [dict(strand=int(seg[0][i]), supp=int(seg[1][i]), secondary=int(seg[2][i]), n0=int(seg[3][i]), n1=int(seg[4][i]), value=float(sv[i]), chrom=int(sc[i]),
                 blocks=B[sbo[i]:sbo[i + 1]].copy()) for i in range(n)]


# ---- chain post-filters + SPLITChain (a9, low-accuracy path) ----------------------------------------------------------
def split_chain(q, t, length, strand, cluster, link, chrom_pos, splitdist=50000, bypass=1):
    """One chain (trace-back order) -> dict(keep, link, splits=[dict(idx, link, type, strand, chrom, box, clusters)], split_link) or None (UB)."""
    L = lib()
    q = np.ascontiguousarray(q, np.uint32); t = np.ascontiguousarray(t, np.uint32); ln = np.ascontiguousarray(length, np.int32)
    st = np.ascontiguousarray(strand, np.uint8); cl = np.ascontiguousarray(cluster, np.int32); lk = np.ascontiguousarray(link, np.uint8)
    pos = np.ascontiguousarray(chrom_pos, np.uint64)
    n = len(q)
    m = max(1, n)
    keep = np.zeros(m, np.uint8); nk = C.c_int(0); lo = np.zeros(m, np.uint8)
    spOff = np.zeros(m + 2, np.int32); spIdx = np.zeros(m, np.int32); spLink = np.zeros(m, np.uint8); spType = np.zeros(m + 1, np.int8)
    spStrand = np.zeros(m + 1, np.uint8); spChrom = np.zeros(m + 1, np.int32); spBox = np.zeros(4 * (m + 1), np.uint32)
    ciOff = np.zeros(m + 2, np.int32); ciIdx = np.zeros(m, np.int32); sl = np.zeros(m + 1, np.uint8); nsl = C.c_int(0)
    L.oracle_split_chain.restype = C.c_int
    r = L.oracle_split_chain(C.c_int(n), _p(q, C.c_uint32), _p(t, C.c_uint32), _p(ln, C.c_int), _p(st, C.c_uint8), _p(cl, C.c_int), _p(lk, C.c_uint8),
                             _p(pos, C.c_uint64), C.c_int(len(pos)), C.c_int(splitdist), C.c_int(bypass), _p(keep, C.c_uint8), C.byref(nk),
                             _p(lo, C.c_uint8), _p(spOff, C.c_int), _p(spIdx, C.c_int), _p(spLink, C.c_uint8), spType.ctypes.data_as(C.c_char_p),
                             _p(spStrand, C.c_uint8), _p(spChrom, C.c_int), _p(spBox, C.c_uint32), _p(ciOff, C.c_int), _p(ciIdx, C.c_int),
                             _p(sl, C.c_uint8), C.byref(nsl))
    if r < 0:
        return None
    splits = []
    for k in range(r):
        a, b = spOff[k], spOff[k + 1]
        splits.append(dict(idx=spIdx[a:b].copy(), link=spLink[a:b - 1].copy() if b > a else spLink[0:0].copy(), type=chr(spType[k]), strand=int(spStrand[k]),
                           chrom=int(spChrom[k]), box=spBox[4 * k:4 * k + 4].copy(), clusters=ciIdx[ciOff[k]:ciOff[k + 1]].copy()))
    return dict(keep=keep[:n].copy(), n_kept=nk.value, link=lo[:max(nk.value - 1, 0)].copy(), splits=splits, split_link=sl[:nsl.value].copy())


# ---- RefineSpace (a11) ------------------------------------------------------------------------------------------------
def store_minimizers_noncanonical64(seq: bytes, k, w):
    L = lib()
    cap = len(seq) + 1
    keys = np.zeros(cap, np.uint64); pos = np.zeros(cap, np.uint32)
    L.oracle_store_minimizers_noncanonical64.restype = C.c_long
    n = L.oracle_store_minimizers_noncanonical64(C.c_char_p(seq), C.c_uint32(len(seq)), int(k), int(w), _p(keys, C.c_uint64), _p(pos, C.c_uint32), C.c_long(cap))
    return keys[:n].copy(), pos[:n].copy()


def refine_space(q: bytes, t: bytes, t_span, K, W, diag, match=4, mismatch=-1, indel=-2, max_freq=15, q_add=0, t_add=0, flip_len=0):
    """-> (pairs_q, pairs_t, identity)"""
    L = lib()
    cap = 4 * (len(q) + len(t)) + 64
    while True:
        oq = np.zeros(cap, np.uint32); ot = np.zeros(cap, np.uint32); ident = C.c_float(0)
        L.oracle_refine_space.restype = C.c_long
        n = L.oracle_refine_space(C.c_char_p(q), len(q), C.c_char_p(t), len(t), C.c_uint32(t_span), int(K), int(W), int(diag), int(match), int(mismatch),
                                  int(indel), C.c_long(max_freq), C.c_uint32(q_add), C.c_uint32(t_add), C.c_uint32(flip_len), _p(oq, C.c_uint32),
                                  _p(ot, C.c_uint32), C.c_long(cap), C.byref(ident))
        if n <= cap:
            return oq[:n].copy(), ot[:n].copy(), ident.value
        cap = n


def between_anchors(q: bytes, t: bytes, cur_read_end, next_read_start, cur_genome_end, next_genome_start, match=4, mismatch=-1, indel=-2, local_band=15,
                    refine_dp=1):
    """RefineByLinearAlignment for one anchor pair -> (blocks [n,3], score) or None (negative span)."""
    L = lib()
    cap = max(8, min(len(q), len(t)) + 8)
    blocks = np.zeros(3 * cap, np.int32); score = C.c_int(0)
    L.oracle_between_anchors.restype = C.c_int
    n = L.oracle_between_anchors(C.c_char_p(q), C.c_char_p(t), C.c_uint32(cur_read_end), C.c_uint32(next_read_start), C.c_uint32(cur_genome_end),
                                 C.c_uint32(next_genome_start), int(match), int(mismatch), int(indel), int(local_band), int(refine_dp),
                                 _p(blocks, C.c_int), cap, C.byref(score))
    if n < 0:
        return None
    return blocks[:3 * n].reshape(n, 3).copy(), score.value


# ---- RefineBreakpoint (a15) -------------------------------------------------------------------------------------------
def refine_breakpoint(read_len, l_blocks, l_strand, l_read: bytes, l_chrom: bytes, r_blocks, r_strand, r_read: bytes, r_chrom: bytes):
    """-> (ret, left blocks [n,3], right blocks [n,3]); ret 1 refined, 0 untouched, -1 the reference reads outside its inputs"""
    L = lib()
    lb = np.ascontiguousarray(l_blocks, np.int32).reshape(-1); rb = np.ascontiguousarray(r_blocks, np.int32).reshape(-1)
    nl, nr = len(lb) // 3, len(rb) // 3
    lo = np.zeros(3 * (nl + 502), np.int32); ro = np.zeros(3 * (nr + 502), np.int32); nlo = C.c_int(0); nro = C.c_int(0)
    L.oracle_refine_breakpoint.restype = C.c_int
    ret = L.oracle_refine_breakpoint(int(read_len), _p(lb, C.c_int), nl, int(l_strand), C.c_char_p(l_read), C.c_char_p(l_chrom), len(l_chrom),
                                     _p(rb, C.c_int), nr, int(r_strand), C.c_char_p(r_read), C.c_char_p(r_chrom), len(r_chrom),
                                     _p(lo, C.c_int), C.byref(nlo), _p(ro, C.c_int), C.byref(nro))
    return ret, lo[:3 * nlo.value].reshape(-1, 3).copy(), ro[:3 * nro.value].reshape(-1, 3).copy()


def filter_chain(q, t, length, strand, link, ops, qend=None):
    """Chain.h filters in the given order (1 small paired indels, 2/3 paired indels with/without refineEnds, 4 spurious anchors, 8 spurious jump).
    link=None for chain types without links.  -> (keep, surviving links)"""
    L = lib()
    q = np.ascontiguousarray(q, np.uint32); t = np.ascontiguousarray(t, np.uint32); ln = np.ascontiguousarray(length, np.int32)
    st = np.ascontiguousarray(strand, np.uint8)
    n = len(q)
    has = link is not None
    lk = np.ascontiguousarray(link if has else np.zeros(max(n - 1, 0)), np.uint8)
    ops = np.ascontiguousarray(ops, np.int32)
    keep = np.zeros(max(1, n), np.uint8); lo = np.zeros(max(1, n), np.uint8); nl = C.c_int(0)
    qe = None if qend is None else np.ascontiguousarray(qend, np.uint32)
    L.oracle_filter_chain_ex.restype = C.c_int
    L.oracle_filter_chain_ex(C.c_int(n), _p(q, C.c_uint32), _p(t, C.c_uint32), _p(ln, C.c_int), None if qe is None else _p(qe, C.c_uint32), _p(st, C.c_uint8),
                             _p(lk, C.c_uint8), C.c_int(int(has)), _p(ops, C.c_int), C.c_int(len(ops)), _p(keep, C.c_uint8), _p(lo, C.c_uint8), C.byref(nl))
    return keep[:n].copy(), lo[:nl.value].copy()


def merge_same_diag(q, t, length, overlap, strand, merge_dist=100):
    """MergeMatchesSameDiag (LinearExtend.h:795) on one extended cluster -> (start, end) anchor index ranges, or None for an empty cluster."""
    L = lib()
    q = np.ascontiguousarray(q, np.uint32); t = np.ascontiguousarray(t, np.uint32); ln = np.ascontiguousarray(length, np.int32)
    ov = np.ascontiguousarray(overlap, np.uint8)
    n = len(q)
    st = np.zeros(max(1, n), np.int32); en = np.zeros(max(1, n), np.int32)
    L.oracle_merge_same_diag.restype = C.c_int
    ng = L.oracle_merge_same_diag(C.c_int(n), _p(q, C.c_uint32), _p(t, C.c_uint32), _p(ln, C.c_int), _p(ov, C.c_uint8), C.c_int(int(strand)), C.c_int(int(merge_dist)),
                                  _p(st, C.c_int), _p(en, C.c_int))
    if ng < 0:
        return None
    return st[:ng].copy(), en[:ng].copy()


def switchindex(ch, link, coarse, cl_qs, cl_qe):
    """switchindex (Mapping_ultility.h:39) on one chain -> (ch, link) or None where the reference reads outside an array."""
    L = lib()
    ch = np.ascontiguousarray(ch, np.uint32); lk = np.ascontiguousarray(link, np.uint8)
    co = np.ascontiguousarray(coarse, np.int32); qs = np.ascontiguousarray(cl_qs, np.uint32); qe = np.ascontiguousarray(cl_qe, np.uint32)
    n = len(ch)
    oc = np.zeros(max(1, n), np.uint32); ol = np.zeros(max(1, n), np.uint8); nl = C.c_int(0)
    lk_in = lk if len(lk) else np.zeros(1, np.uint8)
    L.oracle_switchindex.restype = C.c_int
    r = L.oracle_switchindex(C.c_int(n), _p(ch if n else np.zeros(1, np.uint32), C.c_uint32), C.c_int(len(lk)), _p(lk_in, C.c_uint8), _p(co, C.c_int), _p(qs, C.c_uint32),
                             _p(qe, C.c_uint32), _p(oc, C.c_uint32), _p(ol, C.c_uint8), C.byref(nl))
    if r < 0:
        return None
    return oc[:r].copy(), ol[:nl.value].copy()


def trim_overlapped_anchors(q, t, length, strand):
    """TrimOverlappedAnchors(vector<Cluster>&, start) (LinearExtend.h:574) on one extended cluster -> (q, length) after trimming."""
    L = lib()
    q = np.ascontiguousarray(q, np.uint32).copy(); t = np.ascontiguousarray(t, np.uint32); ln = np.ascontiguousarray(length, np.int32).copy()
    if len(q):
        L.oracle_trim_overlapped_anchors(C.c_int(len(q)), _p(q, C.c_uint32), _p(t, C.c_uint32), _p(ln, C.c_int), C.c_int(int(strand)))
    return q, ln


def switch_to_original_anchors(chain_cluster, chain_entry, group_off, start, end, coarse):
    """SwitchToOriginalAnchors (LocalRefineAlignment.h:187) for one chain -> (anchor index within its cluster, ClusterIndex) arrays."""
    L = lib()
    cc = np.ascontiguousarray(chain_cluster, np.int32); ce = np.ascontiguousarray(chain_entry, np.uint32)
    go = np.ascontiguousarray(group_off, np.uint64); st = np.ascontiguousarray(start, np.uint32); en = np.ascontiguousarray(end, np.uint32)
    co = np.ascontiguousarray(coarse, np.int32)
    cap = int(sum(int(en[int(go[c]) + int(k)]) - int(st[int(go[c]) + int(k)]) for c, k in zip(cc, ce))) + 1
    oa = np.zeros(cap, np.uint32); oc = np.zeros(cap, np.int32)
    L.oracle_switch_to_original_anchors.restype = C.c_long
    m = L.oracle_switch_to_original_anchors(C.c_int(len(cc)), _p(cc if len(cc) else np.zeros(1, np.int32), C.c_int), _p(ce if len(ce) else np.zeros(1, np.uint32), C.c_uint32),
                                            _p(go, C.c_uint64), _p(st, C.c_uint32), _p(en, C.c_uint32), _p(co, C.c_int), _p(oa, C.c_uint32), _p(oc, C.c_int))
    return oa[:m].copy(), oc[:m].copy()


def refine_btwn_space(fwd: bytes, rc: bytes, chrom: bytes, qe, qs, te, ts, st, twoblocks, K=10, W=5, read_type=0, anchorstoosparse=0.005, match=4, mismatch=-1, indel=-2,
                      max_freq=15, lrts=0, lrlength=0):
    """RefineBtwnSpace (ClusterRefine.h:331) on one space -> (decision, pairs_q, pairs_t, eff, reff)"""
    L = lib()
    cap = 4 * ((qe - qs) + (te - ts + lrlength)) + 64
    oq = np.zeros(cap, np.uint32); ot = np.zeros(cap, np.uint32); n = C.c_long(0); eff = C.c_float(0); reff = C.c_float(0)
    L.oracle_refine_btwn_space.restype = C.c_int
    d = L.oracle_refine_btwn_space(int(K), int(W), int(twoblocks), int(read_type), C.c_float(anchorstoosparse), int(match), int(mismatch), int(indel), C.c_long(max_freq),
                                   C.c_char_p(fwd), C.c_char_p(rc), C.c_uint32(len(fwd)), C.c_char_p(chrom), C.c_uint32(qe), C.c_uint32(qs), C.c_uint32(te), C.c_uint32(ts),
                                   int(st), C.c_uint32(lrts), C.c_uint32(lrlength), _p(oq, C.c_uint32), _p(ot, C.c_uint32), C.c_long(cap), C.byref(n), C.byref(eff),
                                   C.byref(reff))
    assert n.value <= cap
    return d, oq[:n.value].copy(), ot[:n.value].copy(), np.float32(eff.value), np.float32(reff.value)
