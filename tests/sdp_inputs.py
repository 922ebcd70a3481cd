"""Helpers (tests / bench cpu_baseline only): the extended clusters SDP#A sees, produced by the ORACLE's a1-a7."""
import numpy as np


def oracle_ext_clusters(oracle, read: bytes, genome_arr, ik, ip, k=17, w=10, mf=150, preset="ONT"):
    """-> (cluster_off, strands, q, t(global), len) as Map_lowacc.h:77-150 hands them to SparseDP (:188)."""
    g = genome_arr.tobytes() + b"\0" * 64
    keys, pos = oracle.store_minimizers(read, k, w)
    sk, sp = oracle.sort_minimizers(keys, pos)
    qi, ti = oracle.compare_lists(sk, sp, ik, ip, mf)
    q, t, key = sp[qi], ip[ti], sk[qi]
    st = oracle.separate_strand(read, g, k, q, t)
    po = dict(oracle.CLEAN_PRESETS[preset]); po["globalK"] = k
    opts = oracle.CleanOpts(**po)
    offs = [0]; strands = []; Q = []; T = []; L = []
    chrom = genome_arr.tobytes()
    for strand in (0, 1):
        sel = st == strand
        oq, ot, cl = oracle.clean_matches(q[sel], t[sel], key[sel], strand, opts, [0, len(genome_arr)])
        for i in range(len(cl["start"])):
            a, b = int(cl["start"][i]), int(cl["end"][i])
            eq, et, el, _ = oracle.linear_extend(oq[a:b], ot[a:b], strand, k, read, chrom)
            Q.extend(eq.tolist()); T.extend(et.tolist()); L.extend(el.tolist())
            strands.append(strand); offs.append(len(Q))
    return (np.array(offs, np.int32), np.array(strands, np.uint8), np.array(Q, np.uint32), np.array(T, np.uint32), np.array(L, np.int32))
