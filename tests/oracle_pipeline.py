"""MapRead_lowacc for ONE read composed from the oracle's stage functions only (test infrastructure: used by tests/ and by bench.py's
cpu_baseline leg, never by the product).  It follows Map_lowacc.h:69-632 stage by stage exactly as lra_amd/csrc/mapread.hip chains the
device stages, so its alignments must equal the GPU path's (tests/test_mapread.py) and its run time is the CPU cost of the same work.
map_read_lowacc / map_reads_lowacc_mt call the C++ composition (oracle/pipeline.cpp: one read; a thread pool over reads);
map_read_lowacc_py is the same composition in Python, kept as a cross-check of the C++ one (tests/test_mapread.py).
chrom_pos (Genome::header.pos) defaults to one chromosome; the stages that take a chromosome's bytes get the slice."""
import ctypes as C
import numpy as np

import oracle_lib as O

ONT = dict(globalK=17, globalW=10, globalMaxFreq=150, localK=10, localW=5, localMaxFreq=15, localIndexWindow=256, refineBand=7, match=4, mismatch=-1,
           indel=-2, refineSpaceDist=30000, anchorstoosparse=0.005, splitdist=50000, window=100, initial_anchorbonus=20.0, second_anchorbonus=2.0,
           alnthres=0.65, SecondCleanMaxDiag=100, refineBreakpoint=False)
CLR = dict(ONT, globalK=15, globalMaxFreq=250, refineBand=20, initial_anchorbonus=15.0, second_anchorbonus=6.0, alnthres=0.50, SecondCleanMaxDiag=120)

_CHROM_CACHE = {}
TRACE = {}                                                                # what the last map_read_lowacc call decided (match_rate, ...): for tests
_COMP = np.zeros(256, np.uint8)
for a, b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    _COMP[a] = b


def revcomp_bytes(s: bytes) -> bytes:
    return _COMP[np.frombuffer(s, np.uint8)][::-1].tobytes()


def seq_offsets(n, window):
    return np.array(list(range(0, n, window)) + [n], np.uint64) if n else np.zeros(1, np.uint64)


def map_read_lowacc_py(read: bytes, genome: bytes, idx_key, idx_pos, g_index, opts=None, clean_opts=None, stats=True, chrom_pos=None):
    """-> (alignments, unaligned).  alignments: list over primary chains p of lists of dict(strand, supp, secondary, n0, n1, value, chrom,
    a13_blocks, blocks (after IndelRefineAlignment), refine_status[, counts, nv, cigar]).  genome: the chromosome's bases (+ padding);
    g_index = (seqOffsets, tupleBoundaries, tuples) of the genome's local index."""
    o = dict(ONT)
    if opts:
        o.update(opts)
    G = len(genome.rstrip(b"\0")) if chrom_pos is None else int(chrom_pos[-1])
    CH = [0, G] if chrom_pos is None else [int(x) for x in chrom_pos]
    def chrom_bytes(c, padded=True):                                      # the chromosome's bases, sliced once per genome object (exact, or + 64 bytes of padding)
        key = (id(genome), len(genome), tuple(CH))
        if _CHROM_CACHE.get("key") != key:
            _CHROM_CACHE.clear(); _CHROM_CACHE["key"] = key
        if (c, padded) not in _CHROM_CACHE:
            if padded:
                _CHROM_CACHE[(c, padded)] = genome if (len(CH) == 2 and len(genome) >= G + 64) else genome[CH[c]:CH[c + 1]] + b"\0" * 64
            else:
                _CHROM_CACHE[(c, padded)] = genome if (len(CH) == 2 and len(genome) == G) else genome[CH[c]:CH[c + 1]]
        return _CHROM_CACHE[(c, padded)]
    L = len(read)
    K = o["globalK"]
    co = clean_opts or O.CleanOpts(**dict(O.CLEAN_PRESETS["ONT"], globalK=K, SecondCleanMaxDiag=o["SecondCleanMaxDiag"]))
    sdp_kw = dict(alnthres=o["alnthres"], globalK=K)
    # a1-a4 (MapRead.h:169-203)
    keys, pos = O.store_minimizers(read, K, o["globalW"])
    sk, sp = O.sort_minimizers(keys, pos)
    qi, ti = O.compare_lists(sk, sp, idx_key, idx_pos, o["globalMaxFreq"])
    st = O.separate_strand(read, genome, K, sp[qi], idx_pos[ti])
    # a5, a7 (Map_lowacc.h:60-184)
    offs = [0]; cst = []; Q = []; T = []; Ln = []
    repetitive = False                                                    # Map_lowacc.h:86-89
    for strand in (0, 1):
        sel = st == strand
        oq, ot, cl = O.clean_matches(sp[qi][sel], idx_pos[ti][sel], sk[qi][sel], strand, co, CH)
        for ci in range(len(cl["start"])):
            a, b = int(cl["start"][ci]), int(cl["end"][ci])
            if 1.0 < float(cl["freq"][ci]) <= 2.0 and b - a >= 500:
                repetitive = True
            c = int(cl["chrom"][ci]); off = np.uint32(CH[c])
            eq, et, el, _ = O.linear_extend(oq[a:b], ot[a:b] - off, strand, K, read, chrom_bytes(c, False))
            Q.append(eq); T.append(et + off); Ln.append(el); cst.append(strand); offs.append(offs[-1] + len(eq))
    if not cst:
        return [], True
    Q = np.concatenate(Q); T = np.concatenate(T); Ln = np.concatenate(Ln)
    # a8: primary chains (Map_lowacc.h:185-188)
    match_rate = 3.0 if repetitive else o["initial_anchorbonus"]          # Map_lowacc.h:184-185
    TRACE["match_rate"] = match_rate
    first = O.sdp_chain(offs, cst, Q, T, Ln, O.sdp_opts(L, rate=match_rate, **sdp_kw))
    if first["status"] < 0 or not first["chains"]:
        return [], True
    offs_a = np.asarray(offs)
    fwd = read; rc = revcomp_bytes(read)
    q_index = [None, None]
    alignments = []
    for p, ch in enumerate(first["chains"]):
        fr = ch["frags"].astype(np.int64)
        cl_of = np.searchsorted(offs_a, fr, side="right") - 1
        cstrand = np.asarray(cst, np.uint8)[cl_of]
        # a9 (Map_lowacc.h:189-245)
        sc = O.split_chain(Q[fr], T[fr], Ln[fr], cstrand, cl_of, ch["link"], CH, o["splitdist"], 1)
        if sc is None or not len(sc["splits"]):                           # Map_lowacc.h:263-267: p == 0 -> unaligned, p > 0 -> break
            if p == 0: return [[]], True
            break
        kb = sc["keep"].astype(bool)
        q, t, al, cl, cs_ = Q[fr][kb], T[fr][kb], Ln[fr][kb], cl_of[kb], cstrand[kb]
        nsp = len(sc["splits"])
        segs = []
        reached = False                                                   # does p get to `alignments.resize(alignments.size() + 1)` (:574)?
        if nsp:
            # a10 (Map_lowacc.h:246-294)
            moff = [0]; mq = []; mt = []; boxes = []; ok = True
            for s in sc["splits"]:
                sd = s["strand"]
                if q_index[sd] is None:
                    tup, bnd = O.local_index_seq(fwd if sd == 0 else rc, o["localK"], o["localW"], o["localIndexWindow"], o["localMaxFreq"])
                    q_index[sd] = (seq_offsets(L, o["localIndexWindow"]), bnd, tup)
                rs = O.refine_splitchain(q, t, al, cl, cs_, s["idx"], s["box"], sd, s["chrom"], s["clusters"], CH, L, q_index[sd], g_index,
                                         window=o["window"], smallK=o["localK"], K=K, limitrefine=True, max_freq=o["localMaxFreq"])
                if rs is None:
                    ok = False
                    break
                mq.extend(rs["q"].tolist()); mt.extend(rs["t"].tolist()); moff.append(len(mq)); boxes.append(rs["box"])
            if ok:
                strands = [s["strand"] for s in sc["splits"]]; chroms = [s["chrom"] for s in sc["splits"]]
                # a11 callers, MergeChain, second LinearExtend + Trim (Map_lowacc.h:362-476)
                eb = O.refine_btwn_splitchain(moff, mq, mt, np.array(boxes, np.uint32).reshape(-1, 4), strands, chroms, sc["split_link"], fwd, rc, genome, CH,
                                              K=o["localK"], W=o["localW"], refineSpaceDist=o["refineSpaceDist"], anchorstoosparse=o["anchorstoosparse"],
                                              match=o["match"], mismatch=o["mismatch"], indel=o["indel"], max_freq=o["localMaxFreq"])
                if eb is not None and len(eb["q"]) > 0:                    # SizeRefinedClusters > 0 (:486-491)
                    reached = True
                    em = O.merge_extend(eb["off"], eb["q"], eb["t"], eb["box"], strands, chroms, fwd, genome, CH, K=o["localK"])
                    # second sparse DP + RemovePairedIndels / RemoveSpuriousAnchors (Map_lowacc.h:521-540)
                    chains = []
                    for g in range(len(em["member"]) - 1):
                        a0, a1 = int(em["anchor_off"][g]), int(em["anchor_off"][g + 1])
                        if a1 == a0:
                            continue
                        sg = int(em["strand"][g])
                        e2 = O.sdp_chain([0, a1 - a0], [sg], em["q"][a0:a1], em["t"][a0:a1], em["len"][a0:a1], O.sdp_opts(L, mode=1, rate=o["second_anchorbonus"], **sdp_kw))
                        if e2["status"] < 0 or not e2["chains"]:
                            continue
                        ix = e2["chains"][0]["frags"].astype(np.int64)
                        cq, ct, cln = em["q"][a0:a1][ix], em["t"][a0:a1][ix], em["len"][a0:a1][ix]
                        keep, _ = O.filter_chain(cq, ct, cln, [sg] * len(ix), None, [2, 4])
                        k2 = keep.astype(bool)
                        chains.append((cq[k2], ct[k2], cln[k2], sg, int(em["chrom"][g]), e2["chains"][0]["value"], len(ix)))
                    if chains:
                        # a13 (Map_lowacc.h:576)
                        off = [0]; aq = []; at = []; aln = []
                        for c in chains:
                            aq.extend(c[0].tolist()); at.extend(c[1].tolist()); aln.extend(c[2].tolist()); off.append(len(aq))
                        segs = O.local_refine_alignment(off, aq, at, aln, [c[3] for c in chains], [c[4] for c in chains], [c[5] for c in chains],
                                                        [len(fr)] * len(chains), [c[6] for c in chains], p, fwd, rc, genome, CH) or []
        if not reached:                                                   # :486-491 (or a stage hit undefined behaviour)
            if p == 0: return [[]], True
            break
        out = []
        for s in segs:
            sb = fwd if s["strand"] == 0 else rc
            # a14 (Map_lowacc.h:582-585)
            refined, rst = O.indel_refine(s["blocks"], sb, chrom_bytes(s["chrom"]), o["refineBand"], o["match"], o["mismatch"], o["indel"])
            out.append(dict(s, a13_blocks=s["blocks"], blocks=refined, refine_status=rst))
        if o["refineBreakpoint"]:                                          # a15 (Map_lowacc.h:586-596): segments come right to left on the read
            for si in range(1, len(out)):
                l, r = out[si], out[si - 1]
                ret, lb, rb = O.refine_breakpoint(L, l["blocks"], l["strand"], fwd if l["strand"] == 0 else rc, chrom_bytes(l["chrom"], False), r["blocks"], r["strand"],
                                                  fwd if r["strand"] == 0 else rc, chrom_bytes(r["chrom"], False))
                if ret >= 0:
                    l["blocks"], r["blocks"] = lb, rb
                l["breakpoint"] = ret
        for d in out:                                                      # a16 (Map_lowacc.h:597-599)
            if stats and d["refine_status"] == 0 and len(d["blocks"]):
                d["stats"] = O.calculate_statistics(d["blocks"], fwd if d["strand"] == 0 else rc, chrom_bytes(d["chrom"]))
        alignments.append(out)
        if p == 0 and not out:
            return alignments, True                                        # Map_lowacc.h:578-581
    return alignments, False


# ---------------------------------------------------------------------------------------------------------------- the C++ composition
class MapOpts(C.Structure):
    """oracle_map_opts (oracle/pipeline.cpp)"""
    _fields_ = ([(n, C.c_int) for n in ("globalK", "globalW", "globalMaxFreq", "localK", "localW", "localMaxFreq", "localIndexWindow", "refineBand", "match", "mismatch",
                                        "indel", "localBand", "refineSpaceDist")] +
                [("anchorstoosparse", C.c_float), ("splitdist", C.c_int), ("window", C.c_int), ("initial_anchorbonus", C.c_float), ("second_anchorbonus", C.c_float),
                 ("alnthres", C.c_float), ("NumAln", C.c_int), ("gapopen", C.c_float), ("gapextend", C.c_float), ("gaproot", C.c_float), ("gapCeiling1", C.c_int),
                 ("gapCeiling2", C.c_int), ("refineBreakpoint", C.c_int), ("stats", C.c_int), ("limitrefine", C.c_int), ("isOnt", C.c_int), ("clean", O.CleanOpts)])


def _c_opts(opts, clean_opts, stats):
    o = dict(ONT)
    if opts:
        o.update(opts)
    m = MapOpts()
    for n in ("globalK", "globalW", "globalMaxFreq", "localK", "localW", "localMaxFreq", "localIndexWindow", "refineBand", "match", "mismatch", "indel", "refineSpaceDist",
              "anchorstoosparse", "splitdist", "window", "initial_anchorbonus", "second_anchorbonus", "alnthres"):
        setattr(m, n, o[n])
    m.localBand = 15; m.NumAln = O.SDP_ONT["NumAln"]
    for n in ("gapopen", "gapextend", "gaproot", "gapCeiling1", "gapCeiling2"):
        setattr(m, n, O.SDP_ONT[n])
    m.refineBreakpoint = int(bool(o["refineBreakpoint"])); m.stats = int(bool(stats)); m.limitrefine = 1; m.isOnt = 1
    m.clean = clean_opts or O.CleanOpts(**dict(O.CLEAN_PRESETS["ONT"], globalK=o["globalK"], SecondCleanMaxDiag=o["SecondCleanMaxDiag"]))
    return m


def seq_offsets_multi(chrom_pos, window):
    """LocalIndex::seqOffsets of several sequences (MMIndex.h:200-245): window ends, restarting at every sequence."""
    out = [np.zeros(1, np.uint64)]
    for c in range(len(chrom_pos) - 1):
        a, b = int(chrom_pos[c]), int(chrom_pos[c + 1])
        e = np.arange(a + window, b, window, dtype=np.uint64)
        out.append(e); out.append(np.array([b], np.uint64))
    return np.concatenate(out)


def _ref_args(genome, idx_key, idx_pos, g_index, chrom_pos):
    G = len(genome.rstrip(b"\0")) if chrom_pos is None else int(chrom_pos[-1])
    cp = np.ascontiguousarray([0, G] if chrom_pos is None else chrom_pos, dtype=np.uint64)
    ik = np.ascontiguousarray(idx_key).view(np.uint64); ip = np.ascontiguousarray(idx_pos, dtype=np.uint32)
    gs, gb, gt = (np.ascontiguousarray(g_index[0], np.uint64), np.ascontiguousarray(g_index[1], np.uint64), np.ascontiguousarray(g_index[2], np.uint32))
    lut = O.log_lookup_table()
    keep = (cp, ik, ip, gs, gb, gt, lut, genome)
    args = (C.c_char_p(genome), C.c_uint64(G), O._p(cp, C.c_uint64), C.c_int(len(cp) - 1), O._p(ik, C.c_uint64), O._p(ip, C.c_uint32), C.c_long(len(ik)),
            C.c_long(len(gs) - 1), O._p(gs, C.c_uint64), O._p(gb, C.c_uint64), O._p(gt, C.c_uint32), O._p(lut, C.c_float))
    return args, keep


def map_read_lowacc(read: bytes, genome: bytes, idx_key, idx_pos, g_index, opts=None, clean_opts=None, stats=True, chrom_pos=None):
    """-> (alignments, unaligned) exactly like map_read_lowacc_py, computed by oracle_map_read_lowacc (oracle/pipeline.cpp)."""
    L = O.lib()
    m = _c_opts(opts, clean_opts, stats)
    args, keep = _ref_args(genome, idx_key, idx_pos, g_index, chrom_pos)
    L.oracle_map_read_lowacc.restype = C.c_long
    n = L.oracle_map_read_lowacc(C.c_char_p(read), C.c_uint32(len(read)), *args, C.byref(m))
    f = np.zeros(max(1, n), np.int32)
    L.oracle_map_read_result(O._p(f, C.c_int32))
    unaligned, ng = bool(f[0]), int(f[1])
    TRACE["match_rate"] = float(f[2:3].view(np.float32)[0])
    at = 3
    alignments = []
    for _ in range(ng):
        ns = int(f[at]); at += 1
        segs = []
        for _ in range(ns):
            h = f[at:at + 14]; at += 14
            counts = f[at:at + 18].astype(np.int64); at += 18
            na, nb, nr = int(h[11]), int(h[12]), int(h[13])
            a13 = f[at:at + 3 * na].reshape(-1, 3).copy(); at += 3 * na
            blocks = f[at:at + 3 * nb].reshape(-1, 3).copy(); at += 3 * nb
            runs = f[at:at + nr].view(np.uint32).copy(); at += nr
            d = dict(strand=int(h[0]), supp=int(h[1]), secondary=int(h[2]), n0=int(h[3]), n1=int(h[4]), chrom=int(h[5]), value=float(h[6:7].view(np.float32)[0]),
                     refine_status=int(h[7]), a13_blocks=a13, blocks=blocks)
            if h[8] != -2:
                d["breakpoint"] = int(h[8])
            if h[9]:
                d["stats"] = (dict(zip(O.STAT_NAMES, counts.tolist())), np.float32(h[10:11].view(np.float32)[0]), runs,
                              "".join("%d%s" % (r >> 4, "=XID"[r & 15]) for r in runs))
            segs.append(d)
        alignments.append(segs)
    return alignments, unaligned


def map_reads_lowacc_mt(reads, off, first, n, genome: bytes, idx_key, idx_pos, g_index, opts=None, chrom_pos=None, n_threads=1, clean_opts=None, stats=True):
    """reads [first, first + n) of a batch (uint8 bases back to back, off) through oracle_map_reads_lowacc_mt on n_threads host threads.
    -> dict(seconds, bases, n_alignments, checksum, n_reads)"""
    L = O.lib()
    m = _c_opts(opts, clean_opts, stats)
    args, keep = _ref_args(genome, idx_key, idx_pos, g_index, chrom_pos)
    r = np.ascontiguousarray(reads, dtype=np.uint8); o_ = np.ascontiguousarray(off, dtype=np.uint64)
    sec, bases, nal, cs = C.c_double(0), C.c_long(0), C.c_long(0), C.c_uint64(0)
    L.oracle_map_reads_lowacc_mt(r.ctypes.data_as(C.c_char_p), O._p(o_, C.c_uint64), C.c_long(first), C.c_long(n), *args, C.byref(m), C.c_int(n_threads),
                                 C.byref(sec), C.byref(bases), C.byref(nal), C.byref(cs))
    return dict(seconds=sec.value, bases=bases.value, n_alignments=nal.value, checksum=cs.value, n_reads=int(n))


# ------------------------------------------------------------------------------------------------------------------------ MapRead_highacc
# -CCS (lra.cpp:306-340) and -CONTIG (lra.cpp:268-305) over the defaults of Options.h:127-230
CCS = dict(globalK=17, globalW=20, globalMaxFreq=150,      # globalK: what ReadIndex leaves (the index's K, 17 for `lra index -CCS`), not the preset's 25
            localK=7, localW=5, localMaxFreq=15, localIndexWindow=256, window=100, readType=2, refineBand=7, match=4, mismatch=-3, indel=-4, localBand=15, NumAln=2,
           alnthres=0.7, initial_anchorbonus=10.0, second_anchorbonus=2.0, splitdist=50000, anchorstoosparse=0.005, merge_dist=100, gapopen=4.0, gapextend=15.0,
           gaproot=1.5, gapCeiling1=2000, gapCeiling2=3000, refineBreakpoint=False, skipBandedRefine=False,
           clean=dict(cleanMaxDiag=150, minDiagCluster=10, bypassClustering=0, cleanClustersize=100, SecondCleanMinDiagCluster=30, SecondCleanMaxDiag=100, punish_anchorfreq=10,
                      anchorPerlength=10),
           fine=dict(RoughClustermaxGap=500, maxDiag=500, maxGap=400, minClusterSize=10, minUniqueStretchNum=1, minUniqueStretchDist=50))
CONTIG = dict(CCS, globalK=19, globalW=10, globalMaxFreq=30, readType=3, refineBand=50, gapextend=20.0, gapCeiling1=3000, gapCeiling2=5000, initial_anchorbonus=1.0,
              clean=dict(CCS["clean"], minDiagCluster=30), fine=dict(CCS["fine"], maxDiag=100, maxGap=500))


def map_read_highacc(read: bytes, genome: bytes, idx_key, idx_pos, opts=None, chrom_pos=None, stats=True, g_index=None):
    """MapRead (MapRead.h:153-263) + MapRead_highacc (Map_highacc.h:37-798) for ONE read, composed from the oracle's stage functions.
    -> (groups, unaligned, note): groups = list over the chains h of Primary_chains[0] that have clusters (dict(h=, segs=[...])), every seg as in
    map_read_lowacc_py plus `stats` with the counters the reference's two CalculateStatistics calls leave (tdel, tins and the six size classes
    accumulate over both, Alignment.h:440-512 / :85-86).  A read that takes the REFINEclusters branch (Map_highacc.h:413-447) needs g_index = (seqOffsets,
    tupleBoundaries, tuples) of the genome's local index (note = "sparse" and groups = None without it); TRACE["sparse"] says which branch ran.  note = "ub"
    where a stage reads outside an array."""
    o = dict(CCS)
    if opts:
        o.update(opts)
    CH = [0, len(genome.rstrip(b"\0"))] if chrom_pos is None else [int(x) for x in chrom_pos]
    gpad = genome if len(genome) >= CH[-1] + 64 else genome + b"\0" * 64
    chrom_b = lambda c: genome[CH[c]:CH[c + 1]]
    L = len(read); K = o["globalK"]; W = o["globalW"]
    fwd = read; rc = revcomp_bytes(read)
    # a1-a4 (MapRead.h:169-203): forward-strand matches first
    keys, pos = O.store_minimizers(read, K, W)
    sk, sp = O.sort_minimizers(keys, pos)
    qi, ti = O.compare_lists(sk, sp, idx_key, idx_pos, o["globalMaxFreq"])
    if len(qi) == 0:
        return [], True, None
    st = O.separate_strand(read, gpad, K, sp[qi], idx_pos[ti])
    f = st == 0
    mq = np.concatenate([sp[qi][f], sp[qi][~f]]); mt = np.concatenate([idx_pos[ti][f], idx_pos[ti][~f]]); mk = np.concatenate([sk[qi][f], sk[qi][~f]])
    # a5 (Map_highacc.h:41-42)
    fc, ust = O.matches_to_fine_clusters(mq, mt, mk, int(f.sum()), O.CleanOpts(globalK=K, **o["clean"]), O.FineOpts(globalK=K, **o["fine"]), CH)
    if ust:
        return None, False, "ub"
    nC = len(fc["strand"])
    if nC == 0:
        return [], True, None
    # a6 (:153-155), a8 SDP#C (:224-229), switchindex (:274)
    sc = O.split_clusters(fc["box"][:, 0], fc["box"][:, 1], fc["box"][:, 2], fc["box"][:, 3], fc["strand"], fc["freq"], fc["off"], fc["q"], contig=o["readType"] == 3, K=K)
    if len(sc["qs"]) == 0:
        return [], True, None
    rate = o["initial_anchorbonus"]
    if len(sc["qs"]) // nC > 20:
        rate = rate / 2.0
    TRACE["rate"] = rate
    sdp_kw = dict(NumAln=o["NumAln"], alnthres=o["alnthres"], gapopen=o["gapopen"], gapextend=o["gapextend"], gaproot=o["gaproot"], gapCeiling1=o["gapCeiling1"],
                  gapCeiling2=o["gapCeiling2"], globalK=K)
    first = O.sdp_chain_boxes(sc["qs"], sc["qe"], sc["ts"], sc["te"], sc["strand"], sc["val"], sc["num"], O.sdp_opts(L, rate=rate, **sdp_kw))
    if first["status"] < 0:
        return None, False, "ub"
    if not first["chains"]:
        return [], True, None
    chains = []
    for ch in first["chains"]:
        sw = O.switchindex(ch["frags"], ch["link"], sc["coarse"], fc["box"][:, 0], fc["box"][:, 1])
        if sw is None:
            return None, False, "ub"
        chains.append(dict(ch=[int(x) for x in sw[0]], link=[int(x) for x in sw[1]], value=ch["value"], n0=ch["num_anchors"]))
    # clusters no chain uses are dropped, the rest renumbered (:285-318)
    used = sorted({c for h in chains for c in h["ch"]})
    if not used:
        return [], True, None
    renum = {c: i for i, c in enumerate(used)}
    for h in chains:
        h["ch"] = [renum[c] for c in h["ch"]]
    cl = []
    for c in used:
        a, b = int(fc["off"][c]), int(fc["off"][c + 1])
        cl.append(dict(q=fc["q"][a:b].copy(), t=fc["t"][a:b].copy(), box=fc["box"][c].astype(np.int64), strand=int(fc["strand"][c]), chrom=int(fc["chrom"][c]),
                       freq=np.float32(fc["freq"][c])))
    # sparse (:413-416): a cluster with at most one anchor per 100 read bases on a read of at most 50 kb
    sparse = any(np.float32(np.float32(len(c["q"])) / np.float32(int(c["box"][1]) - int(c["box"][0]))) <= np.float32(0.01) and L <= 50000 for c in cl)
    TRACE["sparse"] = sparse
    Kg = K                                                                # opts.globalK (REFINEclusters' second options argument)
    if sparse:                                                            # :429-447: REFINEclusters; K, W = glIndex.k, glIndex.w from here on (:466-468)
        if g_index is None:
            return None, False, "sparse"
        q_index = [None, None]
        for c in cl:
            sd = c["strand"]
            if q_index[sd] is None:
                tup, bnd = O.local_index_seq(fwd if sd == 0 else rc, o["localK"], o["localW"], o["localIndexWindow"], o["localMaxFreq"])
                q_index[sd] = (seq_offsets(L, o["localIndexWindow"]), bnd, tup)
            rr = O.refine_cluster(c["q"], c["t"], c["box"], sd, CH, L, q_index[sd], g_index, window=o["window"], smallK=o["localK"], K=Kg, max_freq=o["localMaxFreq"])
            if rr is None:
                return None, False, "ub"
            if rr == "rejected":
                c["q"] = np.zeros(0, np.uint32); c["t"] = np.zeros(0, np.uint32)
            else:
                c["q"], c["t"], c["box"], c["chrom"] = rr["q"], rr["t"], rr["box"].astype(np.int64), rr["chrom"]
        K, W = o["localK"], o["localW"]
        for h in chains:                                                  # :475-487 (`link` keeps its length)
            h["ch"] = [ci for ci in h["ch"] if len(cl[ci]["q"])]
    else:
        for c in cl:                                                      # :449-460
            off = CH[c["chrom"]]
            c["t"] = c["t"] - np.uint32(off); c["box"][2] -= off; c["box"][3] -= off
    if not chains:
        return [], True, None
    # a11 caller (:515-520)
    moff = np.concatenate([[0], np.cumsum([len(c["q"]) for c in cl])])
    coff = np.concatenate([[0], np.cumsum([len(h["ch"]) for h in chains])])
    rb = O.refine_btwn_clusters_chains(moff, np.concatenate([c["q"] for c in cl]), np.concatenate([c["t"] for c in cl]), np.array([c["box"] for c in cl], np.int64),
                                       [c["strand"] for c in cl], [c["chrom"] for c in cl], [c["freq"] for c in cl], coff, [x for h in chains for x in h["ch"]], fwd, rc, gpad, CH,
                                       K=K, W=W, read_type=o["readType"], anchorstoosparse=o["anchorstoosparse"], match=o["match"], mismatch=o["mismatch"], indel=o["indel"],
                                       max_freq=o["localMaxFreq"])
    for i, c in enumerate(cl):
        a, b = int(rb["off"][i]), int(rb["off"][i + 1])
        c["q"] = rb["q"][a:b].copy(); c["t"] = rb["t"][a:b].copy(); c["box"] = rb["box"][i].astype(np.int64); c["freq"] = np.float32(rb["freq"][i])
    # a7 cluster version (:573-582) and MergeMatchesSameDiag (:642)
    ext = []
    for h in chains:
        h["first"] = len(ext)
        for k, ci in enumerate(h["ch"]):
            c = cl[ci]
            pv = cl[h["ch"][k - 1]]["box"] if k > 0 else None
            nx = cl[h["ch"][k + 1]]["box"] if k + 1 < len(h["ch"]) else None
            e = O.linear_extend_cluster(c["q"], c["t"], c["strand"], c["box"], pv, nx, c["freq"], fwd, chrom_b(c["chrom"]), K=K, skiprepetitive=True, trim=True)
            c["q"], c["t"] = e["sorted_q"], e["sorted_t"]                  # LinearExtend sorts RefinedClusters[cm]->matches in place (:201-210)
            ext.append(dict(q=e["q"], t=e["t"], len=e["len"], overlap=e["overlap"], box=e["box"], strand=c["strand"], chrom=c["chrom"]))
    if sum(len(c["q"]) for c in cl) == 0:
        return [], True, None
    for e in ext:
        m = O.merge_same_diag(e["q"], e["t"], e["len"], e["overlap"], e["strand"], o["merge_dist"])
        if m is None:
            return None, False, "ub"
        s_, e_ = m
        last = e_ - 1
        ln = np.where(e["q"][last].astype(np.int64) + e["len"][last] >= e["q"][s_], e["q"][last].astype(np.int64) + e["len"][last] - e["q"][s_], 0)   # Cluster_SameDiag::length
        e["sd"] = dict(start=s_, end=e_, q=e["q"][s_], t=e["t"][s_] if e["strand"] == 0 else e["t"][last], len=ln.astype(np.int32),
                       qend=(e["q"][last].astype(np.int64) + ln).astype(np.uint32))           # GetqStart, GettStart, length, GetqEnd (Clustering.h:366-390)
    lra_kw = dict(localW=o["localW"], globalW=o["localW"], localMaxFreq=o["localMaxFreq"], match=o["match"], mismatch=o["mismatch"], indel=o["indel"], localBand=o["localBand"],
                  refineBySDP=1, isOnt=0, gapopen=o["gapopen"], gapextend=o["gapextend"], gaproot=o["gaproot"], gapCeiling1=o["gapCeiling1"], gapCeiling2=o["gapCeiling2"])
    groups = []
    for hi, h in enumerate(chains):
        n = len(h["ch"])
        if n == 0:
            continue
        E = ext[h["first"]:h["first"] + n]
        # a9 high-accuracy SPLITChain + LSC (:705-707)
        sp = O.split_chain_highacc([e["strand"] for e in E], [e["chrom"] for e in E], [e["box"] for e in E], h["link"], o["splitdist"])
        off = [0]; aq = []; at = []; al = []; cstr = []; cchr = []
        for st_ in range(len(sp["type"])):
            vs = sp["idx"][sp["off"][st_]:sp["off"][st_ + 1]]
            # SDP#D + filters + SwitchToOriginalAnchors (LocalRefineAlignment.h:556-577)
            co2 = np.concatenate([[0], np.cumsum([len(E[v]["sd"]["q"]) for v in vs])])
            sq = np.concatenate([E[v]["sd"]["q"] for v in vs]); st2 = np.concatenate([E[v]["sd"]["t"] for v in vs]); sl = np.concatenate([E[v]["sd"]["len"] for v in vs])
            sqe = np.concatenate([E[v]["sd"]["qend"] for v in vs])
            e2 = O.sdp_chain(co2, [E[v]["strand"] for v in vs], sq, st2, sl, O.sdp_opts(L, mode=1, rate=o["second_anchorbonus"], **sdp_kw))
            if e2["status"] < 0:
                return None, False, "ub"
            uq = []; ut = []; ul = []; ucl = []
            if e2["chains"]:
                ix = e2["chains"][0]["frags"].astype(np.int64)
                which = np.searchsorted(co2, ix, side="right") - 1
                keep, _ = O.filter_chain(sq[ix], st2[ix], sl[ix], [E[vs[w]]["strand"] for w in which], None, [1, 3, 4], qend=sqe[ix])
                for i_, w in zip(ix[keep.astype(bool)], which[keep.astype(bool)]):
                    v = int(vs[w]); k = int(i_ - co2[w]); e = E[v]
                    for j in range(int(e["sd"]["end"][k]) - 1, int(e["sd"]["start"][k]) - 1, -1):
                        uq.append(int(e["q"][j])); ut.append(int(e["t"][j])); ul.append(int(e["len"][j])); ucl.append(v)
            aq.extend(uq); at.extend(ut); al.extend(ul); off.append(len(aq))
            cstr.append(E[ucl[0]]["strand"] if ucl else 0); cchr.append(E[ucl[0]]["chrom"] if ucl else 0)
        nch = len(cstr)
        segs = O.local_refine_alignment(off, aq, at, al, cstr, cchr, [h["value"]] * nch, [h["n0"]] * nch, list(np.diff(off)), hi, fwd, rc, gpad, CH, lsc=sp["lsc"], min_anchors=1,
                                        **lra_kw)
        if segs is None:
            return None, False, "ub"
        out = []
        for s in segs:
            sb = fwd if s["strand"] == 0 else rc
            cb = chrom_b(s["chrom"])
            if o["skipBandedRefine"]:
                refined, rst = s["blocks"], 0
            else:
                refined, rst = O.indel_refine(s["blocks"], sb, cb + b"\0" * 64, o["refineBand"], o["match"], o["mismatch"], o["indel"], end_align=True, read_len=L, chrom_len=len(cb))
            d = dict(s, a13_blocks=s["blocks"], blocks=refined, refine_status=rst)
            if stats and rst == 0 and len(refined):
                d["stats1"] = O.calculate_statistics(refined, sb, cb + b"\0" * 64)
            out.append(d)
        if not o["refineBreakpoint"]:                                      # sic: `if (opts.refineBreakpoint == false)` (Map_highacc.h:723)
            for si in range(1, len(out)):
                l, r = out[si], out[si - 1]
                ret, lb, rb_ = O.refine_breakpoint(L, l["blocks"], l["strand"], fwd if l["strand"] == 0 else rc, chrom_b(l["chrom"]), r["blocks"], r["strand"],
                                                   fwd if r["strand"] == 0 else rc, chrom_b(r["chrom"]))
                if ret >= 0:
                    l["blocks"], r["blocks"] = lb, rb_
                l["breakpoint"] = ret
        for d in out:                                                      # the second CalculateStatistics (:730-732)
            if "stats1" in d and len(d["blocks"]):
                c2, v2, runs2, cg2 = O.calculate_statistics(d["blocks"], fwd if d["strand"] == 0 else rc, chrom_b(d["chrom"]) + b"\0" * 64)
                for k in ("tdel", "tins", "nSmallDel", "nMedDel", "nLargeDel", "nSmallIns", "nMedIns", "nLargeIns"):
                    c2[k] += d["stats1"][0][k]
                d["stats"] = (c2, v2, runs2, cg2)
        groups.append(dict(h=hi, segs=out))
    return groups, len(groups) == 0, None
