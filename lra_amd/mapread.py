"""MapRead_lowacc for a batch of reads (reference: Map_lowacc.h:33-640, called from MapRead, MapRead.h:169-263).

The drop-in boundary is the C ABI: lra_map_reads_lowacc_batch (every stage between the read bases and the alignments' statistics, on the
device) and lra_map_records (SetFromSegAlignment, AlignmentsOrder, SimpleMapQV, OUTPUT on the host) in lra_amd/csrc/mapread.hip;
LowAccMapper.align / .records are thin ctypes wrappers over them.  align_staged / records_staged drive the same stages one library call at
a time from Python -- the form the stage-by-stage parity tests hook into; both forms must give identical results (tests/test_mapread.py).

    mapper = LowAccMapper(ctx, genome, idx_key, idx_pos, ["chr1", ...], [0, ..., G])     # once per reference
    res = mapper.align(seed.ReadBatch(ctx, reads))                                       # device work, results stay in HBM
    sam = mapper.records(res, names, reads)                                              # one text record group per read

Stage order and the reference lines each call replaces:
    a1-a4  seed_batch               StoreMinimizers, sort, CompareLists, SeparateMatchesByStrand        MapRead.h:169-203
    a5     clean_matches_batch      CleanMatches per strand                                              Map_lowacc.h:60-150
    a7     linear_extend_batch      LinearExtend of every cluster                                        Map_lowacc.h:160-184
    a8     sparse_dp_batch          SparseDP over the clusters of the read (primary chains)             Map_lowacc.h:185-188
    a9     split_chains_batch       RemoveSpuriousJump, SPLITChain, RemoveSpuriousSplitChain             Map_lowacc.h:189-245
    a10    LocalIndex, refine_splitchain_batch, refine_btwn_splitchain_batch                             Map_lowacc.h:246-410
    a9/a7  merge_extend_batch       MergeChain, LinearExtend (second pass), TrimOverlappedAnchors        Map_lowacc.h:411-520
    a8     sparse_dp_batch          SparseDP on the merged clusters (ultimate chains)                    Map_lowacc.h:521-540
    a13    local_refine_from_sdp    RemovePairedIndels / RemoveSpuriousAnchors, LocalRefineAlignment     Map_lowacc.h:541-576
    a14    indel_refine_batch       IndelRefineAlignment                                                 Map_lowacc.h:582-585
    a16    stats_of_refined         CalculateStatistics                                                  Map_lowacc.h:597-599
    a16/17 records                  SetFromSegAlignment, AlignmentsOrder::Update, SimpleMapQV, OUTPUT    Map_lowacc.h:600-618
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import chain, cluster, emit, local, refine, seed
from .context import Context


@dataclass
class LowAccOptions:
    """The -ONT preset (lra.cpp:386-431 over the defaults of Options.h:127-230)."""
    globalK: int = 17
    globalW: int = 10
    globalMaxFreq: int = 150
    localK: int = 10
    localW: int = 5
    localMaxFreq: int = 15
    localIndexWindow: int = 256
    refineBand: int = 7
    localMatch: int = 4
    localMismatch: int = -1
    localIndel: int = -2
    refineSpaceDist: int = 30000
    anchorstoosparse: float = 0.005
    splitdist: int = 50000            # Options.h:191
    window: int = 100                 # smallOpts.window (Map_lowacc.h:40)
    initial_anchorbonus: float = 20.0
    second_anchorbonus: float = 2.0
    alnthres: float = 0.65
    SecondCleanMaxDiag: int = 100
    bypassClustering: bool = True
    refineBreakpoint: bool = False     # --refineBreakpoints (lra.cpp:262)
    read_type: str = "ont"
    hardClip: bool = True
    PrintNumAln: int = 1
    printFormat: str = "s"
    deferSeedMatches: int = 0          # lra_map_opts.defer_seed_matches (scheduling only: reads with more tier-1 matches are handed back unmapped; 0 = off)
    deferMatches: int = None           # lra_map_opts.defer_matches (scheduling only; None = the preset's value, 0 = one pass)


GLI_K, GLI_W, GLI_WINDOW = 10, 5, 2048
"""What `lra index` writes into the .gli file under every preset (`LocalIndex glIndex;` lra.cpp:989 -> LocalIndex(0): k = 10, w = 5, windows of 1 << (LOCAL_POS_BITS - 1)
bases, MMIndex.h:110-127) and glIndex.Read hands the path (lra.cpp:627): the genome's local index, the reads' (copied from it, Map_lowacc.h:246-247), smallOpts.globalK / W.
The option classes' defaults are `lra align` WITHOUT a .gli file (glIndex built from opts.localK / localIndexWindow = 256, lra.cpp:619-628)."""


def with_gli(opts, k=GLI_K, w=GLI_W, window=GLI_WINDOW):
    """LowAccOptions after glIndex.Read of a .gli file (lra_map_opts_apply_local_index)."""
    import dataclasses
    return dataclasses.replace(opts, localK=k, localW=w, localIndexWindow=window)


def clr_options(**kw):
    """The -CLR preset (lra.cpp:341-386): what differs from -ONT on this path."""
    d = dict(globalK=15, globalMaxFreq=250, refineBand=20, initial_anchorbonus=15.0, second_anchorbonus=6.0, alnthres=0.50, SecondCleanMaxDiag=120,
             read_type="clr")
    d.update(kw)
    return LowAccOptions(**d)


def seq_offsets(chrom_pos, window):
    """LocalIndex::seqOffsets of the genome (MMIndex.h:200-245): window ends, restarting at each sequence."""
    out = [0]
    for s, e in zip(chrom_pos[:-1], chrom_pos[1:]):
        p = int(s)
        while p < e:
            p = min(p + window, int(e))
            out.append(p)
    return np.array(out, np.int64)


def _log_lookup_table():
    """LogLookUpTable.h:9-15 -- logf from the host libm, as the reference builds it."""
    libm = C.CDLL("libm.so.6")
    libm.logf.restype = C.c_float
    libm.logf.argtypes = [C.c_float]
    return np.array([libm.logf(float(i)) for i in range(1, 10002, 5)], dtype=np.float32)


class MapOpts(C.Structure):
    """lra_map_opts (include/lra_hip.h)"""
    _fields_ = ([(n, C.c_int32) for n in ("globalK", "globalW", "globalMaxFreq", "localK", "localW", "localMaxFreq", "localIndexWindow", "refineBand",
                                          "localMatch", "localMismatch", "localIndel", "localBand", "refineSpaceDist")] +
                [("anchorstoosparse", C.c_float), ("splitdist", C.c_int32), ("window", C.c_int32), ("second_anchorbonus", C.c_float),
                 ("bypassClustering", C.c_int32), ("skipBandedRefine", C.c_int32), ("refineBreakpoint", C.c_int32), ("clean", cluster.CleanOpts), ("sdp", chain.SdpOpts)] +
                [(n, C.c_int32) for n in ("readType", "hardClip", "PrintNumAln", "printFormat")] + [("fine", cluster.FineOpts), ("merge_dist", C.c_int32), ("defer_matches", C.c_int32), ("flagged_unaligned", C.c_int32), ("defer_seed_matches", C.c_int32)])


class MapCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_minimizers", "n_matches", "n_clusters", "n_sdp_anchors", "n_sdp_points", "n_sdp_entries", "n_local_tuples",
                                          "n_local_tasks", "n_local_task_words", "n_local_pairs", "n_refined_matches", "n_btwn_problems", "n_btwn_rounds", "n_refined_after_btwn",
                                          "n_merged_clusters", "n_sdp2_anchors", "n_sdp2_entries", "n_a13_blocks", "n_large_spaces", "n_segments", "n_rows",
                                          "n_cells", "n_aog", "n_deferred_reads", "n_flagged_reads", "n_handed_back_reads")]


class MapResult(C.Structure):
    """lra_map_result (include/lra_hip.h)"""
    _fields_ = ([("n_reads", C.c_int32), ("num_aln", C.c_int32), ("n_jobs", C.c_uint64), ("n_alignments", C.c_uint64), ("n_blocks", C.c_uint64),
                 ("n_runs", C.c_uint64)] +
                [(n, C.c_void_p) for n in ("d_job_aln_off", "d_job_status", "d_job_reached", "d_read_status", "d_aln_read", "d_strand", "d_supp", "d_secondary", "d_n0", "d_n1", "d_chrom",
                                           "d_first_sdp_value", "d_block_off", "d_blocks", "d_refine_status", "d_counts", "d_value", "d_run_off", "d_runs",
                                           "d_strands")] +
                [("rc_base", C.c_uint64), ("counters", MapCounters)])


READ_TYPES = {"ont": 0, "clr": 1, "ccs": 2, "contig": 3}

_RC_TABLE = bytearray(b"N" * 256)
for _a, _b in zip(b"ACGTacgtn", b"TGCAtgcan"):
    _RC_TABLE[_a] = _b
_RC_TABLE = bytes(_RC_TABLE)


def create_rc(read: bytes) -> bytes:
    """CreateRC (SeqUtils.h:151-158, RevCompNuc :112-145): what Alignment::read holds for a reverse-strand segment (strands[1])."""
    return read.translate(_RC_TABLE)[::-1]


class MapBatchResult:
    """What align() leaves in HBM for one batch (context-owned buffers: valid until the next align() on the same context)."""
    pass


class LowAccMapper:
    def __init__(self, ctx: Context, genome, idx_key, idx_pos, chrom_names, chrom_pos, opts: LowAccOptions = None, index_params=None, staged=True):
        """genome: uint8 bases of all sequences back to back (numpy or device tensor); idx_key / idx_pos: the global minimizer index (the
        .mms payload, MMIndex.h:416), or None to build it on the device (lra_ctx_build_global_index: StoreIndex with index_params =
        (K, W, globalMaxFreq, globalWinsize, NumOfminimizersPerWindow), default the -ONT index preset with opts.globalK / globalW);
        chrom_pos: n_chrom + 1 start offsets (Genome::header.pos).  staged=False skips the second device copy of the genome that only
        align_staged needs."""
        from . import index as _index
        self.ctx = ctx
        self.opts = opts or LowAccOptions()
        o = self.opts
        dev = ctx.device
        g = genome if torch.is_tensor(genome) else torch.from_numpy(np.ascontiguousarray(genome, dtype=np.uint8))
        self.G = int(g.numel())
        self.chrom_pos = [int(x) for x in chrom_pos]
        assert self.chrom_pos[0] == 0 and self.chrom_pos[-1] == self.G
        self.chrom_names = [n if isinstance(n, bytes) else str(n).encode() for n in chrom_names]
        _index.load_genome(ctx, g)
        if idx_key is None:
            ip = index_params or (o.globalK, o.globalW, 150, 15, 1)
            self.index_stats = _index.build_global_index(ctx, self.chrom_pos, *ip)
        else:
            k = np.ascontiguousarray(idx_key).view(np.uint64); p = np.ascontiguousarray(idx_pos, dtype=np.uint32)
            ctx.check(ctx.lib.lra_ctx_load_global_index(ctx.h, C.c_void_p(k.ctypes.data), C.c_void_p(p.ctypes.data), C.c_uint64(len(k))))
            self.index_stats = dict(n_index=len(k))
        cp = (C.c_uint64 * len(self.chrom_pos))(*self.chrom_pos)
        ctx.check(ctx.lib.lra_ctx_load_chromosomes(ctx.h, cp, len(self.chrom_pos) - 1))
        ctx.check(ctx.lib.lra_ctx_build_local_index(ctx.h, o.localK, o.localW, o.localIndexWindow, o.localMaxFreq))
        self.copts = self._c_opts()
        self.gdev = torch.cat([g.to(dev), torch.zeros(64, dtype=torch.uint8, device=dev)]) if staged else None
        self.g_off = torch.tensor(self.chrom_pos, dtype=torch.int64, device=dev)
        self._gli = None
        self.gso = torch.from_numpy(seq_offsets(self.chrom_pos, o.localIndexWindow)).to(dev) if staged else None
        self.lut = _log_lookup_table()
        self.sdp_opts = chain.sdp_opts(rate=o.initial_anchorbonus, alnthres=o.alnthres, globalK=o.globalK)
        self.sdp2_opts = chain.sdp_opts(mode=1, rate=o.second_anchorbonus, alnthres=o.alnthres, globalK=o.globalK)   # SparseDP :2287, opts.second_anchorbonus
        self.clean_opts = cluster.CleanOpts(globalK=o.globalK, cleanMaxDiag=200, minDiagCluster=3, bypassClustering=int(o.bypassClustering),
                                            cleanClustersize=100, SecondCleanMinDiagCluster=10, SecondCleanMaxDiag=o.SecondCleanMaxDiag, punish_anchorfreq=5,
                                            anchorPerlength=5)
        self.stats = {}

    @property
    def gli(self):
        """The genome's local index as a Python-side object (align_staged only; the C boundary keeps its own in the context)."""
        if self._gli is None:
            o = self.opts
            self._gli = local.LocalIndex(self.ctx, self.gdev, self.g_off, o.localK, o.localW, o.localIndexWindow, o.localMaxFreq)
        return self._gli

    @classmethod
    def sharing(cls, ctx: Context, other: "LowAccMapper"):
        """A mapper on another context of the same GPU that borrows `other`'s reference data (lra_ctx_share_reference): for sub-batches that
        run on their own HIP streams."""
        m = cls.__new__(cls)
        m.ctx = ctx; m.opts = other.opts; m.G = other.G; m.chrom_pos = other.chrom_pos; m.chrom_names = other.chrom_names; m.index_stats = other.index_stats
        ctx.check(ctx.lib.lra_ctx_share_reference(ctx.h, other.ctx.h))
        m.copts = other.copts; m.gdev = None; m.g_off = other.g_off; m._gli = None; m.gso = None; m.lut = other.lut
        m.sdp_opts, m.sdp2_opts, m.clean_opts = other.sdp_opts, other.sdp2_opts, other.clean_opts
        m.stats = {}
        return m

    def fetch_local_index(self):
        """The genome's local index as the context holds it (what lra_ctx_build_local_index built): host arrays (seqOffsets, tupleBoundaries, tuples)."""
        ctx = self.ctx
        res = local.LocalIndexResult(); gso = C.c_void_p()
        ctx.check(ctx.lib.lra_ctx_local_index(ctx.h, C.byref(res), C.byref(gso)))
        nw, nt = int(res.n_windows), int(res.n_tuples)
        return ctx.to_host(gso.value, nw + 1, np.uint64), ctx.to_host(res.d_tuple_bnd, nw + 1, np.uint64), ctx.to_host(res.d_tuples, nt, np.uint32)

    def _c_opts(self):
        o = self.opts
        m = MapOpts()
        self.ctx.lib.lra_map_opts_preset_ont(C.byref(m))
        for n in ("globalK", "globalW", "globalMaxFreq", "localK", "localW", "localMaxFreq", "localIndexWindow", "refineBand", "localMatch", "localMismatch",
                  "localIndel", "refineSpaceDist", "anchorstoosparse", "splitdist", "window", "second_anchorbonus"):
            setattr(m, n, getattr(o, n))
        m.bypassClustering = int(o.bypassClustering); m.refineBreakpoint = int(o.refineBreakpoint)
        m.clean.globalK = o.globalK; m.clean.bypassClustering = int(o.bypassClustering); m.clean.SecondCleanMaxDiag = o.SecondCleanMaxDiag
        m.sdp.globalK = o.globalK; m.sdp.rate = o.initial_anchorbonus; m.sdp.alnthres = o.alnthres
        m.readType = READ_TYPES[o.read_type]; m.hardClip = int(o.hardClip); m.PrintNumAln = o.PrintNumAln; m.printFormat = ord(o.printFormat)
        if o.deferMatches is not None:
            m.defer_matches = int(o.deferMatches)
        m.defer_seed_matches = int(o.deferSeedMatches)
        return m

    # ------------------------------------------------------------------------------------------------------------------ the C boundary
    def align(self, rbatch) -> MapResult:
        """lra_map_reads_lowacc_batch: the whole device side in one library call."""
        ctx = self.ctx
        res = MapResult()
        ctx.check(ctx.lib.lra_map_reads_lowacc_batch(ctx.h, rbatch.n, C.c_void_p(rbatch.seq.data_ptr()), C.c_void_p(rbatch.off.data_ptr()),
                                                     C.c_uint64(int(rbatch.total_bases)), C.byref(self.copts), C.byref(res)))
        c = res.counters
        self.stats.update({n: int(getattr(c, n)) for n, _ in MapCounters._fields_})
        self.stats.update(n_alignments=int(res.n_alignments), n_blocks=int(res.n_blocks), n_cigar_runs=int(res.n_runs),
                          n_mm=int(c.n_minimizers), n_match=int(c.n_matches), n_seg=int(c.n_segments))
        return res

    # ---- two-stage batches (include/lra_hip.h: lra_map_reads_lowacc_front / _back): front(i + 1) on one host thread beside back(i) on another
    def front(self, rbatch):
        ctx = self.ctx
        ctx.check(ctx.lib.lra_map_reads_lowacc_front(ctx.h, rbatch.n, C.c_void_p(rbatch.seq.data_ptr()), C.c_void_p(rbatch.off.data_ptr()),
                                                     C.c_uint64(int(rbatch.total_bases)), C.byref(self.copts)))

    def back(self):
        """-> (MapResult, the back context its arrays belong to: pack / snapshot / records / fetch through a mapper view on it, then release())"""
        ctx = self.ctx
        res = MapResult(); bh = C.c_void_p()
        rc = ctx.lib.lra_map_reads_lowacc_back(ctx.h, C.byref(self.copts), C.byref(res), C.byref(bh))
        if rc:                                                             # (the back halves' thread leaves its error text on the back context: the front thread writes ctx's)
            from ._lib import LraError
            raise LraError("%d: %s" % (rc, ctx.lib.lra_ctx_last_error(C.c_void_p(bh.value) if bh.value else ctx.h).decode()))
        c = res.counters
        self.stats.update({n: int(getattr(c, n)) for n, _ in MapCounters._fields_})
        self.stats.update(n_alignments=int(res.n_alignments), n_blocks=int(res.n_blocks), n_cigar_runs=int(res.n_runs),
                          n_mm=int(c.n_minimizers), n_match=int(c.n_matches), n_seg=int(c.n_segments))
        return res, Context.borrowed(bh.value, ctx.device)

    def release(self):
        rc = self.ctx.lib.lra_map_back_release(self.ctx.h)
        if rc:
            from ._lib import LraError
            raise LraError("%d: no back half's result is held" % rc)

    def on(self, ctx):
        """This mapper's options and host-side tables bound to another context (the back context of two-stage batches)."""
        import copy
        m = copy.copy(self)
        m.ctx = ctx
        return m

    def fetch(self, res: MapResult):
        """Host copies of a batch result: per job the alignment range, per alignment its fields, refined blocks, counters, NV and CIGAR."""
        ctx = self.ctx
        nA, nJ = int(res.n_alignments), int(res.n_jobs)
        d = {"job_aln_off": ctx.to_host(res.d_job_aln_off, nJ + 1, np.uint64) if nJ else np.zeros(1, np.uint64),
             "job_status": ctx.to_host(res.d_job_status, nJ, np.uint32) if nJ else np.zeros(0, np.uint32),
             "job_reached": ctx.to_host(res.d_job_reached, nJ, np.uint8) if nJ else np.zeros(0, np.uint8),
             "read_status": ctx.to_host(res.d_read_status, int(res.n_reads), np.uint32) if int(res.n_reads) else np.zeros(0, np.uint32)}
        for k, dt in (("aln_read", np.uint32), ("strand", np.int32), ("supp", np.int32), ("secondary", np.int32), ("n0", np.int32), ("n1", np.int32),
                      ("chrom", np.int32), ("first_sdp_value", np.float32), ("refine_status", np.int32), ("value", np.float32)):
            p_ = getattr(res, "d_" + k)
            d[k] = ctx.to_host(p_, nA, dt) if nA and p_ else np.zeros(nA, dt)
        d["block_off"] = ctx.to_host(res.d_block_off, nA + 1, np.uint64) if nA else np.zeros(1, np.uint64)
        d["blocks"] = ctx.to_host(res.d_blocks, 3 * int(res.n_blocks), np.int32).reshape(-1, 3) if nA else np.zeros((0, 3), np.int32)
        d["counts"] = ctx.to_host(res.d_counts, 18 * nA, np.int32).reshape(-1, 18) if nA else np.zeros((0, 18), np.int32)
        d["run_off"] = ctx.to_host(res.d_run_off, nA + 1, np.uint64) if nA else np.zeros(1, np.uint64)
        d["runs"] = ctx.to_host(res.d_runs, int(res.n_runs), np.uint32) if nA else np.zeros(0, np.uint32)
        return d

    def block_records(self, res: MapResult):
        """The refined block triples of a batch as a device tensor (what a rank hands to the gather step)."""
        return self.ctx.to_tensor(res.d_blocks, 3 * int(res.n_blocks), torch.int32)

    def records(self, res: MapResult, names, reads, quals=None, passthrough=None):
        """lra_map_records: one bytes object per read in opts.printFormat."""
        ctx = self.ctx
        n = int(res.n_reads)
        enc = lambda x: x if isinstance(x, bytes) else str(x).encode()
        nm = [enc(x) for x in names]; rd = [bytes(x) for x in reads]
        a_names = (C.c_char_p * n)(*nm); a_reads = (C.c_char_p * n)(*rd)
        a_quals = (C.c_char_p * n)(*[None if q is None else bytes(q) for q in quals]) if quals is not None else None
        a_len = (C.c_int32 * n)(*[len(x) for x in rd])
        a_chr = (C.c_char_p * len(self.chrom_names))(*self.chrom_names)
        ln = C.c_uint64(0)
        roff = (C.c_uint64 * (n + 1))()
        args = (ctx.h, C.byref(res), C.byref(self.copts), a_names, a_reads, a_quals, a_len, a_chr, passthrough)
        ctx.check(ctx.lib.lra_map_records(*args, None, C.c_uint64(0), C.byref(ln), roff))
        buf = C.create_string_buffer(ln.value + 1)
        ctx.check(ctx.lib.lra_map_records(*args, buf, C.c_uint64(ln.value), C.byref(ln), roff))
        raw = buf.raw
        return [raw[roff[i]:roff[i + 1]] for i in range(n)]

    def snapshot(self, res: MapResult, with_blocks=False):
        """lra_map_snapshot: host copy of what the records need; afterwards the context may run the next batch."""
        ctx = self.ctx
        h = C.c_void_p()
        ctx.check(ctx.lib.lra_map_snapshot(ctx.h, C.byref(res), 1 if with_blocks else 0, C.byref(h)))
        return h

    def record_args(self, names, reads, quals=None):
        """The per-read host arrays lra_map_records_host takes (built once per batch; keep the returned object alive during the call)."""
        n = len(names)
        enc = lambda x: x if isinstance(x, bytes) else str(x).encode()
        nm = [enc(x) for x in names]; rd = [bytes(x) for x in reads]
        return dict(n=n, keep=(nm, rd), names=(C.c_char_p * n)(*nm), reads=(C.c_char_p * n)(*rd),
                    quals=(C.c_char_p * n)(*[None if q is None else bytes(q) for q in quals]) if quals is not None else None,
                    lens=(C.c_int32 * n)(*[len(x) for x in rd]), chroms=(C.c_char_p * len(self.chrom_names))(*self.chrom_names))

    def records_host(self, snap, args, passthrough=None, n_threads=0, free=True, as_list=True):
        """lra_map_records_host on a snapshot (host threads only; callable from another Python thread while the device runs the next batch:
        ctypes releases the GIL).  -> list of per-read bytes, or the total number of bytes when as_list is False."""
        lib = self.ctx.lib
        text = C.c_char_p(); ln = C.c_uint64(0); roff = C.POINTER(C.c_uint64)()
        rc = lib.lra_map_records_host(snap, C.byref(self.copts), args["names"], args["reads"], args["quals"], args["lens"], args["chroms"], passthrough,
                                      int(n_threads), C.byref(text), C.byref(ln), C.byref(roff))
        if rc != 0:
            lib.lra_map_host_free(snap)
            raise RuntimeError("lra_map_records_host failed (%d)" % rc)
        out = ln.value
        if as_list:
            raw = C.string_at(text, ln.value)
            out = [raw[roff[i]:roff[i + 1]] for i in range(args["n"])]
        if free:
            lib.lra_map_host_free(snap)
        return out

    # ------------------------------------------------------------------------------------------------------------------ the same, stage by stage
    def align_staged(self, rbatch) -> MapBatchResult:
        ctx, o, st = self.ctx, self.opts, self.stats
        CH, G, gdev = self.chrom_pos, self.G, self.gdev
        nR = rbatch.n
        tot = int(rbatch.total_bases)
        lens = rbatch.off[1:] - rbatch.off[:-1]
        sres = seed.seed_batch(ctx, rbatch, o.globalK, o.globalW, o.globalMaxFreq)
        cres = cluster.clean_matches_batch(ctx, self.clean_opts, CH)
        eres = cluster.linear_extend_batch(ctx, o.globalK, rbatch)
        # match_rate = 3 for a read with a repetitive cluster (Map_lowacc.h:86-89, :184-185)
        rate = C.c_void_p()
        ctx.check(ctx.lib.lra_match_rate_batch(ctx.h, C.byref(cres), C.c_float(o.initial_anchorbonus), C.byref(rate)))
        chres = chain.sparse_dp_batch(ctx, nR, cres.d_cluster_off, eres.d_e_start, eres.d_e_count, cres.d_c_strand, eres.d_e_qpos, eres.d_e_tpos,
                                      eres.d_e_len, rbatch.off, self.sdp_opts, rate=rate.value)
        num_aln = int(chres.num_aln)
        # chains[p].NumOfAnchors0 for a13 (the second sparse DP reuses the first one's buffers)
        slot_n0 = ctx.to_tensor(chres.d_chain_len, nR * num_aln, torch.int32) if nR * num_aln else None
        spres = chain.split_chains_batch(ctx, chres, CH, o.splitdist)
        # the reads forward, then reverse complemented, in one buffer (forwardIndex / reverseIndex, Map_lowacc.h:246-250)
        both = torch.zeros(2 * tot + 64, dtype=torch.uint8, device=ctx.device)
        both[:tot] = rbatch.seq[:tot]
        ctx.check(ctx.lib.lra_create_rc_batch(ctx.h, nR, C.c_void_p(rbatch.seq.data_ptr()), C.c_void_p(rbatch.off.data_ptr()), C.c_void_p(both.data_ptr() + tot)))
        off2 = torch.cat([rbatch.off, rbatch.off[1:] + tot]).contiguous()
        rli = local.LocalIndex(ctx, both, off2, o.localK, o.localW, o.localIndexWindow, o.localMaxFreq)
        rres = chain.refine_splitchain_batch(ctx, chres, spres, rbatch.off, CH, rli, self.gso, self.gli, window=o.window, smallK=o.localK, K=o.globalK,
                                             limitrefine=True, max_freq=o.localMaxFreq)
        bres = chain.refine_btwn_splitchain_batch(ctx, chres, spres, rres, rbatch.off, both, tot, gdev, CH, K=o.localK, W=o.localW,
                                                  refineSpaceDist=o.refineSpaceDist, anchorstoosparse=o.anchorstoosparse, match=o.localMatch,
                                                  mismatch=o.localMismatch, indel=o.localIndel, max_freq=o.localMaxFreq)
        mres = chain.merge_extend_batch(ctx, chres, spres, bres, rbatch.seq, rbatch.off, gdev, CH, K=o.localK)
        # which primary chains reach Map_lowacc.h:574 (split chains left, refined clusters not all empty); before the second sparse DP, which
        # reuses the first one's result arrays
        n_slots = nR * num_aln
        if n_slots:
            nsp = ctx.to_tensor(spres.d_n_split, n_slots, torch.int32).to(torch.int64)
            spst = ctx.to_tensor(spres.d_status, n_slots, torch.int32)
            cs = ctx.to_tensor(chres.d_chain_start, n_slots, torch.int64)
            nch = ctx.to_tensor(chres.d_n_chains, nR, torch.int32).to(torch.int64)
            exists = (torch.arange(n_slots, device=ctx.device) % num_aln) < torch.repeat_interleave(nch, num_aln)
            moff = ctx.to_tensor(bres.d_match_off, int(bres.n_frags) + 1, torch.int64)
            ok = exists & (spst == 0) & (nsp > 0)
            cs = torch.where(ok, cs, torch.zeros_like(cs)); nsp = torch.where(ok, nsp, torch.zeros_like(nsp))
            job_reached = (ok & (moff[cs + nsp] > moff[cs])).cpu().numpy()
        else:
            job_reached = np.zeros(0, bool)
        ch2 = chain.sparse_dp_batch(ctx, int(mres.n_groups), mres.d_iota, mres.d_anchor_off, mres.d_count, mres.d_strand, mres.d_q, mres.d_t, mres.d_len,
                                    mres.d_iota, self.sdp2_opts)
        st.update(n_btwn_problems=bres.n_problems, n_btwn_rounds=bres.n_rounds, n_refined_after_btwn=bres.n_matches, n_merged_clusters=mres.n_groups,
                  n_sdp2_anchors=mres.n_anchors, n_sdp2_entries=ch2.n_subproblem_entries)
        if "n_local_task_words" not in st and rres.n_tasks:
            t4 = [ctx.to_tensor(p_, rres.n_tasks, torch.int64) for p_ in (rres.d_task_q_lo, rres.d_task_q_hi, rres.d_task_t_lo, rres.d_task_t_hi)]
            st["n_local_task_words"] = int((t4[1] - t4[0]).sum() + (t4[3] - t4[2]).sum())
        inp, ares = chain.local_refine_from_sdp(ctx, num_aln, slot_n0, mres, ch2, rbatch.off, both, tot, gdev, CH)
        nA, nJ = int(ares.n_alignments), int(ares.n_jobs)
        aoff = ctx.to_tensor(ares.d_job_aln_off, nJ + 1, torch.int64)
        aln_job = torch.repeat_interleave(torch.arange(nJ, device=ctx.device), aoff[1:] - aoff[:-1])
        aln_read = aln_job // max(num_aln, 1)
        a_strand = ctx.to_tensor(ares.d_strand, nA, torch.int32).to(torch.int64)
        a_chrom = ctx.to_tensor(ares.d_chrom, nA, torch.int32).to(torch.int64)
        fb = refine.refine_batch_from_device(ctx, ctx.to_tensor(ares.d_blocks, 3 * int(ares.n_blocks), torch.int32).view(-1, 3),
                                             ctx.to_tensor(ares.d_block_off, nA + 1, torch.int64), both, rbatch.off[aln_read] + a_strand * tot, lens[aln_read],
                                             gdev, self.g_off[a_chrom], self.g_off[a_chrom + 1] - self.g_off[a_chrom])
        fres = refine.indel_refine_batch(ctx, fb, o.refineBand, o.localMatch, o.localMismatch, o.localIndel)
        refine_status = ctx.to_tensor(fres.d_status, nA, torch.int32)      # lives in scratch the next stage reuses
        tres = refine.stats_of_refined(ctx, fb, fres, self.lut)
        st.update(n_alignments=nA, n_a13_blocks=int(ares.n_blocks), n_large_spaces=int(ares.n_big),
                  n_mm=sres.n_minimizers, n_match=sres.n_matches, n_cells=fres.n_cells, n_rows=fres.n_rows,
                  n_seg=fres.n_segments, n_blocks=fres.n_blocks, n_aog=fres.n_aog, n_clusters=cres.n_clusters, n_cigar_runs=tres.n_runs, n_local_tuples=rli.n_tuples,
                  n_local_tasks=rres.n_tasks, n_local_pairs=rres.n_pairs, n_refined_matches=rres.n_matches,
                  n_sdp_anchors=chres.n_frags, n_sdp_points=chres.n_points, n_sdp_entries=chres.n_subproblem_entries)
        r = MapBatchResult()
        r.n_reads, r.num_aln, r.n_alignments, r.n_jobs = nR, num_aln, nA, nJ
        r.alignments, r.refined, r.stat, r.refine_status = ares, fres, tres, refine_status
        r.aln_job, r.aln_read, r.job_aln_off, r.job_reached = aln_job, aln_read, aoff, job_reached
        r.refine_batch, r.strands, r.rc_base = fb, both, tot
        # the refined block triples: what a rank hands to the gather step
        r.block_records = ctx.to_tensor(fres.d_blocks, 3 * fres.n_blocks, torch.int32)
        return r

    # ------------------------------------------------------------------------------------------------------------------ records
    def records_staged(self, res: MapBatchResult, names, reads, quals=None, passthrough=None):
        """Per read: SetFromSegAlignment -> AlignmentsOrder::Update -> SimpleMapQV -> OUTPUT (Map_lowacc.h:600-618), or output_unaligned
        when its first primary chain produced no alignment (:578-581, :604-607).  names / reads / quals: per-read bytes.  Returns one bytes
        object per read in opts.printFormat ('s' SAM, 'p' / 'P' PAF, 'b' BED; 'a' pairwise only through records())."""
        ctx, o = self.ctx, self.opts
        nA, na = res.n_alignments, max(res.num_aln, 1)
        counts, value, cigars = refine.fetch_stats(ctx, res.stat)
        al = chain.fetch_alignments(ctx, res.alignments)
        rb_off = ctx.to_host(res.refined.d_block_off, nA + 1, np.uint64) if nA else np.zeros(1, np.uint64)
        rblocks = ctx.to_host(res.refined.d_blocks, 3 * int(res.refined.n_blocks), np.int32).reshape(-1, 3) if nA else np.zeros((0, 3), np.int32)
        jo = al["job_aln_off"].astype(np.int64)
        ix = {n: i for i, n in enumerate(refine.STAT_NAMES)}
        out = []
        for r in range(res.n_reads):
            name = names[r] if isinstance(names[r], bytes) else str(names[r]).encode()
            rd = bytes(reads[r])
            rd_rc = None
            ql = None if quals is None else quals[r]
            recs, seg_off = [], [0]
            unaligned = int(jo[r * na + 1] - jo[r * na]) == 0 if res.n_jobs else True                   # p == 0 left no SegAlignment (:578)
            if not unaligned:
                for p in range(na):
                    j = r * na + p
                    if not res.job_reached[j]:
                        break                                                  # Map_lowacc.h:267, :491: the loop over p ends
                    # (a chain that reaches :574 keeps its SegAlignmentGroup even when it is empty)
                    for a in range(int(jo[j]), int(jo[j + 1])):
                        c = counts[a]
                        rec = emit.AlnRecord()
                        if int(al["strand"][a]) and rd_rc is None:
                            rd_rc = create_rc(rd)
                        # Alignment::read = strands[str] (Map_lowacc.h:560, Alignment.h:506-507): a reverse-strand record's SEQ is the reverse complement
                        rec.read_name, rec.read, rec.qual, rec.read_len = name, (rd_rc if int(al["strand"][a]) else rd), ql, len(rd)
                        ci = int(al["chrom"][a])
                        rec.chrom = self.chrom_names[ci]
                        rec.genome_len = self.chrom_pos[ci + 1] - self.chrom_pos[ci]
                        rec.cigar = cigars[a].encode()
                        rec.flag, rec.strand, rec.mapqv = 0, int(al["strand"][a]), 0
                        rec.supplementary, rec.typeofaln, rec.is_secondary = int(al["supp"][a]), 0, int(al["secondary"][a])
                        rec.q_start, rec.q_end, rec.t_start, rec.t_end = (int(c[ix[k]]) for k in ("qStart", "qEnd", "tStart", "tEnd"))
                        rec.pre_clip, rec.suf_clip = int(c[ix["preClip"]]), int(c[ix["sufClip"]])
                        for k in ("nm", "nmm", "nins", "ndel", "tdel", "tins", "nSmallDel", "nMedDel", "nLargeDel", "nSmallIns", "nMedIns", "nLargeIns"):
                            setattr(rec, k, int(c[ix[k]]))
                        rec.value, rec.order, rec.runtime = float(al["value"][a]), 0, 0
                        rec.NumOfAnchors0, rec.NumOfAnchors1 = int(al["n0"][a]), int(al["n1"][a])
                        b = rblocks[int(rb_off[a]):int(rb_off[a + 1])]
                        rec.n_blocks = len(b)
                        rec.first_block_qpos = int(b[0, 0]) if len(b) else 0
                        rec.last_block_qend = int(b[-1, 0] + b[-1, 2]) if len(b) else 0
                        recs.append(rec)
                    seg_off.append(len(recs))
            if unaligned or not recs:
                un = emit.AlnRecord()
                un.read_name, un.read, un.qual, un.read_len = name, rd, ql, len(rd)
                text, _, _, _ = emit.finish_read([], [0], fmt=o.printFormat, unaligned=un)
            else:
                text, _, _, _ = emit.finish_read(recs, seg_off, bypass_clustering=o.bypassClustering, read_type=o.read_type, globalK=o.localK,
                                                 print_num_aln=o.PrintNumAln, fmt=o.printFormat, hard_clip=o.hardClip, passthrough=passthrough)
            out.append(text)
        return out

    def sam_header(self, version=b"lra_amd", command_line=b""):
        names = (C.c_char_p * len(self.chrom_names))(*self.chrom_names)
        pos = (C.c_uint64 * len(self.chrom_pos))(*self.chrom_pos)
        return emit._call("lra_format_sam_header", version, command_line, names, pos, len(self.chrom_names))


class HighAccMapper:
    """MapRead_highacc behind the C boundary (lra_map_reads_highacc_batch): the -CCS / -CONTIG presets.  Reference data as LowAccMapper (the global index is
    built on the device from index_params = (K, W, globalMaxFreq, globalWinsize, NumOfminimizersPerWindow) when idx_key is None); no local index."""

    def __init__(self, ctx: Context, genome, idx_key, idx_pos, chrom_names, chrom_pos, preset="ccs", index_params=None, gli=None, **overrides):
        """gli = (k, w, window): the genome's local index as a .gli file holds it (True: what `lra index` writes, 10 / 5 / 2048); None: built from the options, as
        `lra align` does without a .gli file (localK = 7, windows of 256 bases)."""
        from . import index as _index
        self.ctx = ctx
        g = genome if torch.is_tensor(genome) else torch.from_numpy(np.ascontiguousarray(genome, dtype=np.uint8))
        self.G = int(g.numel())
        self.chrom_pos = [int(x) for x in chrom_pos]
        self.chrom_names = [n if isinstance(n, bytes) else str(n).encode() for n in chrom_names]
        m = MapOpts()
        (ctx.lib.lra_map_opts_preset_contig if preset == "contig" else ctx.lib.lra_map_opts_preset_ccs)(C.byref(m))
        if gli:
            ctx.lib.lra_map_opts_apply_local_index(C.byref(m), *((GLI_K, GLI_W, GLI_WINDOW) if gli is True else gli))
        for k, v in overrides.items():
            obj = m
            *path, leaf = k.split(".")
            for part in path:
                obj = getattr(obj, part)
            setattr(obj, leaf, v)
        self.copts = m
        _index.load_genome(ctx, g)
        if idx_key is None:
            # `lra index -CCS` / `-CONTIG` (lra.cpp:884-896): K 17 / 19, W 10, maxFreq 150 / 30, window 15 / 20, one minimizer per window
            ip = index_params or ((m.globalK, 10, 150, 15, 1) if preset != "contig" else (m.globalK, 10, 30, 20, 1))
            self.index_stats = _index.build_global_index(ctx, self.chrom_pos, *ip)
        else:
            k = np.ascontiguousarray(idx_key).view(np.uint64); p = np.ascontiguousarray(idx_pos, dtype=np.uint32)
            ctx.check(ctx.lib.lra_ctx_load_global_index(ctx.h, C.c_void_p(k.ctypes.data), C.c_void_p(p.ctypes.data), C.c_uint64(len(k))))
            self.index_stats = dict(n_index=len(k))
        cp = (C.c_uint64 * len(self.chrom_pos))(*self.chrom_pos)
        ctx.check(ctx.lib.lra_ctx_load_chromosomes(ctx.h, cp, len(self.chrom_pos) - 1))
        ctx.check(ctx.lib.lra_ctx_build_local_index(ctx.h, m.localK, m.localW, m.localIndexWindow, m.localMaxFreq))     # glIndex: only the REFINEclusters branch reads it
        self.stats = {}

    def align(self, rbatch) -> MapResult:
        ctx = self.ctx
        res = MapResult()
        ctx.check(ctx.lib.lra_map_reads_highacc_batch(ctx.h, rbatch.n, C.c_void_p(rbatch.seq.data_ptr()), C.c_void_p(rbatch.off.data_ptr()),
                                                      C.c_uint64(int(rbatch.total_bases)), C.byref(self.copts), C.byref(res)))
        c = res.counters
        self.stats.update({n: int(getattr(c, n)) for n, _ in MapCounters._fields_})
        self.stats.update(n_alignments=int(res.n_alignments), n_blocks=int(res.n_blocks), n_cigar_runs=int(res.n_runs))
        return res

    fetch = LowAccMapper.fetch
    fetch_local_index = LowAccMapper.fetch_local_index
    records = LowAccMapper.records
    record_args = LowAccMapper.record_args
    snapshot = LowAccMapper.snapshot
    records_host = LowAccMapper.records_host
