/* include/lra_hip.h -- C ABI of liblra_hip.so: the MI355X (gfx950) replacement for the
 * per-read alignment hot path of ChaissonLab/LRA (MapRead.h:153-263 and what it calls).
 *
 * The reference has no FFI seam: every stage is a C++ function template #included into
 * lra.cpp.  Each entry point below replaces ONE of those functions for a whole BATCH of
 * reads / sub-problems (the GPU needs thousands at once), and cites the reference
 * function it replaces.  A maintainer binds them from MapRead.h as shown in
 * INTEGRATION.md.
 *
 * Conventions
 *   - plain C types only; no torch / STL types cross the boundary;
 *   - pointers named d_* are DEVICE (HBM) pointers, h_* are host pointers;
 *   - every call is asynchronous on the context's HIP stream (lra_ctx_set_stream) unless
 *     it says "synchronous"; the caller synchronises the stream before reading results;
 *   - return value: 0 = ok, negative = error (lra_ctx_last_error gives the text); the
 *     library never calls exit() (the reference exit(1)s, lra.cpp:623-640);
 *   - per-item `status` words report conditions that are undefined behaviour or an
 *     endless loop in the reference (so no parity is defined for them).
 *
 * Lifetime of results.  Output arrays marked "context-owned" live in growable buffers of
 * the context and stay valid until a later call reuses their buffer; a buffer that has to
 * grow is freed and allocated again (contents are not preserved).  Which calls share
 * buffers (copy a result out with lra_copy_device if it has to survive one of them):
 *   - every sparse DP call (lra_sparse_dp_batch, lra_sparse_dp_boxes_batch, and the ones
 *     inside lra_local_refine_batch / _highacc_batch) overwrites the previous sparse DP's
 *     lra_chain_result;
 *   - lra_indel_refine_batch and lra_calculate_statistics_batch keep their temporaries in
 *     the sparse DP's arena: a chain result does not survive them either way, their own
 *     results (refined blocks, counters, CIGAR runs) have buffers of their own;
 *   - lra_indel_refine_batch's d_status lives in scratch the statistics stage reuses;
 *   - lra_filter_chains_batch shares its buffer with lra_split_chains_batch,
 *     lra_refine_clusters_batch with lra_refine_splitchain_batch;
 *   - a lra_map_reads_*_batch call invalidates every result of the call before it (and a
 *     pending two-call lra_map_records text); lra_map_snapshot / lra_map_pack take what the
 *     record stage needs out of the context first.
 * The two drivers are the tested orderings of these calls.
 */
#ifndef LRA_HIP_H_
#define LRA_HIP_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lra_ctx lra_ctx;

#define LRA_OK 0
#define LRA_ERR_INVALID (-1)
#define LRA_ERR_HIP (-2)
#define LRA_ERR_NOMEM (-3)

/* per-item status bits */
#define LRA_ST_OOB_SLOT 1      /* reference would index outside its matrices            */
#define LRA_ST_NO_TERMINATION 2 /* reference trace back would loop forever              */
#define LRA_ST_RANGE 4         /* problem too large for the 32-bit device score range   */
#define LRA_ST_CAPACITY 8      /* caller-provided output capacity exceeded              */
#define LRA_ST_REJECTED 16     /* the reference drops the item itself (e.g. a cluster spanning two chromosomes) */
#define LRA_ST_UNSUPPORTED 32  /* the read takes a branch of the reference this library has not built (it gets no record) */
#define LRA_ST_DEFERRED 64     /* scheduling, not an error: with opts.defer_seed_matches the read was handed back unmapped (the caller maps it in a later batch of
                                * its own kind; no record is written for it here).  Also used inside lra_map_reads_lowacc_batch for opts.defer_matches (never in a result) */

/* ---- context ---------------------------------------------------------------------- */
int lra_ctx_create(int device_id, lra_ctx** out);
void lra_ctx_destroy(lra_ctx* ctx);
/* (ABI 9) Frees the context's growable WORK buffers -- its own, its companion contexts' (a second pass, the back half and the handover sets of two-stage batches) -- and
 * keeps everything loaded into it (genome, chromosome table, both indexes).  The buffers grow to what the largest batch needed and are kept
 * from call to call (a batch of 28672 reads of 30 kb holds ~250 GB of them); a caller that wants the memory back -- after a call failed with LRA_ERR_NOMEM and before it
 * retries with a smaller batch, or between jobs -- calls this.  No batch may be in flight on the context (LRA_ERR_INVALID while a batch sits between the halves of a
 * two-stage batch), and every result of an earlier call (lra_map_result and friends: device pointers into these buffers) is void afterwards.  The reference this
 * boundary replaces has no counterpart: its buffers are the process heap's.  Returns the bytes freed through *bytes (may be NULL).                                  */
int lra_ctx_release_buffers(lra_ctx* ctx, uint64_t* bytes);
/* stream = a hipStream_t (NULL = the default stream).  All later calls launch on it. */
int lra_ctx_set_stream(lra_ctx* ctx, void* stream);
const char* lra_ctx_last_error(lra_ctx* ctx);
/* ABI version of the loaded library (tests check it against this header). */
int lra_abi_version(void);
#define LRA_ABI_VERSION 9   /* 9: lra_ctx_release_buffers; 8: lra_map_opts_apply_local_index, lra_ctx_local_index_params, lra_ctx_load_local_index (the .gli file's k / w / window override the options', as glIndex.Read does); 7: lra_sort_pairs_batch;  2: lra_map_opts.defer_matches, lra_map_counters.n_deferred_reads; 3: lra_map_opts.flagged_unaligned, lra_map_counters.n_flagged_reads, lra_map_host_flagged; 4: lra_reads_last_error, a corrupt FASTQ record is LRA_ERR_INVALID; lra_map_opts.defer_seed_matches; 5: lra_seed_prefetch, lra_ctx_adopt_seed, lra_map_reads_lowacc_front / _back, lra_map_back_release; 6: a failed front half hands over an error batch (one back call per front call), separate n_handed_back_reads counter, lra_map_host_trim */

/* Convenience for hosts without their own HIP binding: synchronous device->host copy on the
 * context's stream (a C++ host would call hipMemcpy itself).                                */
int lra_copy_to_host(lra_ctx* ctx, void* h_dst, const void* d_src, uint64_t bytes);

/* Asynchronous device->device copy on the context's stream (e.g. out of context-owned result
 * arrays into caller-owned memory before the next call recycles them).                        */
int lra_copy_device(lra_ctx* ctx, void* d_dst, const void* d_src, uint64_t bytes);

/* Per-kernel device timing (HIP events on the context's stream around every kernel the
 * library launches).  Off by default.  lra_ctx_timing_get synchronises the stream and returns
 * the accumulated milliseconds and launch count of the named kernel since the last reset
 * (names: see DESIGN.md "kernels"); returns LRA_ERR_INVALID for an unknown name.            */
int lra_ctx_timing_enable(lra_ctx* ctx, int on);
int lra_ctx_timing_reset(lra_ctx* ctx);
int lra_ctx_timing_get(lra_ctx* ctx, const char* name, double* total_ms, int* launches);

/* ---- reference data (replicated per GPU) ---------------------------------------------
 * Genome: the concatenated upper-case chromosome bytes, chromosome c at global offset
 * header.pos[c] (Genome.h:59-68, Genome::Read :115-138 upper-cases).  Replaces the role of
 * Genome::seqs / GlobalIndexToSeq (Genome.h:106-112) for the device side.  Synchronous copy. */
int lra_ctx_load_genome(lra_ctx* ctx, const char* h_seq, uint64_t len);
/* Global minimizer index = the payload of `ref.mms` (MMIndex.h:402-424): n GenomeTuples
 * sorted by (t & 2^63-1), passed as two host arrays (t, pos).  Replaces `genomemm`
 * (MapRead.h:153).  Synchronous copy.                                                       */
int lra_ctx_load_global_index(lra_ctx* ctx, const uint64_t* h_key, const uint32_t* h_pos, uint64_t n);
/* The same two from device memory / built on the device:
 *   lra_ctx_load_genome_device  copies len bytes from a device buffer.
 *   lra_ctx_build_global_index  `lra index` on the loaded genome: StoreIndex (MMIndex.h:286-400) -- per chromosome
 *       StoreMinimizers<GenomeTuple,Tuple>(seq, len, k, w) (MinCount.h:8-179), sort by masked key, keys occurring more than
 *       max_freq (opts.globalMaxFreq) times dropped, then at most n_per_window (opts.NumOfminimizersPerWindow) entries per window of
 *       winsize (opts.globalWinsize) bases in CountSort's order (:258-283: frequency ascending, sorted position descending) -- and
 *       installs the result as the context's global index.  Index-time presets (lra.cpp:884-911): -ONT / -CCS 17, 10, 150, 15, 1;
 *       -CLR 15, 10, 250, 12, 1; -CONTIG 19, 10, 30, 20, 1.  h_chrom_pos: n_chrom + 1 cumulative starts (Genome::header.pos).
 *       The entries and their key order are StoreIndex's; equal keys keep their emission order (the reference: libstdc++'s std::sort
 *       permutation), which also decides between two equal-key candidates inside one window.  *status = LRA_ST_OOB_SLOT when the
 *       reference would index winCount out of range (:359-366).  Synchronous.
 *   lra_ctx_global_index        the context's index as device arrays (for lra_write_mms after a copy to the host).                 */
int lra_ctx_load_genome_device(lra_ctx* ctx, const char* d_seq, uint64_t len);
int lra_ctx_build_global_index(lra_ctx* ctx, const uint64_t* h_chrom_pos, int n_chrom, int k, int w, int max_freq, int winsize, int n_per_window,
                               uint64_t* n_minimizers, uint64_t* n_index, int* status);
int lra_ctx_global_index(lra_ctx* ctx, const uint64_t** d_key, const uint32_t** d_pos, uint64_t* n);
/* Index files (host arrays, host I/O), byte layouts of the reference:
 *   .mms  WriteIndex / ReadIndex (MMIndex.h:402-424; Header::Write / Read Genome.h:59-84): int64 n; int32 globalK; int32 nChrom; per
 *         chromosome int32 nameLen + name bytes; uint64 pos[nChrom + 1]; n GenomeTuples of 16 bytes {uint64 t; uint32 pos; 4 bytes of
 *         padding (zeros here)}.  lra_read_mms: first call with key == NULL returns *n, *globalK, *n_chrom, *names_len (bytes for the
 *         names, each NUL-terminated); the second call fills names, chrom_pos[n_chrom + 1], key[n], pos[n].
 *   .gli  LocalIndex::Write / Read (MMIndex.h:138-173): int32 k, w, localIndexWindow, nRegions = n_windows + 1; uint64 seqOffsets[nRegions];
 *         uint64 tupleBoundaries[nRegions]; uint64 nMin; nMin LocalTuples (uint32: t in the low 20 bits, pos in the high 12).
 *         lra_read_gli: first call with seq_offsets == NULL returns the sizes.  The filling calls of both readers take the values the sizing call
 *         left in *n / *n_chrom / *names_len (*n_windows / *n_tuples) as the capacities of the caller's buffers and fail when the file holds more.      */
int lra_write_mms(const char* path, int globalK, const char* const* chrom_names, const uint64_t* chrom_pos, int n_chrom, const uint64_t* key,
                  const uint32_t* pos, uint64_t n);
int lra_read_mms(const char* path, int* globalK, uint64_t* n, int* n_chrom, uint64_t* names_len, char* names, uint64_t* chrom_pos, uint64_t* key,
                 uint32_t* pos);
int lra_write_gli(const char* path, int k, int w, int window, uint64_t n_windows, const uint64_t* seq_offsets, const uint64_t* tuple_bnd,
                  const uint32_t* tuples);
int lra_read_gli(const char* path, int* k, int* w, int* window, uint64_t* n_windows, uint64_t* n_tuples, uint64_t* seq_offsets, uint64_t* tuple_bnd,
                 uint32_t* tuples);

/* ---- a1-a4: tier-1 seeding of a read batch ---------------------------------------------
 * Replaces, per read (MapRead.h:169-203):
 *   StoreMinimizers<GenomeTuple,Tuple>(read.seq, read.length, k, w, readmm, true)  MinCount.h:8
 *   sort(readmm.begin(), readmm.end())                                            MapRead.h:185
 *   CompareLists<GenomeTuple,Tuple>(readmm, genomemm, allMatches, opts, true)     CompareLists.h:9
 *   SeparateMatchesByStrand(read, genome, k, allMatches, forMatches, revMatches)  MapRead.h:109
 * Input: n_reads upper-case ASCII reads concatenated in d_seq, read r = bytes
 * [d_read_off[r], d_read_off[r+1]).  k = opts.globalK, w = opts.globalW,
 * max_freq = opts.globalMaxFreq.
 * Output (device arrays owned by the context, valid until the next lra_seed_batch call on
 * it; CSR by read):
 *   d_mm_*     readmm after the sort (t with the strand flag in bit 63, pos)
 *   d_match_*  allMatches in the reference's order: (index into the read's readmm,
 *              index into the global index) per pair
 *   d_sep_*    forMatches then revMatches of each read (read pos, genome pos of each pair),
 *              d_n_forward[r] = forMatches.size()
 * Synchronous: returns after the results are complete (two host round trips size the
 * outputs).
 * Lifetime of OTHER results on the same context: the sketch is staged in the context's sparse-DP
 * arena -- the allocation that also holds the IndelRefine / CalculateStatistics arrays of the last
 * lra_map_result.  A stage call on a context (this one, lra_sparse_dp_batch, ...) therefore ends the
 * lifetime of the map result of an earlier batch call on THAT context: take what is needed from it
 * first (lra_map_pack / lra_map_snapshot / lra_map_records), or seed on another context
 * (lra_seed_prefetch).                                                                       */
typedef struct lra_seed_result {
  int32_t n_reads;
  uint64_t n_minimizers, n_matches;
  const uint64_t* d_mm_off;    /* [n_reads+1] */
  const uint64_t* d_mm_key;    /* [n_minimizers] */
  const uint32_t* d_mm_pos;    /* [n_minimizers] */
  const uint64_t* d_match_off; /* [n_reads+1] */
  const uint32_t* d_match_qi;  /* [n_matches] */
  const uint32_t* d_match_ti;  /* [n_matches] */
  const uint32_t* d_n_forward; /* [n_reads] */
  const uint32_t* d_sep_qpos;  /* [n_matches] */
  const uint32_t* d_sep_tpos;  /* [n_matches] */
} lra_seed_result;
int lra_seed_batch(lra_ctx* ctx, int n_reads, const char* d_seq, const uint64_t* d_read_off, int k, int w,
                   int max_freq, lra_seed_result* out);

/* The seeding of the NEXT batch beside the current one.  The per-read path is a chain of launches that are as long as their largest reads: much of a batch's time the
 * device has room, and a1-a4 of the batch after it fit there.  `side` is a second context of the same device that shares the mapping context's reference data
 * (lra_ctx_share_reference), driven by a host thread of its own on its own (low-priority) stream: lra_seed_prefetch is lra_seed_batch on it, synchronous, the result
 * kept by `side`.  lra_ctx_adopt_seed(ctx, side) then hands that result to the mapping context (the two contexts exchange their seed-stage batch buffers; no copy);
 * the next lra_map_reads_lowacc_batch / lra_map_reads_highacc_batch on `ctx` with the same n_reads, d_seq, d_read_off and the same globalK / globalW / globalMaxFreq
 * starts from it instead of seeding -- with any other arguments it seeds as usual and the adopted result is dropped.  Same alignments either way: scheduling only.
 * The reads (d_seq, d_read_off) must stay untouched from the prefetch to the batch call.  Not combined with opts.defer_seed_matches (LRA_ERR_INVALID).
 * Replaces nothing in the reference: lra's worker threads (lra.cpp:678-714) each run MapRead start to end; this is the device's way of having two reads in flight. */
int lra_seed_prefetch(lra_ctx* side, int n_reads, const char* d_seq, const uint64_t* d_read_off, int k, int w, int max_freq);
int lra_ctx_adopt_seed(lra_ctx* ctx, lra_ctx* side);

/* CreateRC (SeqUtils.h:151): reverse complement of every read of the batch into d_rc (same offsets);
 * bytes other than ACGTacgtn become 'N' (RevCompNuc, SeqUtils.h:112).  Asynchronous.            */
int lra_create_rc_batch(lra_ctx* ctx, int n_reads, const char* d_seq, const uint64_t* d_read_off, char* d_rc);

/* a2 alone: sorts each list [d_off[i], d_off[i+1]) of (key,pos) tuples IN PLACE exactly as
 * `std::sort(readmm.begin(), readmm.end())` (MapRead.h:185, libstdc++ introsort with
 * GenomeTuple::operator<, TupleOps.h:76) would, including the order it leaves equal keys in.
 * Asynchronous.                                                                               */
int lra_sort_minimizers_batch(lra_ctx* ctx, int n_lists, const uint64_t* d_off, uint64_t* d_key, uint32_t* d_pos);

/* The path's other sorts -- DiagonalSort / AntiDiagonalSort / CartesianSort of a read's (a cluster's, a gap's) matches (Sorting.h:50-150; called from
 * Clustering.h:1840 CleanMatches, ChainRefine.h:767 MergeChain's LinearExtend, LocalRefineAlignment.h:364), the sparse DP's point orders where no two
 * points of a list share a key (SparseDP.h:2171-2174), its diagonal order -- as one primitive: every segment [d_begin[i], d_end[i]) of (64-bit key,
 * 32-bit value) pairs sorted by the key bits [begin_bit, end_bit), STABLY (pairs with equal bits keep their input order: with the packed keys the
 * callers build, that is what std::sort leaves wherever the reference's comparator has no ties, and the callers send the lists with ties through
 * lra_sort_minimizers_batch), from (d_key_in, d_val_in) into (d_key_out, d_val_out); segments may be empty, need not be adjacent and must not overlap.
 * A segment of 257 .. 8192 pairs is sorted by one workgroup in LDS, the others by rocprim's segmented radix sort.  Asynchronous (the context's
 * stream); the work buffer is the context's.                                                                                                 */
int lra_sort_pairs_batch(lra_ctx* ctx, uint64_t n_pairs, uint64_t n_segments, const uint64_t* d_begin, const uint64_t* d_end, const uint64_t* d_key_in,
                         uint64_t* d_key_out, const uint32_t* d_val_in, uint32_t* d_val_out, int begin_bit, int end_bit);

/* ---- a5: match cleaning + diagonal clusters ----------------------------------------------
 * Replaces, per read and strand,   CleanMatches(Matches, clusters, genome, read, opts, timing, ma_strand)
 * (Clustering.h:1840) in the configuration every preset uses (opts.ExtractDiagonalFromClean, lra.cpp:268-431):
 * DiagonalSort / AntiDiagonalSort (Sorting.h:50,113) -> CleanOffDiagonal (Clustering.h:566-798, incl.
 * AVGfreq :550 and SecondRoundCleanOffDiagonal :802-868) -> one Cluster per surviving diagonal run
 * (SetClusterBoundariesFromMatches :308, chromIndex = header.Find(tStart) Genome.h:20).
 * Input: the forMatches / revMatches of lra_seed_batch (the context's current seed result).
 * Output (device arrays owned by the context, CSR by read; forward-strand clusters first, as
 * MapRead_lowacc builds `clusters`, Map_lowacc.h:77-78):
 *   clusters: per cluster start/end into the cleaned match arrays, q/t box, strand, chromIndex, anchorfreq
 *   cl_qpos / cl_tpos: the cleaned, diagonal-sorted matches (genome coordinates)                */
typedef struct lra_clean_opts {
  int32_t globalK, cleanMaxDiag, minDiagCluster, bypassClustering, cleanClustersize;
  int32_t SecondCleanMinDiagCluster, SecondCleanMaxDiag, punish_anchorfreq, anchorPerlength;
} lra_clean_opts;
typedef struct lra_cluster_result {
  int32_t n_reads;
  uint64_t n_clusters, n_matches;
  const uint64_t* d_cluster_off;  /* [n_reads+1] */
  const uint64_t* d_c_start;      /* [n_clusters] first match of the cluster (index into d_cl_*) */
  const uint64_t* d_c_end;        /* [n_clusters] one past its last match */
  const uint32_t* d_c_qStart; const uint32_t* d_c_qEnd; const uint32_t* d_c_tStart; const uint32_t* d_c_tEnd;
  const int32_t* d_c_strand; const int32_t* d_c_chrom; const float* d_c_anchorfreq;
  const uint32_t* d_cl_qpos; const uint32_t* d_cl_tpos;   /* [n_matches] */
} lra_cluster_result;
int lra_clean_matches_batch(lra_ctx* ctx, const lra_clean_opts* opts, const uint64_t* h_chrom_pos, int n_chrom,
                            lra_cluster_result* out);
/* The anchor bonus of the first sparse DP per read (Map_lowacc.h:86-89, :184-185): 3 for a read that has a cluster with anchorfreq in
 * (1, 2] and at least 500 matches ("repetitivecluster"), opts.initial_anchorbonus otherwise.  *d_rate: [n_reads] floats, context-owned;
 * pass it as d_rate to lra_sparse_dp_batch.  Asynchronous on the context's stream.                                                    */
int lra_match_rate_batch(lra_ctx* ctx, const lra_cluster_result* clusters, float initial_anchorbonus, const float** d_rate);

/* ---- a5 (high-accuracy path): MatchesToFineClusters -----------------------------------------------------------------------------------
 * Replaces   MatchesToFineClusters(forMatches, clusters, ...); MatchesToFineClusters(revMatches, clusters, ..., 1)   (Clustering.h:1555-1680,
 *            Map_highacc.h:41-42)
 * behind lra_clean_matches_batch run with the high-accuracy options (bypassClustering = 0: its clusters are the rough clusters of
 * CleanOffDiagonal): CartesianSort of every rough cluster, SplitRoughClustersWithGaps (:1358-1432), StoreFineClusters (:892-1331) for both
 * strands of every read, sharing the read's `clusters` vector as the reference does.  opts: Options::globalK, RoughClustermaxGap, maxDiag,
 * maxGap, minClusterSize, minUniqueStretchNum, minUniqueStretchDist (presets lra.cpp:268-340).
 * Output (context-owned): fine clusters CSR by read (d_cluster_off), their matches CSR by cluster (d_match_off over d_q / d_t, genome
 * coordinates), box {qStart, qEnd, tStart, tEnd}, strand, chromIndex, anchorfreq; d_status[r] = LRA_ST_OOB_SLOT where the reference reads
 * clusters.back() of an empty vector (:1305; the read then has no clusters).  Synchronous.                                               */
typedef struct lra_fine_opts { int32_t globalK, RoughClustermaxGap, maxDiag, maxGap, minClusterSize, minUniqueStretchNum, minUniqueStretchDist; } lra_fine_opts;
typedef struct lra_fine_result {
  int32_t n_reads;
  uint64_t n_clusters, n_matches;
  const uint64_t* d_cluster_off;   /* [n_reads+1] */
  const uint64_t* d_match_off;     /* [n_clusters+1] */
  const uint32_t* d_q; const uint32_t* d_t;   /* [n_matches] */
  const uint32_t* d_box;           /* [4*n_clusters] */
  const int32_t* d_strand; const int32_t* d_chrom; const float* d_anchorfreq;   /* [n_clusters] */
  const uint32_t* d_status;        /* [n_reads] */
} lra_fine_result;
int lra_fine_clusters_batch(lra_ctx* ctx, const lra_cluster_result* rough, const lra_fine_opts* opts, const uint64_t* h_chrom_pos, int n_chrom,
                            lra_fine_result* out);

/* ---- a7: linear extension of the cleaned clusters ------------------------------------------
 * Replaces, per cluster of the context's current lra_clean_matches_batch result,
 *   LinearExtend(&clusters[d].matches, ext.matches, ext.matchesLengths, opts, genome, read,
 *                chromIndex, strand, 1, opts.globalK)            (LinearExtend.h:658-716, Checkbp :50-85)
 *   DecideCoordinates(ext, strand, chromIndex, anchorfreq)        (LinearExtend.h:105-128)
 * as MapRead_lowacc does (Map_lowacc.h:131-137; chromosome offsets handled inside).  d_seq/d_read_off:
 * the reads of the batch (as given to lra_seed_batch).  Output: per cluster its extended anchors
 * (read pos, genome pos, length) at [d_e_start[c], d_e_start[c] + d_e_count[c]) and its new box.    */
typedef struct lra_extend_result {
  uint64_t n_clusters, n_anchors_cap;
  const uint64_t* d_e_start;      /* [n_clusters] */
  const uint32_t* d_e_count;      /* [n_clusters] */
  const uint32_t* d_e_qpos; const uint32_t* d_e_tpos; const int32_t* d_e_len;   /* [n_anchors_cap] */
  const uint32_t* d_box;          /* [4*n_clusters] qStart,qEnd,tStart,tEnd */
} lra_extend_result;
int lra_linear_extend_batch(lra_ctx* ctx, int K, const char* d_seq, const uint64_t* d_read_off, lra_extend_result* out);

/* ---- a7 (high-accuracy path): the cluster version of LinearExtend ---------------------------------------------------------------------
 * Replaces   LinearExtend_chain(chain, ExtendClusters, RefinedClusters, smallOpts, genome, read, start, overlap, skiprepetitive, K)
 *            (LinearExtend.h:783-792) = LinearExtend(vector<Cluster*>, vector<Cluster>&, chain, ...) (:136-352; CheckOverlap :88, Checkbp :50,
 *            DecideCoordinates :105) + TrimOverlappedAnchors(ExtendClusters, start) (:574-649), called at Map_highacc.h:580 for every chain.
 * Item i = one element of one chain: the refined cluster d_item_cluster[i], its neighbours on the chain d_item_prev[i] / d_item_next[i]
 * (cluster indices, -1 at the ends), the read d_item_read[i].  Refined clusters: matches CSR d_match_off over d_mq / d_mt (t relative to the
 * cluster's chromosome; SORTED IN PLACE by diagonal / anti-diagonal as the reference does, :201-210), d_box {qStart, qEnd, tStart, tEnd} (t
 * relative), d_strand, d_chrom, d_anchorfreq.  d_seq / d_read_off: the reads (forward); d_genome: all chromosomes.
 * Output (context-owned): per item its anchors d_anchor_off[i] .. d_anchor_off[i+1] (read pos, chromosome pos, length, Cluster::overlap flag),
 * and what DecideCoordinates leaves: box, strand (-1 for an element without matches: the Cluster() default), chromIndex, anchorfreq.
 * trim != 0 applies TrimOverlappedAnchors to every item's list.  Synchronous.                                                              */
typedef struct lra_ext_clusters_result {
  uint64_t n_items, n_anchors;
  const uint64_t* d_anchor_off;    /* [n_items+1] */
  const uint32_t* d_q; const uint32_t* d_t; const int32_t* d_len; const uint8_t* d_overlap;   /* [n_anchors] */
  const uint32_t* d_box; const int32_t* d_strand; const int32_t* d_chrom; const float* d_anchorfreq;   /* [4*n_items], [n_items] */
} lra_ext_clusters_result;
int lra_linear_extend_clusters_batch(lra_ctx* ctx, uint64_t n_items, const uint32_t* d_item_cluster, const int32_t* d_item_prev, const int32_t* d_item_next,
                                     const uint32_t* d_item_read, uint64_t n_clusters, const uint64_t* d_match_off, uint64_t n_matches, uint32_t* d_mq, uint32_t* d_mt,
                                     const uint32_t* d_box, const int32_t* d_strand, const int32_t* d_chrom, const float* d_anchorfreq, const char* d_seq,
                                     const uint64_t* d_read_off, const char* d_genome, const uint64_t* h_chrom_pos, int n_chrom, int skiprepetitive, int K, int trim,
                                     lra_ext_clusters_result* out);

/* ---- a8: sparse dynamic programming over the extended anchors ("SDP#A") -------------------------
 * Replaces, per read,   SparseDP(ext_clusters, chains, optsSDP, LookUpTable, read, match_rate)
 * (SparseDP.h:2139-2279 as called from Map_lowacc.h:188): insertPointsPair (:79), the SortByRowOp / SortByColOp
 * std::sorts (:2171-2174, libstdc++ permutation of tied points included), GetRowInfo / GetColInfo, the four
 * decompositions DivideSubProbBy{Row,Col}{1,2}, ProcessPoint<Cluster> (:1015) with Maximization /
 * FindValueInBlock / PWL_w (SubRountine.h), PassValueToD1/D2 (:140,:226), the Fragment_valueOrder sort,
 * TraceBack (:1351) and DecidePrimaryChains (:1658).  InitPWL(gapopen, gapextend, gaproot, gapCeiling1,
 * gapCeiling2) (lra.cpp:648) is evaluated on the host with the host libm, as the reference does at start-up.
 * Input: clusters CSR by read (d_cluster_off[n_reads+1]); per cluster its anchors
 * [d_c_start[c], d_c_start[c] + d_c_count[c]) in d_q/d_t/d_len (read pos, GLOBAL genome pos, length -- the layout
 * lra_linear_extend_batch produces) and its strand; read lengths from d_read_off; match_rate per read
 * (d_rate, NULL = opts->rate for every read; Map_lowacc.h:185-186 uses 3 for reads with a repetitive cluster).
 * Output (context-owned, valid until the next call): per read up to opts->NumAln chains in slots
 * [r*NumAln, r*NumAln + d_n_chains[r]); slot s: anchors d_chain_cluster/anchor/link[d_chain_start[s] .. + d_chain_len[s])
 * in trace-back order (last anchor first; cluster = index within the read, anchor = index within the cluster,
 * link[i] = 1 if the step to the next listed anchor is an inversion link), box = QStart,QEnd,TStart,TEnd,
 * value = FirstSDPValue.  d_frag_off[n_reads+1] / d_frag_val: every anchor's final DP value in cluster order.
 * mode LRA_SDP_SINGLE_CLUSTER replaces  SparseDP(ClusterIndex, extend_clusters, ultimatechain, smallOpts, LookUpTable, read)
 * (SparseDP.h:2287-2438, called at Map_lowacc.h:535): every "read" is one cluster job, anchors get only their own family's point pair,
 * rate = opts.second_anchorbonus, and the one chain is the plain TraceBack (:1521) from the first anchor of maximal value (no box).
 * The same mode with several clusters per job is  SparseDP(SplitChain& inputChain, vector<Cluster_SameDiag*>&, FinalChain&, ...)
 * (SparseDP.h:1766-1955, LocalRefineAlignment.h:563): pass the clusters of inputChain in order, anchors = (GetqStart, GettStart, length).
 * SparseDP_ForwardOnly (SparseDP_Forward.h:312-450, called at LocalRefineAlignment.h:378) is this mode on one forward-strand cluster
 * per job with rate = (float)rate: the same points, weights, first-maximum rule and trace back.
 * d_status[r]: LRA_ST_CAPACITY if a work buffer bound was hit, LRA_ST_OOB_SLOT if the reference would read outside
 * its arrays (the read then has no chains).  Synchronous.                                              */
#define LRA_SDP_CLUSTERS 0
#define LRA_SDP_SINGLE_CLUSTER 1
typedef struct lra_sdp_opts {
  float rate; int32_t NumAln; float alnthres;
  float gapopen, gapextend, gaproot; int32_t gapCeiling1, gapCeiling2;
  int32_t mode;   /* LRA_SDP_CLUSTERS (0) or LRA_SDP_SINGLE_CLUSTER (1) */
  int32_t globalK; /* Options::globalK; read by lra_sparse_dp_boxes_batch only (value threshold, SparseDP.h:1592) */
} lra_sdp_opts;
typedef struct lra_chain_result {
  int32_t n_reads, num_aln;
  uint64_t n_frags, n_points, n_subproblem_entries;
  const uint32_t* d_n_chains;       /* [n_reads] */
  const uint64_t* d_chain_start;    /* [n_reads*num_aln] */
  const uint32_t* d_chain_len;      /* [n_reads*num_aln] */
  const uint32_t* d_chain_box;      /* [4*n_reads*num_aln] */
  const float* d_chain_value;       /* [n_reads*num_aln] */
  const uint32_t* d_chain_cluster; const uint32_t* d_chain_anchor; const uint8_t* d_chain_link;   /* [n_frags] */
  const uint32_t* d_chain_q; const uint32_t* d_chain_t; const int32_t* d_chain_alen; const uint8_t* d_chain_strand;   /* [n_frags] the anchors themselves (q, t, length, strand of the cluster) */
  const uint64_t* d_frag_off;       /* [n_reads+1] */
  const float* d_frag_val;          /* [n_frags] */
  const uint32_t* d_status;         /* [n_reads] */
  const int32_t* d_chain_num_anchors; /* [n_reads*num_aln] CHain::NumOfAnchors0 (lra_sparse_dp_boxes_batch; NULL otherwise) */
} lra_chain_result;
int lra_sparse_dp_batch(lra_ctx* ctx, int n_reads, const uint64_t* d_cluster_off, const uint64_t* d_c_start, const uint32_t* d_c_count,
                        const int32_t* d_c_strand, const uint32_t* d_q, const uint32_t* d_t, const int32_t* d_len,
                        const uint64_t* d_read_off, const float* d_rate, const lra_sdp_opts* opts, lra_chain_result* out);

/* The high-accuracy overload SparseDP(vector<Cluster>& splitclusters, vector<Primary_chain>&, opts, LookUpTable, read,
 * rate) (SparseDP.h:1956-2135, called at Map_highacc.h:229): every box contributes the four points s1 (qStart+1, tStart+1), e1 (qEnd-1,
 * tEnd-1), s2 (qStart+1, tEnd-1), e2 (qEnd-1, tStart+1) in that insertion order, weighs Val*rate (ProcessPoint :313), and chains are
 * chosen by DecidePrimaryChains :1587-1655: value threshold max(alnthres*best, best - 130*globalK), TraceBack with `used`, box by
 * min/max over the chain, read-span fraction > 0.005, at most NumAln chains (they are Primary_chains[0].chains).
 * Box i of read r is d_box_off[r] + i: (d_qs, d_qe, d_ts, d_te, d_strand 0 = forward, d_val = Cluster::Val, d_num_anchors =
 * Cluster::NumofAnchors0, may be NULL).  opts->rate (or d_rate[r]) is the caller's `rate` (Map_highacc.h:227-228); opts->mode is
 * ignored.  Result as lra_sparse_dp_batch: d_chain_cluster = box index within the read, d_chain_q/t = (qStart, tStart),
 * d_chain_alen = Val, d_chain_anchor = 0, plus d_chain_num_anchors (ComputeNumOfAnchors :1577).                         */
int lra_sparse_dp_boxes_batch(lra_ctx* ctx, int n_reads, const uint64_t* d_box_off, const uint32_t* d_qs, const uint32_t* d_qe,
                              const uint32_t* d_ts, const uint32_t* d_te, const int32_t* d_strand, const int32_t* d_val,
                              const int32_t* d_num_anchors, const uint64_t* d_read_off, const float* d_rate, const lra_sdp_opts* opts,
                              lra_chain_result* out);

/* ---- a6 (high-accuracy path): SplitClusters + DecideSplitClustersValue -----------------------------------
 * Replaces, for every read of a batch, SplitClusters(clusters, splitclusters, read, opts) (SplitClusters.h:63-171, IntervalSet :18-60)
 * and DecideSplitClustersValue(clusters, splitclusters, opts, read) (:176-249) as called at Map_highacc.h:153-155.
 * Cluster c of read r is d_cluster_off[r] + c: box (d_qs, d_qe, d_ts, d_te), d_strand (0 = forward), d_anchorfreq; its matches' read
 * positions (matches[i].first.pos, in the cluster's own CartesianSort order) are d_match_q[d_match_off[c] .. d_match_off[c+1]).
 * contig = (opts.readType == Options::contig) (only then are sparse clusters kept whole), K = opts.globalK.
 * Output (context-owned): split clusters of read r = [d_split_off[r], d_split_off[r+1]) of d_qs.. in the reference's push order (whole
 * clusters first, then pieces cluster by cluster): box, strand, d_coarse (index of the original within the read), d_val (Cluster::Val),
 * d_num_anchors (NumofAnchors0), d_read; per original cluster d_cluster_val (Cluster::Val) and d_cluster_split (Cluster::split).
 * The arrays feed lra_sparse_dp_boxes_batch unchanged.  Synchronous.                                                              */
typedef struct lra_split_clusters_result {
  int32_t n_reads; uint64_t n_clusters, n_split;
  const uint64_t* d_split_off;                 /* [n_reads+1] */
  const uint32_t* d_qs; const uint32_t* d_qe; const uint32_t* d_ts; const uint32_t* d_te;   /* [n_split] */
  const int32_t* d_strand; const int32_t* d_coarse; const int32_t* d_val; const int32_t* d_num_anchors; const uint32_t* d_read;   /* [n_split] */
  const int32_t* d_cluster_val; const uint8_t* d_cluster_split;   /* [n_clusters] */
} lra_split_clusters_result;
int lra_split_clusters_batch(lra_ctx* ctx, int n_reads, const uint64_t* d_cluster_off, const uint32_t* d_qs, const uint32_t* d_qe,
                             const uint32_t* d_ts, const uint32_t* d_te, const int32_t* d_strand, const float* d_anchorfreq,
                             const uint64_t* d_match_off, const uint32_t* d_match_q, int contig, int K, lra_split_clusters_result* out);

/* ---- a9 (low-accuracy path): chain filters and chain splitting -------------------------------------------
 * Replaces, per chain of an lra_sparse_dp_batch result (mode LRA_SDP_CLUSTERS), what MapRead_lowacc does before tier-2 refinement:
 *   RemoveSpuriousJump<UltimateChain>(chains[p])                                   Chain.h:897-957   (Map_lowacc.h:189-192)
 *   SPLITChain(genome, read, chains[p], spchain, spchain_link, opts)               Mapping_ultility.h:385-441 (push_new :349,
 *     SplitChain::CHROMIndex Chain.h:386, MergeSplitchainINS Mapping_ultility.h:172)
 *   RemoveSpuriousSplitChain(spchain, spchain_link)                                Map_lowacc.h:38-66
 * h_chrom_pos = genome.header.pos (n_chrom + 1 entries).  Output (context-owned), all arrays indexed like the chain arrays: chain
 * slot s occupies [d_chain_start[s], + d_chain_len[s]):
 *   d_keep[i]            1 if anchor i of the chain survives RemoveSpuriousJump; d_n_kept[s]; d_link[.. + n_kept-1) the filtered links
 *   d_n_split[s]         number of split chains; split k of slot s lives at index x = d_chain_start[s] + k of the per-split arrays:
 *   d_sp_beg[x], d_sp_len[x]   its anchors = d_sp_idx[d_chain_start[s] + beg .. + len) (indices into the FILTERED chain, in the
 *                        order SPLITChain leaves them: forward-strand pieces reversed), d_sp_link alongside (len-1 used)
 *   d_sp_type[x] ('N','T','I'), d_sp_strand[x], d_sp_chrom[x], d_sp_box[4x..] (QStart,QEnd,TStart,TEnd)
 *   d_ci_beg[x], d_ci_len[x]   SplitChain::ClusterIndex = d_ci_idx[d_chain_start[s] + beg .. + len)
 *   d_split_link[d_chain_start[s] + k], k < d_n_split_link[s]     spchain_link
 *   d_status[s]          LRA_ST_OOB_SLOT if the reference would read outside its arrays (the slot then has no splits).  Synchronous. */
typedef struct lra_split_result {
  uint64_t n_slots, n_frags;
  const uint8_t* d_keep; const uint32_t* d_n_kept; const uint8_t* d_link;
  const uint32_t* d_n_split; const uint32_t* d_sp_beg; const uint32_t* d_sp_len; const uint32_t* d_sp_idx; const uint8_t* d_sp_link;
  const uint8_t* d_sp_type; const uint8_t* d_sp_strand; const int32_t* d_sp_chrom; const uint32_t* d_sp_box;
  const uint32_t* d_ci_beg; const uint32_t* d_ci_len; const uint32_t* d_ci_idx;
  const uint8_t* d_split_link; const uint32_t* d_n_split_link; const uint32_t* d_status;
  const uint32_t* d_fidx;   /* [n_frags] position in the filtered chain -> position in the chain (both relative to d_chain_start[s]) */
} lra_split_result;
int lra_split_chains_batch(lra_ctx* ctx, const lra_chain_result* chains, const uint64_t* h_chrom_pos, int n_chrom, int splitdist,
                           int bypass_clustering, lra_split_result* out);

/* The chain filters of Chain.h on arbitrary chains (CSR d_off[n_chains+1] over anchors given by read pos, genome pos, length, strand of the
 * cluster, trace-back order; d_link[i] = link bit between anchor i and i+1 of the same chain, NULL if the chain type has none), applied in
 * the order h_ops[0..n_ops):  1 RemoveSmallPairedIndels (:546)   2 RemovePairedIndels (:607)   3 the same with refineEnds = false
 *   5 RemovePairedIndels(GenomePairs&, chain, lengths) (:753-811; strand and link are not read)
 * 4 RemoveSpuriousAnchors (:828; leaves `link` longer than the chain, as the reference does)   8 RemoveSpuriousJump (:897).
 * Map_lowacc.h:538-539 = {2, 4};  LocalRefineAlignment.h:567-571 = {1, 2 (or 3), 4}.
 * Output (context-owned; shares its buffer with lra_split_chains_batch): d_keep per anchor, d_n_kept per chain, the surviving links
 * d_link[d_off[c] .. + d_n_link[c]).  Synchronous.                                                                               */
typedef struct lra_filter_result {
  uint64_t n_chains, n_anchors;
  const uint8_t* d_keep; const uint32_t* d_n_kept; const uint8_t* d_link; const uint32_t* d_n_link;
} lra_filter_result;
int lra_filter_chains_batch(lra_ctx* ctx, uint64_t n_chains, const uint64_t* d_off, uint64_t n_anchors, const uint32_t* d_q, const uint32_t* d_t,
                            const int32_t* d_len, const uint8_t* d_strand, const uint8_t* d_link, const int32_t* h_ops, int n_ops,
                            lra_filter_result* out);
/* The same with the chain's qEnd(i) given explicitly (d_qend[i]; NULL = q + length): FinalChain::qEnd is Cluster_SameDiag::GetqEnd
 * (Clustering.h:378-380), the merged entry's length added to its LAST anchor's read position; the high-accuracy LocalRefineAlignment
 * (LocalRefineAlignment.h:567-571) filters such chains.                                                                              */
int lra_filter_chains_ex_batch(lra_ctx* ctx, uint64_t n_chains, const uint64_t* d_off, uint64_t n_anchors, const uint32_t* d_q, const uint32_t* d_t,
                               const int32_t* d_len, const uint32_t* d_qend, const uint8_t* d_strand, const uint8_t* d_link, const int32_t* h_ops, int n_ops,
                               lra_filter_result* out);

/* ---- a10: tier-2 (local) minimizer index and lookups ----------------------------------------
 * lra_local_index_batch replaces  LocalIndex::IndexSeq(char* seq, int seqLen)  (MMIndex.h:200-245) for
 * n_seqs sequences (read strands, Map_lowacc.h:246-250; or chromosomes = LocalIndex::IndexFile, the
 * `.gli` payload): per `window` bases the non-canonical (w,k) minimizers (MinCount.h:182), sorted by
 * k-mer with std::sort (MMIndex.h:219) and thinned by RemoveFrequent(maxFreq) (MMIndex.h:69).
 * LocalTuple = uint32  t | pos << 20  (TupleOps.h:20-25; pos relative to the window start).
 * Result (one context-owned allocation of `bytes` bytes starting at d_base, valid until the next call;
 * copy it with lra_copy_device to keep several indexes alive): d_win_off[n_seqs+1] (first window of
 * each sequence), d_tuple_bnd[n_windows+1] (tupleBoundaries), d_tuples[n_tuples].  Synchronous.
 *
 * lra_local_compare_batch replaces  CompareLists<LocalTuple,SmallTuple>(qBegin,qEnd,tBegin,tEnd,result,
 * opts,false,maxDiagNum,minDiagNum)  (CompareLists.h:9) for n_tasks (query list, target list) pairs
 * given as index ranges into two tuple arrays; max_freq = opts.localMaxFreq; per-task diagonal bounds
 * (NULL = unbounded; the filter applies only when both are non-zero, CompareLists.h:87).  Output: the
 * emitted pairs as absolute indices into the two tuple arrays, CSR by task, in the reference's order
 * (context-owned, valid until the next call).  Synchronous.                                        */
typedef struct lra_local_index_result {
  int32_t n_seqs;
  uint64_t n_windows, n_tuples, bytes;
  const void* d_base;
  const uint64_t* d_win_off; const uint64_t* d_tuple_bnd; const uint32_t* d_tuples;
} lra_local_index_result;
int lra_local_index_batch(lra_ctx* ctx, int n_seqs, const char* d_seq, const uint64_t* d_seq_off, int k, int w, int window,
                          int max_freq, lra_local_index_result* out);
/* The same with sequences switched off: d_active[s] == 0 leaves sequence s's windows in place (the window numbering is unchanged) but empty.
 * MapRead_lowacc indexes both strands of every read (Map_lowacc.h:249-250) and then only looks up the strands its split chains lie on.       */
int lra_local_index_masked_batch(lra_ctx* ctx, int n_seqs, const char* d_seq, const uint64_t* d_seq_off, const uint8_t* d_active, int k, int w,
                                 int window, int max_freq, lra_local_index_result* out);
typedef struct lra_local_pairs_result {
  uint64_t n_tasks, n_pairs;
  const uint64_t* d_pair_off; const uint32_t* d_pair_qi; const uint32_t* d_pair_ti;
} lra_local_pairs_result;
int lra_local_compare_batch(lra_ctx* ctx, uint64_t n_tasks, const uint32_t* d_q_tuples, const uint64_t* d_q_lo, const uint64_t* d_q_hi,
                            const uint32_t* d_t_tuples, const uint64_t* d_t_lo, const uint64_t* d_t_hi, int max_freq,
                            const int64_t* d_max_diag, const int64_t* d_min_diag, lra_local_pairs_result* out);

/* Refine_splitchain(splitchains, chain, refinedclusters, clusters, genome, read, glIndex, localIndexes, smallOpts, opts)
 * (ChainRefine.h:384-576, called at Map_lowacc.h:294) for every split chain of an lra_split_chains_batch result: the walk over the
 * genome local-index windows under the split chain (LocalIndex::LookupIndex MMIndex.h:175, GenomeHeader::GetNextOffset Genome.h:43),
 * with the extended clusters of the split chain brought to chromosome coordinates on their own strand (SwapStrand ClusterRefine.h:24,
 * K = opts.globalK), CompareLists<LocalTuple,SmallTuple> of every (read window, genome window) it meets, AppendValues
 * (TupleOps.h:159-195), SwapStrand of reverse results (K = smallOpts.globalK), SetClusterBoundariesFromMatches (Clustering.h:308) and
 * refineEffiency.
 * read_index: ONE lra_local_index_batch result over 2*n_reads sequences, the reads forward then their reverse complements
 * (forwardIndex / reverseIndex, Map_lowacc.h:246-250), built with window = opts->local_window.  Genome local index (`.gli` payload):
 * d_g_seq_off[n_g_windows+1] = glIndex.seqOffsets, d_g_tuple_bnd = tupleBoundaries, d_g_tuples = minimizers.
 * opts: window = smallOpts.window, smallK = smallOpts.globalK, K = opts.globalK, limitrefine = opts.limitrefine, max_freq =
 * smallOpts.localMaxFreq, local_window = glIndex.localIndexWindow.
 * UNDEFINED IN THE REFERENCE: with limitrefine (the default) the upper diagonal bound of every window starts from an uninitialised
 * variable (ChainRefine.h:468 `miniMaxDiag = miniMaxDiag;`); here it starts from the first anchor's diagonal like the lower bound.
 * Output (context-owned), indexed like the split arrays (split k of slot s at x = d_chain_start[s] + k): refinedclusters[k].matches =
 * (d_match_q, d_match_t)[d_match_off[x] .. d_match_off[x+1]) (t relative to the chromosome), d_box[4x..] = qStart,qEnd,tStart,tEnd,
 * d_eff[x] = refineEffiency, d_status[x] (LRA_ST_OOB_SLOT where the reference would index outside an array; no matches then).
 * strand / coarse / chromIndex of the refined cluster are the split chain's Strand / k / chromIndex.  Synchronous.            */
typedef struct lra_rsc_opts { int32_t window, smallK, K, limitrefine, max_freq, local_window; } lra_rsc_opts;
typedef struct lra_refined_result {
  uint64_t n_frags, n_tasks, n_pairs, n_matches;
  const uint64_t* d_match_off;               /* [n_frags+1] */
  const uint32_t* d_match_q; const uint32_t* d_match_t;   /* [n_matches] */
  const uint32_t* d_box; const float* d_eff; const uint32_t* d_status;   /* [4*n_frags], [n_frags], [n_frags] */
  const uint64_t* d_task_q_lo; const uint64_t* d_task_q_hi; const uint64_t* d_task_t_lo; const uint64_t* d_task_t_hi;   /* [n_tasks] the CompareLists calls, as tuple ranges */
} lra_refined_result;
int lra_refine_splitchain_batch(lra_ctx* ctx, const lra_chain_result* chains, const lra_split_result* split, const uint64_t* d_read_off,
                                const uint64_t* h_chrom_pos, int n_chrom, const lra_local_index_result* read_index, uint64_t n_g_windows,
                                const uint64_t* d_g_seq_off, const uint64_t* d_g_tuple_bnd, const uint32_t* d_g_tuples,
                                const lra_rsc_opts* opts, lra_refined_result* out);

/* REFINEclusters(clusters, refinedclusters, genome, read, glIndex, localIndexes, smallOpts, opts) (ClusterRefine.h:50-240, called at
 * Map_highacc.h:429-447): the cluster-wise twin of Refine_splitchain on the high-accuracy path.  Cluster c of read r is d_cluster_off[r] + c:
 * its matches are (d_q, d_t genome-wide)[d_c_start[c] .. + d_c_count[c]) (n_matches_cap = the extent of those arrays), its box d_qs/d_qe/
 * d_ts/d_te (genome-wide t), d_c_strand.  read_index / genome index / opts as for lra_refine_splitchain_batch (limitrefine unused).
 * Output (context-owned, shares buffers with lra_refine_splitchain_batch): per cluster d_chrom (Cluster::CHROMIndex), d_status
 * (LRA_ST_REJECTED: the cluster spans two chromosomes and is cleared, :61-65; LRA_ST_OOB_SLOT), refined matches CSR (t relative to the
 * chromosome), box, refineEffiency.  strand / coarse (-1) / refinespace (0) are the caller's.  Synchronous.                          */
typedef struct lra_refined_clusters_result {
  uint64_t n_clusters, n_tasks, n_pairs, n_matches;
  const uint64_t* d_match_off; const uint32_t* d_match_q; const uint32_t* d_match_t;
  const uint32_t* d_box; const float* d_eff; const uint32_t* d_status; const int32_t* d_chrom;
} lra_refined_clusters_result;
int lra_refine_clusters_batch(lra_ctx* ctx, int n_reads, const uint64_t* d_cluster_off, const uint64_t* d_c_start, const uint32_t* d_c_count,
                              const int32_t* d_c_strand, const uint32_t* d_qs, const uint32_t* d_qe, const uint32_t* d_ts, const uint32_t* d_te,
                              const uint32_t* d_q, const uint32_t* d_t, uint64_t n_matches_cap, const uint64_t* d_read_off, const uint64_t* h_chrom_pos,
                              int n_chrom, const lra_local_index_result* read_index, uint64_t n_g_windows, const uint64_t* d_g_seq_off,
                              const uint64_t* d_g_tuple_bnd, const uint32_t* d_g_tuples, const lra_rsc_opts* opts, lra_refined_clusters_result* out);

/* ---- a11: anchors inside one gap ------------------------------------------------------------------------
 * Replaces   float RefineSpace(int K, int W, int refineSpaceDiag, bool consider_str, GenomePairs& EndPairs, const Options& opts,
 *                              Genome&, Read&, char* strands[2], int& ChromIndex, GenomePos qe, GenomePos qs, GenomePos te, GenomePos ts,
 *                              bool st, GenomePos lrts = 0, GenomePos lrlength = 0)                     (ClusterRefine.h:242-325)
 * for n gaps.  Gap p: query = d_qseq[q_off .. + q_len) (= strands[st] + qs, q_len = qe - qs), target = d_tseq[t_off .. + t_len)
 * (= genome.seqs[ChromIndex] + (ts - lrts), t_len = te - ts + lrlength), t_span = te - (ts - lrts) (GenomePos arithmetic, used for the
 * diagonal band), K / W / diag = refineSpaceDiag per gap, q_add = qs, t_add = ts - lrts, flip_len = read.length if (consider_str and
 * st == 1) else 0.  match / mismatch / indel = opts.localMatch / localMismatch / localIndel, max_freq = opts.localMaxFreq of the
 * Options the caller passes (CompareLists with Global = false).  Output (context-owned): EndPairs as (first.pos, second.pos) CSR by gap,
 * in the reference's order; identity (the return value: matching bases / min(span) when both spans are < 1000, else -1); status (the
 * AffineOneGapAlign status bits, LRA_ST_RANGE for W > 32).  Synchronous.                                                      */
typedef struct lra_refine_space_result {
  uint64_t n_problems, n_pairs, n_small;
  const uint64_t* d_pair_off; const uint32_t* d_pair_q; const uint32_t* d_pair_t; const float* d_identity; const uint32_t* d_status;
} lra_refine_space_result;
int lra_refine_space_batch(lra_ctx* ctx, int n, const char* d_qseq, const uint64_t* d_q_off, const int32_t* d_q_len, const char* d_tseq,
                           const uint64_t* d_t_off, const int32_t* d_t_len, const uint32_t* d_t_span, const int32_t* d_K, const int32_t* d_W,
                           const int32_t* d_diag, const uint32_t* d_q_add, const uint32_t* d_t_add, const uint32_t* d_flip_len, int match,
                           int mismatch, int indel, int max_freq, lra_refine_space_result* out);
/* the same with opts.localMaxFreq given per gap (d_max_freq[n]) */
int lra_refine_space_batch_mf(lra_ctx* ctx, int n, const char* d_qseq, const uint64_t* d_q_off, const int32_t* d_q_len, const char* d_tseq,
                              const uint64_t* d_t_off, const int32_t* d_t_len, const uint32_t* d_t_span, const int32_t* d_K, const int32_t* d_W,
                              const int32_t* d_diag, const uint32_t* d_q_add, const uint32_t* d_t_add, const uint32_t* d_flip_len, int match,
                              int mismatch, int indel, const int32_t* d_max_freq, lra_refine_space_result* out);

/* Refine_Btwnsplitchain(spchain, refined_clusters, RevBtwnCluster, tracerev, genome, read, smallOpts, strands, spchain_link)
 * (ChainRefine.h:579-754, called at Map_lowacc.h:362) for every chain of the batch: for each pair of neighbouring refined clusters the
 * case analysis of :587-660 (plain gap, INV, DUP) and RefineBtwnSpace_AppendCloseCluster (:59-121; append_to_closetcluster :22-56) /
 * RefineBtwnSpace (ClusterRefine.h:327-430) on the one or two spaces, then the spaces beyond the first and the last split chain
 * (:684-753), each RefineSpace call going through lra_refine_space_batch.  Gaps of one chain are taken in order (a gap reads the boxes
 * the previous one may have moved): the batch advances in rounds, one gap index per round.
 * refined = the lra_refine_splitchain_batch result of the same chains / split; d_strands = the reads forward, then (at byte rc_base)
 * their reverse complements, laid out by d_read_off; d_genome = all chromosomes back to back (h_chrom_pos).  opts = the Options the
 * reference passes (smallOpts): K / W = globalK / globalW, refineSpaceDist, anchorstoosparse, localMatch / localMismatch / localIndel,
 * max_freq = localMaxFreq.  Read types -ONT / -CLR only (the others leave refineSpaceDiag uninitialised, ChainRefine.h:68-71).
 * Output (context-owned), indexed like the split arrays: the refined clusters after the call -- matches (base matches first, then every
 * appended run in the reference's order), box, refineEffiency, refinespace flag.  RevBtwnCluster / tracerev are dead in the reference
 * (their consumer is commented out, Map_lowacc.h:363-370) and are not produced.  Synchronous.                                    */
typedef struct lra_btwn_opts { int32_t K, W, refineSpaceDist; float anchorstoosparse; int32_t match, mismatch, indel, max_freq; } lra_btwn_opts;
typedef struct lra_btwn_result {
  uint64_t n_frags, n_matches, n_problems, n_pairs; uint32_t n_rounds;
  const uint64_t* d_match_off; const uint32_t* d_match_q; const uint32_t* d_match_t;
  const uint32_t* d_box; const float* d_eff; const uint8_t* d_refinespace;
} lra_btwn_result;
int lra_refine_btwn_splitchain_batch(lra_ctx* ctx, const lra_chain_result* chains, const lra_split_result* split, const lra_refined_result* refined,
                                     const uint64_t* d_read_off, const char* d_strands, uint64_t rc_base, const char* d_genome,
                                     const uint64_t* h_chrom_pos, int n_chrom, const lra_btwn_opts* opts, lra_btwn_result* out);

/* What MapRead_lowacc does with the refined clusters of every chain before the second sparse DP (Map_lowacc.h:440-476):
 *   MergeChain(Refined_Clusters, mergeinfo, merge_spcluster, spcluster)                         ChainRefine.h:767-802
 *   LinearExtend(&Refined_Clusters[cI]->matches, extend_clusters[r].matches, .matchesLengths, smallOpts, genome, read, chromIndex,
 *                st, 0, smallOpts.globalK) for every member of a merged cluster                     LinearExtend.h:658-716
 *   DecideCoordinates(extend_clusters[r], st, chromIndex, anchorfreq)                             LinearExtend.h:105-127
 *   TrimOverlappedAnchors(extend_clusters, 0)                                                     LinearExtend.h:574-649
 * refined = the lra_refine_btwn_splitchain_batch result; d_seq = the reads (forward), d_genome = all chromosomes; K = smallOpts.globalK.
 * Output (context-owned): merged cluster g of chain slot s is d_slot_group_off[s] + g; it covers the refined clusters (dense index
 * d_cluster_base[s] + k for split chain k) d_group_first[g] .. d_group_last[g]; its anchors are (d_q, d_t chromosome-relative, d_len)
 * [d_anchor_off[g], + d_count[g]); d_box[4g..] / d_strand / d_chrom as DecideCoordinates leaves them.
 * The second sparse DP (Map_lowacc.h:535) is lra_sparse_dp_batch in mode LRA_SDP_SINGLE_CLUSTER with n_reads = n_groups,
 * d_cluster_off = d_iota, d_c_start = d_anchor_off, d_c_count = d_count, d_c_strand = d_strand, d_q / d_t / d_len.  Synchronous.   */
typedef struct lra_merge_result {
  uint64_t n_slots, n_groups, n_anchors;
  const uint64_t* d_slot_group_off;   /* [n_slots+1] */
  const uint64_t* d_cluster_base;     /* [n_slots+1] */
  const uint32_t* d_group_slot; const uint32_t* d_group_first; const uint32_t* d_group_last;   /* [n_groups] */
  const uint64_t* d_anchor_off;       /* [n_groups+1] */
  const uint32_t* d_count;            /* [n_groups] */
  const uint32_t* d_q; const uint32_t* d_t; const int32_t* d_len;   /* [n_anchors] */
  const uint32_t* d_box; const int32_t* d_strand; const int32_t* d_chrom;   /* [4*n_groups], [n_groups] */
  const uint64_t* d_iota;             /* [n_groups+1] 0, 1, 2, ... */
} lra_merge_result;
/* TrimOverlappedAnchors(GenomePairs&, vector<int>&) (LinearExtend.h:722-780; used at LocalRefineAlignment.h:371): n_lists anchor lists in CSR
 * d_off, lengths trimmed in place.  Synchronous.                                                                                          */
int lra_trim_anchor_pairs_batch(lra_ctx* ctx, uint64_t n_lists, const uint64_t* d_off, uint64_t n_anchors, uint32_t* d_q, uint32_t* d_t, int32_t* d_len);
/* TrimOverlappedAnchors(vector<Cluster>& extCluster, int start) (LinearExtend.h:574-649; LinearExtend_chain :783, Map_lowacc.h:476): the cluster
 * version -- long anchors are those of 40 bases or more, a reverse-strand cluster (d_strand[c] == 1) is walked by read end and has the
 * read start of the trimmed anchor moved; read positions and lengths are changed in place.  Synchronous.                                    */
int lra_trim_overlapped_anchors_batch(lra_ctx* ctx, uint64_t n_clusters, const uint64_t* d_off, uint64_t n_anchors, const int32_t* d_strand,
                                      uint32_t* d_q, uint32_t* d_t, int32_t* d_len);
int lra_merge_extend_batch(lra_ctx* ctx, const lra_chain_result* chains, const lra_split_result* split, const lra_btwn_result* refined, const char* d_seq,
                           const uint64_t* d_read_off, const char* d_genome, const uint64_t* h_chrom_pos, int n_chrom, int K, lra_merge_result* out);

/* ---- a12: banded one-gap seed-extension DP ------------------------------------------
 * Replaces   int AffineOneGapAlign(string& qSeq, int qLen, string& tSeq, int tLen,
 *                                  int m, int mm, int indel, int k, Alignment& aln,
 *                                  AffineAlignBuffers& b)          (AffineOneGapAlign.h:157)
 * for n independent (q,t,k) problems.  Sequences are ASCII bytes inside device buffers
 * d_qseq / d_tseq (may be the same buffer); problem p uses d_qseq[q_off[p] .. +q_len[p]) and
 * d_tseq[t_off[p] .. +t_len[p]).
 * Outputs per problem: the returned score; the gapless blocks the reference appends to
 * aln.blocks, as (qPos,tPos,length) int32 triples written at d_blocks + 3*d_block_off[p]
 * (capacity d_block_off[p+1]-d_block_off[p] triples; min(q_len,t_len)+1 always suffices);
 * their count; a status word (bits above).                                              */
int lra_affine_one_gap_align_batch(lra_ctx* ctx, int n, const char* d_qseq, const char* d_tseq,
                                   const uint64_t* d_q_off, const int32_t* d_q_len,
                                   const uint64_t* d_t_off, const int32_t* d_t_len,
                                   const int32_t* d_k, int m, int mm, int indel,
                                   int32_t* d_score, int32_t* d_nblocks, int32_t* d_blocks,
                                   const uint64_t* d_block_off, int32_t* d_status);

/* ---- a13 (DP leaf): the alignment between two consecutive chain anchors -------------------------------------
 * Replaces   RefineByLinearAlignment(btc_curReadEnd, btc_curGenomeEnd, btc_nextReadStart, btc_nextGenomeStart, str, chromIndex,
 *                                    alignment, read, genome, strands, ...)                    (LocalRefineAlignment.h:141-185)
 * = SetMatchAndGaps (:93) + RefineSubstrings (:127) + AlignSubstrings (:100) for n anchor pairs: if min(nextReadStart - curReadEnd + 1,
 * nextGenomeStart - curGenomeEnd + 1) > 0 and (opts.refineLevel & REF_DP), AffineOneGapAlign on strands[str][curReadEnd, nextReadStart)
 * x genome.seqs[chromIndex][curGenomeEnd, nextGenomeStart) with band min(2 * |qLen - tLen| + 1, opts.localBand); its blocks, shifted by
 * (curReadEnd, curGenomeEnd), are what the reference appends to alignment->blocks.  d_q_base[p] / d_t_base[p]: offset of strands[str] /
 * genome.seqs[chromIndex] inside d_qseq / d_tseq.  Output (context-owned): blocks CSR by pair (qPos,tPos,length), score, status
 * (AffineOneGapAlign bits; LRA_ST_RANGE if a span is negative).  Synchronous.                                                  */
typedef struct lra_between_result {
  uint64_t n_gaps, n_blocks;
  const uint64_t* d_block_off; const int32_t* d_blocks; const int32_t* d_score; const uint32_t* d_status;
} lra_between_result;
int lra_between_anchors_batch(lra_ctx* ctx, int n, const char* d_qseq, const uint64_t* d_q_base, const uint32_t* d_cur_read_end,
                              const uint32_t* d_next_read_start, const char* d_tseq, const uint64_t* d_t_base, const uint32_t* d_cur_genome_end,
                              const uint32_t* d_next_genome_start, int match, int mismatch, int indel, int local_band, int refine_dp,
                              lra_between_result* out);

/* ---- a13: the chain walk that turns the second sparse DP's chains into alignments ------------------------------------------------
 * Replaces   LocalRefineAlignment(ultimatechains, ext_clusters, alignments, smallOpts, LookUpTable, read, strands, h, genome, LSC, tinyOpts,
 *                                 buff, svsigstrm)                                   (LocalRefineAlignment.h:885-1029, Map_lowacc.h:576)
 * including RefinedAlignmentbtwnAnchors (:203-550) for n_jobs primary chains.  Job j (primary chain h = d_job_h[j] of read d_job_read[j]) has
 * the chains d_job_chain_off[j] .. d_job_chain_off[j+1] (ultimatechains[st], after RemovePairedIndels / RemoveSpuriousAnchors); chain c has
 * the anchors d_chain_anchor_off[c] .. (q on the forward read, t chromosome-relative, length; chain order = trace-back order), its cluster's
 * strand and chromIndex, FirstSDPValue, NumOfAnchors0, NumOfAnchors1.  d_strands = the reads forward, then (at rc_base) reverse complemented.
 * opts = tinyOpts: localW / globalW / localMaxFreq on entry, localMatch / localMismatch / localIndel / localBand, RefineBySDP, readType
 * (is_ont: clr / ont vs contig / ccs), and the PWL parameters of the sparse DP.
 * Output (context-owned): the SegAlignments each job pushes, in order: alignments d_job_aln_off[j] .. d_job_aln_off[j+1], each with strand,
 * Supplymentary, ISsecondary, NumOfAnchors0 / 1, chromIndex, value and its blocks (qPos, tPos, length) d_block_off[a] .. d_block_off[a+1].
 * d_status[j] != 0: the reference would read outside an array on that job.  Synchronous.                                             */
typedef struct lra_lra_opts {
  int32_t localW, globalW, localMaxFreq, match, mismatch, indel, localBand, refineBySDP, isOnt;
  float gapopen, gapextend, gaproot; int32_t gapCeiling1, gapCeiling2;
} lra_lra_opts;
typedef struct lra_alignments_result {
  uint64_t n_jobs, n_alignments, n_blocks, n_big, n_inner_jobs;
  const uint64_t* d_job_aln_off;
  const int32_t* d_strand; const int32_t* d_supp; const int32_t* d_secondary; const int32_t* d_n0; const int32_t* d_n1; const int32_t* d_chrom; const float* d_value;
  const uint64_t* d_block_off; const int32_t* d_blocks; const uint32_t* d_status;
} lra_alignments_result;
int lra_local_refine_batch(lra_ctx* ctx, uint64_t n_jobs, const uint64_t* d_job_chain_off, const uint32_t* d_job_read, const int32_t* d_job_h, uint64_t n_chains,
                           const uint64_t* d_chain_anchor_off, const int32_t* d_chain_strand, const int32_t* d_chain_chrom, const float* d_chain_value,
                           const int32_t* d_chain_n0, const int32_t* d_chain_n1, uint64_t n_anchors, const uint32_t* d_q, const uint32_t* d_t, const int32_t* d_len,
                           const uint64_t* d_read_off, const char* d_strands, uint64_t rc_base, const char* d_genome, const uint64_t* h_chrom_pos, int n_chrom,
                           const lra_lra_opts* opts, lra_alignments_result* out);

/* The walk of the high-accuracy overload  LocalRefineAlignment(Primary_chains, splitchains, ExtendClusters, alignments, smallOpts, LookUpTable, read, strands,
 * p, h, genome, LSC, tinyOpts, buff, svsigstrm, extend_clusters, false)  (LocalRefineAlignment.h:577-766, Map_highacc.h:708) behind its sparse DP, filters and
 * SwitchToOriginalAnchors (:563-576): as lra_local_refine_batch, except that chain st of a job is splitchains[st] (pass empty chains for pieces whose
 * ultimatechain is empty), a chain of one anchor still makes an alignment (:579), Supplymentary = (st != d_job_lsc[job]), d_chain_value /
 * d_chain_n0 = Primary_chains[p].chains[h].value / NumOfAnchors0, d_chain_n1 = the chain's size.                                              */
int lra_local_refine_highacc_batch(lra_ctx* ctx, uint64_t n_jobs, const uint64_t* d_job_chain_off, const uint32_t* d_job_read, const int32_t* d_job_h,
                                   const uint32_t* d_job_lsc, uint64_t n_chains, const uint64_t* d_chain_anchor_off, const int32_t* d_chain_strand,
                                   const int32_t* d_chain_chrom, const float* d_chain_value, const int32_t* d_chain_n0, const int32_t* d_chain_n1, uint64_t n_anchors,
                                   const uint32_t* d_q, const uint32_t* d_t, const int32_t* d_len, const uint64_t* d_read_off, const char* d_strands, uint64_t rc_base,
                                   const char* d_genome, const uint64_t* h_chrom_pos, int n_chrom, const lra_lra_opts* opts, lra_alignments_result* out);

/* From the second sparse DP to lra_local_refine_batch (Map_lowacc.h:530-540, :575): `second` = lra_sparse_dp_batch in single-cluster mode over the
 * merged clusters of `merge`; its chains are filtered (RemovePairedIndels<UltimateChain>, RemoveSpuriousAnchors) and grouped per primary chain.
 * job = chain slot of the first sparse DP (read = slot / num_aln, h = slot % num_aln; d_slot_n0[slot] = chains[p].NumOfAnchors0 kept by the caller, the
 * first DP's result arrays being overwritten by the second; may be NULL), chains = its merged clusters.  Output (context-owned) = the array
 * arguments of lra_local_refine_batch.  Synchronous.                                                                                         */
typedef struct lra_local_refine_inputs {
  uint64_t n_jobs, n_chains, n_anchors;
  const uint64_t* d_job_chain_off; const uint32_t* d_job_read; const int32_t* d_job_h;
  const uint64_t* d_chain_anchor_off; const int32_t* d_chain_strand; const int32_t* d_chain_chrom; const float* d_chain_value; const int32_t* d_chain_n0;
  const int32_t* d_chain_n1; const uint32_t* d_q; const uint32_t* d_t; const int32_t* d_len;
} lra_local_refine_inputs;
int lra_local_refine_inputs_batch(lra_ctx* ctx, int num_aln, const uint32_t* d_slot_n0, const lra_merge_result* merge, const lra_chain_result* second,
                                  lra_local_refine_inputs* out);

/* ---- a14: banded 3-state affine indel refinement ----------------------------------------
 * Replaces   void IndelRefineAlignment(Read& read, Genome& genome, Alignment& alignment,
 *                                      const Options& opts, IndelRefineBuffers& buffers,
 *                                      bool endAlign = false)            (IndelRefine.h:53)
 * for n_aln alignments.  Alignment a: its gapless blocks alignment.blocks as (qPos,tPos,length)
 * int32 triples d_blocks_in[3*d_block_off[a] .. 3*d_block_off[a+1]) (absolute read / chromosome
 * coordinates); alignment.read = the read strand it is on = d_qseq + d_q_off[a] of length
 * d_q_len[a] (read.length); genome.seqs[alignment.chromIndex] = d_tseq + d_t_off[a] of length
 * d_t_len[a] (genome.lengths[chromIndex]).  refine_band = opts.refineBand (2..64), match /
 * mismatch / indel = opts.localMatch / localMismatch / localIndel, end_align as the argument.
 * Output: the refined alignment.blocks of every alignment (CSR, device arrays owned by the
 * context, valid until the next call), a status word per alignment (bits above; LRA_ST_RANGE
 * also flags a row window wider than 64 cells, not supported yet).  Synchronous.            */
typedef struct lra_refine_result {
  int32_t n_aln;
  uint64_t n_blocks, n_segments, n_rows, n_cells, n_aog;
  const uint64_t* d_block_off;   /* [n_aln+1] */
  const int32_t* d_blocks;       /* [3*n_blocks] */
  const int32_t* d_status;       /* [n_aln] */
} lra_refine_result;
int lra_indel_refine_batch(lra_ctx* ctx, int n_aln, const int32_t* d_blocks_in, const uint64_t* d_block_off,
                           uint64_t n_blocks_in, const char* d_qseq, const uint64_t* d_q_off,
                           const int32_t* d_q_len, const char* d_tseq, const uint64_t* d_t_off,
                           const int64_t* d_t_len, int refine_band, int match, int mismatch, int indel,
                           int end_align, lra_refine_result* out);

/* ---- a16: alignment statistics + CIGAR ---------------------------------------------------
 * Replaces   void Alignment::CalculateStatistics(const Options&, ostream*, const vector<float>& LookUpTable)
 * (Alignment.h:513-531 = CreateAlignmentStrings :247 + AlignStringsToCigar :414, opts.showmm) for
 * n_aln alignments given as blocks (same conventions as lra_indel_refine_batch; d_t_off is the
 * chromosome start).  h_lookup = the reference's LookUpTable (LogLookUpTable.h: logf(1), logf(6), ...,
 * 2001 floats, computed by the HOST libm so the float value matches the reference bit for bit).
 * Output per alignment: 18 int32 counters in the order
 *   nm nmm nins ndel tdel tins nSmallDel nMedDel nLargeDel nSmallIns nMedIns nLargeIns preClip sufClip qStart qEnd tStart tEnd
 * named after the reference's MEMBERS after the call (so nins counts deletion runs and ndel insertion
 * runs), the float `value` (NV), and the CIGAR as runs (length << 4 | op,
 * op 0 '=', 1 'X', 2 'I', 3 'D'), CSR by alignment.  Counters are this call's increments of a fresh
 * Alignment (the reference never resets tdel/tins/nSmall*).  Synchronous.                          */
typedef struct lra_stats_result {
  int32_t n_aln;
  uint64_t n_runs;
  const int32_t* d_counts;    /* [18*n_aln] */
  const float* d_value;       /* [n_aln] */
  const uint64_t* d_run_off;  /* [n_aln+1] */
  const uint32_t* d_runs;     /* [n_runs] */
} lra_stats_result;
int lra_calculate_statistics_batch(lra_ctx* ctx, int n_aln, const int32_t* d_blocks, const uint64_t* d_block_off,
                                   const char* d_qseq, const uint64_t* d_q_off, const int32_t* d_q_len,
                                   const char* d_tseq, const uint64_t* d_t_off, const float* h_lookup, int n_lookup,
                                   lra_stats_result* out);

/* ---- a15: junctions of split alignments ---------------------------------------------------------------------
 * Replaces   RefineBreakpoint(read, genome, leftAln, rightAln, opts)   (RefineBreakpoint.h:210-466; Map_lowacc.h:592, Map_highacc.h:725)
 * for n junctions: if the read bases between the two segments (in forward read coordinates) number 1..499, both segments are extended into
 * the gap by a full DP (RSdp: match 2, mismatch -2, gap -4), cut where the summed score is best, and the new blocks are glued on
 * (PrependBlocks / AppendBlocks).  Per junction and side: the segment's blocks (CSR, (qPos,tPos,length) triples), its strand, the offset
 * of Alignment::read (the read strand it is aligned on) in d_seq, the offset and length of its chromosome in d_genome; d_read_len.
 * Output (context-owned): the two block lists after the call, at d_*_blocks + 3 * d_*_off[j] with d_*_n[j] blocks each; status bit
 * 0x10000 = refined, LRA_ST_OOB_SLOT = the reference would read outside the read / chromosome (lists unchanged).  Synchronous.   */
typedef struct lra_breakpoint_result {
  uint64_t n_junctions;
  const int32_t* d_l_blocks; const uint64_t* d_l_off; const int32_t* d_l_n;
  const int32_t* d_r_blocks; const uint64_t* d_r_off; const int32_t* d_r_n;
  const uint32_t* d_status;
} lra_breakpoint_result;
int lra_refine_breakpoint_batch(lra_ctx* ctx, int n, const int32_t* d_read_len, const char* d_seq, const char* d_genome, const int32_t* d_l_blocks,
                                const uint64_t* d_l_off, const int32_t* d_l_strand, const uint64_t* d_l_read_off, const uint64_t* d_l_chrom_off,
                                const int32_t* d_l_chrom_len, const int32_t* d_r_blocks, const uint64_t* d_r_off, const int32_t* d_r_strand,
                                const uint64_t* d_r_read_off, const uint64_t* d_r_chrom_off, const int32_t* d_r_chrom_len, lra_breakpoint_result* out);

/* ---- a17: SAM / PAF / BED records (host code, no device work) ------------------------------------------
 * Byte-for-byte the text of  Alignment::PrintSAM (Alignment.h:658-808), SimplePrintSAM (:811-905), PrintPAF (:600-656) and
 * PrintBed (:591-598) for alignments described by plain records (the fields those functions read).  Tags in the reference's order:
 * SAM  NM MM NX ND TD NI TI NV AS AO N0 RT TP SD ME LD SI MI LI [SA];  simple SAM  RT NM NX ND TD NI TI N0 NV AS AO;
 * PAF  OR NM NX ND TD NI TI SD ME LD SI MI LI N0 NV AS TP [NA] [RT] [CG].  opts.printMD is not supported (the MD string needs the
 * alignment strings).  n_blocks == 0 prints the unaligned record.  lra_format_sam prints segment `as` of the group and lists the other
 * segments, last to first, in SA:Z.  passthrough: the text appended when opts.passthroughtag is set (NULL otherwise).
 * Output: at most cap bytes into out (no terminator); *len = bytes needed.  Returns LRA_ERR_INVALID if cap < *len.            */
typedef struct lra_aln_record {
  const char* read_name; const char* read; const char* qual;   /* qual: NULL, "*" or read_len quality characters (NUL-terminated) */
  int32_t read_len;
  const char* chrom; uint32_t genome_len;                      /* Alignment::chrom, genomeLen */
  const char* cigar;
  uint32_t flag; int32_t strand; uint32_t mapqv; int32_t supplementary, typeofaln;
  uint32_t q_start, q_end, t_start, t_end; int32_t pre_clip, suf_clip;
  int32_t nm, nmm, nins, ndel, tdel, tins, nSmallDel, nMedDel, nLargeDel, nSmallIns, nMedIns, nLargeIns;
  float value; int32_t order, NumOfAnchors0, NumOfAnchors1, runtime;
  int32_t n_blocks; uint32_t first_block_qpos, last_block_qend;   /* blocks[0].qPos and blocks[last].qPos + length (hard-clipped substrings) */
  int32_t is_secondary;                                        /* Alignment::ISsecondary (read by lra_group_alignments, not printed) */
  const char* md;                                              /* NULL, or the MD:Z value (opts.printMD; lra_md_string) */
  /* print format 'a' (PrintPairwise) only, NULL otherwise: the blocks, Alignment::read (the read on the alignment's strand) and a pointer p
   * with p[tPos] = chromosome base tPos for every tPos the blocks cover */
  const int32_t* blocks; const char* strand_read; const char* chrom_text;
} lra_aln_record;
int lra_format_sam(const lra_aln_record* group, int n_group, int as, int hard_clip, const char* passthrough, char* out, uint64_t cap, uint64_t* len);
int lra_format_sam_simple(const lra_aln_record* rec, int hard_clip, const char* passthrough, char* out, uint64_t cap, uint64_t* len);
int lra_format_paf(const lra_aln_record* rec, int print_cigar, char* out, uint64_t cap, uint64_t* len);
int lra_format_bed(const lra_aln_record* rec, char* out, uint64_t cap, uint64_t* len);
/* "@PG\tID:lra\tPN:lra\tVN:<version>\tCL:<command_line>" (lra.cpp:665-671) + one "@SQ\tSN:..\tLN:.." per chromosome (Genome.h:85-89) */
int lra_format_sam_header(const char* version, const char* command_line, const char* const* chrom_names, const uint64_t* chrom_pos, int n_chrom,
                          char* out, uint64_t cap, uint64_t* len);

/* Alignment strings and what is printed from them (host code): CreateAlignmentStrings (Alignment.h:247-331), AlignmentStringsToMD (:204-245,
 * the MD:Z value PrintSAM adds with opts.printMD), PrintPairwise (:564-589, print format "a").  Two-call convention as above.             */
int lra_alignment_strings(const char* query, const char* text, const int32_t* blocks, int n_blocks, char* q_out, char* a_out, char* t_out, uint64_t cap,
                          uint64_t* len, uint32_t* ref_len);
int lra_md_string(const char* query_str, const char* ref_str, uint64_t n, char* out, uint64_t cap, uint64_t* len);
int lra_format_pairwise(const char* read_name, const char* chrom, int n_blocks, int first_q, int first_t, uint32_t ref_len, const char* query_str,
                        const char* align_str, const char* ref_str, uint64_t n, char* out, uint64_t cap, uint64_t* len);

/* ---- a16 / a17: a read's alignments between CalculateStatistics and the text records (host code) -------------------------------
 * lra_group_alignments  = SegAlignmentGroup::SetFromSegAlignment (Alignment.h:944-983) for n_groups alignments whose segment records are
 *                         recs[seg_off[g] .. seg_off[g+1]) (SegAlignment order): sums, flags (REVERSE, SUPPLEMENTARY), ISsecondary.
 * lra_order_alignments  = AlignmentsOrder::Update (:1021-1046; operator() :1048, std::sort): index[old_end..n_groups) ordered by
 *                         (value, NumOfAnchors0) descending, first primary, others secondary (SECONDARY flag, typeofaln 2 unless 3).
 * lra_simple_mapqv      = SimpleMapQV (Mapping_ultility.h:497-595), opts.bypassClustering / readType == clr / == ont / globalK.
 * lra_output_read       = OUTPUT (:453-493) and output_unaligned (:445-451): formats 's' (PrintSAM), 'b' (PrintBed), 'p' / 'P' (PrintPAF
 *                         without / with CIGAR), 'a' (PrintPairwise; needs the records' blocks / strand_read / chrom_text); sets Alignment::order;
 *                         two-call convention of the lra_format_* functions.                                                             */
typedef struct lra_aln_group {
  int32_t first, count;
  uint32_t q_start, q_end, t_start, t_end; int32_t nm, nmm, ndel, nins; int32_t is_secondary; float value; int32_t NumOfAnchors0, NumOfAnchors1;
} lra_aln_group;
int lra_group_alignments(lra_aln_record* recs, const int32_t* seg_off, int n_groups, lra_aln_group* groups);
int lra_order_alignments(lra_aln_group* groups, int n_groups, lra_aln_record* recs, int32_t* index, int old_end);
int lra_simple_mapqv(const lra_aln_group* groups, const int32_t* index, int n_groups, lra_aln_record* recs, int bypass_clustering, int is_clr, int is_ont,
                     int globalK);
int lra_output_read(const lra_aln_group* groups, const int32_t* index, int n_groups, lra_aln_record* recs, int print_num_aln, char format, int hard_clip,
                    const char* passthrough, int read_unaligned, const lra_aln_record* unaligned_rec, char* out, uint64_t cap, uint64_t* len);

/* ---- a7 (high-accuracy path): MergeMatchesSameDiag -----------------------------------------------------------------------------------
 * Replaces   MergeMatchesSameDiag(extend_clusters, samediag_clusters, opts)                    (LinearExtend.h:795-829, Map_highacc.h:642)
 * for n_clusters extended clusters: cluster c has the anchors d_anchor_off[c] .. d_anchor_off[c+1] (read pos, chromosome pos, length,
 * Cluster::overlap flag, in the cluster's order) and its strand.  Output: Cluster_SameDiag::start / end of cluster c =
 * d_start / d_end [d_group_off[c] .. d_group_off[c+1]) (anchor indices relative to the cluster); d_status[c] = LRA_ST_OOB_SLOT for an
 * empty cluster (the reference reads matches[0]).  Synchronous.                                                                        */
typedef struct lra_same_diag_result {
  uint64_t n_clusters, n_groups;
  const uint64_t* d_group_off; const uint32_t* d_start; const uint32_t* d_end; const uint32_t* d_status;
} lra_same_diag_result;
int lra_merge_same_diag_batch(lra_ctx* ctx, uint64_t n_clusters, const uint64_t* d_anchor_off, const uint32_t* d_q, const uint32_t* d_t,
                              const int32_t* d_len, const uint8_t* d_overlap, const int32_t* d_strand, int merge_dist, lra_same_diag_result* out);

/* ---- a11 (high-accuracy path): RefineBtwnSpace --------------------------------------------------------------------------------------
 * Replaces   int RefineBtwnSpace(int K, int W, vector<Cluster>& RevBtwnCluster, bool twoblocks, Cluster* cluster, const Options& opts, Genome&, Read&,
 *                                char* strands[2], GenomePos qe, GenomePos qs, GenomePos te, GenomePos ts, bool st, GenomePos lrts = 0,
 *                                GenomePos lrlength = 0)                                   (ClusterRefine.h:331-432; caller RefineBtwnClusters_chain :433)
 * for n spaces, up to the vector insert + SetClusterBoundariesFromMatches it ends with (the caller owns the clusters): per space the read it
 * belongs to, cluster->chromIndex, qs / qe / ts / te (t relative to the chromosome) and st as the caller passes them, twoblocks, lrts / lrlength
 * (NULL = 0).  read_type = LRA_READ_* (opts.readType picks refineSpaceDiag), anchorstoosparse / match / mismatch / indel / max_freq =
 * opts.anchorstoosparse / localMatch / localMismatch / localIndel / localMaxFreq.  Output per space: d_decision 0 nothing (:371), 1 the
 * pairs go into the cluster (:363-369), 3 the same after the reverse strand was tried (:415-421: anchorfreq = 1 too), 2 the pairs form a
 * new RevBtwnCluster on the other strand (:422-431, the reference returns 1); the pairs (CSR), refine efficiencies eff / reff (-1 when the
 * reverse strand was not tried).  Synchronous.                                                                                           */
typedef struct lra_btwn_space_result {
  uint64_t n, n_pairs, n_reverse_tried;
  const uint64_t* d_pair_off; const uint32_t* d_pair_q; const uint32_t* d_pair_t; const int32_t* d_decision; const float* d_eff; const float* d_reff;
} lra_btwn_space_result;
int lra_refine_btwn_space_batch(lra_ctx* ctx, int n, const uint32_t* d_qs, const uint32_t* d_qe, const uint32_t* d_ts, const uint32_t* d_te, const int32_t* d_st,
                                const uint8_t* d_twoblocks, const uint32_t* d_read, const int32_t* d_chrom, const uint32_t* d_lrts, const uint32_t* d_lrlength,
                                const uint64_t* d_read_off, const char* d_strands, uint64_t rc_base, const char* d_genome, const uint64_t* h_chrom_pos, int n_chrom,
                                int K, int W, int read_type, float anchorstoosparse, int match, int mismatch, int indel, int max_freq,
                                lra_btwn_space_result* out);

/* ---- a11 (high-accuracy path): RefineBtwnClusters_chain -------------------------------------------------------------------------------
 * Replaces   for every chain (p, h) of a read, in order: RefineBtwnClusters_chain(K, W, Primary_chains, RefinedClusters, RevBtwnCluster, tracerev,
 *            genome, read, smallOpts, p, h, strands)                               (ClusterRefine.h:433-614, Map_highacc.h:513-518)
 * Reads: d_read_chain_off[n_reads+1] over the chains (a read's chains in the reference's order); chain x = cluster indices
 * d_ch[d_chain_off[x] .. d_chain_off[x+1]) in the chain's order (ch[0] nearest the read's end).  Clusters (the RefinedClusters of the batch): matches
 * CSR d_match_off over d_mq / d_mt (t relative to the chromosome), d_box {qStart, qEnd, tStart, tEnd} = SetClusterBoundariesFromMatches of those
 * matches for K (UPDATED IN PLACE), d_strand, d_chrom, d_anchorfreq (UPDATED IN PLACE: 1 where RefineBtwnSpace decided on the cluster's own
 * strand after trying both, :414-419).  K / W: what the caller passes on (Map_highacc.h:466-468); read_type, anchorstoosparse, match, mismatch,
 * indel, max_freq as lra_refine_btwn_space_batch.  d_strands: the reads forward, then (at rc_base) reverse complemented.
 * Output (context-owned): every cluster's matches with the refined spaces' pairs appended in the reference's order (CSR), Cluster::refinespace.
 * Synchronous.                                                                                                                              */
typedef struct lra_btwn_clusters_result {
  uint64_t n_clusters, n_matches, n_problems, n_pairs_added; uint32_t n_rounds;
  const uint64_t* d_match_off; const uint32_t* d_q; const uint32_t* d_t; const uint8_t* d_refinespace;
} lra_btwn_clusters_result;
int lra_refine_btwn_clusters_batch(lra_ctx* ctx, int n_reads, const uint64_t* d_read_chain_off, uint64_t n_chains, const uint64_t* d_chain_off, const uint32_t* d_ch,
                                   uint64_t n_clusters, const uint64_t* d_match_off, uint64_t n_matches, const uint32_t* d_mq, const uint32_t* d_mt, uint32_t* d_box,
                                   const int32_t* d_strand, const int32_t* d_chrom, float* d_anchorfreq, const uint64_t* d_read_off, const char* d_strands, uint64_t rc_base,
                                   const char* d_genome, const uint64_t* h_chrom_pos, int n_chrom, int K, int W, int read_type, float anchorstoosparse, int match, int mismatch,
                                   int indel, int max_freq, lra_btwn_clusters_result* out);

/* ---- a9 (high-accuracy path): SPLITChain over merged clusters -----------------------------------------------------------------------------
 * Replaces   SPLITChain(read, ExtendClusters, splitchains, Primary_chains[p].chains[h].link, smallOpts)   (Mapping_ultility.h:266-346, Map_highacc.h:706)
 * including MergeSplitchainINS (:172-262) and LargestSplitChain_dist (Chain.h:974-985, Map_highacc.h:707) for n_jobs chains.  Job j = the
 * Cluster_SameDiag elements d_job_off[j] .. d_job_off[j+1] (strand 0 = forward, chromIndex, box = qStart, qEnd, tStart, tEnd with t relative to
 * the chromosome) and the link bits d_link[d_link_off[j] ..] (entry im joins elements im and im + 1).  splitdist = opts.splitdist.
 * Output (context-owned): pieces of job j = d_job_piece_off[j] .. d_job_piece_off[j+1]; piece p = the elements d_sptc[d_piece_off[p] ..
 * d_piece_off[p+1]) (indices inside the job, in SplitChain::sptc order), its type ('T' 'D' 'I' 'N'), Strand, box (QStart, QEnd, TStart, TEnd);
 * d_job_lsc[j] = LSC.  The reference leaves chromIndex of a chain's last piece indeterminate (Chain.h:350-360): it compares unequal to
 * every other piece's here.  Synchronous.                                                                                                  */
typedef struct lra_hsplit_result {
  uint64_t n_jobs, n_pieces, n_elems;
  const uint64_t* d_job_piece_off; const uint64_t* d_piece_off; const uint32_t* d_sptc; const uint8_t* d_piece_type; const uint8_t* d_piece_strand;
  const uint32_t* d_piece_box; const uint32_t* d_piece_job; const uint32_t* d_job_lsc;
} lra_hsplit_result;
int lra_split_chains_highacc_batch(lra_ctx* ctx, uint64_t n_jobs, const uint64_t* d_job_off, uint64_t n_elems, const int32_t* d_strand, const int32_t* d_chrom,
                                   const uint32_t* d_box, const uint64_t* d_link_off, const uint8_t* d_link, int splitdist, lra_hsplit_result* out);

/* ---- GlobalChain / PrioritySearchTree (named by the path's description; not reachable from lra.cpp) -------------------------------------------
 * Replaces   int GlobalChain(vector<Fragment>& fragments, vector<int>& optFragmentChainIndices, vector<Endpoint>& endpoints)   (GlobalChain.h:85-189,
 *            PrioritySearchTree.h; called by TestGlobalChain.cpp:17 only)
 * for n_sets independent fragment sets: set s = fragments d_off[s] .. d_off[s+1] (xl, yl, xh, yh, initial score -- TestGlobalChain.cpp:14 uses xh - xl).
 * Output (context-owned): per fragment its final score and prev (index inside the set, -1 = none); per set the optimal chain's fragment indices
 * d_chain[d_off[s] .. d_off[s] + d_chain_len[s]).  Synchronous.                                                                                  */
typedef struct lra_global_chain_result {
  uint64_t n_sets, n_fragments;
  const int32_t* d_score; const int32_t* d_prev; const int32_t* d_chain; const uint32_t* d_chain_len;
} lra_global_chain_result;
int lra_global_chain_batch(lra_ctx* ctx, uint64_t n_sets, const uint64_t* d_off, uint64_t n_fragments, const int32_t* d_xl, const int32_t* d_yl, const int32_t* d_xh,
                           const int32_t* d_yh, const int32_t* d_score, lra_global_chain_result* out);

/* ---- a13 helper (high-accuracy path): SwitchToOriginalAnchors ------------------------------------------------------------------------
 * Replaces   SwitchToOriginalAnchors(finalchain, ultimatechain, ExtendClusters, extend_clusters)      (LocalRefineAlignment.h:187-199, :576)
 * for n_chains chains over Cluster_SameDiag entries: chain c = elements d_chain_off[c] .. d_chain_off[c+1], element i = entry d_elem_entry[i]
 * of cluster d_elem_cluster[i] (FinalChain::chain / ClusterNum); same_diag = the lra_merge_same_diag_batch result the entries refer to;
 * d_coarse[cluster] = Cluster_SameDiag::coarse.  Output: per chain the original anchors (index inside their cluster, entry by entry from its
 * last anchor to its first) and their ClusterIndex, at d_chain_off[c] .. d_chain_off[c+1] of the result.  Synchronous.                   */
typedef struct lra_original_anchors_result {
  uint64_t n_chains, n_anchors;
  const uint64_t* d_chain_off; const uint32_t* d_anchor; const int32_t* d_cluster;
} lra_original_anchors_result;
int lra_switch_to_original_anchors_batch(lra_ctx* ctx, uint64_t n_chains, const uint64_t* d_chain_off, uint64_t n_elems, const int32_t* d_elem_cluster,
                                         const uint32_t* d_elem_entry, const lra_same_diag_result* same_diag, const int32_t* d_coarse,
                                         lra_original_anchors_result* out);

/* ---- a9 (high-accuracy path): switchindex ---------------------------------------------------------------------------------------------
 * Replaces   switchindex(splitclusters, Primary_chains, clusters, genome, read)                (Mapping_ultility.h:39-168, Map_highacc.h:274)
 * for n_chains chains (every CHain of every Primary_chain of every read): chain c = d_ch[d_chain_off[c] .. d_chain_off[c+1]) (split-cluster
 * indices relative to its read, as lra_sparse_dp_boxes_batch returns them) with d_n_link[c] link bits at d_link[d_chain_off[c] ..];
 * d_split_base[c] / d_cluster_base[c] = where its read's split clusters (d_coarse) / clusters (d_cl_qs, d_cl_qe = Cluster::qStart, qEnd)
 * begin; n_total = d_chain_off[n_chains].  Output (context-owned, same offsets): the rewritten chain d_ch[d_chain_off[c] .. + d_n[c]) of
 * cluster indices and d_n_link[c] links; d_status[c] = LRA_ST_OOB_SLOT where the reference would index past a vector.  Synchronous.     */
typedef struct lra_switchindex_result {
  uint64_t n_chains;
  const uint32_t* d_ch; const uint8_t* d_link; const uint32_t* d_n; const uint32_t* d_n_link; const uint32_t* d_status;
} lra_switchindex_result;
int lra_switchindex_batch(lra_ctx* ctx, uint64_t n_chains, const uint64_t* d_chain_off, const uint32_t* d_ch, const uint8_t* d_link,
                          const uint32_t* d_n_link, const uint64_t* d_split_base, const uint64_t* d_cluster_base, const int32_t* d_coarse,
                          const uint32_t* d_cl_qs, const uint32_t* d_cl_qe, uint64_t n_total, lra_switchindex_result* out);

/* ---- the path behind one call: MapRead_lowacc for a batch of reads -----------------------------------------------------------------
 * Replaces   int MapRead_lowacc(const vector<float>& LookUpTable, Read& read, Genome& genome, vector<GenomeTuple>& genomemm, LocalIndex& glIndex,
 *                               const Options& opts, ostream* output, ostream* svsigstrm, Timing& timing, IndelRefineBuffers&, pthread_mutex_t*)
 *            (Map_lowacc.h:33-640), which MapRead (MapRead.h:153-263) enters for the low-accuracy presets (-ONT, -CLR), called by the MapReads
 *            worker (lra.cpp:117) and the serial loop (lra.cpp:721) once per read.
 * The reference side (shared, read-only in the reference too) is loaded once per context:
 *   lra_ctx_load_genome        genome.seqs back to back                                  (Genome.h:122-137)
 *   lra_ctx_load_chromosomes   genome.header.pos, n_chrom + 1 cumulative starts          (Genome.h:60-83)
 *   lra_ctx_load_global_index  genomemm, the .mms payload                                (MMIndex.h:416)
 *   lra_ctx_build_local_index  glIndex: LocalIndex::IndexSeq of every chromosome on the device (the .gli payload; MMIndex.h:200-254)
 * lra_map_reads_lowacc_batch aligns n_reads reads (d_seq: upper-case bases back to back, >= 64 bytes of padding behind the last read;
 * d_read_off: n_reads + 1 offsets; total_bases = d_read_off[n_reads]) and leaves, in context-owned buffers valid until the next call:
 *   job j = read j / num_aln, primary chain j % num_aln (the loop over chains, Map_lowacc.h:232); its SegAlignments are alignments
 *   d_job_aln_off[j] .. d_job_aln_off[j+1] (empty when the chain produced none); per alignment: read, strand, Supplymentary, ISsecondary,
 *   NumOfAnchors0 / 1, chromIndex, FirstSDPValue (value before CalculateStatistics), the refined blocks (IndelRefineAlignment),
 *   CalculateStatistics' 18 counters (order of lra_calculate_statistics_batch), NV value and CIGAR runs ((length << 4) | op, op in =XID).
 *   d_job_status[j] != 0 / d_refine_status[a] != 0: the reference would read outside an array there (see the stage headers).
 * lra_map_records is the per-read tail (SetFromSegAlignment, AlignmentsOrder::Update, SimpleMapQV, OUTPUT / output_unaligned,
 * Map_lowacc.h:600-618) on the host: text of all reads in input order, two-call convention; rec_off (nullable): n_reads + 1 offsets of
 * each read's records inside the text.  names / reads / quals (nullable, or NULL / "*" entries) / read_len: per read, host.            */
#define LRA_READ_ONT 0
#define LRA_READ_CLR 1
#define LRA_READ_CCS 2
#define LRA_READ_CONTIG 3
typedef struct lra_map_opts {
  int32_t globalK, globalW, globalMaxFreq;
  int32_t localK, localW, localMaxFreq, localIndexWindow;
  int32_t refineBand, localMatch, localMismatch, localIndel, localBand;
  int32_t refineSpaceDist; float anchorstoosparse; int32_t splitdist, window;
  float second_anchorbonus; int32_t bypassClustering, skipBandedRefine, refineBreakpoint;   /* refineBreakpoint: --refineBreakpoints (lra.cpp:262) */
  lra_clean_opts clean; lra_sdp_opts sdp;
  int32_t readType, hardClip, PrintNumAln, printFormat;   /* printFormat: 's' SAM, 'p' / 'P' PAF, 'b' BED, 'a' pairwise (PrintPairwise) */
  lra_fine_opts fine; int32_t merge_dist;                 /* high-accuracy path only: MatchesToFineClusters, MergeMatchesSameDiag */
  int32_t defer_matches;                                  /* low-accuracy path, scheduling only (results do not depend on it): a read with more refined matches than this
                                                           * after Refine_Btwnsplitchain goes on in a second, concurrent pass; 0 = one pass (the presets) */
  int32_t flagged_unaligned;                              /* what lra_map_records* write for a read whose status word is non-zero (an LRA_ST_* condition in some stage: the
                                                           * result is not known to equal the reference's): 0 (the presets) = an empty record, the caller re-runs those reads
                                                           * elsewhere (counters.n_flagged_reads / lra_map_host_flagged say how many and which); 1 = the read's unaligned
                                                           * record (output_unaligned, Mapping_ultility.h:457-463: a flag-4 line in SAM mode), so that the output keeps one
                                                           * record per input read */
  int32_t defer_seed_matches;                             /* low-accuracy path, scheduling only (a read's result does not depend on the batch it is mapped in): a read with
                                                           * more tier-1 matches than this (CompareLists against the global index; a 30 kb read has ~3 k, a read from a
                                                           * satellite array 6-10 k) is HANDED BACK: it leaves the batch behind the seed stage, d_read_status[r] has
                                                           * LRA_ST_DEFERRED (and nothing else), counters.n_handed_back_reads counts them, lra_map_records* write nothing
                                                           * for it.  The caller collects such reads and maps them as batches of their own (with this field 0): their
                                                           * sparse DPs are tens of times larger than a typical read's and would otherwise set the length of every
                                                           * latency-bound launch of the batch they sit in.  0 = off (the presets) */
} lra_map_opts;
typedef struct lra_map_counters {
  uint64_t n_minimizers, n_matches, n_clusters, n_sdp_anchors, n_sdp_points, n_sdp_entries, n_local_tuples, n_local_tasks, n_local_task_words, n_local_pairs, n_refined_matches,
           n_btwn_problems, n_btwn_rounds, n_refined_after_btwn, n_merged_clusters, n_sdp2_anchors, n_sdp2_entries, n_a13_blocks, n_large_spaces, n_segments,
           n_rows, n_cells, n_aog, n_deferred_reads,
           n_flagged_reads,                                  /* reads of the batch with a non-zero d_read_status (no alignment record is written for them) */
           n_handed_back_reads;                              /* (ABI 6) reads handed back UNMAPPED by the seed stage under opts.defer_seed_matches (status LRA_ST_DEFERRED, no record):
                                                              * what a caller sizes its pool of handed-back reads with.  n_deferred_reads counts only the reads that opts.defer_matches
                                                              * moved to the batch's second pass -- those ARE mapped and have their records */
} lra_map_counters;
typedef struct lra_map_result {
  int32_t n_reads, num_aln;
  uint64_t n_jobs, n_alignments, n_blocks, n_runs;
  const uint64_t* d_job_aln_off; const uint32_t* d_job_status;
  const uint8_t* d_job_reached;     /* [n_jobs] 1: primary chain p reached Map_lowacc.h:574 (its SegAlignmentGroup exists, possibly empty).  The reference's loop over p ENDS at
                                     * the first chain that does not (p > 0: break, :267 / :491; p == 0: the read is unaligned): the low-accuracy driver maps no chain behind
                                     * it (their flags are 0, they have no alignments).  One rule is left to the reader of these arrays, as lra_map_records* apply it: a read
                                     * whose chain 0 reached :574 but got no SegAlignment is unaligned whatever its later chains hold (:578-581) */
  const uint32_t* d_read_status;    /* [n_reads] OR of every stage's LRA_ST_* bits for the read; non-zero = not bit-identical, no record is emitted */
  const uint32_t* d_aln_read; const int32_t* d_strand; const int32_t* d_supp; const int32_t* d_secondary; const int32_t* d_n0; const int32_t* d_n1;
  const int32_t* d_chrom; const float* d_first_sdp_value;
  const uint64_t* d_block_off; const int32_t* d_blocks; const int32_t* d_refine_status;
  const int32_t* d_counts; const float* d_value; const uint64_t* d_run_off; const uint32_t* d_runs;
  const char* d_strands; uint64_t rc_base;                 /* the reads forward, then (at rc_base) reverse complemented */
  lra_map_counters counters;
} lra_map_result;
void lra_map_opts_preset_ont(lra_map_opts* opts);          /* -ONT: lra.cpp:386-431 over Options.h:127-230 */
void lra_map_opts_preset_clr(lra_map_opts* opts);          /* -CLR: lra.cpp:341-386 */
int lra_ctx_load_chromosomes(lra_ctx* ctx, const uint64_t* h_chrom_pos, int n_chrom);
int lra_ctx_build_local_index(lra_ctx* ctx, int k, int w, int window, int max_freq);
/* glIndex's k, w and localIndexWindow belong to the INDEX, not to the options: LocalIndex::Read takes them from the .gli file (MMIndex.h:154-173, lra.cpp:627), the reads'
 * indexes copy them (LocalIndex(LocalIndex&), MMIndex.h:128-136, Map_lowacc.h:246-247) and smallOpts.globalK / globalW are glIndex.k / w (Map_lowacc.h:233-234,
 * Map_highacc.h:430-431).  `lra index` writes k = 10, w = 5, windows of 2048 bases under EVERY preset (`LocalIndex glIndex;` lra.cpp:989 -> LocalIndex(0): 1 <<
 * (LOCAL_POS_BITS - 1), MMIndex.h:110-127; RunStoreLocal keeps k = 10, lra.cpp:785-817), so an `lra align` run on indexed files maps with those; only without a .gli file
 * does it build glIndex from opts.localK (10 for -ONT / -CLR, 7 for -CCS / -CONTIG) and opts.localIndexWindow = 256 (lra.cpp:619-621, :628) -- the values the presets
 * below hold.  lra_map_opts_apply_local_index writes an index's three values into the options (localK, localW, localIndexWindow); lra_ctx_local_index_params returns the
 * ones the context's index was built with.  The drivers refuse options that differ from the context's index (LRA_ERR_INVALID).                                        */
void lra_map_opts_apply_local_index(lra_map_opts* opts, int k, int w, int window);
/* glIndex as LocalIndex::Read left it (MMIndex.h:154-173, lra.cpp:627): the .gli payload handed over as it is (host arrays, copied) instead of lra_ctx_build_local_index --
 * glIndex.k / w / localIndexWindow, seqOffsets[n_windows + 1], tupleBoundaries[n_windows + 1], minimizers[n_tuples].  The seqOffsets must be those IndexSeq writes for the
 * loaded chromosome table at this window (an index of another genome is LRA_ERR_INVALID).                                                                            */
int lra_ctx_load_local_index(lra_ctx* ctx, int k, int w, int window, uint64_t n_windows, const uint64_t* h_seq_offsets, const uint64_t* h_tuple_bnd,
                             uint64_t n_tuples, const uint32_t* h_tuples);
int lra_ctx_local_index_params(lra_ctx* ctx, int* k, int* w, int* window);
/* The context's reference data as device pointers: the genome bytes; the genome's local index (the .gli payload: d_tuple_bnd[n_windows + 1],
 * d_tuples[n_tuples]) and its seqOffsets[n_windows + 1].  Valid until the context is destroyed or the data is loaded / built again.        */
/* Several contexts on one GPU (sub-batches on their own HIP streams) share ONE replica of the reference: dst borrows src's genome, global
 * index, chromosome table and local index; src must outlive dst; dst must not hold reference data of its own.                              */
int lra_ctx_share_reference(lra_ctx* dst, lra_ctx* src);
const char* lra_ctx_genome_ptr(lra_ctx* ctx);
int lra_ctx_local_index(lra_ctx* ctx, lra_local_index_result* out, const uint64_t** d_seq_offsets);
int lra_map_reads_lowacc_batch(lra_ctx* ctx, int n_reads, const char* d_seq, const uint64_t* d_read_off, uint64_t total_bases, const lra_map_opts* opts,
                               lra_map_result* out);
/* Two-stage batches (low-accuracy presets): lra_map_reads_lowacc_batch in two halves, so that the FRONT half of batch i + 1 runs beside the BACK half of batch i.
 *   front: a1 .. the second LinearExtend / TrimOverlappedAnchors (Map_lowacc.h:69-476), on `ctx` and its stream, by one host thread;
 *   back : the second sparse DP, LocalRefineAlignment, IndelRefineAlignment, CalculateStatistics (Map_lowacc.h:477-599), on the context's companion context
 *          (made on first use: its own stream -- the device's highest priority unless LRA_BACK_PRIORITY says otherwise --, its own work buffers, the reference data shared), by ANOTHER host thread.
 * Between the halves sits a queue of ONE batch (ABI 6; before: none -- the front half waited for the running back half): the front half writes what it hands over
 * into one of two sets of handover buffers, taken in turn, and lra_map_reads_lowacc_front returns when the batch is handed over; it waits, at its very end, only until
 * the batch BEFORE its own has been taken by a back call -- the back half that is running is not waited for, so the back context goes from one batch straight to the
 * next.  lra_map_reads_lowacc_back waits for a handed-over batch, runs its back half and returns the result of the whole batch exactly as lra_map_reads_lowacc_batch
 * would (same alignments, same counters); the result's arrays belong to *back_ctx: lra_map_pack / lra_map_snapshot / lra_map_records are called on THAT context, then
 * lra_map_back_release(ctx) ends the result's lifetime (the next back call needs it: LRA_ERR_INVALID while a result is held).  Up to three batches are in flight (one in
 * its back half, one handed over, one in its front half): a batch's reads (d_seq, d_read_off) stay untouched until ITS back call has returned.  Per thread the calls are
 * in batch order: front(0), front(1), ... on one, back(0), release(0), back(1), ... on the other.  Scheduling only; opts.defer_matches and opts.defer_seed_matches do not
 * combine with it (LRA_ERR_INVALID).  The reference's counterpart is its pool of worker threads (lra.cpp:678-714): several reads in flight at different points of MapRead.
 * Errors (ABI 6): a front call that FAILS -- whatever the reason: reference not loaded, out of memory, a stage's error -- still hands over a batch, an ERROR batch: the
 * back call that takes it runs nothing, returns the front call's code and holds nothing (NO lra_map_back_release for it; one would return LRA_ERR_INVALID).  So the
 * contract for the two host threads is one back call per front call, whatever either returned; the thread of the back halves is never left waiting for a batch that
 * does not come, and the next front call is not blocked by a failed one.  A back call whose own half fails returns its code with the result slot still held: release
 * it as after a success.  The back calls leave their error text on the back context (lra_ctx_last_error(*back_ctx); the front thread owns ctx's).
 * The back context's view of the reference data (borrowed from ctx) is refreshed by each back call before it runs, on its own thread; ctx's reference must not be
 * loaded / built again while a batch is in flight (the halves in flight read it). */
int lra_map_reads_lowacc_front(lra_ctx* ctx, int n_reads, const char* d_seq, const uint64_t* d_read_off, uint64_t total_bases, const lra_map_opts* opts);
int lra_map_reads_lowacc_back(lra_ctx* ctx, const lra_map_opts* opts, lra_map_result* out, lra_ctx** back_ctx);
int lra_map_back_release(lra_ctx* ctx);

/* The same boundary for the high-accuracy presets: MapRead_highacc (Map_highacc.h:37-798) behind MapRead (MapRead.h:169-239), which the reference enters
 * when opts.bypassClustering == 0 (-CCS, -CONTIG).  Arguments and result as lra_map_reads_lowacc_batch; job j = read j / num_aln, chain h = j % num_aln of
 * Primary_chains[0] (num_aln = opts.NumAln); d_job_reached[j] = the chain has clusters and got its SegAlignmentGroup (:697-699); d_first_sdp_value =
 * Primary_chains[0].chains[h].value.  The counters of CalculateStatistics are what the reference's two calls (:721, :731) leave: tdel, tins and the six size
 * classes accumulate over both.  Needs the genome, the chromosome table and the global index; the genome's local index only when a read takes the REFINEclusters
 * branch (:413-447: a second pass over those reads with K = glIndex.k, merged into the result; their d_job_reached carries bit 1).  lra_map_records / lra_map_snapshot / lra_map_pack serve both paths
 * (opts.bypassClustering tells the record stage which tail to follow).                                                                          */
void lra_map_opts_preset_ccs(lra_map_opts* opts);          /* -CCS: lra.cpp:306-340 */
void lra_map_opts_preset_contig(lra_map_opts* opts);       /* -CONTIG: lra.cpp:268-305 */
int lra_map_reads_highacc_batch(lra_ctx* ctx, int n_reads, const char* d_seq, const uint64_t* d_read_off, uint64_t total_bases, const lra_map_opts* opts,
                                lra_map_result* out);
/* ---- the input side ---------------------------------------------------------------------------------------------------
 * lra_reads_open / lra_reads_next_batch / lra_reads_close replace Input::Initialize, Input::GetNext (FASTA and FASTQ) and Input::BufferedRead (Input.h:87-168,
 * :182-283, :405-421): files are read one after the other; a batch takes reads while it holds fewer than max_bases bases (so it ends with the read that
 * crosses the limit).  The batch's arrays are owned by the reader and valid until the next call: seq = the reads' bases, upper-cased, back to back + 64 bytes
 * of padding; off[n_reads + 1]; names / reads / quals / read_len as lra_map_records takes them (quals[i] == NULL for FASTA reads).
 * A FASTQ record whose quality string has another length than its read (the reference asserts, Input.h:287) makes lra_reads_next_batch return LRA_ERR_INVALID
 * on that call and on every later one: the batch of that call holds the reads in front of the record, lra_reads_last_error names the record.
 * lra_map_reads_host is the boundary with host buffers: it copies the batch to the device and calls the driver opts->bypassClustering selects.
 * Not supported: streamed input ("-", "stdin", "/dev/stdin": the format sniffing seeks; the reference reads those through htslib) and BAM input (htslib).  */
typedef struct lra_reads lra_reads;
typedef struct lra_read_batch {
  int32_t n_reads; uint64_t total_bases;
  const char* seq; const uint64_t* off; const int32_t* read_len;
  const char* const* names; const char* const* reads; const char* const* quals;
} lra_read_batch;
int lra_reads_open(const char* const* files, int n_files, lra_reads** out);
int lra_reads_next_batch(lra_reads* r, uint64_t max_bases, lra_read_batch* batch);
const char* lra_reads_last_error(const lra_reads* r);
uint64_t lra_map_host_trim(uint64_t keep_bytes);   /* (ABI 6) the record stage keeps its threads' text parts between batches (process-wide, at most LRA_PARTS_POOL_MB, default 4096, of
                                                     * host memory; a fresh 100 MB part is 25 000 page faults): release them down to keep_bytes (0 = all); returns the bytes still held */
int lra_host_thread_budget(void);   /* host threads lra_map_records* use when asked for 0: hardware threads, capped by the container's CPU quota (cgroup cpu.max) less four */
void lra_reads_close(lra_reads* r);
int lra_map_reads_host(lra_ctx* ctx, int n_reads, const char* h_seq, const uint64_t* h_off, const lra_map_opts* opts, lra_map_result* out);
int lra_map_records(lra_ctx* ctx, const lra_map_result* res, const lra_map_opts* opts, const char* const* names, const char* const* reads,
                    const char* const* quals, const int32_t* read_len, const char* const* chrom_names, const char* passthrough, char* out, uint64_t cap,
                    uint64_t* len, uint64_t* rec_off);
/* lra_map_records in two halves, so that the host tail of batch i runs beside the device side of batch i + 1 (the reference interleaves them per
 * thread, lra.cpp:117-158):
 *   lra_map_snapshot      copies what the records need (per-alignment fields, counters, CIGAR runs, block ends; with_blocks != 0 also every block
 *                         and the chromosome text under it, needed by print format 'a' only) from the context's result buffers to a host object;
 *                         after it returns the context may run the next batch.
 *   lra_map_records_host  SetFromSegAlignment / AlignmentsOrder::Update / SimpleMapQV / OUTPUT for every read on n_threads host threads (0: up to
 *                         16); touches neither the context nor the device.  *text (owned by the snapshot, valid until it is freed or reused),
 *                         *len bytes, (*rec_off)[n_reads + 1] record boundaries; a read with a non-zero status word has an empty record.
 *   lra_map_host_free     releases the snapshot.                                                                                              */
typedef struct lra_map_host lra_map_host;
int lra_map_snapshot(lra_ctx* ctx, const lra_map_result* res, int with_blocks, lra_map_host** out);
int lra_map_records_host(lra_map_host* snap, const lra_map_opts* opts, const char* const* names, const char* const* reads, const char* const* quals,
                         const int32_t* read_len, const char* const* chrom_names, const char* passthrough, int n_threads, const char** text, uint64_t* len,
                         const uint64_t** rec_off);
void lra_map_host_free(lra_map_host* snap);
/* the reads of a snapshot whose status word is non-zero: their number; *status (optional) = the snapshot's status array [n_reads], owned by the snapshot */
uint64_t lra_map_host_flagged(const lra_map_host* snap, const uint32_t** status);
/* The record buffer of a batch as ONE device buffer -- what a rank sends to rank 0 in the single exchange step of the multi-GPU path (the
 * reference's ordered output, lra.cpp:145-166): lra_map_pack lays the same arrays out behind a 128-byte header in a context-owned buffer
 * (valid until the next pack on this context); lra_map_unpack_host turns a host copy of such a buffer (from any rank) into a snapshot for
 * lra_map_records_host.  lra_map_snapshot = pack + copy to the host + unpack.                                                               */
int lra_map_pack(lra_ctx* ctx, const lra_map_result* res, int with_blocks, const void** d_buf, uint64_t* bytes);
int lra_map_unpack_host(const void* h_buf, uint64_t bytes, lra_map_host** out);

#ifdef __cplusplus
}
#endif
#endif /* LRA_HIP_H_ */
