// lra_amd/csrc/common.h -- shared host-side plumbing of liblra_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <memory>
#include <string>
#include <string.h>
#include <vector>
#include "../../include/lra_hip.h"

struct lra_seed_state;
struct lra_cluster_state;
struct lra_map_state;
struct lra_handover;
void lra_handover_free(lra_ctx* ctx);   // mapread.hip
// The generation of a context's reference data, shared (refcounted) between the owner and every context that borrows from it, so that a borrower's staleness
// check never reads the owner's state: `gen` is bumped by each of the owner's loaders, `dead` is set when the owner is destroyed (its device buffers are freed).
struct lra_gen_cell { std::atomic<uint64_t> gen{0}; std::atomic<bool> dead{false}; };
struct lra_time_rec { const char* name; hipEvent_t a, b; hipStream_t stream; };

struct lra_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  // growable device scratch arenas (never shrunk; freed in lra_ctx_destroy)
  void* scratch[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t scratch_bytes[4] = {0, 0, 0, 0};
  int num_cu = 256;
  lra_seed_state* seed = nullptr;
  lra_cluster_state* clus = nullptr;
  lra_map_state* map = nullptr;                       // mapread.hip: chromosome table, the genome's local index
  void* aux = nullptr; size_t aux_bytes = 0;          // AffineOneGapAlign blocks of refine fallbacks
  void* out_buf = nullptr; size_t out_bytes = 0;
  void* pin_buf = nullptr; size_t pin_bytes = 0;      // page-locked host staging of the large device-to-host copies (lra_pinned)
  uint64_t* scan_tmp = nullptr;
  // lra_side_fork / lra_side_join: further streams for kernels that would only extend a stage's tail on the context's own stream
  static constexpr int N_SIDE = 4;
  hipStream_t side[N_SIDE] = {}; hipEvent_t ev_fork = nullptr, ev_join[N_SIDE] = {}, ev_mid = nullptr;   // ev_mid: a point on a side stream the main stream waits for (sdp.hip: the large reads' build)
  void* gbuf[192] = {};   // growable result / work buffers (lra_ensure)
  size_t gbytes[192] = {};
  // kernel timing
  bool sdp_inner = false;                    // local_refine.hip: its small inner sparse DP is timed under "sdp_inner_*"
  bool sort_short = false;                   // the exact sort: a launch of its own for the short lists (set by the sparse DP around its sorts: hundreds of tuples per list)
  const char* sort_tag = "sort"; const char* sort_fb_tag = "sort_fallback";   // timing names of the exact-sort kernels (sdp.hip retags them)
  bool timing = false;
  std::vector<lra_time_rec> recs;
  std::vector<hipEvent_t> free_events;
  lra_ctx* child = nullptr;                  // mapread.hip: the context of a batch's second, concurrent pass (shares this one's reference; destroyed with it)
  bool owns_stream = false;
  bool pipelined = false;                    // two-stage batches (lra_map_reads_lowacc_front / _back): another batch's half runs beside this context's launches, so the
                                             // sparse DP chooses for device time (fewer workgroup-per-read jobs: one per CU) rather than for the shortest tail
  struct lra_handover* handover = nullptr;   // mapread.hip: lra_map_reads_lowacc_front / _back (a batch between its two halves)
  // lra_seed_prefetch / lra_ctx_adopt_seed: a seed result made ahead of its batch -- on the side context the one lra_seed_prefetch left, on the mapping context the
  // one it adopted and lra_map_reads_lowacc_batch / lra_map_reads_highacc_batch will use instead of seeding when they are called with the same reads
  struct { bool valid = false; int n_reads = 0; const char* d_seq = nullptr; const uint64_t* d_read_off = nullptr; int k = 0, w = 0, max_freq = 0; lra_seed_result res; } ahead;
  bool low_priority = false; int prio = 0;   // the second pass's streams (its side streams too) are created with the device's lowest priority
};

// Fork: work queued on the returned stream starts after everything queued on ctx->stream so far; join: ctx->stream waits for it.
hipStream_t lra_side_fork(lra_ctx* ctx, int i = 0);
void lra_side_join(lra_ctx* ctx, int i = 0);
// segsort.hip: rocprim::segmented_radix_sort_pairs' interface for (uint64 key, uint32 value) pairs; segments of 257 .. 8192 pairs are sorted by one workgroup in LDS
hipError_t lra_segsort_pairs(lra_ctx* ctx, void* temp, size_t& temp_bytes, const uint64_t* kin, uint64_t* kout, const uint32_t* vin, uint32_t* vout, unsigned int total,
                             unsigned int nseg, const uint64_t* b, const uint64_t* e, int begin_bit, int end_bit, hipStream_t st);
// timing records (lra_ctx_timing_get); stream = nullptr: the context's stream
void lra_time_begin(lra_ctx* ctx, const char* name, hipStream_t stream = nullptr);
void lra_time_end(lra_ctx* ctx, hipStream_t stream = nullptr);
void lra_seed_free(lra_ctx* ctx);
void lra_cluster_free(lra_ctx* ctx);
void lra_map_free(lra_ctx* ctx);

int lra_set_err(lra_ctx* ctx, int code, const char* fmt, ...);
// How many host threads a burst of work may use: the hardware threads, or -- inside a container with a CPU bandwidth quota (cgroup cpu.max / cfs_quota_us) -- the
// quota's whole CPUs less four (the driving thread, the runtime's helper threads and a tail thread need theirs).  More runnable threads than the quota use the period's budget up in a few milliseconds and the kernel then throttles EVERY thread of the
// process until the next period, the one that drives the device included (measured: 128 record threads on a 16-CPU quota = a 30-55 ms hole in every step)
int lra_host_threads();
// returns a device buffer of >= bytes (slot 0..3), growing it if needed (synchronises the
// stream before freeing the old one)
void* lra_scratch(lra_ctx* ctx, int slot, size_t bytes);
// page-locked host buffer of at least `bytes`, kept by the context: the landing place of the large device-to-host copies.  A copy into pageable memory makes
// the runtime pin and unpin the destination's pages, and the unpin stalls every queue of the process for tens of milliseconds (0.7 GB of records per batch: ~50 ms)
void* lra_pinned(lra_ctx* ctx, size_t bytes);
// growable buffer `idx` of at least `bytes` (contents NOT preserved on growth)
void* lra_ensure(lra_ctx* ctx, int idx, size_t bytes);

// seed.hip: std::sort-identical segmented sort of (key, payload) lists whose keys rarely repeat inside a list
int lra_sort_mostly_unique_batch(lra_ctx* ctx, int n_lists, const uint64_t* d_off, uint64_t total, uint64_t* d_key, uint32_t* d_pos,
                                 uint64_t* tmp_key, uint32_t* tmp_pos, int end_bit);

// cluster.hip: LinearExtend (pair version, LinearExtend.h:658) + DecideCoordinates box on caller-supplied clusters of diagonal-sorted matches
int lra_launch_linear_extend(lra_ctx* ctx, uint64_t n_clusters, int K, const uint64_t* c_start, const uint64_t* c_end, const int* c_strand, const int* c_chrom,
                             int* c_read, const uint32_t* cl_q, const uint32_t* cl_t, const uint64_t* d_chrom_pos, const unsigned char* genome,
                             const unsigned char* seq, const uint64_t* read_off, uint32_t* e_q, uint32_t* e_t, int* e_len, uint32_t* e_count, uint32_t* box,
                             const int* c_K = nullptr);

// emit.hip / rank.hip: the record text appended to a string (the C entry points lra_format_* / lra_output_read wrap these)
int lra_format_bed_str(const lra_aln_record* r, std::string& s_);
int lra_format_paf_str(const lra_aln_record* r, int print_cigar, std::string& s_);
int lra_format_sam_str(const lra_aln_record* g, int n_group, int as, int hard_clip, const char* passthrough, std::string& s_);
int lra_format_sam_simple_str(const lra_aln_record* rp, int hard_clip, const char* passthrough, std::string& s_);
int lra_output_read_str(const lra_aln_group* groups, const int32_t* index, int n_groups, lra_aln_record* recs, int print_num_aln, char format, int hard_clip,
                        const char* passthrough, int read_unaligned, const lra_aln_record* unaligned_rec, std::string& text);

#define LRA_HIP_CHECK(ctx, call)                                                         \
  do {                                                                                   \
    hipError_t e__ = (call);                                                             \
    if (e__ != hipSuccess)                                                               \
      return lra_set_err(ctx, LRA_ERR_HIP, "%s failed: %s (%s:%d)", #call,               \
                         hipGetErrorString(e__), __FILE__, __LINE__);                    \
  } while (0)
