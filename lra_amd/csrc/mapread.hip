// lra_amd/csrc/mapread.hip -- the drop-in boundary of the path: MapRead_lowacc for a batch of reads behind ONE call (gfx950 only).
//
// Replaces, for n_reads reads at a time, the body of
//     int MapRead_lowacc(LookUpTable, Read&, Genome&, genomemm, glIndex, opts, output, svsigstrm, timing, indelRefineBuffers, semaphore)
// (reference: Map_lowacc.h:33-640, entered from MapRead, MapRead.h:169-263) between "the read's bases" and "its alignments with their
// statistics" (lra_map_reads_lowacc_batch), and the per-read tail SetFromSegAlignment / AlignmentsOrder::Update / SimpleMapQV / OUTPUT
// (Map_lowacc.h:600-618; lra_map_records).  The stages are the library's own batched entry points, called in the reference's order; the
// only work done here is the glue the reference does with std::vector moves: keeping NumOfAnchors0 of the first sparse DP, building the
// forward + reverse-complement read buffer, and addressing every alignment's strand / chromosome for IndelRefineAlignment and
// CalculateStatistics.
#include "common.h"
#include "seed_state.h"
#include "scan.h"
#include "map_state.h"
#include "map_merge.h"
#include <chrono>
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>

void lra_map_free(lra_ctx* ctx) {
  lra_map_state* m = ctx->map;
  if (!m) return;
  if (!m->borrowed) {
    m->cell->dead = true;                                                  // borrowers hold the cell, not this state
    if (m->d_chrom_pos) (void)hipFree(m->d_chrom_pos);
    if (m->gli_buf) (void)hipFree(m->gli_buf);
    if (m->d_gso) (void)hipFree(m->d_gso);
  }
  delete m;
  ctx->map = nullptr;
}

namespace {

lra_map_state* map_state(lra_ctx* ctx) {
  if (!ctx->map) ctx->map = new lra_map_state();
  return ctx->map;
}

// a loader on a context that borrows its reference data: drop the owner's pointers, own what is loaded from here on
void map_disown(lra_map_state* m) {
  if (!m->borrowed) return;
  m->d_chrom_pos = nullptr; m->gli_buf = nullptr; m->gli = lra_local_index_result{}; m->d_gso = nullptr; m->n_gwin = 0; m->gli_window = 0;
  m->borrowed = false; m->owner_cell.reset(); m->owner_generation = 0;
}

__global__ void k_add_off(int n, const uint64_t* __restrict__ off, uint64_t add, uint64_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= n) out[i] = off[i];                       // [0..n]: the reads forward
  if (i >= 1 && i <= n) out[n + i] = off[i] + add;   // [n+1..2n]: their reverse complements
}

// Map_lowacc.h:86-89, :184-185: a read with a cluster of anchorfreq in (1, 2] and >= 500 matches runs its first sparse DP with anchor bonus 3
// instead of opts.initial_anchorbonus.  One wave per read over its clusters.
__global__ void __launch_bounds__(64) k_match_rate(int n_reads, const uint64_t* __restrict__ cluster_off, const uint64_t* __restrict__ c_start,
                                                   const uint64_t* __restrict__ c_end, const float* __restrict__ anchorfreq, float rate, float* __restrict__ out) {
  const int r = blockIdx.x;
  if (r >= n_reads) return;
  bool rep = false;
  for (uint64_t c = cluster_off[r] + threadIdx.x; c < cluster_off[r + 1]; c += 64) {
    const float f = anchorfreq[c];
    if (f > 1.0f && f <= 2.0f && c_end[c] - c_start[c] >= 500) rep = true;
  }
  const bool any = __ballot(rep) != 0;
  if (threadIdx.x == 0) out[r] = any ? 3.0f : rate;
}

// ---- per-read status word: the OR of every stage's per-item status (LRA_ST_* bits), so that no flagged item is emitted as an ordinary record
__global__ void k_or_status_div(uint64_t n, const uint32_t* __restrict__ status, int div, uint32_t* __restrict__ read_status) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && status[i]) atomicOr(&read_status[i / (uint64_t)div], status[i]);
}
__global__ void k_or_status_idx(uint64_t n, const uint32_t* __restrict__ status, const uint32_t* __restrict__ idx, int div, uint32_t* __restrict__ read_status) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && status[i]) atomicOr(&read_status[idx[i] / (uint32_t)div], status[i]);
}
// ---- the reads whose chains are far larger than anything else in the batch (the second, concurrent pass of lra_map_reads_lowacc_batch)
// load[r] = the number of refined matches of read r's split chains after Refine_Btwnsplitchain: what its second sparse DP will chain
__global__ void k_read_load(uint64_t n_slots, int num_aln, const uint32_t* __restrict__ n_chains, const uint64_t* __restrict__ chain_start,
                            const uint32_t* __restrict__ n_split, const uint32_t* __restrict__ sp_status, const uint64_t* __restrict__ match_off, uint32_t* __restrict__ load) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;
  const uint64_t r = s / (uint64_t)num_aln;
  if ((uint32_t)(s % (uint64_t)num_aln) >= n_chains[r] || sp_status[s] || !n_split[s]) return;
  const uint64_t c0 = chain_start[s];
  atomicAdd(&load[r], (uint32_t)(match_off[c0 + n_split[s]] - match_off[c0]));
}
// a read above the threshold leaves this pass: its split chains are marked (every later stage skips a marked slot), and so is its status word.  The second pass gets
// the mirror image: sp2 (only the deferred reads' slots are on), their job_reached flags, a clean status array
__global__ void k_mark_deferred(int n_reads, int num_aln, const uint32_t* __restrict__ load, uint32_t threshold, uint32_t* __restrict__ sp_status,
                                uint32_t* __restrict__ read_status, uint8_t* __restrict__ deferred, uint32_t* __restrict__ sp2, uint8_t* __restrict__ reached,
                                uint8_t* __restrict__ reached2, uint32_t* __restrict__ rstat2) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_reads) return;
  const bool d = load[r] > threshold && read_status[r] == 0;
  deferred[r] = d ? 1 : 0;
  rstat2[r] = 0;
  for (int h = 0; h < num_aln; h++) {
    const uint64_t s = (uint64_t)r * num_aln + h;
    sp2[s] = d ? sp_status[s] : (uint32_t)LRA_ST_DEFERRED;
    reached2[s] = d ? reached[s] : 0;
    if (d) { sp_status[s] |= LRA_ST_DEFERRED; reached[s] = 0; }
  }
  if (d) read_status[r] |= LRA_ST_DEFERRED;
}
// opts.defer_seed_matches: the seed stage's flags become LRA_ST_DEFERRED in the reads' status words (no record is written for such a read)
__global__ void k_mark_handed_back(int n_reads, const uint8_t* __restrict__ flag, uint32_t* __restrict__ read_status, unsigned long long* count) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const bool d = r < n_reads && flag[r];
  if (d) read_status[r] |= LRA_ST_DEFERRED;
  const unsigned long long m = __ballot(d);
  if ((threadIdx.x & (warpSize - 1)) == 0 && m) atomicAdd(count, (unsigned long long)__popcll(m));
}
__global__ void k_src_slot(uint64_t S, int na, const int32_t* __restrict__ inB, uint64_t* __restrict__ src) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  const int b = inB[s / (uint64_t)na];
  src[s] = b >= 0 ? (((uint64_t)b * na + s % (uint64_t)na) | lra_merge::FROM_B) : s;
}
__global__ void k_merge_reached(uint64_t S, const uint64_t* __restrict__ src, const uint8_t* __restrict__ A, const uint8_t* __restrict__ B, uint8_t* __restrict__ out) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  const uint64_t x = src[s];
  out[s] = (x & lra_merge::FROM_B) ? B[x & ~lra_merge::FROM_B] : A[x];
}
__global__ void k_merge_read_status(int R, const int32_t* __restrict__ inB, const uint32_t* __restrict__ A, const uint32_t* __restrict__ B, uint32_t* __restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  out[r] = inB[r] >= 0 ? B[inB[r]] : A[r];
}
// per-split arrays live at chain_start[s] + k, k < n_split[s] (the split / refined-cluster numbering of chain_split.hip, refine_splitchain.hip, refine_btwn.hip)
__global__ void k_or_status_split(uint64_t n_slots, int num_aln, const uint32_t* __restrict__ n_chains, const uint64_t* __restrict__ chain_start,
                                  const uint32_t* __restrict__ n_split, const uint32_t* __restrict__ sp_status, const uint32_t* __restrict__ status,
                                  uint32_t* __restrict__ read_status) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;
  const uint64_t r = s / (uint64_t)num_aln;
  if ((uint32_t)(s % (uint64_t)num_aln) >= n_chains[r] || sp_status[s]) return;
  uint32_t v = 0;
  for (uint64_t c = chain_start[s]; c < chain_start[s] + n_split[s]; c++) v |= status[c];
  if (v) atomicOr(&read_status[r], v);
}
// Which primary chains p reach `alignments.resize(alignments.size() + 1)` (Map_lowacc.h:574): the chain exists, SPLITChain +
// RemoveSpuriousSplitChain left a split chain (:263-267) and the refined clusters hold at least one match (:486-491).  A chain that
// does not ends the loop over p (p > 0: break) or the read (p == 0: unaligned); one that does adds a SegAlignmentGroup even when
// LocalRefineAlignment then produces no SegAlignment.
__global__ void k_job_reached(uint64_t n_slots, int num_aln, const uint32_t* __restrict__ n_chains, const uint64_t* __restrict__ chain_start,
                              const uint32_t* __restrict__ n_split, const uint32_t* __restrict__ sp_status, const uint64_t* __restrict__ match_off,
                              uint8_t* __restrict__ reached) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;
  bool ok = (uint32_t)(s % (uint64_t)num_aln) < n_chains[s / (uint64_t)num_aln] && !sp_status[s] && n_split[s] > 0;
  if (ok) { const uint64_t cs = chain_start[s]; ok = match_off[cs + n_split[s]] > match_off[cs]; }
  reached[s] = ok ? 1 : 0;
}

// The loop over p ENDS at the first chain that does not reach :574 (p > 0: break, :267 / :491; p == 0: the read is unaligned): the chains behind it are never mapped.
// Their slots are switched off here (every later stage skips a marked slot, as for a deferred read), so the device's result holds no alignment the reference would not
// have made -- lra_map_records* ends a read's loop at the same place either way.
__global__ void k_cut_behind_unreached(int n_reads, int num_aln, uint32_t* __restrict__ sp_status, uint8_t* __restrict__ reached) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_reads) return;
  bool off = false;
  for (int h = 0; h < num_aln; h++) {
    const uint64_t s = (uint64_t)r * num_aln + h;
    if (off) { if (reached[s]) { reached[s] = 0; sp_status[s] |= (uint32_t)LRA_ST_DEFERRED; } }
    else if (!reached[s]) off = true;
  }
}

// per alignment: which read, where its strand's bases start, where its chromosome starts and how long it is
__global__ void k_aln_address(uint64_t n_jobs, int num_aln, const uint64_t* __restrict__ job_aln_off, const int32_t* __restrict__ strand,
                              const int32_t* __restrict__ chrom, const uint64_t* __restrict__ read_off, uint64_t rc_base,
                              const uint64_t* __restrict__ chrom_pos, uint32_t* __restrict__ aln_read, uint64_t* __restrict__ q_off,
                              int32_t* __restrict__ q_len, uint64_t* __restrict__ t_off, int64_t* __restrict__ t_len) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_jobs) return;
  const uint32_t r = (uint32_t)(j / (uint64_t)num_aln);
  for (uint64_t a = job_aln_off[j]; a < job_aln_off[j + 1]; a++) {
    aln_read[a] = r;
    q_off[a] = read_off[r] + (strand[a] ? rc_base : 0);
    q_len[a] = (int32_t)(read_off[r + 1] - read_off[r]);
    const int c = chrom[a];
    t_off[a] = chrom_pos[c];
    t_len[a] = (int64_t)(chrom_pos[c + 1] - chrom_pos[c]);
  }
}

// which (strand, read) sequences of the forward + reverse-complement buffer Refine_splitchain will look up: those a split chain lies on
__global__ void k_mark_strands(uint64_t n_slots, int num_aln, int n_reads, const uint32_t* __restrict__ n_split, const uint64_t* __restrict__ chain_start,
                               const uint8_t* __restrict__ sp_strand, const uint32_t* __restrict__ status, uint8_t* __restrict__ active) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots || status[s]) return;
  const uint32_t r = (uint32_t)(s / (uint64_t)num_aln);
  for (uint32_t k = 0; k < n_split[s]; k++) active[(sp_strand[chain_start[s] + k] ? n_reads : 0) + r] = 1;
}

// ---- RefineBreakpoint between consecutive SegAlignments of a job (Map_lowacc.h:586-596), one round per junction index
__global__ void k_bp_params(int n, const uint32_t* __restrict__ jl, const uint32_t* __restrict__ jr, const uint64_t* __restrict__ boff,
                            const int32_t* __restrict__ strand, const uint64_t* __restrict__ q_off, const int32_t* __restrict__ q_len,
                            const uint64_t* __restrict__ t_off, const int64_t* __restrict__ t_len, uint32_t* l_cnt, uint32_t* r_cnt, int32_t* read_len,
                            int32_t* l_strand, uint64_t* l_read, uint64_t* l_coff, int32_t* l_clen, int32_t* r_strand, uint64_t* r_read, uint64_t* r_coff,
                            int32_t* r_clen) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint32_t a = jl[j], b = jr[j];
  l_cnt[j] = (uint32_t)(boff[a + 1] - boff[a]); r_cnt[j] = (uint32_t)(boff[b + 1] - boff[b]);
  read_len[j] = q_len[a];
  l_strand[j] = strand[a]; l_read[j] = q_off[a]; l_coff[j] = t_off[a]; l_clen[j] = (int32_t)t_len[a];
  r_strand[j] = strand[b]; r_read[j] = q_off[b]; r_coff[j] = t_off[b]; r_clen[j] = (int32_t)t_len[b];
}
__global__ void __launch_bounds__(64) k_bp_gather(int n, const uint32_t* __restrict__ jl, const uint32_t* __restrict__ jr, const uint64_t* __restrict__ boff,
                                                  const int32_t* __restrict__ blocks, const uint64_t* __restrict__ l_off, const uint64_t* __restrict__ r_off,
                                                  int32_t* l_blocks, int32_t* r_blocks) {
  const int j = blockIdx.x >> 1, side = blockIdx.x & 1;
  if (j >= n) return;
  const uint32_t a = side ? jr[j] : jl[j];
  const int32_t* s = blocks + 3 * boff[a];
  int32_t* d = side ? r_blocks + 3 * r_off[j] : l_blocks + 3 * l_off[j];
  const uint64_t w = 3 * (boff[a + 1] - boff[a]);
  for (uint64_t x = threadIdx.x; x < w; x += 64) d[x] = s[x];
}
__global__ void k_bp_counts(uint64_t nA, const uint64_t* __restrict__ boff, uint32_t* cnt, int32_t* touched) {
  const uint64_t a = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= nA) return;
  cnt[a] = (uint32_t)(boff[a + 1] - boff[a]); touched[a] = -1;
}
__global__ void k_bp_touch(int n, const uint32_t* __restrict__ jl, const uint32_t* __restrict__ jr, const int32_t* __restrict__ l_n, const int32_t* __restrict__ r_n,
                           uint32_t* cnt, int32_t* touched) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  cnt[jl[j]] = (uint32_t)l_n[j]; touched[jl[j]] = 2 * j;
  cnt[jr[j]] = (uint32_t)r_n[j]; touched[jr[j]] = 2 * j + 1;
}
__global__ void __launch_bounds__(64) k_bp_scatter(uint64_t nA, const uint64_t* __restrict__ old_off, const int32_t* __restrict__ old_blocks,
                                                   const uint64_t* __restrict__ new_off, const int32_t* __restrict__ touched, const int32_t* __restrict__ l_blocks,
                                                   const uint64_t* __restrict__ l_off, const int32_t* __restrict__ r_blocks, const uint64_t* __restrict__ r_off,
                                                   int32_t* new_blocks) {
  const uint64_t a = blockIdx.x;
  if (a >= nA) return;
  const int32_t tch = touched[a];
  const int32_t* s = tch < 0 ? old_blocks + 3 * old_off[a] : (tch & 1) ? r_blocks + 3 * r_off[tch >> 1] : l_blocks + 3 * l_off[tch >> 1];
  int32_t* d = new_blocks + 3 * new_off[a];
  const uint64_t w = 3 * (new_off[a + 1] - new_off[a]);
  for (uint64_t x = threadIdx.x; x < w; x += 64) d[x] = s[x];
}

// tuple words the local compare stage reads (the algorithmic bytes of local_compare): sum over tasks of both list lengths
__global__ void k_task_words(uint64_t n, const uint64_t* __restrict__ qlo, const uint64_t* __restrict__ qhi, const uint64_t* __restrict__ tlo,
                             const uint64_t* __restrict__ thi, unsigned long long* sum) {
  unsigned long long v = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) v += (qhi[i] - qlo[i]) + (thi[i] - tlo[i]);
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0 && v) atomicAdd(sum, v);
}

}  // namespace

extern "C" void lra_map_opts_preset_ont(lra_map_opts* o) {
  if (!o) return;
  memset(o, 0, sizeof *o);
  // -ONT (lra.cpp:386-431) over the defaults of Options.h:127-230
  o->globalK = 17; o->globalW = 10; o->globalMaxFreq = 150;
  o->localK = 10; o->localW = 5; o->localMaxFreq = 15; o->localIndexWindow = 256;
  o->refineBand = 7; o->localMatch = 4; o->localMismatch = -1; o->localIndel = -2; o->localBand = 15;
  o->refineSpaceDist = 30000; o->anchorstoosparse = 0.005f; o->splitdist = 50000; o->window = 100;
  o->second_anchorbonus = 2.0f; o->bypassClustering = 1; o->skipBandedRefine = 0;
  o->clean.globalK = 17; o->clean.cleanMaxDiag = 200; o->clean.minDiagCluster = 3; o->clean.bypassClustering = 1; o->clean.cleanClustersize = 100;
  o->clean.SecondCleanMinDiagCluster = 10; o->clean.SecondCleanMaxDiag = 100; o->clean.punish_anchorfreq = 5; o->clean.anchorPerlength = 5;
  o->sdp.rate = 20.0f; o->sdp.NumAln = 2; o->sdp.alnthres = 0.65f; o->sdp.gapopen = 7.0f; o->sdp.gapextend = 10.0f; o->sdp.gaproot = 1.5f;
  o->sdp.gapCeiling1 = 1500; o->sdp.gapCeiling2 = 3000; o->sdp.mode = 0; o->sdp.globalK = 17;
  o->readType = LRA_READ_ONT; o->hardClip = 1; o->PrintNumAln = 1; o->printFormat = 's';
  o->flagged_unaligned = 0;    // a flagged read gets an empty record (the caller re-runs it; counters.n_flagged_reads)
  o->defer_matches = 0;        // one pass (lra_map_reads_lowacc_batch: the second, concurrent pass is built and tested, and measured to be no gain on this device)
}

extern "C" void lra_map_opts_preset_clr(lra_map_opts* o) {
  if (!o) return;
  lra_map_opts_preset_ont(o);
  // -CLR (lra.cpp:341-386): what differs from -ONT on this path
  o->globalK = 15; o->globalMaxFreq = 250; o->refineBand = 20; o->second_anchorbonus = 6.0f;
  o->clean.globalK = 15; o->clean.SecondCleanMaxDiag = 120;
  o->sdp.rate = 15.0f; o->sdp.alnthres = 0.50f; o->sdp.globalK = 15;
  o->readType = LRA_READ_CLR;
}

// What glIndex.Read leaves in the options' place (lra.cpp:627, MMIndex.h:154-173): the .gli file's k, w and localIndexWindow are the genome index's AND, through the copy
// constructor (MMIndex.h:128-136, Map_lowacc.h:246-247), the read indexes'; smallOpts.globalK / globalW are glIndex.k / w (Map_lowacc.h:233-234, Map_highacc.h:430-431).
// `lra index` writes k = 10, w = 5, windows of 2048 bases under every preset (LocalIndex(0): 1 << (LOCAL_POS_BITS - 1), MMIndex.h:110-127; RunStoreLocal, lra.cpp:778-850);
// without a .gli file `lra align` builds glIndex from opts.localK / localIndexWindow = 256 (lra.cpp:619-621, :628) -- the presets' values.
extern "C" void lra_map_opts_apply_local_index(lra_map_opts* o, int k, int w, int window) {
  if (!o) return;
  o->localK = k; o->localW = w; o->localIndexWindow = window;
}
extern "C" int lra_ctx_local_index_params(lra_ctx* ctx, int* k, int* w, int* window) {
  if (!ctx || !ctx->map || !ctx->map->gli_window) return LRA_ERR_INVALID;
  if (k) *k = ctx->map->gli_k;
  if (w) *w = ctx->map->gli_w;
  if (window) *window = ctx->map->gli_window;
  return LRA_OK;
}

extern "C" int lra_ctx_load_chromosomes(lra_ctx* ctx, const uint64_t* h_chrom_pos, int n_chrom) {
  if (!ctx || !h_chrom_pos || n_chrom < 1) return LRA_ERR_INVALID;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  lra_map_state* m = map_state(ctx);
  map_disown(m); m->cell->gen++;
  m->chrom_pos.assign(h_chrom_pos, h_chrom_pos + n_chrom + 1);
  if (m->d_chrom_pos) (void)hipFree(m->d_chrom_pos);
  LRA_HIP_CHECK(ctx, hipMalloc((void**)&m->d_chrom_pos, (size_t)(n_chrom + 1) * 8));
  LRA_HIP_CHECK(ctx, hipMemcpy(m->d_chrom_pos, h_chrom_pos, (size_t)(n_chrom + 1) * 8, hipMemcpyHostToDevice));
  if (m->lut.empty()) for (int i = 1; i < 10002; i += 5) m->lut.push_back(logf((float)i));   // LogLookUpTable.h:9-15
  return LRA_OK;
}

extern "C" int lra_ctx_build_local_index(lra_ctx* ctx, int k, int w, int window, int max_freq) {
  if (!ctx || !ctx->map || ctx->map->chrom_pos.size() < 2) return ctx ? lra_set_err(ctx, LRA_ERR_INVALID, "load the chromosome table first") : LRA_ERR_INVALID;
  if (!ctx->seed || !ctx->seed->genome) return lra_set_err(ctx, LRA_ERR_INVALID, "load the genome first");
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  lra_map_state* m = ctx->map;
  if (m->borrowed) return lra_set_err(ctx, LRA_ERR_INVALID, "this context shares another context's reference data: load its own chromosome table first");
  m->cell->gen++;
  const int n_chrom = (int)m->chrom_pos.size() - 1;
  if (m->chrom_pos[n_chrom] != ctx->seed->genome_len) return lra_set_err(ctx, LRA_ERR_INVALID, "chromosome table does not cover the genome");
  lra_local_index_result r;
  int rc = lra_local_index_batch(ctx, n_chrom, (const char*)ctx->seed->genome, m->d_chrom_pos, k, w, window, max_freq, &r);
  if (rc) return rc;
  // the result lives in a context buffer the reads' index will reuse: keep a copy
  if (m->gli_buf) (void)hipFree(m->gli_buf);
  LRA_HIP_CHECK(ctx, hipMalloc(&m->gli_buf, r.bytes + 256));
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(m->gli_buf, r.d_base, r.bytes, hipMemcpyDeviceToDevice, ctx->stream));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  m->gli = r;
  const char* ob = (const char*)r.d_base; char* nb = (char*)m->gli_buf;
  m->gli.d_base = nb;
  m->gli.d_win_off = (const uint64_t*)(nb + ((const char*)r.d_win_off - ob));
  m->gli.d_tuple_bnd = (const uint64_t*)(nb + ((const char*)r.d_tuple_bnd - ob));
  m->gli.d_tuples = (const uint32_t*)(nb + ((const char*)r.d_tuples - ob));
  m->gli_window = window; m->gli_k = k; m->gli_w = w;
  // LocalIndex::seqOffsets (MMIndex.h:200-245): window ends, restarting at each sequence
  std::vector<uint64_t> gso; gso.push_back(0);
  for (int c = 0; c < n_chrom; c++)
    for (uint64_t p = m->chrom_pos[c]; p < m->chrom_pos[c + 1];) { p = std::min<uint64_t>(p + (uint64_t)window, m->chrom_pos[c + 1]); gso.push_back(p); }
  if (gso.size() != r.n_windows + 1) return lra_set_err(ctx, LRA_ERR_INVALID, "local index window count mismatch");
  if (m->d_gso) (void)hipFree(m->d_gso);
  LRA_HIP_CHECK(ctx, hipMalloc((void**)&m->d_gso, gso.size() * 8));
  LRA_HIP_CHECK(ctx, hipMemcpy(m->d_gso, gso.data(), gso.size() * 8, hipMemcpyHostToDevice));
  m->n_gwin = r.n_windows;
  return LRA_OK;
}

// glIndex as LocalIndex::Read left it (MMIndex.h:154-173): the .gli file's payload handed over as it is, instead of building the index again on the device.  The three
// arrays are copied; seq_offsets must be what IndexSeq writes for the loaded chromosome table and this window (MMIndex.h:200-245: window ends, restarting at every
// sequence) -- an index of another genome is refused.
extern "C" int lra_ctx_load_local_index(lra_ctx* ctx, int k, int w, int window, uint64_t n_windows, const uint64_t* h_seq_offsets, const uint64_t* h_tuple_bnd,
                                        uint64_t n_tuples, const uint32_t* h_tuples) {
  if (!ctx || !ctx->map || ctx->map->chrom_pos.size() < 2) return ctx ? lra_set_err(ctx, LRA_ERR_INVALID, "load the chromosome table first") : LRA_ERR_INVALID;
  if (!h_seq_offsets || !h_tuple_bnd || (n_tuples && !h_tuples)) return LRA_ERR_INVALID;
  if (k < 1 || k > 10 || w < 1 || w > 16 || window < w + k || window > 4096) return lra_set_err(ctx, LRA_ERR_INVALID, "need 1<=k<=10 (20-bit LocalTuple), 1<=w<=16, w+k<=window<=4096");
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  lra_map_state* m = ctx->map;
  if (m->borrowed) return lra_set_err(ctx, LRA_ERR_INVALID, "this context shares another context's reference data: load its own chromosome table first");
  const int n_chrom = (int)m->chrom_pos.size() - 1;
  std::vector<uint64_t> gso; gso.push_back(0);
  std::vector<uint64_t> win_off((size_t)n_chrom + 1, 0);
  for (int c = 0; c < n_chrom; c++) {
    for (uint64_t p_ = m->chrom_pos[c]; p_ < m->chrom_pos[c + 1];) { p_ = std::min<uint64_t>(p_ + (uint64_t)window, m->chrom_pos[c + 1]); gso.push_back(p_); }
    win_off[c + 1] = gso.size() - 1;
  }
  if (gso.size() != n_windows + 1 || memcmp(gso.data(), h_seq_offsets, gso.size() * 8) != 0)
    return lra_set_err(ctx, LRA_ERR_INVALID, "the local index's seqOffsets are not those of the loaded chromosome table at windows of %d bases", window);
  if (h_tuple_bnd[0] != 0 || h_tuple_bnd[n_windows] != n_tuples) return lra_set_err(ctx, LRA_ERR_INVALID, "tupleBoundaries do not cover the tuples");
  for (uint64_t i = 0; i < n_windows; i++) if (h_tuple_bnd[i + 1] < h_tuple_bnd[i]) return lra_set_err(ctx, LRA_ERR_INVALID, "tupleBoundaries decrease");
  m->cell->gen++;
  auto sz = [](size_t n, size_t e) { return (n * e + 255) & ~(size_t)255; };
  const size_t NW = (size_t)n_windows + 2;
  const size_t need = sz((size_t)n_chrom + 1, 8) + sz(NW, 8) + sz((size_t)n_tuples + 1, 4);
  if (m->gli_buf) { (void)hipFree(m->gli_buf); m->gli_buf = nullptr; }
  LRA_HIP_CHECK(ctx, hipMalloc(&m->gli_buf, need + 256));
  char* nb = (char*)m->gli_buf;
  uint64_t* o_win = (uint64_t*)nb; uint64_t* o_bnd = (uint64_t*)(nb + sz((size_t)n_chrom + 1, 8)); uint32_t* o_tup = (uint32_t*)((char*)o_bnd + sz(NW, 8));
  LRA_HIP_CHECK(ctx, hipMemcpy(o_win, win_off.data(), ((size_t)n_chrom + 1) * 8, hipMemcpyHostToDevice));
  LRA_HIP_CHECK(ctx, hipMemcpy(o_bnd, h_tuple_bnd, ((size_t)n_windows + 1) * 8, hipMemcpyHostToDevice));
  if (n_tuples) LRA_HIP_CHECK(ctx, hipMemcpy(o_tup, h_tuples, (size_t)n_tuples * 4, hipMemcpyHostToDevice));
  m->gli = lra_local_index_result{};
  m->gli.n_seqs = n_chrom; m->gli.n_windows = n_windows; m->gli.n_tuples = n_tuples; m->gli.bytes = need;
  m->gli.d_base = nb; m->gli.d_win_off = o_win; m->gli.d_tuple_bnd = o_bnd; m->gli.d_tuples = o_tup;
  m->gli_window = window; m->gli_k = k; m->gli_w = w;
  if (m->d_gso) (void)hipFree(m->d_gso);
  LRA_HIP_CHECK(ctx, hipMalloc((void**)&m->d_gso, gso.size() * 8));
  LRA_HIP_CHECK(ctx, hipMemcpy(m->d_gso, gso.data(), gso.size() * 8, hipMemcpyHostToDevice));
  m->n_gwin = n_windows;
  return LRA_OK;
}

// Several contexts on one GPU (sub-batches on their own HIP streams, so that the serial tails of one sub-batch's kernels overlap the other's work)
// share ONE replica of the reference: dst borrows src's genome, global index + directory, chromosome table and local index.  src must outlive dst.
int lra_seed_share(lra_ctx* dst, lra_ctx* src);   // seed.hip
extern "C" int lra_ctx_share_reference(lra_ctx* dst, lra_ctx* src) {
  if (!dst || !src || dst == src || !src->map || !src->seed || dst->device != src->device) return LRA_ERR_INVALID;
  if (dst->map) return lra_set_err(dst, LRA_ERR_INVALID, "context already holds reference data");
  int rc = lra_seed_share(dst, src);
  if (rc) return rc;
  lra_map_state* m = map_state(dst);
  const lra_map_state* s = src->map;
  m->chrom_pos = s->chrom_pos; m->d_chrom_pos = s->d_chrom_pos; m->gli_buf = s->gli_buf; m->gli = s->gli; m->d_gso = s->d_gso; m->n_gwin = s->n_gwin;
  m->gli_window = s->gli_window; m->gli_k = s->gli_k; m->gli_w = s->gli_w; m->lut = s->lut; m->borrowed = true;
  m->owner_cell = s->borrowed ? s->owner_cell : s->cell; m->owner_generation = s->borrowed ? s->owner_generation : s->cell->gen.load();
  return LRA_OK;
}

// LRA_OK, or LRA_ERR_INVALID when this context borrows reference data (lra_ctx_share_reference) that its owner has replaced since
int lra_seed_check_shared(lra_ctx* ctx);   // seed.hip
int lra_map_check_shared(lra_ctx* ctx) {
  int rc = lra_seed_check_shared(ctx);
  if (rc) return rc;
  const lra_map_state* m = ctx->map;
  if (m && m->borrowed && m->owner_cell && (m->owner_cell->dead.load() || m->owner_cell->gen.load() != m->owner_generation))
    return lra_set_err(ctx, LRA_ERR_INVALID, "the context this one shares its reference data with has %s: call lra_ctx_share_reference again",
                       m->owner_cell->dead.load() ? "been destroyed" : "reloaded it");
  return LRA_OK;
}

// the context's reference data, for callers that want to write it to files (lra_write_gli) or hand it to another consumer
extern "C" const char* lra_ctx_genome_ptr(lra_ctx* ctx) { return (ctx && ctx->seed) ? (const char*)ctx->seed->genome : nullptr; }
extern "C" int lra_ctx_local_index(lra_ctx* ctx, lra_local_index_result* out, const uint64_t** d_seq_offsets) {
  if (!ctx || !ctx->map || !ctx->map->gli_buf || !out) return LRA_ERR_INVALID;
  *out = ctx->map->gli;
  if (d_seq_offsets) *d_seq_offsets = ctx->map->d_gso;
  return LRA_OK;
}

// RefineBreakpoint(read, genome, *SegAlignment[s], *SegAlignment[s-1], opts) for s = 1, 2, ... of every job: round k runs junction k of all
// jobs that have one (segment k is "left", segment k - 1 -- already refined against k - 2 in the round before -- is "right").
int lra_refine_breakpoints(lra_ctx* ctx, uint64_t nJ, uint64_t nA, const uint64_t* d_job_aln_off, const int32_t* d_strand, const uint64_t* q_off, const int32_t* q_len,
                              const uint64_t* t_off, const int64_t* t_len, const char* strands, const char* genome, lra_refine_result* fres) {
  hipStream_t st = ctx->stream;
  std::vector<uint64_t> jo(nJ + 1);
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(jo.data(), d_job_aln_off, (nJ + 1) * 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  uint64_t max_seg = 0;
  for (uint64_t j = 0; j < nJ; j++) max_seg = std::max(max_seg, jo[j + 1] - jo[j]);
  const int32_t* cur_blocks = fres->d_blocks; const uint64_t* cur_off = fres->d_block_off;
  uint64_t n_blocks = fres->n_blocks;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  for (uint64_t k = 1; k < max_seg; k++) {
    std::vector<uint32_t> hl, hr;
    for (uint64_t j = 0; j < nJ; j++) if (jo[j + 1] - jo[j] > k) { hl.push_back((uint32_t)(jo[j] + k)); hr.push_back((uint32_t)(jo[j] + k - 1)); }
    const int n = (int)hl.size();
    if (!n) break;
    const size_t n1 = (size_t)n + 2;
    char* w = (char*)lra_ensure(ctx, 72, al(n1 * 4) * 9 + al(n1 * 8) * 6 + al((nA + 2) * 4) * 2 + al((nA + 2) * 8) + 4096);
    if (!w) return LRA_ERR_NOMEM;
    auto take = [&](size_t bytes) { char* r = w; w += al(bytes); return r; };
    uint32_t* jl = (uint32_t*)take(n1 * 4); uint32_t* jr = (uint32_t*)take(n1 * 4); uint32_t* l_cnt = (uint32_t*)take(n1 * 4); uint32_t* r_cnt = (uint32_t*)take(n1 * 4);
    int32_t* read_len = (int32_t*)take(n1 * 4); int32_t* l_strand = (int32_t*)take(n1 * 4); int32_t* r_strand = (int32_t*)take(n1 * 4);
    int32_t* l_clen = (int32_t*)take(n1 * 4); int32_t* r_clen = (int32_t*)take(n1 * 4);
    uint64_t* l_read = (uint64_t*)take(n1 * 8); uint64_t* r_read = (uint64_t*)take(n1 * 8); uint64_t* l_coff = (uint64_t*)take(n1 * 8); uint64_t* r_coff = (uint64_t*)take(n1 * 8);
    uint64_t* l_off = (uint64_t*)take(n1 * 8); uint64_t* r_off = (uint64_t*)take(n1 * 8);
    uint32_t* cnt = (uint32_t*)take((nA + 2) * 4); int32_t* touched = (int32_t*)take((nA + 2) * 4); uint64_t* new_off_tmp = (uint64_t*)take((nA + 2) * 8);
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(jl, hl.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(jr, hr.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_bp_params, dim3((n + 255) / 256), dim3(256), 0, st, n, jl, jr, cur_off, d_strand, q_off, q_len, t_off, t_len, l_cnt, r_cnt, read_len, l_strand,
                       l_read, l_coff, l_clen, r_strand, r_read, r_coff, r_clen);
    int rc;
    if ((rc = lra_exclusive_scan<uint32_t>(ctx, n, l_cnt, l_off)) || (rc = lra_exclusive_scan<uint32_t>(ctx, n, r_cnt, r_off))) return rc;
    uint64_t tl = 0, tr = 0;
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(&tl, l_off + n, 8, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(&tr, r_off + n, 8, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
    int32_t* lb = (int32_t*)lra_ensure(ctx, 73, al((tl + 1) * 12) + al((tr + 1) * 12) + 512);
    if (!lb) return LRA_ERR_NOMEM;
    int32_t* rb = (int32_t*)((char*)lb + al((tl + 1) * 12));
    hipLaunchKernelGGL(k_bp_gather, dim3(2 * n), dim3(64), 0, st, n, jl, jr, cur_off, cur_blocks, l_off, r_off, lb, rb);
    lra_breakpoint_result br;
    if ((rc = lra_refine_breakpoint_batch(ctx, n, read_len, strands, genome, lb, l_off, l_strand, l_read, l_coff, l_clen, rb, r_off, r_strand, r_read, r_coff, r_clen, &br)))
      return rc;
    hipLaunchKernelGGL(k_bp_counts, dim3((unsigned)((nA + 255) / 256)), dim3(256), 0, st, nA, cur_off, cnt, touched);
    hipLaunchKernelGGL(k_bp_touch, dim3((n + 255) / 256), dim3(256), 0, st, n, jl, jr, br.d_l_n, br.d_r_n, cnt, touched);
    if ((rc = lra_exclusive_scan<uint32_t>(ctx, (long)nA, cnt, new_off_tmp))) return rc;
    uint64_t nb = 0;
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(&nb, new_off_tmp + nA, 8, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
    const int slot = 74 + (int)(k & 1);                                   // ping-pong: the other slot may hold the current blocks
    char* nbuf = (char*)lra_ensure(ctx, slot, al((nb + 1) * 12) + al((nA + 2) * 8) + 512);
    if (!nbuf) return LRA_ERR_NOMEM;
    int32_t* new_blocks = (int32_t*)nbuf; uint64_t* new_off = (uint64_t*)(nbuf + al((nb + 1) * 12));
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(new_off, new_off_tmp, (nA + 1) * 8, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_bp_scatter, dim3((unsigned)nA), dim3(64), 0, st, nA, cur_off, cur_blocks, new_off, touched, br.d_l_blocks, br.d_l_off, br.d_r_blocks, br.d_r_off,
                       new_blocks);
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
    cur_blocks = new_blocks; cur_off = new_off; n_blocks = nb;
  }
  fres->d_blocks = cur_blocks; fres->d_block_off = cur_off; fres->n_blocks = n_blocks;
  return LRA_OK;
}

extern "C" int lra_match_rate_batch(lra_ctx* ctx, const lra_cluster_result* clusters, float initial_anchorbonus, const float** d_rate) {
  if (!ctx || !clusters || !d_rate) return LRA_ERR_INVALID;
  *d_rate = nullptr;
  const int n_reads = clusters->n_reads;
  float* rate = (float*)lra_ensure(ctx, 80, ((size_t)n_reads + 1) * 4);
  if (!rate) return LRA_ERR_NOMEM;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (n_reads) hipLaunchKernelGGL(k_match_rate, dim3(n_reads), dim3(64), 0, ctx->stream, n_reads, clusters->d_cluster_off, clusters->d_c_start, clusters->d_c_end,
                                  clusters->d_c_anchorfreq, initial_anchorbonus, rate);
  *d_rate = rate;
  return LRA_OK;
}

// One pass of MapRead_lowacc over a batch.  defer_threshold > 0: the reads with more refined matches than that after Refine_Btwnsplitchain leave the pass there (no
// alignments, LRA_ST_DEFERRED in their status word); *deferred lists them and on_deferred runs as soon as the list is known.
// What the stages behind the split point need from the ones in front of it (device pointers into the first pass's buffers, which it leaves alone from there on)
struct LowaccTailIn {
  int n_reads = 0, num_aln = 1; uint64_t n_slots = 0, tot = 0;
  const uint64_t* d_read_off = nullptr; const char* d_seq = nullptr; const char* both = nullptr;
  const uint32_t* slot_n0 = nullptr;
  lra_merge_result mres;
  uint8_t* job_reached = nullptr; uint32_t* read_status = nullptr;
  lra_map_counters counters;
};
static int lowacc_tail(lra_ctx* ctx, const LowaccTailIn& in, const lra_map_opts* o, lra_map_result* out);

// A batch between its two halves (lra_map_reads_lowacc_front / _back): what the tail needs, and a queue of ONE batch between the threads of the two halves.
// The front half writes the batch it hands over (MergeChain .. TrimOverlappedAnchors' results, the reads with their reverse complements, the chains' NumOfAnchors0, the
// slots reached, the status words) into one of two sets of buffers -- the handover contexts hand[0 / 1], taken in turn -- so it never waits for the back half that is
// RUNNING, only for the batch before its own to have been taken: when batch i - 1 has been taken, batch i - 2 has been released, and set i % 2 is free.
struct lra_handover {
  std::mutex mu; std::condition_variable cv;
  bool pending = false;    // a batch is handed over, its back half not yet started
  bool busy = false;       // the back half runs, or its result is still in use (until lra_map_back_release)
  uint64_t seq = 0;        // batches handed over so far (error batches do not count: they use no buffers)
  lra_ctx* hand[2] = {nullptr, nullptr};   // owned through the companion's child chain (destroyed and timed with it)
  LowaccTailIn in;
  int rc = LRA_OK;         // pending only: the front half of this batch FAILED with this code (nothing to run: the back call returns it)
  std::string err;
};
void lra_handover_free(lra_ctx* ctx) { delete ctx->handover; ctx->handover = nullptr; }
bool lra_handover_idle(lra_ctx* ctx) {                                     // lra_ctx_release_buffers: nothing handed over and not taken, no back half running or unreleased
  lra_handover* H = ctx->handover;
  if (!H) return true;
  std::lock_guard<std::mutex> lk(H->mu);
  return !H->pending && !H->busy;
}
static lra_handover* handover_of(lra_ctx* ctx) {                         // (the two halves' threads may both be the first to ask)
  static std::mutex make;
  std::lock_guard<std::mutex> lk(make);
  if (!ctx->handover) ctx->handover = new lra_handover();
  return ctx->handover;
}

static int child_refresh(lra_ctx* ctx);
static int lowacc_core(lra_ctx* ctx, int n_reads, const char* d_seq, const uint64_t* d_read_off, uint64_t total_bases, const lra_map_opts* o, lra_map_result* out,
                       uint32_t defer_threshold, std::vector<uint32_t>* deferred, lra_ctx* second, LowaccTailIn* second_in, const std::function<int()>& on_deferred,
                       lra_handover* H = nullptr) {
  memset(out, 0, sizeof *out);
  lra_map_state* m = ctx->map;
  out->n_reads = n_reads;
  if (n_reads == 0) return LRA_OK;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  auto grid = [](uint64_t n) { return dim3((unsigned)((n + 255) / 256)); };
  uint32_t* read_status = (uint32_t*)lra_ensure(ctx, 81, ((size_t)n_reads + 1) * 4);
  if (!read_status) return LRA_ERR_NOMEM;
  LRA_HIP_CHECK(ctx, hipMemsetAsync(read_status, 0, (size_t)n_reads * 4, st));
  const uint64_t* CH = m->chrom_pos.data();
  const int nCh = (int)m->chrom_pos.size() - 1;
  const char* genome = (const char*)ctx->seed->genome;
  const uint64_t tot = total_bases;
  int rc;
  // LRA_STAGE_DBG=1: wall time of every stage call (device work + the host-side sizing round trips around it)
  const bool sdbg = getenv("LRA_STAGE_DBG") != nullptr;
  auto wall = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_prev = 0;
  if (sdbg) { (void)hipStreamSynchronize(st); t_prev = wall(); }
  auto stage = [&](const char* name) { if (!sdbg) return; (void)hipStreamSynchronize(st); const double t = wall(); fprintf(stderr, "[stage%s] %-28s %8.1f ms\n", ctx->owns_stream ? " 2nd" : "", name, t - t_prev); t_prev = t; };
  // a1-a4
  lra_seed_result sres;
  // opts.defer_seed_matches: the reads with more tier-1 matches than that are handed back (the seed stage empties their match lists: no later stage sees them)
  const uint32_t seedT = o->defer_seed_matches > 0 ? (uint32_t)o->defer_seed_matches : 0;
  // (a result made ahead of the call -- lra_seed_prefetch on a side context, lra_ctx_adopt_seed -- from these reads with these parameters is what lra_seed_batch would make)
  const bool ahead = ctx->ahead.valid && ctx->ahead.n_reads == n_reads && ctx->ahead.d_seq == d_seq && ctx->ahead.d_read_off == d_read_off &&
                     ctx->ahead.k == o->globalK && ctx->ahead.w == o->globalW && ctx->ahead.max_freq == o->globalMaxFreq;
  ctx->ahead.valid = false;
  if (ahead) {
    if (seedT) return lra_set_err(ctx, LRA_ERR_INVALID, "defer_seed_matches and a seed result adopted ahead of the call do not combine");
    sres = ctx->ahead.res;
  } else {
    ctx->seed->defer_T = seedT;
    rc = lra_seed_batch(ctx, n_reads, d_seq, d_read_off, o->globalK, o->globalW, o->globalMaxFreq, &sres);
    ctx->seed->defer_T = 0;
    if (rc) return rc;
  }
  uint64_t n_handed_back = 0;
  if (seedT) {
    unsigned long long* dcnt = (unsigned long long*)lra_ensure(ctx, 191, 64);
    if (!dcnt) return LRA_ERR_NOMEM;
    LRA_HIP_CHECK(ctx, hipMemsetAsync(dcnt, 0, 8, st));
    hipLaunchKernelGGL(k_mark_handed_back, grid((uint64_t)n_reads), dim3(256), 0, st, n_reads, (const uint8_t*)ctx->seed->defer_flag, read_status, dcnt);
    unsigned long long h = 0;
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(&h, dcnt, 8, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
    n_handed_back = h;
  }
  stage("seed");
  // a5, a7
  lra_cluster_result cres;
  if ((rc = lra_clean_matches_batch(ctx, &o->clean, CH, nCh, &cres))) return rc;
  stage("clean");
  lra_extend_result eres;
  if ((rc = lra_linear_extend_batch(ctx, o->globalK, d_seq, d_read_off, &eres))) return rc;
  stage("linear_extend");
  // a8: the primary chains (Map_lowacc.h:184-188); match_rate = 3 for reads with a repetitive cluster (:86-89)
  const float* match_rate = nullptr;
  if ((rc = lra_match_rate_batch(ctx, &cres, o->sdp.rate, &match_rate))) return rc;
  lra_chain_result chres;
  if ((rc = lra_sparse_dp_batch(ctx, n_reads, cres.d_cluster_off, eres.d_e_start, eres.d_e_count, cres.d_c_strand, eres.d_e_qpos, eres.d_e_tpos, eres.d_e_len,
                                d_read_off, match_rate, &o->sdp, &chres))) return rc;
  stage("sdp#A");
  const int num_aln = chres.num_aln;
  const uint64_t n_slots = (uint64_t)n_reads * (uint64_t)num_aln;
  hipLaunchKernelGGL(k_or_status_div, grid(n_reads), dim3(256), 0, st, (uint64_t)n_reads, chres.d_status, 1, read_status);
  // chains[p].NumOfAnchors0 (the second sparse DP reuses the first one's buffers)
  uint32_t* slot_n0 = (uint32_t*)lra_ensure(ctx, 56, (n_slots + 1) * 4);
  if (!slot_n0) return LRA_ERR_NOMEM;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(slot_n0, chres.d_chain_len, n_slots * 4, hipMemcpyDeviceToDevice, st));
  // a9
  lra_split_result spres;
  if ((rc = lra_split_chains_batch(ctx, &chres, CH, nCh, o->splitdist, o->bypassClustering, &spres))) return rc;
  stage("split_chains");
  hipLaunchKernelGGL(k_or_status_div, grid(n_slots), dim3(256), 0, st, n_slots, spres.d_status, num_aln, read_status);
  // a10: the reads forward, then reverse complemented, in one buffer + its local index (Map_lowacc.h:246-250)
  char* both = (char*)lra_ensure(ctx, 57, 2 * tot + 64);
  uint64_t* off2 = (uint64_t*)lra_ensure(ctx, 58, (2 * (size_t)n_reads + 2) * 8);
  if (!both || !off2) return LRA_ERR_NOMEM;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(both, d_seq, tot, hipMemcpyDeviceToDevice, st));
  LRA_HIP_CHECK(ctx, hipMemsetAsync(both + 2 * tot, 0, 64, st));
  if ((rc = lra_create_rc_batch(ctx, n_reads, d_seq, d_read_off, both + tot))) return rc;
  hipLaunchKernelGGL(k_add_off, dim3((n_reads + 256) / 256), dim3(256), 0, st, n_reads, d_read_off, tot, off2);
  // the reference indexes both strands of every read; only the strands with a split chain are ever looked up, so only those get tuples
  uint8_t* active = (uint8_t*)lra_ensure(ctx, 65, 2 * (size_t)n_reads + 64);
  if (!active) return LRA_ERR_NOMEM;
  LRA_HIP_CHECK(ctx, hipMemsetAsync(active, 0, 2 * (size_t)n_reads, st));
  hipLaunchKernelGGL(k_mark_strands, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, st, n_slots, num_aln, n_reads, spres.d_n_split, chres.d_chain_start,
                     spres.d_sp_strand, spres.d_status, active);
  lra_local_index_result rli;
  if ((rc = lra_local_index_masked_batch(ctx, 2 * n_reads, both, off2, active, o->localK, o->localW, o->localIndexWindow, o->localMaxFreq, &rli))) return rc;
  stage("read local index");
  lra_rsc_opts ro; ro.window = o->window; ro.smallK = o->localK; ro.K = o->globalK; ro.limitrefine = 1; ro.max_freq = o->localMaxFreq; ro.local_window = o->localIndexWindow;
  lra_refined_result rres;
  if ((rc = lra_refine_splitchain_batch(ctx, &chres, &spres, d_read_off, CH, nCh, &rli, m->n_gwin, m->d_gso, m->gli.d_tuple_bnd, m->gli.d_tuples, &ro, &rres))) return rc;
  stage("refine_splitchain");
  uint64_t task_words = 0;
  if (rres.n_tasks) {
    unsigned long long* d_sum = (unsigned long long*)lra_scratch(ctx, 3, 256);
    if (!d_sum) return LRA_ERR_NOMEM;
    LRA_HIP_CHECK(ctx, hipMemsetAsync(d_sum, 0, 8, st));
    hipLaunchKernelGGL(k_task_words, dim3((unsigned)std::min<uint64_t>((rres.n_tasks + 255) / 256, 1024)), dim3(256), 0, st, rres.n_tasks, rres.d_task_q_lo,
                       rres.d_task_q_hi, rres.d_task_t_lo, rres.d_task_t_hi, d_sum);
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(&task_words, d_sum, 8, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  }
  // a11 callers
  lra_btwn_opts bo; bo.K = o->localK; bo.W = o->localW; bo.refineSpaceDist = o->refineSpaceDist; bo.anchorstoosparse = o->anchorstoosparse;
  bo.match = o->localMatch; bo.mismatch = o->localMismatch; bo.indel = o->localIndel; bo.max_freq = o->localMaxFreq;
  lra_btwn_result bres;
  if ((rc = lra_refine_btwn_splitchain_batch(ctx, &chres, &spres, &rres, d_read_off, both, tot, genome, CH, nCh, &bo, &bres))) return rc;
  stage("refine_btwn_splitchain");
  uint8_t* job_reached = (uint8_t*)lra_ensure(ctx, 82, n_slots + 64);
  if (!job_reached) return LRA_ERR_NOMEM;
  hipLaunchKernelGGL(k_job_reached, grid(n_slots), dim3(256), 0, st, n_slots, num_aln, chres.d_n_chains, chres.d_chain_start, spres.d_n_split, spres.d_status,
                     bres.d_match_off, job_reached);
  if (rres.n_frags) hipLaunchKernelGGL(k_or_status_split, grid(n_slots), dim3(256), 0, st, n_slots, num_aln, chres.d_n_chains, chres.d_chain_start, spres.d_n_split,
                                       spres.d_status, rres.d_status, read_status);
  hipLaunchKernelGGL(k_cut_behind_unreached, grid((uint64_t)n_reads), dim3(256), 0, st, n_reads, num_aln, (uint32_t*)spres.d_status, job_reached);
  // counters of the stages so far
  lra_map_counters cnt0; memset(&cnt0, 0, sizeof cnt0);
  cnt0.n_minimizers = sres.n_minimizers; cnt0.n_matches = sres.n_matches; cnt0.n_clusters = cres.n_clusters; cnt0.n_sdp_anchors = chres.n_frags; cnt0.n_sdp_points = chres.n_points;
  cnt0.n_sdp_entries = chres.n_subproblem_entries; cnt0.n_local_tuples = rli.n_tuples; cnt0.n_local_tasks = rres.n_tasks; cnt0.n_local_task_words = task_words; cnt0.n_local_pairs = rres.n_pairs;
  cnt0.n_handed_back_reads = n_handed_back; cnt0.n_refined_matches = rres.n_matches; cnt0.n_btwn_problems = bres.n_problems; cnt0.n_btwn_rounds = bres.n_rounds; cnt0.n_refined_after_btwn = bres.n_matches;
  LowaccTailIn in;
  in.n_reads = n_reads; in.num_aln = num_aln; in.n_slots = n_slots; in.tot = tot; in.d_read_off = d_read_off; in.d_seq = d_seq; in.both = both; in.slot_n0 = slot_n0;
  in.job_reached = job_reached; in.read_status = read_status; in.counters = cnt0;
  if (const char* dumpPath = getenv("LRA_LOAD_DUMP")) {                  // analysis: per read, the tier-1 matches and the refined matches its second sparse DP will chain
    uint32_t* load = (uint32_t*)lra_ensure(ctx, 181, ((size_t)n_reads + 1) * 4);
    if (!load) return LRA_ERR_NOMEM;
    LRA_HIP_CHECK(ctx, hipMemsetAsync(load, 0, (size_t)n_reads * 4, st));
    hipLaunchKernelGGL(k_read_load, grid(n_slots), dim3(256), 0, st, n_slots, num_aln, chres.d_n_chains, chres.d_chain_start, spres.d_n_split, spres.d_status, bres.d_match_off, load);
    std::vector<uint32_t> hl((size_t)n_reads), hs((size_t)n_slots); std::vector<uint64_t> hm((size_t)n_reads + 1), hq((size_t)n_reads + 1);
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(hl.data(), load, (size_t)n_reads * 4, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(hs.data(), spres.d_n_split, (size_t)n_slots * 4, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(hm.data(), sres.d_match_off, ((size_t)n_reads + 1) * 8, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(hq.data(), sres.d_mm_off, ((size_t)n_reads + 1) * 8, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (FILE* f = fopen(dumpPath, "wb")) {
      const uint32_t hdr[2] = {(uint32_t)n_reads, (uint32_t)num_aln};
      fwrite(hdr, 4, 2, f); fwrite(hm.data(), 8, hm.size(), f); fwrite(hq.data(), 8, hq.size(), f); fwrite(hl.data(), 4, hl.size(), f); fwrite(hs.data(), 4, hs.size(), f);
      fclose(f);
    }
  }
  // ---- the split point: reads with more refined matches than the threshold go on in the second context (from here: MergeChain onwards), beside this pass
  if (defer_threshold && deferred && second && second_in) {
    uint32_t* load = (uint32_t*)lra_ensure(ctx, 181, ((size_t)n_reads + 1) * 4);
    uint8_t* dflag = (uint8_t*)lra_ensure(ctx, 182, (size_t)n_reads + 64);
    if (!load || !dflag) return LRA_ERR_NOMEM;
    LRA_HIP_CHECK(ctx, hipMemsetAsync(load, 0, (size_t)n_reads * 4, st));
    hipLaunchKernelGGL(k_read_load, grid(n_slots), dim3(256), 0, st, n_slots, num_aln, chres.d_n_chains, chres.d_chain_start, spres.d_n_split, spres.d_status, bres.d_match_off, load);
    uint32_t* sp2 = (uint32_t*)lra_ensure(second, 183, (n_slots + 1) * 4);    // the second pass's view of the split chains: everything but the deferred reads' slots is off
    uint8_t* reached2 = (uint8_t*)lra_ensure(second, 82, n_slots + 64);
    uint32_t* rstat2 = (uint32_t*)lra_ensure(second, 81, ((size_t)n_reads + 1) * 4);
    if (!sp2 || !reached2 || !rstat2) return LRA_ERR_NOMEM;
    hipLaunchKernelGGL(k_mark_deferred, grid(n_reads), dim3(256), 0, st, n_reads, num_aln, (const uint32_t*)load, defer_threshold, (uint32_t*)spres.d_status, read_status, dflag,
                       sp2, job_reached, reached2, rstat2);
    std::vector<uint8_t> hf((size_t)n_reads);
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(hf.data(), dflag, (size_t)n_reads, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
    deferred->clear();
    for (int r = 0; r < n_reads; r++) if (hf[r]) deferred->push_back((uint32_t)r);
    if (!deferred->empty()) {
      // MergeChain .. TrimOverlappedAnchors of the deferred reads' chains, into the second context's buffers but queued on THIS stream: it reads the first sparse DP's
      // arrays, which this pass's second sparse DP is about to reuse
      lra_split_result spB = spres; spB.d_status = sp2;
      const hipStream_t keep = second->stream;
      second->stream = st;
      rc = lra_merge_extend_batch(second, &chres, &spB, &bres, d_seq, d_read_off, genome, CH, nCh, o->localK, &second_in->mres);
      second->stream = keep;
      if (rc) return lra_set_err(ctx, rc, "second pass, MergeChain: %s", second->err.c_str());
      LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
      { const lra_merge_result keepM = second_in->mres; *second_in = in; second_in->mres = keepM; second_in->job_reached = reached2; second_in->read_status = rstat2; }
      if (on_deferred && (rc = on_deferred())) return rc;
    }
    stage("deferred reads");
  }
  if (H) {
    // The front half ends here (lra_map_reads_lowacc_front): MergeChain .. TrimOverlappedAnchors write into the handover buffers of this batch's turn (queued on this
    // stream: they read the first sparse DP's and the refinement's arrays), and the four buffers of this half that the tail reads -- the reads with their reverse
    // complements, the chains' NumOfAnchors0, the slots reached, the reads' status words -- change owner with that set's (no copy).  Not before the batch before this one
    // has been TAKEN by a back call (then the set's last user, the batch before that, has been released); the back half that is running is not waited for.
    const double tw0 = wall();
    { std::unique_lock<std::mutex> lk(H->mu); H->cv.wait(lk, [&] { return !H->pending; }); }
    if (getenv("LRA_TWO_STAGE_DBG")) fprintf(stderr, "[two-stage] front half waited %.0f ms for the batch before it to be taken\n", wall() - tw0);
    lra_ctx* hs = H->hand[H->seq & 1];
    for (int slot : {56, 57, 81, 82}) { std::swap(ctx->gbuf[slot], hs->gbuf[slot]); std::swap(ctx->gbytes[slot], hs->gbytes[slot]); }
    hs->stream = st;                                                       // (a handover context has no stream of its own: its one stage runs on the front half's)
    // (the stage's work arrays -- per refined match, dead when it returns -- and its sort's scratch are this context's, lent for the call: one set, not one per handover context)
    auto lend = [&]() { std::swap(ctx->gbuf[100], hs->gbuf[100]); std::swap(ctx->gbytes[100], hs->gbytes[100]); std::swap(ctx->scratch[2], hs->scratch[2]); std::swap(ctx->scratch_bytes[2], hs->scratch_bytes[2]); };
    lend();
    rc = lra_merge_extend_batch(hs, &chres, &spres, &bres, d_seq, d_read_off, genome, CH, nCh, o->localK, &in.mres);
    lend();
    if (rc) return lra_set_err(ctx, rc, "MergeChain into the handover buffers: %s", hs->err.c_str());
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
    stage("merge_extend");
    { std::lock_guard<std::mutex> lk(H->mu); H->in = in; H->rc = LRA_OK; H->pending = true; H->seq++; }
    H->cv.notify_all();
    return LRA_OK;
  }
  // a9 MergeChain, a7 second pass (Map_lowacc.h:411-476)
  if ((rc = lra_merge_extend_batch(ctx, &chres, &spres, &bres, d_seq, d_read_off, genome, CH, nCh, o->localK, &in.mres))) return rc;
  stage("merge_extend");
  return lowacc_tail(ctx, in, o, out);
}

// a8 second sparse DP, a13, a14, a16 (Map_lowacc.h:477-599) on the merged clusters of `in`
static int lowacc_tail(lra_ctx* ctx, const LowaccTailIn& in, const lra_map_opts* o, lra_map_result* out) {
  lra_map_state* m = ctx->map;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  auto grid = [](uint64_t n) { return dim3((unsigned)((n + 255) / 256)); };
  const int n_reads = in.n_reads, num_aln = in.num_aln; const uint64_t n_slots = in.n_slots, tot = in.tot;
  const uint64_t* d_read_off = in.d_read_off; const char* both = in.both; const uint32_t* slot_n0 = in.slot_n0;
  const lra_merge_result& mres = in.mres;
  uint8_t* job_reached = in.job_reached; uint32_t* read_status = in.read_status;
  const uint64_t* CH = m->chrom_pos.data();
  const int nCh = (int)m->chrom_pos.size() - 1;
  const char* genome = (const char*)ctx->seed->genome;
  int rc;
  const bool sdbg = getenv("LRA_STAGE_DBG") != nullptr;
  auto wall = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_prev = 0;
  if (sdbg) { (void)hipStreamSynchronize(st); t_prev = wall(); }
  auto stage = [&](const char* name) { if (!sdbg) return; (void)hipStreamSynchronize(st); const double t = wall(); fprintf(stderr, "[stage%s] %-28s %8.1f ms\n", ctx->owns_stream ? " 2nd" : "", name, t - t_prev); t_prev = t; };
  (void)n_slots;
  out->n_reads = n_reads;
  lra_sdp_opts s2 = o->sdp; s2.mode = 1; s2.rate = o->second_anchorbonus;      // SparseDP :2287 with opts.second_anchorbonus (Options.h:221)
  lra_chain_result ch2;
  if ((rc = lra_sparse_dp_batch(ctx, (int)mres.n_groups, mres.d_iota, mres.d_anchor_off, mres.d_count, mres.d_strand, mres.d_q, mres.d_t, mres.d_len, mres.d_iota,
                                nullptr, &s2, &ch2))) return rc;
  stage("sdp#2");
  if (mres.n_groups) hipLaunchKernelGGL(k_or_status_idx, grid(mres.n_groups), dim3(256), 0, st, mres.n_groups, ch2.d_status, mres.d_group_slot, num_aln, read_status);
  // a13
  lra_local_refine_inputs inp;
  if ((rc = lra_local_refine_inputs_batch(ctx, num_aln, slot_n0, &mres, &ch2, &inp))) return rc;
  stage("local_refine_inputs");
  lra_lra_opts lo; lo.localW = o->localW; lo.globalW = o->localW; lo.localMaxFreq = o->localMaxFreq; lo.match = o->localMatch; lo.mismatch = o->localMismatch;
  lo.indel = o->localIndel; lo.localBand = o->localBand; lo.refineBySDP = 1; lo.isOnt = (o->readType == LRA_READ_ONT || o->readType == LRA_READ_CLR) ? 1 : 0;
  lo.gapopen = o->sdp.gapopen; lo.gapextend = o->sdp.gapextend; lo.gaproot = o->sdp.gaproot; lo.gapCeiling1 = o->sdp.gapCeiling1; lo.gapCeiling2 = o->sdp.gapCeiling2;
  lra_alignments_result ares;
  if ((rc = lra_local_refine_batch(ctx, inp.n_jobs, inp.d_job_chain_off, inp.d_job_read, inp.d_job_h, inp.n_chains, inp.d_chain_anchor_off, inp.d_chain_strand,
                                   inp.d_chain_chrom, inp.d_chain_value, inp.d_chain_n0, inp.d_chain_n1, inp.n_anchors, inp.d_q, inp.d_t, inp.d_len, d_read_off,
                                   both, tot, genome, CH, nCh, &lo, &ares))) return rc;
  stage("local_refine");
  const uint64_t nA = ares.n_alignments, nJ = ares.n_jobs;
  if (nJ) hipLaunchKernelGGL(k_or_status_div, grid(nJ), dim3(256), 0, st, nJ, ares.d_status, num_aln, read_status);
  // a14, a16 on every SegAlignment (Map_lowacc.h:582-599)
  uint32_t* aln_read = (uint32_t*)lra_ensure(ctx, 59, (nA + 1) * 4);
  uint64_t* q_off = (uint64_t*)lra_ensure(ctx, 60, (nA + 1) * 8);
  int32_t* q_len = (int32_t*)lra_ensure(ctx, 61, (nA + 1) * 4);
  uint64_t* t_off = (uint64_t*)lra_ensure(ctx, 62, (nA + 1) * 8);
  int64_t* t_len = (int64_t*)lra_ensure(ctx, 63, (nA + 1) * 8);
  if (!aln_read || !q_off || !q_len || !t_off || !t_len) return LRA_ERR_NOMEM;
  if (nJ) hipLaunchKernelGGL(k_aln_address, dim3((unsigned)((nJ + 255) / 256)), dim3(256), 0, st, nJ, num_aln, ares.d_job_aln_off, ares.d_strand, ares.d_chrom,
                             d_read_off, tot, m->d_chrom_pos, aln_read, q_off, q_len, t_off, t_len);
  lra_refine_result fres; memset(&fres, 0, sizeof fres);
  lra_stats_result tres; memset(&tres, 0, sizeof tres);
  if (nA) {
    if (o->skipBandedRefine) {
      fres.n_aln = (int)nA; fres.n_blocks = ares.n_blocks; fres.d_block_off = ares.d_block_off; fres.d_blocks = ares.d_blocks; fres.d_status = nullptr;
    } else if ((rc = lra_indel_refine_batch(ctx, (int)nA, ares.d_blocks, ares.d_block_off, ares.n_blocks, both, q_off, q_len, genome, t_off, t_len, o->refineBand,
                                            o->localMatch, o->localMismatch, o->localIndel, 0, &fres))) return rc;
    if (fres.d_status) {                                                  // the refine stage's status array lives in scratch the next stage reuses
      int32_t* keep = (int32_t*)lra_ensure(ctx, 64, (nA + 1) * 4);
      if (!keep) return LRA_ERR_NOMEM;
      LRA_HIP_CHECK(ctx, hipMemcpyAsync(keep, fres.d_status, nA * 4, hipMemcpyDeviceToDevice, st));
      fres.d_status = keep;
      hipLaunchKernelGGL(k_or_status_idx, grid(nA), dim3(256), 0, st, nA, (const uint32_t*)keep, aln_read, 1, read_status);
    }
    if (o->refineBreakpoint && (rc = lra_refine_breakpoints(ctx, nJ, nA, ares.d_job_aln_off, ares.d_strand, q_off, q_len, t_off, t_len, both, genome, &fres))) return rc;
    stage("indel_refine (+breakpoints)");
    if ((rc = lra_calculate_statistics_batch(ctx, (int)nA, fres.d_blocks, fres.d_block_off, both, q_off, q_len, genome, t_off, m->lut.data(), (int)m->lut.size(), &tres)))
      return rc;
    stage("statistics");
  }
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  LRA_HIP_CHECK(ctx, hipGetLastError());
  out->num_aln = num_aln; out->n_jobs = nJ; out->n_alignments = nA; out->n_blocks = fres.n_blocks; out->n_runs = tres.n_runs;
  out->d_job_aln_off = ares.d_job_aln_off; out->d_job_status = ares.d_status; out->d_job_reached = job_reached; out->d_read_status = read_status;
  out->d_aln_read = aln_read; out->d_strand = ares.d_strand; out->d_supp = ares.d_supp; out->d_secondary = ares.d_secondary; out->d_n0 = ares.d_n0; out->d_n1 = ares.d_n1;
  out->d_chrom = ares.d_chrom; out->d_first_sdp_value = ares.d_value;
  out->d_block_off = fres.d_block_off; out->d_blocks = fres.d_blocks; out->d_refine_status = fres.d_status;
  out->d_counts = tres.d_counts; out->d_value = tres.d_value; out->d_run_off = tres.d_run_off; out->d_runs = tres.d_runs;
  out->d_strands = both; out->rc_base = tot;
  if (getenv("LRA_MEM_DBG")) {
    size_t tot = 0;
    for (int i = 0; i < 192; i++) { tot += ctx->gbytes[i]; if (ctx->gbytes[i] > (size_t(1) << 30)) fprintf(stderr, "[mem] gbuf %d %.1f GB\n", i, ctx->gbytes[i] / 1e9); }
    for (int i = 0; i < 4; i++) { tot += ctx->scratch_bytes[i]; fprintf(stderr, "[mem] scratch %d %.1f GB\n", i, ctx->scratch_bytes[i] / 1e9); }
    fprintf(stderr, "[mem] aux %.1f out %.1f GB total %.1f GB\n", ctx->aux_bytes / 1e9, ctx->out_bytes / 1e9, (tot + ctx->aux_bytes + ctx->out_bytes) / 1e9);
  }
  // counters of the batch (what bench.py prices the roofline with)
  lra_map_counters& c = out->counters;
  c = in.counters;
  c.n_merged_clusters = mres.n_groups; c.n_sdp2_anchors = mres.n_anchors; c.n_sdp2_entries = ch2.n_subproblem_entries; c.n_a13_blocks = ares.n_blocks;
  c.n_large_spaces = ares.n_big; c.n_segments = fres.n_segments; c.n_rows = fres.n_rows; c.n_cells = fres.n_cells; c.n_aog = fres.n_aog;
  return LRA_OK;
}

// lra_map_reads_lowacc_batch: the pass above.  With opts.defer_matches > 0 the batch's most repetitive reads -- a read inside a satellite array ends up with tens of
// thousands of refined matches where a typical 30 kb read has three thousand, and its second sparse DP keeps one workgroup busy for half a second -- leave the pass after
// Refine_Btwnsplitchain and go on, from MergeChain, in a child context on its own lowest-priority stream and host thread BESIDE the rest of the pass; the two results
// are merged on the device (a read's alignments do not depend on which pass computed them: tests/test_mapread.py ont-defer*).  Off in the presets: on this device it is
// no gain at any threshold -- the repetitive reads' work is throughput that the pass's own launches already overlap, not an idle tail.
namespace {
__global__ void k_count_flagged(int n, const uint32_t* __restrict__ st, unsigned long long* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long m = __ballot(i < n && st[i] != 0 && st[i] != (uint32_t)LRA_ST_DEFERRED);   // (a handed-back read is not a flagged one: counters.n_handed_back_reads)
  if ((threadIdx.x & (warpSize - 1)) == 0 && m) atomicAdd(out, (unsigned long long)__popcll(m));
}
}  // namespace
// counters.n_flagged_reads of a finished batch: the reads whose status word is non-zero get no alignment record (lra_map_records*), the caller must know how many
int lra_map_count_flagged(lra_ctx* ctx, lra_map_result* out) {
  out->counters.n_flagged_reads = 0;
  if (!out->d_read_status || out->n_reads <= 0) return LRA_OK;
  unsigned long long* d = (unsigned long long*)lra_ensure(ctx, 191, 64);
  if (!d) return LRA_ERR_NOMEM;
  LRA_HIP_CHECK(ctx, hipMemsetAsync(d, 0, 8, ctx->stream));
  hipLaunchKernelGGL(k_count_flagged, dim3((out->n_reads + 255) / 256), dim3(256), 0, ctx->stream, out->n_reads, out->d_read_status, d);
  unsigned long long h = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, ctx->stream));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  out->counters.n_flagged_reads = h;
  return LRA_OK;
}

// The context's companion (a batch's second, concurrent pass; the back half of two-stage batches): a context of its own -- stream, work buffers -- that borrows this
// one's reference data.  `lowest`: its streams at the device's lowest priority (the second pass fills the gaps the first one leaves; at equal priority the two passes'
// queues slow each other down far beyond the work involved, measured); otherwise at LRA_BACK_PRIORITY (default: the device's highest -- the back half of a batch is the longer one, the front half of the next fills in).
static int child_create(lra_ctx* ctx, bool lowest) {
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (!ctx->child) {
    lra_ctx* c = nullptr;
    int rc = lra_ctx_create(ctx->device, &c);
    if (rc) return lra_set_err(ctx, rc, "companion context");
    if ((rc = lra_ctx_share_reference(c, ctx))) { lra_ctx_destroy(c); return lra_set_err(ctx, rc, "companion context: sharing the reference"); }
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    int prio = least;
    if (!lowest) { prio = getenv("LRA_BACK_PRIORITY") ? atoi(getenv("LRA_BACK_PRIORITY")) : greatest; prio = std::max(greatest, std::min(least, prio)); }
    c->low_priority = true; c->prio = prio;                               // (low_priority: the context keeps the priority it was made with, lra_ctx_set_stream)
    if (hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio) != hipSuccess) { c->stream = nullptr; lra_ctx_destroy(c); return lra_set_err(ctx, LRA_ERR_HIP, "companion stream"); }
    c->owns_stream = true; c->timing = ctx->timing;
    ctx->child = c;
  }
  return LRA_OK;
}
// The companion's view of the parent's reference data (the parent's reference may have been loaded / built again since the last batch).  Writes the companion's
// state: only while nothing runs on the companion (two-stage batches: the front half calls it once it holds the back context, never while a back half may be running).
static int child_refresh(lra_ctx* ctx) {
  {
    lra_ctx* c = ctx->child;
    int rc = lra_seed_share(c, ctx);
    if (rc) return lra_set_err(c, rc, "companion context: sharing the reference");   // (on the companion: in two-stage batches this runs on the back halves' thread)
    lra_map_state* d = c->map; const lra_map_state* s = ctx->map;
    d->chrom_pos = s->chrom_pos; d->d_chrom_pos = s->d_chrom_pos; d->gli_buf = s->gli_buf; d->gli = s->gli; d->d_gso = s->d_gso; d->n_gwin = s->n_gwin;
    d->gli_window = s->gli_window; d->gli_k = s->gli_k; d->gli_w = s->gli_w; d->lut = s->lut; d->borrowed = true;
    d->owner_cell = s->borrowed ? s->owner_cell : s->cell; d->owner_generation = s->borrowed ? s->owner_generation : s->cell->gen.load();
  }
  return LRA_OK;
}
static int ensure_child(lra_ctx* ctx, bool lowest) {
  int rc = child_create(ctx, lowest);
  if (rc) return rc;
  if ((rc = child_refresh(ctx))) return lra_set_err(ctx, rc, "%s", ctx->child->err.c_str());
  return LRA_OK;
}

static int lowacc_batch_impl(lra_ctx* ctx, int n_reads, const char* d_seq, const uint64_t* d_read_off, uint64_t total_bases, const lra_map_opts* o, lra_map_result* out);
extern "C" int lra_map_reads_lowacc_batch(lra_ctx* ctx, int n_reads, const char* d_seq, const uint64_t* d_read_off, uint64_t total_bases,
                                          const lra_map_opts* o, lra_map_result* out) {
  int rc = lowacc_batch_impl(ctx, n_reads, d_seq, d_read_off, total_bases, o, out);
  if (rc == LRA_OK && out && n_reads > 0) rc = lra_map_count_flagged(ctx, out);
  return rc;
}
static int lowacc_batch_impl(lra_ctx* ctx, int n_reads, const char* d_seq, const uint64_t* d_read_off, uint64_t total_bases, const lra_map_opts* o, lra_map_result* out) {
  if (!ctx || !o || !out || n_reads < 0) return LRA_ERR_INVALID;
  memset(out, 0, sizeof *out);
  lra_map_state* m = ctx->map;
  if (!m || !m->gli_buf || !ctx->seed || !ctx->seed->genome || !ctx->seed->idx_key)
    return lra_set_err(ctx, LRA_ERR_INVALID, "reference not loaded (genome, global index, chromosome table, local index)");
  if (m->gli_window != o->localIndexWindow || m->gli_k != o->localK || m->gli_w != o->localW)
    return lra_set_err(ctx, LRA_ERR_INVALID, "the genome's local index has k = %d, w = %d, windows of %d bases; the options say %d, %d, %d (lra_map_opts_apply_local_index: glIndex.Read overrides them)",
                       m->gli_k, m->gli_w, m->gli_window, o->localK, o->localW, o->localIndexWindow);
  { int rcs = lra_map_check_shared(ctx); if (rcs) return rcs; }
  out->n_reads = n_reads;
  m->last_text.clear(); m->last_sig = lra_map_sig{};         // a sizing call of lra_map_records for an earlier batch is void now
  ctx->pipelined = false;                                    // (the one call: nothing runs beside it)
  if (n_reads == 0) return LRA_OK;
  uint32_t threshold = o->defer_matches > 0 ? (uint32_t)o->defer_matches : 0;
  if (const char* e = getenv("LRA_DEFER_MATCHES")) threshold = (uint32_t)std::max(0, atoi(e));
  const std::function<int()> none;
  if (!threshold) return lowacc_core(ctx, n_reads, d_seq, d_read_off, total_bases, o, out, 0, nullptr, nullptr, nullptr, none);
  { int rcc = ensure_child(ctx, true); if (rcc) return rcc; }
  std::vector<uint32_t> picked;
  std::thread second;
  int rc2 = LRA_OK;
  LowaccTailIn in2;
  lra_map_result o2; memset(&o2, 0, sizeof o2);
  const bool ddbg = getenv("LRA_DEFER_DBG") != nullptr;
  auto wall = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = wall();
  double t_split = 0, t_second = 0;
  auto start_second = [&]() -> int {
    t_split = wall();
    if (getenv("LRA_DEFER_DROP")) { picked.clear(); return LRA_OK; }      // (experiment: the first pass without the deferred reads, nothing else running)
    second = std::thread([&, c = ctx->child]() {
      rc2 = lowacc_tail(c, in2, o, &o2);
      t_second = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    });
    return LRA_OK;
  };
  int rc = lowacc_core(ctx, n_reads, d_seq, d_read_off, total_bases, o, out, threshold, &picked, ctx->child, &in2, start_second);
  const double t1 = wall();
  if (second.joinable()) second.join();
  if (ddbg) fprintf(stderr, "[defer] %d reads; split at %.0f ms, second pass done at %.0f ms, first pass done at %.0f ms\n", (int)picked.size(), t_split - t0, t_second - t0, t1 - t0);
  if (rc) return rc;
  if (picked.empty()) return LRA_OK;
  if (rc2) return lra_set_err(ctx, rc2, "second pass (%d reads): %s", (int)picked.size(), ctx->child ? ctx->child->err.c_str() : "");
  // ---- merge: the job slots of the second pass's reads from its result, every other slot from the first pass's
  hipStream_t st = ctx->stream;
  auto grid = [](uint64_t n) { return dim3((unsigned)((n + 255) / 256)); };
  const int na = out->num_aln, R = n_reads, R2 = (int)picked.size();
  if (o2.num_aln != na) return lra_set_err(ctx, LRA_ERR_INVALID, "passes disagree on NumAln");
  const uint64_t S = (uint64_t)R * na;
  std::vector<int32_t> inB((size_t)R, -1);
  for (int i = 0; i < R2; i++) inB[picked[i]] = (int32_t)picked[i];    // (the second pass keeps the batch's slot numbering)
  int32_t* d_inB = (int32_t*)lra_ensure(ctx, 173, ((size_t)R + 4) * 4);
  uint64_t* d_src = (uint64_t*)lra_ensure(ctx, 175, (S + 4) * 8);
  if (!d_inB || !d_src) return LRA_ERR_NOMEM;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(d_inB, inB.data(), (size_t)R * 4, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_src_slot, grid(S), dim3(256), 0, st, S, na, (const int32_t*)d_inB, d_src);
  const lra_map_result a = *out;
  lra_merge::PassView A = lra_merge::view_of(a), B = lra_merge::view_of(o2);
  char* extra = nullptr;
  lra_map_result mo; memset(&mo, 0, sizeof mo);
  if ((rc = lra_merge::merge_passes(ctx, 171, S, na, d_src, A, B, a.n_alignments + o2.n_alignments, a.n_blocks + o2.n_blocks, a.n_runs + o2.n_runs,
                                    lra_merge::al256(S + 64) + ((size_t)R + 4) * 4, &extra, &mo))) return rc;
  uint8_t* reached = (uint8_t*)extra; uint32_t* rstat = (uint32_t*)(extra + lra_merge::al256(S + 64));
  hipLaunchKernelGGL(k_merge_reached, grid(S), dim3(256), 0, st, S, (const uint64_t*)d_src, a.d_job_reached, o2.d_job_reached, reached);
  hipLaunchKernelGGL(k_merge_read_status, grid(R), dim3(256), 0, st, R, (const int32_t*)d_inB, a.d_read_status, o2.d_read_status, rstat);
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));                          // (inB is pageable; and the result is the caller's to read now)
  LRA_HIP_CHECK(ctx, hipGetLastError());
  *out = mo;
  out->n_reads = R; out->num_aln = na; out->d_job_reached = reached; out->d_read_status = rstat; out->d_strands = a.d_strands; out->rc_base = a.rc_base;
  // counters: the stages up to the split point saw every read in the first pass; the later ones saw each read in one pass only
  lra_map_counters c = a.counters; const lra_map_counters& b = o2.counters;
  c.n_merged_clusters += b.n_merged_clusters; c.n_sdp2_anchors += b.n_sdp2_anchors; c.n_sdp2_entries += b.n_sdp2_entries; c.n_a13_blocks += b.n_a13_blocks;
  c.n_large_spaces += b.n_large_spaces; c.n_segments += b.n_segments; c.n_rows += b.n_rows; c.n_cells += b.n_cells; c.n_aog += b.n_aog;
  c.n_deferred_reads = a.counters.n_deferred_reads + (uint64_t)R2;
  out->counters = c;
  return LRA_OK;
}

// ---- two-stage batches: a batch's front half (a1 .. the second LinearExtend) on the context, its back half (second sparse DP .. statistics) on the companion
// context, so that batch i + 1's front half runs BESIDE batch i's back half (two host threads).  The front half is short wide kernels and rounds of small latency-bound
// launches, the back half is the sparse DP over the merged clusters and the banded refinement: side by side each fills what the other leaves idle.
static int front_checks(lra_ctx* ctx, int n_reads, const lra_map_opts* o) {
  if (!ctx || !o || n_reads < 0) return LRA_ERR_INVALID;
  lra_map_state* m = ctx->map;
  if (!m || !m->gli_buf || !ctx->seed || !ctx->seed->genome || !ctx->seed->idx_key)
    return lra_set_err(ctx, LRA_ERR_INVALID, "reference not loaded (genome, global index, chromosome table, local index)");
  if (m->gli_window != o->localIndexWindow || m->gli_k != o->localK || m->gli_w != o->localW)
    return lra_set_err(ctx, LRA_ERR_INVALID, "the genome's local index has k = %d, w = %d, windows of %d bases; the options say %d, %d, %d (lra_map_opts_apply_local_index: glIndex.Read overrides them)",
                       m->gli_k, m->gli_w, m->gli_window, o->localK, o->localW, o->localIndexWindow);
  if (o->defer_matches > 0 || o->defer_seed_matches > 0 || getenv("LRA_DEFER_MATCHES")) return lra_set_err(ctx, LRA_ERR_INVALID, "two-stage batches do not combine with defer_matches / defer_seed_matches");
  return lra_map_check_shared(ctx);
}
// A front half that fails still hands over a batch -- an error batch: the back call that takes it returns the front half's code and holds nothing, so the
// thread that runs the back halves is never left waiting for a batch that will not come (one back call per front call, whatever the front call returned).
static int front_failed(lra_ctx* ctx, lra_handover* H, int rc) {
  const std::string msg = ctx->err;
  { std::unique_lock<std::mutex> lk(H->mu); H->cv.wait(lk, [&] { return !H->pending; }); H->in = LowaccTailIn(); H->in.n_reads = 0; H->rc = rc; H->err = msg; H->pending = true; }
  H->cv.notify_all();
  return rc;
}
// the back context and the two handover contexts (b -> hand[0] -> hand[1] on the child chain: destroyed with the context, timed with it); made once, by the first front call
static int two_stage_contexts(lra_ctx* ctx, lra_handover* H) {
  if (!ctx->child) { int rc = ensure_child(ctx, false); if (rc) return rc; }
  {
    // A companion the one call's second pass made earlier (defer_matches: lowest priority, the one call's tuning) becomes the back context: the back half's priority
    // and the choices made for device time, whoever made it.
    lra_ctx* c = ctx->child;
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    int want = getenv("LRA_BACK_PRIORITY") ? atoi(getenv("LRA_BACK_PRIORITY")) : greatest;
    want = std::max(greatest, std::min(least, want));
    if (c->owns_stream && c->stream && c->prio != want && !H->hand[0]) {     // (only before the first two-stage batch: nothing of a pipeline runs on it yet)
      LRA_HIP_CHECK(ctx, hipStreamSynchronize(c->stream));
      hipStream_t ns = nullptr;
      if (hipStreamCreateWithPriority(&ns, hipStreamNonBlocking, want) != hipSuccess) return lra_set_err(ctx, LRA_ERR_HIP, "companion stream");
      (void)hipStreamDestroy(c->stream);
      c->stream = ns; c->prio = want;
    }
    c->pipelined = true;
  }
  lra_ctx* tail = ctx->child;
  for (int i = 0; i < 2; i++) {
    if (!H->hand[i]) {
      if (tail->child) return lra_set_err(ctx, LRA_ERR_INVALID, "the companion context already has a companion of its own");
      lra_ctx* c = nullptr;
      int rc = lra_ctx_create(ctx->device, &c);
      if (rc) return lra_set_err(ctx, rc, "handover context");
      c->timing = ctx->timing;
      tail->child = c; H->hand[i] = c;
    }
    tail = H->hand[i];
  }
  return LRA_OK;
}
extern "C" int lra_map_reads_lowacc_front(lra_ctx* ctx, int n_reads, const char* d_seq, const uint64_t* d_read_off, uint64_t total_bases, const lra_map_opts* o) {
  if (!ctx) return LRA_ERR_INVALID;
  lra_handover* H = handover_of(ctx);
  { int rc = front_checks(ctx, n_reads, o); if (rc) return front_failed(ctx, H, rc); }
  { int rc = two_stage_contexts(ctx, H); if (rc) return front_failed(ctx, H, rc); }
  ctx->pipelined = true;
  if (n_reads == 0) {                                                     // an empty batch still takes its turn
    { std::unique_lock<std::mutex> lk(H->mu); H->cv.wait(lk, [&] { return !H->pending; }); H->in = LowaccTailIn(); H->in.n_reads = 0; H->rc = LRA_OK; H->pending = true; }
    H->cv.notify_all();
    return LRA_OK;
  }
  lra_map_result tmp;
  const std::function<int()> none;
  const int rc = lowacc_core(ctx, n_reads, d_seq, d_read_off, total_bases, o, &tmp, 0, nullptr, nullptr, nullptr, none, H);
  return rc ? front_failed(ctx, H, rc) : LRA_OK;                         // (lowacc_core hands the batch over as its last act: a failure means it has not)
}
extern "C" int lra_map_reads_lowacc_back(lra_ctx* ctx, const lra_map_opts* o, lra_map_result* out, lra_ctx** back_ctx) {
  if (!ctx || !o || !out) return LRA_ERR_INVALID;
  lra_handover* H = handover_of(ctx);
  LowaccTailIn in;
  int frc = LRA_OK; std::string ferr;
  const double tw0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
  {                                                                      // (waits for a front half, however long; a front half that fails hands over an error batch)
    std::unique_lock<std::mutex> lk(H->mu);
    if (H->busy) return lra_set_err(ctx->child ? ctx->child : ctx, LRA_ERR_INVALID, "the result of the back half before this one is still held (lra_map_back_release)");
    H->cv.wait(lk, [&] { return H->pending; });
    in = H->in; frc = H->rc; ferr = H->err; H->rc = LRA_OK; H->err.clear();
    H->pending = false;
    H->busy = frc == LRA_OK;                                             // an error batch holds nothing
  }
  H->cv.notify_all();                                                    // (the front half may hand over the next batch now)
  if (getenv("LRA_TWO_STAGE_DBG")) fprintf(stderr, "[two-stage] back half waited %.0f ms for a front half\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - tw0);
  memset(out, 0, sizeof *out);
  lra_ctx* b = ctx->child;                                               // (made by the first front half)
  if (back_ctx) *back_ctx = b;
  // (this thread's error text goes to the BACK context -- the front thread writes ctx's; without a back context the front half has failed before making one, and is not running)
  if (frc) return lra_set_err(b ? b : ctx, frc, "front half of this batch failed: %s", ferr.c_str());
  // The back context's view of the reference data (borrowed from ctx), refreshed here -- by the thread that owns the back context, with nothing running on it.
  // (Reloading ctx's reference is for when no batch is in flight: the halves of the batches in flight read it.)
  int rc = child_refresh(ctx);
  if (rc == LRA_OK) {
    b->map->last_text.clear(); b->map->last_sig = lra_map_sig{};
    out->n_reads = in.n_reads;
    if (in.n_reads == 0) return LRA_OK;
    rc = lowacc_tail(b, in, o, out);
  }
  if (rc == LRA_OK) rc = lra_map_count_flagged(b, out);
  if (rc) { const std::string msg = b->err; return lra_set_err(b, rc, "back half: %s", msg.c_str()); }
  return LRA_OK;
}
extern "C" int lra_map_back_release(lra_ctx* ctx) {
  if (!ctx || !ctx->handover) return LRA_ERR_INVALID;
  lra_handover* H = ctx->handover;
  { std::lock_guard<std::mutex> lk(H->mu); if (!H->busy) return lra_set_err(ctx->child ? ctx->child : ctx, LRA_ERR_INVALID, "no back half's result is held"); H->busy = false; }
  H->cv.notify_all();
  return LRA_OK;
}

// ---------------------------------------------------------------------------------------------------------------- records (host)
// lra_map_snapshot copies what the records need from the context's result buffers to the host (so the next batch may overwrite them);
// lra_map_records_host turns a snapshot into text with host threads only -- it touches neither the context nor the device, so it runs
// beside the next batch's lra_map_reads_lowacc_batch.  lra_map_records = both, with the reference's two-call output convention.
struct lra_map_host {
  int32_t n_reads = 0, num_aln = 1; uint64_t nJ = 0, nA = 0;
  std::vector<uint64_t> jo, roff, boff; std::vector<int32_t> strand, supp, sec, n0, n1, chrom, counts, blocks; std::vector<float> fval;
  lra_pod_buf<uint32_t> runs;                              // the CIGAR runs: 0.7 GB per 32768 reads of 30 kb
  std::vector<uint32_t> rstat, ends;                       // ends: per alignment first block's qPos, last block's qPos + length
  std::vector<uint8_t> reached;
  std::vector<uint64_t> chrom_pos;
  std::vector<std::string> segText; std::vector<uint32_t> segStart;   // print format 'a' only
  lra_text_buf text; std::vector<uint64_t> rec_off;        // what lra_map_records_host produced last
};

namespace {
template <typename T>
int fetch(lra_ctx* ctx, std::vector<T>& v, const T* d, size_t n) {
  v.resize(n);
  if (n && d) LRA_HIP_CHECK(ctx, hipMemcpy(v.data(), d, n * sizeof(T), hipMemcpyDeviceToHost));
  return LRA_OK;
}
__global__ void k_block_ends(uint64_t nA, const uint64_t* __restrict__ boff, const int32_t* __restrict__ blocks, uint32_t* __restrict__ ends) {
  const uint64_t a = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= nA) return;
  const uint64_t b0 = boff[a], b1 = boff[a + 1];
  ends[2 * a] = b1 > b0 ? (uint32_t)blocks[3 * b0] : 0;
  ends[2 * a + 1] = b1 > b0 ? (uint32_t)(blocks[3 * (b1 - 1)] + blocks[3 * (b1 - 1) + 2]) : 0;
}
}  // namespace

extern "C" void lra_map_host_free(lra_map_host* h) { delete h; }
extern "C" uint64_t lra_map_host_flagged(const lra_map_host* h, const uint32_t** status) {
  if (status) *status = nullptr;
  if (!h) return 0;
  uint64_t n = 0;
  for (uint32_t v : h->rstat) n += v != 0 && v != (uint32_t)LRA_ST_DEFERRED;   // (a handed-back read is not a flagged one)
  if (status && !h->rstat.empty()) *status = h->rstat.data();
  return n;
}

namespace {
std::mutex g_pool_mu;
struct PoolBlock { void* p; size_t cap; };
std::vector<PoolBlock> g_pool;                                           // at most POOL_KEEP blocks, the largest ones
constexpr size_t POOL_KEEP = 6, POOL_MIN = 8u << 20;
}  // namespace
// the record threads' parts of a batch's text (lra_map_records_host): strings kept between batches with their capacity -- a part is ~100 MB, and a fresh one is
// 25 000 page faults and a chain of doubling reallocations
namespace {
std::mutex g_parts_mu;
std::vector<std::string> g_parts;
size_t g_parts_bytes = 0;                                                // capacity held by the pool
size_t parts_cap() {                                                     // at most this much is kept between batches (LRA_PARTS_POOL_MB; lra_map_host_trim lowers what is held now)
  static const size_t cap = [] { const char* e = getenv("LRA_PARTS_POOL_MB"); return (size_t)(e ? std::max(0, atoi(e)) : 4096) << 20; }();
  return cap;
}
std::string part_take(size_t want) {
  std::string s;
  {
    std::lock_guard<std::mutex> lk(g_parts_mu);
    int best = -1;                                                       // the smallest string that holds `want`, else the largest there is
    for (int i = 0; i < (int)g_parts.size(); i++) {
      const size_t c = g_parts[i].capacity();
      if (best < 0) { best = i; continue; }
      const size_t b = g_parts[best].capacity();
      if (b >= want ? (c >= want && c < b) : c > b) best = i;
    }
    if (best >= 0) { g_parts_bytes -= g_parts[best].capacity(); s.swap(g_parts[best]); g_parts.erase(g_parts.begin() + best); }
  }
  s.clear();
  if (s.capacity() < want) s.reserve(want);
  return s;
}
void part_give(std::string& s) {
  s.clear();
  std::lock_guard<std::mutex> lk(g_parts_mu);
  if (g_parts.size() < 64 && s.capacity() >= (8u << 20) && g_parts_bytes + s.capacity() <= parts_cap()) { g_parts_bytes += s.capacity(); g_parts.emplace_back(); g_parts.back().swap(s); }
  else std::string().swap(s);
}
}  // namespace
// Host memory the record stage keeps between batches (the threads' text parts): released down to keep_bytes (0: all of it).  Returns the bytes still held.
extern "C" uint64_t lra_map_host_trim(uint64_t keep_bytes) {
  std::lock_guard<std::mutex> lk(g_parts_mu);
  while (!g_parts.empty() && g_parts_bytes > keep_bytes) {
    int big = 0;
    for (int i = 1; i < (int)g_parts.size(); i++) if (g_parts[i].capacity() > g_parts[big].capacity()) big = i;
    g_parts_bytes -= g_parts[big].capacity();
    g_parts.erase(g_parts.begin() + big);
  }
  if (g_parts.empty()) std::vector<std::string>().swap(g_parts);
  return g_parts_bytes;
}
void* lra_host_pool_get(size_t bytes, size_t* cap) {
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    int best = -1;
    for (int i = 0; i < (int)g_pool.size(); i++) if (g_pool[i].cap >= bytes && (best < 0 || g_pool[i].cap < g_pool[best].cap)) best = i;
    if (best >= 0 && g_pool[best].cap <= 2 * bytes + POOL_MIN) { void* p = g_pool[best].p; *cap = g_pool[best].cap; g_pool.erase(g_pool.begin() + best); return p; }
  }
  *cap = bytes;
  return malloc(bytes);
}
void lra_host_pool_put(void* p, size_t cap) {
  if (!p) return;
  if (cap >= POOL_MIN) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (g_pool.size() < POOL_KEEP) { g_pool.push_back({p, cap}); return; }
    int smallest = 0;
    for (int i = 1; i < (int)g_pool.size(); i++) if (g_pool[i].cap < g_pool[smallest].cap) smallest = i;
    if (g_pool[smallest].cap < cap) { void* q = g_pool[smallest].p; g_pool[smallest] = {p, cap}; p = q; }
  }
  free(p);
}

// ---- the record buffer of a batch: everything the host tail needs, packed into one device buffer (what a rank sends to rank 0)
//   int64 header[16] = {magic, n_reads, num_aln, nJ, nA, n_blocks (0 unless with_blocks), n_runs, n_chrom, has_reached, has_rstat, ...}
//   then, each padded to 8 bytes:  chrom_pos u64[n_chrom+1] | reached u8[nJ] | rstat u32[n_reads] | jo u64[nJ+1] | strand, supp, sec, n0, n1, chrom
//   i32[nA] each | fval f32[nA] | counts i32[18 nA] | boff u64[nA+1] | ends u32[2 nA] | roff u64[nA+1] | runs u32[n_runs] | blocks i32[3 n_blocks]
namespace {
constexpr int64_t PACK_MAGIC = 0x4c52414d41503031LL;   // "LRAMAP01"
inline size_t pad8(size_t n) { return (n + 7) & ~(size_t)7; }
struct PackLayout {
  size_t off[17]; size_t total;
  PackLayout(uint64_t n_reads, uint64_t nJ, uint64_t nA, uint64_t n_blocks, uint64_t n_runs, uint64_t n_chrom) {
    const size_t sz[17] = {16 * 8, (n_chrom + 1) * 8, nJ, n_reads * 4, (nJ + 1) * 8, nA * 4, nA * 4, nA * 4, nA * 4, nA * 4, nA * 4, nA * 4, 18 * nA * 4, (nA + 1) * 8,
                           2 * nA * 4, (nA + 1) * 8, n_runs * 4};
    size_t at = 0;
    for (int i = 0; i < 17; i++) { off[i] = at; at += pad8(sz[i]); }
    blocks_off = at; at += pad8(3 * n_blocks * 4);
    total = at;
  }
  size_t blocks_off;
};
}  // namespace

extern "C" int lra_map_pack(lra_ctx* ctx, const lra_map_result* res, int with_blocks, const void** d_buf, uint64_t* bytes) {
  if (!ctx || !res || !d_buf || !bytes) return LRA_ERR_INVALID;
  lra_map_state* m = ctx->map;
  if (!m) return LRA_ERR_INVALID;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const uint64_t nR = (uint64_t)res->n_reads, nJ = res->n_jobs, nA = res->n_alignments, nB = with_blocks ? res->n_blocks : 0, nRuns = res->n_runs;
  const uint64_t nCh = m->chrom_pos.size() - 1;
  const PackLayout L(nR, nJ, nA, nB, nRuns, nCh);
  char* buf = (char*)lra_ensure(ctx, 84, L.total + 64);
  if (!buf) return LRA_ERR_NOMEM;
  const int64_t hdr[16] = {PACK_MAGIC, (int64_t)nR, std::max(res->num_aln, 1), (int64_t)nJ, (int64_t)nA, (int64_t)nB, (int64_t)nRuns, (int64_t)nCh,
                           res->d_job_reached ? 1 : 0, res->d_read_status ? 1 : 0, 0, 0, 0, 0, 0, 0};
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(buf + L.off[0], hdr, sizeof hdr, hipMemcpyHostToDevice, st));
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(buf + L.off[1], m->d_chrom_pos, (nCh + 1) * 8, hipMemcpyDeviceToDevice, st));
  auto put = [&](int slot, const void* src, size_t n) -> hipError_t { return (n && src) ? hipMemcpyAsync(buf + L.off[slot], src, n, hipMemcpyDeviceToDevice, st) : hipSuccess; };
  if (nA) {
    uint32_t* d_ends = (uint32_t*)(buf + L.off[14]);
    hipLaunchKernelGGL(k_block_ends, dim3((unsigned)((nA + 255) / 256)), dim3(256), 0, st, nA, res->d_block_off, res->d_blocks, d_ends);
  }
  LRA_HIP_CHECK(ctx, put(2, res->d_job_reached, nJ));
  LRA_HIP_CHECK(ctx, put(3, res->d_read_status, nR * 4));
  LRA_HIP_CHECK(ctx, put(4, res->d_job_aln_off, nJ ? (nJ + 1) * 8 : 0));
  LRA_HIP_CHECK(ctx, put(5, res->d_strand, nA * 4)); LRA_HIP_CHECK(ctx, put(6, res->d_supp, nA * 4)); LRA_HIP_CHECK(ctx, put(7, res->d_secondary, nA * 4));
  LRA_HIP_CHECK(ctx, put(8, res->d_n0, nA * 4)); LRA_HIP_CHECK(ctx, put(9, res->d_n1, nA * 4)); LRA_HIP_CHECK(ctx, put(10, res->d_chrom, nA * 4));
  LRA_HIP_CHECK(ctx, put(11, res->d_first_sdp_value, nA * 4)); LRA_HIP_CHECK(ctx, put(12, res->d_counts, 18 * nA * 4));
  LRA_HIP_CHECK(ctx, put(13, res->d_block_off, nA ? (nA + 1) * 8 : 0)); LRA_HIP_CHECK(ctx, put(15, res->d_run_off, nA ? (nA + 1) * 8 : 0));
  LRA_HIP_CHECK(ctx, put(16, res->d_runs, nRuns * 4));
  if (nB) LRA_HIP_CHECK(ctx, hipMemcpyAsync(buf + L.blocks_off, res->d_blocks, 3 * nB * 4, hipMemcpyDeviceToDevice, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  *d_buf = buf; *bytes = L.total;
  return LRA_OK;
}

// a packed record buffer (host memory) -> snapshot
extern "C" int lra_map_unpack_host(const void* h_buf, uint64_t bytes, lra_map_host** out) {
  if (!h_buf || !out || bytes < 16 * 8) return LRA_ERR_INVALID;
  *out = nullptr;
  const char* b = (const char*)h_buf;
  int64_t hdr[16];
  memcpy(hdr, b, sizeof hdr);
  if (hdr[0] != PACK_MAGIC) return LRA_ERR_INVALID;
  const uint64_t nR = (uint64_t)hdr[1], nJ = (uint64_t)hdr[3], nA = (uint64_t)hdr[4], nB = (uint64_t)hdr[5], nRuns = (uint64_t)hdr[6], nCh = (uint64_t)hdr[7];
  const PackLayout L(nR, nJ, nA, nB, nRuns, nCh);
  if (L.total > bytes) return LRA_ERR_INVALID;
  lra_map_host* h = new lra_map_host();
  h->n_reads = (int32_t)nR; h->num_aln = (int)hdr[2]; h->nJ = nJ; h->nA = nA;
  auto get = [&](auto& v, int slot, size_t n) { v.resize(n); if (n) memcpy(v.data(), b + L.off[slot], n * sizeof(v[0])); };
  get(h->chrom_pos, 1, nCh + 1);
  if (hdr[8]) get(h->reached, 2, nJ);
  if (hdr[9]) get(h->rstat, 3, nR);
  get(h->jo, 4, nJ ? nJ + 1 : 0);
  get(h->strand, 5, nA); get(h->supp, 6, nA); get(h->sec, 7, nA); get(h->n0, 8, nA); get(h->n1, 9, nA); get(h->chrom, 10, nA); get(h->fval, 11, nA);
  get(h->counts, 12, 18 * nA); get(h->boff, 13, nA ? nA + 1 : 0); get(h->ends, 14, 2 * nA); get(h->roff, 15, nA ? nA + 1 : 0);
  if (nRuns) {                                                           // the one large array: uninitialised (pooled) memory, copied by a few threads
    if (!h->runs.alloc(nRuns)) { delete h; return LRA_ERR_NOMEM; }
    const char* src = b + L.off[16]; char* dst = (char*)h->runs.data(); const size_t tot = nRuns * 4;
    const int T = tot > (64u << 20) ? std::min(16, lra_host_threads()) : 1;
    auto cp = [&](int t) { const size_t lo = tot * t / T, hi = tot * (t + 1) / T; memcpy(dst + lo, src + lo, hi - lo); };
    if (T == 1) cp(0);
    else { std::vector<std::thread> th; for (int t = 0; t < T; t++) th.emplace_back(cp, t); for (auto& x : th) x.join(); }
  }
  if (nB) { h->blocks.resize(3 * nB); memcpy(h->blocks.data(), b + L.blocks_off, 3 * nB * 4); }
  *out = h;
  return LRA_OK;
}

extern "C" int lra_map_snapshot(lra_ctx* ctx, const lra_map_result* res, int with_blocks, lra_map_host** out) {
  if (!ctx || !res || !out) return LRA_ERR_INVALID;
  *out = nullptr;
  const void* d_buf = nullptr; uint64_t bytes = 0;
  int rc = lra_map_pack(ctx, res, with_blocks, &d_buf, &bytes);
  if (rc) return rc;
  char* hb = (char*)lra_pinned(ctx, bytes);                                // (page-locked and kept: see lra_pinned)
  if (!hb) return LRA_ERR_NOMEM;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(hb, d_buf, bytes, hipMemcpyDeviceToHost, ctx->stream));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  lra_map_host* h = nullptr;
  if ((rc = lra_map_unpack_host(hb, bytes, &h))) return rc;
  if (with_blocks) {
    // print format 'a': the pairwise text needs the chromosome bases under every alignment (and the read on its strand)
    lra_map_state* m = ctx->map;
    if (!ctx->seed || !ctx->seed->genome) { delete h; return lra_set_err(ctx, LRA_ERR_INVALID, "genome not loaded"); }
    const size_t nA = h->nA;
    h->segText.resize(nA); h->segStart.assign(nA, 0);
    for (size_t a = 0; a < nA; a++) {
      const uint64_t b0 = h->boff[a], b1 = h->boff[a + 1];
      if (b1 == b0) continue;
      const uint32_t t0 = (uint32_t)h->blocks[3 * b0 + 1], t1 = (uint32_t)(h->blocks[3 * (b1 - 1) + 1] + h->blocks[3 * (b1 - 1) + 2]);
      h->segStart[a] = t0;
      h->segText[a].resize((size_t)(t1 - t0) + 1);
      if (hipMemcpy(&h->segText[a][0], ctx->seed->genome + m->chrom_pos[h->chrom[a]] + t0, t1 - t0, hipMemcpyDeviceToHost) != hipSuccess) { delete h; return LRA_ERR_HIP; }
    }
  }
  *out = h;
  return LRA_OK;
}

extern "C" int lra_map_records_host(lra_map_host* h, const lra_map_opts* o, const char* const* names, const char* const* reads, const char* const* quals,
                                    const int32_t* read_len, const char* const* chrom_names, const char* passthrough, int n_threads, const char** text,
                                    uint64_t* len, const uint64_t** rec_off) {
  if (!h || !o || !names || !reads || !read_len || !chrom_names || !len) return LRA_ERR_INVALID;
  const size_t nA = h->nA, nJ = h->nJ;
  (void)nA;
  const int na = h->num_aln;
  const std::vector<uint64_t>& jo = h->jo; const std::vector<uint64_t>& roff = h->roff; const std::vector<uint64_t>& boff = h->boff;
  const std::vector<int32_t>&strand = h->strand, &supp = h->supp, &sec = h->sec, &n0 = h->n0, &n1 = h->n1, &chrom = h->chrom, &counts = h->counts, &blocks = h->blocks;
  const std::vector<float>& fval = h->fval; const lra_pod_buf<uint32_t>& runs = h->runs; const std::vector<uint32_t>&rstat = h->rstat, &ends = h->ends; const std::vector<uint8_t>& reached = h->reached;
  const bool pairwise = o->printFormat == 'a';
  const bool hi = !o->bypassClustering;                                   // MapRead_highacc's tail (Map_highacc.h:733-789)
  if (pairwise && h->segText.size() != h->nA) return LRA_ERR_INVALID;
  // every read is independent: host threads take contiguous ranges of reads, each builds its own text; ranges are joined in read order
  const int n_reads = h->n_reads;
  int T = n_threads > 0 ? n_threads : lra_host_threads();                  // n_threads = 0: what the host allows (a 30 kb read's record is ~43 KB of text: 1.4 GB per 32768 reads)
  T = std::max(1, std::min(T, n_reads / 32 + 1));
  if (const char* e = getenv("LRA_RECORD_THREADS")) T = std::max(1, atoi(e));
  std::vector<std::string> part(T);
  std::vector<std::vector<uint64_t>> plen(T);
  std::vector<int> prc(T, LRA_OK);
  auto work = [&](int tix) {
    const int lo = (int)((long)n_reads * tix / T), hi = (int)((long)n_reads * (tix + 1) / T);
    std::string& text = part[tix];
    {                                                                     // room for the range's text: the reads, their CIGAR runs (~3.3 characters each), the tags
      size_t want = 4096;
      for (int r = lo; r < hi; r++) want += (size_t)read_len[r] + 700;
      if (nJ && jo.size() > (size_t)hi * na) { const uint64_t a0 = jo[(size_t)lo * na], a1 = jo[(size_t)hi * na]; if (a1 < roff.size() && a0 <= a1) want += (size_t)((roff[a1] - roff[a0]) * 7 / 2) + (size_t)(a1 - a0) * 600; }
      text = part_take(want + want / 16);
    }
    std::vector<std::string> cigars;
    std::vector<lra_aln_record> recs;
    std::vector<int32_t> seg_off, index;
    std::vector<lra_aln_group> groups;
    std::vector<char> buf;
    std::string rcRead;
    int rc = LRA_OK;
    for (int r = lo; r < hi; r++) {
      recs.clear(); cigars.clear(); seg_off.assign(1, 0); rcRead.clear();
      const bool flagged = !rstat.empty() && rstat[r];
      if (flagged && (!o->flagged_unaligned || (rstat[r] & LRA_ST_DEFERRED))) { plen[tix].push_back(0); continue; }   // flagged read: no record (the caller routes it elsewhere; d_read_status, lra_map_host_flagged)
      // low-accuracy path: p == 0 left no SegAlignment (Map_lowacc.h:578-581); high-accuracy path: read.unaligned or alignments.size() == 0
      // (Map_highacc.h:778-781) = no chain of the read got its SegAlignmentGroup
      bool unaligned = flagged || nJ == 0 || jo[(size_t)r * na + 1] == jo[(size_t)r * na];   // (opts.flagged_unaligned: a flagged read is written as an unaligned one)
      bool sparseRead = false;                                            // the read took the REFINEclusters branch: smallOpts.globalK = glIndex.k (Map_highacc.h:430)
      if (hi && nJ && !flagged) {
        unaligned = true;
        for (int p = 0; p < na; p++) if (!reached.empty() && reached[(size_t)r * na + p]) { unaligned = false; sparseRead |= (reached[(size_t)r * na + p] & 2) != 0; }
      }
      if (!unaligned) {
        size_t total = 0;
        for (int p = 0; p < na; p++) total += (size_t)(jo[(size_t)r * na + p + 1] - jo[(size_t)r * na + p]);
        cigars.reserve(total);                                            // the records keep pointers into these strings
        for (int p = 0; p < na; p++) {
          const size_t j = (size_t)r * na + p;
          // a chain that never reaches :574 ends the loop over p (:267, :491); one that does keeps its (possibly empty) group (:574-600)
          // (on the high-accuracy path a chain without clusters is skipped, Map_highacc.h:697, and the loop goes on)
          if (!reached.empty() ? !reached[j] : jo[j + 1] == jo[j]) { if (hi) continue; break; }
          for (uint64_t a = jo[j]; a < jo[j + 1]; a++) {
            // (a 30 kb read at 10 % error has ~6000 runs, nearly all of one or two digits: written through a pointer into room for the longest form, not appended one
            // by one -- the CIGAR strings were two thirds of the record threads' time)
            std::string cg;
            const size_t nRuns = (size_t)(roff[a + 1] - roff[a]);
            cg.resize(nRuns * 11 + 1);
            char* w = &cg[0];
            for (uint64_t x = roff[a]; x < roff[a + 1]; x++) {
              uint32_t v = runs[x] >> 4;
              if (v < 10) *w++ = (char)('0' + v);
              else if (v < 100) { *w++ = (char)('0' + v / 10); *w++ = (char)('0' + v % 10); }
              else { char tmp[12]; int k = 12; do { tmp[--k] = (char)('0' + v % 10); v /= 10; } while (v); memcpy(w, tmp + k, (size_t)(12 - k)); w += 12 - k; }
              *w++ = "=XID"[runs[x] & 15];
            }
            cg.resize((size_t)(w - &cg[0]));
            cigars.push_back(std::move(cg));
            const int32_t* c = &counts[18 * a];
            lra_aln_record rec; memset(&rec, 0, sizeof rec);
            rec.read_name = names[r]; rec.read = reads[r]; rec.qual = quals ? quals[r] : nullptr; rec.read_len = read_len[r];
            rec.chrom = chrom_names[chrom[a]]; rec.genome_len = (uint32_t)(h->chrom_pos[chrom[a] + 1] - h->chrom_pos[chrom[a]]);
            rec.cigar = cigars.back().c_str();
            rec.strand = strand[a]; rec.supplementary = supp[a]; rec.is_secondary = sec[a];
            rec.nm = c[0]; rec.nmm = c[1]; rec.nins = c[2]; rec.ndel = c[3]; rec.tdel = c[4]; rec.tins = c[5]; rec.nSmallDel = c[6]; rec.nMedDel = c[7]; rec.nLargeDel = c[8];
            rec.nSmallIns = c[9]; rec.nMedIns = c[10]; rec.nLargeIns = c[11]; rec.pre_clip = c[12]; rec.suf_clip = c[13];
            rec.q_start = (uint32_t)c[14]; rec.q_end = (uint32_t)c[15]; rec.t_start = (uint32_t)c[16]; rec.t_end = (uint32_t)c[17];
            rec.value = fval[a]; rec.NumOfAnchors0 = n0[a]; rec.NumOfAnchors1 = n1[a];
            const uint64_t b0 = boff[a], b1 = boff[a + 1];
            rec.n_blocks = (int32_t)(b1 - b0);
            rec.first_block_qpos = ends[2 * a];
            rec.last_block_qend = ends[2 * a + 1];
            // Alignment::read is the strand the segment lies on: strands[str] (the constructor call Map_lowacc.h:560 / Map_highacc.h:704, UpdateParameters Alignment.h:506-507),
            // so a reverse-strand record's SEQ is the read's reverse complement (its quality string stays as it came, Alignment.h:717-733).  Rounds 1-5 wrote the read as
            // it came for both strands: the emitters were pinned with the read they were GIVEN, and nothing pinned which read the composition gives them.
            if (strand[a] && rcRead.empty()) {                            // CreateRC (SeqUtils.h:151)
              const int L = read_len[r];
              rcRead.resize((size_t)L);
              for (int x = 0; x < L; x++) {
                const char ch = reads[r][L - 1 - x];
                rcRead[x] = ch == 'A' ? 'T' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch == 'T' ? 'A' : ch == 'a' ? 't' : ch == 'c' ? 'g' : ch == 'g' ? 'c' : ch == 't' ? 'a' : ch == 'n' ? 'n' : 'N';
              }
            }
            if (strand[a]) rec.read = rcRead.c_str();
            if (pairwise) {
              rec.blocks = &blocks[3 * b0];
              rec.strand_read = strand[a] ? rcRead.c_str() : reads[r];
              rec.chrom_text = h->segText[a].data() - h->segStart[a];     // chrom_text[tPos] for the covered tPos only
            }
            recs.push_back(rec);
          }
          seg_off.push_back((int32_t)recs.size());
        }
      }
      uint64_t need = 0;
      if (hi && !unaligned && recs.empty()) need = 0;                     // OUTPUT prints nothing: groups exist, the first has no segment, read.unaligned == 0 (Mapping_ultility.h:467-492)
      else if (unaligned || recs.empty()) {
        lra_aln_record un; memset(&un, 0, sizeof un);
        un.read_name = names[r]; un.read = reads[r]; un.qual = quals ? quals[r] : nullptr; un.read_len = read_len[r];
        const size_t before = text.size();
        if ((rc = lra_output_read_str(nullptr, nullptr, 0, nullptr, o->PrintNumAln, (char)o->printFormat, o->hardClip, passthrough, 1, &un, text))) break;
        need = text.size() - before;
      } else {
        const int n = (int)seg_off.size() - 1;
        groups.assign(n, lra_aln_group()); index.assign(n, 0);
        if ((rc = lra_group_alignments(recs.data(), seg_off.data(), n, groups.data())) || (rc = lra_order_alignments(groups.data(), n, recs.data(), index.data(), 0)) ||
            (rc = lra_simple_mapqv(groups.data(), index.data(), n, recs.data(), o->bypassClustering, o->readType == LRA_READ_CLR, o->readType == LRA_READ_ONT,
                                   (hi && !sparseRead) ? o->globalK : o->localK)))                         // SimpleMapQV(alignmentsOrder, read, smallOpts): smallOpts.globalK = glIndex.k (Map_lowacc.h:233, :610); = opts.globalK on the high-accuracy path (Map_highacc.h:402, :736)
          break;
        // (the records' text goes straight into the thread's part, written once: the sizing-then-filling calls of the C entry points formatted every record four times)
        const size_t before = text.size();
        if ((rc = lra_output_read_str(groups.data(), index.data(), n, recs.data(), o->PrintNumAln, (char)o->printFormat, o->hardClip, passthrough, 0, nullptr, text))) break;
        need = text.size() - before;
      }
      plen[tix].push_back(need);
    }
    prc[tix] = rc;
  };
  const bool rdbg = getenv("LRA_RECORD_DBG") != nullptr;
  auto wallr = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double tr0 = wallr();
  if (T == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
  if (rdbg) fprintf(stderr, "[records] %d threads: per-read work %.0f ms\n", T, wallr() - tr0);
  for (int t = 0; t < T; t++) if (prc[t]) return prc[t];
  // the ranges' texts joined in read order: every thread copies its own part to its place (the pages of the joined text are first touched by 256 threads, not one)
  lra_text_buf& out = h->text;
  std::vector<size_t> pstart((size_t)T + 1, 0);
  for (int t = 0; t < T; t++) pstart[t + 1] = pstart[t] + part[t].size();
  out.alloc(pstart[T]);
  if (pstart[T] && !out.data()) return LRA_ERR_NOMEM;
  h->rec_off.assign((size_t)n_reads + 1, 0);
  uint64_t at = 0; size_t r = 0;
  for (int t = 0; t < T; t++) for (uint64_t l : plen[t]) { h->rec_off[r++] = at; at += l; }
  h->rec_off[n_reads] = at;
  {
    auto copy = [&](int t) { if (!part[t].empty()) memcpy(out.data() + pstart[t], part[t].data(), part[t].size()); part_give(part[t]); };
    if (T == 1) copy(0);
    else { std::vector<std::thread> th; for (int t = 0; t < T; t++) th.emplace_back(copy, t); for (auto& x : th) x.join(); }
  }
  *len = out.size();
  if (text) *text = out.data();
  if (rec_off) *rec_off = h->rec_off.data();
  if (rdbg) fprintf(stderr, "[records] joined at %.0f ms (%.2f GB)\n", wallr() - tr0, out.size() / 1e9);
  return LRA_OK;
}

extern "C" int lra_map_records(lra_ctx* ctx, const lra_map_result* res, const lra_map_opts* o, const char* const* names, const char* const* reads,
                               const char* const* quals, const int32_t* read_len, const char* const* chrom_names, const char* passthrough, char* out,
                               uint64_t cap, uint64_t* len, uint64_t* rec_off) {
  if (!ctx || !res || !o || !names || !reads || !read_len || !chrom_names || !len) return LRA_ERR_INVALID;
  lra_map_state* m = ctx->map;
  if (!m) return LRA_ERR_INVALID;
  // two-call convention: the sizing call keeps its text, the filling call for the same result and format hands it over
  const lra_map_sig sig{res->d_blocks, res->d_runs, res->n_reads, res->n_alignments, o->printFormat, o->PrintNumAln, o->hardClip, passthrough, o->flagged_unaligned, res->d_read_status};
  if (out && m->last_sig == sig && !m->last_text.empty() && cap >= m->last_text.size()) {
    memcpy(out, m->last_text.data(), m->last_text.size());
    *len = m->last_text.size();
    if (rec_off) memcpy(rec_off, m->last_off.data(), m->last_off.size() * 8);
    m->last_text.clear(); m->last_sig = lra_map_sig{};
    return LRA_OK;
  }
  lra_map_host* h = nullptr;
  int rc = lra_map_snapshot(ctx, res, o->printFormat == 'a', &h);
  if (rc) return rc;
  const char* text = nullptr; const uint64_t* ro = nullptr;
  rc = lra_map_records_host(h, o, names, reads, quals, read_len, chrom_names, passthrough, 0, &text, len, &ro);
  if (rc) { lra_map_host_free(h); return rc; }
  if (rec_off) memcpy(rec_off, ro, ((size_t)res->n_reads + 1) * 8);
  if (!out) {                                                            // sizing call: remember text and offsets
    m->last_text.swap(h->text); m->last_off.swap(h->rec_off); m->last_sig = sig;
    lra_map_host_free(h);
    return LRA_OK;
  }
  m->last_sig = lra_map_sig{};
  if (cap < *len) { lra_map_host_free(h); return LRA_ERR_INVALID; }
  memcpy(out, text, *len);
  lra_map_host_free(h);
  return LRA_OK;
}
