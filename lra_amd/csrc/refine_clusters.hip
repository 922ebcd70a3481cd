// lra_amd/csrc/refine_clusters.hip -- SURVEY §8a row a10 (high-accuracy path): REFINEclusters (ClusterRefine.h:50-240, called at
// Map_highacc.h:429-447) for every cluster of a batch.  gfx950 only.
//   Cluster::CHROMIndex   Clustering.h:326-336      SwapStrand   ClusterRefine.h:24-31      CartesianTargetSort / bounds   Sorting.h:183-221
//   LocalIndex::LookupIndex   MMIndex.h:175-190     AppendValues   TupleOps.h:159-195       SetClusterBoundariesFromMatches   Clustering.h:308
// Mapping (the cluster-wise twin of refine_splitchain.hip).  One wave per cluster brings its matches to chromosome coordinates on its own
// strand, packs them as (t, q) keys and reduces the diagonal range; one segmented radix sort orders every cluster (CartesianTargetSort:
// only identical matches tie).  One lane per cluster then walks the genome local-index windows under the cluster: the two bounds on the
// sorted keys give the read span of the window, every (read window, genome window) it meets becomes a task (count, then emit).
// lra_local_compare_batch intersects all tasks; one lane per task applies AppendValues' test (count, then emit); one wave per cluster
// swaps reverse results back, reduces the box and refineEffiency.
#include "common.h"
#include "append_values.h"
#include "scan.h"
#include <rocprim/rocprim.hpp>
#include <algorithm>

namespace {

struct RclArgs {
  uint64_t nc; int n_reads;
  const uint64_t* cluster_off; const uint64_t* c_start; const uint32_t* c_count; const int32_t* c_strand;
  const uint32_t* bqs; const uint32_t* bqe; const uint32_t* bts; const uint32_t* bte;
  const uint32_t* mq; const uint32_t* mt;
  const uint64_t* read_off; const uint64_t* pos; int npos;
  const uint64_t* qWinOff; const uint64_t* qBnd; const uint64_t* gSeqOff; uint64_t nWg; const uint64_t* gBnd;
  int window, smallK, K, lwin;
  uint32_t* cRead; int32_t* chrom; uint32_t* status; int64_t* maxD; int64_t* minD; uint32_t* qS; uint32_t* qE;   // per cluster
  uint64_t* key; const uint64_t* skey;                                                                          // per match (c_start layout)
  uint32_t* taskCnt; const uint64_t* taskOff;
  uint64_t* qLo; uint64_t* qHi; uint64_t* tLo; uint64_t* tHi; uint32_t* qAdd; uint32_t* tAdd; int64_t* mx; int64_t* mn; uint32_t* tbox;
};

__device__ int hfind(const uint64_t* pos, int npos, uint64_t query, bool& ub) {      // Genome.h:20-32
  if (npos > 0 && query == pos[0]) return 0;
  int lo = 0, cnt = npos;
  while (cnt > 0) { const int step = cnt >> 1; if (pos[lo + step] < query) { lo += step + 1; cnt -= step + 1; } else cnt = step; }
  if (lo == npos) { ub = true; return lo - 1; }
  if (query == pos[lo]) return lo;
  return lo - 1;
}
__device__ long lookup(const uint64_t* so, long n, uint64_t pos, bool& ub) {         // MMIndex.h:175-190
  long lo = 0, cnt = n;
  while (cnt > 0) { const long step = cnt >> 1; if (so[lo + step] < pos) { lo += step + 1; cnt -= step + 1; } else cnt = step; }
  if (lo == n) { ub = true; return lo - 1; }
  if (so[lo] != pos) return lo - 1;
  return lo;
}

__global__ void rcl_ends(uint64_t nc, const uint64_t* __restrict__ s, const uint32_t* __restrict__ n, uint64_t* e) {
  const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c < nc) e[c] = s[c] + n[c];
}

__global__ void rcl_reads(RclArgs a) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < a.n_reads) for (uint64_t c = a.cluster_off[r]; c < a.cluster_off[r + 1]; c++) a.cRead[c] = r;
}

// :58-83: CHROMIndex, chromosome coordinates, SwapStrand, diagonal range, (t, q) keys.  One wave per cluster.
__global__ void __launch_bounds__(64) rcl_prep(RclArgs a) {
  const int lane = threadIdx.x;
  for (uint64_t c = blockIdx.x; c < a.nc; c += gridDim.x) {
    const uint32_t n = a.c_count[c];
    if (lane == 0) { a.status[c] = 0; a.chrom[c] = 0; }
    if (n == 0) continue;
    bool ub = false;
    const int c0 = hfind(a.pos, a.npos, (uint64_t)a.bts[c] + 1, ub), c1 = hfind(a.pos, a.npos, a.bte[c], ub);
    if (ub) { if (lane == 0) a.status[c] = LRA_ST_OOB_SLOT; continue; }
    if (c0 != c1) { if (lane == 0) a.status[c] = LRA_ST_REJECTED; continue; }       // pass == 1: the cluster is cleared (:61-65)
    if (c1 + 1 >= a.npos) { if (lane == 0) a.status[c] = LRA_ST_OOB_SLOT; continue; }   // GetNextOffset past the table
    const uint32_t r = a.cRead[c];
    const uint32_t readLen = (uint32_t)(a.read_off[r + 1] - a.read_off[r]);
    const uint32_t coff = (uint32_t)a.pos[c0];
    const int strand = a.c_strand[c] != 0;
    const uint64_t b = a.c_start[c];
    int64_t mxd = INT64_MIN, mnd = INT64_MAX;
    for (uint32_t i = lane; i < n; i += 64) {
      uint32_t q = a.mq[b + i]; const uint32_t t = a.mt[b + i] - coff;
      if (strand) q = readLen - (q + (uint32_t)a.K);
      const int64_t d = (int64_t)t - (int64_t)q;
      mxd = max(mxd, d); mnd = min(mnd, d);
      a.key[b + i] = ((uint64_t)t << 32) | q;
    }
    for (int o = 32; o > 0; o >>= 1) { mxd = max(mxd, (int64_t)__shfl_xor((long long)mxd, o)); mnd = min(mnd, (int64_t)__shfl_xor((long long)mnd, o)); }
    if (lane == 0) {
      a.chrom[c] = c0; a.maxD[c] = mxd + 100; a.minD[c] = mnd - 100;
      a.qS[c] = strand ? readLen - a.bqe[c] : a.bqs[c]; a.qE[c] = strand ? readLen - a.bqs[c] : a.bqe[c];
    }
  }
}

// :84-177, one lane per cluster
template <bool EMIT>
__global__ void rcl_tasks(RclArgs a) {
  const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= a.nc) return;
  if (!EMIT) a.taskCnt[c] = 0;
  const uint32_t n = a.c_count[c];
  if (n == 0 || a.status[c]) return;
  const uint32_t r = a.cRead[c];
  const uint32_t readLen = (uint32_t)(a.read_off[r + 1] - a.read_off[r]);
  const int strand = a.c_strand[c] != 0;
  const int ci = a.chrom[c];
  const uint32_t chromOffset = (uint32_t)a.pos[ci], chromEndOffset = (uint32_t)a.pos[ci + 1];   // GetNextOffset(tEnd) = pos[Find(tEnd) + 1], Find(tEnd) = chromIndex
  const uint32_t tStart = a.bts[c], tEnd = a.bte[c];
  const uint64_t* K = a.skey + a.c_start[c];
  uint32_t wts, wte;
  if (chromOffset + (uint32_t)a.window > tStart) wts = chromOffset; else wts = tStart - a.window;
  if (tEnd + (uint32_t)a.window > chromEndOffset) wte = chromEndOffset - 1; else wte = tEnd + a.window;
  bool ub = false;
  const long ls = lookup(a.gSeqOff, (long)a.nWg + 1, wts, ub), le = lookup(a.gSeqOff, (long)a.nWg + 1, wte, ub);
  if (ub) { if (!EMIT) a.status[c] = LRA_ST_OOB_SLOT; return; }
  const long nWq = ((long)readLen + a.lwin - 1) / a.lwin;
  const uint64_t w0 = a.qWinOff[(strand ? a.n_reads : 0) + r];
  uint32_t ntask = 0;
  uint64_t to = EMIT ? a.taskOff[c] : 0;
  for (long lsi = ls; lsi <= le; lsi++) {
    if (lsi + 1 > (long)a.nWg) { if (!EMIT) { a.status[c] = LRA_ST_OOB_SLOT; a.taskCnt[c] = 0; } return; }
    if (a.gSeqOff[lsi] < chromOffset || a.gSeqOff[lsi + 1] < chromOffset) continue;
    const uint32_t gStart = (uint32_t)(a.gSeqOff[lsi] - chromOffset), gEnd = (uint32_t)(a.gSeqOff[lsi + 1] - 1 - chromOffset);
    if (gStart >= gEnd) continue;
    // CartesianTargetLowerBound(gStart) / UpperBound(gEnd) with first.pos = 0 in the query (Sorting.h:209-221): plain bounds on the packed keys
    uint32_t lo = 0, cnt = n;
    const uint64_t k0 = (uint64_t)gStart << 32;
    while (cnt > 0) { const uint32_t s = cnt >> 1; if (K[lo + s] < k0) { lo += s + 1; cnt -= s + 1; } else cnt = s; }
    const uint32_t matchStart = lo;
    const uint64_t k1 = (uint64_t)gEnd << 32;
    cnt = n - matchStart;
    while (cnt > 0) { const uint32_t s = cnt >> 1; if (K[lo + s] <= k1) { lo += s + 1; cnt -= s + 1; } else cnt = s; }
    uint32_t matchEnd = lo;
    if (matchEnd == n) matchEnd--;
    if (matchStart >= n) continue;
    uint32_t readStart = (uint32_t)K[matchStart], readEnd = (uint32_t)K[matchEnd];
    if (readStart == readEnd) { if (lsi > ls && readStart > 0) readStart = 0; }      // prev_readEnd is 0 here (:123, :128-132)
    if (lsi == ls) { if (readStart < (uint32_t)a.window) readStart = 0; else readStart -= a.window; }
    if (lsi == le) { if (readEnd + (uint32_t)a.window > readLen) readEnd = readLen; else readEnd += a.window; }
    if (readStart > readEnd) continue;
    if (readStart > readLen) { if (!EMIT) { a.status[c] = LRA_ST_OOB_SLOT; a.taskCnt[c] = 0; } return; }
    const long qi0 = readStart == readLen ? nWq : (long)(readStart / (uint32_t)a.lwin);
    const long qi1 = (long)(min(readEnd, readLen - 1) / (uint32_t)a.lwin);
    for (long qi = qi0; qi <= qi1; ++qi) {
      if (EMIT && !a.status[c]) {
        a.qLo[to] = a.qBnd[w0 + qi]; a.qHi[to] = a.qBnd[w0 + qi + 1]; a.tLo[to] = a.gBnd[lsi]; a.tHi[to] = a.gBnd[lsi + 1];
        a.qAdd[to] = (uint32_t)min((long)readLen, qi * (long)a.lwin); a.tAdd[to] = gStart; a.mx[to] = a.maxD[c]; a.mn[to] = a.minD[c];
        a.tbox[4 * to] = a.qS[c]; a.tbox[4 * to + 1] = a.qE[c]; a.tbox[4 * to + 2] = tStart - chromOffset; a.tbox[4 * to + 3] = tEnd - chromOffset;
        to++;
      }
      ntask++;
    }
  }
  if (!EMIT) a.taskCnt[c] = ntask;
}

typedef AvArgs FArgs;                                                   // (append_values.h: a wave per 64 tasks reads their pairs 64 at a time)

struct FinArgs {
  uint64_t nc; int smallK; const uint32_t* cRead; const int32_t* c_strand; const uint64_t* read_off; const uint64_t* taskOff; const uint64_t* outOff;
  uint64_t* matchOff; uint32_t* oq; const uint32_t* ot; uint32_t* box; float* eff;
};
__global__ void rcl_match_off(FinArgs a) {
  const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c <= a.nc) a.matchOff[c] = a.outOff[a.taskOff[c]];
}
__global__ void __launch_bounds__(64) rcl_finish(FinArgs a) {            // :229-237
  const int lane = threadIdx.x;
  for (uint64_t c = blockIdx.x; c < a.nc; c += gridDim.x) {
    const uint64_t m0 = a.matchOff[c], m1 = a.matchOff[c + 1];
    if (m1 == m0) { if (lane == 0) { a.box[4 * c] = a.box[4 * c + 1] = a.box[4 * c + 2] = a.box[4 * c + 3] = 0; a.eff[c] = 0; } continue; }
    const uint32_t r = a.cRead[c];
    const uint32_t readLen = (uint32_t)(a.read_off[r + 1] - a.read_off[r]);
    const bool rev = a.c_strand[c] != 0;
    uint32_t qmin = 0xFFFFFFFFu, qmax = 0, tmin = 0xFFFFFFFFu, tmax = 0;
    for (uint64_t i = m0 + lane; i < m1; i += 64) {
      uint32_t q = a.oq[i];
      if (rev) { q = readLen - (q + (uint32_t)a.smallK); a.oq[i] = q; }
      const uint32_t t = a.ot[i];
      qmin = min(qmin, q); qmax = max(qmax, q + (uint32_t)a.smallK); tmin = min(tmin, t); tmax = max(tmax, t + (uint32_t)a.smallK);
    }
    for (int o = 32; o > 0; o >>= 1) {
      qmin = min(qmin, __shfl_xor(qmin, o)); qmax = max(qmax, __shfl_xor(qmax, o)); tmin = min(tmin, __shfl_xor(tmin, o)); tmax = max(tmax, __shfl_xor(tmax, o));
    }
    if (lane == 0) {
      a.box[4 * c] = qmin; a.box[4 * c + 1] = qmax; a.box[4 * c + 2] = tmin; a.box[4 * c + 3] = tmax;
      a.eff[c] = ((float)(m1 - m0)) / (float)min(qmax - qmin, tmax - tmin);
    }
  }
}

inline size_t sz(size_t n, size_t e) { return (n * e + 255) / 256 * 256; }

}  // namespace

extern "C" int lra_refine_clusters_batch(lra_ctx* ctx, int n_reads, const uint64_t* d_cluster_off, const uint64_t* d_c_start, const uint32_t* d_c_count,
                                         const int32_t* d_c_strand, const uint32_t* d_qs, const uint32_t* d_qe, const uint32_t* d_ts, const uint32_t* d_te,
                                         const uint32_t* d_q, const uint32_t* d_t, uint64_t n_matches_cap, const uint64_t* d_read_off,
                                         const uint64_t* h_chrom_pos, int n_chrom, const lra_local_index_result* read_index, uint64_t n_g_windows,
                                         const uint64_t* d_g_seq_off, const uint64_t* d_g_tuple_bnd, const uint32_t* d_g_tuples, const lra_rsc_opts* opts,
                                         lra_refined_clusters_result* out) {
  if (!ctx || !out || !opts || !h_chrom_pos || n_chrom < 1 || !read_index || n_reads < 0) return LRA_ERR_INVALID;
  if (opts->local_window <= 0) return lra_set_err(ctx, LRA_ERR_INVALID, "local_window must be positive");
  if (read_index->n_seqs != 2 * n_reads) return lra_set_err(ctx, LRA_ERR_INVALID, "read_index must hold the reads forward, then reverse-complemented");
  memset(out, 0, sizeof *out);
  if (n_reads == 0) return LRA_OK;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  uint64_t NC = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&NC, d_cluster_off + n_reads, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  out->n_clusters = NC;
  if (NC == 0) return LRA_OK;
  const uint64_t NM = n_matches_cap;
  const size_t npos = (size_t)n_chrom + 1;
  auto take = [](char*& p, size_t n, size_t e) { char* r = p; p += sz(n, e); return r; };
  char* w = (char*)lra_ensure(ctx, 27, sz(NC + 2, 4) * 6 + sz(NC + 2, 8) * 4 + sz(4 * NC + 4, 4) + sz(NC + 1, 4) + sz(npos, 8) + sz(NM + 1, 8) * 2 + 4096);
  if (!w) return LRA_ERR_NOMEM;
  RclArgs a;
  memset(&a, 0, sizeof a);
  a.nc = NC; a.n_reads = n_reads; a.cluster_off = d_cluster_off; a.c_start = d_c_start; a.c_count = d_c_count; a.c_strand = d_c_strand;
  a.bqs = d_qs; a.bqe = d_qe; a.bts = d_ts; a.bte = d_te; a.mq = d_q; a.mt = d_t; a.read_off = d_read_off;
  a.qWinOff = read_index->d_win_off; a.qBnd = read_index->d_tuple_bnd; a.gSeqOff = d_g_seq_off; a.nWg = n_g_windows; a.gBnd = d_g_tuple_bnd;
  a.window = opts->window; a.smallK = opts->smallK; a.K = opts->K; a.lwin = opts->local_window;
  a.cRead = (uint32_t*)take(w, NC + 2, 4); a.chrom = (int32_t*)take(w, NC + 2, 4); a.status = (uint32_t*)take(w, NC + 2, 4); a.qS = (uint32_t*)take(w, NC + 2, 4);
  a.qE = (uint32_t*)take(w, NC + 2, 4); a.taskCnt = (uint32_t*)take(w, NC + 2, 4);
  a.maxD = (int64_t*)take(w, NC + 2, 8); a.minD = (int64_t*)take(w, NC + 2, 8);
  uint64_t* taskOff = (uint64_t*)take(w, NC + 2, 8); uint64_t* matchOff = (uint64_t*)take(w, NC + 2, 8);
  uint32_t* box = (uint32_t*)take(w, 4 * NC + 4, 4); float* eff = (float*)take(w, NC + 1, 4);
  uint64_t* dpos = (uint64_t*)take(w, npos, 8);
  uint64_t* key = (uint64_t*)take(w, NM + 1, 8); uint64_t* skey = (uint64_t*)take(w, NM + 1, 8);
  a.pos = dpos; a.npos = (int)npos; a.key = key; a.skey = skey; a.taskOff = taskOff;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(dpos, h_chrom_pos, npos * 8, hipMemcpyHostToDevice, st));
  LRA_HIP_CHECK(ctx, hipMemsetAsync(a.taskCnt, 0, (NC + 2) * 4, st));
  const unsigned gc = (unsigned)((NC + 255) / 256), gw = (unsigned)std::min<uint64_t>(NC, (uint64_t)ctx->num_cu * 32);
  lra_time_begin(ctx, "rcl_tasks");
  hipLaunchKernelGGL(rcl_reads, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, st, a);
  hipLaunchKernelGGL(rcl_prep, dim3(gw), dim3(64), 0, st, a);
  lra_time_end(ctx);
  if (NM > 0) {                                                           // CartesianTargetSort: segments = the clusters' match ranges
    uint64_t* cend = matchOff;                                            // (free until the end) segment ends = c_start + c_count
    hipLaunchKernelGGL(rcl_ends, dim3(gc), dim3(256), 0, st, NC, d_c_start, d_c_count, cend);
    size_t temp_bytes = 0;
    (void)rocprim::segmented_radix_sort_keys(nullptr, temp_bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (unsigned int)NM, (unsigned int)NC, (uint64_t*)nullptr,
                                             (uint64_t*)nullptr, 0, 64, st);
    void* temp = lra_scratch(ctx, 2, temp_bytes + 256);
    if (!temp) return LRA_ERR_NOMEM;
    lra_time_begin(ctx, "rcl_tasks");
    hipError_t e = rocprim::segmented_radix_sort_keys(temp, temp_bytes, key, skey, (unsigned int)NM, (unsigned int)NC, d_c_start, (const uint64_t*)cend, 0, 64, st);
    lra_time_end(ctx);
    if (e != hipSuccess) return lra_set_err(ctx, LRA_ERR_HIP, "segmented sort: %s", hipGetErrorString(e));
  }
  lra_time_begin(ctx, "rcl_tasks");
  hipLaunchKernelGGL(rcl_tasks<false>, dim3(gc), dim3(256), 0, st, a);
  lra_time_end(ctx);
  { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)NC + 1, a.taskCnt, taskOff); if (rc) return rc; }
  uint64_t NT = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&NT, taskOff + NC + 1, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  out->n_tasks = NT;
  char* wt = (char*)lra_ensure(ctx, 28, sz(NT + 2, 8) * 7 + sz(NT + 1, 4) * 3 + sz(4 * NT + 4, 4) + 4096);
  if (!wt) return LRA_ERR_NOMEM;
  a.qLo = (uint64_t*)take(wt, NT + 2, 8); a.qHi = (uint64_t*)take(wt, NT + 2, 8); a.tLo = (uint64_t*)take(wt, NT + 2, 8); a.tHi = (uint64_t*)take(wt, NT + 2, 8);
  a.mx = (int64_t*)take(wt, NT + 2, 8); a.mn = (int64_t*)take(wt, NT + 2, 8); uint64_t* outOff = (uint64_t*)take(wt, NT + 2, 8);
  a.qAdd = (uint32_t*)take(wt, NT + 1, 4); a.tAdd = (uint32_t*)take(wt, NT + 1, 4); uint32_t* passCnt = (uint32_t*)take(wt, NT + 1, 4);
  a.tbox = (uint32_t*)take(wt, 4 * NT + 4, 4);
  uint64_t NMo = 0;
  uint32_t* oq = nullptr; uint32_t* ot = nullptr;
  if (NT > 0) {
    lra_time_begin(ctx, "rcl_tasks");
    hipLaunchKernelGGL(rcl_tasks<true>, dim3(gc), dim3(256), 0, st, a);
    lra_time_end(ctx);
    lra_local_pairs_result pr;
    { int rc = lra_local_compare_batch(ctx, NT, read_index->d_tuples, a.qLo, a.qHi, d_g_tuples, a.tLo, a.tHi, opts->max_freq, nullptr, nullptr, &pr); if (rc) return rc; }
    out->n_pairs = pr.n_pairs;
    FArgs fa;
    memset(&fa, 0, sizeof fa);
    fa.n_tasks = NT; fa.pairOff = pr.d_pair_off; fa.pqi = pr.d_pair_qi; fa.pti = pr.d_pair_ti; fa.qTup = read_index->d_tuples; fa.gTup = d_g_tuples;
    fa.qAdd = a.qAdd; fa.tAdd = a.tAdd; fa.mx = a.mx; fa.mn = a.mn; fa.tbox = a.tbox; fa.cnt = passCnt; fa.outOff = outOff;
    const unsigned gt = (unsigned)((NT + 63) / 64);
    lra_time_begin(ctx, "rcl_filter");
    hipLaunchKernelGGL(av_filter<false>, dim3(gt), dim3(64), 0, st, fa);
    lra_time_end(ctx);
    { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)NT, passCnt, outOff); if (rc) return rc; }
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(&NMo, outOff + NT, 8, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
    char* wm = (char*)lra_ensure(ctx, 29, sz(NMo + 1, 4) * 2 + 1024);
    if (!wm) return LRA_ERR_NOMEM;
    oq = (uint32_t*)take(wm, NMo + 1, 4); ot = (uint32_t*)take(wm, NMo + 1, 4);
    fa.oq = oq; fa.ot = ot;
    lra_time_begin(ctx, "rcl_filter");
    hipLaunchKernelGGL(av_filter<true>, dim3(gt), dim3(64), 0, st, fa);
    lra_time_end(ctx);
  } else {
    LRA_HIP_CHECK(ctx, hipMemsetAsync(outOff, 0, 16, st));
  }
  FinArgs f;
  memset(&f, 0, sizeof f);
  f.nc = NC; f.smallK = opts->smallK; f.cRead = a.cRead; f.c_strand = d_c_strand; f.read_off = d_read_off; f.taskOff = taskOff; f.outOff = outOff;
  f.matchOff = matchOff; f.oq = oq; f.ot = ot; f.box = box; f.eff = eff;
  lra_time_begin(ctx, "rcl_filter");
  hipLaunchKernelGGL(rcl_match_off, dim3((unsigned)((NC + 256) / 256)), dim3(256), 0, st, f);
  hipLaunchKernelGGL(rcl_finish, dim3(gw), dim3(64), 0, st, f);
  lra_time_end(ctx);
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  LRA_HIP_CHECK(ctx, hipGetLastError());
  out->n_matches = NMo; out->d_match_off = matchOff; out->d_match_q = oq; out->d_match_t = ot; out->d_box = box; out->d_eff = eff; out->d_status = a.status;
  out->d_chrom = a.chrom;
  return LRA_OK;
}
