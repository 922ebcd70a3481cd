// lra_amd/csrc/split_clusters.hip -- SURVEY §8a row a6 (high-accuracy path, Map_highacc.h:153-155), for all reads of a batch.  gfx950 only.
//   IntervalSet (box diagonal, mixed q/t comparator)   SplitClusters.h:18-60
//   SplitClusters                                        SplitClusters.h:63-171
//   DecideSplitClustersValue                             SplitClusters.h:176-249  (CartesianLowerBound Sorting.h:171-178)
// Mapping.  The two std::set<GenomePos> of a read become two sorted, de-duplicated coordinate lists (segmented radix sort + one wave per
// read compacting in place).  Then one lane per cluster: count the cut coordinates strictly inside its box (two binary searches per
// axis), gather them, order them with the reference's comparator by a faithful libstdc++ std::sort (the comparator only ties a q cut
// with a t cut on the same point of the diagonal, and the permutation std::sort leaves among ties decides which cut is taken first),
// walk the cuts and write the pieces.  The pieces of a read are laid out the way the reference pushes them: whole (unsplit, contig
// reads only) clusters first, then the pieces cluster by cluster.  Values and anchor counts have closed forms per piece (see
// sc_values).  All double arithmetic is spelled with the *_rn intrinsics: the host compiler of the reference does not contract
// a*b+c into an fma, hipcc would.
// Algorithmic bytes: 28 B per cluster in, 32 B per piece out, 4 B per match (value pass) -- far below every other stage.
#include "common.h"
#include "scan.h"
#include "std_sort.h"
#include <rocprim/rocprim.hpp>
#include <vector>

namespace {

constexpr uint32_t SENT = 0xFFFFFFFFu;

struct ScArgs {
  uint64_t nc; int n_reads;
  const uint64_t* cluster_off;
  const uint32_t* qs; const uint32_t* qe; const uint32_t* ts; const uint32_t* te; const int32_t* strand; const float* anchorfreq;
  const uint64_t* match_off; const uint32_t* match_q;
  int contig, K;
  uint32_t* clusRead; uint8_t* split;
  uint32_t* qk; uint32_t* tk; uint64_t* off2; uint32_t* nq; uint32_t* nt;             // per read: unique counts
  uint32_t* cutCnt; const uint64_t* cutOff; uint64_t* cuts;                            // per cluster cut list (isT << 32 | coordinate)
  uint32_t* tmp;                                                                       // pieces before layout, 4 words each, at cutOff[c] + c
  uint32_t* cntU; uint32_t* cntP; const uint64_t* U; const uint64_t* P;
  uint64_t* splitOff;
  uint32_t* oqs; uint32_t* oqe; uint32_t* ots; uint32_t* ote; int32_t* ostrand; int32_t* ocoarse; int32_t* oval; int32_t* onum; uint32_t* oread;
  int32_t* clusterVal;
};

__global__ void sc_read_off(ScArgs a) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > a.n_reads) return;
  const uint64_t c0 = a.cluster_off[r];
  a.splitOff[r] = a.U[c0] + a.P[c0];
}

__global__ void sc_reads(ScArgs a) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > a.n_reads) return;
  a.off2[r] = 2 * a.cluster_off[r];
  if (r < a.n_reads) for (uint64_t c = a.cluster_off[r]; c < a.cluster_off[r + 1]; c++) a.clusRead[c] = r;
}

// SplitClusters.h:69-98: which clusters are cut; their box corners go into the read's coordinate lists
__global__ void sc_coords(ScArgs a) {
  const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= a.nc) return;
  const uint32_t qs = a.qs[c], qe = a.qe[c], ts = a.ts[c], te = a.te[c];
  const float af = a.anchorfreq[c];
  bool sp = true;
  if (a.contig) sp = af <= 3.0f || (af <= 5.0f && max(te - ts, qe - qs) <= 2000u);
  a.split[c] = sp;
  a.cntU[c] = sp ? 0 : 1;
  a.qk[2 * c] = sp ? qs : SENT; a.qk[2 * c + 1] = sp ? qe : SENT;
  a.tk[2 * c] = sp ? ts : SENT; a.tk[2 * c + 1] = sp ? te : SENT;
}

// one wave per (read, axis): sorted list -> sorted unique list in place (the std::set)
__global__ void __launch_bounds__(64) sc_unique(ScArgs a) {
  const int r = blockIdx.x >> 1, axis = blockIdx.x & 1;
  uint32_t* v = (axis ? a.tk : a.qk) + a.off2[r];
  const uint32_t n = (uint32_t)(a.off2[r + 1] - a.off2[r]);
  const int lane = threadIdx.x;
  uint32_t outN = 0, prevLast = SENT;
  for (uint32_t base = 0; base < n; base += 64) {
    const uint32_t i = base + lane;
    const uint32_t x = i < n ? v[i] : SENT;
    uint32_t p = __shfl_up(x, 1);
    if (lane == 0) p = prevLast;
    const bool keep = i < n && x != SENT && (i == 0 || x != p);
    const uint64_t m = __ballot(keep);
    const uint32_t pos = outN + __popcll(m & ((1ull << lane) - 1));
    prevLast = __shfl(x, 63);
    if (keep) v[pos] = x;                                                // pos <= i, and every lane has read its element already
    outN += __popcll(m);
  }
  if (lane == 0) (axis ? a.nt : a.nq)[r] = outN;
}

__device__ __forceinline__ uint32_t lb(const uint32_t* v, uint32_t n, uint32_t x) {    // first index with v[i] >= x
  uint32_t lo = 0, cnt = n;
  while (cnt > 0) { const uint32_t s = cnt >> 1; if (v[lo + s] < x) { lo += s + 1; cnt -= s + 1; } else cnt = s; }
  return lo;
}
__device__ __forceinline__ uint32_t ub(const uint32_t* v, uint32_t n, uint32_t x) {    // first index with v[i] > x
  uint32_t lo = 0, cnt = n;
  while (cnt > 0) { const uint32_t s = cnt >> 1; if (v[lo + s] <= x) { lo += s + 1; cnt -= s + 1; } else cnt = s; }
  return lo;
}

__global__ void sc_count(ScArgs a) {
  const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= a.nc) return;
  uint32_t cnt = 0;
  if (a.split[c]) {
    const uint32_t r = a.clusRead[c];
    const uint32_t* Q = a.qk + a.off2[r]; const uint32_t* T = a.tk + a.off2[r];
    const uint32_t nq = a.nq[r], nt = a.nt[r];
    const uint32_t q0 = ub(Q, nq, a.qs[c]), q1 = lb(Q, nq, a.qe[c]), t0 = ub(T, nt, a.ts[c]), t1 = lb(T, nt, a.te[c]);
    cnt = (q1 > q0 ? q1 - q0 : 0) + (t1 > t0 ? t1 - t0 : 0);
  }
  a.cutCnt[c] = cnt;
}

struct Line { double slope, intercept; bool strand; };

// IntervalSet::operator() SplitClusters.h:39-55 on packed cuts (bit 32: 1 = t coordinate)
struct CutLess {
  Line L;
  __device__ bool operator()(uint64_t a, uint64_t b) const {
    const bool at = a >> 32, bt = b >> 32;
    const uint32_t av = (uint32_t)a, bv = (uint32_t)b;
    if (at == bt && !at) return av < bv;
    else if (at == bt) return L.strand == 0 ? av < bv : av > bv;
    else if (!at && bt) {
      const double k = __dadd_rn(__dmul_rn((double)av, L.slope), L.intercept);
      return L.strand == 0 ? k < (double)bv : k > (double)bv;
    } else {
      const double k = __dadd_rn(__dmul_rn((double)bv, L.slope), L.intercept);
      return L.strand == 0 ? (double)av < k : (double)av > k;
    }
  }
};

__device__ __forceinline__ uint32_t to_pos(double x) { return (uint32_t)(long long)x; }   // what x86-64 emits for (GenomePos)double

// SplitClusters.h:103-170, one lane per cut cluster
__global__ void sc_emit(ScArgs a) {
  const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= a.nc) return;
  if (!a.split[c]) { a.cntP[c] = 0; return; }
  const uint32_t r = a.clusRead[c];
  const uint32_t qs = a.qs[c], qe = a.qe[c], ts = a.ts[c], te = a.te[c];
  const int strand = a.strand[c] != 0;
  Line L;
  {
    const long long dq = (long long)qe - (long long)qs;
    L.slope = __ddiv_rn((double)((long long)te - (long long)ts), (double)dq);
    if (strand == 0) L.intercept = __ddiv_rn((double)((long long)qe * (long long)ts - (long long)qs * (long long)te), (double)dq);
    else {
      L.slope = __dmul_rn(-1.0, L.slope);
      L.intercept = __ddiv_rn((double)((long long)qs * (long long)ts - (long long)qe * (long long)te), (double)((long long)qs - (long long)qe));
    }
    L.strand = strand;
  }
  uint64_t* cut = a.cuts + a.cutOff[c];
  uint32_t n = 0;
  {
    const uint32_t* Q = a.qk + a.off2[r]; const uint32_t* T = a.tk + a.off2[r];
    const uint32_t nq = a.nq[r], nt = a.nt[r];
    const uint32_t q0 = ub(Q, nq, qs), q1 = lb(Q, nq, qe), t0 = ub(T, nt, ts), t1 = lb(T, nt, te);
    for (uint32_t i = q0; i < q1; i++) cut[n++] = Q[i];
    for (uint32_t i = t0; i < t1; i++) cut[n++] = (1ull << 32) | T[i];
  }
  CutLess lt{L};
  lra_std_sort::std_sort(cut, (long)n, lt);
  uint32_t* out = a.tmp + 4 * (a.cutOff[c] + c);
  uint32_t np = 0;
  auto push = [&](uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1) { out[4 * np] = x0; out[4 * np + 1] = x1; out[4 * np + 2] = y0; out[4 * np + 3] = y1; np++; };
  uint32_t pq = qs, pt = strand == 0 ? ts : te;
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t v = (uint32_t)cut[i];
    if ((cut[i] >> 32) == 0) {                                            // cut on a q coordinate :127-139
      const uint32_t t = to_pos(ceil(__dadd_rn(__dmul_rn(L.slope, (double)v), L.intercept)));
      if (pq < v) {
        if (strand == 0 && v >= pq + 3 && t >= pt + 3) push(pq, v, pt, t);
        else if (strand == 1 && v >= pq + 3 && pt >= t + 3) push(pq, v, t, pt);
      } else continue;
      pq = v; pt = t;
    } else {                                                              // cut on a t coordinate :140-154
      const uint32_t q = to_pos(ceil(__ddiv_rn(__dsub_rn((double)v, L.intercept), L.slope)));
      if (pq < q) {
        if (strand == 0 && q >= pq + 3 && v >= pt + 3) push(pq, q, pt, v);
        else if (strand == 1 && q >= pq + 3 && pt >= v + 3) push(pq, q, v, pt);
      } else continue;
      pq = q; pt = v;
    }
  }
  if (pq < qe) {                                                          // :157-168
    if (strand == 0 && qe >= pq + 3 && te >= pt + 3) push(pq, qe, pt, te);
    else if (strand == 1 && qe >= pq + 3 && pt >= ts + 3) push(pq, qe, ts, pt);
  }
  a.cntP[c] = np;
}

// layout in push order: the read's whole clusters, then its pieces
__global__ void sc_layout(ScArgs a) {
  const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= a.nc) return;
  const uint32_t r = a.clusRead[c];
  const uint64_t c0 = a.cluster_off[r], c1 = a.cluster_off[r + 1];
  const uint64_t base = a.U[c0] + a.P[c0];
  const int32_t coarse = (int32_t)(c - c0);
  if (!a.split[c]) {
    const uint64_t o = base + (a.U[c] - a.U[c0]);
    a.oqs[o] = a.qs[c]; a.oqe[o] = a.qe[c]; a.ots[o] = a.ts[c]; a.ote[o] = a.te[c]; a.ostrand[o] = a.strand[c] != 0; a.ocoarse[o] = coarse; a.oread[o] = r;
    return;
  }
  const uint32_t np = a.cntP[c];
  const uint32_t* in = a.tmp + 4 * (a.cutOff[c] + c);
  uint64_t o = base + (a.U[c1] - a.U[c0]) + (a.P[c] - a.P[c0]);
  for (uint32_t k = 0; k < np; k++, o++) {
    a.oqs[o] = in[4 * k]; a.oqe[o] = in[4 * k + 1]; a.ots[o] = in[4 * k + 2]; a.ote[o] = in[4 * k + 3]; a.ostrand[o] = a.strand[c] != 0;
    a.ocoarse[o] = coarse; a.oread[o] = r;
  }
}

// clusters[m].Val: read bases covered by the cluster's matches (:179-194); left 0 when the read has no split cluster (:177)
__global__ void sc_cluster_val(ScArgs a) {
  const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= a.nc) return;
  const uint32_t r = a.clusRead[c];
  int32_t val = 0;
  if (a.splitOff[r + 1] > a.splitOff[r]) {
    const uint64_t b = a.match_off[c], e = a.match_off[c + 1];
    if (e > b) {
      uint32_t cur = a.match_q[b], mat = 0;
      for (uint64_t i = b; i < e; i++) {
        const uint32_t p = a.match_q[i];
        if (cur > p) mat += p + a.K - cur; else mat += a.K;
        cur = p + a.K;
      }
      val = (int32_t)mat;
    }
  }
  a.clusterVal[c] = val;
}

// per piece (:195-248): Val = (int)(Val[coarse] * min-side ratio); NumofAnchors0 = matches of the coarse cluster between this piece's
// qStart and the next piece's (the running matchS / matchE of the reference, written per piece)
__global__ void sc_values(ScArgs a, uint64_t ns) {
  const uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= ns) return;
  const uint32_t r = a.oread[m];
  const uint64_t s0 = a.splitOff[r], s1 = a.splitOff[r + 1];
  const uint64_t c0 = a.cluster_off[r];
  const int32_t ic = a.ocoarse[m];
  const uint64_t c = c0 + ic;
  const float pika = (float)min(a.oqe[m] - a.oqs[m], a.ote[m] - a.ots[m]) / (float)min(a.qe[c] - a.qs[c], a.te[c] - a.ts[c]);
  a.oval[m] = (int32_t)((float)a.clusterVal[c] * pika);
  const uint32_t* mq = a.match_q + a.match_off[c];
  const uint32_t sz = (uint32_t)(a.match_off[c + 1] - a.match_off[c]);
  const uint32_t S = (m > s0 && a.ocoarse[m - 1] == ic) ? lb(mq, sz, a.oqs[m]) : 0;
  const uint32_t E = (m + 1 < s1 && a.ocoarse[m + 1] == ic) ? lb(mq, sz, a.oqs[m + 1]) : sz;
  a.onum[m] = (int32_t)E - (int32_t)S;
}

inline size_t sz(size_t n, size_t e) { return (n * e + 255) / 256 * 256; }

}  // namespace

extern "C" int lra_split_clusters_batch(lra_ctx* ctx, int n_reads, const uint64_t* d_cluster_off, const uint32_t* d_qs, const uint32_t* d_qe,
                                        const uint32_t* d_ts, const uint32_t* d_te, const int32_t* d_strand, const float* d_anchorfreq,
                                        const uint64_t* d_match_off, const uint32_t* d_match_q, int contig, int K,
                                        lra_split_clusters_result* out) {
  if (!ctx || !out || n_reads < 0) return LRA_ERR_INVALID;
  memset(out, 0, sizeof *out);
  out->n_reads = n_reads;
  if (n_reads == 0) return LRA_OK;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t n1 = (size_t)n_reads + 1;
  uint64_t NC = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&NC, d_cluster_off + n_reads, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  out->n_clusters = NC;
  auto take = [](char*& p, size_t n, size_t e) { char* r = p; p += sz(n, e); return r; };
  char* w = (char*)lra_ensure(ctx, 24, sz(NC + 1, 4) * 6 + sz(NC + 1, 1) + sz(2 * NC + 2, 4) * 2 + sz(n1, 8) * 2 + sz(n1, 4) * 2 + sz(NC + 2, 8) * 3 + 4096);
  if (!w) return LRA_ERR_NOMEM;
  ScArgs a;
  memset(&a, 0, sizeof a);
  a.nc = NC; a.n_reads = n_reads; a.cluster_off = d_cluster_off; a.qs = d_qs; a.qe = d_qe; a.ts = d_ts; a.te = d_te; a.strand = d_strand;
  a.anchorfreq = d_anchorfreq; a.match_off = d_match_off; a.match_q = d_match_q; a.contig = contig; a.K = K;
  a.clusRead = (uint32_t*)take(w, NC + 1, 4); a.cutCnt = (uint32_t*)take(w, NC + 1, 4); a.cntU = (uint32_t*)take(w, NC + 1, 4);
  a.cntP = (uint32_t*)take(w, NC + 1, 4); a.clusterVal = (int32_t*)take(w, NC + 1, 4); (void)take(w, NC + 1, 4);
  a.split = (uint8_t*)take(w, NC + 1, 1);
  a.qk = (uint32_t*)take(w, 2 * NC + 2, 4); a.tk = (uint32_t*)take(w, 2 * NC + 2, 4);
  a.off2 = (uint64_t*)take(w, n1, 8); a.splitOff = (uint64_t*)take(w, n1, 8); a.nq = (uint32_t*)take(w, n1, 4); a.nt = (uint32_t*)take(w, n1, 4);
  uint64_t* cutOff = (uint64_t*)take(w, NC + 2, 8); uint64_t* U = (uint64_t*)take(w, NC + 2, 8); uint64_t* P = (uint64_t*)take(w, NC + 2, 8);
  a.cutOff = cutOff; a.U = U; a.P = P;
  out->d_split_off = a.splitOff; out->d_cluster_val = a.clusterVal; out->d_cluster_split = a.split;
  const unsigned gc = (unsigned)((NC + 255) / 256);
  hipLaunchKernelGGL(sc_reads, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, st, a);
  if (NC == 0) { LRA_HIP_CHECK(ctx, hipMemsetAsync(a.splitOff, 0, n1 * 8, st)); LRA_HIP_CHECK(ctx, hipStreamSynchronize(st)); return LRA_OK; }
  lra_time_begin(ctx, "split_clusters");
  hipLaunchKernelGGL(sc_coords, dim3(gc), dim3(256), 0, st, a);
  lra_time_end(ctx);
  {
    // the two coordinate sets: any sorted order serves, so a segmented radix sort; sorted into scratch and copied back
    size_t temp_bytes = 0;
    (void)rocprim::segmented_radix_sort_keys(nullptr, temp_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, (unsigned int)(2 * NC), (unsigned int)n_reads,
                                             (uint64_t*)nullptr, (uint64_t*)nullptr, 0, 32, st);
    char* tmp = (char*)lra_scratch(ctx, 2, temp_bytes + 256 + sz(2 * NC + 2, 4));
    if (!tmp) return LRA_ERR_NOMEM;
    uint32_t* sorted = (uint32_t*)tmp;
    void* temp = tmp + sz(2 * NC + 2, 4);
    lra_time_begin(ctx, "split_clusters");
    for (int axis = 0; axis < 2; axis++) {
      uint32_t* k = axis ? a.tk : a.qk;
      hipError_t e = rocprim::segmented_radix_sort_keys(temp, temp_bytes, k, sorted, (unsigned int)(2 * NC), (unsigned int)n_reads, a.off2, a.off2 + 1, 0,
                                                        32, st);
      if (e != hipSuccess) { lra_time_end(ctx); return lra_set_err(ctx, LRA_ERR_HIP, "segmented sort: %s", hipGetErrorString(e)); }
      LRA_HIP_CHECK(ctx, hipMemcpyAsync(k, sorted, 2 * NC * 4, hipMemcpyDeviceToDevice, st));
    }
    hipLaunchKernelGGL(sc_unique, dim3(2 * (unsigned)n_reads), dim3(64), 0, st, a);
    hipLaunchKernelGGL(sc_count, dim3(gc), dim3(256), 0, st, a);
    lra_time_end(ctx);
  }
  { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)NC, a.cutCnt, cutOff); if (rc) return rc; }
  { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)NC, a.cntU, U); if (rc) return rc; }
  uint64_t TC = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&TC, cutOff + NC, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  char* wc = (char*)lra_ensure(ctx, 25, sz(TC + 1, 8) + sz(4 * (TC + NC) + 4, 4) + 1024);
  if (!wc) return LRA_ERR_NOMEM;
  a.cuts = (uint64_t*)take(wc, TC + 1, 8); a.tmp = (uint32_t*)take(wc, 4 * (TC + NC) + 4, 4);
  lra_time_begin(ctx, "split_clusters");
  hipLaunchKernelGGL(sc_emit, dim3(gc), dim3(256), 0, st, a);
  lra_time_end(ctx);
  { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)NC, a.cntP, P); if (rc) return rc; }
  uint64_t nU = 0, nP = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&nU, U + NC, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&nP, P + NC, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  const uint64_t NS = nU + nP;
  out->n_split = NS;
  char* wo = (char*)lra_ensure(ctx, 26, sz(NS + 1, 4) * 9 + 1024);
  if (!wo) return LRA_ERR_NOMEM;
  a.oqs = (uint32_t*)take(wo, NS + 1, 4); a.oqe = (uint32_t*)take(wo, NS + 1, 4); a.ots = (uint32_t*)take(wo, NS + 1, 4); a.ote = (uint32_t*)take(wo, NS + 1, 4);
  a.ostrand = (int32_t*)take(wo, NS + 1, 4); a.ocoarse = (int32_t*)take(wo, NS + 1, 4); a.oval = (int32_t*)take(wo, NS + 1, 4);
  a.onum = (int32_t*)take(wo, NS + 1, 4); a.oread = (uint32_t*)take(wo, NS + 1, 4);
  out->d_qs = a.oqs; out->d_qe = a.oqe; out->d_ts = a.ots; out->d_te = a.ote; out->d_strand = a.ostrand; out->d_coarse = a.ocoarse; out->d_val = a.oval;
  out->d_num_anchors = a.onum; out->d_read = a.oread;
  lra_time_begin(ctx, "split_clusters");
  hipLaunchKernelGGL(sc_read_off, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, st, a);
  hipLaunchKernelGGL(sc_layout, dim3(gc), dim3(256), 0, st, a);
  hipLaunchKernelGGL(sc_cluster_val, dim3(gc), dim3(256), 0, st, a);
  if (NS > 0) hipLaunchKernelGGL(sc_values, dim3((unsigned)((NS + 255) / 256)), dim3(256), 0, st, a, NS);
  lra_time_end(ctx);
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  LRA_HIP_CHECK(ctx, hipGetLastError());
  return LRA_OK;
}


// ---------------------------------------------------------------------------------------------------------------- switchindex
// Mapping_ultility.h:39-168 (Map_highacc.h:274): the chains of the box-fragment sparse DP are lists of split clusters; map them back to the
// clusters they were cut from, drop the links between pieces of one cluster, merge repeats (adjacent ones by std::unique, spread ones by
// keeping the first occurrence and skipping to behind the last), and drop clusters whose read interval lies inside their predecessor's.
// Chains are short lists rewritten in place by data-dependent steps: one lane per chain.
namespace {
struct SwArgs {
  uint64_t n_chains;
  const uint64_t* off; const uint32_t* ch; const uint8_t* link; const uint32_t* n_link;
  const uint64_t* sp_base; const uint64_t* cl_base; const int32_t* coarse; const uint32_t* cl_qs; const uint32_t* cl_qe;
  uint32_t* o_ch; uint8_t* o_link; uint32_t* o_n; uint32_t* o_nlink; uint32_t* status;
  uint32_t* t_ch; uint8_t* t_link; uint32_t* iv_s; uint32_t* iv_e;
};
__global__ void sw_kernel(SwArgs a) {
  const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= a.n_chains) return;
  const uint64_t b = a.off[c];
  long n = (long)(a.off[c + 1] - b);
  long nl = a.n_link[c];
  uint32_t* A = a.o_ch + b; uint8_t* L = a.o_link + b;
  uint32_t* TA = a.t_ch + b; uint8_t* TL = a.t_link + b; uint32_t* IS = a.iv_s + b; uint32_t* IE = a.iv_e + b;
  const int32_t* coarse = a.coarse + a.sp_base[c];
  const uint32_t* qs = a.cl_qs + a.cl_base[c]; const uint32_t* qe = a.cl_qe + a.cl_base[c];
  auto fail = [&]() { a.status[c] = LRA_ST_OOB_SLOT; a.o_n[c] = 0; a.o_nlink[c] = 0; };
  for (long i = 0; i < n; i++) A[i] = (uint32_t)coarse[a.ch[b + i]];                                    // :42-48
  for (long i = 0; i < nl; i++) L[i] = a.link[b + i];
  if (nl > 0) {                                                                                         // :52-69
    for (long i = 1; i < n; i++) if (A[i] == A[i - 1] && i - 1 >= nl) { fail(); return; }
    long sm = 0;
    for (long i = 0; i < nl; i++) { const bool rm = (i + 1 < n) && A[i + 1] == A[i]; if (!rm) { L[sm] = L[i]; sm++; } }
    nl = sm;
  }
  { long m = 0; for (long i = 0; i < n; i++) if (i == 0 || A[i] != A[m - 1]) A[m++] = A[i]; n = m; }     // std::unique :73-80
  if (n > 0) {                                                                                          // :84-143
    long niv = 0;
    for (long i = 0; i < n; i++) {
      bool first = true;
      for (long j = 0; j < i; j++) if (A[j] == A[i]) { first = false; break; }
      if (!first) continue;
      long last = i;
      for (long j = i + 1; j < n; j++) if (A[j] == A[i]) last = j;
      if (last + 1 > i + 1) { IS[niv] = (uint32_t)i; IE[niv] = (uint32_t)(last + 1); niv++; }
    }
    long m = 0, ml = 0, ste = 0, nc = 0;
    while (ste < niv) {
      while (nc <= (long)IS[ste]) {
        TA[m++] = A[nc];
        if (m > 1) { if (nc - 1 < 0 || nc - 1 >= nl) { fail(); return; } TL[ml++] = L[nc - 1]; }
        nc++;
      }
      nc = (long)IE[ste];
      ste++;
    }
    while (nc < n) {
      TA[m++] = A[nc];
      if (m > 1) { if (nc - 1 < 0 || nc - 1 >= nl) { fail(); return; } TL[ml++] = L[nc - 1]; }
      nc++;
    }
    for (long i = 0; i < m; i++) A[i] = TA[i];
    for (long i = 0; i < ml; i++) L[i] = TL[i];
    n = m; nl = ml;
  }
  {                                                                                                     // :147-166
    long sc = 0;
    bool prevRemoved = false;
    uint32_t prevCluster = 0;
    for (long i = 0; i < n; i++) {
      const uint32_t cr = A[i];
      bool rem = false;
      if (i >= 1 && !prevRemoved && qs[cr] >= qs[prevCluster] && qe[cr] <= qe[prevCluster]) rem = true;
      prevRemoved = rem; prevCluster = cr;                               // cremove[c - 1] and ch[c - 1] of the next step (ch is compacted afterwards)
      if (!rem) {
        A[sc] = cr;
        if (sc >= 1) { if (i - 1 >= nl || sc - 1 >= nl) { fail(); return; } L[sc - 1] = L[i - 1]; }
        sc++;
      }
    }
    if (sc - 1 < 0) { fail(); return; }
    n = sc; nl = sc - 1;
  }
  a.o_n[c] = (uint32_t)n; a.o_nlink[c] = (uint32_t)nl; a.status[c] = 0;
}
}  // namespace

extern "C" int lra_switchindex_batch(lra_ctx* ctx, uint64_t n_chains, const uint64_t* d_chain_off, const uint32_t* d_ch, const uint8_t* d_link,
                                     const uint32_t* d_n_link, const uint64_t* d_split_base, const uint64_t* d_cluster_base, const int32_t* d_coarse,
                                     const uint32_t* d_cl_qs, const uint32_t* d_cl_qe, uint64_t n_total, lra_switchindex_result* out) {
  if (!ctx || !out) return LRA_ERR_INVALID;
  memset(out, 0, sizeof *out);
  out->n_chains = n_chains;
  if (n_chains == 0) return LRA_OK;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t N = (size_t)n_total + 1, C1 = (size_t)n_chains + 1;
  char* w = (char*)lra_ensure(ctx, 69, al(N * 4) * 4 + al(N) * 2 + al(C1 * 4) * 3 + 4096);
  if (!w) return LRA_ERR_NOMEM;
  SwArgs a; memset(&a, 0, sizeof a);
  a.n_chains = n_chains; a.off = d_chain_off; a.ch = d_ch; a.link = d_link; a.n_link = d_n_link; a.sp_base = d_split_base; a.cl_base = d_cluster_base;
  a.coarse = d_coarse; a.cl_qs = d_cl_qs; a.cl_qe = d_cl_qe;
  a.o_ch = (uint32_t*)w; w += al(N * 4); a.t_ch = (uint32_t*)w; w += al(N * 4); a.iv_s = (uint32_t*)w; w += al(N * 4); a.iv_e = (uint32_t*)w; w += al(N * 4);
  a.o_link = (uint8_t*)w; w += al(N); a.t_link = (uint8_t*)w; w += al(N);
  a.o_n = (uint32_t*)w; w += al(C1 * 4); a.o_nlink = (uint32_t*)w; w += al(C1 * 4); a.status = (uint32_t*)w;
  lra_time_begin(ctx, "switchindex");
  hipLaunchKernelGGL(sw_kernel, dim3((unsigned)((n_chains + 63) / 64)), dim3(64), 0, st, a);
  lra_time_end(ctx);
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  LRA_HIP_CHECK(ctx, hipGetLastError());
  out->d_ch = a.o_ch; out->d_link = a.o_link; out->d_n = a.o_n; out->d_n_link = a.o_nlink; out->d_status = a.status;
  return LRA_OK;
}
