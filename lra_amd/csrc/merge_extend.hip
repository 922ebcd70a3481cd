// lra_amd/csrc/merge_extend.hip -- SURVEY §8a rows a9 (MergeChain) and a7 (second LinearExtend, DecideCoordinates, TrimOverlappedAnchors):
// what MapRead_lowacc does with the refined clusters of every chain before the second sparse DP (Map_lowacc.h:440-476).  gfx950 only.
//   MergeChain                                    ChainRefine.h:767-802
//   LinearExtend (pair version, skipsorting = 0)  LinearExtend.h:658-716 (DiagonalSort Sorting.h:36-62; the kernel of cluster.hip)
//   DecideCoordinates                             LinearExtend.h:105-127
//   TrimOverlappedAnchors                         LinearExtend.h:574-649 (LongAnchors :11-47)
// Mapping.  One lane per chain slot runs MergeChain (a scan over its one to three refined clusters).  All refined clusters of the batch
// are then diagonal-sorted together (segmented radix sort of (q - t, q) keys: only identical pairs tie) and extended by the same wave-per-
// cluster kernel the first LinearExtend uses.  One wave per merged cluster concatenates its members' anchors and reduces the
// DecideCoordinates box; one lane per merged cluster trims: the long anchors (>= 40) are ordered with LongAnchors' comparator by a
// faithful libstdc++ std::sort (anchors with equal keys but different lengths would tie) and walked in order.
// Algorithmic bytes: 8 B per refined match in, 12 B per extended anchor out (+ the bases Checkbp compares).
#include "common.h"
#include "scan.h"
#include "std_sort.h"
#include <rocprim/rocprim.hpp>

namespace {

struct MeArgs {
  uint64_t n_slots; int numAln; int K;
  const uint32_t* nChains; const uint64_t* chainStart; const uint32_t* nSplit; const uint32_t* spStatus; const uint8_t* spStrand; const int32_t* spChrom;
  const uint64_t* matchOff; const uint32_t* mq; const uint32_t* mt; const uint32_t* box;
  const uint64_t* pos;
  uint32_t* nCl; uint32_t* nGr; const uint64_t* clBase; const uint64_t* grBase;
  // per refined cluster (dense)
  uint64_t* cStart; uint64_t* cEnd; int* cStrand; int* cChrom; int* cRead; uint32_t* cGroup;
  // per merged cluster
  uint32_t* gFirst; uint32_t* gLast; uint32_t* gSlot;
  const uint32_t* eCount; uint32_t* gSize; const uint64_t* anchorOff;
  const uint32_t* eq; const uint32_t* et; const int* el;
  uint32_t* aq; uint32_t* at; int32_t* alen; uint32_t* gbox; int32_t* gstrand; int32_t* gchrom; uint64_t* scratch;
};

// MergeChain ChainRefine.h:767-802 (sp[t] = t); PASS 0 counts, PASS 1 fills the dense cluster / group tables
template <int PASS>
__global__ void me_merge(MeArgs a) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= a.n_slots) return;
  const uint64_t r = s / a.numAln;
  uint32_t nsp = 0;
  if ((uint32_t)(s % a.numAln) < a.nChains[r] && !a.spStatus[s]) nsp = a.nSplit[s];
  if (PASS == 0) { a.nCl[s] = nsp; a.nGr[s] = 0; }
  if (nsp == 0) return;
  const uint64_t cs = a.chainStart[s];
  const uint64_t cb = PASS ? a.clBase[s] : 0, gb = PASS ? a.grBase[s] : 0;
  uint32_t g = 0, g0 = 0;
  for (uint32_t t = 0; t < nsp; t++) {
    const uint64_t x = cs + t;
    if (t > 0) {
      const uint64_t xp = x - 1;
      int qdist = 9999, tdist = 9999;
      if (a.spChrom[xp] == a.spChrom[x] && a.spStrand[xp] == a.spStrand[x]) {
        const uint32_t pqs = a.box[4 * xp], pts = a.box[4 * xp + 2], pte = a.box[4 * xp + 3];
        const uint32_t cqe = a.box[4 * x + 1], cts = a.box[4 * x + 2], cte = a.box[4 * x + 3];
        qdist = pqs > cqe ? (int)(pqs - cqe) : 0;
        if (a.spStrand[xp] == 0) tdist = pts >= cte ? (int)(pts - cte) : 9999;
        else tdist = pte <= cts ? (int)(cts - pte) : 9999;
      }
      if (!(qdist <= 500 && tdist <= 500)) {
        if (PASS) { a.gFirst[gb + g] = (uint32_t)(cb + g0); a.gLast[gb + g] = (uint32_t)(cb + t - 1); a.gSlot[gb + g] = (uint32_t)s; }
        g++; g0 = t;
      }
    }
    if (PASS) {
      const uint64_t c = cb + t;
      a.cStart[c] = a.matchOff[x]; a.cEnd[c] = a.matchOff[x + 1]; a.cStrand[c] = a.spStrand[x]; a.cChrom[c] = a.spChrom[x]; a.cRead[c] = (int)r;
      a.cGroup[c] = (uint32_t)(gb + g);
    }
  }
  if (PASS) { a.gFirst[gb + g] = (uint32_t)(cb + g0); a.gLast[gb + g] = (uint32_t)(cb + nsp - 1); a.gSlot[gb + g] = (uint32_t)s; }
  g++;
  if (PASS == 0) a.nGr[s] = g;
}

// DiagonalSortOp (Sorting.h:36-47): (first.pos - second.pos, first.pos); one wave per refined cluster
__global__ void __launch_bounds__(64) me_keys(uint64_t ncl, const uint64_t* __restrict__ cStart, const uint64_t* __restrict__ cEnd, const uint32_t* __restrict__ mq,
                                              const uint32_t* __restrict__ mt, uint64_t* key, uint32_t* val) {
  for (uint64_t c = blockIdx.x; c < ncl; c += gridDim.x)
    for (uint64_t i = cStart[c] + threadIdx.x; i < cEnd[c]; i += 64) {
      const uint32_t q = mq[i], t = mt[i];
      key[i] = ((uint64_t)((long long)q - (long long)t + (1LL << 32)) << 31) | q;      // q < 2^31
      val[i] = (uint32_t)i;
    }
}

__global__ void __launch_bounds__(64) me_gather_sorted(uint64_t ncl, const uint64_t* __restrict__ cStart, const uint64_t* __restrict__ cEnd, const int* __restrict__ cChrom,
                                                       const uint64_t* __restrict__ pos, const uint32_t* __restrict__ val, const uint32_t* __restrict__ mq,
                                                       const uint32_t* __restrict__ mt, uint32_t* sq, uint32_t* st) {
  for (uint64_t c = blockIdx.x; c < ncl; c += gridDim.x) {
    const uint32_t coff = (uint32_t)pos[cChrom[c]];
    for (uint64_t i = cStart[c] + threadIdx.x; i < cEnd[c]; i += 64) { const uint32_t v = val[i]; sq[i] = mq[v]; st[i] = mt[v] + coff; }   // the kernel takes genome-wide t
  }
}

__global__ void me_group_size(uint64_t ng, MeArgs a) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ng) return;
  uint32_t n = 0;
  for (uint32_t c = a.gFirst[g]; c <= a.gLast[g]; c++) n += a.eCount[c];
  a.gSize[g] = n;
}

// Map_lowacc.h:458-468: members' extended anchors back to back, then DecideCoordinates with the last member's strand / chromosome
__global__ void __launch_bounds__(64) me_concat(uint64_t ng, MeArgs a) {
  const int lane = threadIdx.x;
  for (uint64_t g = blockIdx.x; g < ng; g += gridDim.x) {
    uint64_t o = a.anchorOff[g];
    uint32_t qS = 0xFFFFFFFFu, qE = 0, tS = 0xFFFFFFFFu, tE = 0;
    for (uint32_t c = a.gFirst[g]; c <= a.gLast[g]; c++) {
      const uint64_t b = a.cStart[c];
      const uint32_t n = a.eCount[c];
      const uint32_t coff = (uint32_t)a.pos[a.cChrom[c]];
      for (uint32_t i = lane; i < n; i += 64) {
        const uint32_t q = a.eq[b + i], t = a.et[b + i] - coff; const int l = a.el[b + i];
        a.aq[o + i] = q; a.at[o + i] = t; a.alen[o + i] = l;
        qS = min(qS, q); qE = max(qE, q + (uint32_t)l); tS = min(tS, t); tE = max(tE, t + (uint32_t)l);
      }
      o += n;
    }
    for (int off = 32; off > 0; off >>= 1) {
      qS = min(qS, (uint32_t)__shfl_xor(qS, off)); qE = max(qE, (uint32_t)__shfl_xor(qE, off));
      tS = min(tS, (uint32_t)__shfl_xor(tS, off)); tE = max(tE, (uint32_t)__shfl_xor(tE, off));
    }
    if (lane == 0) {
      const bool any = o > a.anchorOff[g];
      a.gbox[4 * g] = any ? qS : 0; a.gbox[4 * g + 1] = any ? qE : 0; a.gbox[4 * g + 2] = any ? tS : 0; a.gbox[4 * g + 3] = any ? tE : 0;
      a.gstrand[g] = any ? a.cStrand[a.gLast[g]] : 0; a.gchrom[g] = any ? a.cChrom[a.gLast[g]] : 0;
    }
  }
}

struct LongLess {                                                        // LongAnchors::operator() LinearExtend.h:26-43
  const uint32_t* Q; const uint32_t* T; const int32_t* L; int strand;
  __device__ bool operator()(uint64_t i, uint64_t j) const {
    if (strand == 0) { if (Q[i] != Q[j]) return Q[i] < Q[j]; return T[i] < T[j]; }
    const uint32_t ei = Q[i] + (uint32_t)L[i], ej = Q[j] + (uint32_t)L[j];
    if (ei != ej) return ei > ej;
    return T[i] < T[j];
  }
};

// TrimOverlappedAnchors LinearExtend.h:574-649, one lane per merged cluster
// (the GenomePairs version :722-780 is the same walk with strand 0 and long = 50: gstrand == NULL, minLen = 50)
__global__ void me_trim(uint64_t ng, const uint64_t* __restrict__ anchorOff, uint32_t* aq, uint32_t* at, int32_t* alen, const int32_t* __restrict__ gstrand,
                        uint64_t* scratch, int minLen) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ng) return;
  const uint64_t b = anchorOff[g];
  const uint32_t n = (uint32_t)(anchorOff[g + 1] - b);
  uint32_t* Q = aq + b; uint32_t* T = at + b; int32_t* L = alen + b;
  uint64_t* idx = scratch + b;
  const int S = gstrand ? gstrand[g] : 0;
  uint32_t m = 0;
  for (uint32_t i = 0; i < n; i++) if (L[i] >= minLen) idx[m++] = i;
  LongLess lt{Q, T, L, S};
  lra_std_sort::std_sort(idx, (long)m, lt);
  for (uint32_t ln = 1; ln < m; ln++) {
    const uint32_t prev = (uint32_t)idx[ln - 1], cur = (uint32_t)idx[ln];
    int overlap_r = 0, overlap_g = 0;
    if (S == 0) {
      if (Q[cur] < Q[prev] + (uint32_t)L[prev] && Q[cur] >= Q[prev] + (uint32_t)L[prev] - 30u) overlap_r = (int)(Q[prev] + (uint32_t)L[prev] - Q[cur]);
    } else {
      if (Q[cur] + (uint32_t)L[cur] > Q[prev] && Q[cur] + (uint32_t)L[cur] <= Q[prev] + 30u) overlap_r = (int)(Q[cur] + (uint32_t)L[cur] - Q[prev]);
    }
    if (T[cur] < T[prev] + (uint32_t)L[prev] && T[cur] >= T[prev] + (uint32_t)L[prev] - 30u) overlap_g = (int)(T[prev] + (uint32_t)L[prev] - T[cur]);
    if (overlap_r > 0 || overlap_g > 0) {
      const int overlap = max(overlap_r, overlap_g);
      if (S == 1) Q[prev] += (uint32_t)(overlap + 1);
      L[prev] -= overlap + 1;
    }
  }
}

__global__ void me_iota(uint64_t n, uint64_t* out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= n) out[i] = i;
}

__global__ void me_gather_off(uint64_t n, const uint64_t* idx, const uint64_t* src, uint64_t* out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= n) out[i] = src[idx[i]];
}

inline size_t sz(size_t n, size_t e) { return (n * e + 255) / 256 * 256; }

}  // namespace

extern "C" int lra_merge_extend_batch(lra_ctx* ctx, const lra_chain_result* ch, const lra_split_result* sp, const lra_btwn_result* bt, const char* d_seq,
                                      const uint64_t* d_read_off, const char* d_genome, const uint64_t* h_chrom_pos, int n_chrom, int K,
                                      lra_merge_result* out) {
  if (!ctx || !ch || !sp || !bt || !out || !h_chrom_pos || n_chrom < 1 || K < 1) return LRA_ERR_INVALID;
  memset(out, 0, sizeof *out);
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const uint64_t slots = sp->n_slots, NM = bt->n_matches;
  out->n_slots = slots;
  if (slots == 0) return LRA_OK;
  const size_t npos = (size_t)n_chrom + 1;
  auto take = [](char*& p, size_t n, size_t e) { char* r = p; p += sz(n, e); return r; };
  char* w = (char*)lra_ensure(ctx, 37, sz(slots + 2, 4) * 2 + sz(slots + 2, 8) * 2 + sz(npos, 8) + 4096);
  if (!w) return LRA_ERR_NOMEM;
  MeArgs a;
  memset(&a, 0, sizeof a);
  a.n_slots = slots; a.numAln = ch->num_aln; a.K = K; a.nChains = ch->d_n_chains; a.chainStart = ch->d_chain_start; a.nSplit = sp->d_n_split; a.spStatus = sp->d_status;
  a.spStrand = sp->d_sp_strand; a.spChrom = sp->d_sp_chrom; a.matchOff = bt->d_match_off; a.mq = bt->d_match_q; a.mt = bt->d_match_t; a.box = bt->d_box;
  a.nCl = (uint32_t*)take(w, slots + 2, 4); a.nGr = (uint32_t*)take(w, slots + 2, 4);
  uint64_t* clBase = (uint64_t*)take(w, slots + 2, 8); uint64_t* grBase = (uint64_t*)take(w, slots + 2, 8);
  uint64_t* dpos = (uint64_t*)take(w, npos, 8);
  a.clBase = clBase; a.grBase = grBase; a.pos = dpos;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(dpos, h_chrom_pos, npos * 8, hipMemcpyHostToDevice, st));
  const unsigned gs = (unsigned)((slots + 255) / 256);
  lra_time_begin(ctx, "merge_extend");
  hipLaunchKernelGGL(me_merge<0>, dim3(gs), dim3(256), 0, st, a);
  lra_time_end(ctx);
  { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)slots, a.nCl, clBase); if (rc) return rc; }
  { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)slots, a.nGr, grBase); if (rc) return rc; }
  uint64_t NCL = 0, NG = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&NCL, clBase + slots, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&NG, grBase + slots, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  out->n_groups = NG; out->d_slot_group_off = grBase;
  // (the per-cluster / per-group arrays are part of the result; the per-match arrays -- sort keys, gathered and extended matches -- are dead when the call returns: a
  // slot of their own, so that a caller with several result contexts can lend them ONE, lra_map_reads_lowacc_front)
  char* wc = (char*)lra_ensure(ctx, 38, sz(NCL + 1, 8) * 2 + sz(NCL + 1, 4) * 5 + sz(4 * NCL + 4, 4) + sz(NG + 2, 4) * 6 + sz(4 * NG + 4, 4) + sz(NG + 2, 8) * 2 + 4096);
  char* wm = (char*)lra_ensure(ctx, 100, sz(NM + 1, 8) * 2 + sz(NM + 1, 4) * 6 + 4096);
  if (!wc || !wm) return LRA_ERR_NOMEM;
  a.cStart = (uint64_t*)take(wc, NCL + 1, 8); a.cEnd = (uint64_t*)take(wc, NCL + 1, 8);
  a.cStrand = (int*)take(wc, NCL + 1, 4); a.cChrom = (int*)take(wc, NCL + 1, 4); a.cRead = (int*)take(wc, NCL + 1, 4); a.cGroup = (uint32_t*)take(wc, NCL + 1, 4);
  uint32_t* eCount = (uint32_t*)take(wc, NCL + 1, 4); uint32_t* ebox = (uint32_t*)take(wc, 4 * NCL + 4, 4);
  a.gFirst = (uint32_t*)take(wc, NG + 2, 4); a.gLast = (uint32_t*)take(wc, NG + 2, 4); a.gSlot = (uint32_t*)take(wc, NG + 2, 4); a.gSize = (uint32_t*)take(wc, NG + 2, 4);
  a.gstrand = (int32_t*)take(wc, NG + 2, 4); a.gchrom = (int32_t*)take(wc, NG + 2, 4); a.gbox = (uint32_t*)take(wc, 4 * NG + 4, 4);
  uint64_t* anchorOff = (uint64_t*)take(wc, NG + 2, 8); uint64_t* iota = (uint64_t*)take(wc, NG + 2, 8);
  uint64_t* key = (uint64_t*)take(wm, NM + 1, 8); uint64_t* key2 = (uint64_t*)take(wm, NM + 1, 8);
  uint32_t* val = (uint32_t*)take(wm, NM + 1, 4); uint32_t* val2 = (uint32_t*)take(wm, NM + 1, 4); uint32_t* sq = (uint32_t*)take(wm, NM + 1, 4);
  uint32_t* stt = (uint32_t*)take(wm, NM + 1, 4); uint32_t* eq = (uint32_t*)take(wm, NM + 1, 4); uint32_t* et = (uint32_t*)take(wm, NM + 1, 4);
  int* el = (int*)key;                                                   // the sort keys are dead once the matches are gathered
  a.eCount = eCount; a.eq = eq; a.et = et; a.el = el; a.anchorOff = anchorOff; a.scratch = key2;
  if (NCL == 0) { LRA_HIP_CHECK(ctx, hipMemsetAsync(anchorOff, 0, 16, st)); LRA_HIP_CHECK(ctx, hipStreamSynchronize(st)); out->d_anchor_off = anchorOff; return LRA_OK; }
  lra_time_begin(ctx, "merge_extend");
  hipLaunchKernelGGL(me_merge<1>, dim3(gs), dim3(256), 0, st, a);
  lra_time_end(ctx);
  const unsigned gw = (unsigned)std::min<uint64_t>(NCL, (uint64_t)ctx->num_cu * 32);
  if (NM > 0) {
    size_t temp_bytes = 0;
    (void)lra_segsort_pairs(ctx, nullptr, temp_bytes, nullptr, nullptr, nullptr, nullptr, (unsigned int)NM, (unsigned int)NCL, nullptr, nullptr, 0, 64, st);
    void* temp = lra_scratch(ctx, 2, temp_bytes + 256);
    if (!temp) return LRA_ERR_NOMEM;
    lra_time_begin(ctx, "merge_extend");
    hipLaunchKernelGGL(me_keys, dim3(gw), dim3(64), 0, st, NCL, (const uint64_t*)a.cStart, (const uint64_t*)a.cEnd, a.mq, a.mt, key, val);
    hipError_t e = lra_segsort_pairs(ctx, temp, temp_bytes, key, key2, val, val2, (unsigned int)NM, (unsigned int)NCL, a.cStart, a.cEnd, 0, 64, st);
    if (e != hipSuccess) { lra_time_end(ctx); return lra_set_err(ctx, LRA_ERR_HIP, "segmented sort: %s", hipGetErrorString(e)); }
    hipLaunchKernelGGL(me_gather_sorted, dim3(gw), dim3(64), 0, st, NCL, (const uint64_t*)a.cStart, (const uint64_t*)a.cEnd, (const int*)a.cChrom, (const uint64_t*)dpos,
                       (const uint32_t*)val2, a.mq, a.mt, sq, stt);
    lra_time_end(ctx);
  }
  { int rc = lra_launch_linear_extend(ctx, NCL, K, a.cStart, a.cEnd, a.cStrand, a.cChrom, a.cRead, sq, stt, dpos, (const unsigned char*)d_genome,
                                      (const unsigned char*)d_seq, d_read_off, eq, et, el, eCount, ebox); if (rc) return rc; }
  lra_time_begin(ctx, "merge_extend");
  hipLaunchKernelGGL(me_group_size, dim3((unsigned)((NG + 255) / 256)), dim3(256), 0, st, NG, a);
  lra_time_end(ctx);
  { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)NG, a.gSize, anchorOff); if (rc) return rc; }
  uint64_t NA = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&NA, anchorOff + NG, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  char* wa = (char*)lra_ensure(ctx, 39, sz(NA + 1, 4) * 3 + 1024);
  if (!wa) return LRA_ERR_NOMEM;
  a.aq = (uint32_t*)take(wa, NA + 1, 4); a.at = (uint32_t*)take(wa, NA + 1, 4); a.alen = (int32_t*)take(wa, NA + 1, 4);
  lra_time_begin(ctx, "merge_extend");
  hipLaunchKernelGGL(me_concat, dim3((unsigned)std::min<uint64_t>(NG, (uint64_t)ctx->num_cu * 32)), dim3(64), 0, st, NG, a);
  hipLaunchKernelGGL(me_trim, dim3((unsigned)((NG + 63) / 64)), dim3(64), 0, st, NG, (const uint64_t*)anchorOff, a.aq, a.at, a.alen, (const int32_t*)a.gstrand, a.scratch, 40);
  hipLaunchKernelGGL(me_iota, dim3((unsigned)((NG + 256) / 256)), dim3(256), 0, st, NG, iota);
  lra_time_end(ctx);
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  LRA_HIP_CHECK(ctx, hipGetLastError());
  out->n_anchors = NA; out->d_anchor_off = anchorOff; out->d_count = a.gSize; out->d_q = a.aq; out->d_t = a.at; out->d_len = a.alen; out->d_box = a.gbox;
  out->d_strand = a.gstrand; out->d_chrom = a.gchrom; out->d_group_slot = a.gSlot; out->d_group_first = a.gFirst; out->d_group_last = a.gLast; out->d_cluster_base = clBase;
  out->d_iota = iota;
  return LRA_OK;
}

// TrimOverlappedAnchors(GenomePairs& ExtendPairs, vector<int>& ExtendPairsMatchesLengths) (LinearExtend.h:722-780) on n_lists anchor lists
// (CSR d_off; lengths are modified in place).  Used inside RefinedAlignmentbtwnAnchors (LocalRefineAlignment.h:371).
extern "C" int lra_trim_anchor_pairs_batch(lra_ctx* ctx, uint64_t n_lists, const uint64_t* d_off, uint64_t n_anchors, uint32_t* d_q, uint32_t* d_t, int32_t* d_len) {
  if (!ctx) return LRA_ERR_INVALID;
  if (n_lists == 0 || n_anchors == 0) return LRA_OK;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  uint64_t* scratch = (uint64_t*)lra_scratch(ctx, 2, (n_anchors + 1) * 8);
  if (!scratch) return LRA_ERR_NOMEM;
  lra_time_begin(ctx, "merge_extend");
  hipLaunchKernelGGL(me_trim, dim3((unsigned)((n_lists + 63) / 64)), dim3(64), 0, ctx->stream, n_lists, d_off, d_q, d_t, d_len, (const int32_t*)nullptr, scratch, 50);
  lra_time_end(ctx);
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  LRA_HIP_CHECK(ctx, hipGetLastError());
  return LRA_OK;
}

// TrimOverlappedAnchors(vector<Cluster>&, start) (LinearExtend.h:574-649; LinearExtend_chain :783-793, Map_lowacc.h:476): the same walk per
// extended cluster with its strand (long anchors >= 40; reverse clusters are sorted by read end and have their read start moved)
extern "C" int lra_trim_overlapped_anchors_batch(lra_ctx* ctx, uint64_t n_clusters, const uint64_t* d_off, uint64_t n_anchors, const int32_t* d_strand,
                                                 uint32_t* d_q, uint32_t* d_t, int32_t* d_len) {
  if (!ctx || (n_clusters && !d_strand)) return LRA_ERR_INVALID;
  if (n_clusters == 0 || n_anchors == 0) return LRA_OK;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  uint64_t* scratch = (uint64_t*)lra_scratch(ctx, 2, (n_anchors + 1) * 8);
  if (!scratch) return LRA_ERR_NOMEM;
  lra_time_begin(ctx, "merge_extend");
  hipLaunchKernelGGL(me_trim, dim3((unsigned)((n_clusters + 63) / 64)), dim3(64), 0, ctx->stream, n_clusters, d_off, d_q, d_t, d_len, d_strand, scratch, 40);
  lra_time_end(ctx);
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  LRA_HIP_CHECK(ctx, hipGetLastError());
  return LRA_OK;
}

// ---------------------------------------------------------------------------------------------------------------- MergeMatchesSameDiag
// LinearExtend.h:795-829 (Map_highacc.h:642): whether anchor q opens a new Cluster_SameDiag entry depends on the pair (q - 1, q) alone, so
// every anchor tests its own pair and the entries are the stretches between the heads: one wave per cluster, count then emit.
namespace {
struct SdArgs {
  uint64_t n_clusters;
  const uint64_t* a_off; const uint32_t* q; const uint32_t* t; const int32_t* len; const uint8_t* overlap; const int32_t* strand;
  int merge_dist;
  uint32_t* cnt; const uint64_t* g_off; uint32_t* g_start; uint32_t* g_end; uint32_t* status;
};
__device__ __forceinline__ bool sd_head(const SdArgs& a, uint64_t base, long i, int strand) {   // does anchor i (>= 1) start a new entry?
  const uint64_t x = base + i, p = x - 1;
  const long dq = strand == 0 ? (long)a.t[x] - (long)a.q[x] : (long)a.q[x] + (long)a.t[x] + a.len[x];
  const long dp = strand == 0 ? (long)a.t[p] - (long)a.q[p] : (long)a.q[p] + (long)a.t[p] + a.len[p];
  const uint32_t prevEnd = a.q[p] + (uint32_t)a.len[p];
  const long gap = labs((long)a.q[x] - ((long)a.q[p] + a.len[p]));
  return !(a.overlap[p] == 0 && a.overlap[x] == 0 && dp == dq && prevEnd < a.q[x] && gap <= a.merge_dist);
}
template <bool EMIT>
__global__ void __launch_bounds__(64) sd_kernel(SdArgs a) {
  const uint64_t c = blockIdx.x;
  const int lane = threadIdx.x;
  if (c >= a.n_clusters) return;
  const uint64_t base = a.a_off[c];
  const long n = (long)(a.a_off[c + 1] - base);
  if (n <= 0) { if (!EMIT && lane == 0) { a.cnt[c] = 0; a.status[c] = LRA_ST_OOB_SLOT; } return; }   // the reference reads matches[0]
  const int strand = a.strand[c];
  const unsigned long long below = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));
  uint32_t ng = 0;
  const uint64_t go = EMIT ? a.g_off[c] : 0;
  for (long i0 = 0; i0 < n; i0 += 64) {
    const long i = i0 + lane;
    const bool head = i < n && (i == 0 || sd_head(a, base, i, strand));
    const unsigned long long m = __ballot(head);
    if (EMIT && head) {
      const uint32_t g = ng + (uint32_t)__popcll(m & below);
      a.g_start[go + g] = (uint32_t)i;
      if (g > 0) a.g_end[go + g - 1] = (uint32_t)i;                      // the previous entry ends where this one starts
    }
    ng += (uint32_t)__popcll(m);
  }
  if (lane == 0) {
    if (EMIT) a.g_end[go + ng - 1] = (uint32_t)n;
    else { a.cnt[c] = ng; a.status[c] = 0; }
  }
}
}  // namespace

extern "C" int lra_merge_same_diag_batch(lra_ctx* ctx, uint64_t n_clusters, const uint64_t* d_anchor_off, const uint32_t* d_q, const uint32_t* d_t,
                                         const int32_t* d_len, const uint8_t* d_overlap, const int32_t* d_strand, int merge_dist,
                                         lra_same_diag_result* out) {
  if (!ctx || !out) return LRA_ERR_INVALID;
  memset(out, 0, sizeof *out);
  out->n_clusters = n_clusters;
  if (n_clusters == 0) return LRA_OK;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t n1 = (size_t)n_clusters + 2;
  char* w = (char*)lra_ensure(ctx, 67, ((n1 * 4 + 255) & ~(size_t)255) * 2 + ((n1 * 8 + 255) & ~(size_t)255) + 4096);
  if (!w) return LRA_ERR_NOMEM;
  uint32_t* cnt = (uint32_t*)w; w += (n1 * 4 + 255) & ~(size_t)255;
  uint32_t* status = (uint32_t*)w; w += (n1 * 4 + 255) & ~(size_t)255;
  uint64_t* g_off = (uint64_t*)w;
  SdArgs a; memset(&a, 0, sizeof a);
  a.n_clusters = n_clusters; a.a_off = d_anchor_off; a.q = d_q; a.t = d_t; a.len = d_len; a.overlap = d_overlap; a.strand = d_strand; a.merge_dist = merge_dist;
  a.cnt = cnt; a.status = status;
  lra_time_begin(ctx, "merge_same_diag");
  hipLaunchKernelGGL(sd_kernel<false>, dim3((unsigned)n_clusters), dim3(64), 0, st, a);
  lra_time_end(ctx);
  { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)n_clusters, cnt, g_off); if (rc) return rc; }
  uint64_t ng = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&ng, g_off + n_clusters, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  uint32_t* gs = (uint32_t*)lra_ensure(ctx, 68, (ng + 1) * 8 + 512);
  if (!gs) return LRA_ERR_NOMEM;
  uint32_t* ge = gs + ((ng + 64) & ~(uint64_t)63);
  a.g_off = g_off; a.g_start = gs; a.g_end = ge;
  lra_time_begin(ctx, "merge_same_diag");
  hipLaunchKernelGGL(sd_kernel<true>, dim3((unsigned)n_clusters), dim3(64), 0, st, a);
  lra_time_end(ctx);
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  LRA_HIP_CHECK(ctx, hipGetLastError());
  out->n_groups = ng; out->d_group_off = g_off; out->d_start = gs; out->d_end = ge; out->d_status = status;
  return LRA_OK;
}


// ---------------------------------------------------------------------------------------------------------------- SwitchToOriginalAnchors
// LocalRefineAlignment.h:187-199 (:576): the chain of the SplitChain sparse DP runs over Cluster_SameDiag entries; put the original anchors
// back (entry k of cluster c -> anchors end[k]-1 .. start[k], ClusterIndex = coarse).  Count, scan, one wave per chain element to emit.
namespace {
struct SoArgs {
  uint64_t n_elems;
  const int32_t* e_cluster; const uint32_t* e_entry; const uint64_t* g_off; const uint32_t* start; const uint32_t* end; const int32_t* coarse;
  uint32_t* cnt; const uint64_t* out_off; uint32_t* out_anchor; int32_t* out_cluster;
};
__global__ void so_count(SoArgs a) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n_elems) return;
  const uint64_t g = a.g_off[a.e_cluster[i]] + a.e_entry[i];
  a.cnt[i] = a.end[g] > a.start[g] ? a.end[g] - a.start[g] : 0u;
}
__global__ void __launch_bounds__(64) so_emit(SoArgs a) {
  const uint64_t i = blockIdx.x;
  if (i >= a.n_elems) return;
  const int c = a.e_cluster[i];
  const uint64_t g = a.g_off[c] + a.e_entry[i];
  const uint32_t s = a.start[g], e = a.end[g];
  const uint64_t o = a.out_off[i];
  const int co = a.coarse[c];
  for (uint32_t x = threadIdx.x; s + x < e; x += 64) { a.out_anchor[o + x] = e - 1 - x; a.out_cluster[o + x] = co; }
}
}  // namespace

extern "C" int lra_switch_to_original_anchors_batch(lra_ctx* ctx, uint64_t n_chains, const uint64_t* d_chain_off, uint64_t n_elems, const int32_t* d_elem_cluster,
                                                    const uint32_t* d_elem_entry, const lra_same_diag_result* same_diag, const int32_t* d_coarse,
                                                    lra_original_anchors_result* out) {
  if (!ctx || !out || !same_diag) return LRA_ERR_INVALID;
  memset(out, 0, sizeof *out);
  out->n_chains = n_chains;
  if (n_chains == 0 || n_elems == 0) return LRA_OK;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  char* w = (char*)lra_ensure(ctx, 76, al((n_elems + 2) * 4) + al((n_elems + 2) * 8) + al((n_chains + 2) * 8) + 4096);
  if (!w) return LRA_ERR_NOMEM;
  SoArgs a; memset(&a, 0, sizeof a);
  a.n_elems = n_elems; a.e_cluster = d_elem_cluster; a.e_entry = d_elem_entry; a.g_off = same_diag->d_group_off; a.start = same_diag->d_start; a.end = same_diag->d_end;
  a.coarse = d_coarse;
  a.cnt = (uint32_t*)w; w += al((n_elems + 2) * 4);
  uint64_t* e_off = (uint64_t*)w; w += al((n_elems + 2) * 8);
  uint64_t* c_off = (uint64_t*)w;
  hipLaunchKernelGGL(so_count, dim3((unsigned)((n_elems + 255) / 256)), dim3(256), 0, st, a);
  { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)n_elems, a.cnt, e_off); if (rc) return rc; }
  uint64_t total = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&total, e_off + n_elems, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  uint32_t* oa = (uint32_t*)lra_ensure(ctx, 77, al((total + 1) * 4) * 2 + 512);
  if (!oa) return LRA_ERR_NOMEM;
  int32_t* oc = (int32_t*)((char*)oa + al((total + 1) * 4));
  a.out_off = e_off; a.out_anchor = oa; a.out_cluster = oc;
  hipLaunchKernelGGL(so_emit, dim3((unsigned)n_elems), dim3(64), 0, st, a);
  // chain c's anchors start where its first element's do
  hipLaunchKernelGGL(me_gather_off, dim3((unsigned)((n_chains + 256) / 256)), dim3(256), 0, st, n_chains, d_chain_off, (const uint64_t*)e_off, c_off);
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  LRA_HIP_CHECK(ctx, hipGetLastError());
  out->n_anchors = total; out->d_chain_off = c_off; out->d_anchor = oa; out->d_cluster = oc;
  return LRA_OK;
}
