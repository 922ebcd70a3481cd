// lra_amd/csrc/refine_space.hip -- SURVEY §8a row a11, the gap-seeding function RefineSpace (ClusterRefine.h:242-325) for a batch
// of (read span x genome span) gaps.  gfx950 only.
//   both spans < 1000:  AffineOneGapAlign(query, ref, localMatch, localMismatch, localIndel, 30) (aog.hip), then the exact K-mers every
//                       K bases of its blocks and identity = matching bases / min(span)                      :262-292
//   otherwise:          StoreMinimizers_noncanonical<GenomeTuple,Tuple> (MinCount.h:182-338) of both spans, std::sort (the library's
//                       libstdc++-exact sort), CompareLists<GenomeTuple,Tuple> with the diagonal band        :305-311
//   then "+= qs", "+= ts - lrts" and the reverse-strand flip                                                  :313-323
// Mapping: the short-gap branch (the common one: gaps between chained anchors) is the batched AffineOneGapAlign plus one lane per gap
// for the K-mer scan; the long-gap branch runs one lane per (gap, side) for the sketch and one lane per gap for the list walk --
// literal serial code, it is rare and bounded by refineSpaceDist.  Algorithmic bytes: qLen + tLen + 12 per block + 8 per pair.
#include "common.h"
#include <stdlib.h>
#include "scan.h"
#include <algorithm>

int lra_aog_launch_device(lra_ctx* ctx, int n, const char* d_qseq, const char* d_tseq, const uint64_t* d_q_off, const int32_t* d_q_len,
                          const uint64_t* d_t_off, const int32_t* d_t_len, const int32_t* d_k, int m, int mm, int indel, int32_t* d_score,
                          int32_t* d_nblocks, int32_t* d_blocks, const uint64_t* d_block_off, int32_t* d_status);

namespace {

constexpr uint64_t FOR_MASK = 0x7FFFFFFFFFFFFFFFULL;
constexpr int MAX_W = 32;

__device__ __forceinline__ int code_n(unsigned char c) {                 // seqMapN (SeqUtils.h:42-75)
  if (c < 8) return c & 3;
  switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}
__device__ __forceinline__ int code(unsigned char c) { const int v = code_n(c); return v > 3 ? 0 : v; }   // seqMap

struct RsArgs {
  int n;
  const char* qseq; const char* tseq; const uint64_t* q_off; const int32_t* q_len; const uint64_t* t_off; const int32_t* t_len;
  const uint32_t* t_span; const int32_t* K; const int32_t* W; const int32_t* diag; const uint32_t* q_add; const uint32_t* t_add; const uint32_t* flip;
  long maxFreq; const int32_t* maxFreqArr;   // per-problem localMaxFreq when non-null
  // classification
  uint32_t* isSmall; uint64_t* smallPos; uint32_t* smallIdx; uint32_t* largeIdx;   // smallPos: exclusive scan of isSmall
  // small branch (compact, indexed by position among the small ones)
  uint64_t* sq_off; int32_t* sq_len; uint64_t* st_off; int32_t* st_len; int32_t* sk; uint32_t* bcap; const uint64_t* boff;
  const int32_t* nblocks; const int32_t* blocks; const int32_t* aogStatus;
  // large branch: lists 2*j (target), 2*j+1 (query) of large problem j
  uint32_t* lcnt; const uint64_t* loff; uint64_t* lkey; uint32_t* lpos;
  int4* rects; uint32_t* nrect;              // the (target run x query run) rectangles the list walk emits, per large problem
  // output
  uint32_t* cnt; const uint64_t* pair_off; uint32_t* outQ; uint32_t* outT; float* identity; uint32_t* status;
};

__global__ void rs_classify(RsArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  a.isSmall[i] = (a.q_len[i] < 1000 && a.t_len[i] < 1000) ? 1u : 0u;   // querySeq.size() < 1000 and refSeq.size() < 1000  :262
  a.status[i] = 0; a.identity[i] = -1.f; a.cnt[i] = 0;
}

__global__ void rs_split(RsArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const uint64_t sp = a.smallPos[i];
  if (a.isSmall[i]) {
    a.smallIdx[sp] = (uint32_t)i;
    a.sq_off[sp] = a.q_off[i]; a.sq_len[sp] = a.q_len[i]; a.st_off[sp] = a.t_off[i]; a.st_len[sp] = a.t_len[i]; a.sk[sp] = 30;
    a.bcap[sp] = (uint32_t)(a.q_len[i] + a.t_len[i] + 8);
  } else a.largeIdx[(uint64_t)i - sp] = (uint32_t)i;
}

// short gaps: matching bases and exact K-mers of the AffineOneGapAlign blocks (:265-291)
template <bool EMIT>
__global__ void rs_small(RsArgs a, int nSmall) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nSmall) return;
  const uint32_t i = a.smallIdx[s];
  const char* q = a.qseq + a.q_off[i]; const char* t = a.tseq + a.t_off[i];
  const int K = a.K[i];
  const int32_t* B = a.blocks + 3 * a.boff[s];
  const int nb = a.nblocks[s];
  int nMatch = 0; uint32_t np = 0;
  const uint64_t o = EMIT ? a.pair_off[i] : 0;
  const uint32_t qAdd = a.q_add[i], tAdd = a.t_add[i], flip = a.flip[i];
  for (int b = 0; b < nb; b++) {
    const int bq = B[3 * b], bt = B[3 * b + 1], bl = B[3 * b + 2];
    if (!EMIT) for (int x = 0; x < bl; x++) nMatch += q[bq + x] == t[bt + x];
    if (bl > K)
      for (int bp = 0; bp + K < bl; bp += K) {
        bool mis = false;
        for (int x = 0; x < K; x++) if (t[bt + bp + x] != q[bq + bp + x]) { mis = true; break; }
        if (!mis) {
          if (EMIT) { uint32_t pq = (uint32_t)(bq + bp) + qAdd; if (flip) pq = flip - pq - (uint32_t)K; a.outQ[o + np] = pq; a.outT[o + np] = (uint32_t)(bt + bp) + tAdd; }
          np++;
        }
      }
  }
  if (!EMIT) {
    a.cnt[i] = np;
    a.identity[i] = nMatch / (float)min(a.q_len[i], a.t_len[i]);
    if (a.aogStatus[s]) a.status[i] |= (uint32_t)a.aogStatus[s];
  }
}

// long gaps: StoreMinimizers_noncanonical<GenomeTuple,Tuple>(seq, seqLen, K, W, out, false)   MinCount.h:182-338
// One WAVE per (gap, side): the scan is a serial state machine, so every lane runs it on the same (uniform) values and lane 0 writes;
// what the wave buys is the memory system -- the bases come through a 1 KB tile in LDS that the 64 lanes refill together, and the
// w-entry ring lives in LDS instead of per-lane scratch, so no step waits on HBM.
constexpr int SK_TILE = 1024, SK_BACK = 256;
__device__ __forceinline__ void rs_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
template <bool EMIT>
__global__ void __launch_bounds__(64) rs_sketch(RsArgs a, int nLarge) {
  __shared__ unsigned char tile[SK_TILE];
  __shared__ uint64_t ringT[MAX_W];
  __shared__ uint32_t ringP[MAX_W];
  const int l = blockIdx.x, lane = threadIdx.x;
  if (l >= 2 * nLarge) return;
  const uint32_t i = a.largeIdx[l >> 1];
  const bool isQ = l & 1;
  const unsigned char* gseq = (const unsigned char*)(isQ ? a.qseq + a.q_off[i] : a.tseq + a.t_off[i]);
  const uint32_t seqLen = (uint32_t)(isQ ? a.q_len[i] : a.t_len[i]);
  long tileBase = -(long)SK_TILE - 1;
  auto at = [&](long x) -> unsigned char {                               // seq[x], x uniform over the wave
    if (x < tileBase || x >= tileBase + SK_TILE) {
      rs_wave_sync();
      tileBase = x > SK_BACK ? ((x - SK_BACK) & ~15L) : 0;
      for (int o = lane * 16; o < SK_TILE; o += 64 * 16)
        for (int b = 0; b < 16; b++) { const long g = tileBase + o + b; tile[o + b] = g < (long)seqLen ? gseq[g] : 0; }
      rs_wave_sync();
    }
    return tile[x - tileBase];
  };
  struct SeqView { decltype(at)& f; __device__ unsigned char operator[](long x) const { return f(x); } };
  SeqView seq{at};
  const int k = a.K[i], w = a.W[i];
  uint32_t n = 0;
  const uint64_t o = EMIT ? a.loff[l] : 0;
  auto emit = [&](uint64_t t, uint32_t p) { if (EMIT && lane == 0) { a.lkey[o + n] = t; a.lpos[o + n] = p; } n++; };
  auto done = [&]() { if (lane == 0) a.lcnt[l] = n; };
  if (w > MAX_W || w < 1 || k < 1 || k > 31) { if (lane == 0) { a.lcnt[l] = 0; atomicOr(&a.status[i], (uint32_t)LRA_ST_RANGE); } return; }
  if (seqLen < (uint32_t)k) { done(); return; }
  const int span = w + k - 1;
  if (seqLen < (uint32_t)span) { done(); return; }
  if (w >= 2 && w <= 16) {
    // ---- the same scan, 64 positions per round (the serial machine below is what it restates; w > 16 takes the serial machine; the maps are 8 nibbles in a 32-bit word
    // up to w = 8 and 16 nibbles in a 64-bit word beyond -- -CONTIG sketches its gaps with w = 10, and a 50 kb gap through the serial machine was a 26 ms launch).
    // The active minimizer's VALUE at p is the minimum of window (p - w, p]; only WHICH of several equal k-mers is active depends on the past, through the offset
    // o = p - actP in [0, w - 1].  Step p maps o to:  p - R(p) if o = w - 1 (the active one leaves the window: the ring rescan, first minimum in ring-slot order, i.e.
    // by (value, position mod w));  0 if k-mer p is smaller than the previous window's minimum;  o + 1 otherwise.  These maps (w entries of 4 bits) compose, so the
    // offsets of 64 consecutive positions are one wave-wide prefix "sum" of maps applied to the offset carried in.  A tuple goes out at p when the offset was reset
    // (either way) and the reference's validity window allows it: no N in [s, p + k - 1] with s = (last N at or before p + k - 1) + 1, p >= s + w - 1 and
    // s < seqLen - span (the very first window's tuple, s = 0, goes out unconditionally) -- what nvStart / nvEnd / find_valid amount to position by position.
    __shared__ uint64_t kk[128];                                           // k-mers of the last 128 positions (position & 127)
    __shared__ unsigned char cd[128];
    if (seqLen == (uint32_t)span) { done(); return; }                      // find_valid's strict bound: a sequence of exactly one window has no valid start
    const long pLast = (long)seqLen - k;                                   // last k-mer position
    uint64_t kmask = 0;
    for (int x = 0; x < k; x++) { kmask <<= 2; kmask += 3; }
    long lastN = -1;                                                       // last N at or before position pb + k - 2
    { long m = -1; for (int x = lane; x < k - 1; x += 64) if (code_n(gseq[x]) > 3) m = x; for (int o2 = 32; o2 > 0; o2 >>= 1) m = max(m, __shfl_xor(m, o2)); lastN = m; }
    int carryO = 0; uint64_t mPrev = 0;                                    // offset of the active minimizer at pb - 1; minimum of window (pb - 1 - w, pb - 1]
    for (long pb = 0; pb <= pLast; pb += 64) {
      const long p = pb + lane;
      const bool act = p <= pLast;
      // codes of positions pb .. pb + 63 + k - 1
      rs_wave_sync();
      for (int x = lane; x < 64 + k - 1; x += 64) { const long g = pb + x; cd[x] = g < (long)seqLen ? (unsigned char)code_n(gseq[g]) : 4; }
      rs_wave_sync();
      uint64_t c = 0;
      if (act) { for (int j = 0; j < k; j++) { const int v = cd[lane + j]; c = (c << 2) + (uint64_t)(v > 3 ? 0 : v); } c &= kmask; c &= FOR_MASK; }
      kk[p & 127] = c;
      const bool badNew = act && cd[lane + k - 1] > 3;                      // position p + k - 1
      rs_wave_sync();
      // window minimum, the rescan's choice, and (first window only) the earliest minimum
      uint64_t mWin = ~0ULL; long rPos = p, ePos = p;
      if (act && p >= w - 1) {
        uint64_t rVal = ~0ULL; int rSlot = 1 << 30;
        for (int j = 0; j < w; j++) {
          const long q = p - j; const uint64_t v = kk[q & 127]; const int slot = (int)(q % w);
          if (v < mWin || (v == mWin && q < ePos)) { mWin = v; ePos = q; }
          if (v < rVal || (v == rVal && slot < rSlot)) { rVal = v; rSlot = slot; rPos = q; }
        }
      }
      uint64_t mP1 = __shfl_up(mWin, 1); if (lane == 0) mP1 = mPrev;        // minimum of window (p - 1 - w, p - 1]
      const bool lt = act && p >= w && c < mP1;
      // the step's map
      int oNow;
      if (w <= 8) {
        uint32_t F = 0x76543210u;
        if (act && p >= w) {
          F = 0;
          for (int o2 = 0; o2 < w; o2++) { const uint32_t to = (o2 == w - 1) ? (uint32_t)(p - rPos) : (lt ? 0u : (uint32_t)(o2 + 1)); F |= to << (4 * o2); }
        } else if (act && p == w - 1) {                                    // the first window: whatever came in, the offset is that of the earliest minimum
          const uint32_t to = (uint32_t)(p - ePos);
          F = 0; for (int o2 = 0; o2 < 8; o2++) F |= to << (4 * o2);
        }
        uint32_t G = F;                                                    // inclusive prefix: G = F_lane o ... o F_0
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t E = __shfl_up(G, d);
          if (lane >= d) { uint32_t r = 0; for (int o2 = 0; o2 < 8; o2++) { const uint32_t g1 = (E >> (4 * o2)) & 15u; r |= ((G >> (4 * g1)) & 15u) << (4 * o2); } G = r; }
        }
        oNow = (int)((G >> (4 * carryO)) & 15u);
      } else {                                                             // the same with 16 nibbles
        uint64_t F = 0xFEDCBA9876543210ull;
        if (act && p >= w) {
          F = 0;
          for (int o2 = 0; o2 < w; o2++) { const uint64_t to = (o2 == w - 1) ? (uint64_t)(p - rPos) : (lt ? 0ull : (uint64_t)(o2 + 1)); F |= to << (4 * o2); }
        } else if (act && p == w - 1) {
          const uint64_t to = (uint64_t)(p - ePos);
          F = 0; for (int o2 = 0; o2 < 16; o2++) F |= to << (4 * o2);
        }
        uint64_t G = F;
        for (int d = 1; d < 64; d <<= 1) {
          const uint64_t E = (uint64_t)__shfl_up((unsigned long long)G, d);
          if (lane >= d) { uint64_t r = 0; for (int o2 = 0; o2 < 16; o2++) { const uint32_t g1 = (uint32_t)((E >> (4 * o2)) & 15ull); r |= ((G >> (4 * g1)) & 15ull) << (4 * o2); } G = r; }
        }
        oNow = (int)((G >> (4 * carryO)) & 15ull);
      }
      int oBefore = __shfl_up(oNow, 1); if (lane == 0) oBefore = carryO;
      const bool event = act && (p == w - 1 || (p >= w && (oBefore >= w - 1 || lt)));
      // validity
      long ln = badNew ? p + k - 1 : -1;
      for (int d = 1; d < 64; d <<= 1) { const long o3 = __shfl_up(ln, d); if (lane >= d) ln = max(ln, o3); }
      ln = max(ln, lastN);
      const bool V = act && p >= ln + w && ln + 1 < (long)seqLen - span;     // s = ln + 1: p >= s + w - 1, and find_valid's strict bound on s
      const bool out = event && V;
      const unsigned long long om = __ballot(out);
      if (out) {
        const long ap = p - oNow;
        const uint32_t at2 = n + (uint32_t)__popcll(om & ((lane == 0) ? 0ULL : (~0ULL >> (64 - lane))));
        if (EMIT) { a.lkey[o + at2] = kk[ap & 127]; a.lpos[o + at2] = (uint32_t)ap; }
      }
      n += (uint32_t)__popcll(om);
      // carries: the last active lane's offset, window minimum, last N
      const int lastLane = (int)min(63L, pLast - pb);
      carryO = __shfl(oNow, lastLane); mPrev = __shfl(mWin, lastLane); lastN = __shfl(ln, lastLane);
    }
    done();
    return;
  }
  uint64_t mask = 0;
  for (int x = 0; x < k; x++) { mask <<= 2; mask += 3; }
  long nvStart = 0, nvEnd = 0;
  auto find_valid = [&]() -> bool {
    bool valid = false;
    while ((uint32_t)nvStart < seqLen - (uint32_t)span && !valid) {
      valid = true;
      for (long x = nvStart; valid && x < nvStart + span; x++) if (code_n(seq[x]) > 3) { nvStart = x + 1; valid = false; }
    }
    return valid;
  };
  if (!find_valid()) { done(); return; }
  nvEnd = nvStart + span;
  uint64_t cur = 0;
  for (int p = 0; p < k; p++) { cur <<= 2; cur += (uint64_t)code(seq[p]); }
  uint64_t actT = cur; uint32_t actP = 0;
  ringT[0] = actT; ringP[0] = 0;
  uint32_t p;
  for (p = 1; p < (uint32_t)w && p < seqLen - k + 1; p++) {
    cur = ((cur << 2) & mask) + (uint64_t)code(seq[p + k - 1]);
    const uint64_t c = cur & FOR_MASK;
    if (c < actT) { actT = c; actP = p; }
    ringT[p % w] = c; ringP[p % w] = p;
  }
  if (nvEnd == span) emit(actT, actP);
  for (p = w; p < seqLen - k + 1; p++) {
    cur = ((cur << 2) & mask) + (uint64_t)code(seq[p + k - 1]);
    const uint64_t c = cur & FOR_MASK;
    if (nvEnd == (long)(p + k - 1)) {
      if (code_n(seq[p + k - 1]) <= 3) nvEnd++;
      else {
        nvStart = p + k;
        if (!find_valid()) { done(); return; }
        nvEnd = nvStart + span;
      }
    }
    ringT[p % w] = c; ringP[p % w] = p;
    if (p - w >= actP) {
      actT = ringT[0]; actP = ringP[0];
      for (int j = 1; j < w; j++) if ((ringT[j] & FOR_MASK) < (actT & FOR_MASK)) { actT = ringT[j]; actP = ringP[j]; }
      if (nvEnd == (long)(p + k)) emit(actT, actP);
    } else if ((c & FOR_MASK) < (actT & FOR_MASK)) {
      actT = c; actP = p;
      if (nvEnd == (long)(p + k)) emit(actT, actP);
    }
  }
  done();
}

// capacity of list l in the single-pass sketch: every k-mer position could be emitted
__global__ void rs_caps(RsArgs a, int nLarge, uint32_t* cap) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= 2 * nLarge) return;
  const uint32_t i = a.largeIdx[l >> 1];
  const long len = (l & 1) ? a.q_len[i] : a.t_len[i];
  const long k = a.K[i];
  cap[l] = (uint32_t)(len >= k && k >= 1 ? len - k + 1 : 0);
}
// one wave per list: raw (capacity-spaced) tuples -> the CSR lists
__global__ void __launch_bounds__(64) rs_compact(int nLists, const uint64_t* __restrict__ capOff, const uint64_t* __restrict__ loff, const uint64_t* __restrict__ rk,
                                                  const uint32_t* __restrict__ rp, uint64_t* __restrict__ lkey, uint32_t* __restrict__ lpos) {
  const int l = blockIdx.x;
  if (l >= nLists) return;
  const uint64_t s = capOff[l], d = loff[l], n = loff[l + 1] - d;
  for (uint64_t x = threadIdx.x; x < n; x += 64) { lkey[d + x] = rk[s + x]; lpos[d + x] = rp[s + x]; }
}

// long gaps: CompareLists<GenomeTuple,Tuple>(query, target, ..., Global = false, maxDiagNum, minDiagNum, canonical = false)  CompareLists.h:9-146
// One WAVE per gap, the walk on uniform values as in rs_sketch; the two sorted key lists are staged in LDS (when they fit) so that the
// searches and run scans of the walk never wait on HBM.  The walk runs once: every emission of the reference is a rectangle (a run of
// equal target keys x a run of equal query keys, target-major); the walk records the rectangle and counts its cells that pass the
// diagonal band with all lanes, rs_emit then writes the pairs of all rectangles in the same order.
constexpr int CMP_LDS_KEYS = 7168;                                        // 56 KB of keys per workgroup
struct RsBand { long long minDiag, maxDiag; };
__device__ __forceinline__ RsBand rs_band(const RsArgs& a, uint32_t i) {
  const long long diag2 = (long long)a.t_span[i] - (long long)(uint32_t)a.q_len[i];
  return RsBand{min(0LL, diag2) - a.diag[i], max(0LL, diag2) + a.diag[i]};
}
__device__ __forceinline__ bool rs_pass(const RsBand& bd, const uint32_t* tp, const uint32_t* qp, long qi, long ti) {
  if (bd.maxDiag != 0 && bd.minDiag != 0) {                              // :87
    const long long d = (long long)tp[ti] - (long long)qp[qi];
    if (!(d <= bd.maxDiag && d >= bd.minDiag)) return false;
  }
  return true;
}
__global__ void __launch_bounds__(64) rs_compare(RsArgs a, int nLarge, int ldsKeys) {
  extern __shared__ uint64_t skeys[];
  const int j = blockIdx.x, lane = threadIdx.x;
  if (j >= nLarge) return;
  const uint32_t i = a.largeIdx[j];
  const uint64_t* tkG = a.lkey + a.loff[2 * j]; const uint32_t* tp = a.lpos + a.loff[2 * j];
  const uint64_t* qkG = a.lkey + a.loff[2 * j + 1]; const uint32_t* qp = a.lpos + a.loff[2 * j + 1];
  const long nt = (long)(a.loff[2 * j + 1] - a.loff[2 * j]), nq = (long)(a.loff[2 * j + 2] - a.loff[2 * j + 1]);
  const bool staged = nt + nq <= ldsKeys;
  if (staged) {                                                          // the two lists are adjacent in lkey: one copy
    for (long x = lane; x < nt + nq; x += 64) skeys[x] = tkG[x];
    rs_wave_sync();
  }
  const RsBand bd = rs_band(a, i);
  const long maxFreq = a.maxFreqArr ? (long)a.maxFreqArr[i] : a.maxFreq;
  int4* rects = a.rects + a.loff[2 * j] + 2 * (uint64_t)j;               // nt + nq + 2 slots: every iteration of the walk emits at most one
  const long rcap = nt + nq + 2;
  uint32_t n = 0, nrect = 0;
  // The walk, once for keys staged in LDS and once for keys left in HBM: through one pointer that may be either, every read would be a flat_load.
  auto walk = [&](const uint64_t* tk, const uint64_t* qk) __attribute__((always_inline)) {
  auto rect = [&](long t0, long t1, long q0, long q1) {                   // ti in [t0, t1), qi in [q0, q1]
    if (nrect < rcap) { if (lane == 0) rects[nrect] = make_int4((int)t0, (int)t1, (int)q0, (int)q1); }
    else if (lane == 0) atomicOr(&a.status[i], (uint32_t)LRA_ST_CAPACITY);
    nrect++;
    const long w = q1 - q0 + 1, cells = (t1 - t0) * w;
    uint32_t c = 0;
    for (long x = lane; x < cells; x += 64) c += rs_pass(bd, tp, qp, q0 + x % w, t0 + x / w) ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    n += c;
  };
#define Qk(x) (qk[x] & FOR_MASK)
#define Tk(x) (tk[x] & FOR_MASK)
  // The walk's loops, 64 list entries per round (every lane holds the same cursors; the lists are sorted, the predicates of the searches monotone):
  // fwd(from, to, pred): the smallest x in [from, to) with !pred(x), else to  == `while (x < to && pred(x)) x++`
  auto fwd = [&](long from, long to, auto pred) -> long {
    long x = from;
    while (x < to) {
      const long idx = x + lane;
      const unsigned long long m = __ballot(!(idx < to && pred(idx)));
      if (m) return min(x + (long)(__ffsll((long long)m) - 1), to);
      x += 64;
    }
    return to;
  };
  // bwd(from, stop, pred): the largest x in (stop, from] with !pred(x), else stop  == `while (x > stop && pred(x)) x--`
  auto bwd = [&](long from, long stop, auto pred) -> long {
    long x = from;
    while (x > stop) {
      const long idx = x - lane;
      const unsigned long long m = __ballot(!(idx > stop && pred(idx)));
      if (m) return max(x - (long)(__ffsll((long long)m) - 1), stop);
      x -= 64;
    }
    return stop;
  };
  // lower(lo, hi, pred): the binary search `while (lo < hi) { mid; if (pred(mid)) lo = mid + 1; else hi = mid; }` for a monotone pred, 64 probes per round
  auto lower = [&](long lo, long hi, auto pred) -> long {
    while (hi - lo > 64) {
      const long step = (hi - lo + 63) / 64;
      const long idx = lo + step * (lane + 1) - 1;
      const int c = __popcll(__ballot(idx < hi && pred(idx)));            // the probes that hold: a prefix
      const long idxC = lo + step * (c + 1) - 1;                           // the first probe that does not (or lies past hi)
      lo = lo + step * c; hi = min(hi, idxC);
    }
    return fwd(lo, hi, pred);
  };
  if (nq > 0 && nt > 0) {
    long qs = 0, qe = nq - 1, ts = 0, te = nt;
    do {
      { const uint64_t t0 = Tk(ts); qs = fwd(qs, qe + 1, [&](long x) { return Qk(x) < t0; }); }   // :47-49
      if (qs >= qe) break;                                               // :51-53
      const uint64_t startGap = Qk(qs) - Tk(ts);
      if (te > ts) { const uint64_t t1 = Tk(te - 1); qe = bwd(qe, qs, [&](long x) { return Qk(x) > t1; }); }
      const uint64_t endGap = Tk(te - 1) - Qk(qe);
      if (startGap == 0 || (startGap & FOR_MASK) > (endGap & FOR_MASK)) {
        const long tsOrig = ts, qsOrig = qs;
        const uint64_t qv = Qk(qs);
        ts = lower(ts, te, [&](long x) { return Tk(x) < qv; });
        if (ts < te && Tk(ts) == qv) {
          const long tsStart = ts;
          const long tsi = fwd(ts, te, [&](long x) { return Tk(x) == qv; });
          const long qsStart = qs;
          qs = fwd(qs + 1, qe + 1, [&](long x) { return Qk(x) == qv; }) - 1;   // while (qs < qe && Qk(qs + 1) == Qk(qs)) qs++
          if (qs - qsStart < maxFreq && tsi > tsStart) rect(tsStart, tsi, qsStart, qs);   // for ti .. if (..) for qi .. emit(qi, ti)
        }
        { const uint64_t r = tk[tsOrig]; ts = fwd(ts, te, [&](long x) { return tk[x] == r; }); }   // :101 raw compare
        { const uint64_t r = qk[qsOrig]; qs = fwd(qs, qe, [&](long x) { return qk[x] == r; }); }   // :102
      } else {
        const uint64_t qv = Qk(qe);
        if (te != nt && Tk(te - 1) == qv) {
        } else {
          te = lower(ts, te, [&](long x) { return !(qv < Tk(x)); });
        }
        const long teStart = te;
        const long tei = bwd(te - 1, ts - 1, [&](long x) { return Tk(x) == qv; }) + 1;   // while (tei > ts && Tk(tei - 1) == Qk(qe)) tei--
        if (tei < teStart && teStart > 0) {
          const long qeStart = qe;
          qe = bwd(qe - 1, qs - 1, [&](long x) { return Qk(x) == qv; }) + 1;             // while (qe > qs && Qk(qe) == Qk(qe - 1)) qe--
          if (qeStart - qe < maxFreq) rect(tei, teStart, qe, qeStart);
        }
        te = tei;
      }
    } while (qs < qe && ts < te);
  }
#undef Qk
#undef Tk
  };
  if (staged) walk(skeys, skeys + nt); else walk(tkG, qkG);
  if (lane == 0) { a.cnt[i] = n; a.nrect[j] = nrect < rcap ? nrect : (uint32_t)rcap; }
}

// the pairs of the recorded rectangles, in the reference's order (rectangle by rectangle, target-major)
__global__ void __launch_bounds__(64) rs_emit(RsArgs a, int nLarge) {
  const int j = blockIdx.x, lane = threadIdx.x;
  if (j >= nLarge) return;
  const uint32_t i = a.largeIdx[j];
  const uint32_t* tp = a.lpos + a.loff[2 * j]; const uint32_t* qp = a.lpos + a.loff[2 * j + 1];
  const RsBand bd = rs_band(a, i);
  const int K = a.K[i];
  const uint32_t qAdd = a.q_add[i], tAdd = a.t_add[i], flip = a.flip[i];
  const int4* rects = a.rects + a.loff[2 * j] + 2 * (uint64_t)j;
  const uint32_t nrect = a.nrect[j];
  const uint64_t o = a.pair_off[i];
  const unsigned long long below = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));
  uint32_t n = 0;
  for (uint32_t r = 0; r < nrect; r++) {
    const int4 R = rects[r];
    const long w = (long)R.w - R.z + 1, cells = ((long)R.y - R.x) * w;
    for (long base = 0; base < cells; base += 64) {
      const long x = base + lane;
      const long qi = R.z + x % w, ti = R.x + x / w;
      const bool ok = x < cells && rs_pass(bd, tp, qp, qi, ti);
      const unsigned long long m = __ballot(ok);
      if (ok) {
        uint32_t pq = qp[qi] + qAdd;
        if (flip) pq = flip - pq - (uint32_t)K;
        const uint64_t at = o + n + __popcll(m & below);
        a.outQ[at] = pq; a.outT[at] = tp[ti] + tAdd;
      }
      n += (uint32_t)__popcll(m);
    }
  }
}

inline size_t sz(size_t n, size_t e) { return (n * e + 255) / 256 * 256; }

}  // namespace

static int refine_space_impl(lra_ctx* ctx, int n, const char* d_qseq, const uint64_t* d_q_off, const int32_t* d_q_len, const char* d_tseq,
                             const uint64_t* d_t_off, const int32_t* d_t_len, const uint32_t* d_t_span, const int32_t* d_K, const int32_t* d_W,
                             const int32_t* d_diag, const uint32_t* d_q_add, const uint32_t* d_t_add, const uint32_t* d_flip_len, int match,
                             int mismatch, int indel, int max_freq, const int32_t* d_max_freq, lra_refine_space_result* out) {
  if (!ctx || !out || n < 0) return LRA_ERR_INVALID;
  memset(out, 0, sizeof *out);
  out->n_problems = (uint64_t)n;
  if (n == 0) return LRA_OK;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t n1 = (size_t)n + 2;
  auto take = [](char*& p, size_t cnt, size_t e) { char* r = p; p += sz(cnt, e); return r; };
  size_t needW = sz(n1, 4) * 8 + sz(n1, 8) * 6 + 2 * (sz(2 * n1, 4) + sz(2 * n1 + 2, 8)) + 4096;
  char* w = (char*)lra_ensure(ctx, 14, needW);
  if (!w) return LRA_ERR_NOMEM;
  RsArgs a;
  memset(&a, 0, sizeof a);
  a.n = n; a.qseq = d_qseq; a.tseq = d_tseq; a.q_off = d_q_off; a.q_len = d_q_len; a.t_off = d_t_off; a.t_len = d_t_len; a.t_span = d_t_span; a.K = d_K; a.W = d_W;
  a.diag = d_diag; a.q_add = d_q_add; a.t_add = d_t_add; a.flip = d_flip_len; a.maxFreq = max_freq; a.maxFreqArr = d_max_freq;
  a.isSmall = (uint32_t*)take(w, n1, 4); a.smallIdx = (uint32_t*)take(w, n1, 4); a.largeIdx = (uint32_t*)take(w, n1, 4); a.sq_len = (int32_t*)take(w, n1, 4);
  a.st_len = (int32_t*)take(w, n1, 4); a.sk = (int32_t*)take(w, n1, 4); a.bcap = (uint32_t*)take(w, n1, 4); a.cnt = (uint32_t*)take(w, n1, 4);
  a.smallPos = (uint64_t*)take(w, n1, 8); a.sq_off = (uint64_t*)take(w, n1, 8); a.st_off = (uint64_t*)take(w, n1, 8);
  uint64_t* boff = (uint64_t*)take(w, n1, 8); uint64_t* pair_off = (uint64_t*)take(w, n1, 8); uint64_t* spare8 = (uint64_t*)take(w, n1, 8); (void)spare8;
  a.lcnt = (uint32_t*)take(w, 2 * n1, 4); uint64_t* loff = (uint64_t*)take(w, 2 * n1 + 2, 8);
  // results that outlive the call
  size_t needR = sz(n1, 8) + sz(n1, 4) * 2 + 4096;
  const unsigned g = (unsigned)((n + 255) / 256);
  float* identity; uint32_t* status; uint64_t* pairOffOut;
  {
    char* r0 = (char*)lra_ensure(ctx, 15, needR);
    if (!r0) return LRA_ERR_NOMEM;
    pairOffOut = (uint64_t*)take(r0, n1, 8); identity = (float*)take(r0, n1, 4); status = (uint32_t*)take(r0, n1, 4);
  }
  a.identity = identity; a.status = status;
  lra_time_begin(ctx, "refine_space");
  hipLaunchKernelGGL(rs_classify, dim3(g), dim3(256), 0, st, a);
  lra_time_end(ctx);
  { int rc = lra_exclusive_scan<uint32_t>(ctx, n, a.isSmall, a.smallPos); if (rc) return rc; }
  uint64_t nSmall64 = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&nSmall64, a.smallPos + n, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  const int nSmall = (int)nSmall64, nLarge = n - nSmall;
  hipLaunchKernelGGL(rs_split, dim3(g), dim3(256), 0, st, a);
  // ---- short gaps
  int32_t* blocks = nullptr;
  if (nSmall > 0) {
    { int rc = lra_exclusive_scan<uint32_t>(ctx, nSmall, a.bcap, boff); if (rc) return rc; }
    uint64_t totalCap = 0;
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(&totalCap, boff + nSmall, 8, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
    size_t needB = sz(3 * totalCap + 3, 4) + sz((size_t)nSmall + 1, 4) * 3 + 4096;
    char* wb = (char*)lra_ensure(ctx, 16, needB);
    if (!wb) return LRA_ERR_NOMEM;
    blocks = (int32_t*)take(wb, 3 * totalCap + 3, 4);
    int32_t* score = (int32_t*)take(wb, (size_t)nSmall + 1, 4); int32_t* nblocks = (int32_t*)take(wb, (size_t)nSmall + 1, 4);
    int32_t* ast = (int32_t*)take(wb, (size_t)nSmall + 1, 4);
    int rc = lra_aog_launch_device(ctx, nSmall, d_qseq, d_tseq, a.sq_off, a.sq_len, a.st_off, a.st_len, a.sk, match, mismatch, indel, score, nblocks, blocks,
                                   boff, ast);
    if (rc) return rc;
    a.boff = boff; a.nblocks = nblocks; a.blocks = blocks; a.aogStatus = ast;
    lra_time_begin(ctx, "refine_space");
    hipLaunchKernelGGL(rs_small<false>, dim3((nSmall + 63) / 64), dim3(64), 0, st, a, nSmall);
    lra_time_end(ctx);
  }
  // ---- long gaps
  if (nLarge > 0) {
    // one serial pass per list into capacity-spaced slots (a list cannot hold more tuples than k-mer positions), then a parallel compaction
    uint32_t* lcap = (uint32_t*)take(w, 2 * n1, 4); uint64_t* capOff = (uint64_t*)take(w, 2 * n1 + 2, 8);
    hipLaunchKernelGGL(rs_caps, dim3((2 * nLarge + 255) / 256), dim3(256), 0, st, a, nLarge, lcap);
    { int rc = lra_exclusive_scan<uint32_t>(ctx, 2 * nLarge, lcap, capOff); if (rc) return rc; }
    uint64_t totalCap = 0;
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(&totalCap, capOff + 2 * nLarge, 8, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
    char* wr = (char*)lra_ensure(ctx, 66, sz(totalCap + 1, 8) + sz(totalCap + 1, 4) + 4096);
    if (!wr) return LRA_ERR_NOMEM;
    uint64_t* rawKey = (uint64_t*)take(wr, totalCap + 1, 8); uint32_t* rawPos = (uint32_t*)take(wr, totalCap + 1, 4);
    a.lkey = rawKey; a.lpos = rawPos; a.loff = capOff;
    lra_time_begin(ctx, "rs_long_sketch");
    hipLaunchKernelGGL(rs_sketch<true>, dim3(2 * nLarge), dim3(64), 0, st, a, nLarge);
    lra_time_end(ctx);
    { int rc = lra_exclusive_scan<uint32_t>(ctx, 2 * nLarge, a.lcnt, loff); if (rc) return rc; }
    uint64_t totalMm = 0;
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(&totalMm, loff + 2 * nLarge, 8, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (getenv("LRA_RS_DBG")) fprintf(stderr, "[rs] n %d nLarge %d minimizers %llu\n", n, nLarge, (unsigned long long)totalMm);
    char* wl = (char*)lra_ensure(ctx, 17, sz(totalMm + 1, 8) + sz(totalMm + 1, 4) + sz(totalMm + 2 * (size_t)nLarge + 4, 16) + sz((size_t)nLarge + 1, 4) + 4096);
    if (!wl) return LRA_ERR_NOMEM;
    a.lkey = (uint64_t*)take(wl, totalMm + 1, 8); a.lpos = (uint32_t*)take(wl, totalMm + 1, 4); a.loff = loff;
    a.rects = (int4*)take(wl, totalMm + 2 * (size_t)nLarge + 4, 16); a.nrect = (uint32_t*)take(wl, (size_t)nLarge + 1, 4);
    hipLaunchKernelGGL(rs_compact, dim3(2 * nLarge), dim3(64), 0, st, 2 * nLarge, capOff, loff, rawKey, rawPos, a.lkey, a.lpos);
    { int rc = lra_sort_minimizers_batch(ctx, 2 * nLarge, loff, a.lkey, a.lpos); if (rc) return rc; }   // sort(EndGenomeTup), sort(EndReadTup)  :306,:308
    lra_time_begin(ctx, "rs_long_compare");
    static const int ldsKeys = getenv("LRA_RS_LDS_KEYS") ? std::max(0, std::min(CMP_LDS_KEYS, atoi(getenv("LRA_RS_LDS_KEYS")))) : 1536;   // (measured, headline batch, lists of 2-4 k keys per gap: 7168 keys of LDS per wave 54 ms, 3072: 50, 1536: 49, 512: 52 -- the waves per CU matter more than where the keys sit)
    hipLaunchKernelGGL(rs_compare, dim3(nLarge), dim3(64), (size_t)ldsKeys * 8, st, a, nLarge, ldsKeys);
    lra_time_end(ctx);
  }
  // ---- pairs
  { int rc = lra_exclusive_scan<uint32_t>(ctx, n, a.cnt, pair_off); if (rc) return rc; }
  uint64_t nPairs = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&nPairs, pair_off + n, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  char* r = (char*)lra_ensure(ctx, 11, sz(nPairs + 1, 4) * 2 + 4096);
  if (!r) return LRA_ERR_NOMEM;
  uint32_t* outQ = (uint32_t*)take(r, nPairs + 1, 4); uint32_t* outT = (uint32_t*)take(r, nPairs + 1, 4);
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(pairOffOut, pair_off, ((size_t)n + 1) * 8, hipMemcpyDeviceToDevice, st));
  a.pair_off = pairOffOut; a.outQ = outQ; a.outT = outT;
  lra_time_begin(ctx, "refine_space");
  if (nSmall > 0) hipLaunchKernelGGL(rs_small<true>, dim3((nSmall + 63) / 64), dim3(64), 0, st, a, nSmall);
  lra_time_end(ctx);
  if (nLarge > 0) {
    lra_time_begin(ctx, "rs_long_compare");
    hipLaunchKernelGGL(rs_emit, dim3(nLarge), dim3(64), 0, st, a, nLarge);
    lra_time_end(ctx);
  }
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  LRA_HIP_CHECK(ctx, hipGetLastError());
  out->n_pairs = nPairs; out->n_small = (uint64_t)nSmall; out->d_pair_off = pairOffOut; out->d_pair_q = outQ; out->d_pair_t = outT; out->d_identity = identity;
  out->d_status = status;
  return LRA_OK;
}

extern "C" int lra_refine_space_batch(lra_ctx* ctx, int n, const char* d_qseq, const uint64_t* d_q_off, const int32_t* d_q_len, const char* d_tseq,
                                      const uint64_t* d_t_off, const int32_t* d_t_len, const uint32_t* d_t_span, const int32_t* d_K, const int32_t* d_W,
                                      const int32_t* d_diag, const uint32_t* d_q_add, const uint32_t* d_t_add, const uint32_t* d_flip_len, int match,
                                      int mismatch, int indel, int max_freq, lra_refine_space_result* out) {
  return refine_space_impl(ctx, n, d_qseq, d_q_off, d_q_len, d_tseq, d_t_off, d_t_len, d_t_span, d_K, d_W, d_diag, d_q_add, d_t_add, d_flip_len, match, mismatch,
                           indel, max_freq, nullptr, out);
}

// the same with opts.localMaxFreq per problem (RefinedAlignmentbtwnAnchors raises it to 50 for spaces under 500, LocalRefineAlignment.h:272-277)
extern "C" int lra_refine_space_batch_mf(lra_ctx* ctx, int n, const char* d_qseq, const uint64_t* d_q_off, const int32_t* d_q_len, const char* d_tseq,
                                         const uint64_t* d_t_off, const int32_t* d_t_len, const uint32_t* d_t_span, const int32_t* d_K, const int32_t* d_W,
                                         const int32_t* d_diag, const uint32_t* d_q_add, const uint32_t* d_t_add, const uint32_t* d_flip_len, int match,
                                         int mismatch, int indel, const int32_t* d_max_freq, lra_refine_space_result* out) {
  if (!d_max_freq) return LRA_ERR_INVALID;
  return refine_space_impl(ctx, n, d_qseq, d_q_off, d_q_len, d_tseq, d_t_off, d_t_len, d_t_span, d_K, d_W, d_diag, d_q_add, d_t_add, d_flip_len, match, mismatch,
                           indel, 0, d_max_freq, out);
}
