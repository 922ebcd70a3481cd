// lra_amd/csrc/seed.hip -- tier-1 seeding of a read batch on gfx950.
//
// Replaces, for a whole batch of reads resident in HBM, the first stages of MapRead
// (reference: MapRead.h:169-203):
//   a1  StoreMinimizers<GenomeTuple,Tuple>(read, globalK, globalW)        MinCount.h:8-179
//   a2  sort(readmm.begin(), readmm.end())                                MapRead.h:185
//   a3  CompareLists<GenomeTuple,Tuple>(readmm, genomemm, allMatches)     CompareLists.h:9-146
//   a4  SeparateMatchesByStrand(read, genome, k, allMatches, for, rev)    MapRead.h:109-150
//
// Layout in HBM (structure of arrays, CSR by read):
//   reads    : uint8 seq[], uint64 read_off[n+1]
//   minimizers: uint64 mm_key[] (strand in bit 63, TupleOps.h:68), uint32 mm_pos[], uint64 mm_off[n+1]
//   index    : uint64 idx_key[] sorted by (key & 2^63-1), uint32 idx_pos[]   (the .mms payload, MMIndex.h:416)
//   matches  : uint32 match_qi[] (index into the read's sorted minimizers), uint32 match_ti[]
//              (index into the global index), uint64 match_off[n+1], in the reference's
//              discovery order; then per read forward-strand matches first, reverse after:
//              uint32 sep_qpos[], sep_tpos[], uint32 n_forward[n].
//
// Why these stages are emulated step by step instead of re-derived: their outputs depend
// on implementation details -- the unmasked first-window comparison and ring-index
// tie-break of the minimizer scan, the permutation libstdc++'s introsort leaves among
// equal keys (MapRead.h:185 sorts with a non-total order), and CompareLists' alternating
// two-ended walk with raw-key run skipping (which re-emits or drops pairs depending on
// that permutation).  Parallelism is across reads (one lane per read for the serial
// state machines) and across query tuples (one lane per tuple for the index searches,
// which are the HBM-heavy part).
#include "common.h"
#include <rocprim/rocprim.hpp>
#include "scan.h"

namespace {

constexpr uint64_t FOR_MASK = ~(1ULL << 63);  // lra.cpp:1008-1012
constexpr uint64_t REV_MASK = (1ULL << 63);
constexpr int MAX_W = 32;
constexpr int SORT_CAP = 9000;   // tuples per read the LDS sort holds (16 B each + tables <= 160 KiB)

__device__ __forceinline__ int code_n(unsigned char c) {  // SeqUtils.h:42 (seqMapN)
  if (c < 8) return c & 3;
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
  }
}
__device__ __forceinline__ uint64_t code2(unsigned char c) {  // SeqUtils.h:7 (seqMap)
  int v = code_n(c);
  return v > 3 ? 0 : (uint64_t)v;
}

// ------------------------------------------------------------------------------------ a1
// One lane per read; the state machine of MinCount.h:8-179 verbatim in behaviour.
// EMIT=false only counts.  The w-entry ring lives in LDS, lane-interleaved.
template <bool EMIT>
__global__ void __launch_bounds__(64) sketch_kernel(int n_reads, const unsigned char* __restrict__ seq_all,
                                                    const uint64_t* __restrict__ read_off, int k, int w,
                                                    const uint64_t* __restrict__ mm_off, uint64_t* __restrict__ mm_key,
                                                    uint32_t* __restrict__ mm_pos, uint32_t* __restrict__ counts,
                                                    const int* __restrict__ only_flagged) {
  __shared__ uint64_t ringT[MAX_W * 64];
  __shared__ uint32_t ringP[MAX_W * 64];
  const int lane = threadIdx.x;
  const int r = blockIdx.x * 64 + lane;
  if (r >= n_reads) return;
  if (only_flagged && !only_flagged[r]) return;
  const unsigned char* seq = seq_all + read_off[r];
  const uint32_t seqLen = (uint32_t)(read_off[r + 1] - read_off[r]);
  uint64_t* okey = EMIT ? mm_key + mm_off[r] : nullptr;
  uint32_t* opos = EMIT ? mm_pos + mm_off[r] : nullptr;
  uint32_t n = 0;
#define SK_EMIT(T_, P_) do { if (EMIT) { okey[n] = (T_); opos[n] = (P_); } n++; } while (0)
#define SK_DONE() do { if (!EMIT) counts[r] = n; return; } while (0)
  const int span = w + k - 1;                                          // :17
  if (seqLen < (uint32_t)k || seqLen < (uint32_t)span) SK_DONE();      // :12,:26
  const uint64_t kmask = (k >= 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
  long nvStart = 0, nvEnd = 0;
  bool valid = false;
  auto find_valid = [&]() -> bool {                                    // :27-41 / :117-131
    valid = false;
    while ((uint32_t)nvStart < seqLen - (uint32_t)span && !valid) {
      valid = true;
      for (long x = nvStart; valid && x < nvStart + span; x++)
        if (code_n(seq[x]) > 3) { nvStart = x + 1; valid = false; }
    }
    return valid;
  };
  if (!find_valid()) SK_DONE();
  nvEnd = nvStart + span;
  uint64_t cur = 0, rc = 0;
  for (int p = 0; p < k; p++) cur = (cur << 2) + code2(seq[p]);        // StoreTuple TupleOps.h:104
  {
    uint64_t a = cur;                                                  // TupleRC TupleOps.h:125
    for (int i = 0; i < k; i++) { rc = (rc << 2) + ((~a) & 3ULL); a >>= 2; }
  }
  auto canon = [&]() -> uint64_t {
    return ((cur & FOR_MASK) < (rc & FOR_MASK)) ? (cur & FOR_MASK) : (rc | REV_MASK);
  };
  auto shift = [&](uint32_t at) {                                      // TupleOps.h:114-123
    uint64_t c = code2(seq[at]);
    cur = ((cur << 2) & kmask) + c;
    rc = (rc >> 2) + (((~c) & 3ULL) << (2 * ((uint64_t)k - 1)));
  };
  uint64_t actT = canon();
  uint32_t actP = 0;
  ringT[0 * 64 + lane] = actT; ringP[0 * 64 + lane] = 0;
  uint32_t p;
  const uint32_t nk = seqLen - k + 1;
  for (p = 1; p < (uint32_t)w && p < nk; p++) {                        // :77-96
    shift(p + k - 1);
    uint64_t c = canon();
    if (c < actT) { actT = c; actP = p; }                              // unmasked compare (:91)
    ringT[(p % w) * 64 + lane] = c; ringP[(p % w) * 64 + lane] = p;
  }
  if (nvEnd == span) SK_EMIT(actT, actP);                              // :100-102
  uint32_t slot = (uint32_t)w % (uint32_t)w;                           // p % w, advanced incrementally
  for (p = w; p < nk; p++) {                                           // :105-178
    if (nvEnd == (long)(p + k - 1)) {
      if (code_n(seq[p + k - 1]) <= 3) nvEnd++;
      else {
        nvStart = p + k;
        if (!find_valid()) SK_DONE();
        nvEnd = nvStart + span;
      }
    }
    shift(p + k - 1);
    uint64_t c = canon();
    ringT[slot * 64 + lane] = c; ringP[slot * 64 + lane] = p;
    if (++slot == (uint32_t)w) slot = 0;
    if (p - w >= actP) {                                               // :148-162 re-scan, ring order
      actT = ringT[lane]; actP = ringP[lane];
      for (int j = 1; j < w; j++) {
        uint64_t t = ringT[j * 64 + lane];
        if ((t & FOR_MASK) < (actT & FOR_MASK)) { actT = t; actP = ringP[j * 64 + lane]; }
      }
      if (nvEnd == (long)(p + k)) SK_EMIT(actT, actP);
    } else if ((c & FOR_MASK) < (actT & FOR_MASK)) {                   // :164-173
      actT = c; actP = p;
      if (nvEnd == (long)(p + k)) SK_EMIT(actT, actP);
    }
  }
  SK_DONE();
#undef SK_EMIT
#undef SK_DONE
}

// ---- a1, fast path: one WAVE per read (reads without non-ACGT bytes; the others are flagged
// for the serial kernel above).  64 positions per tile:
//   * 2-bit codes of the tile are packed with two ballots; every lane cuts its own k-mer out of
//     the packed words and derives forward / reverse-complement keys with bit tricks;
//   * the serial state of MinCount.h is only "which position is the active minimizer".  After
//     position 2w-1 the active one is always a minimum (by masked key) of the current window
//     (every element that entered after the first window was compared against it on entry), so
//       act(p) = p                      if key[p] <  min of the previous window
//              = ring-order argmin(p)   if act(p-1) just left the window      (:148-154)
//              = act(p-1)               otherwise,
//     and a tuple is emitted in the first two cases (:155-172).  Wherever the window minimum is
//     unique act(p) is that position regardless of history, so lanes resolve independently and
//     only lanes inside a run of tied minima are walked in order;
//   * positions below 2w (first window chosen with the UNMASKED comparison, :91) are replayed
//     literally by the whole wave on uniform values.
__device__ __forceinline__ uint64_t spread32(uint64_t x) {
  x = (x | (x << 16)) & 0x0000FFFF0000FFFFULL;
  x = (x | (x << 8)) & 0x00FF00FF00FF00FFULL;
  x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0FULL;
  x = (x | (x << 2)) & 0x3333333333333333ULL;
  x = (x | (x << 1)) & 0x5555555555555555ULL;
  return x;
}

// One pass: a read's minimizers are written where the read's bases begin in a staging array as long as the batch's bases (a read of L bases has at most L - k + 1
// of them), its count and whether it holds an N come out with them; sketch_compact moves the lists to their places once the counts are scanned.  (Two passes --
// count, scan, emit -- ran the whole window machine twice: 13.7 + 12.9 ms.)
template <bool EMIT>
__global__ void __launch_bounds__(64) sketch_wave_kernel(int n_reads, const unsigned char* __restrict__ seq_all,
                                                         const uint64_t* __restrict__ read_off, int k, int w,
                                                         const uint64_t* __restrict__ mm_off, uint64_t* __restrict__ mm_key,
                                                         uint32_t* __restrict__ mm_pos, uint32_t* __restrict__ counts, int* flagN) {
  __shared__ uint64_t kbuf[128];
  const int lane = threadIdx.x;
  const uint64_t kbits = (k >= 32) ? 0xFFFFFFFFULL : ((1ULL << k) - 1);
  const uint64_t mask2k = (k >= 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
  const unsigned long long below = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));
  for (int r = blockIdx.x; r < n_reads; r += gridDim.x) {
    const unsigned char* seq = seq_all + read_off[r];
    const uint32_t seqLen = (uint32_t)(read_off[r + 1] - read_off[r]);
    uint64_t* okey = EMIT ? mm_key + mm_off[r] : nullptr;
    uint32_t* opos = EMIT ? mm_pos + mm_off[r] : nullptr;
    uint32_t nout = 0;
    const int span = w + k - 1;
    if (seqLen < (uint32_t)k || seqLen <= (uint32_t)span) { if (lane == 0) { counts[r] = 0; flagN[r] = 0; } continue; }   // :12,:26-27
    const uint32_t nk = seqLen - k + 1;
    const uint32_t P0 = min((uint32_t)(2 * w), nk);
    bool hasN = false;
    uint64_t carry_m = 0; uint32_t carry_act = 0;
    // the bases come in 64-byte pieces, one per tile, asked for two tiles ahead: piece j is tile j's own bases and (its first k - 1) the tail of tile j - 1's k-mers;
    // a tile's step is otherwise a chain that begins with a load from HBM and nothing to do until it is back
    unsigned char pc0 = lane < (int)seqLen ? seq[lane] : (unsigned char)'A', pc1 = 64u + lane < seqLen ? seq[64 + lane] : (unsigned char)'A';
    for (uint32_t B = 0; B < nk; B += 64) {
      // ---- keys of positions B..B+63
      const uint32_t p = B + lane;
      const unsigned char pn = p + 128 < seqLen ? seq[p + 128] : (unsigned char)'A';
      int c0 = 0, c1 = 0;
      if (p < seqLen) { int c = code_n(pc0); hasN |= c > 3; c0 = c & 3; }
      if (lane < k - 1 && p + 64 < seqLen) { int c = code_n(pc1); hasN |= c > 3; c1 = c > 3 ? 0 : c; }
      pc0 = pc1; pc1 = pn;
      const unsigned long long b0 = __ballot(c0 & 1), b1 = __ballot(c0 & 2), t0 = __ballot(c1 & 1), t1 = __ballot(c1 & 2);
      uint64_t x0 = b0 >> lane, x1 = b1 >> lane;
      if (lane) { x0 |= t0 << (64 - lane); x1 |= t1 << (64 - lane); }
      x0 &= kbits; x1 &= kbits;
      const uint64_t LE = spread32(x0) | (spread32(x1) << 1);
      const uint64_t rc = (~LE) & mask2k;
      uint64_t v = __brevll(LE);
      v = ((v >> 1) & 0x5555555555555555ULL) | ((v & 0x5555555555555555ULL) << 1);
      const uint64_t fwd = (k >= 32) ? v : (v >> (64 - 2 * k));
      const uint64_t key = ((fwd & FOR_MASK) < (rc & FOR_MASK)) ? (fwd & FOR_MASK) : (rc | REV_MASK);   // :60-61
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      kbuf[p & 127] = key;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      if (B == 0) {
        // ---- literal replay of positions 0..P0-1 on uniform values (all lanes identical)
        uint64_t actT = kbuf[0]; uint32_t actP = 0;
        for (uint32_t q = 1; q < (uint32_t)w && q < nk; q++) { uint64_t c = kbuf[q]; if (c < actT) { actT = c; actP = q; } }   // :77-96
        if (lane == 0 && EMIT) { okey[nout] = actT; opos[nout] = actP; }                                                       // :100-102
        nout++;
        for (uint32_t q = w; q < P0; q++) {                                                                                  // :105-178
          const uint64_t c = kbuf[q];
          bool em = false;
          if (q - w >= actP) {
            // ring slot j holds the newest position == j (mod w)
            uint64_t bt = 0; uint32_t bp = 0;
            for (int j = 0; j < w; j++) {
              uint32_t x = q - ((q - (uint32_t)j) % (uint32_t)w);
              uint64_t t = kbuf[x];
              if (j == 0 || (t & FOR_MASK) < (bt & FOR_MASK)) { bt = t; bp = x; }
            }
            actT = bt; actP = bp; em = true;
          } else if ((c & FOR_MASK) < (actT & FOR_MASK)) { actT = c; actP = q; em = true; }
          if (em) { if (lane == 0 && EMIT) { okey[nout] = actT; opos[nout] = actP; } nout++; }
        }
        carry_m = actT & FOR_MASK; carry_act = actP;
      }
      // ---- positions >= P0 of this tile, in parallel
      const bool live = p >= P0 && p < nk;
      uint64_t bk = key & FOR_MASK; uint32_t bpos = p; int br = (int)(p % (uint32_t)w), cnt = 1;
      if (live) {
        int rx = br;
        for (int d = 1; d < w; d++) {
          rx = (rx == 0) ? w - 1 : rx - 1;
          const uint32_t x = p - d;
          const uint64_t kx = kbuf[x & 127] & FOR_MASK;
          if (kx < bk) { bk = kx; bpos = x; br = rx; cnt = 1; }
          else if (kx == bk) { cnt++; if (rx < br) { br = rx; bpos = x; } }
        }
      }
      // min of the previous window = lane-1's window min (lane 0: carried over)
      uint64_t mprev = __shfl_up(bk, 1);
      const bool prevLive = lane > 0 && (p - 1) >= P0;
      if (!prevLive) mprev = carry_m;           // first live lane of the read, or lane 0 of a later tile
      const bool strictNew = live && (key & FOR_MASK) < mprev;
      uint32_t A = (cnt == 1 || strictNew) ? (strictNew ? p : bpos) : 0xFFFFFFFFu;
      // resolve lanes whose window minimum is tied, in order
      unsigned long long unk = __ballot(live && A == 0xFFFFFFFFu);
      while (unk) {
        const int l = __ffsll((long long)unk) - 1;
        unk &= unk - 1;
        uint32_t pa = __shfl(A, (l + 63) & 63);
        const uint32_t pl = B + l;
        if (l == 0 || pl - 1 < P0) pa = carry_act;
        const uint32_t posR = __shfl(bpos, l);
        const uint32_t a = (pa == pl - w) ? posR : pa;
        if (lane == l) A = a;
      }
      uint32_t prevA = __shfl_up(A, 1);
      if (!prevLive) prevA = carry_act;
      const bool em = live && (prevA == p - (uint32_t)w || strictNew);
      const unsigned long long me = __ballot(em);
      if (EMIT && em) {
        const uint32_t o = nout + __popcll(me & below);
        okey[o] = kbuf[A & 127]; opos[o] = A;
      }
      nout += __popcll(me);
      // carries: state after the last live position of this tile
      const unsigned long long ml = __ballot(live);
      if (ml) {
        const int last = 63 - __clzll((long long)ml);
        carry_m = __shfl(bk, last); carry_act = __shfl(A, last);
      }
      __builtin_amdgcn_wave_barrier();
    }
    const bool anyN = __ballot(hasN) != 0;
    if (lane == 0) { counts[r] = nout; flagN[r] = anyN ? 1 : 0; }
  }
}

// the lists from the staging array (at the reads' base offsets) to their places; a read with an N is left to sketch_kernel
__global__ void __launch_bounds__(256) sketch_compact(int n_reads, const uint64_t* __restrict__ read_off, const uint64_t* __restrict__ mm_off, const int* __restrict__ flagN,
                                                      const uint64_t* __restrict__ wkey, const uint32_t* __restrict__ wpos, uint64_t* __restrict__ mm_key,
                                                      uint32_t* __restrict__ mm_pos) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int r = blockIdx.x * 4 + wv; r < n_reads; r += gridDim.x * 4) {
    if (flagN[r]) continue;
    const uint64_t src = read_off[r], dst = mm_off[r];
    const uint32_t n = (uint32_t)(mm_off[r + 1] - dst);
    for (uint32_t i = lane; i < n; i += 64) { mm_key[dst + i] = wkey[src + i]; mm_pos[dst + i] = wpos[src + i]; }
  }
}

// ------------------------------------------------------------------------------------ a2
// libstdc++ std::sort (bits/stl_algo.h: __introsort_loop + __final_insertion_sort, threshold
// 16, median-of-three to first, unguarded Hoare partition, heap-sort fall-back) restated on
// (key,pos) pairs with the reference's masked comparison (TupleOps.h:76).  The permutation of
// equal keys it leaves is what MapRead.h:185 produces, and CompareLists depends on it.
struct SortSeg {
  uint64_t* k; uint32_t* p;
  __device__ __forceinline__ bool lt(long a, long b) const { return (k[a] & FOR_MASK) < (k[b] & FOR_MASK); }
  __device__ __forceinline__ void swap(long a, long b) {
    uint64_t tk = k[a]; k[a] = k[b]; k[b] = tk;
    uint32_t tp = p[a]; p[a] = p[b]; p[b] = tp;
  }
  __device__ __forceinline__ void mv(long dst, long src) { k[dst] = k[src]; p[dst] = p[src]; }
};

__device__ void adjust_heap(SortSeg& s, long first, long hole, long len, uint64_t vk, uint32_t vp) {
  const long top = hole;
  long child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (s.lt(first + child, first + (child - 1))) child--;
    s.mv(first + hole, first + child);
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    s.mv(first + hole, first + (child - 1));
    hole = child - 1;
  }
  long parent = (hole - 1) / 2;                                        // __push_heap
  while (hole > top && (s.k[first + parent] & FOR_MASK) < (vk & FOR_MASK)) {
    s.mv(first + hole, first + parent);
    hole = parent;
    parent = (hole - 1) / 2;
  }
  s.k[first + hole] = vk; s.p[first + hole] = vp;
}

__device__ void heap_sort(SortSeg& s, long first, long last) {         // __partial_sort(first,last,last)
  long len = last - first;
  if (len >= 2) {                                                      // __make_heap
    long parent = (len - 2) / 2;
    while (true) {
      uint64_t vk = s.k[first + parent]; uint32_t vp = s.p[first + parent];
      adjust_heap(s, first, parent, len, vk, vp);
      if (parent == 0) break;
      parent--;
    }
  }
  while (last - first > 1) {                                           // __sort_heap / __pop_heap
    --last;
    uint64_t vk = s.k[last]; uint32_t vp = s.p[last];
    s.mv(last, first);
    adjust_heap(s, first, 0, last - first, vk, vp);
  }
}

__device__ __forceinline__ void unguarded_linear_insert(SortSeg& s, long last) {
  uint64_t vk = s.k[last]; uint32_t vp = s.p[last];
  long next = last - 1;
  while ((vk & FOR_MASK) < (s.k[next] & FOR_MASK)) { s.mv(last, next); last = next; --next; }
  s.k[last] = vk; s.p[last] = vp;
}

__device__ void insertion_sort(SortSeg& s, long first, long last) {
  if (first == last) return;
  for (long i = first + 1; i != last; ++i) {
    if (s.lt(i, first)) {
      uint64_t vk = s.k[i]; uint32_t vp = s.p[i];
      for (long x = i; x > first; --x) s.mv(x, x - 1);                 // move_backward
      s.k[first] = vk; s.p[first] = vp;
    } else unguarded_linear_insert(s, i);
  }
}

__global__ void __launch_bounds__(64) sort_kernel(int n_reads, const uint64_t* __restrict__ mm_off, uint64_t* mm_key, uint32_t* mm_pos,
                                                  const int* __restrict__ only_flagged) {
  const int r = blockIdx.x * 64 + threadIdx.x;
  if (r >= n_reads) return;
  if (only_flagged && !only_flagged[r]) return;
  SortSeg s{mm_key + mm_off[r], mm_pos + mm_off[r]};
  const long n = (long)(mm_off[r + 1] - mm_off[r]);
  if (n < 2) return;
  // __introsort_loop with an explicit stack (segments are disjoint, so their order is free)
  int sp = 0;
  long stF[64], stL[64]; int stD[64];
  int lg = 63 - __clzll((unsigned long long)n);
  stF[0] = 0; stL[0] = n; stD[0] = 2 * lg; sp = 1;
  while (sp > 0) {
    --sp;
    long first = stF[sp], last = stL[sp]; int depth = stD[sp];
    while (last - first > 16) {
      if (depth == 0) { heap_sort(s, first, last); break; }
      --depth;
      long mid = first + (last - first) / 2;
      long a = first + 1, b = mid, c = last - 1;                       // __move_median_to_first
      if (s.lt(a, b)) {
        if (s.lt(b, c)) s.swap(first, b);
        else if (s.lt(a, c)) s.swap(first, c);
        else s.swap(first, a);
      } else if (s.lt(a, c)) s.swap(first, a);
      else if (s.lt(b, c)) s.swap(first, c);
      else s.swap(first, b);
      long f = first + 1, l = last;                                    // __unguarded_partition
      const uint64_t pv = s.k[first] & FOR_MASK;
      while (true) {
        while ((s.k[f] & FOR_MASK) < pv) ++f;
        --l;
        while (pv < (s.k[l] & FOR_MASK)) --l;
        if (!(f < l)) break;
        s.swap(f, l);
        ++f;
      }
      if (sp < 64) { stF[sp] = f; stL[sp] = last; stD[sp] = depth; sp++; }
      last = f;
    }
  }
  if (n > 16) {                                                        // __final_insertion_sort
    insertion_sort(s, 0, 16);
    for (long i = 16; i != n; ++i) unguarded_linear_insert(s, i);
  } else insertion_sort(s, 0, n);
}

// ---- a2, fast path: the same std::sort, one 256-thread workgroup per read, data in LDS.
// Two facts make introsort data-parallel without changing its result:
//  (1) the segments the loop recurses into are disjoint, so they can be processed level by
//      level, all segments of a level at once;
//  (2) libstdc++'s unguarded Hoare partition of [first+1,last) around *first is a closed form:
//      with a_1<a_2<... the positions holding x >= pivot (ascending) and b_1>b_2>... those
//      holding x <= pivot (descending), it swaps (a_i,b_i) for i <= m = #{i: a_i < b_i} and
//      returns cut = min(a_{m+1}, b_m).  Ranks come from prefix counts, swaps are independent.
// The final insertion sort is stable, and the <=16-element leftovers are mutually ordered, so
// it equals a stable insertion sort of every leftover block on its own.
constexpr int SORT_NT = 1024;
constexpr int SORT_NW = SORT_NT / 64;

template <typename IT>
struct LdsSeg {                     // (key, original index) pairs in LDS (or, for the larger size classes, in a per-workgroup global scratch)
  uint64_t* k; IT* p;
  __device__ __forceinline__ bool lt(int a, int b) const { return (k[a] & FOR_MASK) < (k[b] & FOR_MASK); }
  __device__ __forceinline__ void swap(int a, int b) {
    uint64_t tk = k[a]; k[a] = k[b]; k[b] = tk;
    IT tp = p[a]; p[a] = p[b]; p[b] = tp;
  }
  __device__ __forceinline__ void mv(int dst, int src) { k[dst] = k[src]; p[dst] = p[src]; }
};

template <typename IT>
__device__ void lds_adjust_heap(LdsSeg<IT>& s, int first, int hole, int len, uint64_t vk, IT vp) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (s.lt(first + child, first + (child - 1))) child--;
    s.mv(first + hole, first + child);
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    s.mv(first + hole, first + (child - 1));
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && (s.k[first + parent] & FOR_MASK) < (vk & FOR_MASK)) {
    s.mv(first + hole, first + parent);
    hole = parent;
    parent = (hole - 1) / 2;
  }
  s.k[first + hole] = vk; s.p[first + hole] = vp;
}

template <typename IT>
__device__ void lds_heap_sort(LdsSeg<IT>& s, int first, int last) {
  int len = last - first;
  if (len >= 2) {
    int parent = (len - 2) / 2;
    while (true) {
      uint64_t vk = s.k[first + parent]; IT vp = s.p[first + parent];
      lds_adjust_heap(s, first, parent, len, vk, vp);
      if (parent == 0) break;
      parent--;
    }
  }
  while (last - first > 1) {
    --last;
    uint64_t vk = s.k[last]; IT vp = s.p[last];
    s.mv(last, first);
    lds_adjust_heap(s, first, 0, last - first, vk, vp);
  }
}

template <typename IT>
__device__ void lds_insertion_sort(LdsSeg<IT>& s, int first, int last) {   // == __insertion_sort on an isolated block
  for (int i = first + 1; i < last; ++i) {
    uint64_t vk = s.k[i]; IT vp = s.p[i];
    int j = i;
    while (j > first && (vk & FOR_MASK) < (s.k[j - 1] & FOR_MASK)) { s.mv(j, j - 1); --j; }
    s.k[j] = vk; s.p[j] = vp;
  }
}

// MODE 1 (BIG): the same algorithm for lists of up to 65534 tuples (beyond the LDS capacity; the value order of a large read's sparse-DP fragments, the point
// lists of a large read, the k-mer lists of a 30 kb gap): the element arrays live in a per-workgroup global scratch, only the segment tables stay in
// LDS; it takes the lists the LDS kernel flagged and clears the flag of those it sorted.
// MODE 2 (HUGE): lists beyond that (a 1 Mb assembly contig has ~180 k minimizers; one lane per list with the literal introsort took 19.9 of the 24.3 s of a
// 256-contig -CONTIG batch): 32-bit indices, the segment tables in the global scratch as well, the prefix counts as two 32-bit halves of a 64-bit word.
// MODE 1 also reports what it had to leave behind (stat[0] = the longest such list, stat[1] = how many), so that the host sizes MODE 2's scratch and
// launches it -- and the one-lane-per-list kernel behind it -- only when there is something to do.
template <int MODE> struct SortTypes { typedef unsigned short IT; typedef uint32_t TT; };
template <> struct SortTypes<2> { typedef uint32_t IT; typedef uint64_t TT; };
// bytes of per-workgroup global scratch for lists of up to cap tuples (MODE 1: elements only; MODE 2: elements + tables + start bits)
__host__ __device__ inline size_t sort_scratch_bytes(int mode, size_t cap) {
  const size_t isz = mode == 2 ? 4 : 2, maxseg = (cap / 16 + 8 + 1) & ~(size_t)1;
  size_t b = cap * 8 + 4 * cap * isz + 64;
  if (mode == 2) b += 10 * maxseg * isz + (cap / 32 + 2) * 4 + 64;
  return (b + 255) & ~(size_t)255;
}
template <int MODE>
__global__ void __launch_bounds__(SORT_NT) sort_wg_kernel(int n_reads, const uint64_t* __restrict__ mm_off, uint64_t* mm_key, uint32_t* mm_pos,
                                                        void* tscratch, int cap, int* __restrict__ fallback,
                                                        const int* __restrict__ only, char* gscr, int* stat, int minLen = 0) {
  constexpr bool BIG = MODE >= 1, HUGE = MODE == 2;
  typedef typename SortTypes<MODE>::IT IT;
  typedef typename SortTypes<MODE>::TT TT;
  constexpr IT S_NONE = (IT)~(IT)0;
  constexpr int HB = HUGE ? 32 : 16;                                      // bits per half of a prefix word
  constexpr TT HM = (TT)(((TT)1 << HB) - 1);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NT_ = (int)blockDim.x, NW_ = NT_ >> 6;                         // (1024 threads per list; 256 for the launch that takes the short lists four to a CU)
  const int maxseg = (cap / 16 + 8 + 1) & ~1;                            // even: the word array behind the ten tables stays 4-byte aligned
  char* ebase = BIG ? gscr + (size_t)blockIdx.x * sort_scratch_bytes(MODE, (size_t)cap) : smem;
  uint64_t* key = (uint64_t*)ebase;
  IT* idx = (IT*)(key + cap);
  IT* seg = idx + cap;
  IT* pa = seg + cap;
  IT* pb = pa + cap;
  IT* sF = HUGE ? pb + cap : BIG ? (IT*)smem : pb + cap;                  // segment tables (current)
  IT* sL = sF + maxseg;
  IT* sD = sL + maxseg;
  IT* nF = sD + maxseg;         // next level
  IT* nL = nF + maxseg;
  IT* nD = nL + maxseg;
  IT* sCut = nD + maxseg;
  IT* sLid = sCut + maxseg;
  IT* sRid = sLid + maxseg;
  IT* sM = sRid + maxseg;
  unsigned int* startBits = (unsigned int*)(sM + maxseg + (maxseg & 1));   // leftover-block start markers, cap/32+1 words
  __shared__ int cnt[4];                    // nseg, nnext
  __shared__ unsigned int waveTot[2 * SORT_NW];
  TT* t32 = (TT*)tscratch + (size_t)blockIdx.x * (size_t)(cap + 64);
  LdsSeg<IT> S{key, idx};
  for (int r = blockIdx.x; r < n_reads; r += gridDim.x) {
    const long base = (long)mm_off[r];
    const int n = (int)(mm_off[r + 1] - mm_off[r]);
    if (n < 2) continue;
    if (BIG) {
      if (!fallback[r]) continue;                                         // (every thread reads the flag before anyone clears it: the barrier below)
      if (n > cap) { if (MODE == 1 && stat && tid == 0) { atomicMax(&stat[0], n); atomicAdd(&stat[1], 1); } continue; }
    } else {
      if (only && !only[r]) continue;
      if (n < minLen) continue;                                           // (a shorter list was the previous launch's)
      if (n > cap) { if (fallback && tid == 0) { fallback[r] = 1; if (stat) atomicAdd(&stat[2], 1); } continue; }   // (stat[2]: lists left to the launch for large lists)
    }
    __syncthreads();
    if (BIG && tid == 0) fallback[r] = 0;
    for (int p = tid; p < n; p += NT_) { key[p] = mm_key[base + p]; idx[p] = (IT)p; seg[p] = 0; }
    for (int x = tid; x < cap / 32 + 1; x += NT_) startBits[x] = 0;
    if (tid == 0) {
      if (n > 16) { sF[0] = 0; sL[0] = (IT)n; sD[0] = (IT)(2 * (31 - __clz(n))); cnt[0] = 1; }
      else cnt[0] = 0;
      cnt[1] = 0;
    }
    __syncthreads();
    if (tid == 0) startBits[0] = 1u;
    int nseg = cnt[0];
    while (nseg > 0) {
      // (a) depth check / median of three -> pivot at `first`
      for (int s = tid; s < nseg; s += NT_) {
        const int first = (int)sF[s], last = (int)sL[s];
        sM[s] = 0;
        if (sD[s] == 0) { lds_heap_sort(S, first, last); sM[s] = S_NONE; continue; }
        sD[s] = sD[s] - 1;
        const int a = first + 1, b = first + (last - first) / 2, c = last - 1;
        if (S.lt(a, b)) {
          if (S.lt(b, c)) S.swap(first, b);
          else if (S.lt(a, c)) S.swap(first, c);
          else S.swap(first, a);
        } else if (S.lt(a, c)) S.swap(first, a);
        else if (S.lt(b, c)) S.swap(first, c);
        else S.swap(first, b);
      }
      __syncthreads();
      // (b) exclusive prefix counts of the two stopper flags over all positions (wave w owns a
      //     contiguous range; ballots give in-row ranks)
      const int rows = (n + 63) / 64, rpw = (rows + NW_ - 1) / NW_;
      const int r0 = wave * rpw, r1 = min(rows, r0 + rpw);
      auto flags = [&](int p, bool& A, bool& B) {
        A = false; B = false;
        if (p < n) {
          const IT sg = seg[p];
          if (sg != S_NONE && sM[sg] != S_NONE && p != (int)sF[sg]) {
            const uint64_t piv = key[sF[sg]] & FOR_MASK, km = key[p] & FOR_MASK;
            A = km >= piv; B = km <= piv;
          }
        }
      };
      unsigned int totA = 0, totB = 0;
      for (int rw = r0; rw < r1; rw++) {
        bool A, B; flags(rw * 64 + lane, A, B);
        totA += __popcll(__ballot(A)); totB += __popcll(__ballot(B));
      }
      if (lane == 0) { waveTot[wave] = totA; waveTot[NW_ + wave] = totB; }
      __syncthreads();
      unsigned int baseA = 0, baseB = 0;
      for (int w = 0; w < wave; w++) { baseA += waveTot[w]; baseB += waveTot[NW_ + w]; }
      const unsigned long long below = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));
      for (int rw = r0; rw < r1; rw++) {
        const int p = rw * 64 + lane;
        bool A, B; flags(p, A, B);
        const unsigned long long mA = __ballot(A), mB = __ballot(B);
        if (p <= n) t32[p] = ((TT)(baseA + __popcll(mA & below)) & HM) | ((TT)(baseB + __popcll(mB & below)) << HB);
        baseA += __popcll(mA); baseB += __popcll(mB);
      }
      if (wave == NW_ - 1 && lane == 0 && (n & 63) == 0) t32[n] = ((TT)baseA & HM) | ((TT)baseB << HB);   // prefix at n when n is a row boundary
      __syncthreads();
      // (c) scatter stopper positions: pa ascending, pb descending, both stored from first+1
      for (int p = tid; p < n; p += NT_) {
        bool A, B; flags(p, A, B);
        if (A || B) {
          const int sg = (int)seg[p], b0 = (int)sF[sg] + 1;
          const TT t = t32[p], tb = t32[b0], te = t32[sL[sg]];
          if (A) pa[b0 + (int)(((t & HM) - (tb & HM)) & HM)] = (IT)p;
          if (B) pb[b0 + (int)(((te >> HB) - (t >> HB) - 1) & HM)] = (IT)p;
        }
      }
      __syncthreads();
      // (d) m = number of leading pairs with a_i < b_i
      for (int p = tid; p < n; p += NT_) {
        const IT sg = seg[p];
        if (sg == S_NONE || sM[sg] == S_NONE) continue;
        const int b0 = (int)sF[sg] + 1, i = p - b0;
        if (i < 0) continue;
        const TT tb = t32[b0], te = t32[sL[sg]];
        const int nA = (int)(((te & HM) - (tb & HM)) & HM), nB = (int)(((te >> HB) - (tb >> HB)) & HM);
        const int lim = min(nA, nB);
        if (i < lim && pa[b0 + i] < pb[b0 + i] && !(i + 1 < lim && pa[b0 + i + 1] < pb[b0 + i + 1])) sM[sg] = (IT)(i + 1);
      }
      __syncthreads();
      // (e) the swaps
      for (int p = tid; p < n; p += NT_) {
        const IT sg = seg[p];
        if (sg == S_NONE || sM[sg] == S_NONE) continue;
        const int b0 = (int)sF[sg] + 1, i = p - b0;
        if (i >= 0 && i < (int)sM[sg]) S.swap(pa[b0 + i], pb[b0 + i]);
      }
      __syncthreads();
      // (f) cut -> children
      for (int s = tid; s < nseg; s += NT_) {
        const int first = (int)sF[s], last = (int)sL[s];
        if (sM[s] == S_NONE) { sCut[s] = (IT)last; sLid[s] = S_NONE; sRid[s] = S_NONE; continue; }   // heap sorted: finished
        const int b0 = first + 1, m = (int)sM[s];
        const TT tb = t32[b0], te = t32[last];
        const int nA = (int)(((te & HM) - (tb & HM)) & HM);
        int cut = 0x7fffffff;
        if (m < nA) cut = pa[b0 + m];
        if (m >= 1) cut = min(cut, (int)pb[b0 + m - 1]);
        sCut[s] = (IT)cut;
        IT lid = S_NONE, rid = S_NONE;
        if (cut - first > 16) { int x = atomicAdd(&cnt[1], 1); nF[x] = (IT)first; nL[x] = (IT)cut; nD[x] = sD[s]; lid = (IT)x; }
        if (last - cut > 16) { int x = atomicAdd(&cnt[1], 1); nF[x] = (IT)cut; nL[x] = (IT)last; nD[x] = sD[s]; rid = (IT)x; }
        atomicOr(&startBits[cut >> 5], 1u << (cut & 31));
        sLid[s] = lid; sRid[s] = rid;
      }
      __syncthreads();
      // (g) relabel elements, swap tables
      for (int p = tid; p < n; p += NT_) {
        const IT sg = seg[p];
        if (sg != S_NONE) seg[p] = (p < (int)sCut[sg]) ? sLid[sg] : sRid[sg];
      }
      nseg = cnt[1];
      __syncthreads();
      for (int s = tid; s < nseg; s += NT_) { sF[s] = nF[s]; sL[s] = nL[s]; sD[s] = nD[s]; }
      if (tid == 0) cnt[1] = 0;
      __syncthreads();
    }
    // final insertion sort of every leftover block (blocks start at the marked positions)
    for (int p = tid; p < n; p += NT_) {
      if ((startBits[p >> 5] >> (p & 31)) & 1u) {
        int q = p + 1;
        while (q < n && !((startBits[q >> 5] >> (q & 31)) & 1u)) q++;
        lds_insertion_sort(S, p, q);
      }
    }
    __syncthreads();
    uint32_t* tmp = HUGE ? (uint32_t*)seg : (uint32_t*)pa;            // pa|pb = 4*cap bytes (32-bit indices: seg, dead by now)
    for (int p = tid; p < n; p += NT_) tmp[p] = mm_pos[base + idx[p]];
    __syncthreads();
    for (int p = tid; p < n; p += NT_) { mm_key[base + p] = key[p]; mm_pos[base + p] = tmp[p]; }
  }
}

// ------------------------------------------------------------------------------------ a3
// (i) per query tuple: global lower/upper bound of its masked key in the index.  Because the
// index is sorted by masked key, lower_bound over any sub-range [ts,te) is the global bound
// clamped to [ts,te] -- so the serial walk of CompareLists needs no searches of its own.
// A bucket directory over the top bits of the masked key (built once at index load) confines each
// search to a handful of entries: dir[b] = first index whose (key >> shift) >= b.
__global__ void dir_build_kernel(uint32_t nbuckets, int shift, const uint64_t* __restrict__ idx_key, uint64_t n_idx, uint32_t* __restrict__ dir) {
  uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b > nbuckets) return;
  if (b == nbuckets) { dir[b] = (uint32_t)n_idx; return; }
  const uint64_t q = b << shift;
  uint64_t lo = 0, hi = n_idx;
  while (lo < hi) { uint64_t mid = lo + ((hi - lo) >> 1); if ((idx_key[mid] & FOR_MASK) < q) lo = mid + 1; else hi = mid; }
  dir[b] = (uint32_t)lo;
}

// Also hands the walk the index keys it will look at when it lands on these bounds (T[lb], T[lb-1],
// T[ub-1]), so the serial walk itself makes no dependent loads from the index.
__global__ void bounds_kernel(uint64_t total, const uint64_t* __restrict__ mm_key, const uint64_t* __restrict__ idx_key,
                              uint64_t n_idx, const uint32_t* __restrict__ dir, uint32_t nbuckets, int shift,
                              uint32_t* __restrict__ lb, uint32_t* __restrict__ ub, uint64_t* __restrict__ tk_lb,
                              uint64_t* __restrict__ tk_lbm1, uint64_t* __restrict__ tk_ubm1) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint64_t q = mm_key[i] & FOR_MASK;
  uint64_t bkt = q >> shift;
  if (bkt >= nbuckets) bkt = nbuckets - 1;
  uint64_t lo = dir[bkt], hi = dir[bkt + 1];
  if (bkt == nbuckets - 1) hi = n_idx;
  const uint64_t hi0 = hi;
  while (lo < hi) { uint64_t mid = lo + ((hi - lo) >> 1); if ((idx_key[mid] & FOR_MASK) < q) lo = mid + 1; else hi = mid; }
  const uint64_t l = lo;
  // the end of the equal run: it starts at l and is short (most index keys occur once), so a few forward steps replace the second binary search from scratch
  // (the kernel is bandwidth-bound: every probe of a 1.5 GB key array is a cache line); a long run finishes with the search on what is left
  hi = hi0;
  { int steps = 0; while (lo < hi && steps < 4 && (idx_key[lo] & FOR_MASK) == q) { lo++; steps++; }
    if (steps == 4) { while (lo < hi) { uint64_t mid = lo + ((hi - lo) >> 1); if (!(q < (idx_key[mid] & FOR_MASK))) lo = mid + 1; else hi = mid; } } }
  const uint64_t u = lo;
  lb[i] = (uint32_t)l; ub[i] = (uint32_t)u;
  tk_lb[i] = (l < n_idx) ? idx_key[l] : 0;
  tk_lbm1[i] = (l > 0) ? idx_key[l - 1] : 0;
  tk_ubm1[i] = (u > 0) ? idx_key[u - 1] : 0;
}

// per-read upper bound on emitted pairs: a (query tuple, index tuple) pair with equal keys can be
// emitted by a front step, re-emitted once by the raw-key quirk (:101-102), and once by a back step
// opts.defer_seed_matches (scheduling only): the reads whose walk found more matches than T keep none of them -- every later stage sees a read without matches -- and are
// flagged; the driver marks them LRA_ST_DEFERRED and the caller maps them in a later batch of their own kind
__global__ void defer_heavy_kernel(int n_reads, uint32_t T, uint64_t* __restrict__ counts, uint8_t* __restrict__ flag) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_reads) return;
  const bool d = counts[r] > (uint64_t)T;
  flag[r] = d ? 1 : 0;
  if (d) counts[r] = 0;
}
__global__ void __launch_bounds__(64) match_capacity_kernel(int n_reads, const uint64_t* __restrict__ mm_off, const uint32_t* __restrict__ lb,
                                                            const uint32_t* __restrict__ ub, uint64_t* __restrict__ cap) {
  const int lane = threadIdx.x;
  for (int r = blockIdx.x; r < n_reads; r += gridDim.x) {
    uint64_t sum = 0;
    for (uint64_t i = mm_off[r] + lane; i < mm_off[r + 1]; i += 64) sum += (uint64_t)(ub[i] - lb[i]);
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane == 0) cap[r] = 3 * sum;
  }
}

// (ii) the alternating two-ended walk of CompareLists.h:43-143, one lane per read.  T[ts] and
// T[te-1] live in registers and are refreshed from the prefetched neighbourhood arrays whenever
// ts / te jump to a bound; only the raw-key run skip (:101) still reads the index itself.
__global__ void __launch_bounds__(64) compare_kernel(int FLAT_LANES, int n_reads, const uint64_t* __restrict__ mm_off, const uint64_t* __restrict__ mm_key,
                                                     const uint32_t* __restrict__ lbA, const uint32_t* __restrict__ ubA,
                                                     const uint64_t* __restrict__ tkLbA, const uint64_t* __restrict__ tkLbm1A,
                                                     const uint64_t* __restrict__ tkUbm1A,
                                                     const uint64_t* __restrict__ idx_key, long n_idx, long maxFreq,
                                                     const uint64_t* __restrict__ match_off, uint32_t* __restrict__ match_qi,
                                                     uint32_t* __restrict__ match_ti, uint64_t* __restrict__ counts) {
  if (threadIdx.x >= FLAT_LANES) return;                                 // the walks of a wave's reads diverge: 32 reads per wave measured best (16: 39 ms, 32: 37, 64: 45)
  const int r = blockIdx.x * FLAT_LANES + threadIdx.x;
  if (r >= n_reads) return;
  const uint64_t* qk = mm_key + mm_off[r];
  const uint32_t* LB = lbA + mm_off[r];
  const uint32_t* UB = ubA + mm_off[r];
  const uint64_t* TKLB = tkLbA + mm_off[r];
  const uint64_t* TKLBM1 = tkLbm1A + mm_off[r];
  const uint64_t* TKUBM1 = tkUbm1A + mm_off[r];
  // (32-bit cursors: a read's tuples and the index both stay below 2^32 -- lb / ub are 32-bit -- and the walk is bound by instruction issue, where every 64-bit
  // compare or add is two)
  const int nq = (int)(mm_off[r + 1] - mm_off[r]);
  const uint32_t nt = (uint32_t)n_idx;
  const int mf = maxFreq > 0x7fffffffL ? 0x7fffffff : (int)maxFreq;
  uint32_t* oq = match_qi + match_off[r];
  uint32_t* ot = match_ti + match_off[r];
  const uint64_t room = match_off[r + 1] - match_off[r];
  uint64_t n = 0;
  const uint64_t M = FOR_MASK;
  if (nq != 0 && nt != 0) {                                            // :27-30
    int qs = 0, qe = nq - 1; uint32_t ts = 0, te = nt;
    uint64_t Tts = idx_key[0], Tte1 = idx_key[nt - 1];                 // raw T[ts], T[te-1]
    do {
      // Every load of a step depends on qs / qe only, which are known here: the step's tuples (three keys from each end, the bounds and index keys of qs and qe)
      // are asked for in one go -- one memory round trip per step instead of one per basic block (~6) -- and the loops below fall back to loads beyond them.
      const int pq = qs, eq = qe;
      const uint64_t pK0 = qk[pq], pK1 = qk[pq + 1 < nq ? pq + 1 : pq], pK2 = qk[pq + 2 < nq ? pq + 2 : pq];
      const uint32_t pLB = LB[pq], pUB = UB[pq]; const uint64_t pTL = TKLB[pq];
      const uint64_t eK0 = qk[eq], eK1 = qk[eq >= 1 ? eq - 1 : eq], eK2 = qk[eq >= 2 ? eq - 2 : eq];
      const uint32_t eLB = LB[eq], eUB = UB[eq]; const uint64_t eTU1 = TKUBM1[eq], eTL1 = TKLBM1[eq];
      auto QF = [&](int i) -> uint64_t { const int d = i - pq; return d == 0 ? pK0 : d == 1 ? pK1 : d == 2 ? pK2 : qk[i]; };   // raw key of tuple i, front / back
      auto QB = [&](int i) -> uint64_t { const int d = eq - i; return d == 0 ? eK0 : d == 1 ? eK1 : d == 2 ? eK2 : qk[i]; };
      while (qs <= qe && (QF(qs) & M) < (Tts & M)) qs++;               // :47-49
      if (qs >= qe) break;                                             // :51-53
      const uint64_t Qs = QF(qs) & M;
      uint64_t startGap = Qs - (Tts & M);
      while (qe > qs && te > ts && (QB(qe) & M) > (Tte1 & M)) qe--;    // :63-65
      const uint64_t Qe = QB(qe) & M;
      uint64_t endGap = (Tte1 & M) - Qe;
      if (startGap == 0 || (startGap & M) > (endGap & M)) {            // :69
        const uint32_t tsOrig = ts; const int qsOrig = qs;
        const uint64_t rawOrig = Tts;
        const uint32_t lo = qs == pq ? pLB : LB[qs];                   // lower_bound on [ts,te)  (:76)
        if (lo > ts) {
          if (lo >= te) ts = te;
          else { ts = lo; Tts = (qs == pq ? pTL : TKLB[qs]); }
        }
        if (ts < te && (Tts & M) == Qs) {
          const uint32_t tsStart = ts;
          uint32_t tsi = ts;
          { uint32_t e = qs == pq ? pUB : UB[qs]; e = e > te ? te : e; if (e > tsi) tsi = e; }   // end of the equal run inside [ts,te)
          const uint32_t qsStart = (uint32_t)qs;
          while (qs < qe && (QF(qs + 1) & M) == Qs) qs++;
          if (qs - (int)qsStart < mf) {
            for (uint32_t ti = tsStart; ti != tsi; ti++)
              for (uint32_t qi = qsStart; qi <= (uint32_t)qs; qi++) {
                if (n < room) { oq[n] = qi; ot[n] = ti; }
                n++;
              }
          }
        }
        if (ts == tsOrig) {                                            // :101 (a jump lands on a different key: no skip)
          while (ts < te && Tts == rawOrig) { ts++; if (ts < nt) Tts = idx_key[ts]; }
        }
        { const uint64_t raw = QF(qsOrig); while (qs < qe && QF(qs) == raw) qs++; }             // :102
      } else {
        if (te != nt && (Tte1 & M) == Qe) {                            // :112-114
        } else {                                                       // upper_bound on [ts,te) (:116-118)
          const uint32_t hi = qe == eq ? eUB : UB[qe];
          if (hi < te) {
            if (hi <= ts) te = ts;
            else { te = hi; Tte1 = (qe == eq ? eTU1 : TKUBM1[qe]); }
          }
        }
        const uint32_t teStart = te;
        uint32_t tei = te;
        if (tei > ts && (Tte1 & M) == Qe) {                      // start of the equal run inside [ts,te)
          const uint32_t b = qe == eq ? eLB : LB[qe];
          if (b <= ts) tei = ts;
          else { tei = b; Tte1 = (qe == eq ? eTL1 : TKLBM1[qe]); }
        }
        if (tei < teStart && teStart > 0) {
          const uint32_t qeStart = (uint32_t)qe;
          while (qe > qs && (QB(qe) & M) == (QB(qe - 1) & M)) qe--;
          if ((int)qeStart - qe < mf) {
            for (uint32_t ti = tei; ti < teStart; ti++)
              for (uint32_t qi = (uint32_t)qe; qi <= qeStart; qi++) {
                if (n < room) { oq[n] = qi; ot[n] = ti; }
                n++;
              }
          }
        }
        te = tei;
      }
    } while (qs < qe && ts < te);
  }
  counts[r] = n;
}

// ------------------------------------------------------------------------------------ a4
// One wave per read: strand flag per match (k-byte compare of read vs genome), then a stable
// partition (forward first) with ballot/popcount prefix sums.
__global__ void __launch_bounds__(64) strand_kernel(int n_reads, const unsigned char* __restrict__ seq_all, const uint64_t* __restrict__ read_off,
                                                    const unsigned char* __restrict__ genome, int k,
                                                    const uint64_t* __restrict__ mm_off, const uint32_t* __restrict__ mm_pos,
                                                    const uint32_t* __restrict__ idx_pos,
                                                    const uint64_t* __restrict__ match_off, const uint64_t* __restrict__ src_off,
                                                    const uint32_t* __restrict__ src_qi, const uint32_t* __restrict__ src_ti,
                                                    uint32_t* __restrict__ match_qi, uint32_t* __restrict__ match_ti, uint32_t* __restrict__ sep_qpos,
                                                    uint32_t* __restrict__ sep_tpos, uint32_t* __restrict__ n_forward,
                                                    const uint64_t* __restrict__ mm_key, uint64_t* __restrict__ sep_qkey) {
  const int lane = threadIdx.x;
  for (int r = blockIdx.x; r < n_reads; r += gridDim.x) {
    const unsigned char* read = seq_all + read_off[r];
    const uint32_t* qpos_of = mm_pos + mm_off[r];
    const uint64_t m0 = match_off[r], m1 = match_off[r + 1];
    // pass 0: compact the pairs out of the capacity-spaced walk buffer
    for (uint64_t i = m0 + lane; i < m1; i += 64) { match_qi[i] = src_qi[src_off[r] + (i - m0)]; match_ti[i] = src_ti[src_off[r] + (i - m0)]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // pass 1: count forward matches
    uint64_t nf = 0;
    for (uint64_t base = m0; base < m1; base += 64) {
      uint64_t i = base + lane;
      bool fwd = false;
      if (i < m1) {
        const unsigned char* a = read + qpos_of[match_qi[i]];
        const unsigned char* b = genome + idx_pos[match_ti[i]];
        // (k bytes of each side, eight at a time and all asked for at once: a byte loop with an early exit is k round trips in a row)
        fwd = true;
        int x = 0;
        for (; x + 8 <= k; x += 8) { unsigned long long wa, wb; __builtin_memcpy(&wa, a + x, 8); __builtin_memcpy(&wb, b + x, 8); fwd = fwd && wa == wb; }
        for (; x < k; x++) fwd = fwd && a[x] == b[x];
        if (fwd) match_qi[i] |= 0x80000000u;                               // remembered for pass 2 (the same lane reads it back): the k-byte compare costs two cache
      }                                                                    // lines of random access per match -- 12 GB per batch -- and was made twice
      nf += __popcll(__ballot(fwd));
    }
    if (lane == 0) n_forward[r] = (uint32_t)nf;
    // pass 2: stable partition
    uint64_t fcur = m0, rcur = m0 + nf;
    for (uint64_t base = m0; base < m1; base += 64) {
      uint64_t i = base + lane;
      bool in = i < m1, fwd = false;
      uint32_t qp = 0, tp = 0;
      uint64_t qk = 0;
      if (in) {
        const uint32_t mq = match_qi[i], qi = mq & 0x7fffffffu;
        fwd = (mq >> 31) != 0;
        if (fwd) match_qi[i] = qi;
        qp = qpos_of[qi]; tp = idx_pos[match_ti[i]]; qk = mm_key[mm_off[r] + qi];
      }
      unsigned long long mf = __ballot(in && fwd), mr = __ballot(in && !fwd);
      unsigned long long below = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));
      if (in) {
        uint64_t dst = fwd ? fcur + __popcll(mf & below) : rcur + __popcll(mr & below);
        sep_qpos[dst] = qp; sep_tpos[dst] = tp; sep_qkey[dst] = qk;
      }
      fcur += __popcll(mf); rcur += __popcll(mr);
    }
  }
}

// ------------------------------------------------------------------------------------ scans
// exclusive scan of n (<= a few million) counts into n+1 offsets; one workgroup.

}  // namespace

// ======================================================================================
#include "seed_state.h"

static lra_seed_state* seed_state(lra_ctx* ctx) {
  if (!ctx->seed) ctx->seed = new lra_seed_state();
  return ctx->seed;
}

template <typename T>
static bool regrow(T*& p, size_t n) {
  if (p) (void)hipFree(p);
  p = nullptr;
  return hipMalloc((void**)&p, n * sizeof(T) + 64) == hipSuccess;
}

// a loader on a context that borrows its reference data: the borrowed pointers are dropped (they belong to the owner), the context owns what it loads from here on
static void seed_disown(lra_seed_state* s) {
  if (!s->borrowed) return;
  s->genome = nullptr; s->genome_len = 0; s->idx_key = nullptr; s->idx_pos = nullptr; s->n_idx = 0; s->dir = nullptr; s->nbuckets = 0; s->dir_shift = 0;
  s->borrowed = false; s->owner_cell.reset(); s->owner_generation = 0;
}
// LRA_OK, or LRA_ERR_INVALID when the context borrows reference data its owner has replaced since (lra_ctx_share_reference again)
int lra_seed_check_shared(lra_ctx* ctx) {
  const lra_seed_state* s = ctx->seed;
  if (!s || !s->borrowed || !s->owner_cell) return LRA_OK;
  if (s->owner_cell->dead.load() || s->owner_cell->gen.load() != s->owner_generation)
    return lra_set_err(ctx, LRA_ERR_INVALID, "the context this one shares its reference data with has %s: call lra_ctx_share_reference again",
                       s->owner_cell->dead.load() ? "been destroyed" : "reloaded it");
  return LRA_OK;
}

// lra_ctx_release_buffers: the seed stage's batch arrays (grown to the largest batch's minimizers / matches; the match walk's capacity-spaced buffers are the largest: three
// slots per candidate pair) back to the device, the reference (genome, global index, bucket directory) kept.  -> bytes freed are not tracked per array: the caller reads
// the device's free memory.
void lra_seed_release_batch(lra_ctx* ctx) {
  lra_seed_state* s = ctx->seed;
  if (!s) return;
  void** ptrs[] = {(void**)&s->counts32, (void**)&s->counts64, (void**)&s->mm_off, (void**)&s->match_off, (void**)&s->n_forward, (void**)&s->mm_key, (void**)&s->mm_pos,
                   (void**)&s->lb, (void**)&s->ub, (void**)&s->match_qi, (void**)&s->match_ti, (void**)&s->sep_qpos, (void**)&s->sep_tpos, (void**)&s->tmp_qi, (void**)&s->tmp_ti,
                   (void**)&s->cap_cnt, (void**)&s->cap_off, (void**)&s->tk_lb, (void**)&s->tk_lbm1, (void**)&s->tk_ubm1, (void**)&s->sep_qkey, (void**)&s->defer_flag};
  for (void** p : ptrs) if (*p) { (void)hipFree(*p); *p = nullptr; }
  s->cap_reads = 0; s->cap_mm = 0; s->cap_match = 0; s->cap_tmp = 0; s->cap_defer = 0;
  s->last_n_reads = 0; s->last_n_matches = 0;
}

void lra_seed_free(lra_ctx* ctx) {
  lra_seed_state* s = ctx->seed;
  if (!s) return;
  if (s->borrowed) { s->genome = nullptr; s->idx_key = nullptr; s->idx_pos = nullptr; s->dir = nullptr; }
  else s->cell->dead = true;                                               // borrowers hold the cell, not this state
  void* ptrs[] = {s->genome, s->idx_key, s->idx_pos, s->counts32, s->counts64, s->mm_off, s->match_off, s->n_forward,
                  s->mm_key, s->mm_pos, s->lb, s->ub, s->match_qi, s->match_ti, s->sep_qpos, s->sep_tpos, s->tmp_qi, s->tmp_ti, s->cap_cnt, s->cap_off, s->tk_lb, s->tk_lbm1, s->tk_ubm1, s->dir, s->sep_qkey, s->defer_flag};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  delete s;
  ctx->seed = nullptr;
}

extern "C" int lra_ctx_load_genome(lra_ctx* ctx, const char* h_seq, uint64_t len) {
  if (!ctx || (!h_seq && len)) return LRA_ERR_INVALID;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  lra_seed_state* s = seed_state(ctx);
  seed_disown(s); s->cell->gen++; ctx->ahead.valid = false;   // (a seed result adopted ahead of its batch was made from the old data)
  if (!regrow(s->genome, len + 64)) return lra_set_err(ctx, LRA_ERR_NOMEM, "genome alloc");
  s->genome_len = len;
  LRA_HIP_CHECK(ctx, hipMemcpy(s->genome, h_seq, len, hipMemcpyHostToDevice));
  LRA_HIP_CHECK(ctx, hipMemset(s->genome + len, 0, 64));
  return LRA_OK;
}

// adopts device arrays (hipMalloc'ed, n + 1 entries at least) as the context's global index and builds the bucket directory over the key's top bits
int lra_seed_install_index(lra_ctx* ctx, uint64_t* d_key, uint32_t* d_pos, uint64_t n) {
  lra_seed_state* s = seed_state(ctx);
  seed_disown(s); s->cell->gen++; ctx->ahead.valid = false;   // (a seed result adopted ahead of its batch was made from the old data)
  if (s->idx_key) (void)hipFree(s->idx_key);
  if (s->idx_pos) (void)hipFree(s->idx_pos);
  s->idx_key = d_key; s->idx_pos = d_pos; s->n_idx = n;
  if (!d_key || !d_pos) {
    s->idx_key = nullptr; s->idx_pos = nullptr; s->n_idx = 0;
    if (!regrow(s->idx_key, 2) || !regrow(s->idx_pos, 2)) return lra_set_err(ctx, LRA_ERR_NOMEM, "index alloc");
    n = 0;
  }
  // bucket directory: ~2 buckets per entry, on the top bits of the largest masked key
  int dbits = 1;
  while ((1ULL << dbits) < 2 * n && dbits < 27) dbits++;
  uint64_t maxkey = 0;
  if (n) { LRA_HIP_CHECK(ctx, hipMemcpy(&maxkey, s->idx_key + n - 1, 8, hipMemcpyDeviceToHost)); maxkey &= FOR_MASK; }
  int kbits = 1;
  while (kbits < 63 && (maxkey >> kbits)) kbits++;
  s->dir_shift = kbits > dbits ? kbits - dbits : 0;
  s->nbuckets = (uint32_t)((maxkey >> s->dir_shift) + 1);
  if (!regrow(s->dir, (size_t)s->nbuckets + 2)) return lra_set_err(ctx, LRA_ERR_NOMEM, "index directory");
  hipLaunchKernelGGL(dir_build_kernel, dim3((s->nbuckets + 256) / 256), dim3(256), 0, ctx->stream, s->nbuckets, s->dir_shift, s->idx_key, n, s->dir);
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return LRA_OK;
}

extern "C" int lra_ctx_load_global_index(lra_ctx* ctx, const uint64_t* h_key, const uint32_t* h_pos, uint64_t n) {
  if (!ctx || n >= (1ULL << 32)) return LRA_ERR_INVALID;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  uint64_t* dk = nullptr; uint32_t* dp = nullptr;
  if (!regrow(dk, n + 1) || !regrow(dp, n + 1)) { if (dk) (void)hipFree(dk); return lra_set_err(ctx, LRA_ERR_NOMEM, "index alloc"); }
  LRA_HIP_CHECK(ctx, hipMemcpy(dk, h_key, n * 8, hipMemcpyHostToDevice));
  LRA_HIP_CHECK(ctx, hipMemcpy(dp, h_pos, n * 4, hipMemcpyHostToDevice));
  return lra_seed_install_index(ctx, dk, dp, n);
}

int lra_seed_share(lra_ctx* dst, lra_ctx* src) {
  if (dst->seed && !dst->seed->borrowed && (dst->seed->genome || dst->seed->idx_key)) return lra_set_err(dst, LRA_ERR_INVALID, "context already holds reference data");
  lra_seed_state* d = seed_state(dst);
  const lra_seed_state* s = src->seed;
  d->borrowed = true; d->owner_cell = s->borrowed ? s->owner_cell : s->cell; d->owner_generation = s->borrowed ? s->owner_generation : s->cell->gen.load();
  d->genome = s->genome; d->genome_len = s->genome_len; d->idx_key = s->idx_key; d->idx_pos = s->idx_pos; d->n_idx = s->n_idx;
  d->dir = s->dir; d->nbuckets = s->nbuckets; d->dir_shift = s->dir_shift;
  return LRA_OK;
}

// the context's global index (device arrays; the .mms payload as two columns)
extern "C" int lra_ctx_global_index(lra_ctx* ctx, const uint64_t** d_key, const uint32_t** d_pos, uint64_t* n) {
  if (!ctx || !ctx->seed || !n) return LRA_ERR_INVALID;
  if (d_key) *d_key = ctx->seed->idx_key;
  if (d_pos) *d_pos = ctx->seed->idx_pos;
  *n = ctx->seed->n_idx;
  return LRA_OK;
}

// the genome from a device buffer (copied: the context owns its reference data)
extern "C" int lra_ctx_load_genome_device(lra_ctx* ctx, const char* d_seq, uint64_t len) {
  if (!ctx || (!d_seq && len)) return LRA_ERR_INVALID;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  lra_seed_state* s = seed_state(ctx);
  seed_disown(s); s->cell->gen++; ctx->ahead.valid = false;   // (a seed result adopted ahead of its batch was made from the old data)
  if (!regrow(s->genome, len + 64)) return lra_set_err(ctx, LRA_ERR_NOMEM, "genome alloc");
  s->genome_len = len;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(s->genome, d_seq, len, hipMemcpyDeviceToDevice, ctx->stream));
  LRA_HIP_CHECK(ctx, hipMemsetAsync(s->genome + len, 0, 64, ctx->stream));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return LRA_OK;
}

// CreateRC (SeqUtils.h:151-158, RevCompNuc :112-146): dest[l-i-1] = complement(seq[i]); anything that is
// not one of ACGTacgtn becomes 'N'.
__global__ void create_rc_kernel(int n_reads, const unsigned char* __restrict__ seq, const uint64_t* __restrict__ off, unsigned char* __restrict__ dst) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
  for (int r = wave; r < n_reads; r += nw) {
    const uint64_t a = off[r], L = off[r + 1] - a;
    for (uint64_t i = lane; i < L; i += 64) {
      const unsigned char c = seq[a + i];
      unsigned char o = 'N';
      switch (c) {
        case 'A': o = 'T'; break; case 'C': o = 'G'; break; case 'G': o = 'C'; break; case 'T': o = 'A'; break;
        case 'a': o = 't'; break; case 'c': o = 'g'; break; case 'g': o = 'c'; break; case 't': o = 'a'; break;
        case 'n': o = 'n'; break;
        default: break;
      }
      dst[a + L - 1 - i] = o;
    }
  }
}

extern "C" int lra_create_rc_batch(lra_ctx* ctx, int n_reads, const char* d_seq, const uint64_t* d_read_off, char* d_rc) {
  if (!ctx || n_reads < 0) return LRA_ERR_INVALID;
  if (n_reads == 0) return LRA_OK;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  lra_time_begin(ctx, "create_rc");
  hipLaunchKernelGGL(create_rc_kernel, dim3(ctx->num_cu * 8), dim3(256), 0, ctx->stream, n_reads, (const unsigned char*)d_seq, d_read_off, (unsigned char*)d_rc);
  lra_time_end(ctx);
  LRA_HIP_CHECK(ctx, hipGetLastError());
  return LRA_OK;
}

static int launch_sort(lra_ctx* ctx, int n_reads, const uint64_t* mm_off, uint64_t* mm_key, uint32_t* mm_pos, const int* only = nullptr) {
  hipStream_t st = ctx->stream;
  const int nb = (n_reads + 63) / 64;
  const int cap = SORT_CAP, capB = 65534;                                 // unsigned short indices, 0xFFFF = none
  const int maxseg = (cap / 16 + 8 + 1) & ~1, maxsegB = (capB / 16 + 8 + 1) & ~1;
  const size_t lds = (size_t)cap * 16 + (size_t)maxseg * 20 + 8 + (size_t)(cap / 32 + 2) * 4;
  const size_t ldsB = (size_t)maxsegB * 20 + 8 + (size_t)(capB / 32 + 2) * 4 + 64;
  const int grid = n_reads < ctx->num_cu ? n_reads : ctx->num_cu;
  const int gridB = std::min(grid, 64);
  const size_t tszB = (size_t)gridB * (capB + 64) * 4, esz = (size_t)gridB * sort_scratch_bytes(1, (size_t)capB);
  uint32_t* tscr = (uint32_t*)lra_scratch(ctx, 0, (size_t)grid * (cap + 64) * 4 + (size_t)n_reads * 4);
  char* big = (char*)lra_ensure(ctx, 85, tszB + esz + 1024);               // its own buffer: callers hold pointers into scratch 0 across a sort
  int* stat = (int*)lra_ensure(ctx, 97, 64);                                 // slots 97 / 98 belong to the sort alone (86 / 87 hold fine_clusters.hip's results across sorts)
  if (!tscr || !big || !stat) return LRA_ERR_NOMEM;
  uint32_t* tscrB = (uint32_t*)big; char* gscr = big + tszB;
  int* flags = (int*)(tscr + (size_t)grid * (cap + 64));
  LRA_HIP_CHECK(ctx, hipMemsetAsync(flags, 0, (size_t)n_reads * 4, st));
  LRA_HIP_CHECK(ctx, hipMemsetAsync(stat, 0, 12, st));
  LRA_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)sort_wg_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  LRA_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)sort_wg_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsB));
  lra_time_begin(ctx, ctx->sort_tag);
  // Lists of at most capS tuples first, 256 threads and ~36 KB of LDS each: four of them fit a CU where the 1024-thread / 156 KB launch holds one, and a short list's sort
  // is a chain of barriers either way (the sparse DP's value lists: hundreds of tuples, tens of thousands of lists).  Then the rest, as before.
  static const int capS = getenv("LRA_SORT_SMALL_CAP") ? std::max(0, std::min(cap, atoi(getenv("LRA_SORT_SMALL_CAP")))) : 2048;
  if (capS >= 64 && ctx->sort_short) {
    const int maxsegS = (capS / 16 + 8 + 1) & ~1;
    const size_t ldsS = (size_t)capS * 16 + (size_t)maxsegS * 20 + 8 + (size_t)(capS / 32 + 2) * 4;
    const int gridS = std::min(n_reads, std::min(ctx->num_cu * 4, (int)(((size_t)grid * (cap + 64)) / (size_t)(capS + 64))));
    hipLaunchKernelGGL(sort_wg_kernel<0>, dim3(gridS), dim3(256), ldsS, st, n_reads, mm_off, mm_key, mm_pos, (void*)tscr, capS, (int*)nullptr, only, (char*)nullptr, (int*)nullptr, 0);
  }
  hipLaunchKernelGGL(sort_wg_kernel<0>, dim3(grid), dim3(SORT_NT), lds, st, n_reads, mm_off, mm_key, mm_pos, (void*)tscr, cap, flags, only, (char*)nullptr, stat, (capS >= 64 && ctx->sort_short) ? capS + 1 : 0);
  lra_time_end(ctx);
  // the launch for lists beyond the LDS capacity (1024-thread workgroups again, 64 of them, each waiting for room beside another batch's half) only when the LDS launch left
  // a list behind: it says how many, and the round trip that asks takes the place of the one behind the large-list launch
  int h_left = 1;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&h_left, stat + 2, 4, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  if (h_left == 0) return LRA_OK;
  lra_time_begin(ctx, ctx->sort_tag);
  hipLaunchKernelGGL(sort_wg_kernel<1>, dim3(gridB), dim3(SORT_NT), ldsB, st, n_reads, mm_off, mm_key, mm_pos, (void*)tscrB, capB, flags, only, gscr, stat);
  lra_time_end(ctx);
  // what is left: lists of more than 65534 tuples (the minimizers of a contig of several hundred kb) -- how long, how many
  int h_stat[2] = {0, 0};
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(h_stat, stat, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  if (h_stat[1] == 0) return LRA_OK;
  const int capH = std::min(h_stat[0] + 64, (1 << 26));                    // (a list beyond 2^26 tuples stays with the one-lane kernel)
  // (a workgroup per list beyond 65534 tuples, as many at a time as the device has CUs: a -CONTIG batch of 1024 contigs has 1024 such lists of ~180 k tuples, 7 MB of
  // scratch each; with a quarter of the CUs the launch took 370 ms)
  const int gridH = std::min(h_stat[1], std::max(1, ctx->num_cu));
  const size_t eszH = sort_scratch_bytes(2, (size_t)capH), tszH = (size_t)(capH + 64) * 8;
  char* huge = (char*)lra_ensure(ctx, 98, (size_t)gridH * (eszH + tszH) + 1024);
  if (!huge) return LRA_ERR_NOMEM;
  lra_time_begin(ctx, ctx->sort_tag);
  hipLaunchKernelGGL(sort_wg_kernel<2>, dim3(gridH), dim3(SORT_NT), 0, st, n_reads, mm_off, mm_key, mm_pos, (void*)huge, capH, flags, only, huge + (size_t)gridH * tszH, (int*)nullptr);
  lra_time_end(ctx);
  if (h_stat[0] > capH) {
    lra_time_begin(ctx, ctx->sort_fb_tag);
    hipLaunchKernelGGL(sort_kernel, dim3(nb), dim3(64), 0, st, n_reads, mm_off, mm_key, mm_pos, (const int*)flags);
    lra_time_end(ctx);
  }
  return LRA_OK;
}

// A list whose keys are all different has only one sorted order, so any sort reproduces std::sort on it.  Lists are radix-sorted into
// (tmp_key, tmp_pos); one block per list then looks for two equal neighbours: without any, the sorted list is copied over the input,
// with one the input is left alone and the list is marked for the exact sort.
__global__ void __launch_bounds__(256) sort_adopt_kernel(int n_lists, const uint64_t* __restrict__ off, uint64_t* key, uint32_t* pos,
                                                         const uint64_t* __restrict__ tkey, const uint32_t* __restrict__ tpos, int* ties, int* n_ties) {
  const int r = blockIdx.x;
  const uint64_t b = off[r], e = off[r + 1];
  int tie = 0;
  for (uint64_t i = b + threadIdx.x; i + 1 < e; i += 256) tie |= (tkey[i] & FOR_MASK) == (tkey[i + 1] & FOR_MASK);   // (what the exact sort's comparison sees: bit 63 is a flag)
  tie = __syncthreads_or(tie);
  if (tie) { if (threadIdx.x == 0) { ties[r] = 1; atomicAdd(n_ties, 1); } return; }
  if (threadIdx.x == 0) ties[r] = 0;
  for (uint64_t i = b + threadIdx.x; i < e; i += 256) { key[i] = tkey[i]; pos[i] = tpos[i]; }
}

// Same result as lra_sort_minimizers_batch, for keys that rarely repeat within a list (the sparse DP's point orders): keys below
// 2^end_bit, `total` = d_off[n_lists], (tmp_key, tmp_pos) = scratch of the same size as the input.
int lra_sort_mostly_unique_batch(lra_ctx* ctx, int n_lists, const uint64_t* d_off, uint64_t total, uint64_t* d_key, uint32_t* d_pos,
                                 uint64_t* tmp_key, uint32_t* tmp_pos, int end_bit) {
  if (n_lists == 0 || total == 0) return LRA_OK;
  hipStream_t st = ctx->stream;
  size_t temp_bytes = 0;
  (void)lra_segsort_pairs(ctx, nullptr, temp_bytes, nullptr, nullptr, nullptr, nullptr, (unsigned int)total, (unsigned int)n_lists, nullptr, nullptr, 0, end_bit, st);
  char* temp = (char*)lra_scratch(ctx, 2, temp_bytes + 512 + (size_t)n_lists * 4);
  if (!temp) return LRA_ERR_NOMEM;
  int* ties = (int*)(temp + ((temp_bytes + 255) & ~(size_t)255));
  int* n_ties = ties + n_lists + 1;
  LRA_HIP_CHECK(ctx, hipMemsetAsync(n_ties, 0, 4, st));
  lra_time_begin(ctx, ctx->sort_tag);
  hipError_t e = lra_segsort_pairs(ctx, temp, temp_bytes, d_key, tmp_key, d_pos, tmp_pos, (unsigned int)total, (unsigned int)n_lists, d_off, d_off + 1, 0, end_bit, st);
  if (e != hipSuccess) { lra_time_end(ctx); return lra_set_err(ctx, LRA_ERR_HIP, "segmented sort: %s", hipGetErrorString(e)); }
  hipLaunchKernelGGL(sort_adopt_kernel, dim3(n_lists), dim3(256), 0, st, n_lists, d_off, d_key, d_pos, (const uint64_t*)tmp_key, (const uint32_t*)tmp_pos, ties, n_ties);
  lra_time_end(ctx);
  // Lists with a repeated key are rare in the sparse DP's point orders: without any, the exact sort's three launches -- workgroups that need a whole CU's LDS each, and wait
  // for it beside another batch's half -- and their closing round trip are left out (this round trip takes its place)
  int h_ties = 1;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&h_ties, n_ties, 4, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  if (h_ties == 0) return LRA_OK;
  // lra_scratch slot 0 is launch_sort's own; the ties mask lives in slot 2 and stays valid through it
  int rc = launch_sort(ctx, n_lists, d_off, d_key, d_pos, ties);
  if (rc) return rc;
  LRA_HIP_CHECK(ctx, hipGetLastError());
  return LRA_OK;
}

extern "C" int lra_sort_minimizers_batch(lra_ctx* ctx, int n_lists, const uint64_t* d_off, uint64_t* d_key, uint32_t* d_pos) {
  if (!ctx || n_lists < 0) return LRA_ERR_INVALID;
  if (n_lists == 0) return LRA_OK;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int rc = launch_sort(ctx, n_lists, d_off, d_key, d_pos);
  if (rc) return rc;
  LRA_HIP_CHECK(ctx, hipGetLastError());
  return LRA_OK;
}

extern "C" int lra_seed_batch(lra_ctx* ctx, int n_reads, const char* d_seq, const uint64_t* d_read_off, int k, int w,
                              int max_freq, lra_seed_result* out) {
  if (!ctx || !out || n_reads < 0) return LRA_ERR_INVALID;
  if (k < 1 || k > 32 || w < 1 || w > MAX_W) return lra_set_err(ctx, LRA_ERR_INVALID, "k must be 1..32 and w 1..%d", MAX_W);
  lra_seed_state* s = seed_state(ctx);
  ctx->ahead.valid = false;                                               // (this call overwrites the buffers of a result adopted ahead of its batch, if there is one)
  if (!s->genome || !s->idx_key) return lra_set_err(ctx, LRA_ERR_INVALID, "load genome and global index first");
  { int rcs = lra_seed_check_shared(ctx); if (rcs) return rcs; }
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  memset(out, 0, sizeof(*out));
  out->n_reads = n_reads;
  if (n_reads == 0) return LRA_OK;
  if ((size_t)n_reads > s->cap_reads) {
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
    size_t c = (size_t)n_reads + n_reads / 4 + 64;
    if (!regrow(s->counts32, c) || !regrow(s->counts64, c) || !regrow(s->mm_off, c + 1) || !regrow(s->match_off, c + 1) ||
        !regrow(s->n_forward, c) || !regrow(s->cap_cnt, c) || !regrow(s->cap_off, c + 1))
      { s->cap_reads = 0; return lra_set_err(ctx, LRA_ERR_NOMEM, "per-read arrays"); }   // (capacity 0: whatever the failed group holds is re-made by the next call)
    s->cap_reads = c;
  }
  const unsigned char* seq = (const unsigned char*)d_seq;
  const int nb = (n_reads + 63) / 64;
  // ---- a1: count, scan, emit
  const int gridW = n_reads < ctx->num_cu * 32 ? n_reads : ctx->num_cu * 32;
  int* flagN = (int*)s->n_forward;   // reused before a4 writes it
  uint64_t total_bases = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&total_bases, d_read_off + n_reads, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  // (the staging arrays -- 12 bytes per base of the batch -- live in the sparse DP's arena, slot 12: the seed stage comes first in a batch, the arena is dead until the
  // first sparse DP, and what the batch before left there -- IndelRefine's and CalculateStatistics' arrays -- belonged to a result that ends with this call)
  char* stg = (char*)lra_ensure(ctx, 12, (size_t)total_bases * 12 + 1024);
  if (!stg) return LRA_ERR_NOMEM;
  uint64_t* wkey = (uint64_t*)stg;
  uint32_t* wpos = (uint32_t*)(stg + (((size_t)total_bases * 8 + 255) & ~(size_t)255));
  lra_time_begin(ctx, "sketch_emit");
  hipLaunchKernelGGL(sketch_wave_kernel<true>, dim3(gridW), dim3(64), 0, st, n_reads, seq, d_read_off, k, w, d_read_off, wkey, wpos, s->counts32, flagN);
  lra_time_end(ctx);
  lra_time_begin(ctx, "sketch_serial");
  hipLaunchKernelGGL(sketch_kernel<false>, dim3(nb), dim3(64), 0, st, n_reads, seq, d_read_off, k, w, (const uint64_t*)nullptr,
                     (uint64_t*)nullptr, (uint32_t*)nullptr, s->counts32, (const int*)flagN);
  lra_time_end(ctx);
  if (lra_exclusive_scan<uint32_t>(ctx, (long)n_reads, s->counts32, s->mm_off)) return LRA_ERR_HIP;
  uint64_t total_mm = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&total_mm, s->mm_off + n_reads, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  if (total_mm >= (1ULL << 32)) return lra_set_err(ctx, LRA_ERR_INVALID, "batch too large: %llu minimizers", (unsigned long long)total_mm);
  if (total_mm > s->cap_mm) {
    size_t c = total_mm + total_mm / 4 + 1024;
    if (!regrow(s->mm_key, c) || !regrow(s->mm_pos, c) || !regrow(s->lb, c) || !regrow(s->ub, c) || !regrow(s->tk_lb, c) ||
        !regrow(s->tk_lbm1, c) || !regrow(s->tk_ubm1, c))
      { s->cap_mm = 0; return lra_set_err(ctx, LRA_ERR_NOMEM, "minimizer arrays"); }
    s->cap_mm = c;
  }
  lra_time_begin(ctx, "sketch_compact");
  hipLaunchKernelGGL(sketch_compact, dim3(std::min((n_reads + 3) / 4, ctx->num_cu * 32)), dim3(256), 0, st, n_reads, d_read_off, (const uint64_t*)s->mm_off, (const int*)flagN,
                     (const uint64_t*)wkey, (const uint32_t*)wpos, s->mm_key, s->mm_pos);
  lra_time_end(ctx);
  lra_time_begin(ctx, "sketch_serial");
  hipLaunchKernelGGL(sketch_kernel<true>, dim3(nb), dim3(64), 0, st, n_reads, seq, d_read_off, k, w, s->mm_off, s->mm_key, s->mm_pos,
                     (uint32_t*)nullptr, (const int*)flagN);
  lra_time_end(ctx);
  // ---- a2: a read outside the repeats has no k-mer twice among its minimizers, and a list without equal keys has one sorted order only: the radix path takes nearly
  // all reads, the exact (libstdc++-identical) sort the ones with a repeated k-mer.  The temporaries are a3's outputs, not written yet.
  static const bool exactOnly = getenv("LRA_SEED_EXACT_SORT") != nullptr;
  if (exactOnly || total_mm == 0) { int rc = launch_sort(ctx, n_reads, s->mm_off, s->mm_key, s->mm_pos); if (rc) return rc; }
  else { int rc = lra_sort_mostly_unique_batch(ctx, n_reads, s->mm_off, total_mm, s->mm_key, s->mm_pos, s->tk_lb, s->lb, std::min(2 * k, 63)); if (rc) return rc; }
  // ---- a3
  if (total_mm) {
    lra_time_begin(ctx, "index_bounds");
    hipLaunchKernelGGL(bounds_kernel, dim3((unsigned)((total_mm + 255) / 256)), dim3(256), 0, st, total_mm, s->mm_key, s->idx_key, s->n_idx,
                       s->dir, s->nbuckets, s->dir_shift, s->lb, s->ub, s->tk_lb, s->tk_lbm1, s->tk_ubm1);
    lra_time_end(ctx);
  }
  hipLaunchKernelGGL(match_capacity_kernel, dim3(gridW), dim3(64), 0, st, n_reads, s->mm_off, s->lb, s->ub, s->cap_cnt);
  if (lra_exclusive_scan<uint64_t>(ctx, (long)n_reads, s->cap_cnt, s->cap_off)) return LRA_ERR_HIP;
  uint64_t total_cap = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&total_cap, s->cap_off + n_reads, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  if (total_cap > s->cap_tmp) {
    size_t c = total_cap + total_cap / 4 + 1024;
    if (!regrow(s->tmp_qi, c) || !regrow(s->tmp_ti, c)) { s->cap_tmp = 0; return lra_set_err(ctx, LRA_ERR_NOMEM, "match walk buffers (%llu)", (unsigned long long)total_cap); }
    s->cap_tmp = c;
  }
  // reads per wave of the walk (a lane per read): 32 for a full batch (1024 waves; more waves cost more rounds than the divergence of 32 walks costs), fewer for a small one --
  // a batch of 256 contigs as 256 one-lane waves walks 9 % faster than as 8 waves of 32 diverging lanes (-CONTIG: 3787 -> 3461 ms per batch)
  static const int lanesEnv = getenv("LRA_COMPARE_LANES") ? std::max(1, std::min(64, atoi(getenv("LRA_COMPARE_LANES")))) : 0;
  const int FLAT_LANES = lanesEnv ? lanesEnv : std::max(1, std::min(32, n_reads / 1024));
  lra_time_begin(ctx, "compare");
  hipLaunchKernelGGL(compare_kernel, dim3((n_reads + FLAT_LANES - 1) / FLAT_LANES), dim3(64), 0, st, FLAT_LANES, n_reads, s->mm_off, s->mm_key, s->lb, s->ub, s->tk_lb, s->tk_lbm1, s->tk_ubm1, s->idx_key,
                     (long)s->n_idx, (long)max_freq, s->cap_off, s->tmp_qi, s->tmp_ti, s->counts64);
  lra_time_end(ctx);
  if (s->defer_T) {
    if ((size_t)n_reads > s->cap_defer) {
      LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
      if (!regrow(s->defer_flag, (size_t)n_reads + n_reads / 4 + 64)) { s->cap_defer = 0; return lra_set_err(ctx, LRA_ERR_NOMEM, "defer flags"); }
      s->cap_defer = (size_t)n_reads + n_reads / 4 + 64;
    }
    hipLaunchKernelGGL(defer_heavy_kernel, dim3((n_reads + 255) / 256), dim3(256), 0, st, n_reads, s->defer_T, s->counts64, s->defer_flag);
  }
  if (lra_exclusive_scan<uint64_t>(ctx, (long)n_reads, s->counts64, s->match_off)) return LRA_ERR_HIP;
  uint64_t total_m = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&total_m, s->match_off + n_reads, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  if (total_m > total_cap) return lra_set_err(ctx, LRA_ERR_INVALID, "match capacity bound violated (%llu > %llu)", (unsigned long long)total_m, (unsigned long long)total_cap);
  if (total_m > s->cap_match) {
    size_t c = total_m + total_m / 4 + 1024;
    if (!regrow(s->match_qi, c) || !regrow(s->match_ti, c) || !regrow(s->sep_qpos, c) || !regrow(s->sep_tpos, c) || !regrow(s->sep_qkey, c))
      { s->cap_match = 0; return lra_set_err(ctx, LRA_ERR_NOMEM, "match arrays (%llu matches)", (unsigned long long)total_m); }
    s->cap_match = c;
  }
  // ---- a4
  lra_time_begin(ctx, "strand");
  hipLaunchKernelGGL(strand_kernel, dim3(n_reads < 4096 ? n_reads : 4096), dim3(64), 0, st, n_reads, seq, d_read_off, s->genome, k, s->mm_off,
                     s->mm_pos, s->idx_pos, s->match_off, s->cap_off, s->tmp_qi, s->tmp_ti, s->match_qi, s->match_ti, s->sep_qpos, s->sep_tpos, s->n_forward, s->mm_key, s->sep_qkey);
  lra_time_end(ctx);
  LRA_HIP_CHECK(ctx, hipGetLastError());
  s->last_n_reads = n_reads; s->last_n_matches = total_m;
  out->n_minimizers = total_mm; out->n_matches = total_m;
  out->d_mm_off = s->mm_off; out->d_mm_key = s->mm_key; out->d_mm_pos = s->mm_pos;
  out->d_match_off = s->match_off; out->d_match_qi = s->match_qi; out->d_match_ti = s->match_ti;
  out->d_n_forward = s->n_forward; out->d_sep_qpos = s->sep_qpos; out->d_sep_tpos = s->sep_tpos;
  return LRA_OK;
}

extern "C" int lra_seed_prefetch(lra_ctx* side, int n_reads, const char* d_seq, const uint64_t* d_read_off, int k, int w, int max_freq) {
  if (!side) return LRA_ERR_INVALID;
  side->ahead.valid = false;
  lra_seed_result res;
  int rc = lra_seed_batch(side, n_reads, d_seq, d_read_off, k, w, max_freq, &res);
  if (rc) return rc;
  LRA_HIP_CHECK(side, hipStreamSynchronize(side->stream));               // (the strand pass is queued, not waited for, by lra_seed_batch)
  side->ahead.n_reads = n_reads; side->ahead.d_seq = d_seq; side->ahead.d_read_off = d_read_off; side->ahead.k = k; side->ahead.w = w; side->ahead.max_freq = max_freq;
  side->ahead.res = res; side->ahead.valid = true;
  return LRA_OK;
}

// The two contexts exchange the seed stage's batch buffers (the result's pointers go with them); the reference data, its ownership and the directory stay.
extern "C" int lra_ctx_adopt_seed(lra_ctx* ctx, lra_ctx* side) {
  if (!ctx || !side || ctx == side) return LRA_ERR_INVALID;
  if (ctx->device != side->device) return lra_set_err(ctx, LRA_ERR_INVALID, "the side context is on another device");
  if (!side->ahead.valid) return lra_set_err(ctx, LRA_ERR_INVALID, "the side context holds no prefetched seed result");
  lra_seed_state* a = seed_state(ctx); lra_seed_state* b = seed_state(side);
  { int rcs = lra_seed_check_shared(side); if (rcs) { side->ahead.valid = false; return lra_set_err(ctx, LRA_ERR_INVALID, "the side context's reference data is stale: share it again"); } }
  if (a->genome != b->genome || a->idx_key != b->idx_key || a->idx_pos != b->idx_pos || a->n_idx != b->n_idx)
    return lra_set_err(ctx, LRA_ERR_INVALID, "the side context does not hold this context's reference data (lra_ctx_share_reference)");
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));                 // nothing of the mapping context's last batch still reads its seed buffers
#define SW(f) std::swap(a->f, b->f)
  SW(counts32); SW(counts64); SW(mm_off); SW(match_off); SW(n_forward); SW(cap_reads);
  SW(mm_key); SW(mm_pos); SW(lb); SW(ub); SW(cap_mm); SW(tk_lb); SW(tk_lbm1); SW(tk_ubm1);
  SW(match_qi); SW(match_ti); SW(sep_qpos); SW(sep_tpos); SW(sep_qkey); SW(cap_match);
  SW(last_n_reads); SW(last_n_matches); SW(defer_flag); SW(cap_defer);
  SW(tmp_qi); SW(tmp_ti); SW(cap_tmp); SW(cap_cnt); SW(cap_off);
#undef SW
  ctx->ahead = side->ahead;
  side->ahead.valid = false;
  return LRA_OK;
}
