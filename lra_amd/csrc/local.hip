// lra_amd/csrc/local.hip -- tier-2 (local) minimizer index and lookups on gfx950.
//
//   lra_local_index_batch     LocalIndex::IndexSeq (MMIndex.h:200-245): per 256-base window the
//                             non-canonical (w,k)-minimizers (StoreMinimizers_noncanonical,
//                             MinCount.h:182-338), std::sort by k-mer (MMIndex.h:219), RemoveFrequent
//                             (MMIndex.h:69-84).  Used for reads (both strands, Map_lowacc.h:246-250) and
//                             for genomes (LocalIndex::IndexFile, MMIndex.h:247-254 = the `.gli` payload).
//   lra_local_compare_batch   CompareLists<LocalTuple,SmallTuple> with Global=false and a diagonal band
//                             (CompareLists.h:9-146) over (read-window list, genome-window list) tasks,
//                             the lookup Refine_splitchain / REFINEclusters perform (ChainRefine.h:384,
//                             ClusterRefine.h:50).
// A LocalTuple is the word  t | pos << 20  (TupleOps.h:20-25).  Windows hold ~40-80 tuples, so every
// window / task is one lane: 7.7 M windows per 32 k-read batch keep the chip full.
#include "common.h"
#include <vector>
#include "scan.h"
#include <algorithm>
#include <hipcub/hipcub.hpp>

namespace {

constexpr uint32_t TMASK = 0xFFFFF;
__device__ __forceinline__ uint32_t T_(uint32_t v) { return v & TMASK; }
__device__ __forceinline__ uint32_t P_(uint32_t v) { return v >> 20; }

__device__ __forceinline__ int code_n(unsigned char c) {
  if (c < 8) return c & 3;
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
  }
}
__device__ __forceinline__ uint32_t code2(unsigned char c) { int v = code_n(c); return v > 3 ? 0u : (uint32_t)v; }

constexpr int MAXW = 16;

// window -> (sequence, start, length)
// (sequences masked out keep their windows -- the window numbering is part of the index -- but with length 0: no tuples)
__global__ void window_map(int n_seqs, const uint64_t* __restrict__ seq_off, int window, const uint64_t* __restrict__ win_off,
                           const uint8_t* __restrict__ active, uint32_t* __restrict__ w_seq, uint64_t* __restrict__ w_start, uint32_t* __restrict__ w_len) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_seqs) return;
  const uint64_t a = seq_off[s], L = seq_off[s + 1] - a;
  const bool on = !active || active[s];
  uint64_t wi = win_off[s];
  for (uint64_t p = 0; p < L; p += window, wi++) { w_seq[wi] = s; w_start[wi] = a + p; w_len[wi] = on ? (uint32_t)std::min<uint64_t>((uint64_t)window, L - p) : 0u; }
}
__global__ void window_count(int n_seqs, const uint64_t* __restrict__ seq_off, int window, uint32_t* __restrict__ nwin) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_seqs) return;
  const uint64_t L = seq_off[s + 1] - seq_off[s];
  nwin[s] = (uint32_t)(L / window + (L % window != 0 ? 1 : 0));        // MMIndex.h:201-206
}

// StoreMinimizers_noncanonical for one window per lane.  EMIT=false counts.  The 64 windows of a wave
// are first copied into LDS with coalesced loads (lane-private rows of WSTRIDE bytes, an odd number of
// dwords apart so the per-lane byte reads spread over the banks): the serial scan then never touches
// HBM (the unstaged version fetched ~40x the sequence bytes because 64 private streams thrash L1).
constexpr int WMAX = 256;            // staged window length; longer windows read from HBM directly
// One pass: window wi writes its tuples at raw + wi * stride (stride = window - k + 1 slots: a window cannot emit more tuples than it has
// k-mer positions) and its count; the sort / filter and the compaction read the slab through (wi * stride, counts[wi]).
__device__ __forceinline__ uint32_t spread16(uint32_t x) {            // bit i of x -> bit 2 i
  x = (x | (x << 8)) & 0x00FF00FFu; x = (x | (x << 4)) & 0x0F0F0F0Fu; x = (x | (x << 2)) & 0x33333333u; x = (x | (x << 1)) & 0x55555555u;
  return x;
}
// The serial scan of a window is a chain of LDS round trips (base, ring), so the kernel's time is that chain over the waves a CU holds: the window's bases are staged
// as 2-bit codes + an N mask (24 words per lane instead of 260 bytes; a word serves 16 positions from a register), the ring entry is the packed tuple itself
// (t | pos << 20) and the ring has w slots, not MAXW: ~7 KB of LDS per wave instead of 25 -- 21 waves per CU instead of 6.
__global__ void __launch_bounds__(64) local_sketch(uint64_t n_win, const unsigned char* __restrict__ seq_all, const uint64_t* __restrict__ w_start,
                                                   const uint32_t* __restrict__ w_len, int k, int w, uint64_t stride,
                                                   uint32_t* __restrict__ raw, uint32_t* __restrict__ counts) {
  constexpr bool EMIT = true;
  extern __shared__ uint32_t ls_lds[];
  uint32_t* codes = ls_lds;                         // [WMAX / 16][64]: 16 codes per word, lane-interleaved
  uint32_t* nmask = codes + (WMAX / 16) * 64;       // [WMAX / 32][64]
  uint32_t* ring = nmask + (WMAX / 32) * 64;        // [w][64]
  const int lane = threadIdx.x;
  const uint64_t w0 = (uint64_t)blockIdx.x * 64;
  const uint64_t wi = w0 + lane;
  // cooperative staging of the wave's windows: 64 bases per step, packed with three ballots
  for (int x = 0; x < 64; x++) {
    const uint64_t wx = w0 + x;
    if (wx >= n_win) break;
    const uint32_t L = w_len[wx];
    if (L > WMAX) continue;
    const unsigned char* src = seq_all + w_start[wx];
    for (uint32_t it = 0; it * 64 < L; it++) {
      const uint32_t p = it * 64 + lane;
      const int c = p < L ? code_n(src[p]) : 0;
      const unsigned long long b0 = __ballot(c & 1), b1 = __ballot(c & 2), bn = __ballot(c > 3);
      if (lane < 4) codes[(it * 4 + lane) * 64 + x] = spread16((uint32_t)(b0 >> (16 * lane)) & 0xFFFFu) | (spread16((uint32_t)(b1 >> (16 * lane)) & 0xFFFFu) << 1);
      if (lane < 2) nmask[(it * 2 + lane) * 64 + x] = (uint32_t)(bn >> (32 * lane));
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  if (wi >= n_win) return;
  const uint32_t seqLen = w_len[wi];
  const bool staged = seqLen <= WMAX;
  const unsigned char* gseq = seq_all + w_start[wi];
  auto C2 = [&](uint32_t p) -> uint32_t { return staged ? (codes[(p >> 4) * 64 + lane] >> ((p & 15) * 2)) & 3u : code2(gseq[p]); };       // seqMap: N -> 0
  auto ISN = [&](uint32_t p) -> bool { return staged ? ((nmask[(p >> 5) * 64 + lane] >> (p & 31)) & 1u) != 0 : code_n(gseq[p]) > 3; };
  uint32_t* out = raw + wi * stride;
  uint32_t n = 0;
#define LS_EMIT(T__, P__) do { if (EMIT) out[n] = ((T__) & TMASK) | (((P__) & 0xFFFu) << 20); n++; } while (0)
#define LS_DONE() do { counts[wi] = n; return; } while (0)
  const int span = w + k - 1;
  if (seqLen < (uint32_t)k || seqLen < (uint32_t)span) LS_DONE();        // :186,:199
  const uint32_t kmask = (k >= 16) ? 0xFFFFFFFFu : ((1u << (2 * k)) - 1);
  long nvStart = 0, nvEnd = 0;
  bool valid = false;
  auto find_valid = [&]() -> bool {                                      // :200-214
    valid = false;
    while ((uint32_t)nvStart < seqLen - (uint32_t)span && !valid) {
      valid = true;
      for (long x = nvStart; valid && x < nvStart + span; x++)
        if (ISN((uint32_t)x)) { nvStart = x + 1; valid = false; }
    }
    return valid;
  };
  if (!find_valid()) LS_DONE();
  nvEnd = nvStart + span;
  uint32_t cur = 0;
  for (int p = 0; p < k; p++) cur = ((cur << 2) + C2(p)) & TMASK;
  // the stream of bases the scan shifts in, position after position: a staged window's words are held in registers for 16 / 32 positions
  uint32_t cw = 0, nw = 0;
  auto next_base = [&](uint32_t at, uint32_t& c2, bool& isn) {
    if (staged) {
      if ((at & 15) == 0 || at == (uint32_t)k) cw = codes[(at >> 4) * 64 + lane];
      if ((at & 31) == 0 || at == (uint32_t)k) nw = nmask[(at >> 5) * 64 + lane];
      c2 = (cw >> ((at & 15) * 2)) & 3u; isn = ((nw >> (at & 31)) & 1u) != 0;
    } else { const int c = code_n(gseq[at]); c2 = c > 3 ? 0u : (uint32_t)c; isn = c > 3; }
  };
  auto shift_in = [&](uint32_t c2) { cur = ((((cur << 2) & TMASK) & kmask) + c2) & TMASK; };
  uint32_t actT = cur, actP = 0;
  ring[lane] = actT;                                                      // (position 0 in the high bits)
  // ring entries carry p in their 12 high bits: exact because a window has at most 4096 positions (lra_local_index_masked_batch rejects window > 4096,
  // the width of LocalTuple::pos)
  uint32_t p;
  const uint32_t nk = seqLen - k + 1;
  for (p = 1; p < (uint32_t)w && p < nk; p++) {                          // :251-270
    uint32_t c2; bool isn; next_base(p + k - 1, c2, isn);
    shift_in(c2);
    if (cur < actT) { actT = cur; actP = p; }
    ring[(p % w) * 64 + lane] = cur | (p << 20);
  }
  if (nvEnd == span) LS_EMIT(actT, actP);
  uint32_t slot = 0;                                                     // p % w for p = w
  for (p = w; p < nk; p++) {                                             // :276-337
    uint32_t c2; bool isn; next_base(p + k - 1, c2, isn);
    shift_in(c2);
    if (nvEnd == (long)(p + k - 1)) {
      if (!isn) nvEnd++;
      else {
        nvStart = p + k;
        if (!find_valid()) LS_DONE();
        nvEnd = nvStart + span;
      }
    }
    ring[slot * 64 + lane] = cur | (p << 20);
    if (++slot == (uint32_t)w) slot = 0;
    if (p - w >= actP) {
      { const uint32_t e = ring[lane]; actT = e & TMASK; actP = e >> 20; }
      for (int j = 1; j < w; j++) { const uint32_t e = ring[j * 64 + lane]; if ((e & TMASK) < actT) { actT = e & TMASK; actP = e >> 20; } }
      if (nvEnd == (long)(p + k)) LS_EMIT(actT, actP);
    } else if (cur < actT) {
      actT = cur; actP = p;
      if (nvEnd == (long)(p + k)) LS_EMIT(actT, actP);
    }
  }
  LS_DONE();
#undef LS_EMIT
#undef LS_DONE
}

// ---- libstdc++ std::sort on a list of LocalTuple words (comparison on t only), one lane per list
__device__ __forceinline__ bool wlt(uint32_t a, uint32_t b) { return T_(a) < T_(b); }

__device__ void w_adjust_heap(uint32_t* v, long first, long hole, long len, uint32_t val) {
  const long top = hole;
  long child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (wlt(v[first + child], v[first + child - 1])) child--;
    v[first + hole] = v[first + child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    v[first + hole] = v[first + child - 1];
    hole = child - 1;
  }
  long parent = (hole - 1) / 2;
  while (hole > top && wlt(v[first + parent], val)) { v[first + hole] = v[first + parent]; hole = parent; parent = (hole - 1) / 2; }
  v[first + hole] = val;
}

__device__ void w_std_sort(uint32_t* v, long n) {
  if (n < 2) return;
  long stF[40], stL[40]; int stD[40];
  int sp = 0;
  stF[0] = 0; stL[0] = n; stD[0] = 2 * (63 - __clzll((unsigned long long)n)); sp = 1;
  while (sp > 0) {
    --sp;
    long first = stF[sp], last = stL[sp]; int depth = stD[sp];
    while (last - first > 16) {
      if (depth == 0) {                                                  // __partial_sort(first,last,last)
        long len = last - first;
        for (long parent = (len - 2) / 2;; parent--) { w_adjust_heap(v, first, parent, len, v[first + parent]); if (parent == 0) break; }
        long l2 = last;
        while (l2 - first > 1) { --l2; uint32_t val = v[l2]; v[l2] = v[first]; w_adjust_heap(v, first, 0, l2 - first, val); }
        break;
      }
      --depth;
      const long a = first + 1, b = first + (last - first) / 2, c = last - 1;
      auto sw = [&](long x, long y) { uint32_t t = v[x]; v[x] = v[y]; v[y] = t; };
      if (wlt(v[a], v[b])) { if (wlt(v[b], v[c])) sw(first, b); else if (wlt(v[a], v[c])) sw(first, c); else sw(first, a); }
      else if (wlt(v[a], v[c])) sw(first, a);
      else if (wlt(v[b], v[c])) sw(first, c);
      else sw(first, b);
      long f = first + 1, l = last;
      const uint32_t pv = v[first];
      while (true) {
        while (wlt(v[f], pv)) ++f;
        --l;
        while (wlt(pv, v[l])) --l;
        if (!(f < l)) break;
        sw(f, l);
        ++f;
      }
      if (sp < 40) { stF[sp] = f; stL[sp] = last; stD[sp] = depth; sp++; }
      last = f;
    }
  }
  auto ins = [&](long first, long last) {                                // __insertion_sort
    for (long i = first + 1; i < last; ++i) {
      uint32_t val = v[i];
      if (wlt(val, v[first])) { for (long x = i; x > first; --x) v[x] = v[x - 1]; v[first] = val; }
      else { long j = i; while (wlt(val, v[j - 1])) { v[j] = v[j - 1]; --j; } v[j] = val; }
    }
  };
  if (n > 16) {
    ins(0, 16);
    for (long i = 16; i < n; ++i) { uint32_t val = v[i]; long j = i; while (wlt(val, v[j - 1])) { v[j] = v[j - 1]; --j; } v[j] = val; }
  } else ins(0, n);
}

// sort + RemoveFrequent in place; counts[wi] = surviving tuples.  Lists of <= LCAP tuples are sorted in a
// lane-private LDS row (copied in and out with coalesced accesses); longer ones in HBM.
constexpr int LCAP = 160;
constexpr int STAGE_NT = 256;                                           // 4 waves stage a block's 64 lists (memory parallelism), wave 0 works on them
// (a window holds ~45 tuples: the 64 lists of a block are packed behind one another in LS_WORDS words of LDS -- 24 KB, six blocks per CU -- instead of 64 rows of LCAP;
// a list that no longer fits, like one above LCAP, is sorted in HBM)
constexpr int LS_WORDS = 6144;
// sel != null: the kernel's windows are sel[0 .. *nSel) (the lists local_sort_unique left: a key twice), 64 per block as before.
__global__ void __launch_bounds__(STAGE_NT) local_sort_filter(uint64_t n_win_all, uint64_t stride, uint32_t* raw, int maxFreq, uint32_t* counts,
                                                              const uint32_t* __restrict__ sel, const uint64_t* __restrict__ nSel) {
  __shared__ uint32_t stage[LS_WORDS];
  __shared__ uint32_t kept[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t n_win = sel ? *nSel : n_win_all;
  const uint64_t w0 = (uint64_t)blockIdx.x * 64;
  if (w0 >= n_win) return;
  const uint64_t wi = (w0 + lane < n_win) ? (sel ? (uint64_t)sel[w0 + lane] : w0 + lane) : ~0ULL;   // the window this lane sorts (~0: none)
  const uint64_t myA = wi * stride, myN = wi != ~0ULL ? counts[wi] : 0;       // raw count in, filtered count out
  const uint32_t need = myN <= (uint64_t)LCAP ? (uint32_t)myN : 0u;
  uint32_t incl = need;
  for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= d) incl += o; }
  const uint32_t myOff = incl - need;
  const bool myFits = myN <= (uint64_t)LCAP && incl <= (uint32_t)LS_WORDS;
#pragma unroll 4
  for (int x = wave; x < 64; x += STAGE_NT / 64) {
    const uint64_t a = __shfl(myA, x), n = __shfl(myN, x);
    const uint32_t so = __shfl(myOff, x);
    if (!__shfl((int)myFits, x)) continue;
    for (uint32_t p = lane; p < n; p += 64) stage[so + p] = raw[a + p];
  }
  __syncthreads();
  if (wave == 0) {
    long c = 0;
    if (wi != ~0ULL) {
      const long n = (long)myN;
      const bool staged = myFits;
      // (once per address space, so that the list in LDS is read with ds_read and the one in HBM with global_load instead of flat_load through an either-or pointer)
      auto work = [&](uint32_t* v) __attribute__((always_inline)) {
        w_std_sort(v, n);                                                // MMIndex.h:219
        long x = 0;                                                      // RemoveFrequent MMIndex.h:69-84
        while (x < n) {
          long ne = x;
          while (ne < n && T_(v[ne]) == T_(v[x])) ne++;
          if (ne - x < maxFreq) for (long y = x; y < ne; y++) v[c++] = v[y];
          x = ne;
        }
      };
      if (staged) work(stage + myOff); else work(raw + myA);
      counts[wi] = (uint32_t)c;
    }
    kept[lane] = (uint32_t)c;
  }
  __syncthreads();
  // write the surviving tuples back (coalesced, one window per wave at a time)
  for (int x = wave; x < 64; x += STAGE_NT / 64) {
    const uint64_t a = __shfl(myA, x);
    const uint32_t so = __shfl(myOff, x);
    if (w0 + x >= n_win || !__shfl((int)myFits, x)) continue;
    const uint32_t cx = kept[x];
    for (uint32_t p = lane; p < cx; p += 64) raw[a + p] = stage[so + p];
  }
}

// ---- sort(minimizers) + RemoveFrequent for the windows whose tuples have pairwise different keys -- nearly all: two equal 10-mers among a window's ~85 minimizers need a
// repeat inside 256 bases.  Such a list has ONE sorted order (the comparison is on t only, and no two t are equal), so libstdc++'s permutation does not matter and the
// sort is a rank count: a wave per window, a lane per tuple, rank = the tuples with a smaller key (every key read once from LDS, broadcast); nothing is removed
// (every key occurs once, and 1 < maxFreq).  A window with a key twice, or with more than RANK_CAP tuples, is left as it is and flagged: local_sort_filter's exact
// sort takes those (one lane per list walking LDS: the chain the whole stage used to be, 19 ms).
constexpr int RANK_CAP = 256;
__global__ void __launch_bounds__(256) local_sort_unique(uint64_t n_win, uint64_t stride, uint32_t* raw, const uint32_t* __restrict__ counts, uint32_t* __restrict__ flag) {
  __shared__ uint32_t keys[4][RANK_CAP];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t wi = (uint64_t)blockIdx.x * 4 + wv;
  if (wi >= n_win) return;
  const uint32_t n = counts[wi];
  if (n < 2) { if (lane == 0) flag[wi] = 0; return; }
  if (n > (uint32_t)RANK_CAP) { if (lane == 0) flag[wi] = 1; return; }
  uint32_t* v = raw + wi * stride;
  uint32_t* K = keys[wv];
  uint32_t e[RANK_CAP / 64];
#pragma unroll
  for (int c = 0; c < RANK_CAP / 64; c++) { const uint32_t i = c * 64 + lane; e[c] = i < n ? v[i] : 0xFFFFFFFFu; if (i < n) K[i] = T_(e[c]); }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  uint32_t rank[RANK_CAP / 64] = {}, eq[RANK_CAP / 64] = {};
  const int nc = (int)((n + 63) / 64);
  for (uint32_t j = 0; j < n; j++) {
    const uint32_t tj = K[j];
#pragma unroll
    for (int c = 0; c < RANK_CAP / 64; c++) if (c < nc) { const uint32_t ti = T_(e[c]); rank[c] += tj < ti; eq[c] += tj == ti; }
  }
  bool tie = false;
#pragma unroll
  for (int c = 0; c < RANK_CAP / 64; c++) if (c < nc && (uint32_t)(c * 64 + lane) < n && eq[c] > 1) tie = true;
  const bool anyTie = __ballot(tie) != 0ULL;
  if (!anyTie) {
#pragma unroll
    for (int c = 0; c < RANK_CAP / 64; c++) if (c < nc && (uint32_t)(c * 64 + lane) < n) v[rank[c]] = e[c];
  }
  if (lane == 0) flag[wi] = anyTie ? 1u : 0u;
}
// The same for the longer lists (windows beyond 256 bases: the .gli file `lra index` writes has windows of 2048 -- LocalIndex(0), MMIndex.h:110-127 --, ~700 tuples a
// list, and the reads' indexes copy its window, Map_lowacc.h:246): a workgroup per window sorts the words in LDS by their low 20 bits (hipcub::BlockRadixSort, stable),
// neighbours are compared, and a list without a repeated key -- one sorted order, nothing for RemoveFrequent to remove -- is written back; one with a repeated key is left
// as it was, flagged, for the exact sort (the permutation libstdc++'s introsort leaves among equal keys is part of the index).  Size classes by NT * IPT; flag[wi] is 1 on
// entry for every list above RANK_CAP (local_sort_unique).
template <int NT, int IPT>
__global__ void __launch_bounds__(NT) local_sort_radix(uint64_t n_win, uint64_t stride, uint32_t* raw, const uint32_t* __restrict__ counts, uint32_t* __restrict__ flag, int minLen) {
  typedef hipcub::BlockRadixSort<uint32_t, NT, IPT> Sort;
  __shared__ typename Sort::TempStorage tmp;
  __shared__ uint32_t edge[NT + 1];
  __shared__ int dup;
  for (uint64_t wi = blockIdx.x; wi < n_win; wi += gridDim.x) {
    const int n = (int)counts[wi];
    if (n <= minLen || n > NT * IPT) continue;                            // (another class's; at most RANK_CAP: sorted already)
    uint32_t* v = raw + wi * stride;
    uint32_t k[IPT];
#pragma unroll
    for (int i = 0; i < IPT; i++) { const int p = (int)threadIdx.x * IPT + i; k[i] = p < n ? v[p] : 0xFFFFFFFFu; }   // (the padding: behind every tuple of its key -- the sort is stable)
    if (threadIdx.x == 0) dup = 0;
    Sort(tmp).Sort(k, 0, 20);
    edge[threadIdx.x] = k[0];
    __syncthreads();
    bool d = false;
#pragma unroll
    for (int i = 0; i < IPT; i++) {
      const int p = (int)threadIdx.x * IPT + i;
      const uint32_t nx = i + 1 < IPT ? k[i + 1 < IPT ? i + 1 : i] : (threadIdx.x + 1 < NT ? edge[threadIdx.x + 1] : 0xFFFFFFFFu);
      if (p + 1 < n && T_(k[i]) == T_(nx)) d = true;
    }
    if (d) dup = 1;
    __syncthreads();
    if (!dup) {
#pragma unroll
      for (int i = 0; i < IPT; i++) { const int p = (int)threadIdx.x * IPT + i; if (p < n) v[p] = k[i]; }
      if (threadIdx.x == 0) flag[wi] = 0;
    }
    __syncthreads();
  }
}

// ---- the exact sort of the long lists with a repeated key, a WAVE per list (local_sort_filter walks such a list with one lane: 52-62 ms per batch at windows of 2048
// bases, a third of the windows).  libstdc++'s introsort is data-parallel as it stands (seed.hip, sort_wg_kernel): the segments the loop recurses into are disjoint,
// and the unguarded Hoare partition of [first + 1, last) around *first is a closed form -- with a_0 < a_1 < .. the positions holding x >= pivot and b_0 > b_1 > .. those
// holding x <= pivot it swaps (a_i, b_i) for i < m = #{i: a_i < b_i} and returns cut = min(a_m, b_(m-1)); the final insertion sort equals a stable insertion sort of
// every block between two cuts.  Here the list lies in the wave's LDS; a segment is taken from a stack, its stoppers ranked by ballots into two position lists, m counted
// (the test is true on a prefix), the swaps done side by side, the children pushed; then a lane per leftover block; then RemoveFrequent (MMIndex.h:69-84) as three
// sweeps (run start, run end, compaction) and the list goes back in place.
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__global__ void __launch_bounds__(64) local_sort_exact_wave(uint64_t stride, uint32_t* raw, int maxFreq, uint32_t* counts, const uint32_t* __restrict__ sel,
                                                            const uint64_t* __restrict__ nSel, int cap, int minN) {
  extern __shared__ uint32_t xs_lds[];
  uint32_t* v = xs_lds;
  unsigned short* pa = (unsigned short*)(v + cap);
  unsigned short* pb = pa + cap;
  uint32_t* sbits = (uint32_t*)(pb + cap);                                 // block starts, cap / 32 + 1 words
  int* stF = (int*)(sbits + cap / 32 + 1); int* stL = stF + 64; int* stD = stL + 64;
  const int lane = threadIdx.x;
  const unsigned long long below = lane == 0 ? 0ULL : (~0ULL >> (64 - lane));
  const uint64_t nS = *nSel;
  for (uint64_t bi = blockIdx.x; bi < nS; bi += gridDim.x) {
    const uint64_t wi = sel[bi];
    const int n = (int)counts[wi];
    if (n > cap || n <= minN) continue;                                    // (another size class's launch: a list's LDS is its class's cap, and LDS is what limits the waves per CU)
    uint32_t* g = raw + wi * stride;
    for (int p = lane; p < n; p += 64) v[p] = g[p];
    for (int x = lane; x < cap / 32 + 1; x += 64) sbits[x] = x == 0 ? 1u : 0u;
    int sp = 0;
    if (lane == 0 && n > 16) { stF[0] = 0; stL[0] = n; stD[0] = 2 * (31 - __clz(n)); }
    if (n > 16) sp = 1;
    wave_sync_lds();
    while (sp > 0) {
      --sp;
      const int first = stF[sp], last = stL[sp]; int depth = stD[sp];
      wave_sync_lds();                                                     // (everyone has read the entry before it is overwritten)
      if (depth == 0) {                                                    // __partial_sort(first, last, last): heap sort, one lane
        if (lane == 0) {
          const long len = last - first;
          for (long parent = (len - 2) / 2;; parent--) { w_adjust_heap(v, first, parent, len, v[first + parent]); if (parent == 0) break; }
          long l2 = last;
          while (l2 - first > 1) { --l2; const uint32_t val = v[l2]; v[l2] = v[first]; w_adjust_heap(v, first, 0, l2 - first, val); }
        }
        wave_sync_lds();
        continue;
      }
      --depth;
      {                                                                    // __move_median_to_first(first, first + 1, mid, last - 1)
        const int a = first + 1, b = first + (last - first) / 2, c = last - 1;
        const uint32_t ka = T_(v[a]), kb = T_(v[b]), kc = T_(v[c]);
        int w;
        if (ka < kb) w = kb < kc ? b : (ka < kc ? c : a);
        else w = ka < kc ? a : (kb < kc ? c : b);
        if (lane == 0) { const uint32_t t0 = v[first]; v[first] = v[w]; v[w] = t0; }
      }
      wave_sync_lds();
      const uint32_t piv = T_(v[first]);
      const int f1 = first + 1;
      int nA = 0, nB = 0;
      for (int base = f1; base < last; base += 64) {
        const int p = base + lane;
        const bool ok = p < last;
        const uint32_t k = ok ? T_(v[p]) : 0u;
        const bool A = ok && k >= piv, B = ok && k <= piv;
        const unsigned long long mA = __ballot(A), mB = __ballot(B);
        if (A) pa[f1 + nA + __popcll(mA & below)] = (unsigned short)p;
        if (B) pb[f1 + nB + __popcll(mB & below)] = (unsigned short)p;
        nA += __popcll(mA); nB += __popcll(mB);
      }
      wave_sync_lds();
      const int lim = min(nA, nB);
      int m = 0;
      for (int i0 = 0; i0 < lim; i0 += 64) {
        const int i = i0 + lane;
        const bool ok = i < lim && pa[f1 + i] < pb[f1 + nB - 1 - i];
        const unsigned long long mk = __ballot(ok);
        m += __popcll(mk);
        if (__popcll(mk) < min(64, lim - i0)) break;
      }
      for (int i0 = 0; i0 < m; i0 += 64) {
        const int i = i0 + lane;
        if (i < m) { const int x = pa[f1 + i], y = pb[f1 + nB - 1 - i]; const uint32_t t0 = v[x]; v[x] = v[y]; v[y] = t0; }
      }
      int cut = 0x7fffffff;
      if (m < nA) cut = pa[f1 + m];
      if (m >= 1) cut = min(cut, (int)pb[f1 + nB - m]);
      wave_sync_lds();
      if (lane == 0) {
        atomicOr(&sbits[cut >> 5], 1u << (cut & 31));
        int s2 = sp;
        if (cut - first > 16) { stF[s2] = first; stL[s2] = cut; stD[s2] = depth; s2++; }
        if (last - cut > 16) { stF[s2] = cut; stL[s2] = last; stD[s2] = depth; s2++; }
      }
      sp += (cut - first > 16) + (last - cut > 16);
      wave_sync_lds();
    }
    // the final insertion sort: a lane per block between two marks
    for (int p = lane; p < n; p += 64) {
      if ((sbits[p >> 5] >> (p & 31)) & 1u) {
        int q = p + 1;
        while (q < n && !((sbits[q >> 5] >> (q & 31)) & 1u)) q++;
        for (int i = p + 1; i < q; ++i) {
          const uint32_t val = v[i];
          int j = i;
          while (j > p && T_(val) < T_(v[j - 1])) { v[j] = v[j - 1]; --j; }
          v[j] = val;
        }
      }
    }
    wave_sync_lds();
    // RemoveFrequent: a run of maxFreq or more equal keys goes
    {
      int carry = 0;
      for (int base = 0; base < n; base += 64) {                           // run start of every position
        const int p = base + lane;
        const bool st = p < n && (p == 0 || T_(v[p]) != T_(v[p - 1]));
        const unsigned long long ms = __ballot(st);
        const unsigned long long upto = ms & (below | (1ULL << lane));
        if (p < n) pa[p] = (unsigned short)(upto ? base + 63 - __clzll(upto) : carry);
        if (ms) carry = base + 63 - __clzll(ms);
      }
      int nxt = n;
      for (int base = ((n - 1) / 64) * 64; base >= 0; base -= 64) {        // run end (the next start) of every position
        const int p = base + lane;
        const bool st = p < n && (p == 0 || T_(v[p]) != T_(v[p - 1]));
        const unsigned long long ms = __ballot(st);
        const unsigned long long above = lane == 63 ? 0ULL : (ms & (~0ULL << (lane + 1)));
        if (p < n) pb[p] = (unsigned short)(above ? base + __ffsll((long long)above) - 1 : nxt);
        if (ms) nxt = base + __ffsll((long long)ms) - 1;
      }
      wave_sync_lds();
      int kept = 0;
      for (int base = 0; base < n; base += 64) {
        const int p = base + lane;
        const bool keep = p < n && (int)pb[p] - (int)pa[p] < maxFreq;
        const unsigned long long mk = __ballot(keep);
        if (keep) g[kept + __popcll(mk & below)] = v[p];
        kept += __popcll(mk);
      }
      if (lane == 0) counts[wi] = (uint32_t)kept;
    }
    wave_sync_lds();
  }
}

__global__ void local_sel(uint64_t n_win, const uint32_t* __restrict__ flag, const uint64_t* __restrict__ off, uint32_t* __restrict__ sel) {
  const uint64_t wi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (wi < n_win && flag[wi]) sel[off[wi]] = (uint32_t)wi;
}

// the surviving tuples of every window, packed: a wave takes 64 windows, one after the other, its lanes side by side (coalesced both ways)
__global__ void __launch_bounds__(64) local_compact(uint64_t n_win, uint64_t stride, const uint32_t* __restrict__ raw,
                                                    const uint64_t* __restrict__ bnd, uint32_t* __restrict__ out) {
  const int lane = threadIdx.x;
  const uint64_t w0 = (uint64_t)blockIdx.x * 64;
  const uint64_t mine = w0 + lane < n_win ? bnd[w0 + lane] : 0, mineEnd = w0 + lane < n_win ? bnd[w0 + lane + 1] : 0;
  for (int x = 0; x < 64 && w0 + x < n_win; x++) {
    const uint64_t d0 = __shfl(mine, x), n = __shfl(mineEnd, x) - d0;
    const uint32_t* s = raw + (w0 + x) * stride;
    for (uint64_t i = lane; i < n; i += 64) out[d0 + i] = s[i];
  }
}

// ---- CompareLists<LocalTuple,SmallTuple>, one lane per task
struct CmpArgs {
  uint64_t n_tasks;
  const uint32_t* q; const uint64_t* q_lo; const uint64_t* q_hi;
  const uint32_t* t; const uint64_t* t_lo; const uint64_t* t_hi;
  long maxFreq; const int64_t* maxDiag; const int64_t* minDiag;
  const uint64_t* out_off; uint32_t* out_qi; uint32_t* out_ti; uint32_t* counts;
  uint16_t* capped; int* nOver;                 // MODE 2: the pairs of task x as (qi << 8 | ti) at capped[x * CMP_CAP ..], the tasks that do not fit counted
};
constexpr int CMP_CAP = 128;                    // pairs per task kept by the one-pass form (a task of two ~30-tuple lists yields ~20; the headline batch's largest 104)

// A block's 64 tasks share CMP_WORDS words of LDS, each task's query list then its target list packed behind the previous task's (a task of more than CMP_TASK_MAX words,
// or one that no longer fits, is walked in HBM): a typical task has ~180 words, so 40 KB hold a block and four blocks fit a CU (fixed 257-word rows: two).
constexpr int CMP_WORDS = 10240, CMP_TASK_MAX = 512;
constexpr int CW_ROW = 512;                     // pairs per task kept by the one-walk form for large tasks (packed qi << 16 | ti; a task with more is walked again)
// MODE 0: count; 1: write the pairs at out_off (after a scan of the counts: the walk runs twice); 2: count AND keep the pairs, packed, in a fixed row per task --
// local_compact_pairs then lays them out by the scan of the counts.  A task with a list of more than 255 tuples or more than CMP_CAP pairs sends the batch
// through modes 0 + 1 (nOver).
// BIG: the batch's tasks are beyond the staging (windows of 2048 bases -- the .gli file `lra index` writes -- hold ~700 tuples a list): a lane per task walks its lists
// where they lie, every lane of the block (the 16-lanes-per-wave form exists for the LDS rows).  Its two searches start where the walk stands: lower_bound from ts
// upwards and upper_bound from te downwards by doubling steps, then by halving inside the bracket -- the bound of a sorted range does not depend on the probes that
// find it, and the two lists interleave, so it is one or two elements away (the literal halving of [ts, te) is ten scattered probes).
template <int MODE, bool BIG = false>
__global__ void __launch_bounds__(STAGE_NT) local_compare(CmpArgs A, uint32_t* __restrict__ rows = nullptr) {
  constexpr bool EMIT = MODE == 1;
  __shared__ uint32_t stage[BIG ? 1 : CMP_WORDS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t x0 = (uint64_t)blockIdx.x * (BIG ? STAGE_NT : 64);
  const uint64_t xl = x0 + (BIG ? threadIdx.x : lane);
  const bool live = xl < A.n_tasks;
  const uint64_t myQa = live ? A.q_lo[xl] : 0, myTa = live ? A.t_lo[xl] : 0;
  const uint64_t myQn = live ? A.q_hi[xl] - myQa : 0, myTn = live ? A.t_hi[xl] - myTa : 0;
  uint64_t x; long nq, nt; const uint32_t* q; const uint32_t* t;
  if constexpr (BIG) {
    if (!live) return;
    if (MODE == 1 && rows && A.counts[xl] <= (uint32_t)CW_ROW) return;   // (its pairs came out of its row: local_compact_rows)
    x = xl; nq = (long)myQn; nt = (long)myTn; q = A.q + myQa; t = A.t + myTa;
  } else {
  // where task `lane`'s lists sit in the block's LDS (every wave computes the same prefix)
  const uint32_t need = (myQn + myTn <= (uint64_t)CMP_TASK_MAX) ? (uint32_t)(myQn + myTn) : 0u;
  uint32_t incl = need;
  for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= d) incl += o; }
  const uint32_t myOff = incl - need;
  const bool myFits = need > 0 && incl <= (uint32_t)CMP_WORDS;
#pragma unroll 4
  for (int i = wave; i < 64; i += STAGE_NT / 64) {                       // coalesced staging, one task per wave at a time
    const uint64_t qa = __shfl(myQa, i), qn = __shfl(myQn, i), ta = __shfl(myTa, i), tn = __shfl(myTn, i);
    const uint32_t so = __shfl(myOff, i);
    if (!__shfl((int)myFits, i)) continue;
    for (uint32_t p = lane; p < qn; p += 64) stage[so + p] = A.q[qa + p];
    for (uint32_t p = lane; p < tn; p += 64) stage[so + (uint32_t)qn + p] = A.t[ta + p];
  }
  __syncthreads();
  // the walks diverge from task to task: every wave of the block takes 16 of its 64 tasks (16 active lanes), so the four SIMDs of the CU
  // work on the block's lists at once and a wave only waits for the longest of 16 walks
  constexpr int TPW = 64 / (STAGE_NT / 64);
  const int slot = wave * TPW + lane;                                    // the task (and its LDS row) this lane walks
  const uint64_t sQa = __shfl(myQa, slot & 63), sTa = __shfl(myTa, slot & 63), sQn = __shfl(myQn, slot & 63), sTn = __shfl(myTn, slot & 63);
  const uint32_t sOff = __shfl(myOff, slot & 63); const bool staged = __shfl((int)myFits, slot & 63) != 0;
  if (lane >= TPW || x0 + slot >= A.n_tasks) return;
  x = x0 + slot;
  nq = (long)sQn; nt = (long)sTn;
  // (one walk through pointers that are LDS or HBM per lane -- flat loads: a block's last tasks often do not fit its LDS, and a walk per address space makes every
  // wave that holds one of them run both, one after the other: measured 44 -> 60 ms)
  q = staged ? stage + sOff : A.q + sQa;
  t = staged ? stage + sOff + (uint32_t)nq : A.t + sTa;
  }
  const int64_t maxDiag = A.maxDiag ? A.maxDiag[x] : 0, minDiag = A.minDiag ? A.minDiag[x] : 0;
  const long maxFreq = A.maxFreq;
  uint32_t* oq = EMIT ? A.out_qi + A.out_off[x] : nullptr; uint32_t* ot = EMIT ? A.out_ti + A.out_off[x] : nullptr;
  uint16_t* cp = (MODE == 2 && !BIG) ? A.capped + x * (uint64_t)CMP_CAP : nullptr;
  uint32_t* row = (MODE == 2 && BIG) ? rows + x * (uint64_t)CW_ROW : nullptr;
  uint32_t n = 0;
  auto emit = [&](long qi, long ti) {                                    // :87-97
    if (maxDiag != 0 && minDiag != 0) {
      const int64_t d = (int64_t)P_(t[ti]) - (int64_t)P_(q[qi]);
      if (!(d <= maxDiag && d >= minDiag)) return;
    }
    if (EMIT) { oq[n] = (uint32_t)(A.q_lo[x] + qi); ot[n] = (uint32_t)(A.t_lo[x] + ti); }
    if (MODE == 2 && !BIG && n < (uint32_t)CMP_CAP) cp[n] = (uint16_t)((qi << 8) | ti);
    if (MODE == 2 && BIG && n < (uint32_t)CW_ROW) row[n] = ((uint32_t)qi << 16) | (uint32_t)ti;
    n++;
  };
#define Q(i) T_(q[(i)])
#define TT(i) T_(t[(i)])
  if (nq != 0 && nt != 0) {
    long qs = 0, qe = nq - 1, ts = 0, te = nt;
    do {
      while (qs <= qe && Q(qs) < TT(ts)) qs++;
      if (qs >= qe) break;
      const uint32_t startGap = (Q(qs) - TT(ts)) & TMASK;
      while (qe > qs && te > ts && Q(qe) > TT(te - 1)) qe--;
      const uint32_t endGap = (TT(te - 1) - Q(qe)) & TMASK;
      if (startGap == 0 || startGap > endGap) {
        const long tsOrig = ts, qsOrig = qs;
        long lo = ts, hi = te;
        if (BIG) { const uint32_t key = Q(qs); for (long s_ = 1; lo < hi; s_ <<= 1) { const long p = lo + s_ - 1; if (p >= hi) break; if (TT(p) < key) lo = p + 1; else { hi = p; break; } } }
        while (lo < hi) { long mid = lo + (hi - lo) / 2; if (TT(mid) < Q(qs)) lo = mid + 1; else hi = mid; }
        ts = lo;
        if (ts < te && TT(ts) == Q(qs)) {
          long tsi = ts;
          while (tsi != te && Q(qs) == TT(tsi)) tsi++;
          const long qsStart = qs;
          while (qs < qe && Q(qs + 1) == Q(qs)) qs++;
          if (qs - qsStart < maxFreq)
            for (long ti = ts; ti != tsi; ti++)
              for (long qi = qsStart; qi <= qs; qi++) emit(qi, ti);
        }
        { const uint32_t raw = TT(tsOrig); while (ts < te && TT(ts) == raw) ts++; }
        { const uint32_t raw = Q(qsOrig); while (qs < qe && Q(qs) == raw) qs++; }
      } else {
        if (te != nt && TT(te - 1) == Q(qe)) {
        } else {
          long lo = ts, hi = te;
          if (BIG) { const uint32_t key = Q(qe); for (long s_ = 1; lo < hi; s_ <<= 1) { const long p = hi - s_; if (p < lo) break; if (!(key < TT(p))) { lo = p + 1; break; } else hi = p; } }
          while (lo < hi) { long mid = lo + (hi - lo) / 2; if (!(Q(qe) < TT(mid))) lo = mid + 1; else hi = mid; }
          te = lo;
        }
        const long teStart = te;
        long tei = te;
        while (tei > ts && TT(tei - 1) == Q(qe)) tei--;
        if (tei < teStart && teStart > 0) {
          const long qeStart = qe;
          while (qe > qs && Q(qe) == Q(qe - 1)) qe--;
          if (qeStart - qe < maxFreq)
            for (long ti = tei; ti < teStart; ti++)
              for (long qi = qe; qi <= qeStart; qi++) emit(qi, ti);
        }
        te = tei;
      }
    } while (qs < qe && ts < te);
  }
#undef Q
#undef TT
  if (!EMIT) A.counts[x] = n;
  if (MODE == 2 && !BIG && (n > (uint32_t)CMP_CAP || nq > 255 || nt > 255)) atomicAdd(A.nOver, 1);
  if (MODE == 2 && BIG && n > (uint32_t)CW_ROW) atomicAdd(A.nOver, 1);
}


// ---- CompareLists for batches of LARGE tasks (two lists of ~700 tuples: the 2048-base windows of a .gli file): the walk is a chain of dependent reads, one lane per
// task, and with the lists where they lie every step of a wave is 64 scattered lines (142 ms per batch: each line fetched for one word and gone again before the next).
// Here a wave takes CW_TPB tasks at a time: all its lanes copy the tasks' lists into LDS (every byte read once, coalesced), then CW_TPB lanes walk them -- the walk's
// round trips are LDS round trips.  One walk per task: its pairs go to a row of CW_ROW packed pairs (qi << 16 | ti), local_compact_rows lays them out by the scan of
// the counts; a task with more pairs is walked again, alone with its likes, writing at its offset (MODE 1).
constexpr int CW_TPB = 4, CW_SLOT = 1536;
template <int MODE>
__global__ void __launch_bounds__(64) local_compare_wave(CmpArgs A, uint32_t* __restrict__ rows) {
  __shared__ uint32_t stage[CW_TPB * CW_SLOT];
  const int lane = threadIdx.x;
  for (uint64_t g0 = (uint64_t)blockIdx.x * CW_TPB; g0 < A.n_tasks; g0 += (uint64_t)gridDim.x * CW_TPB) {
    const uint64_t xl = g0 + (uint64_t)(lane & (CW_TPB - 1));             // lane i (and i + CW_TPB, ..) looks at task g0 + i
    bool live = xl < A.n_tasks;
    if (MODE == 1 && live) live = A.counts[xl] > (uint32_t)CW_ROW;
    const uint64_t myQa = live ? A.q_lo[xl] : 0, myTa = live ? A.t_lo[xl] : 0;
    const int myQn = live ? (int)(A.q_hi[xl] - myQa) : 0, myTn = live ? (int)(A.t_hi[xl] - myTa) : 0;
    const bool myFits = myQn + myTn <= CW_SLOT;
#pragma unroll
    for (int i = 0; i < CW_TPB; i++) {
      const uint64_t qa = __shfl(myQa, i), ta = __shfl(myTa, i);
      const int qn = __shfl(myQn, i), tn = __shfl(myTn, i);
      if (!__shfl((int)myFits, i)) continue;
      uint32_t* d = stage + i * CW_SLOT;
      for (int p = lane; p < qn; p += 64) d[p] = A.q[qa + p];
      for (int p = lane; p < tn; p += 64) d[qn + p] = A.t[ta + p];
    }
    __syncthreads();
    if (lane < CW_TPB && live) {
      const uint64_t x = xl;
      const int nq = myQn, nt = myTn;
      const uint32_t* q = myFits ? stage + lane * CW_SLOT : A.q + myQa;
      const uint32_t* t = myFits ? stage + lane * CW_SLOT + nq : A.t + myTa;
      const int64_t maxDiag = A.maxDiag ? A.maxDiag[x] : 0, minDiag = A.minDiag ? A.minDiag[x] : 0;
      const bool banded = maxDiag != 0 && minDiag != 0;
      const int maxFreq = (int)A.maxFreq;
      uint32_t* oq = MODE == 1 ? A.out_qi + A.out_off[x] : nullptr; uint32_t* ot = MODE == 1 ? A.out_ti + A.out_off[x] : nullptr;
      uint32_t* row = rows + x * (uint64_t)CW_ROW;
      const uint32_t qb = (uint32_t)myQa, tb = (uint32_t)myTa;
      uint32_t n = 0;
      auto emit = [&](int qi, int ti) {                                    // :87-97
        if (banded) {
          const int64_t d = (int64_t)P_(t[ti]) - (int64_t)P_(q[qi]);
          if (!(d <= maxDiag && d >= minDiag)) return;
        }
        if (MODE == 1) { oq[n] = qb + (uint32_t)qi; ot[n] = tb + (uint32_t)ti; }
        else if (n < (uint32_t)CW_ROW) row[n] = ((uint32_t)qi << 16) | (uint32_t)ti;
        n++;
      };
#define Q(i) T_(q[(i)])
#define TT(i) T_(t[(i)])
      if (nq != 0 && nt != 0) {
        int qs = 0, qe = nq - 1, ts = 0, te = nt;
        do {
          { const uint32_t k0 = TT(ts); while (qs <= qe && Q(qs) < k0) qs++; }
          if (qs >= qe) break;
          const uint32_t kq = Q(qs);
          const uint32_t startGap = (kq - TT(ts)) & TMASK;
          { const uint32_t k1 = TT(te - 1); while (qe > qs && te > ts && Q(qe) > k1) qe--; }
          const uint32_t ke = Q(qe);
          const uint32_t endGap = (TT(te - 1) - ke) & TMASK;
          if (startGap == 0 || startGap > endGap) {
            const int tsOrig = ts;
            int lo = ts, hi = te;                                              // lower_bound(T[ts, te), Q[qs]) from ts upwards (see local_compare<.., BIG>)
            for (int s_ = 1; lo < hi; s_ <<= 1) { const int p = lo + s_ - 1; if (p >= hi) break; if (TT(p) < kq) lo = p + 1; else { hi = p; break; } }
            while (lo < hi) { const int mid = lo + (hi - lo) / 2; if (TT(mid) < kq) lo = mid + 1; else hi = mid; }
            ts = lo;
            if (ts < te && TT(ts) == kq) {
              int tsi = ts;
              while (tsi != te && kq == TT(tsi)) tsi++;
              const int qsStart = qs;
              while (qs < qe && Q(qs + 1) == kq) qs++;
              if (qs - qsStart < maxFreq)
                for (int ti = ts; ti != tsi; ti++)
                  for (int qi = qsStart; qi <= qs; qi++) emit(qi, ti);
            }
            { const uint32_t raw = TT(tsOrig); while (ts < te && TT(ts) == raw) ts++; }
            while (qs < qe && Q(qs) == kq) qs++;
          } else {
            if (te != nt && TT(te - 1) == ke) {
            } else {
              int lo = ts, hi = te;                                            // upper_bound(T[ts, te), Q[qe]) from te downwards
              for (int s_ = 1; lo < hi; s_ <<= 1) { const int p = hi - s_; if (p < lo) break; if (!(ke < TT(p))) { lo = p + 1; break; } else hi = p; }
              while (lo < hi) { const int mid = lo + (hi - lo) / 2; if (!(ke < TT(mid))) lo = mid + 1; else hi = mid; }
              te = lo;
            }
            const int teStart = te;
            int tei = te;
            while (tei > ts && TT(tei - 1) == ke) tei--;
            if (tei < teStart && teStart > 0) {
              const int qeStart = qe;
              while (qe > qs && Q(qe - 1) == ke) qe--;
              if (qeStart - qe < maxFreq)
                for (int ti = tei; ti < teStart; ti++)
                  for (int qi = qe; qi <= qeStart; qi++) emit(qi, ti);
            }
            te = tei;
          }
        } while (qs < qe && ts < te);
      }
#undef Q
#undef TT
      if (MODE != 1) { A.counts[x] = n; if (n > (uint32_t)CW_ROW) atomicAdd(A.nOver, 1); }
    }
    __syncthreads();
  }
}
// the rows' pairs to their places: a wave per task
__global__ void __launch_bounds__(256) local_compact_rows(CmpArgs A, const uint32_t* __restrict__ rows) {
  const uint64_t x = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (x >= A.n_tasks) return;
  const int l = threadIdx.x & 63;
  const uint32_t n = A.counts[x];
  if (n > (uint32_t)CW_ROW) return;                                        // (walked again: local_compare_wave<1>)
  const uint64_t o = A.out_off[x];
  const uint32_t ql = (uint32_t)A.q_lo[x], tl = (uint32_t)A.t_lo[x];
  const uint32_t* rw = rows + x * (uint64_t)CW_ROW;
  for (uint32_t p = l; p < n; p += 64) { const uint32_t v = rw[p]; A.out_qi[o + p] = ql + (v >> 16); A.out_ti[o + p] = tl + (v & 0xFFFFu); }
}


// ---- CompareLists of large tasks, a lane per task with the four cursors' lines kept in LDS.  The walk reads its lists at four places that only move one way (qs up, qe
// down, ts up, te down); read where they lie, each read is a round trip to L2 / HBM for one word of a line that is gone before the cursor needs the next one.  Here a lane
// owns two lines of CL words per cursor in LDS (word-interleaved over the wave's lanes: bank = lane): an access is an LDS read, a cursor that crosses a line boundary
// fetches the next line whole (two 16-byte loads), so every word of a list is fetched about once.  Probes of a search that land far from where the walk stands read the
// list directly.
constexpr int CL = 8;
struct Cursor {
  const uint32_t* arr; uint64_t off; uint32_t* row; int tag0, tag1;
  __device__ __forceinline__ void init(const uint32_t* a, uint64_t o, uint32_t* r) { arr = a; off = o; row = r; tag0 = -1; tag1 = -1; }
  __device__ __forceinline__ uint32_t get(int i) {
    const uint64_t g = off + (uint64_t)(uint32_t)i;
    const int line = (int)(g >> 3), sl = line & 1;
    if ((sl ? tag1 : tag0) != line) {                                      // (fetching the line the cursor comes to next in the same round trip was measured: slower, 30.9 -> 34.6 ms)
      const uint4* src = (const uint4*)(arr + ((uint64_t)(uint32_t)line << 3));
      const uint4 a = src[0], b = src[1];
      uint32_t* d = row + sl * CL * 64;
      d[0] = a.x; d[64] = a.y; d[128] = a.z; d[192] = a.w; d[256] = b.x; d[320] = b.y; d[384] = b.z; d[448] = b.w;
      if (sl) tag1 = line; else tag0 = line;
    }
    return row[(sl * CL + (int)(g & 7)) * 64];
  }
  __device__ __forceinline__ uint32_t direct(int i) const { return arr[off + (uint64_t)(uint32_t)i]; }
};
template <int MODE>
__global__ void __launch_bounds__(64) local_compare_cached(CmpArgs A, uint32_t* __restrict__ rows, const uint32_t* __restrict__ overList, int nOverList) {
  __shared__ uint32_t lines[4 * 2 * CL * 64];
  const int lane = threadIdx.x;
  // MODE 1: the tasks whose pairs outgrew their row, 64 of them to a wave (taken from a list: a wave of the first pass's order would walk for its one or two such tasks)
  const uint64_t x0 = (uint64_t)blockIdx.x * 64 + lane;
  if (MODE == 1 ? x0 >= (uint64_t)nOverList : x0 >= A.n_tasks) return;
  const uint64_t x = MODE == 1 ? (uint64_t)overList[x0] : x0;
  const uint64_t myQa = A.q_lo[x], myTa = A.t_lo[x];
  const int nq = (int)(A.q_hi[x] - myQa), nt = (int)(A.t_hi[x] - myTa);
  Cursor QF, QB, TF, TB;
  QF.init(A.q, myQa, lines + 0 * 2 * CL * 64 + lane); QB.init(A.q, myQa, lines + 1 * 2 * CL * 64 + lane);
  TF.init(A.t, myTa, lines + 2 * 2 * CL * 64 + lane); TB.init(A.t, myTa, lines + 3 * 2 * CL * 64 + lane);
  const int64_t maxDiag = A.maxDiag ? A.maxDiag[x] : 0, minDiag = A.minDiag ? A.minDiag[x] : 0;
  const bool banded = maxDiag != 0 && minDiag != 0;
  const int maxFreq = (int)A.maxFreq;
  uint32_t* oq = MODE == 1 ? A.out_qi + A.out_off[x] : nullptr; uint32_t* ot = MODE == 1 ? A.out_ti + A.out_off[x] : nullptr;
  uint32_t* row = MODE == 2 ? rows + x * (uint64_t)CW_ROW : nullptr;
  const uint32_t qb = (uint32_t)myQa, tb = (uint32_t)myTa;
  uint32_t n = 0;
  auto put = [&](int qi, int ti) {
    if (MODE == 1) { oq[n] = qb + (uint32_t)qi; ot[n] = tb + (uint32_t)ti; }
    else if (n < (uint32_t)CW_ROW) row[n] = ((uint32_t)qi << 16) | (uint32_t)ti;
    n++;
  };
  if (nq != 0 && nt != 0) {
    int qs = 0, qe = nq - 1, ts = 0, te = nt;
    do {
      { const uint32_t k0 = T_(TF.get(ts)); while (qs <= qe && T_(QF.get(qs)) < k0) qs++; }
      if (qs >= qe) break;
      const uint32_t kq = T_(QF.get(qs));
      const uint32_t startGap = (kq - T_(TF.get(ts))) & TMASK;
      { const uint32_t k1 = T_(TB.get(te - 1)); while (qe > qs && te > ts && T_(QB.get(qe)) > k1) qe--; }
      const uint32_t ke = T_(QB.get(qe));
      const uint32_t endGap = (T_(TB.get(te - 1)) - ke) & TMASK;
      if (startGap == 0 || startGap > endGap) {
        const int tsOrig = ts;
        auto tf = [&](int p) -> uint32_t { return T_((unsigned)(p - tsOrig) < 12u ? TF.get(p) : TF.direct(p)); };
        int lo = ts, hi = te;
        for (int s_ = 1; lo < hi; s_ <<= 1) { const int p = lo + s_ - 1; if (p >= hi) break; if (tf(p) < kq) lo = p + 1; else { hi = p; break; } }
        while (lo < hi) { const int mid = lo + (hi - lo) / 2; if (tf(mid) < kq) lo = mid + 1; else hi = mid; }
        ts = lo;
        if (ts < te && tf(ts) == kq) {
          int tsi = ts;
          while (tsi != te && kq == tf(tsi)) tsi++;
          const int qsStart = qs;
          while (qs < qe && T_(QF.get(qs + 1)) == kq) qs++;
          if (qs - qsStart < maxFreq)
            for (int ti = ts; ti != tsi; ti++) {
              const int64_t tp = banded ? (int64_t)P_((unsigned)(ti - tsOrig) < 12u ? TF.get(ti) : TF.direct(ti)) : 0;
              for (int qi = qsStart; qi <= qs; qi++) {
                if (banded) { const int64_t d = tp - (int64_t)P_(QF.get(qi)); if (!(d <= maxDiag && d >= minDiag)) continue; }
                put(qi, ti);
              }
            }
        }
        { const uint32_t raw = T_(TF.get(tsOrig)); while (ts < te && tf(ts) == raw) ts++; }
        while (qs < qe && T_(QF.get(qs)) == kq) qs++;
      } else {
        const int teOrig = te;
        auto tbk = [&](int p) -> uint32_t { return T_((unsigned)(teOrig - 1 - p) < 12u ? TB.get(p) : TB.direct(p)); };
        if (te != nt && tbk(te - 1) == ke) {
        } else {
          int lo = ts, hi = te;
          for (int s_ = 1; lo < hi; s_ <<= 1) { const int p = hi - s_; if (p < lo) break; if (!(ke < tbk(p))) { lo = p + 1; break; } else hi = p; }
          while (lo < hi) { const int mid = lo + (hi - lo) / 2; if (!(ke < tbk(mid))) lo = mid + 1; else hi = mid; }
          te = lo;
        }
        const int teStart = te;
        int tei = te;
        while (tei > ts && tbk(tei - 1) == ke) tei--;
        if (tei < teStart && teStart > 0) {
          const int qeStart = qe;
          while (qe > qs && T_(QB.get(qe - 1)) == ke) qe--;
          if (qeStart - qe < maxFreq)
            for (int ti = tei; ti < teStart; ti++) {
              const int64_t tp = banded ? (int64_t)P_((unsigned)(teOrig - 1 - ti) < 12u ? TB.get(ti) : TB.direct(ti)) : 0;
              for (int qi = qe; qi <= qeStart; qi++) {
                if (banded) { const int64_t d = tp - (int64_t)P_(QB.get(qi)); if (!(d <= maxDiag && d >= minDiag)) continue; }
                put(qi, ti);
              }
            }
        }
        te = tei;
      }
    } while (qs < qe && ts < te);
  }
  if (MODE != 1) { A.counts[x] = n; if (n > (uint32_t)CW_ROW) { const int at = atomicAdd(A.nOver, 1); if (overList && at < nOverList) ((uint32_t*)overList)[at] = (uint32_t)x; } }
}

// the tasks whose two lists are beyond a staged row (the batch's form of the walk is chosen by their share)
__global__ void k_big_tasks(CmpArgs A, unsigned long long* nBig) {
  const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool big = x < A.n_tasks && (A.q_hi[x] - A.q_lo[x]) + (A.t_hi[x] - A.t_lo[x]) > (uint64_t)CMP_TASK_MAX;
  const unsigned long long m = __ballot(big);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(nBig, (unsigned long long)__popcll(m));
}

// the kept pairs of 4 tasks per wave (16 lanes each) to their places
__global__ void __launch_bounds__(64) local_compact_pairs(CmpArgs A) {
  const uint64_t x = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 4);
  if (x >= A.n_tasks) return;
  const int l = threadIdx.x & 15;
  const uint32_t n = A.counts[x];
  const uint64_t o = A.out_off[x], ql = A.q_lo[x], tl = A.t_lo[x];
  const uint16_t* cp = A.capped + x * (uint64_t)CMP_CAP;
  for (uint32_t p = l; p < n; p += 16) { const uint32_t v = cp[p]; A.out_qi[o + p] = (uint32_t)(ql + (v >> 8)); A.out_ti[o + p] = (uint32_t)(tl + (v & 255u)); }
}

template <typename T>
static T* carve(char*& p, size_t n) { T* r = (T*)p; p += (n * sizeof(T) + 255) & ~(size_t)255; return r; }
static int d2h8(lra_ctx* ctx, uint64_t* dst, const uint64_t* src) {
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, 8, hipMemcpyDeviceToHost, ctx->stream));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return LRA_OK;
}

}  // namespace

extern "C" int lra_local_index_batch(lra_ctx* ctx, int n_seqs, const char* d_seq, const uint64_t* d_seq_off, int k, int w, int window,
                                     int max_freq, lra_local_index_result* out) {
  return lra_local_index_masked_batch(ctx, n_seqs, d_seq, d_seq_off, nullptr, k, w, window, max_freq, out);
}

extern "C" int lra_local_index_masked_batch(lra_ctx* ctx, int n_seqs, const char* d_seq, const uint64_t* d_seq_off, const uint8_t* d_active, int k, int w,
                                            int window, int max_freq, lra_local_index_result* out) {
  if (!ctx || !out || n_seqs < 0) return LRA_ERR_INVALID;
  if (k < 1 || k > 10 || w < 1 || w > MAXW || window < w + k || window > 4096)
    return lra_set_err(ctx, LRA_ERR_INVALID, "need 1<=k<=10 (20-bit LocalTuple), 1<=w<=%d, w+k<=window<=4096", MAXW);
  memset(out, 0, sizeof(*out));
  if (n_seqs == 0) return LRA_OK;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  auto sz = [](size_t n, size_t e) { return (n * e + 255) & ~(size_t)255; };
  // pass 0: windows per sequence
  char* w0 = (char*)lra_scratch(ctx, 0, sz((size_t)n_seqs + 1, 4) + sz((size_t)n_seqs + 1, 8) + 4096);
  if (!w0) return LRA_ERR_NOMEM;
  uint32_t* nwin = carve<uint32_t>(w0, (size_t)n_seqs + 1);
  uint64_t* win_off = carve<uint64_t>(w0, (size_t)n_seqs + 1);
  hipLaunchKernelGGL(window_count, dim3((n_seqs + 255) / 256), dim3(256), 0, st, n_seqs, d_seq_off, window, nwin);
  if (lra_exclusive_scan<uint32_t>(ctx, (long)n_seqs, nwin, win_off)) return LRA_ERR_HIP;
  uint64_t n_win = 0;
  if (d2h8(ctx, &n_win, win_off + n_seqs)) return LRA_ERR_HIP;
  // per-window arrays
  const size_t NW = (size_t)n_win + 2;
  char* w1 = (char*)lra_scratch(ctx, 1, sz(NW, 4) * 3 + sz(NW, 8) * 3 + 4096);
  if (!w1) return LRA_ERR_NOMEM;
  uint32_t* w_seq = carve<uint32_t>(w1, NW); uint32_t* w_len = carve<uint32_t>(w1, NW); uint32_t* cnt = carve<uint32_t>(w1, NW);
  uint64_t* w_start = carve<uint64_t>(w1, NW); uint64_t* spare_ = carve<uint64_t>(w1, NW); (void)spare_; uint64_t* bnd_tmp = carve<uint64_t>(w1, NW);
  const unsigned char* seq = (const unsigned char*)d_seq;
  const unsigned gw = (unsigned)((n_win + 63) / 64);
  uint64_t n_tup = 0;
  const uint64_t stride = (uint64_t)(window - k + 1);
  uint32_t* raw = (uint32_t*)lra_scratch(ctx, 2, ((size_t)n_win * stride + 64) * 4);
  if (!raw) return LRA_ERR_NOMEM;
  if (n_win) {
    hipLaunchKernelGGL(window_map, dim3((n_seqs + 255) / 256), dim3(256), 0, st, n_seqs, d_seq_off, window, win_off, d_active, w_seq, w_start, w_len);
    lra_time_begin(ctx, "local_sketch");
    hipLaunchKernelGGL(local_sketch, dim3(gw), dim3(64), (size_t)((WMAX / 16 + WMAX / 32 + w) * 64 * 4), st, n_win, seq, w_start, w_len, k, w, stride, raw, cnt);
    lra_time_end(ctx);
    static const bool exactOnly = getenv("LRA_LOCAL_EXACT_SORT") != nullptr;   // every list through the exact sort (kept for comparison)
    lra_time_begin(ctx, "local_sort_filter");
    if (exactOnly || max_freq < 2 || n_win >= (1ULL << 32)) hipLaunchKernelGGL(local_sort_filter, dim3(gw), dim3(STAGE_NT), 0, st, n_win, stride, raw, max_freq, cnt, (const uint32_t*)nullptr, (const uint64_t*)nullptr);
    else {
      char* ws = (char*)lra_ensure(ctx, 96, sz(NW, 4) * 2 + sz(NW, 8) + 1024);
      if (!ws) return LRA_ERR_NOMEM;
      uint32_t* flag = carve<uint32_t>(ws, NW); uint32_t* sel = carve<uint32_t>(ws, NW); uint64_t* foff = carve<uint64_t>(ws, NW);
      hipLaunchKernelGGL(local_sort_unique, dim3((unsigned)((n_win + 3) / 4)), dim3(256), 0, st, n_win, stride, raw, (const uint32_t*)cnt, flag);
      if (stride > (uint64_t)RANK_CAP) {                                  // lists beyond the rank sort: LDS radix sorts by size class (a window has at most 4096 - k + 1 tuples)
        const unsigned cu = (unsigned)ctx->num_cu;
        const unsigned gr = (unsigned)std::min<uint64_t>(n_win, (uint64_t)cu * 64);
        hipLaunchKernelGGL((local_sort_radix<256, 4>), dim3(gr), dim3(256), 0, st, n_win, stride, raw, (const uint32_t*)cnt, flag, RANK_CAP);
        if (stride > 1024) hipLaunchKernelGGL((local_sort_radix<256, 8>), dim3(gr), dim3(256), 0, st, n_win, stride, raw, (const uint32_t*)cnt, flag, 1024);
        if (stride > 2048) hipLaunchKernelGGL((local_sort_radix<512, 8>), dim3(gr), dim3(512), 0, st, n_win, stride, raw, (const uint32_t*)cnt, flag, 2048);
      }
      if (lra_exclusive_scan<uint32_t>(ctx, (long)n_win, flag, foff)) return LRA_ERR_HIP;
      hipLaunchKernelGGL(local_sel, dim3((unsigned)((n_win + 255) / 256)), dim3(256), 0, st, n_win, (const uint32_t*)flag, (const uint64_t*)foff, sel);
      static const bool laneExact = getenv("LRA_LOCAL_EXACT_LANES") != nullptr;   // (the one-lane-per-list exact sort for the long lists as well: kept for comparison)
      if (stride > (uint64_t)RANK_CAP && !laneExact) {
        const int capAll = (int)((stride + 63) & ~(uint64_t)63);
        const unsigned ge = (unsigned)std::min<uint64_t>(n_win, (uint64_t)ctx->num_cu * 32);
        int lo = 0;
        for (int cap : {1024, capAll}) {                                   // (a 2048-base window holds ~700 tuples: the 1024 class is nearly all of them, 8.4 KB of LDS a wave)
          cap = std::min(cap, capAll);
          if (cap <= lo) continue;
          const size_t lds = (size_t)cap * 8 + ((size_t)cap / 32 + 1) * 4 + 3 * 64 * 4;
          hipLaunchKernelGGL(local_sort_exact_wave, dim3(ge), dim3(64), lds, st, stride, raw, max_freq, cnt, (const uint32_t*)sel, (const uint64_t*)(foff + n_win), cap, lo);
          lo = cap;
        }
      } else
        hipLaunchKernelGGL(local_sort_filter, dim3(gw), dim3(STAGE_NT), 0, st, n_win, stride, raw, max_freq, cnt, (const uint32_t*)sel, (const uint64_t*)(foff + n_win));
    }
    lra_time_end(ctx);
    if (lra_exclusive_scan<uint32_t>(ctx, (long)n_win, cnt, bnd_tmp)) return LRA_ERR_HIP;
    if (d2h8(ctx, &n_tup, bnd_tmp + n_win)) return LRA_ERR_HIP;
  }
  // results: [win_off: n_seqs+1 u64][bnd: n_win+1 u64][tuples: n_tup u32] in a context-owned buffer
  const uint64_t need = sz((size_t)n_seqs + 1, 8) + sz(NW, 8) + sz((size_t)n_tup + 1, 4);
  out->n_seqs = n_seqs; out->n_windows = n_win; out->n_tuples = n_tup; out->bytes = need;
  char* ob = (char*)lra_ensure(ctx, 6, need + 256);
  if (!ob) return LRA_ERR_NOMEM;
  out->d_base = ob;
  uint64_t* o_win = carve<uint64_t>(ob, (size_t)n_seqs + 1); uint64_t* o_bnd = carve<uint64_t>(ob, NW); uint32_t* o_tup = carve<uint32_t>(ob, (size_t)n_tup + 1);
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(o_win, win_off, ((size_t)n_seqs + 1) * 8, hipMemcpyDeviceToDevice, st));
  if (n_win) {
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(o_bnd, bnd_tmp, ((size_t)n_win + 1) * 8, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(local_compact, dim3(gw), dim3(64), 0, st, n_win, stride, raw, bnd_tmp, o_tup);
  }
  LRA_HIP_CHECK(ctx, hipGetLastError());
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  out->d_win_off = o_win; out->d_tuple_bnd = o_bnd; out->d_tuples = o_tup;
  return LRA_OK;
}

extern "C" int lra_local_compare_batch(lra_ctx* ctx, uint64_t n_tasks, const uint32_t* d_q_tuples, const uint64_t* d_q_lo, const uint64_t* d_q_hi,
                                       const uint32_t* d_t_tuples, const uint64_t* d_t_lo, const uint64_t* d_t_hi, int max_freq,
                                       const int64_t* d_max_diag, const int64_t* d_min_diag, lra_local_pairs_result* out) {
  if (!ctx || !out) return LRA_ERR_INVALID;
  memset(out, 0, sizeof(*out));
  out->n_tasks = n_tasks;
  if (n_tasks == 0) return LRA_OK;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  auto sz = [](size_t n, size_t e) { return (n * e + 255) & ~(size_t)255; };
  char* w = (char*)lra_scratch(ctx, 0, sz(n_tasks + 1, 4) + sz(n_tasks + 1, 8) + 4096);
  if (!w) return LRA_ERR_NOMEM;
  CmpArgs A;
  A.n_tasks = n_tasks; A.q = d_q_tuples; A.q_lo = d_q_lo; A.q_hi = d_q_hi; A.t = d_t_tuples; A.t_lo = d_t_lo; A.t_hi = d_t_hi;
  A.maxFreq = max_freq; A.maxDiag = d_max_diag; A.minDiag = d_min_diag;
  A.counts = carve<uint32_t>(w, n_tasks + 1);
  uint64_t* off = carve<uint64_t>(w, n_tasks + 1);
  A.out_off = off; A.out_qi = nullptr; A.out_ti = nullptr;
  const unsigned g = (unsigned)((n_tasks + 63) / 64);
  static const bool twoPass = getenv("LRA_LOCAL_TWO_PASS") != nullptr;      // count, scan, walk again (kept for comparison; also what a batch with an oversized task falls back to)
  int* nOver = (int*)lra_ensure(ctx, 94, 64);
  if (!nOver) return LRA_ERR_NOMEM;
  A.nOver = nOver;
  LRA_HIP_CHECK(ctx, hipMemsetAsync(nOver, 0, 16, st));
  // the form of the walk: a batch whose tasks are mostly beyond the staged rows (the local index's windows are larger than 256 bases) is walked where its lists lie
  static const int forceBig = getenv("LRA_LOCAL_BIG") ? atoi(getenv("LRA_LOCAL_BIG")) : -1;
  uint64_t h_big = 0;
  hipLaunchKernelGGL(k_big_tasks, dim3((unsigned)((n_tasks + 255) / 256)), dim3(256), 0, st, A, (unsigned long long*)(nOver + 2));
  if (d2h8(ctx, &h_big, (const uint64_t*)(nOver + 2))) return LRA_ERR_HIP;
  const bool big = forceBig >= 0 ? forceBig != 0 : 2 * h_big > n_tasks;
  const unsigned gB = (unsigned)((n_tasks + STAGE_NT - 1) / STAGE_NT);
  static const bool waveForm = getenv("LRA_LOCAL_BIG_WAVE") != nullptr;     // (the large tasks' lists staged in LDS, four walks per wave: measured slower -- 188 ms against 142 --, kept for comparison)
  const bool waveBig = big && waveForm;
  A.capped = (twoPass || big) ? nullptr : (uint16_t*)lra_ensure(ctx, 95, (size_t)n_tasks * CMP_CAP * 2 + 256);
  if (!twoPass && !big && !A.capped) return LRA_ERR_NOMEM;
  constexpr int OVER_CAP = 1 << 20;
  uint32_t* rows = big ? (uint32_t*)lra_ensure(ctx, 95, (size_t)n_tasks * CW_ROW * 4 + (size_t)OVER_CAP * 4 + 512) : nullptr;
  if (big && !rows) return LRA_ERR_NOMEM;
  uint32_t* overList = big ? rows + (((size_t)n_tasks * CW_ROW + 63) & ~(size_t)63) : nullptr;
  const unsigned gW = (unsigned)std::min<uint64_t>((n_tasks + CW_TPB - 1) / CW_TPB, (uint64_t)ctx->num_cu * 24);
  // The lane-per-task walk of large tasks has four cursors a lane, a cache line each: with every wave slot of a CU taken (2048 walks) a line is fetched for one word and is
  // gone before the cursor needs its next word -- the launch moves 40 x its lists.  Dynamic LDS nobody uses keeps it to two blocks (8 waves, 512 walks) per CU, whose lines
  // stay in the CU's share of L2: 77.9 ms per batch -> 47.6 (four blocks: 55.8; one: 61.5).
  static const bool cachedBig = !(getenv("LRA_LOCAL_BIG_CACHED") && getenv("LRA_LOCAL_BIG_CACHED")[0] == '0');   // (0: the lists read where they lie)
  static const size_t bigPad = getenv("LRA_LOCAL_BIG_PAD") ? (size_t)atol(getenv("LRA_LOCAL_BIG_PAD")) : 64000;
  lra_time_begin(ctx, "local_compare");
  if (waveBig) hipLaunchKernelGGL(local_compare_wave<2>, dim3(gW), dim3(64), 0, st, A, rows);
  else if (big && cachedBig) hipLaunchKernelGGL(local_compare_cached<2>, dim3((unsigned)((n_tasks + 63) / 64)), dim3(64), 0, st, A, rows, (const uint32_t*)overList, OVER_CAP);
  else if (big) {
    if (bigPad > 65536) LRA_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)local_compare<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bigPad));
    hipLaunchKernelGGL((local_compare<2, true>), dim3(gB), dim3(STAGE_NT), bigPad, st, A, rows);
  }
  else if (twoPass) hipLaunchKernelGGL(local_compare<0>, dim3(g), dim3(STAGE_NT), 0, st, A);
  else hipLaunchKernelGGL(local_compare<2>, dim3(g), dim3(STAGE_NT), 0, st, A);
  lra_time_end(ctx);
  if (lra_exclusive_scan<uint32_t>(ctx, (long)n_tasks, A.counts, off)) return LRA_ERR_HIP;
  uint64_t total = 0; int h_over = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&h_over, nOver, 4, hipMemcpyDeviceToHost, st));
  if (d2h8(ctx, &total, off + n_tasks)) return LRA_ERR_HIP;
  if (getenv("LRA_LOCAL_DBG")) {
    std::vector<uint32_t> h(n_tasks); (void)hipMemcpy(h.data(), A.counts, n_tasks * 4, hipMemcpyDeviceToHost);
    long over[6] = {0, 0, 0, 0, 0, 0}; const uint32_t lim[6] = {16, 32, 48, 64, 96, 128}; uint32_t mx = 0;
    for (uint32_t v : h) { mx = std::max(mx, v); for (int i = 0; i < 6; i++) if (v > lim[i]) over[i]++; }
    fprintf(stderr, "[local_compare] %llu tasks, %llu pairs, max %u, %d oversized; tasks with more than 16/32/48/64/96/128 pairs: %ld %ld %ld %ld %ld %ld\n", (unsigned long long)n_tasks, (unsigned long long)total, mx, h_over, over[0], over[1], over[2], over[3], over[4], over[5]);
  }
  char* r = (char*)lra_scratch(ctx, 1, sz(total + 1, 4) * 2 + 4096);
  if (!r) return LRA_ERR_NOMEM;
  A.out_qi = carve<uint32_t>(r, total + 1); A.out_ti = carve<uint32_t>(r, total + 1);
  lra_time_begin(ctx, "local_compare");
  if (waveBig) {
    hipLaunchKernelGGL(local_compact_rows, dim3((unsigned)((n_tasks + 3) / 4)), dim3(256), 0, st, A, (const uint32_t*)rows);
    if (h_over > 0) hipLaunchKernelGGL(local_compare_wave<1>, dim3(gW), dim3(64), 0, st, A, rows);
  } else if (big) {
    hipLaunchKernelGGL(local_compact_rows, dim3((unsigned)((n_tasks + 3) / 4)), dim3(256), 0, st, A, (const uint32_t*)rows);
    if (h_over > 0 && cachedBig && h_over <= OVER_CAP) hipLaunchKernelGGL(local_compare_cached<1>, dim3((unsigned)((h_over + 63) / 64)), dim3(64), 0, st, A, rows, (const uint32_t*)overList, h_over);
    else if (h_over > 0 && cachedBig) { hipLaunchKernelGGL((local_compare<1, true>), dim3(gB), dim3(STAGE_NT), std::min<size_t>(bigPad, 65536), st, A, rows); }   // (more such tasks than the list holds: every task looked at)
    else if (h_over > 0) hipLaunchKernelGGL((local_compare<1, true>), dim3(gB), dim3(STAGE_NT), std::min<size_t>(bigPad, 65536), st, A, rows);
  }
  else if (twoPass || h_over > 0) hipLaunchKernelGGL(local_compare<1>, dim3(g), dim3(STAGE_NT), 0, st, A);
  else hipLaunchKernelGGL(local_compact_pairs, dim3((unsigned)((n_tasks + 3) / 4)), dim3(64), 0, st, A);
  lra_time_end(ctx);
  LRA_HIP_CHECK(ctx, hipGetLastError());
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  out->n_pairs = total; out->d_pair_off = off; out->d_pair_qi = A.out_qi; out->d_pair_ti = A.out_ti;
  return LRA_OK;
}
