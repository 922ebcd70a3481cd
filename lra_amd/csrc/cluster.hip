// lra_amd/csrc/cluster.hip -- a5: match cleaning and diagonal clusters for a read batch (gfx950).
//
// Replaces CleanMatches (reference: Clustering.h:1840-1906) in the configuration all presets use
// (opts.ExtractDiagonalFromClean): DiagonalSort / AntiDiagonalSort (Sorting.h:50,113),
// CleanOffDiagonal (Clustering.h:566-798), AVGfreq (:550), SecondRoundCleanOffDiagonal (:802-868),
// Cluster boundaries (:308) and chromIndex (Genome.h:20).
//
//   * the sort: the reference sorts by (diagonal, read position); two matches that compare equal
//     are the same (q,t) pair, so ANY correct sort yields the same array.  One 64-bit key per match
//     ((q - t + 2^32) << 31 | q forward, (uint32)(q + t) << 32 | q reverse) and a segmented radix
//     sort (rocPRIM, a plain library sort) over the 2 x n_reads strand segments of the batch;
//   * the cleaning passes are short serial scans with data-dependent run boundaries: one lane per
//     (read, strand) segment, streaming its own slice of the sorted arrays; the distinct-key count
//     of AVGfreq uses a per-segment open-addressing table in HBM tagged by run number (no clears);
//   * clusters are written at capacity offsets and compacted by one scan.
#include <cstring>
#include <cstdlib>
#include "seed_state.h"
#include "scan.h"
#include <algorithm>
#include <rocprim/rocprim.hpp>

namespace {

struct CleanArgs {
  int n_reads;
  lra_clean_opts o;
  const uint64_t* match_off; const uint32_t* n_forward;
  const uint32_t* sq; const uint32_t* st; const uint64_t* sk;       // sorted matches (q, t, read key)
  uint32_t* cl_q; uint32_t* cl_t;                                   // cleaned matches (capacity layout)
  unsigned char* onDiag; unsigned char* second; unsigned char* fw; unsigned char* rv; int* count; float* freq;
  uint64_t* tab_key; uint32_t* tab_tag;
  const uint64_t* chrom_pos; int n_chrom;
  // clusters at capacity offsets (segment base), per-segment count
  uint64_t* c_start; uint64_t* c_end; uint32_t* c_qs; uint32_t* c_qe; uint32_t* c_ts; uint32_t* c_te; int* c_strand; int* c_chrom; float* c_freq;
  uint32_t* seg_ncl;
};

__global__ void key_build(int n_reads, const uint64_t* __restrict__ match_off, const uint32_t* __restrict__ n_forward,
                          const uint32_t* __restrict__ q, const uint32_t* __restrict__ t, uint64_t* __restrict__ key, uint32_t* __restrict__ val,
                          uint64_t* __restrict__ seg_off) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
  for (int r = wave; r < n_reads; r += nw) {
    const uint64_t m0 = match_off[r], m1 = match_off[r + 1], mf = m0 + n_forward[r];
    if (lane == 0) { seg_off[2 * r] = m0; seg_off[2 * r + 1] = mf; if (r == n_reads - 1) seg_off[2 * n_reads] = m1; }
    for (uint64_t i = m0 + lane; i < m1; i += 64) {
      const uint32_t qq = q[i], tt = t[i];
      key[i] = (i < mf) ? ((((uint64_t)((int64_t)qq - (int64_t)tt + (1LL << 32))) << 31) | (uint64_t)qq)   // Sorting.h:34-47
                        : (((uint64_t)(uint32_t)(qq + tt) << 32) | (uint64_t)qq);                           // Sorting.h:74-88
      val[i] = (uint32_t)i;
    }
  }
}

__global__ void gather_sorted(uint64_t n, const uint32_t* __restrict__ val, const uint32_t* __restrict__ q, const uint32_t* __restrict__ t,
                              const uint64_t* __restrict__ k, uint32_t* __restrict__ sq, uint32_t* __restrict__ st, uint64_t* __restrict__ sk) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t v = val[i];
  sq[i] = q[v]; st[i] = t[v]; sk[i] = k[v];
}

__device__ __forceinline__ long diag_diff(uint32_t aq, uint32_t at, uint32_t bq, uint32_t bt, int strand) {   // Clustering.h:503-514
  if (strand == 0) return ((long)at - (long)aq) - ((long)bt - (long)bq);
  return (long)(uint32_t)(aq + at) - (long)(uint32_t)(bq + bt);
}

__device__ int header_find(const uint64_t* pos, int npos, uint64_t query) {   // Genome.h:20-32
  if (npos > 0 && query == pos[0]) return 0;
  int lo = 0, hi = npos;
  while (lo < hi) { int mid = (lo + hi) >> 1; if (pos[mid] < query) lo = mid + 1; else hi = mid; }
  if (lo < npos && pos[lo] == query) return lo;
  return lo - 1;
}

// SecondRoundCleanOffDiagonal (:802-868) on [os, oe)
__device__ void second_round(const CleanArgs& A, uint64_t base, const uint32_t* q, const uint32_t* t, int out_counter, int MinDiagCluster, int os, int oe, int strand) {
  unsigned char* second = A.second + base; unsigned char* fw = A.fw + base; unsigned char* rv = A.rv + base; int* count = A.count + base;
  if (MinDiagCluster >= oe - os) return;
  if (MinDiagCluster <= 0) { for (int i = os; i < oe; i++) { second[i] = 1; count[i] = out_counter; } return; }
  if (oe - os <= 1) return;
  const int cmd = A.o.SecondCleanMaxDiag;
  for (int i = os; i < oe; i++) { fw[i] = 0; rv[i] = 0; }
  for (int i = os + 1; i < oe; i++) if (labs(diag_diff(q[i], t[i], q[i - 1], t[i - 1], strand)) < cmd) fw[i - 1] = 1;
  bool prev = false; int ds = 0;
  for (int i = os; i < oe; i++) {
    if (!prev && fw[i]) ds = i;
    if (prev && !fw[i]) {
      if (i - ds + 1 < MinDiagCluster) { for (int j = ds; j <= i; j++) fw[j] = 0; }
      else fw[i] = 1;
    }
    prev = fw[i];
  }
  for (int i = oe - 2; i >= os; i--) if (labs(diag_diff(q[i], t[i], q[i + 1], t[i + 1], strand)) < cmd) rv[i + 1] = 1;
  prev = false;
  for (int i = oe - 1; i >= os; i--) {
    if (!prev && rv[i]) ds = i;
    if (prev && !rv[i]) {
      if (ds - i + 1 < MinDiagCluster) { for (int j = i; j <= ds; j++) rv[j] = 0; }
      else rv[i] = 1;
    }
    prev = rv[i];
  }
  for (int i = os; i < oe; i++) {
    if (fw[i] && rv[i]) { second[i] = 1; count[i] = out_counter; }
    else second[i] = 0;
  }
}

constexpr int CLEAN_LANES = 16;
__global__ void __launch_bounds__(64) clean_kernel(CleanArgs A) {
  if (threadIdx.x >= CLEAN_LANES) return;                                // serial scans with dependent loads: fewer lanes per wave, more waves
  const long seg = (long)blockIdx.x * CLEAN_LANES + threadIdx.x;
  if (seg >= 2L * A.n_reads) return;
  const int r = (int)(seg >> 1), strand = (int)(seg & 1);
  const uint64_t m0 = A.match_off[r], mf = m0 + A.n_forward[r], m1 = A.match_off[r + 1];
  const uint64_t base = strand ? mf : m0;
  const int n = (int)((strand ? m1 : mf) - base);
  A.seg_ncl[seg] = 0;
  if (n == 0) return;                                                    // :568-570
  const uint32_t* q = A.sq + base; const uint32_t* t = A.st + base; const uint64_t* key = A.sk + base;
  unsigned char* onDiag = A.onDiag + base; unsigned char* second = A.second + base;
  int* count = A.count + base; float* freq = A.freq + base;
  const lra_clean_opts& o = A.o;
  for (int i = 0; i < n; i++) { onDiag[i] = 0; second[i] = 0; count[i] = -1; freq[i] = 1.0f; }
  if (n > 1 && labs(diag_diff(q[0], t[0], q[1], t[1], strand)) < o.cleanMaxDiag) onDiag[0] = 1;              // :573-576
  for (int i = 1; i < n; i++) if (labs(diag_diff(q[i], t[i], q[i - 1], t[i - 1], strand)) < o.cleanMaxDiag) onDiag[i - 1] = 1;   // :578-584
  bool prev = false, startSet = false;
  int diagStart = 0, largest = 0;
  for (int i = 0; i < n; i++) {                                          // :589-598
    const bool od = onDiag[i];
    if (!prev && od) { diagStart = i; startSet = true; }
    if (prev && !od) largest = max(largest, i - diagStart + 1);
    prev = od;
  }
  if (!startSet) return;                                                 // :600-603
  largest = max(largest, n - diagStart);
  int minDiagCluster = largest / 10;                                     // :608-609
  if (minDiagCluster >= o.minDiagCluster) minDiagCluster = o.minDiagCluster;
  // AVGfreq table: power of two >= 2n slots at 4*base
  uint32_t tsz = 1; while (tsz < 2u * (uint32_t)n) tsz <<= 1;
  uint64_t* tk = A.tab_key + 4 * base; uint32_t* tg = A.tab_tag + 4 * base;
  int counter = 0;
  prev = false;
  if (minDiagCluster >= 0) {
    for (int i = 0; i < n; i++) {                                        // :620-722
      const bool od = onDiag[i];
      if (!prev && od) diagStart = i;
      if (prev && !od) {
        const int len = i - diagStart + 1;
        if (len >= minDiagCluster) {
          int distinct = 0;                                              // AVGfreq :550-564
          const uint32_t tag = (uint32_t)counter + 1;
          for (int x = diagStart; x <= i; x++) {
            const uint64_t kk = key[x];
            uint32_t h = (uint32_t)((kk * 0x9E3779B97F4A7C15ULL) >> 40) & (tsz - 1);
            while (true) {
              if (tg[h] != tag) { tg[h] = tag; tk[h] = kk; distinct++; break; }
              if (tk[h] == kk) break;
              h = (h + 1) & (tsz - 1);
            }
          }
          const float avgfreq = (float)len / (float)distinct;
          for (int j = diagStart; j <= i; j++) freq[j] = avgfreq;
          const int cc = o.cleanClustersize;
          int MinDiagCluster = 0;
          bool keepAll = false;
          if (o.bypassClustering) {                                      // :635-657
            if (avgfreq >= 3.0f && len < 10) {}
            else if (avgfreq >= 2.0f && len >= cc) {
              MinDiagCluster = (int)((float)o.SecondCleanMinDiagCluster + floorf((avgfreq - 1.5f) / 1.0f) * (float)o.punish_anchorfreq + (float)(((len - cc) / cc) * o.anchorPerlength));
              second_round(A, base, q, t, counter, MinDiagCluster, diagStart, i + 1, strand);
            } else if (avgfreq >= 1.5f && len >= cc) {
              MinDiagCluster = (int)((float)o.SecondCleanMinDiagCluster + floorf((avgfreq - 1.5f) / 1.5f) * (float)o.punish_anchorfreq + (float)(((len - cc) / cc) * o.anchorPerlength));
              second_round(A, base, q, t, counter, MinDiagCluster, diagStart, i + 1, strand);
            } else keepAll = true;
          } else {                                                       // :659-693
            if (avgfreq >= 3.0f && len < 10) {}
            else if (avgfreq >= 4.0f && len >= cc) {
              MinDiagCluster = (int)((float)o.SecondCleanMinDiagCluster + floorf((avgfreq - 1.5f) / 1.0f) * (float)o.punish_anchorfreq + (float)(((len - cc) / cc) * o.anchorPerlength));
              second_round(A, base, q, t, counter, MinDiagCluster, diagStart, i + 1, strand);
            } else if (avgfreq >= 1.5f && len >= cc) {
              MinDiagCluster = (int)((float)o.SecondCleanMinDiagCluster + floorf((avgfreq - 1.5f) / 1.5f) * (float)o.punish_anchorfreq + (float)(((len - cc) / cc) * o.anchorPerlength));
              second_round(A, base, q, t, counter, MinDiagCluster, diagStart, i + 1, strand);
            } else if (avgfreq > 1.0f && len >= cc) {
              MinDiagCluster = (int)((float)o.SecondCleanMinDiagCluster - (5.0f - floorf((avgfreq - 1.0f) / 0.1f)) * (float)(o.punish_anchorfreq / 2) + (float)(((len - cc) / cc) * (o.anchorPerlength / 2)));
              second_round(A, base, q, t, counter, MinDiagCluster, diagStart, i + 1, strand);
            } else if (avgfreq > 1.0f) {
              MinDiagCluster = (int)((float)o.SecondCleanMinDiagCluster - (5.0f - floorf((avgfreq - 1.0f) / 0.1f)) * (float)(o.punish_anchorfreq / 2) - (float)(((cc - i + diagStart - 1) / 15) * (o.anchorPerlength / 2)));
              second_round(A, base, q, t, counter, MinDiagCluster, diagStart, i + 1, strand);
            } else keepAll = true;
          }
          if (keepAll) for (int j = diagStart; j <= i; j++) { second[j] = 1; count[j] = counter; }
        }
        counter++;
      }
      prev = od;
    }
  }
  // compaction (:728-738) and clusters (:740-797)
  uint32_t* oq = A.cl_q + base; uint32_t* ot = A.cl_t + base;
  int c = 0;
  for (int i = 0; i < n; i++)
    if (second[i]) { oq[c] = q[i]; ot[c] = t[i]; freq[c] = freq[i]; count[c] = count[i]; c++; }
  uint32_t ncl = 0;
  auto emit = [&](int s, int e) {
    uint32_t qS = oq[s], qE = oq[s] + (uint32_t)o.globalK, tS = ot[s], tE = ot[s] + (uint32_t)o.globalK;
    for (int b = s; b < e; b++) {
      qS = min(qS, oq[b]); qE = max(qE, oq[b] + (uint32_t)o.globalK);
      tS = min(tS, ot[b]); tE = max(tE, ot[b] + (uint32_t)o.globalK);
    }
    const uint64_t x = base + ncl;
    A.c_start[x] = base + s; A.c_end[x] = base + e; A.c_qs[x] = qS; A.c_qe[x] = qE; A.c_ts[x] = tS; A.c_te[x] = tE;
    A.c_strand[x] = strand; A.c_freq[x] = freq[s];
    A.c_chrom[x] = header_find(A.chrom_pos, A.n_chrom + 1, tS);
    ncl++;
  };
  int count_s = 0, cc2 = 1;
  while (cc2 < c) {
    if (count[cc2] == count[cc2 - 1]) { cc2++; continue; }
    emit(count_s, cc2);
    count_s = cc2;
    cc2++;
  }
  if (cc2 == c && count_s < cc2) emit(count_s, cc2);
  A.seg_ncl[seg] = ncl;
}

// ---- the same stage with a WAVE per (read, strand) segment.  clean_kernel walks a segment with one lane: every pass over its matches is a chain of dependent loads, and a
// read out of a satellite array has tens of thousands of matches -- the launch lasted as long as its largest segment.  Every pass is a map, a scan or a reduction:
//   * the diagonal runs (:589-722): from the ballots of the neighbour flags, 64 matches per step (runs(): calls back once per run, wave-uniformly, in order);
//   * AVGfreq (:550-564): the run's distinct read k-mers through a compare-and-swap hash table (the run's own region of tab_key: cleared, then one CAS per probe);
//   * SecondRoundCleanOffDiagonal (:802-868) in closed form.  Its forward pass clears the runs of neighbours that are too short, but once a run of MinDiagCluster
//     elements has closed, `prev` stays set and every later element is flagged (the run start is never moved again, so every later run "is long enough"); the backward
//     pass does the same downwards.  What survives both is the span from the start of the FIRST long-enough run to the closing element of the LAST one;
//   * compaction and the clusters' boxes: ballot prefix sums and min / max reductions.
__device__ __forceinline__ void cl_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// flag(i) for i in [lo, hi), flag(hi - 1) == 0: calls fn(ds, e) for every maximal run of set flags [ds, e - 1] (e: the element that closes it), in order.
template <typename Flag, typename Fn>
__device__ __forceinline__ void runs(int lo, int hi, int lane, Flag flag, Fn fn) {
  int start = -1;                                                          // start of a run that is still open
  for (int b = lo; b < hi; b += 64) {
    const int i = b + lane;
    const unsigned long long m = __ballot(i < hi && flag(i));
    int pos = 0;
    while (pos < 64) {
      if (start >= 0) {
        const unsigned long long z = ~m >> pos;                           // (bits shifted in at the top are set flags' complements of nothing: zeros of ~m are ones of m)
        if (z == 0ULL) break;
        const int p = pos + (__ffsll((long long)z) - 1);
        if (p >= 64) break;
        fn(start, b + p);
        start = -1; pos = p + 1;
      } else {
        const unsigned long long o = m >> pos;
        if (o == 0ULL) break;
        const int p = pos + (__ffsll((long long)o) - 1);
        start = b + p; pos = p + 1;
      }
    }
  }
}
__device__ __forceinline__ int cl_wave_sum(int v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ __forceinline__ uint32_t cl_wave_min(uint32_t v) { for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o)); return v; }
__device__ __forceinline__ uint32_t cl_wave_max(uint32_t v) { for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o)); return v; }

__global__ void __launch_bounds__(64) clean_wave_kernel(CleanArgs A) {
  const int lane = threadIdx.x;
  const long seg = blockIdx.x;
  if (seg >= 2L * A.n_reads) return;
  const int r = (int)(seg >> 1), strand = (int)(seg & 1);
  const uint64_t m0 = A.match_off[r], mf = m0 + A.n_forward[r], m1 = A.match_off[r + 1];
  const uint64_t base = strand ? mf : m0;
  const int n = (int)((strand ? m1 : mf) - base);
  if (lane == 0) A.seg_ncl[seg] = 0;
  if (n == 0) return;                                                    // :568-570
  const uint32_t* q = A.sq + base; const uint32_t* t = A.st + base; const uint64_t* key = A.sk + base;
  unsigned char* second = A.second + base;
  int* count = A.count + base; float* freq = A.freq + base;
  const lra_clean_opts& o = A.o;
  for (int i = lane; i < n; i += 64) { second[i] = 0; count[i] = -1; freq[i] = 1.0f; }
  auto onDiag = [&](int i) -> bool { return i + 1 < n && labs(diag_diff(q[i + 1], t[i + 1], q[i], t[i], strand)) < o.cleanMaxDiag; };   // :573-584
  int largest = 0, lastStart = -1;
  runs(0, n, lane, onDiag, [&](int ds, int e) { largest = max(largest, e - ds + 1); lastStart = ds; });                        // :589-598
  if (lastStart < 0) return;                                             // :600-603
  largest = max(largest, n - lastStart);
  int minDiagCluster = largest / 10;                                     // :608-609
  if (minDiagCluster >= o.minDiagCluster) minDiagCluster = o.minDiagCluster;
  cl_wave_sync();
  int counter = 0;
  if (minDiagCluster >= 0) {
    runs(0, n, lane, onDiag, [&](int diagStart, int i) {                 // :620-722
      const int len = i - diagStart + 1;
      if (len >= minDiagCluster) {
        // AVGfreq :550-564
        uint32_t tsz = 2; while (tsz < 2u * (uint32_t)len) tsz <<= 1;
        unsigned long long* T = (unsigned long long*)(A.tab_key + 4 * (base + (uint64_t)diagStart));
        for (uint32_t x = lane; x < tsz; x += 64) T[x] = ~0ULL;
        cl_wave_sync();
        int d = 0;
        for (int x = diagStart + lane; x <= i; x += 64) {
          const unsigned long long kk = key[x];
          uint32_t h = (uint32_t)((kk * 0x9E3779B97F4A7C15ULL) >> 40) & (tsz - 1);
          while (true) {
            const unsigned long long old = atomicCAS(&T[h], ~0ULL, kk);
            if (old == ~0ULL) { d++; break; }
            if (old == kk) break;
            h = (h + 1) & (tsz - 1);
          }
        }
        const int distinct = cl_wave_sum(d);
        const float avgfreq = (float)len / (float)distinct;
        for (int j = diagStart + lane; j <= i; j += 64) freq[j] = avgfreq;
        const int cc = o.cleanClustersize;
        int MinDiagCluster = 0;
        bool keepAll = false, secondRound = false;
        if (o.bypassClustering) {                                        // :635-657
          if (avgfreq >= 3.0f && len < 10) {}
          else if (avgfreq >= 2.0f && len >= cc) {
            MinDiagCluster = (int)((float)o.SecondCleanMinDiagCluster + floorf((avgfreq - 1.5f) / 1.0f) * (float)o.punish_anchorfreq + (float)(((len - cc) / cc) * o.anchorPerlength));
            secondRound = true;
          } else if (avgfreq >= 1.5f && len >= cc) {
            MinDiagCluster = (int)((float)o.SecondCleanMinDiagCluster + floorf((avgfreq - 1.5f) / 1.5f) * (float)o.punish_anchorfreq + (float)(((len - cc) / cc) * o.anchorPerlength));
            secondRound = true;
          } else keepAll = true;
        } else {                                                         // :659-693
          if (avgfreq >= 3.0f && len < 10) {}
          else if (avgfreq >= 4.0f && len >= cc) {
            MinDiagCluster = (int)((float)o.SecondCleanMinDiagCluster + floorf((avgfreq - 1.5f) / 1.0f) * (float)o.punish_anchorfreq + (float)(((len - cc) / cc) * o.anchorPerlength));
            secondRound = true;
          } else if (avgfreq >= 1.5f && len >= cc) {
            MinDiagCluster = (int)((float)o.SecondCleanMinDiagCluster + floorf((avgfreq - 1.5f) / 1.5f) * (float)o.punish_anchorfreq + (float)(((len - cc) / cc) * o.anchorPerlength));
            secondRound = true;
          } else if (avgfreq > 1.0f && len >= cc) {
            MinDiagCluster = (int)((float)o.SecondCleanMinDiagCluster - (5.0f - floorf((avgfreq - 1.0f) / 0.1f)) * (float)(o.punish_anchorfreq / 2) + (float)(((len - cc) / cc) * (o.anchorPerlength / 2)));
            secondRound = true;
          } else if (avgfreq > 1.0f) {
            MinDiagCluster = (int)((float)o.SecondCleanMinDiagCluster - (5.0f - floorf((avgfreq - 1.0f) / 0.1f)) * (float)(o.punish_anchorfreq / 2) - (float)(((cc - i + diagStart - 1) / 15) * (o.anchorPerlength / 2)));
            secondRound = true;
          } else keepAll = true;
        }
        if (secondRound) {                                               // SecondRoundCleanOffDiagonal (:802-868) on [os, oe)
          const int os = diagStart, oe = i + 1;
          if (MinDiagCluster >= oe - os) {}
          else if (MinDiagCluster <= 0) keepAll = true;
          else if (oe - os <= 1) {}
          else {
            const int cmd = o.SecondCleanMaxDiag;
            int aF = 0x7fffffff, bL = -1;
            runs(os, oe, lane, [&](int x) -> bool { return x + 1 < oe && labs(diag_diff(q[x + 1], t[x + 1], q[x], t[x], strand)) < cmd; },
                 [&](int ds, int e) { if (e - ds + 1 >= MinDiagCluster) { aF = min(aF, ds); bL = max(bL, e); } });
            for (int j = os + lane; j < oe; j += 64) {
              if (j >= aF && j <= bL) { second[j] = 1; count[j] = counter; }
              else second[j] = 0;
            }
          }
        }
        if (keepAll) for (int j = diagStart + lane; j <= i; j += 64) { second[j] = 1; count[j] = counter; }
      }
      counter++;
    });
  }
  cl_wave_sync();
  // compaction (:728-738) and clusters (:740-797)
  uint32_t* oq = A.cl_q + base; uint32_t* ot = A.cl_t + base;
  const unsigned long long below = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));
  int c = 0;
  for (int b = 0; b < n; b += 64) {
    const int i = b + lane;
    const bool keep = i < n && second[i];
    uint32_t vq = 0, vt = 0; float vf = 0; int vc = 0;
    if (keep) { vq = q[i]; vt = t[i]; vf = freq[i]; vc = count[i]; }
    const unsigned long long m = __ballot(keep);
    cl_wave_sync();                                                        // (freq / count are compacted in place: this step's reads before its writes)
    if (keep) { const int d = c + __popcll(m & below); oq[d] = vq; ot[d] = vt; freq[d] = vf; count[d] = vc; }
    c += __popcll(m);
  }
  cl_wave_sync();
  uint32_t ncl = 0;
  auto emit = [&](int s0, int e0) {
    uint32_t qS = 0xFFFFFFFFu, qE = 0, tS = 0xFFFFFFFFu, tE = 0;
    for (int b = s0 + lane; b < e0; b += 64) {
      qS = min(qS, oq[b]); qE = max(qE, oq[b] + (uint32_t)o.globalK);
      tS = min(tS, ot[b]); tE = max(tE, ot[b] + (uint32_t)o.globalK);
    }
    qS = cl_wave_min(qS); qE = cl_wave_max(qE); tS = cl_wave_min(tS); tE = cl_wave_max(tE);
    if (lane == 0) {
      const uint64_t x = base + ncl;
      A.c_start[x] = base + s0; A.c_end[x] = base + e0; A.c_qs[x] = qS; A.c_qe[x] = qE; A.c_ts[x] = tS; A.c_te[x] = tE;
      A.c_strand[x] = strand; A.c_freq[x] = freq[s0];
      A.c_chrom[x] = header_find(A.chrom_pos, A.n_chrom + 1, tS);
    }
    ncl++;
  };
  int count_s = 0;
  for (int b = 0; b < c; b += 64) {                                        // a cluster ends where the run counter changes
    const int i = b + lane;
    unsigned long long m = __ballot(i >= 1 && i < c && count[i] != count[i - 1]);
    while (m) { const int p = __ffsll((long long)m) - 1; m &= m - 1; emit(count_s, b + p); count_s = b + p; }
  }
  if (c > 0 && count_s < c) emit(count_s, c);
  if (lane == 0) A.seg_ncl[seg] = ncl;
}

// compaction of the per-segment cluster records (capacity layout: segment base = its first match)
struct CompactArgs {
  long n_seg; const uint64_t* seg_base; const uint64_t* seg_coff;
  const uint64_t* c_start; const uint64_t* c_end; const uint32_t* c_qs; const uint32_t* c_qe; const uint32_t* c_ts; const uint32_t* c_te;
  const int* c_strand; const int* c_chrom; const float* c_freq;
  uint64_t* o_start; uint64_t* o_end; uint32_t* o_qs; uint32_t* o_qe; uint32_t* o_ts; uint32_t* o_te; int* o_strand; int* o_chrom; float* o_freq;
  uint64_t* cluster_off; int n_reads;
};
__global__ void compact_clusters(CompactArgs C) {
  const long seg = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (seg <= C.n_reads && 2 * seg <= C.n_seg) C.cluster_off[seg] = C.seg_coff[2 * seg];
  if (seg >= C.n_seg) return;
  const uint64_t b = C.seg_base[seg], o = C.seg_coff[seg], n = C.seg_coff[seg + 1] - o;
  for (uint64_t x = 0; x < n; x++) {
    C.o_start[o + x] = C.c_start[b + x]; C.o_end[o + x] = C.c_end[b + x];
    C.o_qs[o + x] = C.c_qs[b + x]; C.o_qe[o + x] = C.c_qe[b + x]; C.o_ts[o + x] = C.c_ts[b + x]; C.o_te[o + x] = C.c_te[b + x];
    C.o_strand[o + x] = C.c_strand[b + x]; C.o_chrom[o + x] = C.c_chrom[b + x]; C.o_freq[o + x] = C.c_freq[b + x];
  }
}

template <typename T>
static T* carve(char*& p, size_t n) {
  T* r = (T*)p;
  p += (n * sizeof(T) + 255) & ~(size_t)255;
  return r;
}

}  // namespace

extern "C" int lra_clean_matches_batch(lra_ctx* ctx, const lra_clean_opts* opts, const uint64_t* h_chrom_pos, int n_chrom,
                                       lra_cluster_result* out) {
  if (!ctx || !opts || !out || !h_chrom_pos || n_chrom < 1) return LRA_ERR_INVALID;
  lra_seed_state* s = ctx->seed;
  if (!s || !s->sep_qpos) return lra_set_err(ctx, LRA_ERR_INVALID, "run lra_seed_batch first");
  memset(out, 0, sizeof(*out));
  const int n_reads = s->last_n_reads;
  const uint64_t nm = s->last_n_matches;
  out->n_reads = n_reads;
  if (n_reads == 0) return LRA_OK;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t N = (size_t)nm + 64, NS = 2 * (size_t)n_reads + 2;
  // ---- work buffers (gbuf 3) and results (gbuf 4)
  size_t temp_bytes = 0;
  if (nm) {
    (void)lra_segsort_pairs(ctx, nullptr, temp_bytes, nullptr, nullptr, nullptr, nullptr, (unsigned int)nm, (unsigned int)(2 * n_reads), nullptr, nullptr, 0, 64, st);
  }
  auto sz = [](size_t n, size_t e) { return (n * e + 255) & ~(size_t)255; };
  size_t needW = sz(N, 8) * 2 + sz(N, 4) * 2 + sz(NS, 8) + sz(N, 4) * 2 + sz(N, 8) + sz(N, 1) * 4 + sz(N, 4) * 2 + sz(4 * N, 8) + sz(4 * N, 4) +
                 sz(n_chrom + 2, 8) + sz(N, 8) * 2 + sz(N, 4) * 6 + sz(N, 4) + sz(NS, 4) + sz(NS, 8) + temp_bytes + 4096;
  char* w = (char*)lra_ensure(ctx, 3, needW);
  if (!w) return LRA_ERR_NOMEM;
  uint64_t* key_in = carve<uint64_t>(w, N); uint64_t* key_out = carve<uint64_t>(w, N);
  uint32_t* val_in = carve<uint32_t>(w, N); uint32_t* val_out = carve<uint32_t>(w, N);
  uint64_t* seg_off = carve<uint64_t>(w, NS);
  CleanArgs A;
  A.n_reads = n_reads; A.o = *opts; A.match_off = s->match_off; A.n_forward = s->n_forward;
  uint32_t* sq = carve<uint32_t>(w, N); uint32_t* stt = carve<uint32_t>(w, N); uint64_t* sk = carve<uint64_t>(w, N);
  A.sq = sq; A.st = stt; A.sk = sk;
  A.onDiag = carve<unsigned char>(w, N); A.second = carve<unsigned char>(w, N); A.fw = carve<unsigned char>(w, N); A.rv = carve<unsigned char>(w, N);
  A.count = carve<int>(w, N); A.freq = carve<float>(w, N);
  A.tab_key = carve<uint64_t>(w, 4 * N); A.tab_tag = carve<uint32_t>(w, 4 * N);
  uint64_t* d_chrom = carve<uint64_t>(w, n_chrom + 2);
  A.chrom_pos = d_chrom; A.n_chrom = n_chrom;
  A.c_start = carve<uint64_t>(w, N); A.c_end = carve<uint64_t>(w, N);
  A.c_qs = carve<uint32_t>(w, N); A.c_qe = carve<uint32_t>(w, N); A.c_ts = carve<uint32_t>(w, N); A.c_te = carve<uint32_t>(w, N);
  A.c_strand = carve<int>(w, N); A.c_chrom = carve<int>(w, N); A.c_freq = carve<float>(w, N);
  A.seg_ncl = carve<uint32_t>(w, NS);
  uint64_t* seg_coff = carve<uint64_t>(w, NS);
  void* temp = (void*)w;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(d_chrom, h_chrom_pos, (size_t)(n_chrom + 1) * 8, hipMemcpyHostToDevice, st));
  static const bool cleanSerial = getenv("LRA_CLEAN_SERIAL") != nullptr;   // the one-lane-per-segment walk (kept for comparison)
  if (cleanSerial) LRA_HIP_CHECK(ctx, hipMemsetAsync(A.tab_tag, 0, 4 * N * 4, st));   // (its hash table's tags; clean_wave_kernel clears what it uses)
  // results: cleaned matches
  char* rbuf = (char*)lra_ensure(ctx, 4, sz(N, 4) * 2 + 4096);
  if (!rbuf) return LRA_ERR_NOMEM;
  A.cl_q = carve<uint32_t>(rbuf, N); A.cl_t = carve<uint32_t>(rbuf, N);
  // ---- sort
  lra_time_begin(ctx, "clean_sort");
  hipLaunchKernelGGL(key_build, dim3(ctx->num_cu * 8), dim3(256), 0, st, n_reads, s->match_off, s->n_forward, s->sep_qpos, s->sep_tpos, key_in, val_in, seg_off);
  if (nm) {
    hipError_t e = lra_segsort_pairs(ctx, temp, temp_bytes, key_in, key_out, val_in, val_out, (unsigned int)nm, (unsigned int)(2 * n_reads), seg_off, seg_off + 1, 0, 64, st);
    if (e != hipSuccess) return lra_set_err(ctx, LRA_ERR_HIP, "segmented sort: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(gather_sorted, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, st, nm, val_out, s->sep_qpos, s->sep_tpos, s->sep_qkey, sq, stt, sk);
  }
  lra_time_end(ctx);
  // ---- clean
  lra_time_begin(ctx, "clean");
  if (cleanSerial) hipLaunchKernelGGL(clean_kernel, dim3((2 * n_reads + CLEAN_LANES - 1) / CLEAN_LANES), dim3(64), 0, st, A);
  else hipLaunchKernelGGL(clean_wave_kernel, dim3(2 * n_reads), dim3(64), 0, st, A);
  lra_time_end(ctx);
  if (lra_exclusive_scan<uint32_t>(ctx, 2L * n_reads, A.seg_ncl, seg_coff)) return LRA_ERR_HIP;
  uint64_t ncl = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&ncl, seg_coff + 2 * n_reads, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  const size_t NC = (size_t)ncl + 8;
  char* cb = (char*)lra_ensure(ctx, 5, sz(NC, 8) * 2 + sz(NC, 4) * 7 + sz((size_t)n_reads + 2, 8) + 4096);
  if (!cb) return LRA_ERR_NOMEM;
  CompactArgs C;
  C.n_seg = 2L * n_reads; C.seg_base = seg_off; C.seg_coff = seg_coff;
  C.c_start = A.c_start; C.c_end = A.c_end; C.c_qs = A.c_qs; C.c_qe = A.c_qe; C.c_ts = A.c_ts; C.c_te = A.c_te;
  C.c_strand = A.c_strand; C.c_chrom = A.c_chrom; C.c_freq = A.c_freq;
  C.o_start = carve<uint64_t>(cb, NC); C.o_end = carve<uint64_t>(cb, NC);
  C.o_qs = carve<uint32_t>(cb, NC); C.o_qe = carve<uint32_t>(cb, NC); C.o_ts = carve<uint32_t>(cb, NC); C.o_te = carve<uint32_t>(cb, NC);
  C.o_strand = carve<int>(cb, NC); C.o_chrom = carve<int>(cb, NC); C.o_freq = carve<float>(cb, NC);
  C.cluster_off = carve<uint64_t>(cb, (size_t)n_reads + 2); C.n_reads = n_reads;
  hipLaunchKernelGGL(compact_clusters, dim3((unsigned)((2L * n_reads + 256) / 256)), dim3(256), 0, st, C);
  LRA_HIP_CHECK(ctx, hipGetLastError());
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  out->n_clusters = ncl; out->n_matches = nm;
  out->d_cluster_off = C.cluster_off; out->d_c_start = C.o_start; out->d_c_end = C.o_end;
  out->d_c_qStart = C.o_qs; out->d_c_qEnd = C.o_qe; out->d_c_tStart = C.o_ts; out->d_c_tEnd = C.o_te;
  out->d_c_strand = C.o_strand; out->d_c_chrom = C.o_chrom; out->d_c_anchorfreq = C.o_freq;
  out->d_cl_qpos = A.cl_q; out->d_cl_tpos = A.cl_t;
  if (!ctx->clus) ctx->clus = new lra_cluster_state();
  lra_cluster_state* cs = ctx->clus;
  cs->n_reads = n_reads; cs->n_clusters = ncl; cs->n_matches = nm;
  cs->cluster_off = C.cluster_off; cs->c_start = C.o_start; cs->c_end = C.o_end; cs->c_strand = C.o_strand; cs->c_chrom = C.o_chrom;
  cs->cl_q = A.cl_q; cs->cl_t = A.cl_t; cs->chrom_pos = d_chrom; cs->n_chrom = n_chrom;
  return LRA_OK;
}

void lra_cluster_free(lra_ctx* ctx) { delete ctx->clus; ctx->clus = nullptr; }

// ======================================================================================== a7
namespace {

struct ExtArgs {
  uint64_t n_clusters; int n_reads; int K;
  const uint64_t* cluster_off; const uint64_t* c_start; const uint64_t* c_end; const int* c_strand; const int* c_chrom;
  const uint32_t* cl_q; const uint32_t* cl_t; const uint64_t* chrom_pos;
  const unsigned char* genome; const unsigned char* seq; const uint64_t* read_off;
  int* c_read; const int* c_K;             // per-cluster K when non-null (the gap seeds of RefinedAlignmentbtwnAnchors use 9 or 12)
  uint32_t* e_q; uint32_t* e_t; int* e_len; uint32_t* e_count; uint32_t* box;
};

__global__ void cluster_read_map(int n_reads, const uint64_t* cluster_off, int* c_read) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_reads) return;
  for (uint64_t c = cluster_off[r]; c < cluster_off[r + 1]; c++) c_read[c] = r;
}

// One wave per cluster.  Whether two consecutive matches fuse (same diagonal and either overlapping
// or joined by an exact-match walk, Checkbp) depends on that pair alone, so every pair is tested by
// its own lane; the extended anchors are the stretches between the break points.
__global__ void __launch_bounds__(64) linear_extend_kernel(ExtArgs A) {
  const int lane = threadIdx.x;
  const unsigned long long below = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));
  for (uint64_t x = blockIdx.x; x < A.n_clusters; x += gridDim.x) {
    const uint32_t K = (uint32_t)(A.c_K ? A.c_K[x] : A.K);
    const uint64_t s = A.c_start[x], e = A.c_end[x];
    const int strand = A.c_strand[x], chrom = A.c_chrom[x], r = A.c_read[x];
    const uint64_t coff = A.chrom_pos[chrom];
    const uint32_t chromLen = (uint32_t)(A.chrom_pos[chrom + 1] - coff);
    const unsigned char* G = A.genome + coff;
    const unsigned char* R = A.seq + A.read_off[r];
    const uint32_t readLen = (uint32_t)(A.read_off[r + 1] - A.read_off[r]);
    uint64_t m = s;                     // start of the open stretch
    uint32_t nout = 0;
    uint32_t bqS = 0xFFFFFFFFu, bqE = 0, btS = 0xFFFFFFFFu, btE = 0;
    auto put = [&](uint32_t idx, uint32_t q, uint32_t t, int len) {
      A.e_q[s + idx] = q; A.e_t[s + idx] = t + (uint32_t)coff; A.e_len[s + idx] = len;
      bqS = min(bqS, q); bqE = max(bqE, q + (uint32_t)len); btS = min(btS, t + (uint32_t)coff); btE = max(btE, t + (uint32_t)coff + (uint32_t)len);
    };
    for (uint64_t base = s; base < e; base += 64) {
      const uint64_t i = base + lane;
      bool brk = false, ext = false;
      uint32_t qe = 0, te = 0, qp = 0, tp = 0;
      if (i > s && i < e) {
        const uint32_t q1 = A.cl_q[i - 1], t1 = A.cl_t[i - 1] - (uint32_t)coff, q2 = A.cl_q[i], t2 = A.cl_t[i] - (uint32_t)coff;
        qp = q1; tp = t1;
        const int64_t d1 = strand == 0 ? (int64_t)q1 - (int64_t)t1 : (int64_t)q1 + (int64_t)t1;
        const int64_t d2 = strand == 0 ? (int64_t)q2 - (int64_t)t2 : (int64_t)q2 + (int64_t)t2;
        if (d1 != d2) brk = true;                                              // :702-707
        else if (q2 >= q1 + K) {                                                // :681-699
          uint32_t curQ = q1 + K, curT;
          if (strand == 0) {                                                    // Checkbp :50-85
            curT = min(chromLen, t1 + K);
            const uint32_t nextT = min(chromLen, t2);
            while (curQ < readLen && curT < chromLen && q2 > curQ && nextT > curT && G[curT] == R[curQ]) { curQ++; curT++; }
            if (!(curQ == q2 && curT == t2)) { brk = true; ext = true; }
          } else {
            curT = min(chromLen - 1, t1 - 1);
            const uint32_t nextT = min(chromLen - 1, t2 + K - 1);
            while (curQ < readLen && q2 > curQ && nextT < curT && G[curT] == R[curQ]) { curQ++; curT--; }
            if (!(curQ == q2 && curT == t2 + K - 1)) { brk = true; ext = true; }
          }
          qe = curQ; te = curT;
        }
      }
      const unsigned long long mb = __ballot(brk);
      if (brk) {
        // the stretch this break closes starts at the previous break of the chunk, or at the carried m
        const unsigned long long prevb = mb & below;
        const uint64_t ms = prevb ? base + (63 - __clzll((long long)prevb)) : m;
        const uint32_t qm = A.cl_q[ms], tm = A.cl_t[ms] - (uint32_t)coff;
        const uint32_t idx = nout + __popcll(prevb);
        if (ext) put(idx, qm, strand == 0 ? tm : te + 1, (int)(qe - qm));
        else put(idx, qm, strand == 0 ? tm : tp, (int)(qp + K - qm));
      }
      if (mb) { m = base + (63 - __clzll((long long)mb)); nout += __popcll(mb); }
    }
    if (e > s) {                                                               // :710-714 the last stretch
      if (lane == 0) {
        const uint32_t qm = A.cl_q[m], tm = A.cl_t[m] - (uint32_t)coff;
        const uint32_t ql = A.cl_q[e - 1], tl = A.cl_t[e - 1] - (uint32_t)coff;
        put(nout, qm, strand == 0 ? tm : tl, (int)(ql + K - qm));
      }
      nout++;
    }
    for (int off = 32; off > 0; off >>= 1) {
      bqS = min(bqS, (uint32_t)__shfl_xor(bqS, off)); bqE = max(bqE, (uint32_t)__shfl_xor(bqE, off));
      btS = min(btS, (uint32_t)__shfl_xor(btS, off)); btE = max(btE, (uint32_t)__shfl_xor(btE, off));
    }
    if (lane == 0) { A.e_count[x] = nout; A.box[4 * x] = bqS; A.box[4 * x + 1] = bqE; A.box[4 * x + 2] = btS; A.box[4 * x + 3] = btE; }
  }
}

}  // namespace

// the pair-version LinearExtend on caller-supplied cluster arrays (merge_extend.hip: the refined clusters before the second sparse DP)
int lra_launch_linear_extend(lra_ctx* ctx, uint64_t n_clusters, int K, const uint64_t* c_start, const uint64_t* c_end, const int* c_strand, const int* c_chrom,
                             int* c_read, const uint32_t* cl_q, const uint32_t* cl_t, const uint64_t* d_chrom_pos, const unsigned char* genome,
                             const unsigned char* seq, const uint64_t* read_off, uint32_t* e_q, uint32_t* e_t, int* e_len, uint32_t* e_count, uint32_t* box,
                             const int* c_K) {
  if (n_clusters == 0) return LRA_OK;
  ExtArgs A;
  memset(&A, 0, sizeof A);
  A.n_clusters = n_clusters; A.K = K; A.c_start = c_start; A.c_end = c_end; A.c_strand = c_strand; A.c_chrom = c_chrom; A.c_read = c_read;
  A.cl_q = cl_q; A.cl_t = cl_t; A.chrom_pos = d_chrom_pos; A.genome = genome; A.seq = seq; A.read_off = read_off;
  A.e_q = e_q; A.e_t = e_t; A.e_len = e_len; A.e_count = e_count; A.box = box; A.c_K = c_K;
  lra_time_begin(ctx, "linear_extend");
  hipLaunchKernelGGL(linear_extend_kernel, dim3((unsigned)std::min<uint64_t>(n_clusters, (uint64_t)ctx->num_cu * 32)), dim3(64), 0, ctx->stream, A);
  lra_time_end(ctx);
  return LRA_OK;
}

extern "C" int lra_linear_extend_batch(lra_ctx* ctx, int K, const char* d_seq, const uint64_t* d_read_off, lra_extend_result* out) {
  if (!ctx || !out || K < 1) return LRA_ERR_INVALID;
  lra_cluster_state* cs = ctx->clus;
  if (!cs || !ctx->seed || !ctx->seed->genome) return lra_set_err(ctx, LRA_ERR_INVALID, "run lra_clean_matches_batch first");
  memset(out, 0, sizeof(*out));
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t NM = (size_t)cs->n_matches + 64, NC = (size_t)cs->n_clusters + 8;
  auto sz = [](size_t n, size_t e) { return (n * e + 255) & ~(size_t)255; };
  char* w = (char*)lra_scratch(ctx, 3, sz(NM, 4) * 3 + sz(NC, 4) * 2 + sz(4 * NC, 4) + 4096);
  if (!w) return LRA_ERR_NOMEM;
  ExtArgs A;
  A.n_clusters = cs->n_clusters; A.n_reads = cs->n_reads; A.K = K; A.c_K = nullptr;
  A.cluster_off = cs->cluster_off; A.c_start = cs->c_start; A.c_end = cs->c_end; A.c_strand = cs->c_strand; A.c_chrom = cs->c_chrom;
  A.cl_q = cs->cl_q; A.cl_t = cs->cl_t; A.chrom_pos = cs->chrom_pos;
  A.genome = ctx->seed->genome; A.seq = (const unsigned char*)d_seq; A.read_off = d_read_off;
  A.e_q = carve<uint32_t>(w, NM); A.e_t = carve<uint32_t>(w, NM); A.e_len = carve<int>(w, NM);
  A.e_count = carve<uint32_t>(w, NC); A.c_read = carve<int>(w, NC); A.box = carve<uint32_t>(w, 4 * NC);
  if (cs->n_clusters) {
    hipLaunchKernelGGL(cluster_read_map, dim3((cs->n_reads + 255) / 256), dim3(256), 0, st, cs->n_reads, cs->cluster_off, A.c_read);
    lra_time_begin(ctx, "linear_extend");
    hipLaunchKernelGGL(linear_extend_kernel, dim3((unsigned)std::min<uint64_t>(cs->n_clusters, (uint64_t)ctx->num_cu * 32)), dim3(64), 0, st, A);
    lra_time_end(ctx);
  }
  LRA_HIP_CHECK(ctx, hipGetLastError());
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  out->n_clusters = cs->n_clusters; out->n_anchors_cap = cs->n_matches;
  out->d_e_start = cs->c_start; out->d_e_count = A.e_count; out->d_e_qpos = A.e_q; out->d_e_tpos = A.e_t; out->d_e_len = A.e_len; out->d_box = A.box;
  return LRA_OK;
}
