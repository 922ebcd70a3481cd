// oracle/store_index.cpp -- TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// CPU restatement of `lra index` for the global minimizer index:
//   StoreIndex                     MMIndex.h:286-400   (StoreMinimizers per sequence MinCount.h:8, std::sort by masked key, frequency filter,
//                                                       CountSort :258-283 by frequency, <= NumOfminimizersPerWindow per globalWinsize window,
//                                                       RemoveFrequent :88-98)
// Parity status: PARITY UNPINNED -- MMIndex.h includes Genome.h / MinCount.h (htslib/kseq.h, not in this image); restated from the source.
//
// `stable`: 0 = the reference's order, std::sort on the minimizers in emission order (libstdc++'s permutation of equal keys decides which of
// two equal-key candidates inside one window survives, and the order of equal keys in the file); 1 = equal keys keep their emission order
// (what the device builder produces with a stable radix sort).  The two differ only inside runs of equal keys.
#include "oracle_common.h"
#include <algorithm>
#include <vector>

extern "C" long oracle_store_minimizers(const char* seq, uint32_t seqLen, int k, int w, uint64_t* keys, uint32_t* poss, long cap);

namespace {
const uint64_t FOR_MASK = ~(1ULL << 63);
struct GTup {
  uint64_t t; uint32_t pos;
  bool operator<(const GTup& b) const { return (t & FOR_MASK) < (b.t & FOR_MASK); }   // TupleOps.h:76
};
}  // namespace

// Returns the number of index entries (only the first cap are written to keys / poss).  *status: 1 if the reference would index winCount
// out of range (a minimizer in a last partial window that `sz` does not cover, MMIndex.h:356-361), else 0.
extern "C" long oracle_store_index(const char* genome, const uint64_t* chromPos, int nChrom, int k, int w, int maxFreq, int winsize, int nPerWin, int stable,
                                   uint64_t* keys, uint32_t* poss, long cap, int* status) {
  std::vector<GTup> mm;
  if (status) *status = 0;
  for (int c = 0; c < nChrom; c++) {                                     // :301-310
    const uint64_t off = chromPos[c], len = chromPos[c + 1] - off;
    const long guess = (long)len + 16;
    std::vector<uint64_t> kk(guess); std::vector<uint32_t> pp(guess);
    const long n = oracle_store_minimizers(genome + off, (uint32_t)len, k, w, kk.data(), pp.data(), guess);
    for (long i = 0; i < n; i++) mm.push_back(GTup{kk[i], (uint32_t)(pp[i] + off)});
  }
  if (stable) std::stable_sort(mm.begin(), mm.end());
  else std::sort(mm.begin(), mm.end());                                  // :314
  const size_t N = mm.size();
  std::vector<uint8_t> Remove(N, 0);
  std::vector<uint32_t> Freq(N, 0);
  uint32_t n = 0, ne = 0, unremoved = 0;
  while (n < N) {                                                        // :331-352
    ne = n + 1;
    while (ne < N && (mm[ne].t & FOR_MASK) == (mm[n].t & FOR_MASK)) ne++;
    const bool rm = ne - n > (uint32_t)maxFreq;
    for (uint32_t i = n; i < ne; i++) { Freq[i] = ne - n; Remove[i] = rm; }
    if (!rm) unremoved += ne - n;
    n = ne;
  }
  const uint64_t G = chromPos[nChrom];
  uint32_t sz = (uint32_t)(G / winsize);                                 // :359-360
  if (G / winsize % winsize > 0) sz += 1;
  std::vector<uint32_t> Sortindex(unremoved, 0);
  {                                                                      // CountSort :258-283
    std::vector<uint32_t> count(maxFreq + 1, 0);
    for (uint32_t i = 0; i < N; i++) if (!Remove[i]) ++count[Freq[i]];
    for (int i = 1; i <= maxFreq; i++) count[i] += count[i - 1];
    for (uint32_t i = 0; i < N; i++) if (!Remove[i]) { Sortindex[count[Freq[i]] - 1] = i; --count[Freq[i]]; }
  }
  std::vector<uint32_t> winCount((size_t)sz + 1, (uint32_t)nPerWin);      // (+1: one slot of slack for the out-of-range case reported in *status)
  for (uint32_t s = 0; s < Sortindex.size(); s++) {                       // :365-376
    const uint32_t id = mm[Sortindex[s]].pos / (uint32_t)winsize;
    if (id >= sz) { if (status) *status = 1; if (id > sz) continue; }
    if (winCount[id] > 0) winCount[id] -= 1;
    else Remove[Sortindex[s]] = 1;
  }
  long out = 0;
  for (size_t i = 0; i < N; i++)                                         // RemoveFrequent :88-98
    if (!Remove[i]) { if (out < cap) { keys[out] = mm[i].t; poss[out] = mm[i].pos; } out++; }
  return out;
}
