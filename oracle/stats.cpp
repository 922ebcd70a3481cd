// oracle/stats.cpp -- TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// CPU restatement of Alignment::CalculateStatistics (reference: Alignment.h:513-531), i.e.
// CreateAlignmentStrings (:247-331) + AlignStringsToCigar (:414-504) with opts.showmm (the default,
// Options.h:124): gapless blocks -> CIGAR runs ('=' 'X' 'I' 'D') + event / base counters + the
// float alignment value (NV).
// Parity status: PINNED -- bit-exact (CIGAR, all counters, the float value's bit pattern) against
// the reference's own Alignment.h compiled in place (oracle/ref_harness/stats_ref.cpp), golden in
// tests/golden/stats_golden.json.
//
// Counter naming follows the MEMBERS of the reference class after the call (CalculateStatistics
// passes (nm,nmm,ndel,nins) into parameters (nm,nmm,nins,ndel), :516 vs :414): nins = number of
// deletion runs, ndel = number of insertion runs, tdel = deleted bases, tins = inserted bases.
#include "oracle_common.h"
#include <cmath>
#include <vector>

// out_counts: nm nmm nins ndel tdel tins nSmallDel nMedDel nLargeDel nSmallIns nMedIns nLargeIns preClip sufClip qStart qEnd tStart tEnd
// runs: (len<<4 | op) with op 0 '=', 1 'X', 2 'I', 3 'D'; returns the number of runs (only cap written).
extern "C" long oracle_calculate_statistics(const int* blocks, long nb, const char* read, long readLen, const char* genome,
                                            const float* lut, long* out_counts, float* out_value, uint32_t* runs, long cap) {
  for (int i = 0; i < 18; i++) out_counts[i] = 0;
  *out_value = 0;
  if (nb == 0) return 0;
  // column stream (:261-330): per block its aligned pairs; between blocks the longer gap's excess
  // (insertion first, then deletion), then min(qGap,tGap) aligned pairs
  std::vector<unsigned char> col;   // 0 '=', 1 'X', 2 'I' (query base, target gap), 3 'D' (query gap, target base)
  long q = blocks[0], t = blocks[1];
  auto pair_col = [&]() { col.push_back(oracle_code((unsigned char)read[q]) != oracle_code((unsigned char)genome[t]) ? 1 : 0); q++; t++; };
  for (long b = 0; b < nb; b++) {
    for (long x = 0; x < blocks[3 * b + 2]; x++) pair_col();
    if (b == nb - 1) continue;
    long qg = (long)blocks[3 * (b + 1)] - blocks[3 * b] - blocks[3 * b + 2];
    long tg = (long)blocks[3 * (b + 1) + 1] - blocks[3 * b + 1] - blocks[3 * b + 2];
    if (qg > 0 || tg > 0) {
      long common = qg > tg ? tg : qg;
      tg -= common; qg -= common;
      for (long g = 0; g < qg; g++, q++) col.push_back(2);
      for (long g = 0; g < tg; g++, t++) col.push_back(3);
      for (long g = 0; g < common; g++) pair_col();
    }
  }
  long nm = 0, nmm = 0, nDrun = 0, nIrun = 0, tdel = 0, tins = 0, sD = 0, mD = 0, lD = 0, sI = 0, mI = 0, lI = 0;
  float value = 0;
  const float coefficient = 3.0f;
  long nr = 0;
  size_t i = 0;
  while (i < col.size()) {                                               // :419-501
    size_t p = i;
    const unsigned char c = col[i];
    while (i < col.size() && col[i] == c) i++;
    const long len = (long)(i - p);
    if (nr < cap) runs[nr] = (uint32_t)(len << 4) | c;
    nr++;
    if (c == 0) { nm += len; value += len; }
    else if (c == 1) { nmm += len; value -= len; }
    else if (c == 3) {                                                   // 'D' :447-470
      tdel += len; nDrun++;
      if (len <= 10) sD++;
      if (len > 10 && len < 50) mD++; else if (len > 50) lD++;
      if (len <= 20) value -= len;
      else if (len <= 10001) { int a = (int)std::floor((len - 1) / 5); value += -coefficient * lut[a] - 1; }
      else if (len <= 100001) value += -1000;
      else value += -2000;
    } else {                                                             // 'I' :472-499
      tins += len; nIrun++;
      if (len <= 10) sI++;
      if (len > 10 && len < 50) mI++; else if (len > 50) lI++;
      if (len <= 20) { value -= len; sI++; }
      else if (len <= 10001) { int a = (int)std::floor((len - 1) / 5); value += -coefficient * lut[a] - 1; }
      else if (len <= 100001) value += -1000;
      else value += -2000;
    }
  }
  const long last = nb - 1;
  long* o = out_counts;
  o[0] = nm; o[1] = nmm; o[2] = nDrun; o[3] = nIrun; o[4] = tdel; o[5] = tins; o[6] = sD; o[7] = mD; o[8] = lD; o[9] = sI; o[10] = mI; o[11] = lI;
  o[12] = blocks[0];                                                     // preClip :522
  o[13] = readLen - blocks[3 * last] - blocks[3 * last + 2];             // sufClip
  o[14] = blocks[0]; o[15] = blocks[3 * last] + blocks[3 * last + 2];
  o[16] = blocks[1]; o[17] = blocks[3 * last + 1] + blocks[3 * last + 2];
  *out_value = value;
  return nr;
}
