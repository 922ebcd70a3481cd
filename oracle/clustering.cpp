// oracle/clustering.cpp -- TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// CPU restatement of CleanMatches (reference: Clustering.h:1840-1906) as every preset runs it
// (opts.ExtractDiagonalFromClean == true): DiagonalSort / AntiDiagonalSort (Sorting.h:50-68,
// 113-136), CleanOffDiagonal (Clustering.h:566-798), AVGfreq (:550-564),
// SecondRoundCleanOffDiagonal (:802-868), Cluster boundaries (:308-322), chromIndex
// (Genome.h:20-32).
// Parity status: PARITY UNPINNED -- Clustering.h needs Genome (-> htslib/kseq.h, absent), so it
// cannot be compiled here; restated from the source text.
#include "oracle_common.h"
#include "../include/lra_hip.h"
#include <algorithm>
#include <cmath>
#include <unordered_map>
#include <vector>

namespace {
struct M { uint32_t q, t; uint64_t key; };

long diag_diff(const M& a, const M& b, int strand) {                    // Clustering.h:503-514
  if (strand == 0) return ((long)a.t - (long)a.q) - ((long)b.t - (long)b.q);
  return (long)(uint32_t)(a.q + a.t) - (long)(uint32_t)(b.q + b.t);   // 32-bit sums, as `long aDiag = a.first.pos + a.second.pos`
}

int header_find(const uint64_t* pos, int npos, uint64_t query) {         // Genome.h:20-32 (pos has nchrom+1 entries)
  if (npos > 0 && query == pos[0]) return 0;
  const uint64_t* it = std::lower_bound(pos, pos + npos, query);
  int i = (int)(it - pos);
  if (it != pos + npos && query == *it) return i;
  return i - 1;
}

void second_round(std::vector<int>& count, int out_counter, std::vector<M>& m, int MinDiagCluster, int CleanMaxDiag,
                  std::vector<char>& orig, int os, int oe, int strand) { // :802-868
  if (MinDiagCluster >= oe - os) return;
  if (MinDiagCluster <= 0) { for (int i = os; i < oe; i++) { orig[i] = 1; count[i] = out_counter; } return; }
  if (oe - os <= 1) return;
  std::vector<char> fw(oe - os, 0), rv(oe - os, 0);
  for (int i = os + 1; i < oe; i++) if (std::labs(diag_diff(m[i], m[i - 1], strand)) < CleanMaxDiag) fw[i - 1 - os] = 1;
  bool prev = false; int ds = 0;
  for (int i = os; i < oe; i++) {
    if (!prev && fw[i - os]) ds = i;
    if (prev && !fw[i - os]) {
      if (i - ds + 1 < MinDiagCluster) { for (int j = ds; j <= i; j++) fw[j - os] = 0; }
      else fw[i - os] = 1;
    }
    prev = fw[i - os];
  }
  for (int i = oe - 2; i >= os; i--) if (std::labs(diag_diff(m[i], m[i + 1], strand)) < CleanMaxDiag) rv[i + 1 - os] = 1;
  prev = false;
  for (int i = oe - 1; i >= os; i--) {
    if (!prev && rv[i - os]) ds = i;
    if (prev && !rv[i - os]) {
      if (ds - i + 1 < MinDiagCluster) { for (int j = i; j <= ds; j++) rv[j - os] = 0; }
      else rv[i - os] = 1;
    }
    prev = rv[i - os];
  }
  for (int i = os; i < oe; i++) {
    if (fw[i - os] && rv[i - os]) { orig[i] = 1; count[i] = out_counter; }
    else orig[i] = 0;
  }
}
}  // namespace

// One strand of one read.  Inputs: the matches (read pos, genome pos, read minimizer key incl. strand
// bit).  Outputs: cleaned sorted matches (out_q/out_t, capacity n) and clusters (capacity n each).
// Returns the number of clusters; *n_clean the number of surviving matches.
extern "C" long oracle_clean_matches(const uint32_t* qpos, const uint32_t* tpos, const uint64_t* qkey, long n, int strand,
                                     const lra_clean_opts* o, const uint64_t* chrom_pos, int n_chrom, uint32_t* out_q,
                                     uint32_t* out_t, long* n_clean, long* c_start, long* c_end, uint32_t* c_qs, uint32_t* c_qe,
                                     uint32_t* c_ts, uint32_t* c_te, int* c_chrom, float* c_freq) {
  *n_clean = 0;
  if (n == 0) return 0;                                                  // :568-570
  std::vector<M> m(n);
  for (long i = 0; i < n; i++) m[i] = {qpos[i], tpos[i], qkey[i]};
  if (strand == 0)                                                       // Sorting.h:34-47
    std::sort(m.begin(), m.end(), [](const M& a, const M& b) {
      long ad = (long)a.q - (long)a.t, bd = (long)b.q - (long)b.t;
      return ad != bd ? ad < bd : a.q < b.q;
    });
  else                                                                   // Sorting.h:74-88 (32-bit sum)
    std::sort(m.begin(), m.end(), [](const M& a, const M& b) {
      uint32_t ad = a.q + a.t, bd = b.q + b.t;
      return ad != bd ? ad < bd : a.q < b.q;
    });
  std::vector<float> freq(n, 1.0f);
  std::vector<char> onDiag(n, 0);
  if (n > 1 && std::labs(diag_diff(m[0], m[1], strand)) < o->cleanMaxDiag) onDiag[0] = 1;            // :573-576
  for (long i = 1; i < n; i++) if (std::labs(diag_diff(m[i], m[i - 1], strand)) < o->cleanMaxDiag) onDiag[i - 1] = 1;   // :578-584
  bool prev = false, startSet = false;
  int diagStart = 0, largest = 0;
  for (long i = 0; i < n; i++) {                                         // :589-598
    if (!prev && onDiag[i]) { diagStart = (int)i; startSet = true; }
    if (prev && !onDiag[i]) largest = std::max(largest, (int)i - diagStart + 1);
    prev = onDiag[i];
  }
  if (!startSet) return 0;                                               // :600-603
  largest = std::max(largest, (int)n - diagStart);
  int minDiagCluster = (int)std::floor(largest / 10);                    // :608-609
  if (minDiagCluster >= o->minDiagCluster) minDiagCluster = o->minDiagCluster;
  std::vector<int> count(n, -1);
  std::vector<char> second(n, 0);
  int counter = 0;
  prev = false;
  if (minDiagCluster >= 0) {
    for (long i = 0; i < n; i++) {                                       // :620-722
      if (!prev && onDiag[i]) diagStart = (int)i;
      if (prev && !onDiag[i]) {
        const int len = (int)i - diagStart + 1;
        if (len < minDiagCluster) {
          for (int j = diagStart; j <= i; j++) second[j] = 0;
        } else {
          std::unordered_map<uint64_t, int> mc;                          // AVGfreq :550-564
          for (int r = diagStart; r <= i; r++) mc[m[r].key]++;
          const float avgfreq = (float)(len) / mc.size();
          for (int j = diagStart; j <= i; j++) freq[j] = avgfreq;
          int MinDiagCluster = 0;
          const int cc = o->cleanClustersize;
          if (o->bypassClustering) {                                     // :635-657
            if (avgfreq >= 3.0f && len < 10) { for (int j = diagStart; j <= i; j++) second[j] = 0; }
            else if (avgfreq >= 2.0f && len >= cc) {
              MinDiagCluster = o->SecondCleanMinDiagCluster + std::floor((avgfreq - 1.5f) / 1.0f) * o->punish_anchorfreq +
                               std::floor((len - cc) / cc) * o->anchorPerlength;
              second_round(count, counter, m, MinDiagCluster, o->SecondCleanMaxDiag, second, diagStart, (int)i + 1, strand);
            } else if (avgfreq >= 1.5f && len >= cc) {
              MinDiagCluster = o->SecondCleanMinDiagCluster + std::floor((avgfreq - 1.5f) / 1.5f) * o->punish_anchorfreq +
                               std::floor((len - cc) / cc) * o->anchorPerlength;
              second_round(count, counter, m, MinDiagCluster, o->SecondCleanMaxDiag, second, diagStart, (int)i + 1, strand);
            } else { for (int j = diagStart; j <= i; j++) { second[j] = 1; count[j] = counter; } }
          } else {                                                       // :659-693
            if (avgfreq >= 3.0f && len < 10) { for (int j = diagStart; j <= i; j++) second[j] = 0; }
            else if (avgfreq >= 4.0f && len >= cc) {
              MinDiagCluster = o->SecondCleanMinDiagCluster + std::floor((avgfreq - 1.5f) / 1.0f) * o->punish_anchorfreq +
                               std::floor((len - cc) / cc) * o->anchorPerlength;
              second_round(count, counter, m, MinDiagCluster, o->SecondCleanMaxDiag, second, diagStart, (int)i + 1, strand);
            } else if (avgfreq >= 1.5f && len >= cc) {
              MinDiagCluster = o->SecondCleanMinDiagCluster + std::floor((avgfreq - 1.5f) / 1.5f) * o->punish_anchorfreq +
                               std::floor((len - cc) / cc) * o->anchorPerlength;
              second_round(count, counter, m, MinDiagCluster, o->SecondCleanMaxDiag, second, diagStart, (int)i + 1, strand);
            } else if (avgfreq > 1.0f && len >= cc) {
              MinDiagCluster = o->SecondCleanMinDiagCluster - (5 - std::floor((avgfreq - 1.0f) / 0.1f)) * (o->punish_anchorfreq / 2) +
                               std::floor((len - cc) / cc) * (o->anchorPerlength / 2);
              second_round(count, counter, m, MinDiagCluster, o->SecondCleanMaxDiag, second, diagStart, (int)i + 1, strand);
            } else if (avgfreq > 1.0f) {
              MinDiagCluster = o->SecondCleanMinDiagCluster - (5 - std::floor((avgfreq - 1.0f) / 0.1f)) * (o->punish_anchorfreq / 2) -
                               std::floor((cc - (int)i + diagStart - 1) / 15) * (o->anchorPerlength / 2);
              second_round(count, counter, m, MinDiagCluster, o->SecondCleanMaxDiag, second, diagStart, (int)i + 1, strand);
            } else { for (int j = diagStart; j <= i; j++) { second[j] = 1; count[j] = counter; } }
          }
        }
        counter++;
      }
      prev = onDiag[i];
    }
  }
  long c = 0;
  for (long i = 0; i < n; i++)                                           // :728-738
    if (second[i]) { m[c] = m[i]; freq[c] = freq[i]; count[c] = count[i]; c++; }
  *n_clean = c;
  for (long i = 0; i < c; i++) { out_q[i] = m[i].q; out_t[i] = m[i].t; }
  // clusters (:740-797): maximal stretches of equal `count`
  long ncl = 0;
  auto emit = [&](long s, long e) {
    uint32_t qS = m[s].q, qE = m[s].q + o->globalK, tS = m[s].t, tE = m[s].t + o->globalK;
    for (long b = s; b < e; b++) {
      qS = std::min(qS, m[b].q); qE = std::max(qE, m[b].q + (uint32_t)o->globalK);
      tS = std::min(tS, m[b].t); tE = std::max(tE, m[b].t + (uint32_t)o->globalK);
    }
    c_start[ncl] = s; c_end[ncl] = e; c_qs[ncl] = qS; c_qe[ncl] = qE; c_ts[ncl] = tS; c_te[ncl] = tE;
    c_freq[ncl] = freq[s];
    c_chrom[ncl] = header_find(chrom_pos, n_chrom + 1, tS);              // :760-761 (only set when bypassClustering; reported always)
    ncl++;
  };
  long count_s = 0, cc2 = 1;
  while (cc2 < c) {
    if (count[cc2] == count[cc2 - 1]) { cc2++; continue; }
    emit(count_s, cc2);
    count_s = cc2;
    cc2++;
  }
  if (cc2 == c && count_s < cc2) emit(count_s, cc2);
  return ncl;
}
