// oracle/fine_clusters.cpp -- TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// CPU restatement of the clustering of the high-accuracy path (MapRead_highacc, Map_highacc.h:41-42), both strands of one read:
//   MatchesToFineClusters          Clustering.h:1555-1680   (DiagonalSort / AntiDiagonalSort + CleanOffDiagonal = oracle_clean_matches,
//                                                            CartesianSort of every rough cluster Sorting.h:157)
//   SplitRoughClustersWithGaps     Clustering.h:1358-1432   (CloseToPreviousCluster :1333, MergeTwoClusters :1351, minGapDifference :532)
//   StoreFineClusters              Clustering.h:892-1331    (DiagonalDifference :503, Cluster::SetClusterBoundariesFromMatches :308,
//                                                            Cluster::CHROMIndex :327)
// Both strands share the reference's `clusters` vector: the `pop_back` pairs of StoreFineClusters (:1301-1306) may look at -- and drop -- the
// cluster pushed before the current one, even one of the other strand.  *status != 0 when the reference would read `clusters.back()` of an
// empty vector there (undefined behaviour).
// Parity status: PARITY UNPINNED -- Clustering.h needs Genome.h (htslib); restated from the source text.
#include "oracle_common.h"
#include "../include/lra_hip.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <list>
#include <vector>

extern "C" long oracle_clean_matches(const uint32_t* qpos, const uint32_t* tpos, const uint64_t* qkey, long n, int strand, const lra_clean_opts* o, const uint64_t* chrom_pos,
                                     int n_chrom, uint32_t* out_q, uint32_t* out_t, long* n_clean, long* c_start, long* c_end, uint32_t* c_qs, uint32_t* c_qe, uint32_t* c_ts,
                                     uint32_t* c_te, int* c_chrom, float* c_freq);

struct oracle_fine_opts { int globalK, RoughClustermaxGap, maxDiag, maxGap, minClusterSize, minUniqueStretchNum, minUniqueStretchDist; };

namespace {

struct Mt { uint32_t q, t; };
struct FCluster { std::vector<Mt> matches; uint32_t qStart = 0, qEnd = 0, tStart = 0, tEnd = 0; int strand = 0, chromIndex = 0; float anchorfreq = 0; };
struct Split { int start, end; uint32_t qStart, qEnd, tStart, tEnd; int strand; float anchorfreq; std::vector<int> idx; };

int header_find(const uint64_t* pos, int npos, uint64_t query) {          // Genome.h:20-32
  if (npos > 0 && query == pos[0]) return 0;
  const uint64_t* it = std::lower_bound(pos, pos + npos, query);
  const int i = (int)(it - pos);
  if (it != pos + npos && query == *it) return i;
  return i - 1;
}
long diag_diff(const Mt& a, const Mt& b, int strand) {                      // DiagonalDifference :503-514
  if (strand == 0) return ((long)a.t - (long)a.q) - ((long)b.t - (long)b.q);
  return (long)(uint32_t)(a.q + a.t) - (long)(uint32_t)(b.q + b.t);        // a.first.pos + a.second.pos is a 32-bit sum
}
long min_gap(const Mt& a, const Mt& b) { return std::min(std::labs((long)b.q - (long)a.q), std::labs((long)b.t - (long)a.t)); }   // :532-536
void set_bounds(FCluster& c, int K) {                                       // :308-322
  c.qStart = c.matches[0].q; c.qEnd = c.qStart + K; c.tStart = c.matches[0].t; c.tEnd = c.tStart + K;
  for (size_t i = 1; i < c.matches.size(); i++) {
    c.tEnd = std::max(c.tEnd, c.matches[i].t + (uint32_t)K); c.tStart = std::min(c.tStart, c.matches[i].t);
    c.qEnd = std::max(c.qEnd, c.matches[i].q + (uint32_t)K); c.qStart = std::min(c.qStart, c.matches[i].q);
  }
}
bool chrom_index(FCluster& c, const uint64_t* pos, int npos) {              // Cluster::CHROMIndex :327-337
  if (c.matches.empty()) return true;
  const int a = header_find(pos, npos, (uint64_t)c.tStart + 1), b = header_find(pos, npos, c.tEnd);
  if (a != b) return true;
  c.chromIndex = a;
  return false;
}

bool close_to_previous(const Split& a, uint32_t qS, uint32_t tS, uint32_t tE, const oracle_fine_opts& o) {   // :1333-1343
  const long aDiff = std::labs((long)qS - (long)a.qEnd);
  const long bDiff = a.strand == 0 ? std::labs((long)tS - (long)a.tEnd) : std::labs((long)a.tStart - (long)tE);
  long aDiag, bDiag;
  if (a.strand == 0) { aDiag = (long)a.tEnd - (long)a.qEnd; bDiag = (long)tS - (long)qS; }
  else { aDiag = (long)a.qEnd + (long)a.tStart; bDiag = (long)qS + (long)tE; }
  return std::min(aDiff, bDiff) <= o.RoughClustermaxGap && std::labs(aDiag - bDiag) < o.maxDiag;
}

// SplitRoughClustersWithGaps :1358-1432 on matches[start, end) (already in Cartesian order)
void split_rough(const std::vector<Mt>& m, int start, int end, int strand, float anchorfreq, uint32_t bq0, uint32_t bq1, uint32_t bt0, uint32_t bt1, std::vector<Split>& split,
                 const oracle_fine_opts& o) {
  if (end - start == 0) return;
  if (anchorfreq >= 10.0f) {
    Split s{start, end, bq0, bq1, bt0, bt1, strand, anchorfreq, {}};
    for (int q = start; q < end; q++) s.idx.push_back(q);
    split.push_back(std::move(s));
    return;
  }
  const size_t cur_s = split.size();
  int split_cs = start;
  uint32_t sq0 = m[split_cs].q, st0 = m[split_cs].t, sq1 = sq0 + o.globalK, st1 = st0 + o.globalK;
  auto merge_or_push = [&](int e, bool lastCall) {
    // (the chromIndex tests of :1388 / :1392 compare values that are all -1 on this path; the final call :1420 has no such test)
    (void)lastCall;
    if (split.size() > cur_s && close_to_previous(split.back(), sq0, st0, st1, o)) {                       // MergeTwoClusters :1351-1355
      Split& a = split.back();
      a.qStart = std::min(a.qStart, sq0); a.qEnd = std::max(a.qEnd, sq1); a.tStart = std::min(a.tStart, st0); a.tEnd = std::max(a.tEnd, st1);
      for (int q = split_cs; q < e; q++) a.idx.push_back(q);
      a.end = e;
    } else {
      Split s{split_cs, e, sq0, sq1, st0, st1, strand, anchorfreq, {}};
      for (int q = split_cs; q < e; q++) s.idx.push_back(q);
      split.push_back(std::move(s));
    }
  };
  for (int i = start + 1; i < end; i++) {
    const long gap = min_gap(m[i], m[i - 1]);
    if (gap > o.RoughClustermaxGap) {
      if (i - split_cs >= o.minClusterSize) merge_or_push(i, false);
      sq0 = m[i].q; st0 = m[i].t; sq1 = sq0 + o.globalK; st1 = st0 + o.globalK; split_cs = i;
    } else {
      sq0 = std::min(sq0, m[i].q); st0 = std::min(st0, m[i].t); sq1 = std::max(sq1, m[i].q + (uint32_t)o.globalK); st1 = std::max(st1, m[i].t + (uint32_t)o.globalK);
    }
  }
  if (end - split_cs >= o.minClusterSize) merge_or_push(end, true);
}

// StoreFineClusters :892-1331 for one split cluster
void store_fine(const std::vector<Mt>& m, const Split& sp, std::vector<FCluster>& clusters, const oracle_fine_opts& o, int strand, const uint64_t* pos, int npos, int& ub) {
  const std::vector<int>& smi = sp.idx;
  const float anchorfreq = sp.anchorfreq;
  const int ri = header_find(pos, npos, sp.tStart);                       // :1611 genome.header.Find(split.tStart)
  const int K = o.globalK;
  if (smi.size() == 1) return;
  auto M = [&](int i) -> const Mt& { return m[smi[i]]; };
  if (std::fabs(anchorfreq - 1.0f) <= 0.005) {                             // :900-942
    clusters.push_back(FCluster()); clusters.back().strand = strand;
    for (size_t i = 0; i < smi.size(); i++) clusters.back().matches.push_back(M((int)i));
    set_bounds(clusters.back(), K);
    clusters.back().chromIndex = ri; clusters.back().anchorfreq = 1.0f;
    if (chrom_index(clusters.back(), pos, npos)) clusters.pop_back();
    return;
  }
  std::vector<int> match_num, pos_start;                                   // :948-965
  int oc = 1, us = 0;
  for (size_t i = 1; i < smi.size(); i++) {
    if (M((int)i).q == M((int)i - 1).q) oc++;
    else { match_num.push_back(oc); pos_start.push_back(us); us = (int)i; oc = 1; }
    if (i == smi.size() - 1) { match_num.push_back(oc); pos_start.push_back(us); }
  }
  int u_start = 0, u_end = 0, u_maxstart = 0, u_maxend = 0, max_pos = 0;
  std::vector<int> Start, End;
  if (match_num.size() == 1) { u_maxstart = 0; u_maxend = 1; Start.push_back(0); End.push_back(1); }          // :972-978
  else {
    int k = 0;
    const int nm = (int)match_num.size();
    while (k < nm - 1) {                                                   // :980-1005
      while (k < nm - 1 && match_num[k] != 1) k++;
      u_start = k; u_end = k + 1;
      while (k < nm - 1 && match_num[k + 1] == match_num[k] && std::labs(diag_diff(M(pos_start[k + 1]), M(pos_start[k]), strand)) < o.maxDiag &&
             min_gap(M(pos_start[k + 1]), M(pos_start[k])) <= o.maxGap) { u_end = k + 2; k++; }
      Start.push_back(u_start); End.push_back(u_end);
      k++;
      if ((u_maxstart == 0 && u_maxend == 0) || (u_maxend - u_maxstart < u_end - u_start)) { u_maxstart = u_start; u_maxend = u_end; max_pos = (int)Start.size() - 1; }
    }
  }
  if (u_maxstart == 0 && u_maxend == 0) return;                            // :1007-1009 no unique stretch
  int c_s = pos_start[u_maxstart], c_e = pos_start[u_maxend - 1] + 1;
  if (!(c_e - c_s >= o.minUniqueStretchNum && (long)M(c_e - 1).q + K - (long)M(c_s).q >= o.minUniqueStretchDist)) return;   // :1044-1045, :1328
  clusters.push_back(FCluster()); clusters.back().strand = strand;
  std::vector<char> AddOrNot(Start.size(), 0);
  if (c_e - c_s == (int)smi.size()) {                                      // :1051-1057
    for (int i = c_s; i < c_e; i++) clusters.back().matches.push_back(M(i));
    clusters.back().anchorfreq = anchorfreq;
    AddOrNot[0] = 1;
  } else {
    std::list<int> StretchOfOne;
    int prev_anchor = c_s;
    auto near_ = [&](int i_m, int pa) {
      return (std::labs(diag_diff(M(i_m), M(pa), strand)) <= o.maxDiag && min_gap(M(i_m), M(pa)) <= o.maxGap) || min_gap(M(i_m), M(pa)) <= o.maxGap / 2;
    };
    if (max_pos >= 0) {                                                    // :1063-1080 towards the start
      StretchOfOne.push_back(max_pos); AddOrNot[max_pos] = 1;
      for (int i = max_pos - 1; i >= 0; i--) {
        const int i_m = pos_start[End[i] - 1];
        if (near_(i_m, prev_anchor)) { StretchOfOne.push_back(i); AddOrNot[i] = 1; prev_anchor = pos_start[Start[i]]; }
      }
    }
    prev_anchor = c_e - 1;                                                 // :1081-1099 towards the end
    if (max_pos < (int)Start.size()) {
      for (int i = max_pos + 1; i < (int)Start.size(); i++) {
        const int i_m = pos_start[Start[i]];
        if (near_(i_m, prev_anchor)) { StretchOfOne.push_front(i); AddOrNot[i] = 1; prev_anchor = pos_start[End[i] - 1]; }
      }
    }
    int prev_stretch = -1, p_s = 0, p_e = 0;
    for (auto it = StretchOfOne.rbegin(); it != StretchOfOne.rend(); ++it) {   // :1103-1163 (reverse = ascending stretch index)
      std::vector<int> Cluster_index;
      c_s = pos_start[Start[*it]]; c_e = pos_start[End[*it] - 1] + 1;
      if (it == StretchOfOne.rbegin()) { p_s = *it == 0 ? 0 : pos_start[End[*it - 1]]; p_e = pos_start[Start[*it]]; }
      else { p_s = pos_start[End[prev_stretch]]; p_e = pos_start[Start[*it]]; }
      prev_stretch = *it;
      int prev_match = c_s;
      for (int si = p_e - 1; si >= p_s; si--)
        if (std::labs(diag_diff(M(si), M(prev_match), strand)) < o.maxDiag) { Cluster_index.push_back(si); prev_match = si; }
      for (auto ci = Cluster_index.rbegin(); ci != Cluster_index.rend(); ++ci) clusters.back().matches.push_back(M(*ci));
      for (int si = c_s; si < c_e; si++) clusters.back().matches.push_back(M(si));
      if (std::next(it) == StretchOfOne.rend()) {                            // the last stretch: the matches behind it
        p_s = pos_start[End[*it] - 1] + 1;
        p_e = (*it == (int)AddOrNot.size() - 1) ? (int)smi.size() : pos_start[Start[*it + 1]];
        prev_match = c_e - 1;
        for (int si = p_s; si < p_e; si++)
          if (std::labs(diag_diff(M(si), M(prev_match), strand)) < o.maxDiag) { clusters.back().matches.push_back(M(si)); prev_match = si; }
      }
    }
    clusters.back().anchorfreq = anchorfreq;
  }
  set_bounds(clusters.back(), K);                                          // :1280
  clusters.back().chromIndex = ri;
  if (!clusters.empty()) {                                                 // :1282-1295
    FCluster& b = clusters.back();
    if (chrom_index(b, pos, npos)) clusters.pop_back();
    else if ((long)b.matches.size() <= o.minClusterSize) clusters.pop_back();
    else if (b.qEnd == b.qStart) clusters.pop_back();
    else if ((long)b.tEnd - (long)b.tStart >= 5 * ((long)b.qEnd - (long)b.qStart)) clusters.pop_back();
  }
  for (size_t ar = 0; ar < AddOrNot.size(); ar++) {                         // :1297-1323 the long stretches that were left out
    if (!AddOrNot[ar] && End[ar] - Start[ar] >= 15) {
      clusters.push_back(FCluster()); clusters.back().strand = strand;
      for (int i = pos_start[Start[ar]]; i < pos_start[End[ar] - 1] + 1; i++) clusters.back().matches.push_back(M(i));
      set_bounds(clusters.back(), K);
      clusters.back().chromIndex = ri; clusters.back().anchorfreq = anchorfreq;
      if (chrom_index(clusters.back(), pos, npos)) clusters.pop_back();
      if (clusters.empty()) { ub = 1; return; }                            // :1305 reads clusters.back() of an empty vector
      const FCluster& b = clusters.back();
      if ((long)b.qEnd - (long)b.qStart == 0) { ub = 1; return; }
      if (((long)b.tEnd - (long)b.tStart) / ((long)b.qEnd - (long)b.qStart) >= 5) clusters.pop_back();
    }
  }
}

}  // namespace

// Both strands of one read: matches (read pos, genome pos, read minimizer key) with the forward-strand ones first (n_forward of them).
// Out: fine clusters in the reference's order -- match lists (CSR c_off over out_q / out_t), box {qStart, qEnd, tStart, tEnd}, strand, chromIndex,
// anchorfreq.  Returns the number of clusters (-1 if a capacity is too small); *n_matches = total matches; *status != 0: undefined behaviour.
extern "C" long oracle_matches_to_fine_clusters(const uint32_t* qpos, const uint32_t* tpos, const uint64_t* qkey, long n, long n_forward, const lra_clean_opts* co,
                                                const oracle_fine_opts* fo, const uint64_t* chrom_pos, int n_chrom, long capM, long capC, uint32_t* out_q, uint32_t* out_t,
                                                long* c_off, uint32_t* box, int* strand_out, int* chrom, float* freq, long* n_matches, int* status) {
  std::vector<FCluster> clusters;
  int ub = 0;
  for (int strand = 0; strand < 2 && !ub; strand++) {
    const long a = strand == 0 ? 0 : n_forward, b = strand == 0 ? n_forward : n, ns = b - a;
    if (ns <= 0) continue;
    std::vector<uint32_t> cq((size_t)ns), ct((size_t)ns), bqs((size_t)ns), bqe((size_t)ns), bts((size_t)ns), bte((size_t)ns);
    std::vector<long> cs((size_t)ns), ce((size_t)ns); std::vector<int> cch((size_t)ns); std::vector<float> cfr((size_t)ns);
    long nclean = 0;
    const long nr = oracle_clean_matches(qpos + a, tpos + a, qkey + a, ns, strand, co, chrom_pos, n_chrom, cq.data(), ct.data(), &nclean, cs.data(), ce.data(), bqs.data(), bqe.data(),
                                         bts.data(), bte.data(), cch.data(), cfr.data());
    std::vector<Mt> m((size_t)nclean);
    for (long i = 0; i < nclean; i++) m[i] = {cq[i], ct[i]};
    std::vector<Split> split;
    for (long c = 0; c < nr; c++) {                                        // :1574-1577 / :1631-1634
      std::sort(m.begin() + cs[c], m.begin() + ce[c], [](const Mt& x, const Mt& y) { return x.q != y.q ? x.q < y.q : x.t < y.t; });   // CartesianSort
      split_rough(m, (int)cs[c], (int)ce[c], strand, cfr[c], bqs[c], bqe[c], bts[c], bte[c], split, *fo);
    }
    for (size_t c = 0; c < split.size() && !ub; c++) store_fine(m, split[c], clusters, *fo, strand, chrom_pos, n_chrom + 1, ub);
  }
  *status = ub;
  long tot = 0;
  if ((long)clusters.size() > capC) return -1;
  for (size_t c = 0; c < clusters.size(); c++) {
    const FCluster& f = clusters[c];
    c_off[c] = tot;
    if (tot + (long)f.matches.size() > capM) return -1;
    for (const Mt& x : f.matches) { out_q[tot] = x.q; out_t[tot] = x.t; tot++; }
    box[4 * c] = f.qStart; box[4 * c + 1] = f.qEnd; box[4 * c + 2] = f.tStart; box[4 * c + 3] = f.tEnd;
    strand_out[c] = f.strand; chrom[c] = f.chromIndex; freq[c] = f.anchorfreq;
  }
  c_off[clusters.size()] = tot;
  *n_matches = tot;
  return (long)clusters.size();
}
