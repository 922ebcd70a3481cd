// oracle/merge_extend.cpp -- TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// CPU restatement of what MapRead_lowacc does with the refined clusters of one chain before the second sparse DP (Map_lowacc.h:440-476):
//   MergeChain                                   ChainRefine.h:767-802
//   LinearExtend (pair version, skipsorting = 0) LinearExtend.h:658-716  (DiagonalSort Sorting.h:36-62; oracle_linear_extend)
//   DecideCoordinates                            LinearExtend.h:105-127
//   TrimOverlappedAnchors                        LinearExtend.h:574-649  (LongAnchors :11-47)
// Parity status: PARITY UNPINNED -- these headers need Genome.h / Clustering.h (htslib); restated from the source text.
#include "oracle_common.h"
#include <algorithm>
#include <utility>
#include <vector>

extern "C" long oracle_linear_extend(const uint32_t* q, const uint32_t* t, long n, int strand, int K, const char* read, uint32_t readLen,
                                     const char* chrom, uint32_t chromLen, uint32_t* eq, uint32_t* et, int* elen, uint32_t* box);

// One chain: nsp refined clusters (matches CSR, box, strand, chromIndex; t chromosome-relative), the forward read, the genome with its
// chromosome table, K = smallOpts.globalK.  Out: merged + extended clusters: groupMember[nG+1] (ranges of refined clusters), anchors CSR
// anchorOff[nG+1] / aq / at / alen (capacity cap), gbox[4 nG], gstrand, gchrom.  Returns nG, or -1 when cap is too small.
extern "C" int oracle_merge_extend(int nsp, const int* matchOff, const uint32_t* mq, const uint32_t* mt, const uint32_t* box, const uint8_t* strand,
                                   const int* chrom, const char* read, uint32_t readLen, const char* genome, const uint64_t* chromPos, int K, long cap,
                                   int* groupMember, int* anchorOff, uint32_t* aq, uint32_t* at, int* alen, uint32_t* gbox, uint8_t* gstrand, int* gchrom) {
  if (nsp == 0) { groupMember[0] = 0; anchorOff[0] = 0; return 0; }
  // MergeChain :767-802 (sp[t] = t: spcluster.sptc[t] = spchain[t].clusterIndex = t, Map_lowacc.h:376)
  std::vector<std::pair<int, int>> groups;                               // [first, last] runs of consecutive clusters
  int g0 = 0;
  for (int t = 1; t < nsp; t++) {
    const int cur = t, prev = t - 1;
    int qdist = 9999, tdist = 9999;
    if (chrom[prev] == chrom[cur] && strand[prev] == strand[cur]) {
      const uint32_t pqs = box[4 * prev], pts = box[4 * prev + 2], pte = box[4 * prev + 3];
      const uint32_t cqe = box[4 * cur + 1], cts = box[4 * cur + 2], cte = box[4 * cur + 3];
      qdist = (pqs > cqe) ? (int)(pqs - cqe) : 0;
      if (strand[prev] == 0) tdist = (pts >= cte) ? (int)(pts - cte) : 9999;
      else tdist = (pte <= cts) ? (int)(cts - pte) : 9999;
    }
    if (qdist <= 500 && tdist <= 500) continue;
    groups.push_back(std::make_pair(g0, t - 1));
    g0 = t;
  }
  groups.push_back(std::make_pair(g0, nsp - 1));
  long total = 0;
  anchorOff[0] = 0; groupMember[0] = 0;
  for (size_t r = 0; r < groups.size(); r++) {
    const long begin = total;
    int st = 0, ci = 0;
    for (int cI = groups[r].first; cI <= groups[r].second; cI++) {       // Map_lowacc.h:458-466
      st = strand[cI]; ci = chrom[cI];
      const int n = matchOff[cI + 1] - matchOff[cI];
      std::vector<std::pair<uint32_t, uint32_t>> P(n);
      for (int i = 0; i < n; i++) P[i] = std::make_pair(mq[matchOff[cI] + i], mt[matchOff[cI] + i]);
      std::sort(P.begin(), P.end(), [](const std::pair<uint32_t, uint32_t>& a, const std::pair<uint32_t, uint32_t>& b) {   // DiagonalSortOp
        const long aDiag = (long)a.first - (long)a.second, bDiag = (long)b.first - (long)b.second;
        if (aDiag != bDiag) return aDiag < bDiag;
        return a.first < b.first;
      });
      if (n == 0) continue;                                              // LinearExtend on an empty list appends nothing
      std::vector<uint32_t> q(n), t(n), eq(n), et(n); std::vector<int> el(n);
      for (int i = 0; i < n; i++) { q[i] = P[i].first; t[i] = P[i].second; }
      uint32_t bx[4];
      const long ne = oracle_linear_extend(q.data(), t.data(), n, st, K, read, readLen, genome + chromPos[ci], (uint32_t)(chromPos[ci + 1] - chromPos[ci]),
                                           eq.data(), et.data(), el.data(), bx);
      if (total + ne > cap) return -1;
      for (long i = 0; i < ne; i++) { aq[total] = eq[i]; at[total] = et[i]; alen[total] = el[i]; total++; }
    }
    // DecideCoordinates :105-127
    uint32_t* gb = gbox + 4 * r;
    gb[0] = gb[1] = gb[2] = gb[3] = 0;
    if (total > begin) {
      uint32_t qS = aq[begin], qE = qS + alen[begin], tS = at[begin], tE = tS + alen[begin];
      for (long i = begin + 1; i < total; i++) {
        qS = std::min(qS, aq[i]); qE = std::max(qE, aq[i] + (uint32_t)alen[i]); tS = std::min(tS, at[i]); tE = std::max(tE, at[i] + (uint32_t)alen[i]);
      }
      gb[0] = qS; gb[1] = qE; gb[2] = tS; gb[3] = tE;
      gstrand[r] = (uint8_t)st; gchrom[r] = ci;
    } else { gstrand[r] = 0; gchrom[r] = 0; }                            // a default-constructed Cluster
    anchorOff[r + 1] = (int)total; groupMember[r + 1] = groups[r].second + 1;
    // TrimOverlappedAnchors :574-649 (start = 0) on this cluster
    {
      const int S = gstrand[r];
      uint32_t* Q = aq + begin; uint32_t* T = at + begin; int* L = alen + begin;
      std::vector<int> idx;
      for (long i = 0; i < total - begin; i++) if (L[i] >= 40) idx.push_back((int)i);
      std::sort(idx.begin(), idx.end(), [&](int i, int j) {              // LongAnchors::operator() :26-43
        if (S == 0) { if (Q[i] != Q[j]) return Q[i] < Q[j]; return T[i] < T[j]; }
        if (Q[i] + L[i] != Q[j] + L[j]) return Q[i] + L[i] > Q[j] + L[j];
        return T[i] < T[j];
      });
      for (size_t ln = 1; ln < idx.size(); ln++) {
        const int prev = idx[ln - 1], cur = idx[ln];
        int overlap_r = 0, overlap_g = 0;
        if (S == 0) {
          if (Q[cur] < Q[prev] + L[prev] && Q[cur] >= Q[prev] + L[prev] - 30) overlap_r = (int)(Q[prev] + L[prev] - Q[cur]);
        } else {
          if (Q[cur] + L[cur] > Q[prev] && Q[cur] + L[cur] <= Q[prev] + 30) overlap_r = (int)(Q[cur] + L[cur] - Q[prev]);
        }
        if (T[cur] < T[prev] + L[prev] && T[cur] >= T[prev] + L[prev] - 30) overlap_g = (int)(T[prev] + L[prev] - T[cur]);
        if (overlap_r > 0 || overlap_g > 0) {
          const int overlap = std::max(overlap_r, overlap_g);
          if (S == 1) Q[prev] += overlap + 1;
          L[prev] -= overlap + 1;
        }
      }
    }
  }
  return (int)groups.size();
}

// TrimOverlappedAnchors(vector<Cluster>&, start) LinearExtend.h:574-649 on one extended cluster: read positions / lengths modified in place
extern "C" void oracle_trim_overlapped_anchors(int n, uint32_t* Q, const uint32_t* T, int* L, int S) {
  std::vector<int> idx;
  for (int i = 0; i < n; i++) if (L[i] >= 40) idx.push_back(i);
  std::sort(idx.begin(), idx.end(), [&](int i, int j) {                  // LongAnchors::operator() :26-43
    if (S == 0) { if (Q[i] != Q[j]) return Q[i] < Q[j]; return T[i] < T[j]; }
    if (Q[i] + L[i] != Q[j] + L[j]) return Q[i] + L[i] > Q[j] + L[j];
    return T[i] < T[j];
  });
  for (size_t ln = 1; ln < idx.size(); ln++) {
    const int prev = idx[ln - 1], cur = idx[ln];
    int overlap_r = 0, overlap_g = 0;
    if (S == 0) {
      if (Q[cur] < Q[prev] + L[prev] && Q[cur] >= Q[prev] + L[prev] - 30) overlap_r = (int)(Q[prev] + L[prev] - Q[cur]);
    } else {
      if (Q[cur] + L[cur] > Q[prev] && Q[cur] + L[cur] <= Q[prev] + 30) overlap_r = (int)(Q[cur] + L[cur] - Q[prev]);
    }
    if (T[cur] < T[prev] + L[prev] && T[cur] >= T[prev] + L[prev] - 30) overlap_g = (int)(T[prev] + L[prev] - T[cur]);
    if (overlap_r > 0 || overlap_g > 0) {
      const int overlap = std::max(overlap_r, overlap_g);
      if (S == 1) Q[prev] += overlap + 1;
      L[prev] -= overlap + 1;
    }
  }
}

// TrimOverlappedAnchors(GenomePairs&, vector<int>&) LinearExtend.h:722-780: one list, lengths modified in place
extern "C" void oracle_trim_anchor_pairs(int n, const uint32_t* Q, const uint32_t* T, int* L) {
  std::vector<int> idx;
  for (int i = 0; i < n; i++) if (L[i] >= 50) idx.push_back(i);
  std::sort(idx.begin(), idx.end(), [&](int i, int j) { if (Q[i] != Q[j]) return Q[i] < Q[j]; return T[i] < T[j]; });   // LongAnchors, strand 0
  for (size_t ln = 1; ln < idx.size(); ln++) {
    const int prev = idx[ln - 1], cur = idx[ln];
    int overlap_r = 0, overlap_g = 0;
    if (Q[cur] < Q[prev] + L[prev] && Q[cur] >= Q[prev] + L[prev] - 30) overlap_r = (int)(Q[prev] + L[prev] - Q[cur]);
    if (T[cur] < T[prev] + L[prev] && T[cur] >= T[prev] + L[prev] - 30) overlap_g = (int)(T[prev] + L[prev] - T[cur]);
    if (overlap_r > 0 || overlap_g > 0) L[prev] -= std::max(overlap_r, overlap_g) + 1;
  }
}

// MergeMatchesSameDiag (LinearExtend.h:795-829; Map_highacc.h:642): consecutive anchors of one extended cluster that lie on the same
// diagonal, carry no overlap flag, follow each other on the read and are at most merge_dist apart fall into one Cluster_SameDiag entry.
// GetDiag (Clustering.h:886-889), GapDifference (:539-542).  Returns the number of entries (start / end = anchor index ranges), -1 for an
// empty cluster (the reference reads matches[0] of it).
extern "C" int oracle_merge_same_diag(int n, const uint32_t* Q, const uint32_t* T, const int* L, const uint8_t* overlap, int strand, int merge_dist,
                                      int* start, int* end) {
  if (n <= 0) return -1;
  auto diag = [&](int i) -> long { return strand == 0 ? (long)T[i] - (long)Q[i] : (long)Q[i] + (long)T[i] + L[i]; };
  int ng = 0;
  start[ng] = 0; end[ng] = 1; ng++;
  long prev_diag = diag(0);
  uint32_t prev_qEnd = Q[0] + (uint32_t)L[0];
  for (int q = 1; q < n; q++) {
    const long cur_diag = diag(q);
    const long gap = labs((long)Q[q] - ((long)Q[q - 1] + L[q - 1]));
    if (overlap[q - 1] == 0 && overlap[q] == 0 && prev_diag == cur_diag && prev_qEnd < Q[q] && gap <= merge_dist) end[ng - 1] = q + 1;
    else { start[ng] = q; end[ng] = q + 1; ng++; }
    prev_qEnd = Q[q] + (uint32_t)L[q];
    prev_diag = cur_diag;
  }
  return ng;
}

// SwitchToOriginalAnchors (LocalRefineAlignment.h:187-199; called at :576): a chain over Cluster_SameDiag entries -> the chain over the
// original anchors of the extended clusters: entry k of cluster c becomes its anchors end[k]-1 .. start[k] (descending), ClusterIndex = coarse.
// chain_cluster[i] / chain_entry[i]: prev.ClusterNum(i) / prev.chain[i]; group_off / start / end: the MergeMatchesSameDiag result.
extern "C" long oracle_switch_to_original_anchors(int n, const int* chain_cluster, const uint32_t* chain_entry, const uint64_t* group_off, const uint32_t* start,
                                                  const uint32_t* end, const int* coarse, uint32_t* out_anchor, int* out_cluster) {
  long m = 0;
  for (int i = 0; i < n; i++) {
    const int c = chain_cluster[i];
    const uint64_t g = group_off[c] + chain_entry[i];
    for (long j = (long)end[g] - 1; j >= (long)start[g]; j--) { out_anchor[m] = (uint32_t)j; out_cluster[m] = coarse[c]; m++; }
  }
  return m;
}
