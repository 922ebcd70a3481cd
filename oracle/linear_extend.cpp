// oracle/linear_extend.cpp -- TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// CPU restatement of the pair version of LinearExtend (reference: LinearExtend.h:658-716), Checkbp
// (:50-85) and DecideCoordinates (:105-128), as MapRead_lowacc applies them to every cleaned cluster
// (Map_lowacc.h:131-137).
// Parity status: PARITY UNPINNED -- LinearExtend.h includes Clustering.h -> Genome (htslib); restated
// from the source text.
#include "oracle_common.h"
#include <algorithm>
#include <vector>

namespace {
void checkbp(uint32_t curq, uint32_t curt, uint32_t nxq, uint32_t nxt, const char* chrom, uint32_t chromLen, const char* read,
             uint32_t readLen, int strand, int K, uint32_t& qe, uint32_t& te) {   // :50-85
  uint32_t curQ, curT, nextQ, nextT;
  if (strand == 0) {
    curQ = curq + K; curT = std::min(chromLen, curt + (uint32_t)K);
    nextQ = nxq; nextT = std::min(chromLen, nxt);
    while (curQ < readLen && curT < chromLen && nextQ > curQ && nextT > curT && chrom[curT] == read[curQ]) { curQ++; curT++; }
  } else {
    curQ = curq + K; curT = std::min(chromLen - 1, curt - 1);
    nextQ = nxq; nextT = std::min(chromLen - 1, nxt + (uint32_t)K - 1);
    while (curQ < readLen && nextQ > curQ && nextT < curT && chrom[curT] == read[curQ]) { curQ++; curT--; }
  }
  qe = curQ; te = curT;
}
}

// q/t: the cluster's matches (t chromosome-relative, diagonal-sorted).  Outputs up to n extended
// anchors (eq, et, elen) and the box of DecideCoordinates.  Returns their number.
extern "C" long oracle_linear_extend(const uint32_t* q, const uint32_t* t, long n, int strand, int K, const char* read, uint32_t readLen,
                                     const char* chrom, uint32_t chromLen, uint32_t* eq, uint32_t* et, int* elen, uint32_t* box) {
  long ne = 0;
  auto push = [&](uint32_t a, uint32_t b, int l) { eq[ne] = a; et[ne] = b; elen[ne] = l; ne++; };
  long i = 1, m = 0;
  while (i < n) {                                                        // :673-708
    int64_t curDiag, nextDiag;
    if (strand == 0) { curDiag = (int64_t)q[i - 1] - (int64_t)t[i - 1]; nextDiag = (int64_t)q[i] - (int64_t)t[i]; }
    else { curDiag = (int64_t)q[i - 1] + (int64_t)t[i - 1]; nextDiag = (int64_t)q[i] + (int64_t)t[i]; }
    if (curDiag == nextDiag) {
      if (q[i] < q[i - 1] + (uint32_t)K) i++;
      else {
        uint32_t qe, te;
        checkbp(q[i - 1], t[i - 1], q[i], t[i], chrom, chromLen, read, readLen, strand, K, qe, te);
        if (strand == 0 && qe == q[i] && te == t[i]) i++;
        else if (strand == 1 && qe == q[i] && te == t[i] + (uint32_t)K - 1) i++;
        else {
          push(q[m], strand == 0 ? t[m] : te + 1, (int)(qe - q[m]));
          m = i; i++;
        }
      }
    } else {
      push(q[m], strand == 0 ? t[m] : t[i - 1], (int)(q[i - 1] + K - q[m]));
      m = i; i++;
    }
  }
  if (i == n) push(q[m], strand == 0 ? t[m] : t[i - 1], (int)(q[i - 1] + K - q[m]));   // :710-714
  if (ne) {                                                              // DecideCoordinates :105-128
    uint32_t qS = eq[0], qE = eq[0] + elen[0], tS = et[0], tE = et[0] + elen[0];
    for (long x = 1; x < ne; x++) {
      qS = std::min(qS, eq[x]); qE = std::max(qE, eq[x] + (uint32_t)elen[x]);
      tS = std::min(tS, et[x]); tE = std::max(tE, et[x] + (uint32_t)elen[x]);
    }
    box[0] = qS; box[1] = qE; box[2] = tS; box[3] = tE;
  }
  return ne;
}

// ---- the cluster version, one element of a chain ---------------------------------------------------------------------------------------
// LinearExtend(vector<Cluster*> clusters, vector<Cluster>& extCluster, vector<Tup>& chain, opts, genome, read, start, overlap, skiprepetitive, K)
// (LinearExtend.h:136-352; LinearExtend_chain :783, Map_highacc.h:580), for chain element c: the refined cluster's matches (t relative to its
// chromosome) are diagonal / anti-diagonal sorted IN PLACE (:201-210: DiagonalSort / AntiDiagonalSort of clusters[cm]->matches), the Set of
// the neighbours' box coordinates that fall strictly inside this cluster's box is built when skiprepetitive and anchorfreq <= 1.1 (:161-192;
// prevBox / nextBox = NULL at the ends of the chain), matches touching the Set become anchors of length K flagged `overlap` (CheckOverlap
// :88-101), everything else is merged along its diagonal as in the pair version.  Out: anchors (eq, et, elen, eovl), the box of
// DecideCoordinates (:105-128), *n_overlap = flagged anchors.  Returns the number of anchors (<= n).
extern "C" long oracle_linear_extend_cluster(long n, uint32_t* q, uint32_t* t, int strand, const uint32_t* box, const uint32_t* prevBox, const uint32_t* nextBox,
                                             float anchorfreq, int skiprepetitive, int K, const char* read, uint32_t readLen, const char* chrom, uint32_t chromLen,
                                             uint32_t* eq, uint32_t* et, int* elen, uint8_t* eovl, uint32_t* obox, int* n_overlap) {
  *n_overlap = 0;
  if (n == 0) return 0;                                                  // :145
  struct SetE { uint32_t v; int isT; };
  std::vector<SetE> Set;
  const uint32_t qsb = box[0], qeb = box[1], tsb = box[2], teb = box[3];
  if (skiprepetitive && anchorfreq <= 1.1f) {
    for (const uint32_t* nb : {prevBox, nextBox}) {
      if (!nb) continue;
      if (nb[0] > qsb && nb[0] < qeb) Set.push_back({nb[0], 0});
      if (nb[1] > qsb && nb[1] < qeb) Set.push_back({nb[1], 0});
      if (nb[2] > tsb && nb[2] < teb) Set.push_back({nb[2], 1});
      if (nb[3] > tsb && nb[3] < teb) Set.push_back({nb[3], 1});
    }
  }
  {                                                                      // :201-210
    std::vector<std::pair<uint32_t, uint32_t>> m((size_t)n);
    for (long i = 0; i < n; i++) m[i] = {q[i], t[i]};
    if (strand == 0) std::sort(m.begin(), m.end(), [](const std::pair<uint32_t, uint32_t>& a, const std::pair<uint32_t, uint32_t>& b) {
      const long ad = (long)a.first - (long)a.second, bd = (long)b.first - (long)b.second; return ad != bd ? ad < bd : a.first < b.first; });
    else std::sort(m.begin(), m.end(), [](const std::pair<uint32_t, uint32_t>& a, const std::pair<uint32_t, uint32_t>& b) {
      const uint32_t ad = a.first + a.second, bd = b.first + b.second; return ad != bd ? ad < bd : a.first < b.first; });
    for (long i = 0; i < n; i++) { q[i] = m[i].first; t[i] = m[i].second; }
  }
  auto ovp = [&](long i) {                                               // CheckOverlap :88-101
    for (const SetE& s : Set) {
      if (s.isT == 0 && s.v >= q[i] && s.v < q[i] + (uint32_t)K) return true;
      if (s.isT == 1 && s.v >= t[i] && s.v < t[i] + (uint32_t)K) return true;
    }
    return false;
  };
  long ne = 0;
  auto push = [&](uint32_t a, uint32_t b, int l, int o) { eq[ne] = a; et[ne] = b; elen[ne] = l; eovl[ne] = (uint8_t)o; ne++; *n_overlap += o; };
  long i = 1, m = 0;
  bool chm = true;
  while (i < n) {                                                        // :218-329
    if (chm) {
      if (ovp(m)) { push(q[m], t[m], K, 1); m = i; i++; chm = true; continue; }
      chm = false;
    }
    if (ovp(i)) {
      push(q[m], strand == 0 ? t[m] : t[i - 1], (int)(q[i - 1] + K - q[m]), 0);
      push(q[i], t[i], K, 1);
      m = i + 1; i = m + 1; chm = true;
      continue;
    }
    int64_t curDiag, nextDiag;
    if (strand == 0) { curDiag = (int64_t)q[i - 1] - (int64_t)t[i - 1]; nextDiag = (int64_t)q[i] - (int64_t)t[i]; }
    else { curDiag = (int64_t)q[i - 1] + (int64_t)t[i - 1]; nextDiag = (int64_t)q[i] + (int64_t)t[i]; }
    if (curDiag == nextDiag) {
      if (q[i] < q[i - 1] + (uint32_t)K) i++;
      else {
        uint32_t qe, te;
        checkbp(q[i - 1], t[i - 1], q[i], t[i], chrom, chromLen, read, readLen, strand, K, qe, te);
        if (strand == 0 && qe == q[i] && te == t[i]) i++;
        else if (strand == 1 && qe == q[i] && te == t[i] + (uint32_t)K - 1) i++;
        else { push(q[m], strand == 0 ? t[m] : te + 1, (int)(qe - q[m]), 0); m = i; i++; }
      }
    } else { push(q[m], strand == 0 ? t[m] : t[i - 1], (int)(q[i - 1] + K - q[m]), 0); m = i; i++; }
    chm = false;
  }
  if (i == n) push(q[m], strand == 0 ? t[m] : t[i - 1], (int)(q[i - 1] + K - q[m]), 0);   // :331-342
  if (ne) {                                                              // DecideCoordinates :105-128
    uint32_t qS = eq[0], qE = eq[0] + elen[0], tS = et[0], tE = et[0] + elen[0];
    for (long x = 1; x < ne; x++) { qS = std::min(qS, eq[x]); qE = std::max(qE, eq[x] + (uint32_t)elen[x]); tS = std::min(tS, et[x]); tE = std::max(tE, et[x] + (uint32_t)elen[x]); }
    obox[0] = qS; obox[1] = qE; obox[2] = tS; obox[3] = tE;
  }
  return ne;
}
