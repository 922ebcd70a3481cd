// oracle/refine_clusters.cpp -- TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// CPU restatement of the tier-2 lookup of the high-accuracy path, one cluster at a time:
//   REFINEclusters                              ClusterRefine.h:50-240   (Map_highacc.h:429-447)
//     Cluster::CHROMIndex                       Clustering.h:326-336     SwapStrand                     ClusterRefine.h:24-31
//     CartesianTargetSort / LowerBound / UpperBound   Sorting.h:183-221  LocalIndex::LookupIndex        MMIndex.h:175-190
//     CompareLists<LocalTuple,SmallTuple>       (oracle_compare_lists_local)    AppendValues             TupleOps.h:159-195
//     Cluster::SetClusterBoundariesFromMatches  Clustering.h:308-322
// Parity status: PARITY UNPINNED -- ClusterRefine.h needs Genome.h / Clustering.h (htslib); restated from the source text.
#include "oracle_common.h"
#include <algorithm>
#include <utility>
#include <vector>

extern "C" long oracle_compare_lists_local(const uint32_t* q, long nq, const uint32_t* t, long nt, long maxFreq, int64_t maxDiag,
                                           int64_t minDiag, uint32_t* out_qi, uint32_t* out_ti, long cap);

namespace {
int header_find(const uint64_t* pos, int npos, uint64_t query, bool& ub) {   // Genome.h:20-32
  if (npos > 0 && query == pos[0]) return 0;
  const uint64_t* it = std::lower_bound(pos, pos + npos, query);
  int i = (int)(it - pos);
  if (i == npos) { ub = true; return i - 1; }
  if (query == *it) return i;
  return i - 1;
}
long lookup_index(const uint64_t* so, long n, uint64_t pos, bool& ub) {      // MMIndex.h:175-190
  if (n == 0) return 0;
  const uint64_t* it = std::lower_bound(so, so + n, pos);
  long index = it - so;
  if (index == n) { ub = true; return index - 1; }
  if (*it != pos) return index - 1;
  return index;
}
typedef std::pair<uint32_t, uint32_t> QT;                                 // (first.pos, second.pos)
bool target_less(const QT& a, const QT& b) { if (a.second != b.second) return a.second < b.second; return a.first < b.first; }   // CartesianTargetSortOp
}  // namespace

struct oracle_rcl_opts { int window, smallK, K, maxFreq; };

// One cluster: n matches (q on the forward read, t genome-wide), its box (qStart, qEnd, tStart, tEnd genome-wide), strand.  Read index of
// that strand / genome index as for oracle_refine_splitchain.  Out: chromIndex, the refined matches (t chromosome-relative; q back on the
// forward read) up to cap, box, refineEffiency.  Returns the number of matches; -1 if the reference reads outside an array; -2 if
// CHROMIndex rejects the cluster (it is cleared, :61-65).
extern "C" long oracle_refine_cluster(int n, const uint32_t* mq, const uint32_t* mt, const uint32_t* box, int strand, const uint64_t* chromPos, int nChrom,
                                      uint32_t readLen, long nWq, const uint64_t* qSeqOff, const uint64_t* qBnd, const uint32_t* qTup, long nWg,
                                      const uint64_t* gSeqOff, const uint64_t* gBnd, const uint32_t* gTup, const oracle_rcl_opts* o, long cap, int* chromOut,
                                      uint32_t* outQ, uint32_t* outT, uint32_t* outBox, float* outEff) {
  if (n == 0) return 0;
  bool ub = false;
  uint32_t qStart = box[0], qEnd = box[1];
  const uint32_t tStart = box[2], tEnd = box[3];
  const int c0 = header_find(chromPos, nChrom + 1, (uint64_t)tStart + 1, ub), c1 = header_find(chromPos, nChrom + 1, tEnd, ub);   // CHROMIndex :326-336
  if (ub) return -1;
  if (c0 != c1) return -2;
  *chromOut = c0;
  const uint32_t chromOffset = (uint32_t)chromPos[c0];
  std::vector<QT> M(n);
  for (int i = 0; i < n; i++) M[i] = QT(mq[i], mt[i] - chromOffset);
  const int fi = header_find(chromPos, nChrom + 1, tEnd, ub);             // GetNextOffset
  if (ub || fi + 1 > nChrom) return -1;
  const uint32_t chromEndOffset = (uint32_t)chromPos[fi + 1];
  if (strand == 1) {                                                      // SwapStrand(read, opts, cluster, opts.globalK)
    for (int i = 0; i < n; i++) M[i].first = readLen - (M[i].first + (uint32_t)o->K);
    const uint32_t r = qStart; qStart = readLen - qEnd; qEnd = readLen - r;
  }
  int64_t maxDN = (int64_t)M[0].second - (int64_t)M[0].first, minDN = maxDN;
  for (int i = 0; i < n; i++) { const int64_t d = (int64_t)M[i].second - (int64_t)M[i].first; maxDN = std::max(maxDN, d); minDN = std::min(minDN, d); }
  int64_t maxDiagNum = maxDN + 100, minDiagNum = minDN - 100;
  std::sort(M.begin(), M.end(), target_less);                             // CartesianTargetSort
  uint32_t wts, wte;
  if (chromOffset + (uint32_t)o->window > tStart) wts = chromOffset; else wts = tStart - o->window;
  if (tEnd + (uint32_t)o->window > chromEndOffset) wte = chromEndOffset - 1; else wte = tEnd + o->window;
  const long ls = lookup_index(gSeqOff, nWg + 1, wts, ub), le = lookup_index(gSeqOff, nWg + 1, wte, ub);
  if (ub) return -1;
  long nOut = 0;
  std::vector<uint32_t> pq, pt;
  for (long lsi = ls; lsi <= le; lsi++) {
    if (lsi + 1 > nWg) return -1;
    if (gSeqOff[lsi] < chromOffset || gSeqOff[lsi + 1] < chromOffset) continue;
    const uint32_t gStart = (uint32_t)(gSeqOff[lsi] - chromOffset), gEnd = (uint32_t)(gSeqOff[lsi + 1] - 1 - chromOffset);
    if (gStart >= gEnd) continue;
    int matchStart = (int)(std::lower_bound(M.begin(), M.end(), QT(0, gStart), target_less) - M.begin());
    int matchEnd = (int)(std::upper_bound(M.begin() + matchStart, M.end(), QT(0, gEnd), target_less) - (M.begin() + matchStart));
    matchEnd += matchStart;
    if (matchEnd == n) matchEnd--;
    if (matchStart >= n) continue;
    uint32_t prev_readEnd = 0;
    uint32_t readStart = M[matchStart].first, readEnd = M[matchEnd].first;
    if (readStart == readEnd) { if (lsi > ls && readStart > prev_readEnd) readStart = prev_readEnd; }
    if (lsi == ls) { if (readStart < (uint32_t)o->window) readStart = 0; else readStart -= o->window; }
    if (lsi == le) { if (readEnd + (uint32_t)o->window > readLen) readEnd = readLen; else readEnd += o->window; }
    if (readStart > readEnd) continue;
    const long qi0 = lookup_index(qSeqOff, nWq + 1, readStart, ub);
    const long qi1 = lookup_index(qSeqOff, nWq + 1, std::min(readEnd, readLen - 1), ub);
    if (ub) return -1;
    for (long qi = qi0; qi <= qi1; ++qi) {
      if (qi + 1 > nWq) return -1;
      const uint64_t qb0 = qBnd[qi], qb1 = qBnd[qi + 1], gb0 = gBnd[lsi], gb1 = gBnd[lsi + 1];
      const uint32_t readSegmentStart = (uint32_t)qSeqOff[qi];
      const long capP = std::max<long>(1, (long)(qb1 - qb0) * (long)std::max<uint64_t>(1, gb1 - gb0));
      pq.resize(capP); pt.resize(capP);
      const long np = oracle_compare_lists_local(qTup + qb0, (long)(qb1 - qb0), gTup + gb0, (long)(gb1 - gb0), o->maxFreq, 0, 0, pq.data(), pt.data(), capP);
      const uint32_t ts = tStart - chromOffset, te = tEnd - chromOffset;
      for (long p = 0; p < np; p++) {                                     // AppendValues
        const uint32_t fp = (qTup[qb0 + pq[p]] >> 20) + readSegmentStart, sp = (gTup[gb0 + pt[p]] >> 20) + gStart;
        const int64_t diag = (int64_t)sp - (int64_t)fp;
        if (diag >= minDiagNum && diag <= maxDiagNum && fp >= qStart && fp < qEnd && sp >= ts && sp < te) {
          if (nOut < cap) { outQ[nOut] = fp; outT[nOut] = sp; }
          nOut++;
        }
      }
    }
  }
  if (nOut == 0 || nOut > cap) return nOut;
  if (strand == 1) for (long i = 0; i < nOut; i++) outQ[i] = readLen - (outQ[i] + (uint32_t)o->smallK);
  uint32_t bqs = outQ[0], bqe = bqs + o->smallK, bts = outT[0], bte = bts + o->smallK;
  for (long i = 1; i < nOut; i++) {
    bte = std::max(bte, outT[i] + (uint32_t)o->smallK); bts = std::min(bts, outT[i]);
    bqe = std::max(bqe, outQ[i] + (uint32_t)o->smallK); bqs = std::min(bqs, outQ[i]);
  }
  outBox[0] = bqs; outBox[1] = bqe; outBox[2] = bts; outBox[3] = bte;
  *outEff = ((float)nOut) / std::min(bqe - bqs, bte - bts);
  return nOut;
}
