// oracle/refine_splitchain.cpp -- TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// CPU restatement of the tier-2 lookup of the low-accuracy path, one split chain at a time:
//   Refine_splitchain                                   ChainRefine.h:384-576  (Map_lowacc.h:294)
//     SwapStrand(read, opts, cluster, K)                ClusterRefine.h:24-31
//     LocalIndex::LookupIndex                           MMIndex.h:175-190
//     GenomeHeader::GetNextOffset / Find                Genome.h:43-47, :20-32
//     CompareLists<LocalTuple,SmallTuple>               CompareLists.h:9   (oracle_compare_lists_local, pinned in test_local.py)
//     AppendValues<LocalPairs>                          TupleOps.h:159-195
//     Cluster::SetClusterBoundariesFromMatches          Clustering.h:308-322
// Parity status: PARITY UNPINNED -- ChainRefine.h needs Genome.h / Clustering.h (htslib); restated from the source text.
//
// UNDEFINED BEHAVIOUR IN THE REFERENCE (default options): with opts.limitrefine (Options.h:234, true unless --skiplimitrefine) the upper
// diagonal bound of every genome window starts from `miniMaxDiag = miniMaxDiag;` (ChainRefine.h:468), an uninitialised long.  SURVEY.md
// H2 measured what the reference binary does: the stale slot holds a pointer-sized value, so there is no upper diagonal bound at all
// (`1L<<60` reproduced the binary on 400/400 reads; "as intended" initialisation changed 13/200 -ONT records).  This restatement and
// the kernel implement the measured behaviour: +infinity.  limitrefine = 0 is fully defined by the source and is restated literally.
#include "oracle_common.h"
#include <algorithm>
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

long lookup_index(const uint64_t* seqOffsets, long n, uint64_t pos, bool& ub) {   // MMIndex.h:175-190 (n = seqOffsets.size())
  if (n == 0) return 0;
  const uint64_t* it = std::lower_bound(seqOffsets, seqOffsets + n, pos);
  long index = it - seqOffsets;
  if (index == n) { ub = true; return index - 1; }                        // *it past the end
  if (*it != pos) return index - 1;
  return index;
}

}  // namespace

struct oracle_rsc_opts {
  int window;          // smallOpts.window (Options.h:158)
  int smallK;          // smallOpts.globalK
  int K;               // opts.globalK (SwapStrand of the extended clusters)
  int limitrefine;     // opts.limitrefine
  int maxFreq;         // smallOpts.localMaxFreq (CompareLists, Global = false)
};

// The chain (after RemoveSpuriousJump) as arrays over its anchors: q, t (genome-wide), len, cluster index, strand of that cluster.
// One split chain: sptc[m] (indices into the chain), box, Strand, chromIndex, ClusterIndex[nci].  chromPos[nChrom+1] = genome.header.pos.
// Read index of strand `Strand`: qSeqOff[nWq+1] (seqOffsets), qBnd[nWq+1] (tupleBoundaries), qTup; genome index likewise (g*).
// Out: the refined cluster's matches (q, t relative to the chromosome; q back on the forward read if Strand) up to cap, its box
// (qStart,qEnd,tStart,tEnd) and refineEffiency.  Returns the number of matches, -1 if the reference would read outside an array.
extern "C" long oracle_refine_splitchain(int n, const uint32_t* q, const uint32_t* t, const int* len, const int* cluster, const uint8_t* cstrand,
                                         int m, const int* sptc, const uint32_t* box, int Strand, int chromIndex, int nci, const int* ci,
                                         const uint64_t* chromPos, int nChrom, uint32_t readLen, long nWq, const uint64_t* qSeqOff,
                                         const uint64_t* qBnd, const uint32_t* qTup, long nWg, const uint64_t* gSeqOff, const uint64_t* gBnd,
                                         const uint32_t* gTup, const oracle_rsc_opts* o, long cap, uint32_t* outQ, uint32_t* outT, uint32_t* outBox,
                                         float* outEff) {
  (void)n;
  if (m == 0) return 0;
  bool ub = false;
  const uint32_t chromOffset = (uint32_t)chromPos[chromIndex];
  // :395-409 the clusters of this split chain are brought to chromosome coordinates on the strand they lie on
  auto flipped = [&](int c) { for (int k = 0; k < nci; k++) if (ci[k] == c) return true; return false; };
  auto tS = [&](int i) { const int a = sptc[i]; return flipped(cluster[a]) ? t[a] - chromOffset : t[a]; };
  auto qS = [&](int i) {
    const int a = sptc[i];
    return (flipped(cluster[a]) && cstrand[a] == 1) ? readLen - (q[a] + (uint32_t)o->K) : q[a];
  };
  auto qE = [&](int i) { return qS(i) + (uint32_t)len[sptc[i]]; };
  const uint32_t QStart = box[0], QEnd = box[1], TStart = box[2], TEnd = box[3];
  const int fi = header_find(chromPos, nChrom + 1, TEnd, ub);             // GetNextOffset :43-47
  if (ub || fi + 1 > nChrom) return -1;
  const uint32_t chromEndOffset = (uint32_t)chromPos[fi + 1];
  int64_t maxDN = (int64_t)tS(0) - (int64_t)qS(0), minDN = maxDN;
  for (int db = 0; db < m; db++) {
    maxDN = std::max(maxDN, (int64_t)tS(db) - (int64_t)qS(db));
    minDN = std::min(minDN, (int64_t)tS(db) - (int64_t)qS(db));
  }
  int64_t maxDiagNum = maxDN + 50, minDiagNum = minDN - 50;
  const uint32_t wts = (TStart >= chromOffset + (uint32_t)o->window) ? TStart - o->window : chromOffset;
  const uint32_t wte = (TEnd + (uint32_t)o->window < chromEndOffset) ? TEnd + o->window : chromEndOffset;
  const long ls = lookup_index(gSeqOff, nWg + 1, wts, ub), le = lookup_index(gSeqOff, nWg + 1, wte, ub);
  if (ub) return -1;
  long nOut = 0;
  std::vector<uint32_t> pq, pt;
  int matchStart = 0, matchEnd = 0;
  for (long lsi = ls; lsi <= le; lsi++) {
    // lsi == nWg happens next to the end of the genome: the reference reads seqOffsets one past its end there, and whatever it finds
    // the window holds no anchor (every tStart is <= the chromosome length), so the iteration adds nothing
    if (lsi + 1 > nWg) continue;
    if (gSeqOff[lsi] < chromOffset || gSeqOff[lsi + 1] < chromOffset) continue;
    const uint32_t gStart = (uint32_t)(gSeqOff[lsi] - chromOffset), gEnd = (uint32_t)(gSeqOff[lsi + 1] - 1 - chromOffset);
    if (gStart >= gEnd) continue;
    while (matchStart < m && tS(matchStart) <= gStart) matchStart++;
    matchEnd = matchStart;
    while (matchEnd < m && tS(matchEnd) < gEnd) matchEnd++;
    if (matchStart >= m) continue;
    if (matchEnd == matchStart) continue;
    uint32_t prev_readEnd = 0, prev_readStart = readLen;
    (void)prev_readStart;
    uint32_t readStart = qS(matchStart), readEnd = qS(matchEnd - 1);
    for (int mi = matchStart; mi < matchEnd; mi++) {
      if (qS(mi) < readStart) readStart = qS(mi);
      if (qE(mi) > readEnd) readEnd = qE(mi);
    }
    if (readStart == readEnd) { if (lsi > ls && readStart > prev_readEnd) readStart = prev_readEnd; }
    int64_t miniMinDiag = 0, miniMaxDiag = 0;
    if (o->limitrefine) {
      miniMinDiag = (int64_t)tS(matchStart) - (int64_t)qS(matchStart);
      // :468 `miniMaxDiag = miniMaxDiag;` reads an uninitialised local.  What the reference binary does (measured on
      // x86-64 Linux, -O0 and -O2, -t 1 and -t 8): the stale stack slot holds a pointer-sized value (~9.4e13) that only grows, so
      // there is NO upper diagonal bound when limitrefine is on -- only miniMinDiag - 100 filters.  Hard-coded as +infinity.
      miniMaxDiag = (int64_t)1 << 60;
      for (int mi = matchStart; mi < matchEnd; mi++)
        miniMinDiag = std::min(miniMinDiag, (int64_t)tS(mi) - (int64_t)qS(mi));
      miniMinDiag -= 100;
    }
    const int sow = 500;
    if (lsi == ls) readStart = (readStart < (uint32_t)sow) ? 0 : readStart - sow;
    if (lsi == le) readEnd = (readEnd + sow > readLen) ? readLen : readEnd + sow;
    if (readStart > readEnd) continue;
    const long qi0 = lookup_index(qSeqOff, nWq + 1, readStart, ub);
    const long qi1 = lookup_index(qSeqOff, nWq + 1, std::min(readEnd, readLen - 1), ub);
    if (ub) return -1;
    uint32_t qStart, qEnd;
    for (long qi = qi0; qi <= qi1; ++qi) {
      if (qi + 1 > nWq) return -1;
      const uint64_t qb0 = qBnd[qi], qb1 = qBnd[qi + 1], gb0 = gBnd[lsi], gb1 = gBnd[lsi + 1];
      const uint32_t readSegmentStart = (uint32_t)qSeqOff[qi];
      const long capP = std::max<long>(1, (long)(qb1 - qb0) * (long)std::max<uint64_t>(1, gb1 - gb0));
      pq.resize(capP); pt.resize(capP);
      const long np = oracle_compare_lists_local(qTup + qb0, (long)(qb1 - qb0), gTup + gb0, (long)(gb1 - gb0), o->maxFreq, 0, 0, pq.data(), pt.data(), capP);
      if (np > capP) return -2;
      if (Strand == 0) { qStart = QStart; qEnd = QEnd; }
      else { qStart = readLen - QEnd; qEnd = readLen - QStart; }
      const int64_t mx = o->limitrefine ? miniMaxDiag : maxDiagNum, mn = o->limitrefine ? miniMinDiag : minDiagNum;
      const uint32_t ts = TStart - chromOffset, te = TEnd - chromOffset;
      for (long p = 0; p < np; p++) {                                     // AppendValues TupleOps.h:159-195
        const uint32_t fp = (qTup[qb0 + pq[p]] >> 20) + readSegmentStart, sp = (gTup[gb0 + pt[p]] >> 20) + gStart;
        const int64_t diag = (int64_t)sp - (int64_t)fp;
        if (diag >= mn && diag <= mx && fp >= qStart && fp < qEnd && sp >= ts && sp < te) {
          if (nOut < cap) { outQ[nOut] = fp; outT[nOut] = sp; }
          nOut++;
          prev_readEnd = std::max(prev_readEnd, fp); prev_readStart = std::min(prev_readStart, fp);
        }
      }
    }
  }
  if (nOut == 0 || nOut > cap) return nOut;
  if (Strand == 1) for (long i = 0; i < nOut; i++) outQ[i] = readLen - (outQ[i] + (uint32_t)o->smallK);   // SwapStrand :24-31
  uint32_t bqs = outQ[0], bqe = bqs + o->smallK, bts = outT[0], bte = bts + o->smallK;                      // Clustering.h:308-322
  for (long i = 1; i < nOut; i++) {
    bte = std::max(bte, outT[i] + (uint32_t)o->smallK); bts = std::min(bts, outT[i]);
    bqe = std::max(bqe, outQ[i] + (uint32_t)o->smallK); bqs = std::min(bqs, outQ[i]);
  }
  outBox[0] = bqs; outBox[1] = bqe; outBox[2] = bts; outBox[3] = bte;
  *outEff = ((float)nOut) / std::min(bqe - bqs, bte - bts);
  return nOut;
}
