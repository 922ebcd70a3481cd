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
