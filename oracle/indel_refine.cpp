// oracle/indel_refine.cpp -- TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// CPU restatement of IndelRefineAlignment (reference: IndelRefine.h:53-784): re-aligns runs
// of gapless blocks separated by small gaps with a 3-state (match / deletion / insertion)
// affine-gap DP inside a per-target-row query window [qS,qE], and falls back to
// AffineOneGapAlign for very short spans.
//
// Parity status: PARITY UNPINNED.  IndelRefine.h includes Genome.h, which includes
// htslib/kseq.h; htslib is neither in this image nor vendored by the reference, so the
// function cannot be compiled here and the reference ships no test or fixture for it.
// Restated from the source text; every step cites the lines it follows.  (Its
// AffineOneGapAlign fallback IS pinned, see aog.cpp.)
#include "oracle_common.h"
#include <algorithm>
#include <vector>

extern "C" int oracle_affine_one_gap_align(const char* q, int qLen, const char* t, int tLen, int m, int mm,
                                           int indel, int k, int* blocks_out, int cap, int* n_blocks, int* status);

namespace {
struct Blk { long q, t, len; };
enum { P_DIAG = 0, P_LEFT = 1, P_DOWN = 2, P_BOUND = 3, P_DELOPEN = 4, P_DELEXT = 5, P_DELCLOSE = 6,
       P_INSOPEN = 7, P_INSEXT = 8, P_INSCLOSE = 9, P_DONE = 20 };
const int BAD = -999999999;  // :368
}

// blocks_in: n_in (qPos,tPos,len) triples; qSeq = the read strand the alignment is on,
// tSeq = the chromosome (both absolute coordinates).  Returns the number of refined blocks
// (written to blocks_out up to cap triples).  *status bit0: the reference would loop forever
// or index out of range; bit1: an AffineOneGapAlign fallback reported a status.
extern "C" long oracle_indel_refine(const int* blocks_in, long n_in, const char* qSeq, long readLen, const char* tSeq,
                                    long chromLen, int refineBand, int match, int mismatch, int indel, int endAlign,
                                    int* blocks_out, long cap, int* status) {
  *status = 0;
  std::vector<Blk> blocks(n_in);
  for (long i = 0; i < n_in; i++) blocks[i] = {blocks_in[3 * i], blocks_in[3 * i + 1], blocks_in[3 * i + 2]};
  std::vector<Blk> refined;
  auto finish = [&](const std::vector<Blk>& v) {
    for (size_t i = 0; i < v.size() && (long)i < cap; i++) {
      blocks_out[3 * i] = (int)v[i].q; blocks_out[3 * i + 1] = (int)v[i].t; blocks_out[3 * i + 2] = (int)v[i].len;
    }
    return (long)v.size();
  };
  const int k = refineBand;                                            // :60
  const int maxGap = k - 1;                                            // :61
  if (blocks.size() == 0 || blocks.size() == 1) return finish(blocks); // :79
  int startMatch = 0, endMatch = 0;
  if (endAlign) {                                                      // :89-130
    long qStart = blocks[0].q, tStart = blocks[0].t;
    int minStart = (int)std::min(qStart, tStart);
    int addStart = 0;
    if (minStart < 40) { tStart -= minStart; qStart -= minStart; startMatch = minStart; addStart = 1; }
    size_t e = blocks.size();
    long qAlnEnd = blocks[e - 1].q + blocks[e - 1].len, tAlnEnd = blocks[e - 1].t + blocks[e - 1].len;
    int minEnd = (int)std::min(readLen - qAlnEnd, chromLen - tAlnEnd);
    int addEnd = 0;
    if (minEnd < 40) { endMatch = minEnd; addEnd++; }
    std::vector<Blk> nb(blocks.size() + addStart + addEnd);
    if (addStart) nb[0] = {qStart, tStart, startMatch};
    std::copy(blocks.begin(), blocks.end(), nb.begin() + addStart);
    if (addEnd) nb[nb.size() - 1] = {qAlnEnd, tAlnEnd, endMatch};
    blocks = nb;
  }
  const long nB = (long)blocks.size();
  long startBlock = 0, endBlock = 0;
  std::vector<int> qS, qE;
  while (endBlock < nB) {                                              // :132
    long qStart = blocks[startBlock].q, tStart = blocks[startBlock].t;
    long qPos = qStart + blocks[startBlock].len, tPos = tStart + blocks[startBlock].len;
    int tGap = 0, qGap = 0;
    if (endBlock < nB - 1) { tGap = (int)(blocks[endBlock + 1].t - tPos); qGap = (int)(blocks[endBlock + 1].q - qPos); }
    while (endBlock < nB - 1 && qGap < maxGap && tGap < maxGap &&      // :148-162
           (startBlock == endBlock || blocks[endBlock].len < 100)) {
      endBlock++;
      qPos = blocks[endBlock].q + blocks[endBlock].len;
      tPos = blocks[endBlock].t + blocks[endBlock].len;
      if (endBlock + 1 < nB - 1) {                                     // (sic) gaps go stale before the last block
        tGap = (int)(blocks[endBlock + 1].t - tPos);
        qGap = (int)(blocks[endBlock + 1].q - qPos);
      }
    }
    Blk alt{0, 0, 0};
    bool usedAlt = false;
    if (endBlock == startBlock) {
      refined.push_back(blocks[startBlock]);                           // :170-173
    } else {
      if (blocks[startBlock].len > maxGap) {                           // :178-196 trim a long first block
        long advanced = blocks[startBlock].len - maxGap;
        blocks[startBlock].len -= maxGap;
        refined.push_back(blocks[startBlock]);
        blocks[startBlock].q += advanced; blocks[startBlock].t += advanced; blocks[startBlock].len = maxGap;
        qStart += advanced; tStart += advanced;
      }
      if (blocks[endBlock].len > maxGap) {                             // :198-211 split a long last block
        usedAlt = true;
        alt = blocks[endBlock];
        alt.q += maxGap; alt.t += maxGap; alt.len -= maxGap;
        blocks[endBlock].len = maxGap;
        qPos = blocks[endBlock].q + maxGap; tPos = blocks[endBlock].t + maxGap;
      }
      const long qEnd = blocks[endBlock].q + blocks[endBlock].len, tEnd = blocks[endBlock].t + blocks[endBlock].len;
      const long tLen = tPos - tStart;                                 // :217-218
      if (tLen <= 0) { *status |= 1; return finish(refined); }
      qS.assign(tLen, -1); qE.assign(tLen, -1);                        // :220-223
      long t = blocks[startBlock].t, q = blocks[startBlock].q;
      long tOff = 0;
      for (long b = startBlock; b <= endBlock; b++) {                  // :232-315 row windows
        int bqGap = 0, btGap = 0;
        long blockLength = blocks[b].len;
        if (b < endBlock) {
          bqGap = (int)(blocks[b + 1].q - (blocks[b].q + blockLength));
          btGap = (int)(blocks[b + 1].t - (blocks[b].t + blockLength));
          if (bqGap > 0 && btGap > 0) { int c = std::min(bqGap, btGap); bqGap -= c; btGap -= c; blockLength += c; }
        }
        for (long bi = 0; bi < blockLength; tOff++, bi++, q++, t++) {  // :252-283
          if (tOff >= tLen) { *status |= 1; return finish(refined); }
          if (qS[tOff] == -1) qS[tOff] = (int)std::max(q - k, qStart);
          else qS[tOff] = (int)std::min((long)qS[tOff], std::max(q - k, qStart));
          if (qE[tOff] == -1 || qE[tOff] < q + k) qE[tOff] = (int)std::min(qEnd - 1, q + k);
          for (int ki = 0; ki < k; ki++) {
            if (tOff - ki >= 0 && qE[tOff - ki] < q) qE[tOff - ki] = (int)q;
            if (tOff + ki < tLen && (qS[tOff + ki] == -1 || qS[tOff + ki] > q)) qS[tOff + ki] = (int)q;
          }
        }
        if (bqGap > btGap) {                                           // :287-305
          for (int qi = 0; qi < bqGap; qi++, q++)
            for (int ki = 0; ki < k; ki++) {
              if (tOff - ki >= 0 && tOff - ki < tLen && qE[tOff - ki] < q) qE[tOff - ki] = (int)q;
              if (tOff + ki < tLen && (qS[tOff + ki] == 0 || qS[tOff + ki] > q)) qS[tOff + ki] = (int)q;   // (sic) == 0
            }
        }
        if (btGap > bqGap) {                                           // :306-314
          for (int ti = 0; ti < btGap; tOff++, ti++, t++) {
            if (tOff >= tLen) { *status |= 1; return finish(refined); }
            qS[tOff] = (int)std::max(q - k, qStart);
            qE[tOff] = (int)std::min(qEnd - 1, q + k);
          }
        }
      }
      long matSize = 0;
      for (long qi = tLen; qi > 1; qi--) if (qS[qi - 1] < qS[qi - 2]) qS[qi - 2] = qS[qi - 1];   // :318-322
      for (long qi = 0; qi < tLen - 1; qi++) if (qE[qi] > qE[qi + 1]) qE[qi + 1] = qE[qi];       // :323-325
      for (long qi = 0; qi < tLen; qi++) matSize += qE[qi] - qS[qi] + 1;                         // :326-328
      const long tSeqLen = tEnd - tStart, qSeqLen = qEnd - qStart;
      const int gap = indel, gapOpen = indel * 2 + 1, gapExtend = 0;   // :338-340
      if (tSeqLen < k || qSeqLen < k) {                                // :344-357 short span
        std::vector<int> ob(3 * (std::min(tSeqLen, qSeqLen) + 2));
        int nb = 0, st = 0;
        oracle_affine_one_gap_align(qSeq + qStart, (int)qSeqLen, tSeq + tStart, (int)tSeqLen, match, mismatch, gap, k,
                                    ob.data(), (int)(ob.size() / 3), &nb, &st);
        if (st) *status |= 2;
        for (int i = 0; i < nb; i++) refined.push_back({ob[3 * i] + qStart, ob[3 * i + 1] + tStart, ob[3 * i + 2]});
      } else {
        for (long r = 0; r < tLen; r++) if (qE[r] < qS[r] || qS[r] < 0) { *status |= 1; return finish(refined); }
        std::vector<int> scoreMat(matSize, 0), pathMat(matSize, P_BOUND), indexMat(matSize, -1);   // :383-405
        std::vector<int> delScore(matSize, BAD), delPath(matSize, P_BOUND), delIndex(matSize, -1);
        std::vector<int> insScore(matSize, BAD), insPath(matSize, P_BOUND), insIndex(matSize, -1);
        indexMat[0] = 0; pathMat[0] = P_DONE;
        long rowStart = 0;
        for (long ti = 0; ti < tLen; ti++) {                           // :410-431 boundaries
          long rowLen = qE[ti] - qS[ti] + 1, rowEnd = rowStart + rowLen - 1;
          if (rowStart > 0) { scoreMat[rowStart] = BAD; pathMat[rowStart] = P_BOUND; insPath[rowStart] = P_BOUND; }
          else for (long qi = 1; qi < rowEnd; qi++) { scoreMat[qi] = scoreMat[qi - 1] + gap; pathMat[qi] = P_LEFT; indexMat[qi] = (int)(qi - 1); }
          if (ti < tLen - 1) { scoreMat[rowEnd] = BAD; pathMat[rowEnd] = P_BOUND; }
          rowStart += rowLen;
        }
        long curRowStart = qE[0] - qS[0] + 1, prevRowStart = 0, prevRowLen = qE[0] - qS[0] + 1;
        for (long ti = 1; ti < tLen; ti++) {                           // :438-622 fill
          long curRowLen = qE[ti] - qS[ti] + 1;
          long curRowOffset = qS[ti] - qS[ti - 1];
          long cur = curRowStart + 1, prev = prevRowStart + curRowOffset + 1;
          long rowEnd = (ti == tLen - 1) ? curRowLen : curRowLen - 1;
          const char tChar = tSeq[ti + tStart];
          for (long qi = 1; qi < rowEnd; qi++, cur++, prev++) {
            const bool aboveIn = qE[ti - 1] >= qi + qS[ti];
            int dOpen, dExt;
            if (aboveIn && pathMat[prev] != P_BOUND) { dOpen = scoreMat[prev] + gapOpen; dExt = delScore[prev] + gapExtend; }
            else { dOpen = BAD; dExt = BAD; }
            int mx = std::max(dOpen, dExt);
            delPath[cur] = (mx == dOpen) ? P_DELOPEN : P_DELEXT;
            delIndex[cur] = (int)prev;
            delScore[cur] = mx;
            int iOpen = scoreMat[cur - 1] + gapOpen, iExt = insScore[cur - 1] + gapExtend;
            mx = std::max(iOpen, iExt);
            insPath[cur] = (mx == iOpen) ? P_INSOPEN : P_INSEXT;
            insIndex[cur] = (int)(cur - 1);
            insScore[cur] = mx;
            int mS;
            if (aboveIn && pathMat[prev - 1] != P_BOUND) mS = scoreMat[prev - 1] + ((tChar == qSeq[qi + qS[ti]]) ? match : mismatch);
            else mS = BAD;
            int iS = scoreMat[cur - 1] + gap;
            int dS = (aboveIn && pathMat[prev] != P_BOUND) ? scoreMat[prev] + gap : BAD;
            int dC = delScore[cur], iC = insScore[cur];
            mx = std::max(mS, std::max(iS, std::max(dS, std::max(dC, iC))));
            scoreMat[cur] = mx;
            if (mx == mS) { pathMat[cur] = P_DIAG; indexMat[cur] = (int)(prev - 1); }
            else if (mx == iS) { pathMat[cur] = P_LEFT; indexMat[cur] = (int)(cur - 1); }
            else if (mx == dS) { pathMat[cur] = P_DOWN; indexMat[cur] = (int)prev; }
            else if (mx == dC) { pathMat[cur] = P_DELCLOSE; indexMat[cur] = (int)cur; }
            else { pathMat[cur] = P_INSCLOSE; indexMat[cur] = (int)cur; }
          }
          prevRowStart += prevRowLen; curRowStart += curRowLen; prevRowLen = curRowLen;
        }
        std::vector<int> path;                                          // :626-674 trace back
        int curMat = 0;
        long pos = matSize - 1, guard = 0;
        while (pos > 0) {
          if (++guard > 4 * matSize + 16) { *status |= 1; return finish(refined); }
          if (curMat == 0) {
            if (pathMat[pos] == P_DELCLOSE) curMat = 1;
            else if (pathMat[pos] == P_INSCLOSE) curMat = 2;
            else path.push_back(pathMat[pos]);
            pos = indexMat[pos];
          } else if (curMat == 1) {
            path.push_back(P_DOWN);
            curMat = (delPath[pos] == P_DELOPEN) ? 0 : 1;
            pos = delIndex[pos];
          } else {
            path.push_back(P_LEFT);
            curMat = (insPath[pos] == P_INSOPEN) ? 0 : 2;
            pos = insIndex[pos];
          }
        }
        path.push_back(P_DIAG);
        std::reverse(path.begin(), path.end());
        long qPath = qStart, tPath = tStart;
        size_t pi = 0;
        while (pi < path.size()) {                                     // :718-745 path -> blocks
          long blockLen = 0;
          while (pi < path.size() && path[pi] == P_DIAG) { blockLen++; pi++; }
          long tg = 0, qg = 0;
          if (pi < path.size()) {
            if (path[pi] == P_LEFT) while (pi < path.size() && path[pi] == P_LEFT) { qg++; pi++; }
            else if (path[pi] == P_DOWN) while (pi < path.size() && path[pi] == P_DOWN) { tg++; pi++; }
            else { *status |= 1; return finish(refined); }              // boundary arrow on the path: endless in the reference
          }
          refined.push_back({qPath, tPath, blockLen});
          qPath += blockLen + qg; tPath += blockLen + tg;
        }
      }
    }
    if (!usedAlt) endBlock++; else blocks[endBlock] = alt;             // :761-766
    startBlock = endBlock;
  }
  return finish(refined);
}
