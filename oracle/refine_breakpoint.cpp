// oracle/refine_breakpoint.cpp -- TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// CPU restatement of RefineBreakpoint (RefineBreakpoint.h:210-466) with RSdp :150-195, FindMax :197-207, StoreQScoreVect :118-146,
// TraceBack :91-116, PathToBlocks :49-82, PrependBlocks :6-27, AppendBlocks :29-46: the unaligned read bases between two adjacent
// segments of a split alignment (< 500) are aligned from both sides by a full DP (match 2, mismatch -2, gap -4), the two local
// alignments are cut where their summed score is best, and the new blocks are glued onto the segments.
// Parity status: PARITY UNPINNED -- the file needs Read / Genome / Alignment (htslib) for its main function; restated from the source.
#include "oracle_common.h"
#include <algorithm>
#include <string>
#include <vector>

namespace {
struct Blk { int q, t, len; };
enum { LEFT = 1, DOWN = 2, DIAG = 3 };

void rsdp(const std::string& q, const std::string& t, std::vector<int>& path, std::vector<int>& score, int mat, int mis, int indel) {   // :150-195
  const int qs = (int)q.size(), ts = (int)t.size(), row = qs + 1;
  path.assign((size_t)(qs + 1) * (ts + 1), -1);
  score.assign((size_t)(qs + 1) * (ts + 1), 0);
  for (int i = 1; i < qs + 1; i++) { path[i] = LEFT; score[i] = score[i - 1] + indel; }
  for (int i = 1; i < ts + 1; i++) { path[row * i] = DOWN; score[i * row] = score[(i - 1) * row] + indel; }
  for (int i = 0; i < ts; i++)
    for (int j = 0; j < qs; j++) {
      int diagScore = score[i * row + j] + (q[j] == t[i] ? mat : mis);
      int leftScore = score[(i + 1) * row + j] + indel, downScore = score[i * row + (j + 1)] + indel;
      int maxScore = std::max(diagScore, std::max(leftScore, downScore));
      score[(i + 1) * row + (j + 1)] = maxScore;
      path[(i + 1) * row + (j + 1)] = maxScore == diagScore ? DIAG : maxScore == leftScore ? LEFT : DOWN;
    }
}
int find_max(const std::vector<int>& score, int row, int& q, int& t) {   // :197-207
  if (score.empty()) { q = t = 0; return 0; }
  int index = (int)(std::max_element(score.begin(), score.end()) - score.begin());
  t = index / row - 1; q = index % row - 1;
  return score[index];
}
void store_q(const std::vector<int>& score, const std::vector<int>& path, int q, int t, int r, std::vector<int>& qv, std::vector<int>& index) {   // :118-146
  qv.assign(r - 1, 0); index.assign(r - 1, 0);
  int i = (t + 1) * r + q + 1;
  q++; t++;
  while (i > 0) {
    if (path[i] == DIAG || path[i] == LEFT) { qv[q - 1] = score[i]; index[q - 1] = i; }
    if (path[i] == DIAG) { q--; t--; }
    if (path[i] == LEFT) q--;
    if (path[i] == DOWN) t--;
    i = t * r + q;
  }
}
void trace_back(const std::vector<int>& path, int q, int t, int r, std::vector<int>& tb) {   // :91-116
  q++; t++;
  int i = t * r + q;
  while (q > 0 || t > 0) {
    if (path[i] == DIAG) { q--; t--; tb.push_back(DIAG); }
    if (path[i] == LEFT) { q--; tb.push_back(LEFT); }
    if (path[i] == DOWN) { t--; tb.push_back(DOWN); }
    i = t * r + q;
  }
  std::reverse(tb.begin(), tb.end());
}
void path_to_blocks(const std::vector<int>& path, std::vector<Blk>& blocks) {   // :49-82
  size_t i = 0; int q = 0, t = 0;
  while (i < path.size() && path[i] != DIAG && (path[i] == LEFT || path[i] == DOWN)) { if (path[i] == LEFT) q++; if (path[i] == DOWN) t++; i++; }
  while (i < path.size()) {
    int ml = 0, qs = q, ts = t;
    while (i < path.size() && path[i] == DIAG) { ml++; q++; t++; i++; }
    while (i < path.size() && (path[i] == LEFT || path[i] == DOWN)) { if (path[i] == LEFT) q++; if (path[i] == DOWN) t++; i++; }
    int match = std::min(q - qs, t - ts);
    if (match > 0) blocks.push_back(Blk{qs, ts, match});
    (void)ml;
  }
}
void prepend_blocks(std::vector<Blk>& src, std::vector<Blk>& dest) {   // :6-27
  if (src.empty()) return;
  if (dest.empty()) { dest = src; return; }
  int last = (int)src.size() - 1;
  if (src[last].t + src[last].len == dest[0].t && src[last].q + src[last].len == dest[0].q) {
    dest[0].t -= src[last].len; dest[0].q -= src[last].len; dest[0].len += src[last].len;
    src.resize(last);
  }
  dest.insert(dest.begin(), src.begin(), src.end());
}
void append_blocks(std::vector<Blk>& src, std::vector<Blk>& dest) {   // :29-46
  if (src.empty()) return;
  if (dest.empty()) { dest = src; return; }
  int last = (int)dest.size() - 1, srcStart = 0;
  if (dest[last].t + dest[last].len == src[0].t && dest[last].q + dest[last].len == src[0].q) { dest[last].len += src[0].len; srcStart = 1; }
  dest.insert(dest.end(), src.begin() + srcStart, src.end());
}
}  // namespace

// left / right: the two segments (blocks as (q,t,len) triples, strand, the read strand they are aligned on = Alignment::read, their chromosome and its
// length).  Outputs the two new block lists (capacity nL + 501 / nR + 501 triples).  Returns 1 if the junction was refined, 0 if left alone,
// -1 if the reference would read outside its inputs (a segment too close to the read start).
extern "C" int oracle_refine_breakpoint(int readLen, const int* lBlocks, int nL, int lStrand, const char* lRead, const char* lChrom, int lChromLen,
                                        const int* rBlocks, int nR, int rStrand, const char* rRead, const char* rChrom, int rChromLen, int* lOut, int* nLOut,
                                        int* rOut, int* nROut) {
  std::vector<Blk> L(nL), R(nR);
  for (int i = 0; i < nL; i++) L[i] = Blk{lBlocks[3 * i], lBlocks[3 * i + 1], lBlocks[3 * i + 2]};
  for (int i = 0; i < nR; i++) R[i] = Blk{rBlocks[3 * i], rBlocks[3 * i + 1], rBlocks[3 * i + 2]};
  auto put = [&](int ret) {
    *nLOut = (int)L.size(); *nROut = (int)R.size();
    for (size_t i = 0; i < L.size(); i++) { lOut[3 * i] = L[i].q; lOut[3 * i + 1] = L[i].t; lOut[3 * i + 2] = L[i].len; }
    for (size_t i = 0; i < R.size(); i++) { rOut[3 * i] = R[i].q; rOut[3 * i + 1] = R[i].t; rOut[3 * i + 2] = R[i].len; }
    return ret;
  };
  const int lqs = nL ? L[0].q : 0, lqe = nL ? L.back().q + L.back().len : 0, lts = nL ? L[0].t : 0, lte = nL ? L.back().t + L.back().len : 0;
  const int rqs = nR ? R[0].q : 0, rqe = nR ? R.back().q + R.back().len : 0, rts = nR ? R[0].t : 0, rte = nR ? R.back().t + R.back().len : 0;
  int flqe = lStrand == 0 ? lqe : readLen - lqs;
  int frqs = rStrand == 0 ? rqs : readLen - rqe;
  const int MAX_GAP = 500;
  if (!(frqs > flqe && frqs - flqe < MAX_GAP)) return put(0);
  const int span = frqs - flqe;
  std::string lq, lt, rq, rt;
  bool lPrefix = false, rPrefix = false;
  if (lStrand == 0) {
    if (lqe + span > readLen) return put(-1);
    lq.assign(lRead + lqe, span);
    int tSpan = std::min(lChromLen - lte, span);
    if (tSpan < 0) return put(-1);
    lt.assign(lChrom + lte, tSpan);
  } else {
    if (lqs - span < 0) return put(-1);
    lq.assign(lRead + (lqs - span), span);
    int ltExtEnd = lts, ltExtStart = std::max(0, ltExtEnd - span);
    lt.assign(lChrom + ltExtStart, ltExtEnd - ltExtStart);
    lPrefix = true;
    std::reverse(lq.begin(), lq.end()); std::reverse(lt.begin(), lt.end());
  }
  std::vector<int> lPath, lScore, rPath, rScore;
  rsdp(lq, lt, lPath, lScore, 2, -2, -4);
  if (rStrand == 0) {
    if (rqs - span < 0) return put(-1);
    rq.assign(rRead + (rqs - span), span);
    int rtSpan = std::min(rts, span);
    rt.assign(rChrom + (rts - rtSpan), rtSpan);
    std::reverse(rq.begin(), rq.end()); std::reverse(rt.begin(), rt.end());
    rPrefix = true;
  } else {
    if (rqe + span > readLen) return put(-1);
    rq.assign(rRead + rqe, span);
    int tSpan = span;
    if (rte + span >= rChromLen) tSpan = rChromLen - rte;
    if (tSpan < 0) return put(-1);
    rt.assign(rChrom + rte, tSpan);
  }
  rsdp(rq, rt, rPath, rScore, 2, -2, -4);
  int mlq, mlt, mrq, mrt;
  find_max(lScore, span + 1, mlq, mlt);
  find_max(rScore, span + 1, mrq, mrt);
  if (!(mlq < span - mrq)) {                                             // the two local alignments overlap on the read: cut where the sum is best
    std::vector<int> lqS, rqS, lqI, rqI;
    store_q(lScore, lPath, mlq, mlt, span + 1, lqS, lqI);
    store_q(rScore, rPath, mrq, mrt, span + 1, rqS, rqI);
    int maxScore = 0, maxL = 0, maxR = 0;
    for (int i = 0; i < (int)lqS.size(); i++)
      if (lqS[i] + rqS[lqS.size() - i - 1] > maxScore) { maxScore = lqS[i] + rqS[lqS.size() - i - 1]; maxL = i; maxR = (int)lqS.size() - i - 1; }
    mlq = maxL; mlt = lqI[maxL] / (span + 1) - 1; mrq = maxR; mrt = rqI[maxR] / (span + 1) - 1;
  }
  std::vector<int> ltb, rtb;
  trace_back(lPath, mlq, mlt, span + 1, ltb);
  trace_back(rPath, mrq, mrt, span + 1, rtb);
  std::vector<Blk> lB, rB;
  int lqStart, ltStart, rqStart, rtStart;
  if (lPrefix) { std::reverse(ltb.begin(), ltb.end()); lqStart = lqs - mlq - 1; ltStart = lts - mlt - 1; } else { lqStart = lqe; ltStart = lte; }
  path_to_blocks(ltb, lB);
  for (auto& b : lB) { b.q += lqStart; b.t += ltStart; }
  if (lPrefix) prepend_blocks(lB, L); else append_blocks(lB, L);
  if (rPrefix) { std::reverse(rtb.begin(), rtb.end()); rqStart = rqs - mrq - 1; rtStart = rts - mrt - 1; } else { rqStart = rqe; rtStart = rte; }
  path_to_blocks(rtb, rB);
  for (auto& b : rB) { b.q += rqStart; b.t += rtStart; }
  if (rPrefix) prepend_blocks(rB, R); else append_blocks(rB, R);
  return put(1);
}
