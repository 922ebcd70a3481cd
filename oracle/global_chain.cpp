// oracle/global_chain.cpp -- TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// CPU restatement of GlobalChain<Fragment, Endpoint> (GlobalChain.h:85-189; Endpoint :10-60, FragmentSetToEndpoints :65-83) over the priority search tree of
// PrioritySearchTree.h (CreateTree :65-104, Activate :196-221, FindIndexOfMaxPoint :106-143, :223-231) -- the chaining component `north_star` names; `lra.cpp`
// does not reach it (the include is commented out, LocalRefineAlignment.h:13).  Kept literal, including what looks unintended: the tree is keyed by y but built
// over endpoints sorted by (x, y); Activate keeps descending by the ORIGINAL point's key and score after it swapped a displaced point in; leaves never hold a
// maxScoreNode.  Keys are compared as `unsigned int` (Endpoint::KeyType).
// Parity status: PINNED -- oracle/ref_harness/globalchain_ref.cpp compiles the reference headers in place; tests/golden/globalchain_golden.json holds its
// answers (TestGlobalChain.cpp's own input among them).
#include "oracle_common.h"
#include <algorithm>
#include <vector>

namespace {
struct Ep { int x, y, fragment, side, score; };             // side 0 = Start, 1 = End
struct Vx { unsigned left = 0, right = 0, leaf = 0, medianKey = 0, maxKey = 0; int pointIndex = -1, maxScoreNode = -1; };
struct Pst {
  std::vector<Vx> t;
  unsigned create(const std::vector<Ep>& p, int start, int end, unsigned& it) {
    const int median = (end + start) / 2;
    const unsigned cur = it;
    t[cur].medianKey = (unsigned)p[median].y;
    if (end == start) { t[cur].pointIndex = start; return t[cur].medianKey; }
    if (end - start == 1) { t[cur].leaf = 1; t[cur].medianKey = (unsigned)p[start].y; t[cur].pointIndex = start; return t[cur].medianKey; }
    t[cur].leaf = 0;
    t[cur].left = ++it;
    const unsigned lk = create(p, start, median, it);
    t[cur].medianKey = lk;
    t[cur].right = ++it;
    const unsigned rk = create(p, median, end, it);
    t[cur].maxKey = rk;
    return rk;
  }
  void activate(const std::vector<Ep>& p, int pointIndex) {
    const int pointScore = p[pointIndex].score;
    unsigned cur = 0;
    const unsigned key = (unsigned)p[pointIndex].y;
    while (pointIndex != -1 && t[cur].leaf == 0) {
      if (t[cur].maxScoreNode == -1 || p[t[cur].maxScoreNode].score <= pointScore) { const int tmp = t[cur].maxScoreNode; t[cur].maxScoreNode = pointIndex; pointIndex = tmp; }
      cur = key <= t[cur].medianKey ? t[cur].left : t[cur].right;
    }
  }
  int find(unsigned cur, const std::vector<Ep>& p, unsigned maxKey, int& maxVal, int& maxIdx) {
    if (t[cur].maxScoreNode == -1) return 0;
    if ((unsigned)p[t[cur].maxScoreNode].y < maxKey) {
      if (p[t[cur].maxScoreNode].score > maxVal) { maxVal = p[t[cur].maxScoreNode].score; maxIdx = t[cur].maxScoreNode; return 1; }
      return 0;
    }
    if (!t[cur].leaf) {
      if (maxKey <= t[cur].medianKey) return find(t[cur].left, p, maxKey, maxVal, maxIdx);
      const int a = find(t[cur].left, p, maxKey, maxVal, maxIdx), b = find(t[cur].right, p, maxKey, maxVal, maxIdx);
      return a || b;
    }
    return 0;
  }
};
}  // namespace

// n fragments (xl, yl, xh, yh, score in / out); prev out; chain out (capacity n).  Returns the chain length.
extern "C" int oracle_global_chain(int n, const int* xl, const int* yl, const int* xh, const int* yh, int* score, int* prev, int* chain) {
  for (int i = 0; i < n; i++) prev[i] = -1;
  if (n == 0) return 0;
  std::vector<Ep> ep(2 * (size_t)n);
  for (int i = 0; i < n; i++) { ep[2 * i] = Ep{xl[i], yl[i], i, 0, 0}; ep[2 * i + 1] = Ep{xh[i], yh[i], i, 1, 0}; }
  std::sort(ep.begin(), ep.end(), [](const Ep& a, const Ep& b) { return a.x != b.x ? a.x < b.x : a.y < b.y; });     // Endpoint::LessThan :37-47
  Pst pst;
  pst.t.resize(ep.size() * 2 - 1);
  unsigned it = 0;
  pst.create(ep, 0, (int)ep.size(), it);
  unsigned maxEp = 0; bool found = false;
  for (unsigned p = 0; p < ep.size(); p++) {
    if (ep[p].side == 0) {
      int mi = 0, mv = -1;
      if (pst.t[0].maxScoreNode != -1 && pst.find(0, ep, (unsigned)ep[p].y, mv, mi)) {
        prev[ep[p].fragment] = ep[mi].fragment;
        score[ep[p].fragment] = score[ep[mi].fragment] + score[ep[p].fragment];
      } else prev[ep[p].fragment] = -1;
    } else {
      ep[p].score = score[ep[p].fragment];
      pst.activate(ep, (int)p);
      if (!found || score[ep[maxEp].fragment] < score[ep[p].fragment]) { maxEp = p; found = true; }
    }
  }
  if (!found) return 0;
  int k = 0;
  for (int f = ep[maxEp].fragment; f != -1 && k < n; f = prev[f]) chain[k++] = f;
  std::reverse(chain, chain + k);
  return k;
}
