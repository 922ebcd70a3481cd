// oracle/aog.cpp -- TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// CPU restatement of AffineOneGapAlign (reference: AffineOneGapAlign.h:157-649).
// Parity status: PINNED -- checked bit-exact against the reference function
// itself compiled from /root/reference (oracle/ref_harness/aog_ref.cpp ->
// oracle/_ref/aog_ref) on the 22 input pairs of TestAffineOneGapAlign.cpp:19-68
// and on seeded random pairs (tests/test_oracle_pinning.py; committed outputs
// in tests/golden/aog_golden.json).
//
// The function is a banded linear-gap global DP done twice (a "prefix" band
// anchored at (0,0) and a "suffix" band anchored at (qLen,tLen)) that are joined
// by one free long gap.  Both score matrices are flat row-major arrays of
// (3+k+diag) rows x (2k+3) columns; the suffix matrix is addressed through a
// shifted origin, so some logical cells fall on neighbouring rows' slots.  The
// restatement keeps the same flat slot arithmetic so those reads see the same
// values.
#include "oracle_common.h"
#include <vector>
#include <algorithm>
#include <limits.h>

namespace {
enum { A_DONE = 0, A_LEFT = 1, A_DOWN = 2, A_DIAG = 3, A_BORDER = 4, A_GAPLEFT = 5, A_GAPDOWN = 6 };
const long MISSING_L = INT_MIN;  // AffineOneGapAlign.h:29

struct Flat {
  std::vector<long> score;
  std::vector<int> path;
  bool oob = false;
  void init(size_t n) { score.assign(n, MISSING_L); path.assign(n, -1); }
  bool ok(long i) { if (i < 0 || (size_t)i >= score.size()) { oob = true; return false; } return true; }
  long S(long i) { return ok(i) ? score[i] : MISSING_L; }
  int P(long i) { return ok(i) ? path[i] : -1; }
  void set(long i, long s, int p) { if (ok(i)) { score[i] = s; path[i] = p; } }
};
}  // namespace

// Returns the alignment score (AffineOneGapAlign.h:648).  blocks_out receives
// (qPos,tPos,len) triples in alignment order; *n_blocks their count (only up to
// cap triples are written).  *status: 0 ok, bit0 = a slot index left the
// matrices (undefined behaviour in the reference), bit1 = traceback did not
// terminate (infinite loop in the reference).
extern "C" int oracle_affine_one_gap_align(const char* q, int qLen, const char* t, int tLen, int m,
                                           int mm, int indel, int k, int* blocks_out, int cap,
                                           int* n_blocks, int* status) {
  *status = 0;
  const int diag = std::max(1, std::min(qLen, tLen));  // :162
  std::vector<int> qc(qLen + 1, 0), tc(tLen + 1, 0);   // :173-182
  for (int s = 0; s < qLen; s++) qc[s + 1] = oracle_code_n((unsigned char)q[s]);
  for (int s = 0; s < tLen; s++) tc[s + 1] = oracle_code_n((unsigned char)t[s]);
  std::vector<int> upMax(diag + 1, INT_MIN), upIdx(diag + 1, 0), loMax(diag + 1, INT_MIN),
      loIdx(diag + 1, 0);  // :183-191

  k = std::min(diag, k);  // :194
  bool top = true;
  if (diag + 2 * k >= std::max(qLen, tLen)) {  // :196-203
    k = 2 * k;
    top = false;
  }
  const int R = 2 * k + 3;                       // :207-209
  const long n = (long)(3 + k + diag) * R;       // :210
  Flat pre, suf;
  pre.init(n);
  suf.init(n);
  auto PI = [&](long i, long j) { return j * R + (i - j) + k + 1; };  // :12-17

  // ---- prefix boundary (:229-306) -- same write order as the reference
  for (int i = 1; i < k + 1; i++) pre.set(PI(i, 0), (long)indel * i, A_LEFT);
  for (int j = 1; j <= k + 1; j++) pre.set(PI(0, j), (long)indel * j, A_DOWN);
  pre.set(PI(0, 0), 0, A_DONE);
  if (qLen >= tLen) {
    for (int i = 0; i <= diag - k - 1; i++) pre.set(PI(i, i + k + 1), MISSING_L, A_BORDER);
    for (int i = 1; i < diag + k - 1; i++) pre.set(PI(i + k + 1, i), MISSING_L, A_BORDER);
    loMax[0] = 0; loIdx[0] = 0;
  }
  if (qLen <= tLen) {
    for (int j = 0; j < diag - 1; j++) pre.set(PI(j + k + 1, j), MISSING_L, A_BORDER);
    for (int j = 1; j < diag + k; j++) pre.set(PI(j - k - 1, j), MISSING_L, A_BORDER);
    upMax[0] = 0; upIdx[0] = 0;
  }
  const int qB = std::min(diag + k, qLen + 1);  // :309-310
  const int tB = std::min(diag + k, tLen + 1);
  // ---- prefix fill (:313-362)
  for (int j = 1; j < tB; j++) {
    for (int i = std::max(1, j - k); i < std::min(qB, j + k + 1); i++) {
      long sIns = pre.S(PI(i - 1, j)) + indel;
      long sDel = pre.S(PI(i, j - 1)) + indel;
      long sMat = pre.S(PI(i - 1, j - 1)) + (qc[i] == tc[j] ? m : mm);
      long best = std::max(sIns, std::max(sDel, sMat));
      int ar = (best == sIns) ? A_LEFT : (best == sDel) ? A_DOWN : A_DIAG;  // :331-339
      pre.set(PI(i, j), best, ar);
      if (i < qLen - k) {                       // :347-352 (ties -> largest i)
        if (j <= diag && best >= loMax[j]) { loMax[j] = (int)best; loIdx[j] = i; }
        else if (j > diag) *status |= 1;
      }
      if (j < tLen && i < diag + 1) {           // :353-360 (ties -> smallest j)
        if (best > upMax[i]) { upMax[i] = (int)best; upIdx[i] = j; }
      }
    }
  }

  std::vector<int> lens, ops;
  auto push = [&](int arrow) {                  // :532-538, :598-604
    if (ops.empty() || ops.back() != arrow) { lens.push_back(1); ops.push_back(arrow); }
    else lens.back()++;
  };
  int i, j;
  int result = -1;
  const long ITER_CAP = 4L * (qLen + tLen + 8);
  if (top) {
    // ---- suffix matrices (:409-518)
    const int qStart = std::max(0, qLen - diag), qEnd = qLen + 1;
    const int tStart = std::max(0, tLen - diag);
    const int tLow = std::max(0, tLen - diag - k - 1 - 1);
    const int qLow = std::max(0, qLen - diag - k - 1);
    const int tEnd = tLen + 1;
    auto SI = [&](long ii, long jj) {           // :19-27
      long a = ii - qLow, b = jj - tLow;
      return b * R + (a - b) + k + 1;
    };
    if (qLen >= tLen) {                         // :419-440
      for (i = qLow, j = 0; i < qStart + k + 1; i++) suf.set(SI(i, j), loMax[j], A_GAPLEFT);
      for (i = qLow, j = 1; i < qLow + diag; i++, j++) suf.set(SI(i, j), loMax[j], A_GAPLEFT);
      for (j = tStart + 1, i = qStart; j < tEnd - k; i++, j++) suf.set(SI(i + k + 1, j), MISSING_L, A_BORDER);
    }
    if (qLen <= tLen) {                         // :441-467
      for (j = tLow, i = qStart; j < tStart + k + 2; j++) suf.set(SI(i, j), upMax[0], A_GAPDOWN);
      for (j = tStart + 1, i = qStart + 1; j < tEnd; i++, j++) suf.set(SI(i, j - k - 1), upMax[i], A_GAPDOWN);
      for (j = tStart, i = qStart; j < tEnd - k - 1; i++, j++) suf.set(SI(i, j + k + 1), MISSING_L, A_BORDER);
    }
    for (j = tLow + 1; j < tEnd; j++) {         // :474-518
      int doff = diag + 1 - (tEnd - j);
      for (i = std::max(qLow + 1, qStart + doff - k); i < std::min(qEnd, qStart + doff + k + 1); i++) {
        long delClose = MISSING_L, insClose = MISSING_L;
        if (qLen >= tLen) delClose = loMax[j];
        if (tLen > qLen) insClose = upMax[i];
        long sIns = suf.S(SI(i - 1, j)) + indel;
        long sDel = suf.S(SI(i, j - 1)) + indel;
        long sMat = suf.S(SI(i - 1, j - 1)) + (qc[i] == tc[j] ? m : mm);
        long best = std::max(delClose, std::max(insClose, std::max(sIns, std::max(sDel, sMat))));
        int ar = -1;
        if (best == sIns) ar = A_LEFT;
        else if (best == sDel) ar = A_DOWN;
        else if (best == sMat) ar = A_DIAG;
        else if (best == delClose) ar = A_GAPLEFT;
        else if (best == insClose) ar = A_GAPDOWN;
        long slot = SI(i, j);
        if (suf.ok(slot)) { suf.score[slot] = best; if (ar >= 0) suf.path[slot] = ar; }
      }
    }
    // ---- suffix trace back (:523-580)
    i = qLen; j = tLen;
    int arrow = suf.P(SI(i, j));
    result = (int)suf.S(SI(i, j));
    long it = 0;
    while (arrow != A_DONE && arrow != A_GAPDOWN && arrow != A_GAPLEFT && i >= 0 && j >= 0) {
      if (++it > ITER_CAP) { *status |= 2; break; }
      if (arrow != A_DIAG && arrow != A_LEFT && arrow != A_DOWN) { *status |= 2; break; }  // endless in the reference
      push(arrow);
      if (arrow == A_DIAG) { i--; j--; }
      else if (arrow == A_LEFT) i--;
      else if (arrow == A_DOWN) j--;
      if (i >= 0 && j >= 0) arrow = suf.P(SI(i, j));
    }
    if (arrow == A_GAPDOWN) {
      lens.push_back(j - upIdx[i]); ops.push_back(arrow); j = upIdx[i];
    }
    if (arrow == A_GAPLEFT) {
      lens.push_back(i - loIdx[j]); ops.push_back(arrow); i = loIdx[j];
    }
  } else {                                      // :582-586
    i = qB - 1; j = tB - 1;
    result = (int)pre.S(PI(i, j));
  }
  // ---- prefix trace back (:589-629)
  {
    int arrow = (i >= 0 && j >= 0) ? pre.P(PI(i, j)) : A_DONE;
    long it = 0;
    while (arrow != A_BORDER && arrow != A_DONE && i >= 0 && j >= 0) {
      if (++it > ITER_CAP) { *status |= 2; break; }
      if (arrow == A_GAPLEFT || arrow == A_GAPDOWN) { push(arrow); break; }
      if (arrow != A_DIAG && arrow != A_LEFT && arrow != A_DOWN) { *status |= 2; break; }  // endless in the reference
      push(arrow);
      if (arrow == A_DIAG) { i--; j--; }
      else if (arrow == A_LEFT) i--;
      else j--;
      if (i < 0 || j < 0) break;                // the reference reads one stale slot, then exits
      arrow = pre.P(PI(i, j));
    }
  }
  if (pre.oob || suf.oob) *status |= 1;
  // ---- ops -> gapless blocks (:630-647)
  int qPos = 0, tPos = 0, nb = 0;
  for (size_t x = lens.size(); x > 0; x--) {
    int op = ops[x - 1], len = lens[x - 1];
    if (op == A_LEFT || op == A_GAPLEFT) qPos += len;
    else if (op == A_DOWN || op == A_GAPDOWN) tPos += len;
    else if (op == A_DIAG) {
      if (nb < cap) { blocks_out[3 * nb] = qPos; blocks_out[3 * nb + 1] = tPos; blocks_out[3 * nb + 2] = len; }
      nb++;
      qPos += len; tPos += len;
    }
  }
  *n_blocks = nb;
  return result;
}
