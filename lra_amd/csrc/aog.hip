// lra_amd/csrc/aog.hip -- batched banded one-gap seed-extension DP for gfx950.
//
// Replaces AffineOneGapAlign (reference: AffineOneGapAlign.h:157-649) for a batch of
// independent (q, t, k) problems: a banded linear-gap global DP filled from (0,0)
// ("prefix" band) and, when the length difference exceeds the band ("alignTop",
// :194-203), a second band anchored at (qLen,tLen) ("suffix") joined to the first by
// one free long gap chosen through per-row / per-column running maxima (:347-360).
//
// MI355X mapping
//   * one 64-lane wavefront owns one problem; its two flat score/arrow matrices, the
//     per-diagonal maxima and the 2-bit sequence codes live in that wave's LDS slice
//     (classes A: <= 10 KB, 4 waves per workgroup; B: <= 64 KB, 1 wave per workgroup),
//     or in an HBM scratch slot for the rare larger problem (class C);
//   * the band is swept by ANTI-DIAGONALS: the <= k+1 cells of one anti-diagonal are
//     independent, one lane each; consecutive sweeps are ordered by wavefront-local
//     fences only (no s_barrier: the waves of a workgroup run different problems);
//   * the per-row / per-column maxima (ties: last row resp. first column, as the
//     reference's >= / > updates give) are taken after the sweep, one lane per row/column;
//   * trace back is a serial walk by lane 0 over 1-byte arrows; blocks are emitted in
//     walk order and reversed cooperatively.
//   Scores are int32 with MISSING = -2^30 (the reference uses INT_MIN in 64-bit cells,
//   :29,:146); every decision compares values whose offsets from MISSING stay below
//   2^28 (checked per problem), so all orderings and equalities are preserved, and the
//   returned score is mapped back to the reference's 32-bit truncation.
//   Slot arithmetic (:12-27) is kept exactly: the suffix matrix is addressed through a
//   shifted origin and some out-of-band neighbour reads land on other rows' slots.
#include "common.h"
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include <type_traits>

namespace {

constexpr int MISS = -(1 << 30);
enum { A_DONE = 0, A_LEFT = 1, A_DOWN = 2, A_DIAG = 3, A_BORDER = 4, A_GAPLEFT = 5, A_GAPDOWN = 6 };

constexpr int CLASS_A_BYTES = 10 * 1024;  // per wave, 4 waves per workgroup
constexpr int CLASS_B_BYTES = 64 * 1024;  // per wave, 1 wave per workgroup
constexpr int CLASS_M1_BYTES = 20 * 1024; // the same with a smaller LDS request: 8 / 5 problems per CU instead of 2
constexpr int CLASS_M2_BYTES = 32 * 1024;
constexpr int LN_MAX = 24, LN_ROWW = 2;
constexpr int LN_ARROW_WORDS = LN_MAX * LN_ROWW, LN_PREV_WORDS = LN_MAX + 2, LN_CODE_BYTES = LN_MAX + 1;
constexpr int LN_BYTES = 64 * (4 * LN_ARROW_WORDS + 4 * LN_PREV_WORDS + 2 * LN_CODE_BYTES);
constexpr int NCLS = 19;
constexpr int CLASS_S_BYTES = 2560;       // per 16-lane group: 4 problems per wave, 16 per workgroup (anti-diagonals <= 16 cells)

__device__ __forceinline__ int code_n(unsigned char c) {  // SeqUtils.h:42-75 (seqMapN)
  if (c < 8) return c & 3;
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
  }
}

struct Geo {
  int qLen, tLen, diag, k, R, qB, tB;
  bool top;
  int n;      // slots per matrix in the reference (bounds checks)
  int nUsed;  // prefix-matrix slots that can ever be read back: rows j < tB when only the prefix band is used
};

__device__ __forceinline__ bool make_geo(int qLen, int tLen, int k0, Geo& g) {
  g.qLen = qLen; g.tLen = tLen;
  g.diag = max(1, min(qLen, tLen));                 // :162
  int k = min(g.diag, k0);                          // :194
  g.top = true;
  if (g.diag + 2 * k >= max(qLen, tLen)) { k *= 2; g.top = false; }  // :196-203
  g.k = k;
  g.R = 2 * k + 3;                                  // :207-209
  long n = (long)(3 + k + g.diag) * g.R;            // :210
  g.qB = min(g.diag + k, qLen + 1);                 // :309-310
  g.tB = min(g.diag + k, tLen + 1);
  if (n > (1L << 28)) { g.n = 0; g.nUsed = 0; return false; }
  g.n = (int)n;
  g.nUsed = g.top ? g.n : min(g.n, g.tB * g.R);
  return true;
}

__host__ __device__ __forceinline__ long align4(long x) { return (x + 3) & ~3L; }

// bytes of working memory one problem needs (same carve order as carve())
__device__ __forceinline__ long need_bytes(const Geo& g) {
  long suf = g.top ? (4L * g.n + align4(g.n)) : 0;
  return 4L * g.nUsed + align4(g.nUsed) + suf + 16L * (g.diag + 1) + align4(g.qLen + 1) + align4(g.tLen + 1);
}

struct Work {
  int* sPre; int* sSuf; int* loMax; int* loIdx; int* upMax; int* upIdx;
  signed char* pPre; signed char* pSuf; unsigned char* qc; unsigned char* tc;
};

template <typename BytePtr>
__device__ __forceinline__ Work carve(BytePtr base, const Geo& g) {
  Work w;
  auto* p = base;
  w.sPre = (int*)p; p += 4L * g.nUsed;
  w.sSuf = (int*)p; if (g.top) p += 4L * g.n;
  w.loMax = (int*)p; p += 4L * (g.diag + 1);
  w.loIdx = (int*)p; p += 4L * (g.diag + 1);
  w.upMax = (int*)p; p += 4L * (g.diag + 1);
  w.upIdx = (int*)p; p += 4L * (g.diag + 1);
  w.pPre = (signed char*)p; p += align4(g.nUsed);
  w.pSuf = (signed char*)p; if (g.top) p += align4(g.n);
  w.qc = (unsigned char*)p; p += align4(g.qLen + 1);
  w.tc = (unsigned char*)p;
  return w;
}

// Orders this wave's earlier LDS / global writes before its later reads.  Workgroup-scope
// fences lower to s_waitcnt only (no barrier, no cache maintenance): one wave's memory
// instructions are serviced in order by its CU's LDS and L1.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// the same for LDS traffic only: a wave's LDS instructions execute in order, so nothing has to be waited for -- in particular not the wave's outstanding
// global stores, which wave_sync() would drain
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct Problem {
  const char* q; const char* t;
  int qLen, tLen, k0, m, mm, indel;
};

__device__ __forceinline__ int dpp_from_prev_lane(int v) { return __builtin_amdgcn_update_dpp(MISS, v, 0x138, 0xf, 0xf, false); }   // wave_shr:1 -- lane n reads lane n - 1
__device__ __forceinline__ int dpp_from_next_lane(int v) { return __builtin_amdgcn_update_dpp(MISS, v, 0x130, 0xf, 0xf, false); }   // wave_shl:1 -- lane n reads lane n + 1

// The sweep of solve_reg (and of solve()'s HBM-resident problems): returns the corner cell's score in every lane of the group.  P: the arrows (LDS or HBM); qc / tc: the
// sequence codes in LDS.  Two steps per iteration (even diagonals, then odd ones, or the other way round), the codes of a lane's next cell on a diagonal loaded one
// round ahead (a cell's successor on its diagonal is (i + 1, j + 1)), and the boundary values looked at only while an anti-diagonal can still touch row / column 0.
template <int G, typename ArrowPtr, typename CodePtr>
__device__ __forceinline__ int reg_fill(int lane, int gbase, const Geo& g, int m, int mm, int indel, ArrowPtr P, CodePtr qc, CodePtr tc) {
  const int qLen = g.qLen, tLen = g.tLen, k = g.k, R = g.R, diag = g.diag, qB = g.qB, tB = g.tB;
  // what the boundary / rail stores of solve() leave in cell (i, j) before the sweep (last store wins; everything else is MISS)
  auto pre = [&](int i, int j) -> int {
    int v = MISS;
    if (j == 0 && i >= 1 && i < k + 1) v = indel * i;
    if (i == 0 && j >= 1 && j <= k + 1) v = indel * j;
    if (i == 0 && j == 0) v = 0;
    if (qLen >= tLen) { if (j == i + k + 1 && i <= diag - k - 1) v = MISS; if (i == j + k + 1 && j >= 1 && j < diag + k - 1) v = MISS; }
    if (qLen <= tLen) { if (i == j + k + 1 && j < diag - 1) v = MISS; if (j == i + k + 1 && j >= 1 && j < diag + k) v = MISS; }
    return v;
  };
  const int Wd = 2 * k + 3;
  const int sLast = max((qB - 1) + (tB - 1), 0);
  const int ci = qB - 1, cj = tB - 1;
  int vE = MISS, vO = MISS;
  const bool firstE = ((0 - k - 1) & 1) == 0;                             // step 0 is an even-diagonal step
  // Per lane and parity, fixed for the whole sweep: the diagonal d, the steps [sLo, sHi] at which its cell is one the sweep computes
  // (1 <= i <= qB - 1, 1 <= j <= tB - 1, |d| <= k: what solve()'s jlo / jhi bounds say, per diagonal), and, advanced by one per own step, the cell (i, j),
  // its arrow's address and whether the codes of the lane's NEXT cell on the diagonal match (loaded a round ahead).
  const int dE = 2 * lane - k - 1, dO = dE + 1;
  const bool okE = 2 * lane < Wd && dE >= -k && dE <= k, okO = 2 * lane + 1 < Wd && dO >= -k && dO <= k;
  const int sLoE = okE ? 2 + abs(dE) : INT_MAX, sHiE = min(2 * (qB - 1) - dE, 2 * (tB - 1) + dE);
  const int sLoO = okO ? 2 + abs(dO) : INT_MAX, sHiO = min(2 * (qB - 1) - dO, 2 * (tB - 1) + dO);
  const int s0E = firstE ? 0 : 1, s0O = firstE ? 1 : 0;
  int iE = (s0E + dE) >> 1, jE = s0E - iE, iO = (s0O + dO) >> 1, jO = s0O - iO;
  int adE = jE * R + 2 * lane, adO = jO * R + 2 * lane + 1;
  auto codes_match = [&](int i, int j) -> bool { return qc[min(max(i, 0), qLen)] == tc[min(max(j, 0), tLen)]; };   // (a cell whose codes are used has 1 <= i <= qLen, 1 <= j <= tLen)
  bool eqE = codes_match(iE, jE), eqO = codes_match(iO, jO);
  auto step = [&](int s, auto isE) {
    constexpr bool E = decltype(isE)::value;
    const int nb1 = E ? dpp_from_prev_lane(vO) : dpp_from_next_lane(vE);   // diagonal dd - 1 (even step) resp. dd + 1 (odd step), one step ago
    const int i = E ? iE : iO, j = E ? jE : jO;
    const bool nxt = codes_match(i + 1, j + 1);                           // (in flight until this lane's next step on this diagonal, two steps on)
    int v = MISS;
    if (s >= (E ? sLoE : sLoO) && s <= (E ? sHiE : sHiO)) {
      const int sIns = (E ? nb1 : vE) + indel;                            // (i - 1, j): diagonal dd - 1
      const int sDel = (E ? vO : nb1) + indel;                            // (i, j - 1): diagonal dd + 1
      const int sMat = (E ? vE : vO) + ((E ? eqE : eqO) ? m : mm);        // (i - 1, j - 1): this diagonal, two steps ago
      const int best = max(sIns, max(sDel, sMat));
      v = best;
      P[E ? adE : adO] = (signed char)((best == sIns) ? A_LEFT : (best == sDel) ? A_DOWN : A_DIAG);
    } else if (s <= k + 1) {                                              // (a boundary value sits in row or column 0: i + j <= k + 1; the rails hold MISS)
      if (2 * lane + (E ? 0 : 1) < Wd && i >= 0 && j >= 0) v = pre(i, j);
    }
    if (E) { vE = v; eqE = nxt; iE++; jE++; adE += R; } else { vO = v; eqO = nxt; iO++; jO++; adO += R; }
  };
  int s = 0;
  bool lastE;
  if (firstE) {
    for (; s + 1 <= sLast; s += 2) { step(s, std::true_type{}); step(s + 1, std::false_type{}); }
    lastE = false;
    if (s <= sLast) { step(s, std::true_type{}); lastE = true; }
  } else {
    for (; s + 1 <= sLast; s += 2) { step(s, std::false_type{}); step(s + 1, std::true_type{}); }
    lastE = true;
    if (s <= sLast) { step(s, std::false_type{}); lastE = false; }
  }
  // the corner (ci, cj) is the last step's cell on diagonal ci - cj
  const int ddc = ci - cj + k + 1;
  const int corner = lastE ? vE : vO;
  return (ci >= 0 && cj >= 0 && ddc >= 0 && ddc < Wd && ((ddc & 1) == 0) == lastE) ? __shfl(corner, gbase + (ddc >> 1)) : MISS;
}

template <int G, typename BytePtr>
__device__ __forceinline__ void solve(int wlane, const Problem& pr, const Geo& g, BytePtr mem, int* out_score,
                      int* out_nb, int* blocks, long cap, int* out_status, int* roll = nullptr, int chunkBytes = 8192) {
  const int qLen = g.qLen, tLen = g.tLen, k = g.k, R = g.R, diag = g.diag, n = g.n, nUsed = g.nUsed;
  const int m = pr.m, mm = pr.mm, indel = pr.indel;
  const int lane = wlane & (G - 1), gbase = wlane - lane;   // G lanes work on this problem
  Work w = carve(mem, g);
  int status = 0;
  auto PI = [&](int i, int j) { return j * R + (i - j) + k + 1; };  // :12-17
  auto inb = [&](int s) { return s >= 0 && s < n; };
#define PSET(slot, sc, ar)                                                         \
  do { int s__ = (slot); if (inb(s__)) { if (s__ < nUsed) { if (!rolling) w.sPre[s__] = (sc); w.pPre[s__] = (ar); } } else status |= LRA_ST_OOB_SLOT; } while (0)
#define SSET(slot, sc, ar)                                                         \
  do { int s__ = (slot); if (inb(s__)) { w.sSuf[s__] = (sc); w.pSuf[s__] = (ar); } else status |= LRA_ST_OOB_SLOT; } while (0)

  // ---- codes + clear (:173-219)
  for (int x = lane; x <= qLen; x += G) w.qc[x] = x ? code_n((unsigned char)pr.q[x - 1]) : 0;
  for (int x = lane; x <= tLen; x += G) w.tc[x] = x ? code_n((unsigned char)pr.t[x - 1]) : 0;
  // roll != NULL (HBM-resident problems that only use the prefix band): the prefix SCORES live in three rotating anti-diagonal windows in LDS (a cell reads its
  // two predecessors' anti-diagonals only, and without the suffix band nothing reads a score again but the corner's); HBM keeps the arrows.  The matrices of
  // such a problem are ~1 MB and every cell was written twice (fill + sweep): 5 B per cell -> 1 B.
  // (with the band's diagonals two to a lane the sweep keeps the scores in registers -- reg_fill -- and the codes may stay in the HBM work area when they outgrow LDS)
  const bool codesInLds = qLen + tLen + 2 <= 4096;
  const bool rolling = roll != nullptr && !g.top && (k + 2 <= G || (2 * k + 3 <= 256 && codesInLds));
  // ... and the sequence codes sit behind the windows (roll[768 ..], one byte each), so that a sweep step touches HBM only to store its arrows
  unsigned char* lq = (unsigned char*)(roll + 768); unsigned char* lt = lq + (qLen + 1);
  if (rolling && codesInLds) {
    for (int x = lane; x <= qLen; x += G) lq[x] = x ? code_n((unsigned char)pr.q[x - 1]) : 0;
    for (int x = lane; x <= tLen; x += G) lt[x] = x ? code_n((unsigned char)pr.t[x - 1]) : 0;
  }
  if (rolling) {                                                          // the arrows' "never stored" value, 16 bytes per lane and store (pPre is 4-byte aligned)
    const long head = min((long)nUsed, (long)((16 - ((uintptr_t)&w.pPre[0] & 15)) & 15));
    for (long x = lane; x < head; x += G) w.pPre[x] = -1;
    const long body = (nUsed - head) >> 4;
    int4* p16 = (int4*)(&w.pPre[0] + head);
    for (long x = lane; x < body; x += G) p16[x] = make_int4(-1, -1, -1, -1);
    for (long x = head + (body << 4) + lane; x < nUsed; x += G) w.pPre[x] = -1;
  }
  else for (int x = lane; x < nUsed; x += G) { w.sPre[x] = MISS; w.pPre[x] = -1; }
  if (g.top)
    for (int x = lane; x < n; x += G) { w.sSuf[x] = MISS; w.pSuf[x] = -1; }
  for (int x = lane; x <= diag; x += G) { w.loMax[x] = MISS; w.loIdx[x] = 0; w.upMax[x] = MISS; w.upIdx[x] = 0; }
  wave_sync();
  // ---- prefix boundary (:229-241), then rails (:248-306); the rails overwrite (0,k+1)
  for (int i = 1 + lane; i < k + 1; i += G) PSET(PI(i, 0), indel * i, A_LEFT);
  for (int j = 1 + lane; j <= k + 1; j += G) PSET(PI(0, j), indel * j, A_DOWN);
  if (lane == 0) PSET(PI(0, 0), 0, A_DONE);
  wave_sync();
  if (qLen >= tLen) {
    for (int i = lane; i <= diag - k - 1; i += G) PSET(PI(i, i + k + 1), MISS, A_BORDER);
    for (int i = 1 + lane; i < diag + k - 1; i += G) PSET(PI(i + k + 1, i), MISS, A_BORDER);
  }
  if (qLen <= tLen) {
    for (int j = lane; j < diag - 1; j += G) PSET(PI(j + k + 1, j), MISS, A_BORDER);
    for (int j = 1 + lane; j < diag + k; j += G) PSET(PI(j - k - 1, j), MISS, A_BORDER);
  }
  wave_sync();
  // ---- prefix fill by anti-diagonals s = i + j (:313-339)
  const int qB = g.qB, tB = g.tB;
  int rollResult = MISS;
  if (rolling) {
    // what the boundary / rail stores above leave in cell (i, j) before the sweep (last store wins; everything else is MISS)
    auto pre = [&](int i, int j) -> int {
      int v = MISS;
      if (j == 0 && i >= 1 && i < k + 1) v = indel * i;
      if (i == 0 && j >= 1 && j <= k + 1) v = indel * j;
      if (i == 0 && j == 0) v = 0;
      if (qLen >= tLen) { if (j == i + k + 1 && i <= diag - k - 1) v = MISS; if (i == j + k + 1 && j >= 1 && j < diag + k - 1) v = MISS; }
      if (qLen <= tLen) { if (i == j + k + 1 && j < diag - 1) v = MISS; if (j == i + k + 1 && j >= 1 && j < diag + k) v = MISS; }
      return v;
    };
    const int Wd = 2 * k + 3;
    const int sLast = (qB - 1) + (tB - 1);
    const bool inRegs = k + 2 <= G;                                      // the band's diagonals fit the lanes two apiece: scores in registers (reg_fill)
    if (inRegs) rollResult = codesInLds ? reg_fill<G>(lane, gbase, g, m, mm, indel, w.pPre, lq, lt) : reg_fill<G>(lane, gbase, g, m, mm, indel, w.pPre, w.qc, w.tc);
    for (int s = 0; !inRegs && s <= max(sLast, 0); s++) {
      int* cur = roll + (s % 3) * 256; const int* p1 = roll + ((s + 2) % 3) * 256; const int* p2 = roll + ((s + 1) % 3) * 256;
      int jlo = max(1, max(s - qB + 1, (s - k + 1) >> 1));
      if (s - k < 0) jlo = max(1, s - qB + 1);
      const int jhi = min(tB - 1, min(s - 1, (s + k) >> 1));
      for (int dd = lane; dd < Wd; dd += G) {
        const int d = dd - k - 1;                                          // i - j
        int v = MISS;
        if (((s + d) & 1) == 0) {
          const int i = (s + d) >> 1, j = s - i;
          if (i >= 0 && j >= 0) {
            if (s >= 2 && j >= jlo && j <= jhi && d >= -k && d <= k) {
              const int sIns = p1[dd - 1] + indel, sDel = p1[dd + 1] + indel;
              const int sMat = p2[dd] + (lq[i] == lt[j] ? m : mm);
              const int best = max(sIns, max(sDel, sMat));
              const int ar = (best == sIns) ? A_LEFT : (best == sDel) ? A_DOWN : A_DIAG;
              v = best;
              w.pPre[PI(i, j)] = (signed char)ar;
            } else v = pre(i, j);
            if (i == qB - 1 && j == tB - 1) rollResult = v;
          }
        }
        cur[dd] = v;
      }
      wave_sync_lds();
    }
    wave_sync();
    if (!inRegs) { const int ddc = (qB - 1) - (tB - 1) + k + 1; rollResult = (ddc >= 0 && ddc < Wd) ? __shfl(rollResult, gbase + (ddc % G)) : MISS; }   // the lane that owns the corner's diagonal
  }
  for (int s = 2; !rolling && s <= (qB - 1) + (tB - 1); s++) {
    int jlo = max(1, max(s - qB + 1, (s - k + 1) >> 1));   // ceil((s-k)/2), s-k may be < 0
    if (s - k < 0) jlo = max(1, s - qB + 1);
    int jhi = min(tB - 1, min(s - 1, (s + k) >> 1));
    for (int j = jlo + lane; j <= jhi; j += G) {
      int i = s - j;
      int sIns = w.sPre[PI(i - 1, j)] + indel;
      int sDel = w.sPre[PI(i, j - 1)] + indel;
      int sMat = w.sPre[PI(i - 1, j - 1)] + (w.qc[i] == w.tc[j] ? m : mm);
      int best = max(sIns, max(sDel, sMat));
      int ar = (best == sIns) ? A_LEFT : (best == sDel) ? A_DOWN : A_DIAG;   // :331-339
      int slot = PI(i, j);
      w.sPre[slot] = best;
      w.pPre[slot] = (signed char)ar;
    }
    wave_sync();
  }

  int ti, tj;        // trace-back cursor
  int result;
  long nb = 0;       // blocks written (walk order)
  const long iter_cap = 4L * (qLen + tLen + 8);
  // A block is a maximal run of diagonal arrows; (i,j) after the run is its start.
  int run = 0;
  auto flush = [&](int i_after, int j_after) {
    if (run > 0) {
      if (nb < cap) { blocks[3 * nb] = i_after; blocks[3 * nb + 1] = j_after; blocks[3 * nb + 2] = run; }
      else status |= LRA_ST_CAPACITY;
      nb++;
      run = 0;
    }
  };

  if (g.top) {
    // ---- per-row / per-column maxima of the prefix band (:347-360)
    //   loMax[j] = max over rows i < qLen-k of column j, ties -> LARGEST i   (>=)
    //   upMax[i] = max over columns j < tLen of row i (i <= diag), ties -> SMALLEST j (>)
    for (int j = 1 + lane; j < tB && j <= diag; j += G) {
      int bm = MISS, bi = 0;
      int ihi = min(qB, j + k + 1);
      for (int i = max(1, j - k); i < ihi && i < qLen - k; i++) {
        int v = w.sPre[PI(i, j)];
        if (v >= bm) { bm = v; bi = i; }
      }
      w.loMax[j] = bm; w.loIdx[j] = bi;
    }
    for (int i = 1 + lane; i <= diag && i < qB; i += G) {
      int bm = MISS, bj = 0;
      int jhi = min(tB - 1, i + k);
      for (int j = max(1, i - k); j <= jhi && j < tLen; j++) {
        int v = w.sPre[PI(i, j)];
        if (v > bm) { bm = v; bj = j; }
      }
      w.upMax[i] = bm; w.upIdx[i] = bj;
    }
    if (lane == 0) {
      if (qLen >= tLen) { w.loMax[0] = 0; w.loIdx[0] = 0; }   // :275-276
      if (qLen <= tLen) { w.upMax[0] = 0; w.upIdx[0] = 0; }   // :300-301
    }
    wave_sync();
    // ---- suffix boundary (:409-467), loop by loop in the reference's order
    const int qStart = max(0, qLen - diag);
    const int tStart = max(0, tLen - diag);
    const int tLow = max(0, tLen - diag - k - 1 - 1);
    const int qLow = max(0, qLen - diag - k - 1);
    const int tEnd = tLen + 1;
    auto SI = [&](int ii, int jj) {                               // :19-27
      int a = ii - qLow, b = jj - tLow;
      return b * R + (a - b) + k + 1;
    };
    if (qLen >= tLen) {
      for (int i = qLow + lane; i < qStart + k + 1; i += G) SSET(SI(i, 0), w.loMax[0], A_GAPLEFT);
      wave_sync();
      for (int j = 1 + lane; j <= diag; j += G) SSET(SI(qLow + j - 1, j), w.loMax[j], A_GAPLEFT);
      wave_sync();
      for (int j = tStart + 1 + lane; j < tEnd - k; j += G) SSET(SI(qStart + (j - tStart - 1) + k + 1, j), MISS, A_BORDER);
      wave_sync();
    }
    if (qLen <= tLen) {
      for (int j = tLow + lane; j < tStart + k + 2; j += G) SSET(SI(qStart, j), w.upMax[0], A_GAPDOWN);
      wave_sync();
      for (int j = tStart + 1 + lane; j < tEnd; j += G) {
        int i = qStart + 1 + (j - tStart - 1);
        SSET(SI(i, j - k - 1), (i <= diag ? w.upMax[i] : MISS), A_GAPDOWN);
      }
      wave_sync();
      for (int j = tStart + lane; j < tEnd - k - 1; j += G) SSET(SI(qStart + (j - tStart), j + k + 1), MISS, A_BORDER);
      wave_sync();
    }
    // ---- suffix fill by anti-diagonals (:474-518).  Row j holds rows
    //      i in [max(qLow+1, j+c0-k), min(qLen, j+c0+k)],  c0 = qStart + diag - tLen.
    const int c0 = qStart + diag - tLen;
    const bool qLong = qLen >= tLen;
    for (int s = (qLow + 1) + (tLow + 1); s <= qLen + tLen; s++) {
      int num = s - c0 - k;
      int jlo = (num >= 0) ? ((num + 1) >> 1) : -((-num) >> 1);           // ceil(num/2)
      jlo = max(jlo, max(tLow + 1, s - qLen));
      int num2 = s - c0 + k;
      int jhi = (num2 >= 0) ? (num2 >> 1) : -((-num2 + 1) >> 1);          // floor(num2/2)
      jhi = min(jhi, min(tLen, s - qLow - 1));
      for (int j = jlo + lane; j <= jhi; j += G) {
        int i = s - j;
        int delClose = MISS, insClose = MISS;
        if (qLong) delClose = (j <= diag) ? w.loMax[j] : MISS;
        else insClose = (i <= diag) ? w.upMax[i] : MISS;
        int a = SI(i - 1, j), b = SI(i, j - 1), c = SI(i - 1, j - 1);
        if (!inb(a) || !inb(b) || !inb(c)) { status |= LRA_ST_OOB_SLOT; continue; }
        int sIns = w.sSuf[a] + indel;
        int sDel = w.sSuf[b] + indel;
        int sMat = w.sSuf[c] + (w.qc[i] == w.tc[j] ? m : mm);
        int best = max(delClose, max(insClose, max(sIns, max(sDel, sMat))));
        int ar = (best == sIns) ? A_LEFT : (best == sDel) ? A_DOWN : (best == sMat) ? A_DIAG
                 : (best == delClose) ? A_GAPLEFT : A_GAPDOWN;              // :502-516
        SSET(SI(i, j), best, (signed char)ar);
      }
      wave_sync();
    }
    // ---- suffix trace back (:523-580), lane 0
    ti = qLen; tj = tLen;
    int s0 = SI(ti, tj);
    int arrow = inb(s0) ? w.pSuf[s0] : -1;
    result = inb(s0) ? w.sSuf[s0] : MISS;
    if (lane == 0) {
      long it = 0;
      while (arrow != A_DONE && arrow != A_GAPDOWN && arrow != A_GAPLEFT && ti >= 0 && tj >= 0) {
        if (++it > iter_cap) { status |= LRA_ST_NO_TERMINATION; break; }
        if (arrow == A_DIAG) { run++; ti--; tj--; }
        else if (arrow == A_LEFT) { flush(ti, tj); ti--; }
        else if (arrow == A_DOWN) { flush(ti, tj); tj--; }
        else { status |= LRA_ST_NO_TERMINATION; break; }   // border / unset arrow: endless in the reference
        if (ti >= 0 && tj >= 0) { int sl = SI(ti, tj); arrow = inb(sl) ? w.pSuf[sl] : -1; if (!inb(sl)) status |= LRA_ST_OOB_SLOT; }
      }
      flush(ti, tj);
      if (arrow == A_GAPDOWN && ti >= 0 && ti <= diag) tj = w.upIdx[ti];      // :568-574
      else if (arrow == A_GAPLEFT && tj >= 0 && tj <= diag) ti = w.loIdx[tj]; // :575-580
    }
  } else {                                                                    // :582-586
    ti = qB - 1; tj = tB - 1;
    result = rolling ? rollResult : w.sPre[PI(ti, tj)];
  }
  // ---- prefix trace back (:589-629), lane 0 -- or, with the arrows in HBM only (rolling), all lanes in step on the same state over a window of rows staged in
  // LDS: the walk is a chain of dependent one-byte loads a row apart (a cache line each); a window serves at least as many steps as it has rows
  if (rolling) {
    unsigned char* chunk = (unsigned char*)(roll + 768 + 1024);          // chunkBytes (LRA_AOG_CHUNK: 4 KB; what a wave asks for sets how many fit a CU)
    const int chRows = max(1, chunkBytes / R);
    int cLo = 1, cHi = 0;
    auto arrowAt = [&](int i, int j) -> int {
      if (j < cLo || j > cHi) {
        cHi = j; cLo = max(0, j - chRows + 1);
        wave_sync();
        const long base = (long)cLo * R, len = (long)(cHi - cLo + 1) * R;
        for (long x = lane; x < len; x += G) chunk[x] = (unsigned char)w.pPre[base + x];
        wave_sync();
      }
      return (int)(signed char)chunk[(j - cLo) * R + (i - j) + k + 1];
    };
    auto flushL = [&](int i_after, int j_after) {
      if (run > 0) {
        if (nb < cap) { if (lane == 0) { blocks[3 * nb] = i_after; blocks[3 * nb + 1] = j_after; blocks[3 * nb + 2] = run; } }
        else status |= LRA_ST_CAPACITY;
        nb++;
        run = 0;
      }
    };
    // lane l looks at the cell l steps down the current diagonal (inside the staged window): a run of diagonal arrows is one round
    constexpr int A_OFF = 100, A_WIN = 101;
    const unsigned long long gmask = (~0ULL >> (64 - G)) << gbase;
    if (ti >= 0 && tj >= 0 && inb(PI(ti, tj))) {
      while (true) {
        (void)arrowAt(ti, tj);                                            // the window holds row tj now
        const int jl = tj - lane;
        const bool on = ti - lane >= 0 && jl >= 0;
        const int a = !on ? A_OFF : jl < cLo ? A_WIN : (int)(signed char)chunk[(jl - cLo) * R + (ti - tj) + k + 1];
        const unsigned long long nd = (__ballot(a != A_DIAG) & gmask) >> gbase;
        const int f = nd ? __ffsll((long long)nd) - 1 : G;
        run += f; ti -= f; tj -= f;
        if (f == G) { if (ti < 0 || tj < 0) break; continue; }
        const int af = __shfl(a, gbase + f);
        if (af == A_WIN) continue;                                        // past the window: stage the next one
        if (af == A_OFF) break;
        if (af == A_LEFT) { flushL(ti, tj); ti--; }
        else if (af == A_DOWN) { flushL(ti, tj); tj--; }
        else { if (af != A_BORDER && af != A_DONE && af != A_GAPLEFT && af != A_GAPDOWN) status |= LRA_ST_NO_TERMINATION; break; }
        if (ti < 0 || tj < 0) break;
      }
    }
    flushL(ti, tj);
  } else if (lane == 0) {
    int arrow = (ti >= 0 && tj >= 0 && inb(PI(ti, tj))) ? w.pPre[PI(ti, tj)] : A_DONE;
    long it = 0;
    while (arrow != A_BORDER && arrow != A_DONE && ti >= 0 && tj >= 0) {
      if (++it > iter_cap) { status |= LRA_ST_NO_TERMINATION; break; }
      if (arrow == A_DIAG) { run++; ti--; tj--; }
      else if (arrow == A_LEFT) { flush(ti, tj); ti--; }
      else if (arrow == A_DOWN) { flush(ti, tj); tj--; }
      else { if (arrow != A_GAPLEFT && arrow != A_GAPDOWN) status |= LRA_ST_NO_TERMINATION; break; }
      if (ti < 0 || tj < 0) break;
      arrow = w.pPre[PI(ti, tj)];
    }
    flush(ti, tj);
  }
#undef PSET
#undef SSET
  // ---- publish: blocks were written in walk (reverse) order with absolute matrix
  //      coordinates; alignment order is the reverse, relative to where the walk ended.
  nb = __shfl(nb, gbase);
  int fi = __shfl(ti, gbase), fj = __shfl(tj, gbase);
  wave_sync();
  long nw = min(nb, cap);
  for (long x = lane; x < (nw + 1) / 2; x += G) {
    long y = nw - 1 - x;
    int a0 = blocks[3 * x], a1 = blocks[3 * x + 1], a2 = blocks[3 * x + 2];
    int b0 = blocks[3 * y], b1 = blocks[3 * y + 1], b2 = blocks[3 * y + 2];
    blocks[3 * x] = b0 - fi; blocks[3 * x + 1] = b1 - fj; blocks[3 * x + 2] = b2;
    if (y != x) { blocks[3 * y] = a0 - fi; blocks[3 * y + 1] = a1 - fj; blocks[3 * y + 2] = a2; }
  }
  // status bits raised by any lane
  for (int off = G / 2; off > 0; off >>= 1) status |= __shfl_xor(status, off);
  if (lane == 0) {
    // map the device score domain back to the reference's (int)(long) truncation
    int r = result;
    if (r < -(1 << 29)) r = (int)(unsigned int)((long)INT_MIN + ((long)r - (long)MISS));
    *out_score = r;
    *out_nb = (int)nb;
    *out_status = status;
  }
}

// ---- the same DP for a problem that only uses the prefix band (!g.top: the common case -- the two sequences differ in length by less than the band), with the SCORES
// in registers.  Nothing reads a prefix score again but the cell's three successors (and the corner's caller), and with lane L holding diagonals dd = 2L (even) and
// 2L + 1 (odd) of the band (dd = i - j + k + 1), a cell's predecessors are: the same diagonal two steps ago (own register), and the two neighbouring diagonals one
// step ago -- one in the lane's other register, one in the neighbouring lane's (a DPP wave shift).  An anti-diagonal step is then ~30 ALU instructions, one DPP move and
// one arrow store, with no fence: the LDS version's step is a chain of LDS round trips (three score loads, the stores, a wave fence) several times as long.
// LDS (or HBM) holds the 1-byte arrows and the sequence codes only: a fifth of the memory, so five times as many problems in flight per CU.
// Trace back: the G lanes look at the next G cells down the current diagonal at once; a run of diagonal arrows (the common step) is taken in one round.
// Needs (2k + 3 + 1) / 2 <= G, i.e. k + 2 <= G.  Bit-identical to solve() by construction: same cell order, same operands, same tie rules.
template <int G, typename BytePtr>
__device__ __forceinline__ void solve_reg(int wlane, const Problem& pr, const Geo& g, BytePtr mem, int* out_score, int* out_nb, int* blocks, long cap, int* out_status) {
  const int qLen = g.qLen, tLen = g.tLen, k = g.k, R = g.R, diag = g.diag, n = g.n, nUsed = g.nUsed;
  const int m = pr.m, mm = pr.mm, indel = pr.indel;
  const int lane = wlane & (G - 1), gbase = wlane - lane;
  auto* P = (signed char*)&mem[0];                                        // arrows [align4(nUsed)], then the codes
  auto* qc = (unsigned char*)&mem[align4(nUsed)];
  auto* tc = (unsigned char*)&mem[align4(nUsed) + align4(qLen + 1)];
  int status = 0;
  auto PI = [&](int i, int j) { return j * R + (i - j) + k + 1; };
  auto inb = [&](int s_) { return s_ >= 0 && s_ < n; };
#define PSETA(slot, ar) do { int s__ = (slot); if (inb(s__)) { if (s__ < nUsed) P[s__] = (ar); } else status |= LRA_ST_OOB_SLOT; } while (0)
  for (int x = lane; x <= qLen; x += G) qc[x] = x ? code_n((unsigned char)pr.q[x - 1]) : 0;
  for (int x = lane; x <= tLen; x += G) tc[x] = x ? code_n((unsigned char)pr.t[x - 1]) : 0;
  { auto* P4 = (int*)&mem[0]; for (int x = lane; x < (int)(align4(nUsed) >> 2); x += G) P4[x] = -1; }
  wave_sync();
  // prefix boundary, then rails (the rails overwrite (0, k + 1)): the arrows of solve()'s stores, in its order
  for (int i = 1 + lane; i < k + 1; i += G) PSETA(PI(i, 0), A_LEFT);
  for (int j = 1 + lane; j <= k + 1; j += G) PSETA(PI(0, j), A_DOWN);
  if (lane == 0) PSETA(PI(0, 0), A_DONE);
  wave_sync();
  if (qLen >= tLen) {
    for (int i = lane; i <= diag - k - 1; i += G) PSETA(PI(i, i + k + 1), A_BORDER);
    for (int i = 1 + lane; i < diag + k - 1; i += G) PSETA(PI(i + k + 1, i), A_BORDER);
  }
  if (qLen <= tLen) {
    for (int j = lane; j < diag - 1; j += G) PSETA(PI(j + k + 1, j), A_BORDER);
    for (int j = 1 + lane; j < diag + k; j += G) PSETA(PI(j - k - 1, j), A_BORDER);
  }
  wave_sync();
#undef PSETA
  const int qB = g.qB, tB = g.tB;
  const int result = reg_fill<G>(lane, gbase, g, m, mm, indel, P, qc, tc);
  wave_sync();
  int ti = qB - 1, tj = tB - 1;
  // ---- trace back (solve()'s prefix walk :589-629): every lane holds the walk's state; lane l looks at the cell l steps down the current diagonal
  long nb = 0; int run = 0;
  auto flushL = [&](int i_after, int j_after) {
    if (run > 0) {
      if (nb < cap) { if (lane == 0) { blocks[3 * nb] = i_after; blocks[3 * nb + 1] = j_after; blocks[3 * nb + 2] = run; } }
      else status |= LRA_ST_CAPACITY;
      nb++;
      run = 0;
    }
  };
  constexpr int A_OFF = 100;                                              // off the matrix: the walk stops after the move that leaves it
  const unsigned long long gmask = (~0ULL >> (64 - G)) << gbase;
  if (ti >= 0 && tj >= 0 && inb(PI(ti, tj))) {
    while (true) {
      const bool on = ti - lane >= 0 && tj - lane >= 0;
      const int a = on ? (int)P[PI(ti, tj) - lane * R] : A_OFF;
      const unsigned long long nd = (__ballot(a != A_DIAG) & gmask) >> gbase;
      const int f = nd ? __ffsll((long long)nd) - 1 : G;                  // diagonal arrows before the first other one
      run += f; ti -= f; tj -= f;
      if (f == G) { if (ti < 0 || tj < 0) break; continue; }
      const int af = __shfl(a, gbase + f);
      if (af == A_OFF) break;                                             // (ti < 0 || tj < 0 after a diagonal move)
      if (af == A_LEFT) { flushL(ti, tj); ti--; }
      else if (af == A_DOWN) { flushL(ti, tj); tj--; }
      else { if (af != A_BORDER && af != A_DONE && af != A_GAPLEFT && af != A_GAPDOWN) status |= LRA_ST_NO_TERMINATION; break; }
      if (ti < 0 || tj < 0) break;
    }
  }
  flushL(ti, tj);
  // ---- publish (as solve())
  wave_sync();
  const long nw = min(nb, cap);
  for (long x = lane; x < (nw + 1) / 2; x += G) {
    const long y = nw - 1 - x;
    const int a0 = blocks[3 * x], a1 = blocks[3 * x + 1], a2 = blocks[3 * x + 2];
    const int b0 = blocks[3 * y], b1 = blocks[3 * y + 1], b2 = blocks[3 * y + 2];
    blocks[3 * x] = b0 - ti; blocks[3 * x + 1] = b1 - tj; blocks[3 * x + 2] = b2;
    if (y != x) { blocks[3 * y] = a0 - ti; blocks[3 * y + 1] = a1 - tj; blocks[3 * y + 2] = a2; }
  }
  for (int off = G / 2; off > 0; off >>= 1) status |= __shfl_xor(status, off);
  if (lane == 0) {
    int r = result;
    if (r < -(1 << 29)) r = (int)(unsigned int)((long)INT_MIN + ((long)r - (long)MISS));
    *out_score = r;
    *out_nb = (int)nb;
    *out_status = status;
  }
}
__device__ __forceinline__ long need_bytes_reg(const Geo& g) { return align4(g.nUsed) + align4(g.qLen + 1) + align4(g.tLen + 1); }
constexpr int REG_S_BYTES = 2048, REG_M_BYTES = 8192, REG_L_BYTES = 32768, REG_X_BYTES = 65536;   // classes 10 (16 lanes), 11 (32 lanes), 12, 13 (64 lanes)

struct BatchArgs {
  int n;
  const char* qseq; const char* tseq;
  const uint64_t* q_off; const int32_t* q_len; const uint64_t* t_off; const int32_t* t_len;
  const int32_t* k;
  int m, mm, indel;
  int32_t* score; int32_t* nblocks; int32_t* blocks; const uint64_t* block_off; int32_t* status;
  // work lists: aog_classify gives every problem a class (cls8) and counts the classes; aog_scatter lays the problems out class by class in ONE list of n entries
  // (class c at offs[c], counts[c] entries).  NCLS classes; 14 .. 18 are the lane-per-problem class cut by size, so that the 64 problems of a wave cost about the same
  int* counts; int* offs; int* cursor; unsigned char* cls8; int* list;
  char* gscratch; long gslot_bytes; int gslots;       // class 2: HBM work slots
  char* gscratchB; long gslotB_bytes; int gslotsB;    // class 6: a few larger ones
  int chunk_bytes;                                    // classes 2 / 6: the trace-back window in LDS
  int use_reg;                                        // classes 10-13 (solve_reg) in use
  int use_lane;                                       // class 14 (aog_lane_kernel) in use
  int regL, regX;                                     // largest arrows + codes footprint of classes 12 and 13 (above: the HBM class, whose sweep is in registers too)
};

__device__ __forceinline__ bool load_problem(const BatchArgs& a, int p, Problem& pr, Geo& g, int& range_ok) {
  pr.q = a.qseq + a.q_off[p]; pr.t = a.tseq + a.t_off[p];
  pr.qLen = a.q_len[p]; pr.tLen = a.t_len[p]; pr.k0 = a.k[p];
  pr.m = a.m; pr.mm = a.mm; pr.indel = a.indel;
  bool ok = pr.qLen >= 0 && pr.tLen >= 0 && pr.k0 >= 1 && make_geo(pr.qLen, pr.tLen, pr.k0, g);
  long mx = max(abs(a.m), max(abs(a.mm), abs(a.indel)));
  range_ok = ok && ((long)(pr.qLen + pr.tLen + 16) * (mx + 1) < (1L << 28));
  return range_ok;
}

// Every workgroup takes CLS_ITEMS x 1024 consecutive problems and touches the global class counters once: one atomic per wave and class on a handful of addresses
// is 10^6 serialized L2 atomics for a batch of 2 x 10^7 problems.
constexpr int CLS_ITEMS = 16;
__device__ __forceinline__ int classify_one(const BatchArgs& a, int p) {
  Problem pr; Geo g; int ok;
  if (!load_problem(a, p, pr, g, ok)) { a.score[p] = 0; a.nblocks[p] = 0; a.status[p] = LRA_ST_RANGE; return -1; }
  long need = need_bytes(g);
  int cls = need <= CLASS_A_BYTES ? 0 : need <= CLASS_M1_BYTES ? 4 : need <= CLASS_M2_BYTES ? 5 : need <= CLASS_B_BYTES ? 1 : 2;
  if (g.k + 1 <= 32) cls = cls == 0 ? 7 : cls == 4 ? 8 : cls == 5 ? 9 : cls;                              // anti-diagonals of at most 32 cells: two problems per wave
  if (need <= CLASS_S_BYTES && g.k + 1 <= 16) cls = 3;
  if (cls == 2 && need > a.gslot_bytes) cls = 6;
  if (cls == 6 && need > a.gslotB_bytes) { a.score[p] = 0; a.nblocks[p] = 0; a.status[p] = LRA_ST_RANGE; return -1; }
  if (!g.top && a.use_reg) {                                               // prefix band only: scores in registers (solve_reg), arrows + codes in LDS
    const long nr = need_bytes_reg(g);
    if (a.use_lane && g.qLen <= LN_MAX && g.tLen <= LN_MAX) { const int mx = max(g.qLen, g.tLen); cls = mx <= 3 ? 14 : mx <= 6 ? 15 : mx <= 10 ? 16 : mx <= 16 ? 17 : 18; }
    else if (g.k + 2 <= 16 && nr <= REG_S_BYTES) cls = 10;
    else if (g.k + 2 <= 32 && nr <= REG_M_BYTES) cls = 11;
    else if (g.k + 2 <= 64 && nr <= a.regL) cls = 12;
    else if (g.k + 2 <= 64 && nr <= a.regX) cls = 13;
  }
  return cls;
}
__global__ void __launch_bounds__(1024) aog_classify(BatchArgs a) {
  __shared__ int s_cnt[NCLS];
  const int lane = threadIdx.x & 63;
  if (threadIdx.x < NCLS) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  for (int it = 0; it < CLS_ITEMS; it++) {
    const long p = ((long)blockIdx.x * CLS_ITEMS + it) * 1024 + threadIdx.x;
    int cls = -1;
    if (p < a.n) { cls = classify_one(a, (int)p); a.cls8[p] = (unsigned char)cls; }
    for (int c = 0; c < NCLS; c++) {
      const unsigned long long m = __ballot(cls == c);
      if (m && lane == __ffsll((long long)m) - 1) atomicAdd(&s_cnt[c], __popcll(m));
    }
  }
  __syncthreads();
  if (threadIdx.x < NCLS && s_cnt[threadIdx.x]) atomicAdd(&a.counts[threadIdx.x], s_cnt[threadIdx.x]);
}
__global__ void aog_offsets(BatchArgs a) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { int at = 0; for (int c = 0; c < NCLS; c++) { a.offs[c] = at; at += a.counts[c]; a.cursor[c] = 0; } }
}
__global__ void __launch_bounds__(1024) aog_scatter(BatchArgs a) {
  __shared__ int s_cnt[NCLS], s_base[NCLS];
  const int lane = threadIdx.x & 63;
  if (threadIdx.x < NCLS) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  int mine[CLS_ITEMS];
#pragma unroll
  for (int it = 0; it < CLS_ITEMS; it++) {                                // how many of each class this workgroup holds
    const long p = ((long)blockIdx.x * CLS_ITEMS + it) * 1024 + threadIdx.x;
    const int cls = p < a.n ? (int)(signed char)a.cls8[p] : -1;
    mine[it] = cls;
    for (int c = 0; c < NCLS; c++) {
      const unsigned long long m = __ballot(cls == c);
      if (m && lane == __ffsll((long long)m) - 1) atomicAdd(&s_cnt[c], __popcll(m));
    }
  }
  __syncthreads();
  if (threadIdx.x < NCLS) { s_base[threadIdx.x] = s_cnt[threadIdx.x] ? a.offs[threadIdx.x] + atomicAdd(&a.cursor[threadIdx.x], s_cnt[threadIdx.x]) : 0; s_cnt[threadIdx.x] = 0; }
  __syncthreads();
  const unsigned long long below = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));
#pragma unroll
  for (int it = 0; it < CLS_ITEMS; it++) {
    const long p = ((long)blockIdx.x * CLS_ITEMS + it) * 1024 + threadIdx.x;
    const int cls = mine[it];
    for (int c = 0; c < NCLS; c++) {
      const unsigned long long m = __ballot(cls == c);
      if (!m) continue;
      int base = 0;
      const int leader = __ffsll((long long)m) - 1;
      if (lane == leader) base = atomicAdd(&s_cnt[c], __popcll(m));
      base = __shfl(base, leader);
      if (cls == c) a.list[s_base[c] + base + __popcll(m & below)] = (int)p;
    }
  }
}

template <int CLS>
__global__ void __launch_bounds__((CLS == 0 || CLS == 3 || CLS == 7) ? 256 : 64) aog_kernel(BatchArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_roll = (int*)smem;                                               // classes 2 / 6 (one wave per workgroup): 3 KB of rolling scores, 4 KB of codes, the trace-back window
  constexpr int G = (CLS == 3) ? 16 : (CLS >= 7) ? 32 : 64;
  constexpr int GPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const int wave_in_wg = threadIdx.x >> 6;
  const int waves_per_wg = blockDim.x >> 6;
  const int group = (blockIdx.x * waves_per_wg + wave_in_wg) * GPW + lane / G;
  const int ngroups = gridDim.x * waves_per_wg * GPW;
  const int count = a.counts[CLS];
  for (int x = group; x < count; x += ngroups) {
    int p = a.list[a.offs[CLS] + x];
    Problem pr; Geo g; int ok;
    load_problem(a, p, pr, g, ok);
    long cap = (long)(a.block_off[p + 1] - a.block_off[p]);
    int* blk = a.blocks + 3 * a.block_off[p];
    if (CLS == 3) solve<16>(lane, pr, g, smem + (wave_in_wg * GPW + lane / G) * CLASS_S_BYTES, &a.score[p], &a.nblocks[p], blk, cap, &a.status[p]);
    else if (CLS == 8) solve<32>(lane, pr, g, smem + (lane / G) * CLASS_M1_BYTES, &a.score[p], &a.nblocks[p], blk, cap, &a.status[p]);
    else if (CLS == 9) solve<32>(lane, pr, g, smem + (lane / G) * CLASS_M2_BYTES, &a.score[p], &a.nblocks[p], blk, cap, &a.status[p]);
    else if (CLS == 7) solve<32>(lane, pr, g, smem + (wave_in_wg * GPW + lane / G) * CLASS_A_BYTES, &a.score[p], &a.nblocks[p], blk, cap, &a.status[p]);
    else if (CLS == 0) solve<64>(lane, pr, g, smem + wave_in_wg * CLASS_A_BYTES, &a.score[p], &a.nblocks[p], blk, cap, &a.status[p]);
    else if (CLS == 1 || CLS == 4 || CLS == 5) solve<64>(lane, pr, g, smem, &a.score[p], &a.nblocks[p], blk, cap, &a.status[p]);
    else if (CLS == 6) solve<64>(lane, pr, g, a.gscratchB + (long)(group % a.gslotsB) * a.gslotB_bytes, &a.score[p], &a.nblocks[p], blk, cap, &a.status[p], s_roll, a.chunk_bytes);
    else solve<64>(lane, pr, g, a.gscratch + (long)(group % a.gslots) * a.gslot_bytes, &a.score[p], &a.nblocks[p], blk, cap, &a.status[p], s_roll, a.chunk_bytes);
    wave_sync();
  }
}

// classes 10-13: solve_reg.  10: 16 lanes per problem, 16 problems per 256-thread workgroup; 11: 32 lanes, 4 problems per 128 threads; 12, 13: a wave per problem
template <int CLS>
__global__ void __launch_bounds__(CLS == 10 ? 256 : CLS == 11 ? 128 : 64) aog_reg_kernel(BatchArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int G = CLS == 10 ? 16 : CLS == 11 ? 32 : 64;
  constexpr int BYTES = CLS == 10 ? REG_S_BYTES : CLS == 11 ? REG_M_BYTES : CLS == 12 ? REG_L_BYTES : REG_X_BYTES;
  constexpr int GPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const int wave_in_wg = threadIdx.x >> 6;
  const int waves_per_wg = blockDim.x >> 6;
  const int group = (blockIdx.x * waves_per_wg + wave_in_wg) * GPW + lane / G;
  const int ngroups = gridDim.x * waves_per_wg * GPW;
  const int count = a.counts[CLS];
  for (int x = group; x < count; x += ngroups) {
    const int p = a.list[a.offs[CLS] + x];
    Problem pr; Geo g; int ok;
    load_problem(a, p, pr, g, ok);
    const long cap = (long)(a.block_off[p + 1] - a.block_off[p]);
    int* blk = a.blocks + 3 * a.block_off[p];
    solve_reg<G>(lane, pr, g, smem + (wave_in_wg * GPW + lane / G) * BYTES, &a.score[p], &a.nblocks[p], blk, cap, &a.status[p]);
    wave_sync();
  }
}


// ---- class 14: the tiny problems (both sequences at most LN_MAX long, prefix band only) -- most of a batch's gaps between neighbouring anchors are a handful of
// bases -- one LANE per problem.  A wave's anti-diagonal sweep spends more on a 7 x 7 problem's set-up than on its 49 cells; here 64 problems go through the wave at
// once, each lane filling its own band row by row (the values do not depend on the order the cells are visited in): the previous row's scores in LDS (one word per
// column, lane-interleaved, updated in place), the arrows two bits apiece (LEFT / DOWN / DIAG; two words per row), the sequence codes one byte each.  The cells the
// sweep never computes -- row / column 0, the rails, everything off the band -- are what solve()'s boundary stores leave there: pre() for the scores, pre_arrow()
// for the arrows.  The walk runs twice over the arrows in LDS (count the blocks and find where it ends; then store them in alignment order).

// MX: the class's largest sequence length (3, 6, 10, 16, 24).  The kernel is a chain of LDS round trips per lane, so its time is that chain over the waves a CU holds:
// the tables are sized for the class (a row of at most 16 arrows is one word), 2.5 / 4.5 / 7 / 11 / 22 KB per wave instead of 22 for all, and a class's problems
// load MX bases per sequence, not 24.
template <int MX>
__global__ void __launch_bounds__(64) aog_lane_kernel(BatchArgs a, int cls) {
  constexpr int LN_MAX = MX, LN_ROWW = MX > 16 ? 2 : 1;
  constexpr int LN_ARROW_WORDS = LN_MAX * LN_ROWW, LN_PREV_WORDS = LN_MAX + 2, LN_CODE_BYTES = LN_MAX + 1;
  constexpr int LN_BYTES = 64 * (4 * LN_ARROW_WORDS + 4 * LN_PREV_WORDS + 2 * LN_CODE_BYTES);
  __shared__ __attribute__((aligned(16))) char smem[LN_BYTES];
  const int lane = threadIdx.x;
  unsigned* AR = (unsigned*)smem;                                         // [LN_ARROW_WORDS][64]
  int* PV = (int*)(smem + 64 * 4 * LN_ARROW_WORDS);                       // [LN_PREV_WORDS][64]
  unsigned char* QC = (unsigned char*)(smem + 64 * 4 * (LN_ARROW_WORDS + LN_PREV_WORDS));   // [LN_CODE_BYTES][64]
  unsigned char* TC = QC + 64 * LN_CODE_BYTES;
  const int count = a.counts[cls];
  for (int x0 = blockIdx.x * 64; x0 < count; x0 += gridDim.x * 64) {
    const int x = x0 + lane;
    if (x < count) {
      const int p = a.list[a.offs[cls] + x];
      Problem pr; Geo g; int ok;
      load_problem(a, p, pr, g, ok);
      const int qLen = g.qLen, tLen = g.tLen, k = g.k, diag = g.diag, qB = g.qB, tB = g.tB;
      const int m = pr.m, mm = pr.mm, indel = pr.indel;
      {                                                                   // (all loads in flight before the first is used)
        unsigned char cq[LN_MAX], ct[LN_MAX];
#pragma unroll
        for (int i = 0; i < LN_MAX; i++) { cq[i] = i < qLen ? (unsigned char)pr.q[i] : 0; ct[i] = i < tLen ? (unsigned char)pr.t[i] : 0; }
#pragma unroll
        for (int i = 0; i < LN_MAX; i++) { QC[(i + 1) * 64 + lane] = (unsigned char)code_n(cq[i]); TC[(i + 1) * 64 + lane] = (unsigned char)code_n(ct[i]); }
      }
      auto in_region = [&](int i, int j) { return i >= 1 && i <= qB - 1 && j >= 1 && j <= tB - 1 && i - j <= k && j - i <= k; };
      auto pre = [&](int i, int j) -> int {
        int v = MISS;
        if (j == 0 && i >= 1 && i < k + 1) v = indel * i;
        if (i == 0 && j >= 1 && j <= k + 1) v = indel * j;
        if (i == 0 && j == 0) v = 0;
        if (qLen >= tLen) { if (j == i + k + 1 && i <= diag - k - 1) v = MISS; if (i == j + k + 1 && j >= 1 && j < diag + k - 1) v = MISS; }
        if (qLen <= tLen) { if (i == j + k + 1 && j < diag - 1) v = MISS; if (j == i + k + 1 && j >= 1 && j < diag + k) v = MISS; }
        return v;
      };
      auto pre_arrow = [&](int i, int j) -> int {                           // the arrows of the same stores (-1: never stored)
        int v = -1;
        if (i < 0 || j < 0) return v;
        if (j == 0 && i >= 1 && i < k + 1) v = A_LEFT;
        if (i == 0 && j >= 1 && j <= k + 1) v = A_DOWN;
        if (i == 0 && j == 0) v = A_DONE;
        if (qLen >= tLen) { if (j == i + k + 1 && i <= diag - k - 1) v = A_BORDER; if (i == j + k + 1 && j >= 1 && j < diag + k - 1) v = A_BORDER; }
        if (qLen <= tLen) { if (i == j + k + 1 && j < diag - 1) v = A_BORDER; if (j == i + k + 1 && j >= 1 && j < diag + k) v = A_BORDER; }
        return v;
      };
      for (int j = 1; j <= tB - 1; j++) {
        const int ilo = max(1, j - k), ihi = min(qB - 1, j + k);
        const int tcj = TC[j * 64 + lane];
        int left = pre(ilo - 1, j);                                        // column 0 or the lower rail
        int dg = in_region(ilo - 1, j - 1) ? PV[(ilo - 1) * 64 + lane] : pre(ilo - 1, j - 1);
        unsigned long long acc = 0;
        const int hi2 = j >= 2 ? min(qB - 1, j - 1 + k) : 0;              // row j - 1 holds computed cells up to column hi2 (and from at most ilo on)
        for (int i = ilo; i <= ihi; i++) {
          // (i, j - 1) off the computed region: row 0 (the boundary value), or the cell past the band's edge (column j + k: a rail or beyond, MISS)
          int up;
          if (i <= hi2) up = PV[i * 64 + lane];
          else up = (j == 1 && i < k + 1) ? indel * i : MISS;
          const int sIns = left + indel, sDel = up + indel, sMat = dg + (QC[i * 64 + lane] == tcj ? m : mm);
          const int best = max(sIns, max(sDel, sMat));
          const int ar = (best == sIns) ? A_LEFT : (best == sDel) ? A_DOWN : A_DIAG;
          acc |= (unsigned long long)ar << (2 * (i - ilo));
          PV[i * 64 + lane] = best;
          dg = up; left = best;
        }
        AR[((j - 1) * LN_ROWW) * 64 + lane] = (unsigned)acc; if (LN_ROWW > 1) AR[((j - 1) * LN_ROWW + 1) * 64 + lane] = (unsigned)(acc >> 32);
      }
      const int ci = qB - 1, cj = tB - 1;
      const int result = in_region(ci, cj) ? PV[ci * 64 + lane] : ((ci >= 0 && cj >= 0) ? pre(ci, cj) : MISS);
      auto arrow_at = [&](int i, int j) -> int {
        if (!in_region(i, j)) return pre_arrow(i, j);
        const int sh = 2 * (i - max(1, j - k));
        const unsigned w = AR[((j - 1) * LN_ROWW + (LN_ROWW > 1 ? (sh >> 5) : 0)) * 64 + lane];
        return (int)((w >> (sh & 31)) & 3u);
      };
      const long cap = (long)(a.block_off[p + 1] - a.block_off[p]);
      int* blocks = a.blocks + 3 * a.block_off[p];
      int status = 0;
      long nb = 0; int fi = ci, fj = cj;
      for (int pass = 0; pass < 2; pass++) {                              // solve()'s prefix walk :589-629; pass 1 stores
        const long nw = min(nb, cap);
        long w = 0; int run = 0; int ti = ci, tj = cj;
        auto flush = [&](int i_after, int j_after) {
          if (run > 0) {
            if (pass == 1 && w < nw) { int* b = blocks + 3 * (nw - 1 - w); b[0] = i_after - fi; b[1] = j_after - fj; b[2] = run; }
            w++; run = 0;
          }
        };
        int arrow = (ti >= 0 && tj >= 0) ? arrow_at(ti, tj) : A_DONE;
        while (arrow != A_BORDER && arrow != A_DONE && ti >= 0 && tj >= 0) {
          if (arrow == A_DIAG) { run++; ti--; tj--; }
          else if (arrow == A_LEFT) { flush(ti, tj); ti--; }
          else if (arrow == A_DOWN) { flush(ti, tj); tj--; }
          else { if (arrow != A_GAPLEFT && arrow != A_GAPDOWN) status |= LRA_ST_NO_TERMINATION; break; }
          if (ti < 0 || tj < 0) break;
          arrow = arrow_at(ti, tj);
        }
        flush(ti, tj);
        if (pass == 0) { nb = w; fi = ti; fj = tj; if (nb > cap) status |= LRA_ST_CAPACITY; }
      }
      int r = result;
      if (r < -(1 << 29)) r = (int)(unsigned int)((long)INT_MIN + ((long)r - (long)MISS));
      a.score[p] = r; a.nblocks[p] = (int)nb; a.status[p] = status;
    }
  }
}

}  // namespace

int lra_aog_launch_device(lra_ctx* ctx, int n, const char* d_qseq, const char* d_tseq, const uint64_t* d_q_off,
                          const int32_t* d_q_len, const uint64_t* d_t_off, const int32_t* d_t_len, const int32_t* d_k,
                          int m, int mm, int indel, int32_t* d_score, int32_t* d_nblocks, int32_t* d_blocks,
                          const uint64_t* d_block_off, int32_t* d_status) {
  if (!ctx) return LRA_ERR_INVALID;
  if (n < 0) return lra_set_err(ctx, LRA_ERR_INVALID, "n < 0");
  if (n == 0) return LRA_OK;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  BatchArgs a;
  a.n = n; a.qseq = d_qseq; a.tseq = d_tseq; a.q_off = d_q_off; a.q_len = d_q_len; a.t_off = d_t_off; a.t_len = d_t_len;
  a.k = d_k; a.m = m; a.mm = mm; a.indel = indel;
  a.score = d_score; a.nblocks = d_nblocks; a.blocks = d_blocks; a.block_off = d_block_off; a.status = d_status;
  // scratch slot 0: counts[32] + offs[32] + cursor[32] + list[n] + cls8[n];  slot 1: class-C HBM work slots
  char* s0 = (char*)lra_scratch(ctx, 0, 384 + sizeof(int) * (size_t)n + (size_t)n + 64);
  if (!s0) return LRA_ERR_NOMEM;
  a.counts = (int*)s0; a.offs = a.counts + 32; a.cursor = a.counts + 64; a.list = (int*)(s0 + 384); a.cls8 = (unsigned char*)(a.list + n);
  // class 2: 4 MiB slots, LRA_AOG_SLOTS (default 8) per CU -- with the scores of most of these problems in LDS (rolling, see solve) a wave's HBM traffic is its
  // arrows, and a CU can keep more of them in flight; class 6: 8 MiB slots, one per CU, for the rare larger problem (1.5 kb x 1.5 kb at k = 60, 5 kb x 5 kb at k = 15)
  const int perCu = std::max(1, getenv("LRA_AOG_SLOTS") ? atoi(getenv("LRA_AOG_SLOTS")) : ctx->pipelined ? 6 : 8);   // (a zero or non-numeric value would leave the HBM class without a slot; two-stage batches: 6 -- step 955 -> 939 ms)
  a.gslots = ctx->num_cu * perCu; a.gslot_bytes = 4L << 20;
  a.chunk_bytes = std::min(8192, std::max(1024, getenv("LRA_AOG_CHUNK") ? atoi(getenv("LRA_AOG_CHUNK")) : 8192)) & ~255;
  a.gslotsB = ctx->num_cu; a.gslotB_bytes = 8L << 20;
  a.gscratch = (char*)lra_scratch(ctx, 1, (size_t)a.gslots * a.gslot_bytes + (size_t)a.gslotsB * a.gslotB_bytes);
  if (!a.gscratch) return LRA_ERR_NOMEM;
  a.gscratchB = a.gscratch + (size_t)a.gslots * a.gslot_bytes;
  if (getenv("LRA_AOG_DBG")) {                                            // shapes of the batch's problems (host side, diagnostic)
    std::vector<int32_t> hq(n), ht(n), hk(n);
    (void)hipMemcpy(hq.data(), d_q_len, (size_t)n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(ht.data(), d_t_len, (size_t)n * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hk.data(), d_k, (size_t)n * 4, hipMemcpyDeviceToHost);
    long nTop = 0, nBigK = 0, steps = 0; long byNeed[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long byK[4] = {0, 0, 0, 0}; long cellsBy[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int p = 0; p < n; p++) {
      const int q = hq[p], t = ht[p]; const int diag = std::max(1, std::min(q, t)); int k = std::min(diag, hk[p]); bool top = true;
      if (diag + 2 * k >= std::max(q, t)) { k *= 2; top = false; }
      const long R = 2 * k + 3, tB = std::min(diag + k, t + 1); const long nU = top ? (3 + k + diag) * R : std::min((3 + k + diag) * R, tB * R);
      if (top) { nTop++; continue; }
      if (k > 62) { nBigK++; continue; }
      const long need = nU + q + t + 8;
      const int b = need <= 2048 ? 0 : need <= 4096 ? 1 : need <= 8192 ? 2 : need <= 16384 ? 3 : need <= 32768 ? 4 : need <= 65536 ? 5 : need <= 160 * 1024 ? 6 : 7;
      byNeed[b]++; cellsBy[b] += nU; steps += q + t;
      byK[k <= 14 ? 0 : k <= 30 ? 1 : k <= 62 ? 2 : 3]++;
    }
    fprintf(stderr, "[aog] n %d top %ld k>62 %ld | arrows+codes <=2K %ld <=4K %ld <=8K %ld <=16K %ld <=32K %ld <=64K %ld <=160K %ld more %ld | cells(M) %.1f %.1f %.1f %.1f %.1f %.1f %.1f %.1f | k<=14 %ld <=30 %ld <=62 %ld | steps(M) %.1f\n",
            n, nTop, nBigK, byNeed[0], byNeed[1], byNeed[2], byNeed[3], byNeed[4], byNeed[5], byNeed[6], byNeed[7], cellsBy[0] / 1e6, cellsBy[1] / 1e6, cellsBy[2] / 1e6, cellsBy[3] / 1e6,
            cellsBy[4] / 1e6, cellsBy[5] / 1e6, cellsBy[6] / 1e6, cellsBy[7] / 1e6, byK[0], byK[1], byK[2], steps / 1e6);
  }
  a.use_reg = getenv("LRA_AOG_NOREG") ? 0 : 1;
  a.use_lane = (a.use_reg && !getenv("LRA_AOG_NOLANE")) ? 1 : 0;
  a.regL = getenv("LRA_AOG_REG_L") ? std::min(atoi(getenv("LRA_AOG_REG_L")), REG_L_BYTES) : 16384;
  a.regX = getenv("LRA_AOG_REG_X") ? std::min(atoi(getenv("LRA_AOG_REG_X")), REG_X_BYTES) : 0;
  if (a.use_reg) LRA_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)aog_reg_kernel<13>, hipFuncAttributeMaxDynamicSharedMemorySize, REG_X_BYTES));
  LRA_HIP_CHECK(ctx, hipMemsetAsync(a.counts, 0, 384, ctx->stream));
  const int cblocks = (int)(((long)n + CLS_ITEMS * 1024 - 1) / (CLS_ITEMS * 1024));
  hipLaunchKernelGGL(aog_classify, dim3(cblocks), dim3(1024), 0, ctx->stream, a);
  hipLaunchKernelGGL(aog_offsets, dim3(1), dim3(64), 0, ctx->stream, a);
  hipLaunchKernelGGL(aog_scatter, dim3(cblocks), dim3(1024), 0, ctx->stream, a);
  int wgA = min((n + 3) / 4, ctx->num_cu * 5);
  int wgB = min(n, ctx->num_cu * 2);
  int wgC = min(n, a.gslots);
  LRA_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)aog_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * CLASS_A_BYTES));
  LRA_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)aog_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, CLASS_B_BYTES));
  LRA_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)aog_kernel<9>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CLASS_M2_BYTES));
  // The classes are independent, and every class's launch ends with a few long problems on a few CUs: they run side by side on the context's side streams
  // (the HBM class, the longest, on the context's own stream) instead of one after the other.  LRA_AOG_SERIAL=1: one stream (for comparisons).
  static const bool serial = getenv("LRA_AOG_SERIAL") != nullptr;
  const hipStream_t sm = ctx->stream;
  const hipStream_t s1 = serial ? sm : lra_side_fork(ctx, 1), s2 = serial ? sm : lra_side_fork(ctx, 2), s3 = serial ? sm : lra_side_fork(ctx, 3);
  lra_time_begin(ctx, "aog_lds_large", s1);
  hipLaunchKernelGGL(aog_kernel<1>, dim3(wgB), dim3(64), CLASS_B_BYTES, s1, a);
  lra_time_end(ctx, s1);
  lra_time_begin(ctx, "aog_lds_small", s2);
  hipLaunchKernelGGL(aog_kernel<7>, dim3(min((n + 7) / 8, ctx->num_cu * 2)), dim3(256), 8 * CLASS_A_BYTES, s2, a);
  hipLaunchKernelGGL(aog_kernel<0>, dim3(wgA), dim3(256), 4 * CLASS_A_BYTES, s2, a);
  lra_time_end(ctx, s2);
  lra_time_begin(ctx, "aog_lds_tiny", s2);
  hipLaunchKernelGGL(aog_kernel<3>, dim3(min((n + 15) / 16, ctx->num_cu * 8)), dim3(256), 16 * CLASS_S_BYTES, s2, a);
  lra_time_end(ctx, s2);
  lra_time_begin(ctx, "aog_lds_medium", s3);
  hipLaunchKernelGGL(aog_kernel<8>, dim3(min((n + 1) / 2, ctx->num_cu * 4)), dim3(64), 2 * CLASS_M1_BYTES, s3, a);
  hipLaunchKernelGGL(aog_kernel<9>, dim3(min((n + 1) / 2, ctx->num_cu * 2)), dim3(64), 2 * CLASS_M2_BYTES, s3, a);
  hipLaunchKernelGGL(aog_kernel<4>, dim3(min(n, ctx->num_cu * 8)), dim3(64), CLASS_M1_BYTES, s3, a);
  hipLaunchKernelGGL(aog_kernel<5>, dim3(min(n, ctx->num_cu * 5)), dim3(64), CLASS_M2_BYTES, s3, a);
  lra_time_end(ctx, s3);
  if (a.use_lane) {
    lra_time_begin(ctx, "aog_lane");
    const int nb64 = (n + 63) / 64;
    hipLaunchKernelGGL(aog_lane_kernel<24>, dim3(min(nb64, ctx->num_cu * 7)), dim3(64), 0, sm, a, 18);
    hipLaunchKernelGGL(aog_lane_kernel<16>, dim3(min(nb64, ctx->num_cu * 14)), dim3(64), 0, sm, a, 17);
    hipLaunchKernelGGL(aog_lane_kernel<10>, dim3(min(nb64, ctx->num_cu * 22)), dim3(64), 0, sm, a, 16);
    hipLaunchKernelGGL(aog_lane_kernel<6>, dim3(min(nb64, ctx->num_cu * 32)), dim3(64), 0, sm, a, 15);
    hipLaunchKernelGGL(aog_lane_kernel<3>, dim3(min(nb64, ctx->num_cu * 32)), dim3(64), 0, sm, a, 14);
    lra_time_end(ctx);
  }
  if (a.use_reg) {
    lra_time_begin(ctx, "aog_reg");
    // the three register classes side by side as well (each ends with a few long problems; one after the other they took 43 ms of the a13 call, LRA_AOG_SERIAL=1)
    hipLaunchKernelGGL(aog_reg_kernel<10>, dim3(min((n + 15) / 16, ctx->num_cu * 10)), dim3(256), 16 * REG_S_BYTES, sm, a);
    lra_time_end(ctx);
    lra_time_begin(ctx, "aog_reg_medium", s2);
    hipLaunchKernelGGL(aog_reg_kernel<11>, dim3(min((n + 3) / 4, ctx->num_cu * 10)), dim3(128), 4 * REG_M_BYTES, s2, a);
    lra_time_end(ctx, s2);
    lra_time_begin(ctx, "aog_reg_large", s3);
    hipLaunchKernelGGL(aog_reg_kernel<12>, dim3(min(n, ctx->num_cu * 10)), dim3(64), a.regL, s3, a);
    lra_time_end(ctx, s3);
    lra_time_begin(ctx, "aog_reg");
    if (a.regX) hipLaunchKernelGGL(aog_reg_kernel<13>, dim3(min(n, ctx->num_cu * 4)), dim3(64), a.regX, sm, a);
    lra_time_end(ctx);
  }
  lra_time_begin(ctx, "aog_hbm");
  hipLaunchKernelGGL(aog_kernel<2>, dim3(wgC), dim3(64), (3 * 256 + 1024) * 4 + a.chunk_bytes, sm, a);
  hipLaunchKernelGGL(aog_kernel<6>, dim3(min(n, a.gslotsB)), dim3(64), (3 * 256 + 1024) * 4 + a.chunk_bytes, sm, a);
  lra_time_end(ctx);
  if (!serial) { lra_side_join(ctx, 1); lra_side_join(ctx, 2); lra_side_join(ctx, 3); }
  LRA_HIP_CHECK(ctx, hipGetLastError());
  return LRA_OK;
}

extern "C" int lra_affine_one_gap_align_batch(lra_ctx* ctx, int n, const char* d_qseq, const char* d_tseq,
                                              const uint64_t* d_q_off, const int32_t* d_q_len,
                                              const uint64_t* d_t_off, const int32_t* d_t_len,
                                              const int32_t* d_k, int m, int mm, int indel,
                                              int32_t* d_score, int32_t* d_nblocks, int32_t* d_blocks,
                                              const uint64_t* d_block_off, int32_t* d_status) {
  return lra_aog_launch_device(ctx, n, d_qseq, d_tseq, d_q_off, d_q_len, d_t_off, d_t_len, d_k, m, mm, indel, d_score, d_nblocks,
                               d_blocks, d_block_off, d_status);
}
