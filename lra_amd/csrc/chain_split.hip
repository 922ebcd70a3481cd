// lra_amd/csrc/chain_split.hip -- SURVEY §8a row a9 (low-accuracy path): what MapRead_lowacc does to every chain of the first
// sparse DP before tier-2 refinement (Map_lowacc.h:189-192, :252-256), for all chains of a batch.  gfx950 only.
//   RemoveSpuriousJump<UltimateChain>     Chain.h:897-957
//   SPLITChain                            Mapping_ultility.h:385-441 (push_new :349-383, SplitChain::CHROMIndex Chain.h:386-394,
//                                         GenomeHeader::Find Genome.h:20-32, UltimateChain::diag Chain.h:243-246)
//   MergeSplitchainINS                    Mapping_ultility.h:172-262
//   RemoveSpuriousSplitChain              Map_lowacc.h:38-66
// Mapping: one lane per chain slot; these are short serial scans (a chain has a few hundred anchors, a read one or two chains), far
// below every other stage -- the point of having them on the device is that the chain never leaves HBM between the sparse DP and the
// refinement stages.  A split chain is a run [a, b) of the filtered chain; MergeSplitchainINS concatenates runs, kept as a linked list
// and laid out at the end.  Algorithmic bytes: 21 B per chain anchor in, ~14 B out.
#include "common.h"
#include "scan.h"
#include <algorithm>

namespace {

constexpr uint32_t NONE = 0xFFFFFFFFu;

struct SplitArgs {
  uint64_t n_slots;
  const uint64_t* chainStart; const uint32_t* chainLen; const uint32_t* nChains; int numAln;
  const uint32_t* cq; const uint32_t* ct; const int32_t* clen; const uint8_t* cstrand; const uint32_t* ccl; const uint8_t* clink;
  const uint64_t* pos; int npos; int splitdist, bypass;
  // outputs
  uint8_t* keep; uint32_t* nKept; uint8_t* link;
  uint32_t* nSplit; uint32_t* spBeg; uint32_t* spLen; uint32_t* spIdx; uint8_t* spLink; uint8_t* spType; uint8_t* spStrand; int32_t* spChrom; uint32_t* spBox;
  uint32_t* ciBeg; uint32_t* ciLen; uint32_t* ciIdx; uint8_t* splitLink; uint32_t* nSplitLink; uint32_t* status;
  // scratch, indexed like the chain arrays
  uint32_t* fidx;      // filtered position -> original position
  uint32_t* runA; uint32_t* runB; uint32_t* runNext; uint32_t* curInd; uint8_t* keepS; uint8_t* spl0;
};

__device__ int header_find(const uint64_t* pos, int npos, uint64_t query, bool& ub) {   // Genome.h:20-32
  if (npos > 0 && query == pos[0]) return 0;
  int lo = 0, cnt = npos;
  while (cnt > 0) { const int step = cnt >> 1; if (pos[lo + step] < query) { lo += step + 1; cnt -= step + 1; } else cnt = step; }
  if (lo == npos) { ub = true; return lo - 1; }
  if (query == pos[lo]) return lo;
  return lo - 1;
}

__global__ void __launch_bounds__(64) split_kernel(SplitArgs a) {
  const uint64_t s = (uint64_t)blockIdx.x * 64 + threadIdx.x;
  if (s >= a.n_slots) return;
  a.nKept[s] = 0; a.nSplit[s] = 0; a.nSplitLink[s] = 0; a.status[s] = 0;
  const uint32_t r = (uint32_t)(s / a.numAln), c = (uint32_t)(s % a.numAln);
  if (c >= a.nChains[r]) return;
  const uint64_t base = a.chainStart[s];
  const int n = (int)a.chainLen[s];
  if (n == 0) return;
  const uint32_t* Q = a.cq + base; const uint32_t* T = a.ct + base; const int32_t* Ln = a.clen + base;
  const uint8_t* St = a.cstrand + base; const uint32_t* Cl = a.ccl + base; const uint8_t* Lk = a.clink + base;
  uint8_t* keep = a.keep + base; uint8_t* link = a.link + base; uint32_t* fidx = a.fidx + base;
  // ---- RemoveSpuriousJump :897-957.  Only adjacent SVs one anchor apart matter, so one pass with the previous SV suffices.
  for (int i = 0; i < n; i++) keep[i] = 1;
  if (n >= 2) {
    int pSV = 0, pPos = -1; bool have = false;
    for (int i = 1; i < n; i++) {
      int sv = 0; bool is = false;
      if (St[i] == St[i - 1]) {
        int Gap;
        if (St[i] == 0) Gap = (int)(((long long)T[i] - (long long)Q[i]) - ((long long)T[i - 1] - (long long)Q[i - 1]));
        else Gap = (int)((long long)(uint32_t)(Q[i] + (uint32_t)Ln[i] + T[i]) - (long long)(uint32_t)(Q[i - 1] + (uint32_t)Ln[i - 1] + T[i - 1]));
        if (abs(Gap) > 100) { sv = Gap; is = true; }
      } else { sv = 0; is = true; }
      if (is) {
        if (have && ((sv >= 0) != (pSV >= 0)) && sv != 0 && pSV != 0 && i - pPos == 1 && Ln[pPos] < 50) keep[pPos] = 0;
        pSV = sv; pPos = i; have = true;
      }
    }
  }
  int N = 0;
  for (int i = 0; i < n; i++)
    if (keep[i]) { fidx[N] = (uint32_t)i; if (N >= 1) link[N - 1] = Lk[i - 1]; N++; }
  a.nKept[s] = (uint32_t)N;
#define FQ(i) Q[fidx[i]]
#define FT(i) T[fidx[i]]
#define FL(i) Ln[fidx[i]]
#define FS(i) St[fidx[i]]
#define FQE(i) (Q[fidx[i]] + (uint32_t)Ln[fidx[i]])
#define FTE(i) (T[fidx[i]] + (uint32_t)Ln[fidx[i]])
  // ---- SPLITChain :385-441; split k = run [runA, runB) of the filtered chain
  uint32_t* runA = a.runA + base; uint32_t* runB = a.runB + base; uint32_t* runNext = a.runNext + base;
  uint8_t* spType = a.spType + base; uint8_t* spStrand = a.spStrand + base; int32_t* spChrom = a.spChrom + base; uint32_t* spBox = a.spBox + 4 * base;
  uint8_t* spl = a.spl0 + base;
  bool ub = false;
  int ns = 0, nl = 0;
  int runStart = 0;
  auto push_new = [&](int from, int to) -> bool {                        // :349-383; the run is [from, to)
    const uint32_t QStart = FQ(to - 1), QEnd = FQE(from);
    uint32_t TStart, TEnd;
    if (FS(from) == 0) { TStart = FT(to - 1); TEnd = FTE(from); } else { TStart = FT(from); TEnd = FTE(to - 1); }
    const int first = header_find(a.pos, a.npos, (uint64_t)TStart + 1, ub), last = header_find(a.pos, a.npos, (uint64_t)TEnd, ub);
    if (first != last) return false;
    runA[ns] = (uint32_t)from; runB[ns] = (uint32_t)to; runNext[ns] = NONE; spType[ns] = 'N'; spStrand[ns] = FS(from); spChrom[ns] = first;
    spBox[4 * ns] = QStart; spBox[4 * ns + 1] = QEnd; spBox[4 * ns + 2] = TStart; spBox[4 * ns + 3] = TEnd;
    ns++;
    return true;
  };
  for (int im = 0; im < N - 1; im++) {
    const int cur = im + 1, prev = im;
    const int qdist = (int)(FQ(prev) - FQE(cur));
    const int tdist = (FT(prev) > FTE(cur)) ? (int)(FT(prev) - FTE(cur)) : (int)(FTE(cur) - FT(prev));
    const int dist = min(qdist, tdist);
    const long long dc = FS(cur) == 1 ? (long long)FQE(cur) + (long long)FT(cur) : (long long)FT(cur) - (long long)FQ(cur);   // Chain.h:243
    const long long dp = FS(prev) == 1 ? (long long)FQE(prev) + (long long)FT(prev) : (long long)FT(prev) - (long long)FQ(prev);
    const long long dd = dc > dp ? dc - dp : dp - dc;
    if (FS(cur) == FS(prev) && dist >= 1000 && (double)dd <= ceil(0.15 * (double)dist)) {
      if (push_new(runStart, cur)) { spl[nl++] = 0; spType[ns - 1] = 'N'; }
      runStart = cur;
    } else if (FT(cur) > FTE(prev) + (uint32_t)a.splitdist || FTE(cur) + (uint32_t)a.splitdist < FT(prev)) {
      if (push_new(runStart, cur)) { spl[nl++] = 0; spType[ns - 1] = 'T'; }
      runStart = cur;
    } else if (FS(cur) != FS(prev)) {
      if (push_new(runStart, cur)) { spType[ns - 1] = 'I'; spl[nl++] = 1; }
      runStart = cur;
    }
  }
  if (N > 0) push_new(runStart, N);
  // ---- MergeSplitchainINS :172-262
  uint8_t* keepS = a.keepS + base;
  for (int k = 0; k < ns; k++) keepS[k] = 1;
  if (ns >= 3) {
    uint32_t* curInd = a.curInd + base;
    for (int k = 0; k < ns; k++) curInd[k] = (uint32_t)k;
    bool change = false;
    int i0 = 0;
    while (i0 + 3 <= ns) {
      const int cc = (int)curInd[i0];
      if (spType[cc] != 'T') { i0++; continue; }
      int nn = (int)curInd[i0 + 2];
      while (nn < ns) {
        const long long cTS = spBox[4 * cc + 2], nTE = spBox[4 * nn + 3];
        const long long tdist = cTS > nTE ? cTS - nTE : nTE - cTS;
        if (tdist > 1500 || spStrand[cc] != spStrand[nn] || spChrom[cc] != spChrom[nn]) { nn++; continue; }
        change = true;
        uint32_t tail = (uint32_t)cc;                                     // append nn's runs behind cc's
        while (runNext[tail] != NONE) tail = runNext[tail];
        runNext[tail] = (uint32_t)nn;
        spBox[4 * cc] = min(spBox[4 * cc], spBox[4 * nn]); spBox[4 * cc + 2] = min(spBox[4 * cc + 2], spBox[4 * nn + 2]);
        spBox[4 * cc + 1] = max(spBox[4 * cc + 1], spBox[4 * nn + 1]); spBox[4 * cc + 3] = max(spBox[4 * cc + 3], spBox[4 * nn + 3]);
        spType[cc] = spType[nn];
        curInd[nn] = curInd[cc];
        keepS[nn] = 0;
        break;
      }
      i0 = nn;
    }
    if (change) {
      int rcount = 0;
      for (int k = 0; k < ns; k++) rcount += keepS[k];
      for (int x = nl; x < rcount - 1; x++) spl[x] = 0;                  // vector<bool>::resize fills with false
      nl = rcount - 1;
      if (a.bypass) { int x = 0; for (int k = 0; k < ns; k++) if (keepS[k]) { if (x > 0) spl[x - 1] = spType[k] == 'I'; x++; } }
    }
  }
  // ---- RemoveSpuriousSplitChain  Map_lowacc.h:38-66  (sizes of the merged pieces), then lay the survivors out
  int total = 0;
  for (int k = 0; k < ns; k++) if (keepS[k]) { for (uint32_t x = (uint32_t)k; x != NONE; x = runNext[x]) total += (int)(runB[x] - runA[x]); }
  const int filter = max((int)floorf(0.02f * (float)total), 2), filterDI = max((int)floorf(0.03f * (float)total), 2);
  uint32_t* spBeg = a.spBeg + base; uint32_t* spLen = a.spLen + base; uint32_t* spIdx = a.spIdx + base; uint8_t* spLink = a.spLink + base;
  uint32_t* ciBeg = a.ciBeg + base; uint32_t* ciLen = a.ciLen + base; uint32_t* ciIdx = a.ciIdx + base; uint8_t* splitLink = a.splitLink + base;
  int outK = 0, o = 0, co = 0, idx = 0;                                    // idx: position among the kept (merged) pieces
  for (int k = 0; k < ns && !ub; k++) {
    if (!keepS[k]) continue;
    int sz = 0;
    for (uint32_t x = (uint32_t)k; x != NONE; x = runNext[x]) sz += (int)(runB[x] - runA[x]);
    bool rm = sz < min(filter, 2);
    if (idx > 0) {
      if (idx - 1 >= nl) { ub = true; break; }
      if (spl[idx - 1] == 1 && sz < min(filterDI, 4)) rm = true;
    }
    if (!rm) {
      // sptc in SPLITChain's order; link[j] joins sptc[j] and sptc[j+1] (0 across a merge); forward pieces are reversed at the end (:436)
      const int beg = o;
      for (uint32_t x = (uint32_t)k; x != NONE; x = runNext[x]) {
        for (uint32_t p = runA[x]; p < runB[x]; p++) {
          spIdx[o] = p;
          if (o > beg) spLink[o - 1] = (p == runA[x]) ? (uint8_t)0 : link[p - 1];   // 0 across a merge (:205)
          o++;
        }
      }
      // ClusterIndex: consecutive-distinct clusters along the piece (push_new :352-356, merge :226-236)
      const int cbeg = co;
      for (uint32_t x = (uint32_t)k; x != NONE; x = runNext[x]) {
        if (x != (uint32_t)k && !a.bypass) break;
        for (uint32_t p = runA[x]; p < runB[x]; p++) { const uint32_t cl = Cl[fidx[p]]; if (co == cbeg || ciIdx[co - 1] != cl) ciIdx[co++] = cl; }
      }
      if (spStrand[k] == 0) {
        for (int x = beg, y = o - 1; x < y; x++, y--) { const uint32_t tv = spIdx[x]; spIdx[x] = spIdx[y]; spIdx[y] = tv; }
        for (int x = beg, y = o - 2; x < y; x++, y--) { const uint8_t tv = spLink[x]; spLink[x] = spLink[y]; spLink[y] = tv; }
      }
      spBeg[outK] = (uint32_t)beg; spLen[outK] = (uint32_t)(o - beg); ciBeg[outK] = (uint32_t)cbeg; ciLen[outK] = (uint32_t)(co - cbeg);
      // per-split attributes move from slot k to slot outK (outK <= k)
      spType[outK] = spType[k]; spStrand[outK] = spStrand[k]; spChrom[outK] = spChrom[k];
      spBox[4 * outK] = spBox[4 * k]; spBox[4 * outK + 1] = spBox[4 * k + 1]; spBox[4 * outK + 2] = spBox[4 * k + 2]; spBox[4 * outK + 3] = spBox[4 * k + 3];
      if (outK > 1) splitLink[outK - 1] = spl[idx - 1];
      else if (outK == 1) splitLink[0] = spl[0];                          // `if (c > 1)` in the reference: slot 0 keeps spchain_link[0]
      outK++;
    }
    idx++;
  }
  if (ub) { a.status[s] = LRA_ST_OOB_SLOT; return; }
  a.nSplit[s] = (uint32_t)outK;
  a.nSplitLink[s] = outK > 1 ? (uint32_t)(outK - 1) : 0;
#undef FQ
#undef FT
#undef FL
#undef FS
#undef FQE
#undef FTE
}

// ---- the chain filters of Chain.h on arbitrary chains: RemoveSmallPairedIndels :546 (op 1), RemovePairedIndels :607 (op 2; op 3 with
// refineEnds = false), RemoveSpuriousAnchors :828 (op 4, leaves `link` alone), RemoveSpuriousJump :897 (op 8), in the order given.
// One lane per chain; every filter is two streaming passes (the reference's SV list is only ever compared with its previous entry).
struct FilterArgs {
  uint64_t n; const uint64_t* off; const uint32_t* q; const uint32_t* t; const int32_t* len; const uint8_t* strand; const uint8_t* link; const uint32_t* qend;
  int ops[8]; int nOps;
  uint8_t* keep; uint32_t* nKept; uint8_t* linkOut; uint32_t* nLink; uint32_t* idx; uint8_t* rm;
};

__global__ void __launch_bounds__(64) filter_kernel(FilterArgs a) {
  const uint64_t c0 = (uint64_t)blockIdx.x * 64 + threadIdx.x;
  if (c0 >= a.n) return;
  const uint64_t base = a.off[c0];
  const int n = (int)(a.off[c0 + 1] - base);
  const uint32_t* Q = a.q + base; const uint32_t* T = a.t + base; const int32_t* Ln = a.len + base; const uint8_t* St = a.strand + base;
  const uint32_t* QE = a.qend ? a.qend + base : nullptr;   // FinalChain::qEnd is not qStart + length (Clustering.h:378-380)
  uint8_t* keep = a.keep + base; uint8_t* lk = a.linkOut + base; uint32_t* idx = a.idx + base; uint8_t* rm = a.rm + base;
  int N = n;
  int nl = (a.link && n > 0) ? n - 1 : 0;
  const bool hasLink = a.link != nullptr && nl > 0;
  for (int i = 0; i < n; i++) idx[i] = (uint32_t)i;
  for (int i = 0; i < nl; i++) lk[i] = a.link[base + i];
#define XQ(i) Q[idx[i]]
#define XT(i) T[idx[i]]
#define XL(i) Ln[idx[i]]
#define XS(i) St[idx[i]]
#define XQE(i) (QE ? QE[idx[i]] : Q[idx[i]] + (uint32_t)Ln[idx[i]])
#define XTE(i) (T[idx[i]] + (uint32_t)Ln[idx[i]])
  for (int oi = 0; oi < a.nOps; oi++) {
    const int op = a.ops[oi];
    if (N < 2) continue;
    for (int i = 0; i < N; i++) rm[i] = 0;
    auto gap_of = [&](int c) -> int {
      if (XS(c) == 0) return (int)(((long long)XT(c) - (long long)XQ(c)) - ((long long)XT(c - 1) - (long long)XQ(c - 1)));
      return (int)((long long)(uint32_t)(XQE(c) + XT(c)) - (long long)(uint32_t)(XQE(c - 1) + XT(c - 1)));
    };
    auto dists = [&](int c, long long& tDist, long long& qDist) {         // Chain.h:617-630
      if (XT(c) > XTE(c - 1)) tDist = (uint32_t)(XT(c) - XTE(c - 1)); else tDist = (uint32_t)(XT(c - 1) - XTE(c));
      if (XQ(c) > XQE(c - 1)) qDist = (uint32_t)(XQ(c) - XTE(c - 1)); else qDist = (uint32_t)(XQ(c - 1) - XQE(c));
    };
    float meanDist = 0, sdDist = 0;
    const bool refineEnds = op == 2;
    if (refineEnds) {
      // `totDistSq += dist*dist` overflows `long` when a distance wrapped around 2^32 (the q distance mixes in tEnd); the reference binary
      // (x86-64, two's complement wrap, signed conversion) then sees a negative sum, a NaN deviation and calls every distance invalid.
      long long totalDist = 0; unsigned long long totDistSqU = 0;
      for (int c = 1; c < N; c++) { long long tD, qD; dists(c, tD, qD); const long long d = min(tD, qD); totDistSqU += (unsigned long long)d * (unsigned long long)d; totalDist += d; }
      const long long totDistSq = (long long)totDistSqU;
      const float nDist = (float)(N - 1);
      meanDist = (float)totalDist / nDist;
      const float varDist = (float)totDistSq / nDist - meanDist * meanDist;
      sdDist = __fsqrt_rn(varDist);
    }
    int pSV = 0, pPos = -1; bool have = false;
    if (op == 5) {                                                       // RemovePairedIndels(GenomePairs&, chain, lengths)  Chain.h:753-811
      int pG = 0;
      for (int c = 1; c < N; c++) {
        const int Gap = (int)(((long long)XT(c) - (long long)XQ(c)) - ((long long)XT(c - 1) - (long long)XQ(c - 1)));
        if (abs(Gap) <= 30) continue;
        const int sv = Gap, g = (int)XT(c);
        if (have) {
          const int blink = max(abs(sv), abs(pSV));
          const bool pos = sv >= 0, differ = (sv >= 0) != (pSV >= 0);
          const int lim = max(2 * blink, 1000);
          const int dIns = abs(g - pG), dDel = abs(g - sv - pG);
          bool hit = false;
          if (differ && abs(sv + pSV) < 600 && abs(sv) != 0 && pSV != 0) hit = (pos && dIns < lim) || (!pos && dDel < lim);
          else if (differ && sv != 0 && pSV != 0 && ((pos && dIns < 500) || (!pos && dDel < 500))) hit = true;
          else if (!differ && sv != 0 && pSV != 0) hit = (pos && dIns < lim) || (!pos && dDel < lim);
          if (hit) for (int i = pPos; i < c; i++) if (XL(i) < 100) rm[i] = 1;
        }
        pSV = sv; pPos = c; pG = g; have = true;
      }
    }
    for (int c = 1; c < N && op != 5; c++) {
      int sv = 0; bool is = false;
      if (XS(c) == XS(c - 1)) {
        const int Gap = gap_of(c), ag = abs(Gap);
        const bool in = op == 1 ? (ag > 5 && ag <= 50) : op == 8 ? ag > 100 : op == 4 ? ag >= 500 : ag > 30;
        if (in) { sv = Gap; is = true; }
      } else { sv = 0; is = true; }
      if (!is) continue;
      if (have) {
        const bool opp = ((sv >= 0) != (pSV >= 0)) && sv != 0 && pSV != 0;
        if (op == 1) { if (opp && abs(sv + pSV) <= 20 && c - pPos < 3) for (int i = pPos; i < c; i++) if (XL(i) <= 50) rm[i] = 1; }
        else if (op == 8) { if (rm[pPos] == 0 && opp && c - pPos == 1) for (int i = pPos; i < c; i++) if (XL(i) < 50) rm[i] = 1; }
        else if (op == 4) {
          if (sv != 0 && pSV != 0 && c - pPos <= 10) {
            bool check = false;
            for (int b = pPos; b < c; b++) if (XL(b) >= 50) { check = true; break; }
            if (!check) for (int i = pPos; i < c; i++) if (XL(i) < 50) rm[i] = 1;
          }
        } else {
          if (opp && abs(sv) >= 300 && abs(pSV) >= 300 && c - pPos < 3) for (int i = pPos; i < c; i++) if (XL(i) < 100) rm[i] = 1;
          if (opp && abs(sv + pSV) < 100 && c - pPos < 3) for (int i = pPos; i < c; i++) if (XL(i) < 100) rm[i] = 1;
        }
      }
      pSV = sv; pPos = c; have = true;
    }
    if (refineEnds) {
      int firstValid = -1, lastValid = -1;
      for (int c = 1; c < N; c++) {
        long long tD, qD; dists(c, tD, qD);
        const int dist = (int)min(tD, qD);
        if ((float)dist < meanDist + 4 * sdDist) { if (firstValid == -1) firstValid = c - 1; lastValid = c; }
      }
      if (lastValid == -1 || firstValid == -1) for (int i = 0; i < N; i++) if (XL(i) < 100) rm[i] = 1;
      if (firstValid > 0 && firstValid < 3) for (int i = 0; i < firstValid; i++) if (XL(i) < 100) rm[i] = 1;
      if (lastValid + 1 <= N && N - lastValid < 3) for (int i = lastValid + 1; i < N; i++) if (XL(i) < 100) rm[i] = 1;
    }
    const bool touchLink = op != 4 && op != 5;
    int m = 0;
    for (int i = 0; i < N; i++)
      if (!rm[i]) {
        idx[m] = idx[i];
        if (touchLink && hasLink && nl > 0 && m >= 1) lk[m - 1] = lk[i - 1];
        m++;
      }
    N = m;
    if (touchLink && hasLink && nl > 0) { if (op == 2 || op == 3) { if (m > 0) nl = m - 1; } else nl = m - 1; }
  }
  for (int i = 0; i < n; i++) keep[i] = 0;
  for (int i = 0; i < N; i++) keep[idx[i]] = 1;
  a.nKept[c0] = (uint32_t)N; a.nLink[c0] = (uint32_t)nl;
#undef XQ
#undef XT
#undef XL
#undef XS
#undef XQE
#undef XTE
}

inline size_t sz(size_t n, size_t e) { return (n * e + 255) / 256 * 256; }

}  // namespace

extern "C" int lra_split_chains_batch(lra_ctx* ctx, const lra_chain_result* ch, const uint64_t* h_chrom_pos, int n_chrom, int splitdist,
                                      int bypass_clustering, lra_split_result* out) {
  if (!ctx || !ch || !out || !h_chrom_pos || n_chrom < 1) return LRA_ERR_INVALID;
  memset(out, 0, sizeof *out);
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const uint64_t slots = (uint64_t)ch->n_reads * ch->num_aln, NF = ch->n_frags;
  out->n_slots = slots; out->n_frags = NF;
  if (slots == 0) return LRA_OK;
  const size_t npos = (size_t)n_chrom + 1;
  size_t need = sz(NF + 1, 1) * 8 + sz(NF + 1, 4) * 12 + sz(4 * NF + 4, 4) + sz(slots + 1, 4) * 4 + sz(npos, 8) + 4096;
  char* w = (char*)lra_ensure(ctx, 13, need);
  if (!w) return LRA_ERR_NOMEM;
  auto take = [](char*& p, size_t n, size_t e) { char* r = p; p += sz(n, e); return r; };
  SplitArgs a;
  a.n_slots = slots; a.chainStart = ch->d_chain_start; a.chainLen = ch->d_chain_len; a.nChains = ch->d_n_chains; a.numAln = ch->num_aln;
  a.cq = ch->d_chain_q; a.ct = ch->d_chain_t; a.clen = ch->d_chain_alen; a.cstrand = ch->d_chain_strand; a.ccl = ch->d_chain_cluster; a.clink = ch->d_chain_link;
  uint64_t* dpos = (uint64_t*)take(w, npos, 8);
  a.pos = dpos; a.npos = (int)npos; a.splitdist = splitdist; a.bypass = bypass_clustering;
  a.keep = (uint8_t*)take(w, NF + 1, 1); a.link = (uint8_t*)take(w, NF + 1, 1); a.spLink = (uint8_t*)take(w, NF + 1, 1); a.spType = (uint8_t*)take(w, NF + 1, 1);
  a.spStrand = (uint8_t*)take(w, NF + 1, 1); a.splitLink = (uint8_t*)take(w, NF + 1, 1); a.keepS = (uint8_t*)take(w, NF + 1, 1); a.spl0 = (uint8_t*)take(w, NF + 1, 1);
  a.spBeg = (uint32_t*)take(w, NF + 1, 4); a.spLen = (uint32_t*)take(w, NF + 1, 4); a.spIdx = (uint32_t*)take(w, NF + 1, 4); a.spChrom = (int32_t*)take(w, NF + 1, 4);
  a.ciBeg = (uint32_t*)take(w, NF + 1, 4); a.ciLen = (uint32_t*)take(w, NF + 1, 4); a.ciIdx = (uint32_t*)take(w, NF + 1, 4); a.fidx = (uint32_t*)take(w, NF + 1, 4);
  a.runA = (uint32_t*)take(w, NF + 1, 4); a.runB = (uint32_t*)take(w, NF + 1, 4); a.runNext = (uint32_t*)take(w, NF + 1, 4); a.curInd = (uint32_t*)take(w, NF + 1, 4);
  a.spBox = (uint32_t*)take(w, 4 * NF + 4, 4);
  a.nKept = (uint32_t*)take(w, slots + 1, 4); a.nSplit = (uint32_t*)take(w, slots + 1, 4); a.nSplitLink = (uint32_t*)take(w, slots + 1, 4); a.status = (uint32_t*)take(w, slots + 1, 4);
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(dpos, h_chrom_pos, npos * 8, hipMemcpyHostToDevice, st));
  lra_time_begin(ctx, "chain_split");
  hipLaunchKernelGGL(split_kernel, dim3((unsigned)((slots + 63) / 64)), dim3(64), 0, st, a);
  lra_time_end(ctx);
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  LRA_HIP_CHECK(ctx, hipGetLastError());
  out->d_keep = a.keep; out->d_n_kept = a.nKept; out->d_link = a.link; out->d_n_split = a.nSplit; out->d_sp_beg = a.spBeg; out->d_sp_len = a.spLen;
  out->d_sp_idx = a.spIdx; out->d_sp_link = a.spLink; out->d_sp_type = a.spType; out->d_sp_strand = a.spStrand; out->d_sp_chrom = a.spChrom; out->d_sp_box = a.spBox;
  out->d_ci_beg = a.ciBeg; out->d_ci_len = a.ciLen; out->d_ci_idx = a.ciIdx; out->d_split_link = a.splitLink; out->d_n_split_link = a.nSplitLink; out->d_status = a.status;
  out->d_fidx = a.fidx;
  return LRA_OK;
}

extern "C" int lra_filter_chains_ex_batch(lra_ctx* ctx, uint64_t n_chains, const uint64_t* d_off, uint64_t n_anchors, const uint32_t* d_q, const uint32_t* d_t,
                                          const int32_t* d_len, const uint32_t* d_qend, const uint8_t* d_strand, const uint8_t* d_link, const int32_t* h_ops, int n_ops,
                                          lra_filter_result* out);
extern "C" int lra_filter_chains_batch(lra_ctx* ctx, uint64_t n_chains, const uint64_t* d_off, uint64_t n_anchors, const uint32_t* d_q, const uint32_t* d_t,
                                       const int32_t* d_len, const uint8_t* d_strand, const uint8_t* d_link, const int32_t* h_ops, int n_ops,
                                       lra_filter_result* out) {
  return lra_filter_chains_ex_batch(ctx, n_chains, d_off, n_anchors, d_q, d_t, d_len, nullptr, d_strand, d_link, h_ops, n_ops, out);
}
extern "C" int lra_filter_chains_ex_batch(lra_ctx* ctx, uint64_t n_chains, const uint64_t* d_off, uint64_t n_anchors, const uint32_t* d_q, const uint32_t* d_t,
                                          const int32_t* d_len, const uint32_t* d_qend, const uint8_t* d_strand, const uint8_t* d_link, const int32_t* h_ops, int n_ops,
                                          lra_filter_result* out) {
  if (!ctx || !out || !h_ops || n_ops < 0 || n_ops > 8) return LRA_ERR_INVALID;
  for (int i = 0; i < n_ops; i++) if (h_ops[i] != 1 && h_ops[i] != 2 && h_ops[i] != 3 && h_ops[i] != 4 && h_ops[i] != 5 && h_ops[i] != 8) return lra_set_err(ctx, LRA_ERR_INVALID, "unknown chain filter %d", h_ops[i]);
  memset(out, 0, sizeof *out);
  out->n_chains = n_chains; out->n_anchors = n_anchors;
  if (n_chains == 0) return LRA_OK;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  auto take = [](char*& p, size_t n, size_t e) { char* r = p; p += sz(n, e); return r; };
  char* w = (char*)lra_ensure(ctx, 13, sz(n_anchors + 1, 1) * 3 + sz(n_anchors + 1, 4) + sz(n_chains + 1, 4) * 2 + 4096);
  if (!w) return LRA_ERR_NOMEM;
  FilterArgs a;
  a.n = n_chains; a.off = d_off; a.q = d_q; a.t = d_t; a.len = d_len; a.strand = d_strand; a.link = d_link; a.qend = d_qend; a.nOps = n_ops;
  for (int i = 0; i < 8; i++) a.ops[i] = i < n_ops ? h_ops[i] : 0;
  a.keep = (uint8_t*)take(w, n_anchors + 1, 1); a.linkOut = (uint8_t*)take(w, n_anchors + 1, 1); a.rm = (uint8_t*)take(w, n_anchors + 1, 1);
  a.idx = (uint32_t*)take(w, n_anchors + 1, 4); a.nKept = (uint32_t*)take(w, n_chains + 1, 4); a.nLink = (uint32_t*)take(w, n_chains + 1, 4);
  lra_time_begin(ctx, "chain_filter");
  hipLaunchKernelGGL(filter_kernel, dim3((unsigned)((n_chains + 63) / 64)), dim3(64), 0, st, a);
  lra_time_end(ctx);
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  LRA_HIP_CHECK(ctx, hipGetLastError());
  out->d_keep = a.keep; out->d_n_kept = a.nKept; out->d_link = a.linkOut; out->d_n_link = a.nLink;
  return LRA_OK;
}

// ---- a9, high-accuracy path: SPLITChain over Cluster_SameDiag elements (Mapping_ultility.h:266-346) + MergeSplitchainINS (:172-262) +
// LargestSplitChain_dist (Chain.h:974-985), one lane per chain.  A piece is a run [a, b) of the chain, merges are a linked list of runs.  The last
// piece's chromIndex is indeterminate in the reference (never assigned, Chain.h:350-360): -1 here, so that it equals no other piece's.
namespace {

struct HSplitArgs {
  uint64_t nJobs;
  const uint64_t* jobOff; const uint64_t* linkOff; const uint8_t* link;
  const int32_t* strand; const int32_t* chrom; const uint32_t* box; int splitdist;
  uint32_t* nSplit; uint32_t* lsc; uint32_t* spLen; uint32_t* spIdx; uint8_t* spType; uint8_t* spStrand; uint32_t* spBox;
  uint32_t* runA; uint32_t* runB; uint32_t* runNext; uint32_t* curInd; uint8_t* keepS; int32_t* spChrom; uint8_t* ty0; uint8_t* st0; uint32_t* bx0;
};

__global__ void __launch_bounds__(64) hsplit_kernel(HSplitArgs a) {
  const uint64_t j = (uint64_t)blockIdx.x * 64 + threadIdx.x;
  if (j >= a.nJobs) return;
  const uint64_t base = a.jobOff[j];
  const int n = (int)(a.jobOff[j + 1] - base);
  a.nSplit[j] = 0; a.lsc[j] = 0;
  if (n == 0) return;
  const int32_t* St = a.strand + base; const int32_t* Ch = a.chrom + base; const uint32_t* B = a.box + 4 * base;
  const uint8_t* Lk = a.link + a.linkOff[j];
  uint32_t* runA = a.runA + base; uint32_t* runB = a.runB + base; uint32_t* runNext = a.runNext + base; uint32_t* curInd = a.curInd + base;
  uint8_t* keepS = a.keepS + base; int32_t* spChrom = a.spChrom + base; uint8_t* ty = a.ty0 + base; uint8_t* sd = a.st0 + base; uint32_t* bx = a.bx0 + 4 * base;
  auto ovl = [&](int x, int y) -> float {                                 // x->OverlaprateOnGenome(y)  Clustering.h:397-406
    const uint32_t xs = B[4 * x + 2], xe = B[4 * x + 3], ys = B[4 * y + 2], ye = B[4 * y + 3];
    if (xe <= ys || ye <= xs) return 0.f;
    const int ovp = (int)(min(xe, ye) - max(xs, ys));
    return __fdiv_rn((float)ovp, (float)(xe - xs));
  };
  int ns = 0, runStart = 0;
  auto push = [&](int from, int to, char type, int chromIndex, int strand) {
    runA[ns] = (uint32_t)from; runB[ns] = (uint32_t)to; runNext[ns] = NONE; ty[ns] = (uint8_t)type; spChrom[ns] = chromIndex; sd[ns] = (uint8_t)strand;
    uint32_t qs = B[4 * from], qe = B[4 * from + 1], ts = B[4 * from + 2], te = B[4 * from + 3];
    for (int v = from + 1; v < to; v++) { qs = min(qs, B[4 * v]); qe = max(qe, B[4 * v + 1]); ts = min(ts, B[4 * v + 2]); te = max(te, B[4 * v + 3]); }
    bx[4 * ns] = qs; bx[4 * ns + 1] = qe; bx[4 * ns + 2] = ts; bx[4 * ns + 3] = te;
    ns++;
  };
  for (int im = 0; im < n - 1; im++) {
    const int cur = im + 1, prev = im;
    const bool rep = ((Lk[im] == 1 && St[cur] == 0 && St[prev] == 0) || (Lk[im] == 0 && St[cur] == 1 && St[prev] == 1)) && (double)ovl(prev, cur) >= 0.6 &&
                     (double)ovl(cur, prev) >= 0.6;
    char type = 0;
    if (B[4 * cur + 2] > B[4 * prev + 3] + (uint32_t)a.splitdist || B[4 * cur + 3] + (uint32_t)a.splitdist < B[4 * prev + 2] || Ch[cur] != Ch[prev]) type = 'T';
    else if (rep) type = 'D';
    else if ((St[cur] == 0 && St[prev] == 1) || (St[cur] == 1 && St[prev] == 0)) type = 'I';
    if (type) { push(runStart, cur, type, Ch[cur], St[prev]); runStart = cur; }
  }
  push(runStart, n, 'N', -1, St[n - 1]);
  for (int k = 0; k < ns; k++) keepS[k] = 1;
  if (ns >= 3) {                                                          // MergeSplitchainINS :172-262
    for (int k = 0; k < ns; k++) curInd[k] = (uint32_t)k;
    int i0 = 0;
    while (i0 + 3 <= ns) {
      const int cc = (int)curInd[i0];
      if (ty[cc] != 'T') { i0++; continue; }
      int nn = (int)curInd[i0 + 2];
      while (nn < ns) {
        const long long cTS = bx[4 * cc + 2], nTE = bx[4 * nn + 3];
        const long long tdist = cTS > nTE ? cTS - nTE : nTE - cTS;
        if (tdist > 1500 || sd[cc] != sd[nn] || spChrom[cc] != spChrom[nn]) { nn++; continue; }
        uint32_t tail = (uint32_t)cc;
        while (runNext[tail] != NONE) tail = runNext[tail];
        runNext[tail] = (uint32_t)nn;
        bx[4 * cc] = min(bx[4 * cc], bx[4 * nn]); bx[4 * cc + 2] = min(bx[4 * cc + 2], bx[4 * nn + 2]);
        bx[4 * cc + 1] = max(bx[4 * cc + 1], bx[4 * nn + 1]); bx[4 * cc + 3] = max(bx[4 * cc + 3], bx[4 * nn + 3]);
        ty[cc] = ty[nn];
        curInd[nn] = curInd[cc];
        keepS[nn] = 0;
        break;
      }
      i0 = nn;
    }
  }
  uint32_t* spLen = a.spLen + base; uint32_t* spIdx = a.spIdx + base; uint8_t* spType = a.spType + base; uint8_t* spStrand = a.spStrand + base; uint32_t* spBox = a.spBox + 4 * base;
  int outK = 0, o = 0, maxi = 0, maxi_d = 0;
  for (int k = 0; k < ns; k++) {
    if (!keepS[k]) continue;
    const int beg = o;
    for (uint32_t x = (uint32_t)k; x != NONE; x = runNext[x]) for (uint32_t v = runA[x]; v < runB[x]; v++) spIdx[o++] = v;
    spLen[outK] = (uint32_t)(o - beg); spType[outK] = ty[k]; spStrand[outK] = sd[k];
    const uint32_t qs = bx[4 * k], qe = bx[4 * k + 1];
    spBox[4 * outK] = qs; spBox[4 * outK + 1] = qe; spBox[4 * outK + 2] = bx[4 * k + 2]; spBox[4 * outK + 3] = bx[4 * k + 3];
    const int d = qe > qs ? (int)(qe - qs) : 0;                           // LargestSplitChain_dist
    if (outK == 0) maxi_d = d; else if (d > maxi_d) { maxi = outK; maxi_d = d; }
    outK++;
  }
  a.nSplit[j] = (uint32_t)outK; a.lsc[j] = (uint32_t)maxi;
}

// CSR of the pieces over all jobs: piece p of job j gets its length, attributes and the offset of its elements
__global__ void hsplit_lay(uint64_t nJobs, const uint64_t* jobOff, const uint64_t* jobPieceOff, const uint32_t* spLen, const uint8_t* spType, const uint8_t* spStrand,
                           const uint32_t* spBox, uint64_t* pieceOff, uint8_t* oType, uint8_t* oStrand, uint32_t* oBox, uint32_t* pieceJob, uint64_t nPieces, uint64_t nElems) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j == 0) pieceOff[nPieces] = nElems;
  if (j >= nJobs) return;
  const uint64_t base = jobOff[j], p0 = jobPieceOff[j];
  const int np = (int)(jobPieceOff[j + 1] - p0);
  uint64_t o = base;
  for (int k = 0; k < np; k++) {
    pieceOff[p0 + k] = o; o += spLen[base + k]; oType[p0 + k] = spType[base + k]; oStrand[p0 + k] = spStrand[base + k]; pieceJob[p0 + k] = (uint32_t)j;
    for (int x = 0; x < 4; x++) oBox[4 * (p0 + k) + x] = spBox[4 * (base + k) + x];
  }
}

}  // namespace

extern "C" int lra_split_chains_highacc_batch(lra_ctx* ctx, uint64_t n_jobs, const uint64_t* d_job_off, uint64_t n_elems, const int32_t* d_strand, const int32_t* d_chrom,
                                              const uint32_t* d_box, const uint64_t* d_link_off, const uint8_t* d_link, int splitdist, lra_hsplit_result* out) {
  if (!ctx || !out) return LRA_ERR_INVALID;
  memset(out, 0, sizeof *out);
  out->n_jobs = n_jobs;
  if (n_jobs == 0) return LRA_OK;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const uint64_t NE = n_elems;
  size_t need = sz(NE + 1, 4) * 8 + sz(NE + 1, 1) * 7 + sz(4 * NE + 4, 4) * 3 + sz(n_jobs + 2, 4) * 2 + sz(n_jobs + 2, 8) + sz(NE + 2, 8) + 4096;
  char* w = (char*)lra_ensure(ctx, 93, need);
  if (!w) return LRA_ERR_NOMEM;
  auto take = [](char*& p, size_t n, size_t e) { char* r = p; p += sz(n, e); return r; };
  HSplitArgs a;
  a.nJobs = n_jobs; a.jobOff = d_job_off; a.linkOff = d_link_off; a.link = d_link; a.strand = d_strand; a.chrom = d_chrom; a.box = d_box; a.splitdist = splitdist;
  a.nSplit = (uint32_t*)take(w, n_jobs + 2, 4); a.lsc = (uint32_t*)take(w, n_jobs + 2, 4);
  uint64_t* jobPieceOff = (uint64_t*)take(w, n_jobs + 2, 8);
  a.spLen = (uint32_t*)take(w, NE + 1, 4); a.spIdx = (uint32_t*)take(w, NE + 1, 4); a.runA = (uint32_t*)take(w, NE + 1, 4); a.runB = (uint32_t*)take(w, NE + 1, 4);
  a.runNext = (uint32_t*)take(w, NE + 1, 4); a.curInd = (uint32_t*)take(w, NE + 1, 4); a.spChrom = (int32_t*)take(w, NE + 1, 4);
  uint32_t* pieceJob = (uint32_t*)take(w, NE + 1, 4);
  a.spType = (uint8_t*)take(w, NE + 1, 1); a.spStrand = (uint8_t*)take(w, NE + 1, 1); a.keepS = (uint8_t*)take(w, NE + 1, 1); a.ty0 = (uint8_t*)take(w, NE + 1, 1);
  a.st0 = (uint8_t*)take(w, NE + 1, 1);
  uint8_t* oType = (uint8_t*)take(w, NE + 1, 1); uint8_t* oStrand = (uint8_t*)take(w, NE + 1, 1);
  a.spBox = (uint32_t*)take(w, 4 * NE + 4, 4); a.bx0 = (uint32_t*)take(w, 4 * NE + 4, 4);
  uint32_t* oBox = (uint32_t*)take(w, 4 * NE + 4, 4);
  uint64_t* pieceOff = (uint64_t*)take(w, NE + 2, 8);
  lra_time_begin(ctx, "chain_split_highacc");
  hipLaunchKernelGGL(hsplit_kernel, dim3((unsigned)((n_jobs + 63) / 64)), dim3(64), 0, st, a);
  lra_time_end(ctx);
  { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)n_jobs, a.nSplit, jobPieceOff); if (rc) return rc; }
  uint64_t nPieces = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&nPieces, jobPieceOff + n_jobs, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  hipLaunchKernelGGL(hsplit_lay, dim3((unsigned)((n_jobs + 255) / 256)), dim3(256), 0, st, n_jobs, d_job_off, (const uint64_t*)jobPieceOff, (const uint32_t*)a.spLen,
                     (const uint8_t*)a.spType, (const uint8_t*)a.spStrand, (const uint32_t*)a.spBox, pieceOff, oType, oStrand, oBox, pieceJob, nPieces, NE);
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  LRA_HIP_CHECK(ctx, hipGetLastError());
  out->n_pieces = nPieces; out->n_elems = NE; out->d_job_piece_off = jobPieceOff; out->d_piece_off = pieceOff; out->d_sptc = a.spIdx; out->d_piece_type = oType;
  out->d_piece_strand = oStrand; out->d_piece_box = oBox; out->d_piece_job = pieceJob; out->d_job_lsc = a.lsc;
  return LRA_OK;
}
