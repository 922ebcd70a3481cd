// oracle/chain_split.cpp -- TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// CPU restatement of what MapRead_lowacc does to every chain of the first sparse DP before tier-2 refinement
// (Map_lowacc.h:189-192, :252-256):
//   RemoveSpuriousJump<UltimateChain>                          Chain.h:897-957
//   SPLITChain(genome, read, chain, spchain, spchain_link)     Mapping_ultility.h:385-441, push_new :349-383,
//     SplitChain::CHROMIndex Chain.h:386-394 (GenomeHeader::Find Genome.h:20-32), UltimateChain::diag Chain.h:243-246
//   MergeSplitchainINS                                         Mapping_ultility.h:172-262
//   RemoveSpuriousSplitChain                                   Map_lowacc.h:38-66
// Parity status: PARITY UNPINNED -- Chain.h / Mapping_ultility.h include Genome.h (htslib); restated from the source text.
#include "oracle_common.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <numeric>
#include <vector>

namespace {

struct Anchor { uint32_t q, t; int len; int strand; int cluster; };   // one chain element with its cluster's strand (0 forward)

struct Split {
  std::vector<int> sptc;
  std::vector<bool> link;
  uint32_t QStart = 0, QEnd = 0, TStart = 0, TEnd = 0;
  int chromIndex = 0;
  bool Strand = 0;
  std::vector<int> ClusterIndex;
  char type = 'N';
};

int header_find(const uint64_t* pos, int npos, uint64_t query, bool& ub) {   // Genome.h:20-32
  if (npos > 0 && query == pos[0]) return 0;
  const uint64_t* it = std::lower_bound(pos, pos + npos, query);
  int i = (int)(it - pos);
  if (i == npos) { ub = true; return i - 1; }
  if (query == *it) return i;
  return i - 1;
}

inline bool sgn(int x) { return x >= 0; }                               // sign() Clustering.h:544

}  // namespace

// In: one chain in trace-back order (anchor i: q, t(global), len, strand of its cluster, cluster index) with its link bits
// (n - 1 of them); the chromosome table pos[0..npos) (genome.header.pos).  splitdist = opts.splitdist, bypass = opts.bypassClustering.
// Out: keep[i] (RemoveSpuriousJump), link of the filtered chain; the split chains as CSR over indices INTO THE FILTERED CHAIN
// (spOff[nSplit+1], spIdx), per split: type char, Strand, chromIndex, box (QStart,QEnd,TStart,TEnd), link bits CSR-aligned with spIdx
// (entry k links spIdx[k] to spIdx[k+1]), ClusterIndex CSR (ciOff, ciIdx); splitLink[nSplit-1].  Returns nSplit, or -1 if the reference
// would read outside its arrays.
extern "C" int oracle_split_chain(int n, const uint32_t* q, const uint32_t* t, const int* len, const uint8_t* strand, const int* cluster,
                                  const uint8_t* link, const uint64_t* pos, int npos, int splitdist, int bypass, uint8_t* keep, int* nKept,
                                  uint8_t* linkOut, int* spOff, int* spIdx, uint8_t* spLink, char* spType, uint8_t* spStrand, int* spChrom,
                                  uint32_t* spBox, int* ciOff, int* ciIdx, uint8_t* splitLink, int* nSplitLink) {
  std::vector<Anchor> ch(n);
  for (int i = 0; i < n; i++) ch[i] = Anchor{q[i], t[i], len[i], (int)strand[i], cluster[i]};
  std::vector<bool> lk(link, link + (n > 0 ? n - 1 : 0));
  auto qS = [&](int i) { return ch[i].q; };
  auto tS = [&](int i) { return ch[i].t; };
  auto qE = [&](int i) { return ch[i].q + (uint32_t)ch[i].len; };
  auto tE = [&](int i) { return ch[i].t + (uint32_t)ch[i].len; };
  // ---- RemoveSpuriousJump :897-957
  std::vector<bool> remove(n, false);
  if (n >= 2) {
    std::vector<int> SV, SVpos;
    for (int c = 1; c < n; c++) {
      if (ch[c].strand == ch[c - 1].strand) {
        int Gap;
        if (ch[c].strand == 0) Gap = (int)(((long)tS(c) - (long)qS(c)) - ((long)tS(c - 1) - (long)qS(c - 1)));
        else Gap = (int)((long)(qE(c) + tS(c)) - (long)(qE(c - 1) + tS(c - 1)));     // 32-bit sums, as written
        if (std::abs(Gap) > 100) { SV.push_back(Gap); SVpos.push_back(c); }
      } else { SVpos.push_back(c); SV.push_back(0); }
    }
    for (size_t c = 1; c < SV.size(); c++) {
      if (remove[SVpos[c - 1]] == 0 && sgn(SV[c]) != sgn(SV[c - 1]) && SV[c] != 0 && SV[c - 1] != 0 && SVpos[c] - SVpos[c - 1] == 1)
        for (int i = SVpos[c - 1]; i < SVpos[c]; i++) if (ch[i].len < 50) remove[i] = true;
    }
    int m = 0;
    for (int i = 0; i < n; i++)
      if (!remove[i]) {
        ch[m] = ch[i];
        if (!lk.empty() && m >= 1) lk[m - 1] = lk[i - 1];
        m++;
      }
    ch.resize(m);
    if (!lk.empty()) lk.resize(m - 1);
  }
  for (int i = 0; i < n; i++) keep[i] = !remove[i];
  const int N = (int)ch.size();
  *nKept = N;
  for (size_t i = 0; i < lk.size(); i++) linkOut[i] = lk[i];
  // ---- SPLITChain :385-441
  bool ub = false;
  std::vector<Split> sp;
  std::vector<bool> spl;
  auto diag = [&](int i) -> long { return ch[i].strand == 1 ? (long)qE(i) + (long)tS(i) : (long)tS(i) - (long)qS(i); };   // Chain.h:243
  std::vector<int> onec;
  std::vector<bool> olk;
  auto push_new = [&](int cur) -> bool {                                 // :349-383
    Split s;
    s.sptc = onec; s.link = olk; s.Strand = ch[onec[0]].strand; s.type = 'N';
    s.ClusterIndex.push_back(ch[onec[0]].cluster);
    for (size_t c = 1; c < onec.size(); c++) if (ch[onec[c]].cluster != s.ClusterIndex.back()) s.ClusterIndex.push_back(ch[onec[c]].cluster);
    s.QStart = qS(onec.back()); s.QEnd = qE(onec[0]);
    if (ch[onec[0]].strand == 0) { s.TStart = tS(onec.back()); s.TEnd = tE(onec[0]); }
    else { s.TStart = tS(onec[0]); s.TEnd = tE(onec.back()); }
    int first = header_find(pos, npos, (uint64_t)s.TStart + 1, ub), last = header_find(pos, npos, (uint64_t)s.TEnd, ub);   // CHROMIndex
    onec.clear(); olk.clear(); onec.push_back(cur);
    if (first != last) return false;
    s.chromIndex = first;
    sp.push_back(s);
    return true;
  };
  if (N == 0) { *nSplitLink = 0; spOff[0] = 0; ciOff[0] = 0; return 0; }
  onec.push_back(0);
  int im = 0, cur = 0, prev = 0;
  while (im < N - 1) {
    cur = im + 1; prev = im;
    int qdist = (int)(qS(prev) - qE(cur));
    int tdist = (tS(prev) > tE(cur)) ? (int)(tS(prev) - tE(cur)) : (int)(tE(cur) - tS(prev));
    int dist = std::min(qdist, tdist);
    if (ch[cur].strand == ch[prev].strand && dist >= 1000 && std::labs(diag(cur) - diag(prev)) <= std::ceil(0.15 * dist)) {
      if (push_new(cur)) { spl.push_back(0); sp.back().type = 'N'; }
    } else if (tS(cur) > tE(prev) + (uint32_t)splitdist || tE(cur) + (uint32_t)splitdist < tS(prev)) {
      if (push_new(cur)) { spl.push_back(0); sp.back().type = 'T'; }
    } else if (ch[cur].strand != ch[prev].strand) {
      if (push_new(cur)) { sp.back().type = 'I'; spl.push_back(1); }
    } else { onec.push_back(cur); olk.push_back(lk[im]); }
    im++;
  }
  if (!onec.empty()) push_new(cur);
  // ---- MergeSplitchainINS :172-262
  if (sp.size() >= 3) {
    std::vector<int> cur_ind(sp.size());
    std::iota(cur_ind.begin(), cur_ind.end(), 0);
    std::vector<bool> keepS(sp.size(), true);
    bool change = false;
    size_t i0 = 0;
    while (i0 + 3 <= sp.size()) {
      int c = cur_ind[i0];
      if (sp[c].type != 'T') { i0++; continue; }
      size_t nn = (size_t)cur_ind[i0 + 2];
      while (nn < sp.size()) {
        long tdist = (sp[c].TStart > sp[nn].TEnd) ? ((long)sp[c].TStart - (long)sp[nn].TEnd) : ((long)sp[nn].TEnd - (long)sp[c].TStart);
        if (tdist > 1500) { nn++; continue; }
        if (sp[c].Strand != sp[nn].Strand) { nn++; continue; }
        if (sp[c].chromIndex != sp[nn].chromIndex) { nn++; continue; }
        change = true;
        int t1 = (int)sp[c].sptc.size(), tt = t1 + (int)sp[nn].sptc.size();
        sp[c].sptc.resize(tt); sp[c].link.resize(tt - 1);
        for (int s = t1; s < tt; s++) {
          sp[c].sptc[s] = sp[nn].sptc[s - t1];
          if (s == t1) sp[c].link[s - 1] = 0;
          else sp[c].link[s - 1] = sp[nn].link[s - t1 - 1];
          sp[c].QStart = std::min(sp[c].QStart, sp[nn].QStart); sp[c].TStart = std::min(sp[c].TStart, sp[nn].TStart);
          sp[c].QEnd = std::max(sp[c].QEnd, sp[nn].QEnd); sp[c].TEnd = std::max(sp[c].TEnd, sp[nn].TEnd);
          sp[c].type = sp[nn].type;
        }
        if (bypass) {
          int pv = sp[c].ClusterIndex.back();
          for (size_t s = 0; s < sp[nn].ClusterIndex.size(); s++) {
            int cu = sp[nn].ClusterIndex[s];
            if (pv != cu) { sp[c].ClusterIndex.push_back(cu); pv = cu; }
          }
        }
        cur_ind[nn] = cur_ind[c];
        keepS[nn] = false;
        break;
      }
      i0 = nn;
    }
    if (change) {
      size_t r = 0;
      for (size_t s = 0; s < sp.size(); s++) if (keepS[s]) { if (r != s) sp[r] = sp[s]; r++; }
      sp.resize(r);
      spl.resize(r - 1);
      if (bypass) for (size_t k = 1; k < sp.size(); k++) spl[k - 1] = sp[k].type == 'I';
    }
  }
  for (auto& s : sp)                                                     // :436-441
    if (s.Strand == 0) { std::reverse(s.sptc.begin(), s.sptc.end()); std::reverse(s.link.begin(), s.link.end()); }
  // ---- RemoveSpuriousSplitChain  Map_lowacc.h:38-66
  {
    int total = 0;
    for (auto& s : sp) total += (int)s.sptc.size();
    int filter = std::max((int)std::floor(0.02f * (float)total), 2);
    int filterDI = std::max((int)std::floor(0.03f * (float)total), 2);
    std::vector<bool> rm(sp.size(), 0);
    for (size_t i = 0; i < sp.size(); i++) {
      if ((int)sp[i].sptc.size() < std::min(filter, 2)) rm[i] = 1;
      if (i > 0) {
        if (i - 1 >= spl.size()) { ub = true; break; }
        if (spl[i - 1] == 1 && (int)sp[i].sptc.size() < std::min(filterDI, 4)) rm[i] = 1;
      }
    }
    if (ub) return -1;
    size_t c = 0;
    for (size_t i = 0; i < rm.size(); i++)
      if (!rm[i]) {
        if (c != i) sp[c] = sp[i];
        if (c > 1) spl[c - 1] = spl[i - 1];
        c++;
      }
    sp.resize(c);
    if (c > 1) spl.resize(c - 1); else spl.clear();
  }
  if (ub) return -1;
  int o = 0, co = 0;
  for (size_t k = 0; k < sp.size(); k++) {
    spOff[k] = o; ciOff[k] = co;
    for (size_t x = 0; x < sp[k].sptc.size(); x++) { spIdx[o + x] = sp[k].sptc[x]; spLink[o + x] = x < sp[k].link.size() ? (uint8_t)sp[k].link[x] : 0; }
    o += (int)sp[k].sptc.size();
    for (int x : sp[k].ClusterIndex) ciIdx[co++] = x;
    spType[k] = sp[k].type; spStrand[k] = sp[k].Strand; spChrom[k] = sp[k].chromIndex;
    spBox[4 * k] = sp[k].QStart; spBox[4 * k + 1] = sp[k].QEnd; spBox[4 * k + 2] = sp[k].TStart; spBox[4 * k + 3] = sp[k].TEnd;
  }
  spOff[sp.size()] = o; ciOff[sp.size()] = co;
  *nSplitLink = (int)spl.size();
  for (size_t k = 0; k < spl.size(); k++) splitLink[k] = spl[k];
  return (int)sp.size();
}

// ---- the other chain filters of Chain.h, applied in the order given by ops[] to one chain (trace-back order) -------------------
//   op 1  RemoveSmallPairedIndels :546-603     op 2  RemovePairedIndels(chain, refineEnds) :607-741 (op 3 = with refineEnds false)
//   op 4  RemoveSpuriousAnchors :828-890 (does not touch `link`)     op 8  RemoveSpuriousJump :897-957
// Map_lowacc.h:538-539 applies {2, 4} to the chain of the second sparse DP; LocalRefineAlignment.h:567-571 {1, 2 or 3, 4}.
// Out: keep[] over the ORIGINAL anchors, the surviving links (linkOut, *nLink of them: RemoveSpuriousAnchors leaves the vector longer than
// the chain, as the reference does).  Returns the number of surviving anchors.
extern "C" int oracle_filter_chain_ex(int n, const uint32_t* q, const uint32_t* t, const int* len, const uint32_t* qend, const uint8_t* strand, const uint8_t* link,
                                      int hasLink, const int* ops, int nOps, uint8_t* keep, uint8_t* linkOut, int* nLink);
extern "C" int oracle_filter_chain(int n, const uint32_t* q, const uint32_t* t, const int* len, const uint8_t* strand, const uint8_t* link, int hasLink,
                                   const int* ops, int nOps, uint8_t* keep, uint8_t* linkOut, int* nLink) {
  return oracle_filter_chain_ex(n, q, t, len, nullptr, strand, link, hasLink, ops, nOps, keep, linkOut, nLink);
}
// qend != NULL: the chain's qEnd(i) is qend[i] rather than q + len -- FinalChain::qEnd = Cluster_SameDiag::GetqEnd (Clustering.h:378-380), which adds the
// merged entry's length to its LAST anchor's read position.
extern "C" int oracle_filter_chain_ex(int n, const uint32_t* q, const uint32_t* t, const int* len, const uint32_t* qend, const uint8_t* strand, const uint8_t* link,
                                      int hasLink, const int* ops, int nOps, uint8_t* keep, uint8_t* linkOut, int* nLink) {
  struct A { uint32_t q, t; int len; int strand; int orig; };
  std::vector<A> ch(n);
  for (int i = 0; i < n; i++) ch[i] = A{q[i], t[i], len[i], (int)strand[i], i};
  std::vector<bool> lk;
  if (hasLink) lk.assign(link, link + (n > 0 ? n - 1 : 0));
  auto qS = [&](int i) { return ch[i].q; };
  auto tS = [&](int i) { return ch[i].t; };
  auto qE = [&](int i) { return qend ? qend[ch[i].orig] : ch[i].q + (uint32_t)ch[i].len; };
  auto tE = [&](int i) { return ch[i].t + (uint32_t)ch[i].len; };
  auto gap_of = [&](int c) -> int {
    if (ch[c].strand == 0) return (int)(((long)tS(c) - (long)qS(c)) - ((long)tS(c - 1) - (long)qS(c - 1)));
    return (int)((long)(qE(c) + tS(c)) - (long)(qE(c - 1) + tS(c - 1)));
  };
  for (int oi = 0; oi < nOps; oi++) {
    const int op = ops[oi];
    const int N = (int)ch.size();
    if (N < 2) continue;
    std::vector<bool> remove(N, false);
    std::vector<int> SV, SVpos;
    bool touchLink = true;
    if (op == 1 || op == 8) {
      for (int c = 1; c < N; c++) {
        if (ch[c].strand == ch[c - 1].strand) {
          int Gap = gap_of(c);
          bool in = op == 1 ? (std::abs(Gap) > 5 && std::abs(Gap) <= 50) : (std::abs(Gap) > 100);
          if (in) { SV.push_back(Gap); SVpos.push_back(c); }
        } else { SVpos.push_back(c); SV.push_back(0); }
      }
      for (size_t c = 1; c < SV.size(); c++) {
        if (op == 1) {
          if (sgn(SV[c]) != sgn(SV[c - 1]) && SV[c] != 0 && SV[c - 1] != 0 && std::abs(SV[c] + SV[c - 1]) <= 20 && SVpos[c] - SVpos[c - 1] < 3)
            for (int i = SVpos[c - 1]; i < SVpos[c]; i++) if (ch[i].len <= 50) remove[i] = true;
        } else if (remove[SVpos[c - 1]] == 0 && sgn(SV[c]) != sgn(SV[c - 1]) && SV[c] != 0 && SV[c - 1] != 0 && SVpos[c] - SVpos[c - 1] == 1)
          for (int i = SVpos[c - 1]; i < SVpos[c]; i++) if (ch[i].len < 50) remove[i] = true;
      }
    } else if (op == 2 || op == 3) {
      const bool refineEnds = op == 2;
      long totalDist = 0; unsigned long totDistSqU = 0;                      // dist*dist wraps like the reference binary's `long` does
      auto dists = [&](int c, long& tDist, long& qDist) {                  // :617-630 (GenomePos arithmetic; the q distance mixes in tEnd, as written)
        if (tS(c) > tE(c - 1)) tDist = tS(c) - tE(c - 1); else tDist = tS(c - 1) - tE(c);
        if (qS(c) > qE(c - 1)) qDist = qS(c) - tE(c - 1); else qDist = qS(c - 1) - qE(c);
      };
      for (int c = 1; c < N; c++) {
        if (refineEnds) { long tD, qD; dists(c, tD, qD); long dist = std::min(tD, qD); totDistSqU += (unsigned long)dist * (unsigned long)dist; totalDist += dist; }
        if (ch[c].strand == ch[c - 1].strand) { int Gap = gap_of(c); if (std::abs(Gap) > 30) { SV.push_back(Gap); SVpos.push_back(c); } }
        else { SVpos.push_back(c); SV.push_back(0); }
      }
      const long totDistSq = (long)totDistSqU;
      float nDist = N - 1;
      float meanDist = totalDist / nDist;
      float varDist = totDistSq / float(nDist) - meanDist * meanDist;
      float sdDist = std::sqrt(varDist);
      int firstValid = -1, lastValid = -1;
      for (size_t c = 1; c < SV.size(); c++) {
        if (sgn(SV[c]) != sgn(SV[c - 1]) && SV[c] != 0 && SV[c - 1] != 0 && std::abs(SV[c]) >= 300 && std::abs(SV[c - 1]) >= 300 && SVpos[c] - SVpos[c - 1] < 3)
          for (int i = SVpos[c - 1]; i < SVpos[c]; i++) if (ch[i].len < 100) remove[i] = true;
        if (sgn(SV[c]) != sgn(SV[c - 1]) && SV[c] != 0 && SV[c - 1] != 0 && std::abs(SV[c] + SV[c - 1]) < 100 && SVpos[c] - SVpos[c - 1] < 3)
          for (int i = SVpos[c - 1]; i < SVpos[c]; i++) if (ch[i].len < 100) remove[i] = true;
      }
      if (refineEnds) {
        for (int c = 1; c < N; c++) {
          long tD, qD; dists(c, tD, qD);
          int dist = (int)std::min(tD, qD);
          if (dist < meanDist + 4 * sdDist) { if (firstValid == -1) firstValid = c - 1; lastValid = c; }
        }
        if (lastValid == -1 || firstValid == -1) for (int i = 0; i < N; i++) if (ch[i].len < 100) remove[i] = true;
        if (firstValid > 0 && firstValid < 3) for (int i = 0; i < firstValid; i++) if (ch[i].len < 100) remove[i] = true;
        if (lastValid + 1 <= N && N - lastValid < 3) for (int i = lastValid + 1; i < N; i++) if (ch[i].len < 100) remove[i] = true;
      }
    } else if (op == 5) {                                                  // RemovePairedIndels(GenomePairs&, chain, lengths)  Chain.h:753-811
      touchLink = false;
      std::vector<int> SVgenome;
      for (int c = 1; c < N; c++) {
        int Gap = (int)(((long)tS(c) - (long)qS(c)) - ((long)tS(c - 1) - (long)qS(c - 1)));
        if (std::abs(Gap) > 30) { SV.push_back(Gap); SVgenome.push_back((int)tS(c)); SVpos.push_back(c); }
      }
      auto sg = [](int x) { return x >= 0; };
      for (size_t c = 1; c < SV.size(); c++) {
        int blink = std::max(std::abs(SV[c]), std::abs(SV[c - 1]));
        if (sg(SV[c]) != sg(SV[c - 1]) && std::abs(SV[c] + SV[c - 1]) < 600 && std::abs(SV[c]) != 0 && SV[c - 1] != 0) {
          if ((sg(SV[c]) == true && std::abs(SVgenome[c] - SVgenome[c - 1]) < std::max(2 * blink, 1000)) ||
              (sg(SV[c]) == false && std::abs(SVgenome[c] - SV[c] - SVgenome[c - 1]) < std::max(2 * blink, 1000)))
            for (int i = SVpos[c - 1]; i < SVpos[c]; i++) if (ch[i].len < 100) remove[i] = true;
        } else if (sg(SV[c]) != sg(SV[c - 1]) && SV[c] != 0 && SV[c - 1] != 0 &&
                   ((sg(SV[c]) == true && std::abs(SVgenome[c] - SVgenome[c - 1]) < 500) || (sg(SV[c]) == false && std::abs(SVgenome[c] - SV[c] - SVgenome[c - 1]) < 500))) {
          for (int i = SVpos[c - 1]; i < SVpos[c]; i++) if (ch[i].len < 100) remove[i] = true;
        } else if (sg(SV[c]) == sg(SV[c - 1]) && SV[c] != 0 && SV[c - 1] != 0) {
          if ((sg(SV[c]) == true && std::abs(SVgenome[c] - SVgenome[c - 1]) < std::max(2 * blink, 1000)) ||
              (sg(SV[c]) == false && std::abs(SVgenome[c] - SV[c] - SVgenome[c - 1]) < std::max(2 * blink, 1000)))
            for (int i = SVpos[c - 1]; i < SVpos[c]; i++) if (ch[i].len < 100) remove[i] = true;
        }
      }
    } else if (op == 4) {
      touchLink = false;
      for (int c = 1; c < N; c++) {
        if (ch[c].strand == ch[c - 1].strand) { int Gap = gap_of(c); if (std::abs(Gap) >= 500) { SV.push_back(Gap); SVpos.push_back(c); } }
        else { SVpos.push_back(c); SV.push_back(0); }
      }
      for (size_t c = 1; c < SV.size(); c++)
        if (SV[c] != 0 && SV[c - 1] != 0 && SVpos[c] - SVpos[c - 1] <= 10) {
          bool check = false;
          for (int b = SVpos[c - 1]; b < SVpos[c]; b++) if (ch[b].len >= 50) { check = true; break; }
          if (!check) for (int i = SVpos[c - 1]; i < SVpos[c]; i++) if (ch[i].len < 50) remove[i] = true;
        }
    }
    int m = 0;
    for (int i = 0; i < N; i++)
      if (!remove[i]) {
        ch[m] = ch[i];
        if (touchLink && !lk.empty() && m >= 1) lk[m - 1] = lk[i - 1];
        m++;
      }
    ch.resize(m);
    if (touchLink && !lk.empty()) { if (op == 2 || op == 3) { if (m > 0) lk.resize(m - 1); } else lk.resize(m - 1); }
  }
  for (int i = 0; i < n; i++) keep[i] = 0;
  for (auto& a : ch) keep[a.orig] = 1;
  *nLink = (int)lk.size();
  for (size_t i = 0; i < lk.size(); i++) linkOut[i] = lk[i];
  return (int)ch.size();
}

// ---- the high-accuracy overload  SPLITChain(read, vector<Cluster_SameDiag*>& ExtendClusters, splitchains, link, opts)
// (Mapping_ultility.h:266-346, called at Map_highacc.h:706) + MergeSplitchainINS (:172-262, bypassClustering == 0 on this path) +
// LargestSplitChain_dist (Chain.h:974-985, Map_highacc.h:707).  Element v of the chain is a Cluster_SameDiag: strand, chromIndex, box
// (qStart, qEnd, tStart, tEnd; t chromosome-relative); link = Primary_chains[p].chains[h].link.  OverlaprateOnGenome: Clustering.h:397-406.
// The reference never sets chromIndex of the LAST piece (SplitChain's constructors leave it indeterminate, Chain.h:350-360; pieces pushed in the
// loop get ExtendClusters[cur]->chromIndex, :291/:299/:308): it is restated as -1, a chromosome no other piece has.  PARITY UNPINNED.
// Out: pieces as CSR over v (spOff, spIdx) in the pieces' final order, per piece type / Strand / box(QStart,QEnd,TStart,TEnd), *lsc.  Returns the piece count.
extern "C" int oracle_split_chain_highacc(int n, const uint8_t* strand, const int* chrom, const uint32_t* box, const uint8_t* link, int splitdist, int* spOff,
                                          int* spIdx, char* spType, uint8_t* spStrand, uint32_t* spBox, int* lsc) {
  struct P { std::vector<int> sptc; uint32_t QS = 0, QE = 0, TS = 0, TE = 0; int chromIndex = -1; bool Strand = 0; char type = 'N'; };
  std::vector<P> sp;
  auto qS = [&](int i) { return box[4 * i]; }; auto qE = [&](int i) { return box[4 * i + 1]; };
  auto tS = [&](int i) { return box[4 * i + 2]; }; auto tE = [&](int i) { return box[4 * i + 3]; };
  auto ovl = [&](int a, int b) -> float {                                 // a->OverlaprateOnGenome(b)
    if (tE(a) <= tS(b) || tE(b) <= tS(a)) return 0;
    const int ovp = (int)(std::min(tE(a), tE(b)) - std::max(tS(a), tS(b)));
    const float denomA = (float)(tE(a) - tS(a));
    return ovp / denomA;
  };
  *lsc = 0; spOff[0] = 0;
  if (n == 0) return 0;
  std::vector<int> onec{0};
  int im = 0, cur = 0, prev = 0;
  auto cut = [&](char type) {
    P p; p.sptc = onec; p.chromIndex = chrom[cur]; p.type = type; p.Strand = strand[prev];
    sp.push_back(p); onec.clear(); onec.push_back(cur);
  };
  while (im < n - 1) {
    cur = im + 1; prev = im;
    bool rep_map = 0;
    if (((link[im] == 1 && strand[cur] == 0 && strand[prev] == 0) || (link[im] == 0 && strand[cur] == 1 && strand[prev] == 1)) && ovl(prev, cur) >= 0.6 &&
        ovl(cur, prev) >= 0.6)
      rep_map = 1;
    if (tS(cur) > tE(prev) + (uint32_t)splitdist || tE(cur) + (uint32_t)splitdist < tS(prev) || chrom[cur] != chrom[prev]) cut('T');
    else if (rep_map) cut('D');
    else if (strand[cur] != strand[prev]) cut('I');
    else onec.push_back(cur);
    im++;
  }
  if (!onec.empty()) { P p; p.sptc = onec; p.type = 'N'; p.Strand = strand[n - 1]; sp.push_back(p); }
  for (auto& p : sp) {
    p.QS = qS(p.sptc[0]); p.QE = qE(p.sptc[0]); p.TS = tS(p.sptc[0]); p.TE = tE(p.sptc[0]);
    for (size_t k = 1; k < p.sptc.size(); k++) {
      p.QS = std::min(p.QS, qS(p.sptc[k])); p.QE = std::max(p.QE, qE(p.sptc[k])); p.TS = std::min(p.TS, tS(p.sptc[k])); p.TE = std::max(p.TE, tE(p.sptc[k]));
    }
  }
  if (sp.size() >= 3) {                                                    // MergeSplitchainINS :172-262
    std::vector<int> cur_ind(sp.size());
    std::iota(cur_ind.begin(), cur_ind.end(), 0);
    std::vector<bool> keepS(sp.size(), true);
    bool change = false;
    size_t i0 = 0;
    while (i0 + 3 <= sp.size()) {
      const int c = cur_ind[i0];
      if (sp[c].type != 'T') { i0++; continue; }
      size_t nn = (size_t)cur_ind[i0 + 2];
      while (nn < sp.size()) {
        const long tdist = (sp[c].TS > sp[nn].TE) ? ((long)sp[c].TS - (long)sp[nn].TE) : ((long)sp[nn].TE - (long)sp[c].TS);
        if (tdist > 1500 || sp[c].Strand != sp[nn].Strand || sp[c].chromIndex != sp[nn].chromIndex) { nn++; continue; }
        change = true;
        sp[c].sptc.insert(sp[c].sptc.end(), sp[nn].sptc.begin(), sp[nn].sptc.end());
        sp[c].QS = std::min(sp[c].QS, sp[nn].QS); sp[c].TS = std::min(sp[c].TS, sp[nn].TS);
        sp[c].QE = std::max(sp[c].QE, sp[nn].QE); sp[c].TE = std::max(sp[c].TE, sp[nn].TE);
        sp[c].type = sp[nn].type;
        cur_ind[nn] = cur_ind[c];
        keepS[nn] = false;
        break;
      }
      i0 = nn;
    }
    if (change) {
      size_t r = 0;
      for (size_t s = 0; s < sp.size(); s++) if (keepS[s]) { if (r != s) sp[r] = sp[s]; r++; }
      sp.resize(r);
    }
  }
  int o = 0;
  for (size_t k = 0; k < sp.size(); k++) {
    for (int v : sp[k].sptc) spIdx[o++] = v;
    spOff[k + 1] = o; spType[k] = sp[k].type; spStrand[k] = sp[k].Strand;
    spBox[4 * k] = sp[k].QS; spBox[4 * k + 1] = sp[k].QE; spBox[4 * k + 2] = sp[k].TS; spBox[4 * k + 3] = sp[k].TE;
  }
  int maxi = 0, maxi_d = sp[0].QE > sp[0].QS ? (int)(sp[0].QE - sp[0].QS) : 0;    // LargestSplitChain_dist
  for (size_t mi = 1; mi < sp.size(); mi++) {
    const int d = sp[mi].QE > sp[mi].QS ? (int)(sp[mi].QE - sp[mi].QS) : 0;
    if (d > maxi_d) { maxi = (int)mi; maxi_d = d; }
  }
  *lsc = maxi;
  return (int)sp.size();
}
