// oracle/refine_btwn.cpp -- TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// CPU restatement of the gap seeding between / beyond the refined clusters of one chain (low-accuracy path, Map_lowacc.h:362):
//   Refine_Btwnsplitchain                     ChainRefine.h:579-754
//   RefineBtwnSpace_AppendCloseCluster        ChainRefine.h:59-121   (append_to_closetcluster :22-56, minGapDifference Clustering.h:532)
//   RefineBtwnSpace                           ClusterRefine.h:327-430
//   RefineSpace                               ClusterRefine.h:242-325 (oracle_refine_space)
//   Cluster::SetClusterBoundariesFromMatches  Clustering.h:308-322
// Parity status: PARITY UNPINNED -- ChainRefine.h / ClusterRefine.h need Genome.h (htslib); restated from the source text.
// Only the -ONT / -CLR read types are restated: for the others RefineBtwnSpace_AppendCloseCluster leaves refineSpaceDiag
// uninitialised (ChainRefine.h:68-71).
#include "oracle_common.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <utility>
#include <vector>

extern "C" long oracle_refine_space(const char* q, int qLen, const char* t, int tLen, uint32_t tSpan, int K, int W, int refineSpaceDiag, int match,
                                    int mismatch, int indel, long maxFreq, uint32_t qAdd, uint32_t tAdd, uint32_t flipLen, uint32_t* outQ,
                                    uint32_t* outT, long cap, float* identity);

struct oracle_btwn_opts {
  int K, W;                 // opts.globalK / globalW of the Options passed (smallOpts: the local index's k and w)
  int refineSpaceDist;      // Options::refineSpaceDist
  float anchorstoosparse;   // Options::anchorstoosparse
  int match, mismatch, indel;   // localMatch / localMismatch / localIndel (RefineSpace's AffineOneGapAlign)
  int maxFreq;              // localMaxFreq (RefineSpace's CompareLists)
};

namespace {

typedef std::pair<uint32_t, uint32_t> Pair;   // (first.pos, second.pos)

struct Clu {
  std::vector<Pair> m;
  uint32_t qStart = 0, qEnd = 0, tStart = 0, tEnd = 0;
  int strand = 0, chrom = 0, refinespace = 0;
  void bounds(int K) {                                                   // Clustering.h:308-322
    qStart = m[0].first; qEnd = qStart + K; tStart = m[0].second; tEnd = tStart + K;
    for (size_t i = 1; i < m.size(); i++) {
      tEnd = std::max(tEnd, m[i].second + (uint32_t)K); tStart = std::min(tStart, m[i].second);
      qEnd = std::max(qEnd, m[i].first + (uint32_t)K); qStart = std::min(qStart, m[i].first);
    }
  }
};

struct Env {
  const char* strands[2]; uint32_t readLen;
  const char* genome; const uint64_t* chromPos;
  const oracle_btwn_opts* o;
  bool bad = false;
};

// RefineSpace(K, W, diag, consider_str = 1, EndPairs, ..., qe, qs, te, ts, st, lrts, lrlength)   ClusterRefine.h:242-325
void refine_space(Env& E, int diag, int chrom, uint32_t qe, uint32_t qs, uint32_t te, uint32_t ts, int st, uint32_t lrts, uint32_t lrlength,
                  std::vector<Pair>& out) {
  const int qLen = (int)(qe - qs), tLen = (int)(te - ts + lrlength);
  const uint32_t tAdd = ts - lrts;
  long cap = 4096;
  std::vector<uint32_t> oq, ot;
  float id;
  for (;;) {
    oq.resize(cap); ot.resize(cap);
    const long n = oracle_refine_space(E.strands[st] + qs, qLen, E.genome + E.chromPos[chrom] + tAdd, tLen, te - tAdd, E.o->K, E.o->W, diag, E.o->match,
                                       E.o->mismatch, E.o->indel, E.o->maxFreq, qs, tAdd, st == 1 ? E.readLen : 0, oq.data(), ot.data(), cap, &id);
    if (n < 0) { E.bad = true; return; }
    if (n <= cap) { out.clear(); for (long i = 0; i < n; i++) out.push_back(Pair(oq[i], ot[i])); return; }
    cap = n;
  }
}

int space_diag(uint32_t qe, uint32_t qs) { return std::min((int)std::floor(std::max(100.f, 0.15f * (qe - qs))), 1000); }   // clr / ont

void append_to_closest(std::vector<Pair>& M, int start, int end, Clu* c, Clu* p, int st, int K) {   // ChainRefine.h:22-56
  uint32_t qStart = M[start].first, qEnd = qStart + K, tStart = M[start].second, tEnd = tStart + K;
  for (int i = start + 1; i < end; i++) {
    tEnd = std::max(tEnd, M[i].second + (uint32_t)K); tStart = std::min(tStart, M[i].second);
    qEnd = std::max(qEnd, M[i].first + (uint32_t)K); qStart = std::min(qStart, M[i].first);
  }
  int qdist = (qStart >= c->qEnd) ? qStart - c->qEnd : 0, tdist;
  if (st == 0) tdist = (tStart >= c->tEnd) ? tStart - c->tEnd : 0;
  else tdist = (c->tStart >= tEnd) ? c->tStart - tEnd : 0;
  const int dist_cur = std::max(qdist, tdist);
  qdist = (p->qStart >= qEnd) ? p->qStart - qEnd : 0;
  if (st == 0) tdist = (p->tStart >= tEnd) ? p->tStart - tEnd : 0;
  else tdist = (tStart >= p->tEnd) ? tStart - p->tEnd : 0;
  const int dist_prev = std::max(qdist, tdist);
  Clu* d = dist_cur <= dist_prev ? c : p;
  d->m.insert(d->m.end(), M.begin() + start, M.begin() + end);
  d->bounds(K);
}

// ChainRefine.h:59-121; returns 1 never (the reverse-cluster branch is commented out in the reference)
void append_close(Env& E, bool twoblocks, Clu* c, Clu* p, uint32_t qe, uint32_t qs, uint32_t te, uint32_t ts, int st) {
  if (st == 1) { const uint32_t t = qs; qs = E.readLen - qe; qe = E.readLen - t; }
  const int diag = space_diag(qe, qs);
  std::vector<Pair> P;
  refine_space(E, diag, c->chrom, qe, qs, te, ts, st, 0, 0, P);
  if (E.bad) return;
  const float eff = ((float)P.size()) / std::min(qe - qs, te - ts);
  if (P.empty()) return;
  if (eff >= E.o->anchorstoosparse * 2) { c->m.insert(c->m.end(), P.begin(), P.end()); c->bounds(E.o->K); c->refinespace = 1; return; }
  if (twoblocks) return;
  std::sort(P.begin(), P.end());                                          // CartesianSort: first.pos, then second.pos
  uint32_t max_pairdist = 0;
  for (size_t e = 1; e < P.size(); e++) max_pairdist = std::max(max_pairdist, P[e].first - (P[e - 1].first + (uint32_t)E.o->K));
  if (max_pairdist <= 100 && eff >= E.o->anchorstoosparse * 2) { append_to_closest(P, 0, (int)P.size(), c, p, st, E.o->K); return; }
  int start = 0, end = 1;
  while (start < (int)P.size()) {
    end = start + 1;
    while (end < (int)P.size() && std::min(std::labs((long)P[end].first - (long)P[end - 1].first), std::labs((long)P[end].second - (long)P[end - 1].second)) <= 200) end++;
    if (end - start >= 4) append_to_closest(P, start, end, c, p, st, E.o->K);
    start = end;
  }
}

// ClusterRefine.h:327-430
void btwn_space(Env& E, int& nRev, bool twoblocks, Clu* c, uint32_t qe, uint32_t qs, uint32_t te, uint32_t ts, int st, uint32_t lrts, uint32_t lrlength) {
  if (st == 1) { const uint32_t t = qs; qs = E.readLen - qe; qe = E.readLen - t; }
  const int diag = space_diag(qe, qs);
  std::vector<Pair> P, R;
  refine_space(E, diag, c->chrom, qe, qs, te, ts, st, lrts, lrlength, P);
  if (E.bad) return;
  const float eff = ((float)P.size()) / std::min(qe - qs, te - ts);
  if ((!P.empty() && twoblocks) || (!P.empty() && eff >= E.o->anchorstoosparse * 2)) {
    c->m.insert(c->m.end(), P.begin(), P.end()); c->bounds(E.o->K); c->refinespace = 1; return;
  }
  if (twoblocks) return;
  const int rst = st == 1 ? 0 : 1;
  const uint32_t t = qs; qs = E.readLen - qe; qe = E.readLen - t;
  refine_space(E, diag, c->chrom, qe, qs, te, ts, rst, lrts, lrlength, R);
  if (E.bad) return;
  const float reff = ((float)R.size()) / std::min(qe - qs, te - ts);
  if (eff >= reff) { c->m.insert(c->m.end(), P.begin(), P.end()); c->bounds(E.o->K); c->refinespace = 1; }   // (P empty: bounds() would read matches[0])
  else nRev++;
}

}  // namespace

// One chain: nsp refined clusters (those of Refine_splitchain, index = split chain index): matches CSR (matchOff, mq, mt), box, strand,
// chromIndex; link[nsp-1] = spchain_link.  fwd / rc = read.seq / readRC (strands[0..1]); genome = all chromosomes back to back,
// chromPos[nChrom+1].  Out: the clusters after the call (matches CSR up to cap, boxes, refinespace flags) and the number of
// reverse clusters pushed (RevBtwnCluster.size()).  Returns total matches, or -1 if the reference would read outside an array.
extern "C" long oracle_refine_btwn_splitchain(int nsp, const int* matchOff, const uint32_t* mq, const uint32_t* mt, const uint32_t* box, const uint8_t* strand,
                                              const int* chrom, const uint8_t* link, const char* fwd, const char* rc, uint32_t readLen, const char* genome,
                                              const uint64_t* chromPos, int nChrom, const oracle_btwn_opts* o, long cap, int* outOff, uint32_t* outQ,
                                              uint32_t* outT, uint32_t* outBox, uint8_t* outRefinespace, int* nRevOut) {
  (void)nChrom;
  std::vector<Clu> C(nsp);
  for (int i = 0; i < nsp; i++) {
    for (int k = matchOff[i]; k < matchOff[i + 1]; k++) C[i].m.push_back(Pair(mq[k], mt[k]));
    C[i].qStart = box[4 * i]; C[i].qEnd = box[4 * i + 1]; C[i].tStart = box[4 * i + 2]; C[i].tEnd = box[4 * i + 3];
    C[i].strand = strand[i]; C[i].chrom = chrom[i];
  }
  Env E; E.strands[0] = fwd; E.strands[1] = rc; E.readLen = readLen; E.genome = genome; E.chromPos = chromPos; E.o = o;
  auto glen = [&](int c) { return (uint32_t)(chromPos[c + 1] - chromPos[c]); };
  int nRev = 0;
  const uint32_t RSD = (uint32_t)o->refineSpaceDist;
  int c = 1;
  bool twoblocks = false;
  int st1 = 0, st2 = 0;
  while (c < nsp) {                                                       // :587-682
    Clu& cur = C[c]; Clu& prev = C[c - 1];
    if (cur.m.empty() || prev.m.empty()) { c++; continue; }
    const uint32_t qs = cur.qEnd, qe = prev.qStart;
    uint32_t ts1 = 0, te1 = 0, ts2 = 0, te2 = 0;
    if (qe <= qs) { c++; continue; }
    if (cur.strand == prev.strand && link[c - 1] == 0) {
      twoblocks = 0; st1 = cur.strand;
      if (cur.tEnd <= prev.tStart) { ts1 = cur.tEnd; te1 = prev.tStart; }
      else if (cur.tStart > prev.tEnd) { ts1 = prev.tEnd; te1 = cur.tStart; }
      else { c++; continue; }
    } else if (cur.strand != prev.strand && link[c - 1] == 1) {
      st1 = cur.strand; st2 = prev.strand; twoblocks = 1;
      if (cur.tEnd <= prev.tStart) {
        if (st1 == 0) { ts1 = cur.tEnd; te1 = ts1 + qe - qs; ts2 = prev.tEnd; te2 = ts2 + qe - qs; }
        else { te1 = cur.tStart; ts1 = (te1 > (qe - qs) ? te1 - (qe - qs) : 0); te2 = prev.tStart; ts2 = (te2 > (qe - qs) ? te2 - (qe - qs) : 0); }
      } else if (cur.tStart > prev.tEnd) {
        if (st1 == 0) { ts1 = cur.tEnd; te1 = ts1 + qe - qs; te2 = cur.tStart; ts2 = (te2 > (qe - qs) ? te2 - (qe - qs) : 0); }
        else { te1 = cur.tStart; ts1 = (te1 > (qe - qs) ? te1 - (qe - qs) : 0); te2 = prev.tStart; ts2 = (te2 > (qe - qs) ? te2 - (qe - qs) : 0); }
      } else { c++; continue; }
    } else if (cur.strand == prev.strand && link[c - 1] == 1) {
      st1 = cur.strand; st2 = st1; twoblocks = 1;
      if (st1 == 0 && cur.tEnd > prev.tStart) { ts1 = cur.tEnd; te1 = ts1 + qe - qs; te2 = prev.tStart; ts2 = (te2 > (qe - qs) ? te2 - (qe - qs) : 0); }
      else if (st1 == 1 && cur.tStart < prev.tEnd) { te1 = cur.tStart; ts1 = (te1 > (qe - qs) ? te1 - (qe - qs) : 0); ts2 = prev.tEnd; te2 = ts2 + (qe - qs); }
      else { c++; continue; }
    }
    // (strands differ with link 0: none of the branches; ts1 = te1 = 0 and twoblocks / st1 keep their previous values)
    if (te1 <= ts1) { c++; continue; }
    if (te1 >= glen(cur.chrom)) { c++; continue; }
    if (std::max(qe - qs, te1 - ts1) >= 5 * RSD) { c++; continue; }
    uint32_t SpaceLength = std::max(qe - qs, te1 - ts1);
    if (SpaceLength >= 20 && SpaceLength <= RSD && cur.chrom == prev.chrom) append_close(E, twoblocks, &cur, &prev, qe, qs, te1, ts1, st1);
    if (E.bad) return -1;
    if (twoblocks) {
      if (te2 <= ts2) { c++; continue; }
      if (te2 >= glen(cur.chrom)) { c++; continue; }
      if (std::max(qe - qs, te2 - ts2) >= 5 * RSD) { c++; continue; }
      SpaceLength = std::max(qe - qs, te2 - ts2);
      if (SpaceLength >= 20 && SpaceLength <= RSD && cur.chrom == prev.chrom) btwn_space(E, nRev, twoblocks, &prev, qe, qs, te2, ts2, st2, 0, 0);
      if (E.bad) return -1;
    }
    c++;
  }
  if (nsp > 0) {
    {                                                                     // :684-724 beyond the first split chain (the read's end)
      Clu& h = C[0];
      if (!h.m.empty()) {
        const int st = h.strand;
        const uint32_t qs = h.qEnd, qe = readLen;
        uint32_t ts = 0, te = 0;
        bool tsSet = true;
        if (st == 0) { ts = h.tEnd; te = ts + qe - qs; }
        else { te = h.tStart; if (te > qe - qs) ts = te - (qe - qs); else { te = 0; tsSet = false; } }
        // (`ts` is a local the reference leaves unset on that last path; te = 0 then fails `te > ts` for every ts)
        (void)tsSet;
        if (qe > qs && te > ts) {
          const uint32_t SpaceLength = std::max(qe - qs, te - ts);
          if (SpaceLength >= 20 && SpaceLength < RSD && te + 500 < glen(h.chrom)) {
            uint32_t lrts = 0, lrlength = 0;
            if (st == 0) { lrts = 0; lrlength = 500; }
            else { if (ts > 500) lrts = 500; lrlength = lrts; }
            btwn_space(E, nRev, 1, &h, qe, qs, te, ts, st, lrts, lrlength);
            if (E.bad) return -1;
          }
        }
      }
    }
    {                                                                     // :725-753 beyond the last split chain (the read's start)
      Clu& h = C[nsp - 1];
      if (!h.m.empty()) {
        const uint32_t qs = 0, qe = h.qStart;
        const int st = h.strand;
        uint32_t ts, te;
        if (st == 0) { te = h.tStart; if (te > qe - qs) ts = te - (qe - qs); else ts = 0; }
        else { ts = h.tEnd; te = ts + (qe - qs); }
        if (qe > qs && te > ts) {
          const uint32_t SpaceLength = std::max(qe - qs, te - ts);
          if (SpaceLength >= 20 && SpaceLength < RSD && te + 500 < glen(h.chrom)) {
            uint32_t lrts = 0, lrlength = 0;
            if (st == 0) { if (ts > 500) lrts = 500; lrlength = lrts; }
            else { lrts = 0; lrlength = 500; }
            btwn_space(E, nRev, 1, &h, qe, qs, te, ts, st, lrts, lrlength);
            if (E.bad) return -1;
          }
        }
      }
    }
  }
  long total = 0;
  outOff[0] = 0;
  for (int i = 0; i < nsp; i++) {
    for (size_t k = 0; k < C[i].m.size(); k++, total++) if (total < cap) { outQ[total] = C[i].m[k].first; outT[total] = C[i].m[k].second; }
    outOff[i + 1] = (int)total;
    outBox[4 * i] = C[i].qStart; outBox[4 * i + 1] = C[i].qEnd; outBox[4 * i + 2] = C[i].tStart; outBox[4 * i + 3] = C[i].tEnd;
    outRefinespace[i] = (uint8_t)C[i].refinespace;
  }
  *nRevOut = nRev;
  return total;
}
