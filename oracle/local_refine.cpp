// oracle/local_refine.cpp -- TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// CPU restatement of the chain walk that turns the chains of the second sparse DP into alignments (low-accuracy path, Map_lowacc.h:575-576):
//   LocalRefineAlignment(ultimatechains, ext_clusters, alignments, smallOpts, ..., h, genome, LSC, tinyOpts, ...)   LocalRefineAlignment.h:885-1029
//   RefinedAlignmentbtwnAnchors                                                                                      LocalRefineAlignment.h:203-550
//   RefineByLinearAlignment :141-185 (oracle_between_anchors), RefineSpace (oracle_refine_space), LinearExtend pair version
//   (oracle_linear_extend after DiagonalSort), TrimOverlappedAnchors pair version (oracle_trim_anchor_pairs), SparseDP_ForwardOnly
//   (oracle_sdp_chain, single-cluster mode on a forward cluster), RemovePairedIndels pair version (oracle_filter_chain op 5)
// Parity status: PARITY UNPINNED -- LocalRefineAlignment.h needs Genome.h / Clustering.h (htslib); restated from the source text.
// The +,-,+ / typeofaln = 3 pass at the end of the function (:997-1023) reads nm, tStart and tEnd of the new alignments, which are
// still the constructor's zeros there (CalculateStatistics runs later, Map_lowacc.h:596), so it never marks anything; restated as such.
#include "oracle_common.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

extern "C" long oracle_refine_space(const char* q, int qLen, const char* t, int tLen, uint32_t tSpan, int K, int W, int refineSpaceDiag, int match,
                                    int mismatch, int indel, long maxFreq, uint32_t qAdd, uint32_t tAdd, uint32_t flipLen, uint32_t* outQ,
                                    uint32_t* outT, long cap, float* identity);
extern "C" long oracle_linear_extend(const uint32_t* q, const uint32_t* t, long n, int strand, int K, const char* read, uint32_t readLen,
                                     const char* chrom, uint32_t chromLen, uint32_t* eq, uint32_t* et, int* elen, uint32_t* box);
extern "C" void oracle_trim_anchor_pairs(int n, const uint32_t* Q, const uint32_t* T, int* L);
extern "C" int oracle_filter_chain(int n, const uint32_t* q, const uint32_t* t, const int* len, const uint8_t* strand, const uint8_t* link, int hasLink,
                                   const int* ops, int nOps, uint8_t* keep, uint8_t* linkOut, int* nLink);
extern "C" int oracle_between_anchors(const char* q, const char* t, uint32_t curReadEnd, uint32_t nextReadStart, uint32_t curGenomeEnd,
                                      uint32_t nextGenomeStart, int match, int mismatch, int indel, int localBand, int refineDp, int* blocks, int cap,
                                      int* score);
struct oracle_sdp_opts { float rate; int NumAln; float alnthres; int readLen; float gapopen, gapextend, gaproot; int gapCeiling1, gapCeiling2; int mode; int globalK; };
extern "C" int oracle_sdp_chain(int nClusters, const int* clusterOff, const uint8_t* clusterStrand, const uint32_t* q, const uint32_t* t,
                                const int* len, const oracle_sdp_opts* o, float* fragVal, long* fragPrevSub, long* fragPrevInd,
                                uint8_t* fragFlags, int maxChains, int* chainOff, uint32_t* chainFrag, uint8_t* chainLink, uint32_t* chainBox,
                                float* chainValue);

struct oracle_lra_opts {
  int localW, globalW, localMaxFreq;     // tinyOpts.localW / globalW / localMaxFreq on entry (Map_lowacc.h:241-242: globalW = localW)
  int match, mismatch, indel, localBand; // localMatch / localMismatch / localIndel / localBand
  int refineBySDP, isOnt;                // Options::RefineBySDP; readType clr / ont (1) or contig / ccs (0)
  float gapopen, gapextend, gaproot; int gapCeiling1, gapCeiling2;   // the PWL table of the sparse DP
};

namespace {

struct Aln { int strand, supp, secondary, N0, N1, chrom; float value; std::vector<int> blocks; };   // blocks: qPos, tPos, length triples

struct Env {
  const char* strands[2]; uint32_t readLen; const char* genome; const uint64_t* chromPos; const oracle_lra_opts* o; bool bad = false;
  const char* chromSeq(int c) const { return genome + chromPos[c]; }
  uint32_t chromLen(int c) const { return (uint32_t)(chromPos[c + 1] - chromPos[c]); }
};

void linear_between(Env& E, uint32_t cre, uint32_t cge, uint32_t nrs, uint32_t ngs, int str, int chrom, Aln* a) {   // RefineByLinearAlignment :141-185
  const int cap = std::max(0, (int)(nrs - cre)) + std::max(0, (int)(ngs - cge)) + 16;
  std::vector<int> b(3 * (size_t)std::max(cap, 16));
  int score = 0;
  const int nb = oracle_between_anchors(E.strands[str], E.chromSeq(chrom), cre, nrs, cge, ngs, E.o->match, E.o->mismatch, E.o->indel, E.o->localBand, 1, b.data(),
                                        std::max(cap, 16), &score);
  if (nb < 0) { E.bad = true; return; }
  a->blocks.insert(a->blocks.end(), b.begin(), b.begin() + 3 * nb);
}

long refine_space(Env& E, int K, int W, int diag, int maxFreq, int chrom, uint32_t qe, uint32_t qs, uint32_t te, uint32_t ts, int st, std::vector<uint32_t>& oq,
                  std::vector<uint32_t>& ot, float& identity) {
  long cap = 4096;
  for (;;) {
    oq.resize(cap); ot.resize(cap);
    const long n = oracle_refine_space(E.strands[st] + qs, (int)(qe - qs), E.chromSeq(chrom) + ts, (int)(te - ts), te - ts, K, W, diag, E.o->match, E.o->mismatch,
                                       E.o->indel, maxFreq, qs, ts, 0, oq.data(), ot.data(), cap, &identity);
    if (n < 0) { E.bad = true; return 0; }
    if (n <= cap) { oq.resize(n); ot.resize(n); return n; }
    cap = n;
  }
}

// RefinedAlignmentbtwnAnchors :203-550.  A = the chain's anchors (q, t, len) in chain order; alns.back() is the alignment being built.
void between_anchors(Env& E, int cur, int next, int str, int inv_str, int chrom, const uint32_t* AQ, const uint32_t* AT, const int* AL, std::vector<Aln>& alns,
                     bool& inversion, bool& breakalignment) {
  Aln* alignment = &alns.back();
  const uint32_t L = E.readLen;
  if (str == 0) { alignment->blocks.push_back((int)AQ[cur]); alignment->blocks.push_back((int)AT[cur]); alignment->blocks.push_back(AL[cur]); }
  else { alignment->blocks.push_back((int)(L - AQ[cur] - AL[cur])); alignment->blocks.push_back((int)AT[cur]); alignment->blocks.push_back(AL[cur]); }
  uint32_t curGenomeEnd, curReadEnd, nextGenomeStart, nextReadStart;
  if (str == 0) { curReadEnd = AQ[cur] + AL[cur]; nextReadStart = AQ[next]; curGenomeEnd = AT[cur] + AL[cur]; nextGenomeStart = AT[next]; }
  else { curReadEnd = L - AQ[cur]; nextReadStart = L - AQ[next] - AL[next]; curGenomeEnd = AT[cur] + AL[cur]; nextGenomeStart = AT[next]; }
  if (!(curGenomeEnd <= nextGenomeStart)) return;
  const long read_dist = (long)nextReadStart - (long)curReadEnd, genome_dist = (long)nextGenomeStart - (long)curGenomeEnd;   // (GenomePos differences widened to long)
  // note: nextReadStart - curReadEnd is unsigned arithmetic in the reference, converted to long: a "negative" distance is a huge value
  const long rd = (long)(uint32_t)(nextReadStart - curReadEnd), gd = (long)(uint32_t)(nextGenomeStart - curGenomeEnd);
  (void)read_dist; (void)genome_dist;
  if (E.o->refineBySDP && std::min(rd, gd) >= 300) {
    int K, W, maxFreq = E.o->localMaxFreq;
    int refineSpaceDiag = 0;
    const int sv_diag = (int)(std::max(rd, gd) - std::min(rd, gd));
    if (!E.o->isOnt) refineSpaceDiag = std::min((int)std::floor(std::max(80.f, 0.01f * rd)), 500);
    else refineSpaceDiag = std::min((int)std::floor(std::max(100.f, 0.15f * rd)), 2000);
    refineSpaceDiag = std::max(2 * sv_diag, refineSpaceDiag);
    float minRatio;
    if (std::max(rd, gd) < 100) { K = 6; W = 5; minRatio = 0.5 / 29.5; }
    else if (std::max(rd, gd) < 500) { K = 9; W = 7; maxFreq = 50; minRatio = 0.5 / 69.1; }
    else { K = 12; W = 7; minRatio = 0.5 / 140.2; }
    std::vector<uint32_t> fq, ft, rq, rt;
    float identity = 0;
    refine_space(E, K, W, refineSpaceDiag, maxFreq, chrom, nextReadStart, curReadEnd, nextGenomeStart, curGenomeEnd, str, fq, ft, identity);
    if (E.bad) return;
    const int minDist = (int)std::min(rd, gd);
    bool useRev = false;
    if (getenv("ORACLE_DEBUG")) fprintf(stderr, "big space: rd %ld gd %ld K %d nF %zu id %f minRatio %f blocks %zu\n", rd, gd, K, fq.size(), identity, minRatio, alns.back().blocks.size() / 3);
    if ((fq.size() / (float)minDist) < minRatio && alns.back().blocks.size() / 3 >= 5 && identity < 0.8) {   // try the other strand
      const uint32_t temp = curReadEnd;
      curReadEnd = L - nextReadStart; nextReadStart = L - temp;
      identity = 0;
      refine_space(E, K, E.o->globalW, refineSpaceDiag, maxFreq, chrom, nextReadStart, curReadEnd, nextGenomeStart, curGenomeEnd, inv_str, rq, rt, identity);
      if (E.bad) return;
      const double driftRate = E.o->isOnt ? 0.10f : 0.01f;
      if (fq.size() == 0 && rq.size() == 0 && minDist > 500 && sv_diag <= std::max((double)50, minDist * driftRate)) { breakalignment = 1; inversion = 0; return; }
      if (identity < 0.8 && rq.size() / (float)minDist < minRatio) { breakalignment = 1; inversion = 0; return; }
      if (fq.size() >= rq.size()) {
        inversion = 0;
        const uint32_t t2 = curReadEnd;
        curReadEnd = L - nextReadStart; nextReadStart = L - t2;
      } else { useRev = true; inversion = 1; }
    }
    std::vector<uint32_t>& bq = useRev ? rq : fq;
    std::vector<uint32_t>& bt = useRev ? rt : ft;
    if (bq.size() > 0) {
      // LinearExtend(BtwnPairs, ..., chromIndex, strand 0, skipsorting 0, K) on read.seq (the forward read, whatever str)
      const size_t n = bq.size();
      std::vector<std::pair<uint32_t, uint32_t>> P(n);
      for (size_t i = 0; i < n; i++) P[i] = std::make_pair(bq[i], bt[i]);
      std::sort(P.begin(), P.end(), [](const std::pair<uint32_t, uint32_t>& a, const std::pair<uint32_t, uint32_t>& b) {
        const long aD = (long)a.first - (long)a.second, bD = (long)b.first - (long)b.second;
        if (aD != bD) return aD < bD;
        return a.first < b.first;
      });
      std::vector<uint32_t> sq(n), st(n), eq(n + 2), et(n + 2); std::vector<int> el(n + 2);
      for (size_t i = 0; i < n; i++) { sq[i] = P[i].first; st[i] = P[i].second; }
      uint32_t bx[4];
      long ne = oracle_linear_extend(sq.data(), st.data(), (long)n, 0, K, E.strands[0], L, E.chromSeq(chrom), E.chromLen(chrom), eq.data(), et.data(), el.data(), bx);
      if (inversion == 0) {
        eq[ne] = nextReadStart; et[ne] = nextGenomeStart; el[ne] = AL[next]; ne++;
        eq[ne] = curReadEnd - AL[cur]; et[ne] = curGenomeEnd - AL[cur]; el[ne] = AL[cur]; ne++;
      }
      oracle_trim_anchor_pairs((int)ne, eq.data(), et.data(), el.data());
      // SparseDP_ForwardOnly(..., rate 2) = the single-cluster sparse DP on one forward cluster; inv_value = its best value
      oracle_sdp_opts so; so.rate = 2.0f; so.NumAln = 1; so.alnthres = 0; so.readLen = (int)L; so.gapopen = E.o->gapopen; so.gapextend = E.o->gapextend;
      so.gaproot = E.o->gaproot; so.gapCeiling1 = E.o->gapCeiling1; so.gapCeiling2 = E.o->gapCeiling2; so.mode = 1; so.globalK = K;
      const int off2[2] = {0, (int)ne}; const uint8_t cst[1] = {0};
      std::vector<float> fv(ne); std::vector<int> coff(2); std::vector<uint32_t> cf(ne); std::vector<uint8_t> cl(ne); uint32_t cb[4]; float cv[1] = {0};
      const int nc = oracle_sdp_chain(1, off2, cst, eq.data(), et.data(), el.data(), &so, fv.data(), nullptr, nullptr, nullptr, 1, coff.data(), cf.data(), cl.data(), cb, cv);
      if (nc < 0) { E.bad = true; return; }
      std::vector<uint32_t> chain(cf.begin(), cf.begin() + (nc > 0 ? coff[1] : 0));
      const float inv_value = cv[0];
      {                                                                   // RemovePairedIndels(ExtendBtwnPairs, BtwnChain, lengths)  Chain.h:753
        const int m = (int)chain.size();
        std::vector<uint32_t> cq(m), ct(m); std::vector<int> cln(m); std::vector<uint8_t> cs(m, 0), keep(std::max(m, 1)), lo(std::max(m, 1));
        for (int i = 0; i < m; i++) { cq[i] = eq[chain[i]]; ct[i] = et[chain[i]]; cln[i] = el[chain[i]]; }
        const int ops[1] = {5}; int nl = 0;
        oracle_filter_chain(m, cq.data(), ct.data(), cln.data(), cs.data(), nullptr, 0, ops, 1, keep.data(), lo.data(), &nl);
        std::vector<uint32_t> c2;
        for (int i = 0; i < m; i++) if (keep[i]) c2.push_back(chain[i]);
        chain.swap(c2);
      }
      if (chain.empty()) { E.bad = true; return; }                        // BtwnChain.back() on an empty vector
      uint32_t btc_curReadEnd = curReadEnd, btc_curGenomeEnd = curGenomeEnd;
      int btc_end = (int)chain.size() - 1, btc_start = 0;
      if (chain.back() == (uint32_t)(ne - 1)) btc_end = (int)chain.size() - 2;
      if (chain[0] == (uint32_t)(ne - 2)) btc_start = 1;
      if (inversion == 1) {
        Aln inv; inv.strand = inv_str; inv.supp = 1; inv.secondary = 0; inv.N0 = (int)chain.size(); inv.N1 = (int)chain.size(); inv.chrom = chrom; inv.value = inv_value;
        alns.push_back(inv);
      }
      for (int btc = btc_end; btc >= btc_start; btc--) {
        const uint32_t ngs = et[chain[btc]], nrs = eq[chain[btc]];
        linear_between(E, btc_curReadEnd, btc_curGenomeEnd, nrs, ngs, str, chrom, &alns.back());
        if (E.bad) return;
        alns.back().blocks.push_back((int)nrs); alns.back().blocks.push_back((int)ngs); alns.back().blocks.push_back(el[chain[btc]]);
        btc_curReadEnd = nrs + el[chain[btc]]; btc_curGenomeEnd = ngs + el[chain[btc]];
      }
      if (nextGenomeStart > btc_curGenomeEnd && nextReadStart > btc_curReadEnd) linear_between(E, btc_curReadEnd, btc_curGenomeEnd, nextReadStart, nextGenomeStart, str, chrom, &alns.back());
    } else linear_between(E, curReadEnd, curGenomeEnd, nextReadStart, nextGenomeStart, str, chrom, &alns.back());
  } else linear_between(E, curReadEnd, curGenomeEnd, nextReadStart, nextGenomeStart, str, chrom, &alns.back());
}

}  // namespace

// One primary chain h of one read: nChains chains (the second sparse DP's, after its filters) as CSR over anchors (q, t chromosome-relative,
// len), per chain strand / chromIndex / FirstSDPValue / NumOfAnchors0 / NumOfAnchors1; LSC = LargestSplitChain.  Out: the SegAlignments pushed
// onto alignments.back() in order: strand, Supplymentary, ISsecondary, NumOfAnchors0/1, value, chromIndex and blocks (CSR of qPos, tPos,
// length).  Returns the number of alignments, -1 when the reference would read outside an array, -2 when a capacity is too small.
extern "C" int oracle_local_refine_alignment_ex(int nChains, const int* chainOff, const uint32_t* aq, const uint32_t* at, const int* alen, const uint8_t* chainStrand,
                                                const int* chainChrom, const float* firstSdp, const int* numAnchors0, const int* numAnchors1, int LSC, int h, int minAnchors,
                                                const char* fwd, const char* rc, uint32_t readLen, const char* genome, const uint64_t* chromPos, const oracle_lra_opts* o,
                                                int maxSeg, int* segStrand, int* segSupp, int* segSecondary, int* segN0, int* segN1, float* segValue, int* segChrom,
                                                int* segBlockOff, int* blocks, long blockCap);
extern "C" int oracle_local_refine_alignment(int nChains, const int* chainOff, const uint32_t* aq, const uint32_t* at, const int* alen, const uint8_t* chainStrand,
                                             const int* chainChrom, const float* firstSdp, const int* numAnchors0, const int* numAnchors1, int LSC, int h,
                                             const char* fwd, const char* rc, uint32_t readLen, const char* genome, const uint64_t* chromPos, const oracle_lra_opts* o,
                                             int maxSeg, int* segStrand, int* segSupp, int* segSecondary, int* segN0, int* segN1, float* segValue, int* segChrom,
                                             int* segBlockOff, int* blocks, long blockCap) {
  return oracle_local_refine_alignment_ex(nChains, chainOff, aq, at, alen, chainStrand, chainChrom, firstSdp, numAnchors0, numAnchors1, LSC, h, 2, fwd, rc, readLen, genome,
                                          chromPos, o, maxSeg, segStrand, segSupp, segSecondary, segN0, segN1, segValue, segChrom, segBlockOff, blocks, blockCap);
}
// minAnchors = 1: the walk of the high-accuracy overload (LocalRefineAlignment.h:577-766; `if (ultimatechain.size() == 0) continue`, :579), whose chains st are
// splitchains[st] (empty where the ultimatechain is empty), LSC = LargestSplitChain_dist, firstSdp / numAnchors0 = chains[h].value / NumOfAnchors0.
extern "C" int oracle_local_refine_alignment_ex(int nChains, const int* chainOff, const uint32_t* aq, const uint32_t* at, const int* alen, const uint8_t* chainStrand,
                                                const int* chainChrom, const float* firstSdp, const int* numAnchors0, const int* numAnchors1, int LSC, int h, int minAnchors,
                                                const char* fwd, const char* rc, uint32_t readLen, const char* genome, const uint64_t* chromPos, const oracle_lra_opts* o,
                                                int maxSeg, int* segStrand, int* segSupp, int* segSecondary, int* segN0, int* segN1, float* segValue, int* segChrom,
                                                int* segBlockOff, int* blocks, long blockCap) {
  Env E; E.strands[0] = fwd; E.strands[1] = rc; E.readLen = readLen; E.genome = genome; E.chromPos = chromPos; E.o = o;
  std::vector<Aln> alns;
  for (int st = 0; st < nChains; st++) {
    const int m = chainOff[st + 1] - chainOff[st];
    if (m < minAnchors) continue;
    const uint32_t* AQ = aq + chainOff[st]; const uint32_t* AT = at + chainOff[st]; const int* AL = alen + chainOff[st];
    const int start = 0, end = m - 1;
    const int str = chainStrand[st], chrom = chainChrom[st];
    auto fresh = [&](int supp) { Aln a; a.strand = str; a.supp = supp; a.secondary = 0; a.N0 = numAnchors0[st]; a.N1 = 0; a.chrom = chrom; a.value = firstSdp[st]; return a; };
    Aln first = fresh(0);
    first.N1 = numAnchors1[st];
    if (h > 0) first.secondary = 1;
    if (st != LSC) first.supp = 1;
    alns.push_back(first);
    bool inversion = 0, breakalignment = 0;
    auto on_event = [&](int inv_str, int n1) {
      alns.back().strand = inversion ? inv_str : str;                     // UpdateParameters(inv_str / str, ...) :506-511
      alns.back().N0 = numAnchors0[st]; alns.back().N1 = n1;
      alns.push_back(fresh(1));
      inversion = 0; breakalignment = 0;
    };
    if (str == 0) {
      int last = end, fl = end;
      const int inv_str = 1;
      while (fl > start) {
        between_anchors(E, fl, fl - 1, str, inv_str, chrom, AQ, AT, AL, alns, inversion, breakalignment);
        if (E.bad) return -1;
        if (inversion || breakalignment) { on_event(inv_str, last - fl); last = fl; }
        fl--;
      }
      alns.back().N0 = numAnchors0[st]; alns.back().N1 = last - fl;
      alns.back().blocks.push_back((int)AQ[start]); alns.back().blocks.push_back((int)AT[start]); alns.back().blocks.push_back(AL[start]);
    } else {
      int last = start, fl = start;
      const int inv_str = 0;
      while (fl < end) {
        between_anchors(E, fl, fl + 1, str, inv_str, chrom, AQ, AT, AL, alns, inversion, breakalignment);
        if (E.bad) return -1;
        if (inversion || breakalignment) { on_event(inv_str, fl - last); last = fl; }
        fl++;
      }
      alns.back().N0 = numAnchors0[st]; alns.back().N1 = fl - last;
      alns.back().blocks.push_back((int)(readLen - AQ[end] - AL[end])); alns.back().blocks.push_back((int)AT[end]); alns.back().blocks.push_back(AL[end]);
    }
    alns.back().strand = str;                                             // UpdateParameters(str, ...) :995
  }
  if ((int)alns.size() > maxSeg) return -2;
  long nb = 0;
  segBlockOff[0] = 0;
  for (size_t i = 0; i < alns.size(); i++) {
    segStrand[i] = alns[i].strand; segSupp[i] = alns[i].supp; segSecondary[i] = alns[i].secondary; segN0[i] = alns[i].N0; segN1[i] = alns[i].N1;
    segValue[i] = alns[i].value; segChrom[i] = alns[i].chrom;
    if (nb + (long)alns[i].blocks.size() / 3 > blockCap) return -2;
    for (size_t k = 0; k < alns[i].blocks.size(); k++) blocks[3 * nb + k] = alns[i].blocks[k];
    nb += (long)alns[i].blocks.size() / 3;
    segBlockOff[i + 1] = (int)nb;
  }
  return (int)alns.size();
}
