// oracle/pipeline.cpp -- TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// MapRead -> MapRead_lowacc for ONE read composed from the oracle's stage functions (MapRead.h:153-263, Map_lowacc.h:69-632), statement by statement
// in the reference's order, and a thread pool over reads for bench.py's cpu_baseline leg (the reference's `-t N` scheduler, lra.cpp:678-714:
// N worker threads pulling reads; reads are independent).  It is what tests/test_mapread.py compares the C boundary of the product with, and
// what the CPU baseline times: the same stages in the same order as the GPU step.  The product never links this file.
//
// Statement map (Map_lowacc.h line -> here):
//   :77-78  CleanMatches fwd / rev            clean_matches x2            :86-89  repetitivecluster                   `repetitive`
//   :81-85  no cluster -> unaligned           early return               :118-137 chromosome offsets, LinearExtend    linear_extend per cluster
//   :184-188 match_rate, SparseDP (SDP#A)     sdp_chain mode 0            :189-192 RemoveSpuriousJump                  inside split_chain (op 8 first)
//   :194-198 no chain -> unaligned            early return               :246-250 read LocalIndex fwd / rc            local_index_seq, lazily per strand
//   :261-267 SPLITChain, RemoveSpurious..., empty -> unaligned (p = 0) / break (p > 0)
//   :298    Refine_splitchain                 refine_splitchain per split :371    Refine_Btwnsplitchain               refine_btwn_splitchain
//   :440-476 MergeChain, LinearExtend, DecideCoordinates, TrimOverlappedAnchors        merge_extend
//   :486-491 SizeRefinedClusters == 0 -> unaligned (p = 0) / break (p > 0)
//   :529-541 SparseDP per merged cluster + RemovePairedIndels + RemoveSpuriousAnchors   sdp_chain mode 1, filter_chain {2, 4}
//   :574-576 alignments.resize, LargestSplitChain, LocalRefineAlignment                  local_refine_alignment
//   :577-580 p == 0 without SegAlignment -> unaligned, break
//   :582    IndelRefineAlignment              indel_refine                :585-595 RefineBreakpoint (opts.refineBreakpoint)
//   :597-599 CalculateStatistics              calculate_statistics        :600-618 SetFromSegAlignment .. OUTPUT: the product's host tail (pinned emitters)
#include "oracle_common.h"
#include "../include/lra_hip.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <string.h>
#include <string>
#include <thread>
#include <vector>

struct oracle_sdp_opts { float rate; int NumAln; float alnthres; int readLen; float gapopen, gapextend, gaproot; int gapCeiling1, gapCeiling2; int mode; int globalK; };
struct oracle_lra_opts { int localW, globalW, localMaxFreq; int match, mismatch, indel, localBand; int refineBySDP, isOnt; float gapopen, gapextend, gaproot; int gapCeiling1, gapCeiling2; };
struct oracle_btwn_opts { int K, W; int refineSpaceDist; float anchorstoosparse; int match, mismatch, indel; int maxFreq; };
struct oracle_rsc_opts { int window, smallK, K, limitrefine, maxFreq; };

extern "C" {
long oracle_store_minimizers(const char* seq, uint32_t seqLen, int k, int w, uint64_t* keys, uint32_t* poss, long cap);
void oracle_sort_minimizers(uint64_t* keys, uint32_t* poss, long n);
long oracle_compare_lists(const uint64_t* qk, const uint32_t* qp, long nq, const uint64_t* tk, const uint32_t* tp, long nt, long maxFreq, int64_t maxDiag, int64_t minDiag,
                          uint32_t* out_qi, uint32_t* out_ti, long cap);
long oracle_separate_strand(const char* read, const char* genome, int k, const uint32_t* qpos, const uint32_t* tpos, long n, uint8_t* strand);
long oracle_clean_matches(const uint32_t* qpos, const uint32_t* tpos, const uint64_t* qkey, long n, int strand, const lra_clean_opts* o, const uint64_t* chrom_pos, int n_chrom,
                          uint32_t* out_q, uint32_t* out_t, long* n_clean, long* c_start, long* c_end, uint32_t* c_qs, uint32_t* c_qe, uint32_t* c_ts, uint32_t* c_te, int* c_chrom,
                          float* c_freq);
long oracle_linear_extend(const uint32_t* q, const uint32_t* t, long n, int strand, int K, const char* read, uint32_t readLen, const char* chrom, uint32_t chromLen, uint32_t* eq,
                          uint32_t* et, int* elen, uint32_t* box);
int oracle_sdp_chain(int nClusters, const int* clusterOff, const uint8_t* clusterStrand, const uint32_t* q, const uint32_t* t, const int* len, const oracle_sdp_opts* o, float* fragVal,
                     long* fragPrevSub, long* fragPrevInd, uint8_t* fragFlags, int maxChains, int* chainOff, uint32_t* chainFrag, uint8_t* chainLink, uint32_t* chainBox, float* chainValue);
int oracle_split_chain(int n, const uint32_t* q, const uint32_t* t, const int* len, const uint8_t* strand, const int* cluster, const uint8_t* link, const uint64_t* pos, int npos,
                       int splitdist, int bypass, uint8_t* keep, int* nKept, uint8_t* linkOut, int* spOff, int* spIdx, uint8_t* spLink, char* spType, uint8_t* spStrand, int* spChrom,
                       uint32_t* spBox, int* ciOff, int* ciIdx, uint8_t* splitLink, int* nSplitLink);
int oracle_filter_chain(int n, const uint32_t* q, const uint32_t* t, const int* len, const uint8_t* strand, const uint8_t* link, int hasLink, const int* ops, int nOps, uint8_t* keep,
                        uint8_t* linkOut, int* nLink);
long oracle_local_index_seq(const char* seq, long seqLen, int k, int w, int window, int maxFreq, uint32_t* tuples, long cap, uint64_t* boundaries);
long oracle_refine_splitchain(int n, const uint32_t* q, const uint32_t* t, const int* len, const int* cluster, const uint8_t* cstrand, int m, const int* sptc, const uint32_t* box,
                              int Strand, int chromIndex, int nci, const int* ci, const uint64_t* chromPos, int nChrom, uint32_t readLen, long nWq, const uint64_t* qSeqOff,
                              const uint64_t* qBnd, const uint32_t* qTup, long nWg, const uint64_t* gSeqOff, const uint64_t* gBnd, const uint32_t* gTup, const oracle_rsc_opts* o,
                              long cap, uint32_t* outQ, uint32_t* outT, uint32_t* outBox, float* outEff);
long oracle_refine_btwn_splitchain(int nsp, const int* matchOff, const uint32_t* mq, const uint32_t* mt, const uint32_t* box, const uint8_t* strand, const int* chrom,
                                   const uint8_t* link, const char* fwd, const char* rc, uint32_t readLen, const char* genome, const uint64_t* chromPos, int nChrom,
                                   const oracle_btwn_opts* o, long cap, int* outOff, uint32_t* outQ, uint32_t* outT, uint32_t* outBox, uint8_t* outRefinespace, int* nRevOut);
int oracle_merge_extend(int nsp, const int* matchOff, const uint32_t* mq, const uint32_t* mt, const uint32_t* box, const uint8_t* strand, const int* chrom, const char* read,
                        uint32_t readLen, const char* genome, const uint64_t* chromPos, int K, long cap, int* groupMember, int* anchorOff, uint32_t* aq, uint32_t* at, int* alen,
                        uint32_t* gbox, uint8_t* gstrand, int* gchrom);
int oracle_local_refine_alignment(int nChains, const int* chainOff, const uint32_t* aq, const uint32_t* at, const int* alen, const uint8_t* chainStrand, const int* chainChrom,
                                  const float* firstSdp, const int* numAnchors0, const int* numAnchors1, int LSC, int h, const char* fwd, const char* rc, uint32_t readLen,
                                  const char* genome, const uint64_t* chromPos, const oracle_lra_opts* o, int maxSeg, int* segStrand, int* segSupp, int* segSecondary, int* segN0,
                                  int* segN1, float* segValue, int* segChrom, int* segBlockOff, int* blocks, long blockCap);
long oracle_indel_refine(const int* blocks_in, long n_in, const char* qSeq, long readLen, const char* tSeq, long chromLen, int refineBand, int match, int mismatch, int indel,
                         int endAlign, int* blocks_out, long cap, int* status);
int oracle_refine_breakpoint(int readLen, const int* lBlocks, int nL, int lStrand, const char* lRead, const char* lChrom, int lChromLen, const int* rBlocks, int nR, int rStrand,
                             const char* rRead, const char* rChrom, int rChromLen, int* lOut, int* nLOut, int* rOut, int* nROut);
long oracle_calculate_statistics(const int* blocks, long nb, const char* read, long readLen, const char* genome, const float* lut, long* out_counts, float* out_value, uint32_t* runs,
                                 long cap);
}

// everything MapRead_lowacc reads from Options on this path
struct oracle_map_opts {
  int globalK, globalW, globalMaxFreq, localK, localW, localMaxFreq, localIndexWindow, refineBand, match, mismatch, indel, localBand, refineSpaceDist;
  float anchorstoosparse; int splitdist, window; float initial_anchorbonus, second_anchorbonus, alnthres; int NumAln;
  float gapopen, gapextend, gaproot; int gapCeiling1, gapCeiling2;
  int refineBreakpoint, stats, limitrefine, isOnt;
  lra_clean_opts clean;
};

namespace {

struct Seg {
  int strand = 0, supp = 0, secondary = 0, n0 = 0, n1 = 0, chrom = 0, refine_status = 0, breakpoint = -2 /* no junction on its left */, has_stats = 0;
  float value = 0, nv = 0;
  long counts[18] = {0};
  std::vector<int> a13, blocks;
  std::vector<uint32_t> runs;
};
struct Result {
  std::vector<std::vector<Seg>> groups;
  bool unaligned = true;
  float match_rate = 0;
};

std::string revcomp(const char* s, long n) {
  std::string r((size_t)n, 'N');
  for (long i = 0; i < n; i++) {
    const char c = s[n - 1 - i];
    r[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c == 'a' ? 't' : c == 'c' ? 'g' : c == 'g' ? 'c' : c == 't' ? 'a' : c == 'n' ? 'n' : 'N';
  }
  return r;
}

struct Ref {
  const char* genome; uint64_t G; const uint64_t* chromPos; int nChrom;
  const uint64_t* idxKey; const uint32_t* idxPos; long nIdx;
  long nWg; const uint64_t* gSeqOff; const uint64_t* gBnd; const uint32_t* gTup;
  const float* lut;
};

void map_read(const char* read, uint32_t L, const Ref& R, const oracle_map_opts& o, Result& out) {
  out.groups.clear(); out.unaligned = true; out.match_rate = o.initial_anchorbonus;
  const int K = o.globalK;
  const uint64_t* CH = R.chromPos;
  // a1-a4 (MapRead.h:169-203)
  std::vector<uint64_t> keys((size_t)L + 1); std::vector<uint32_t> pos((size_t)L + 1);
  const long nmm = oracle_store_minimizers(read, L, K, o.globalW, keys.data(), pos.data(), (long)L + 1);
  oracle_sort_minimizers(keys.data(), pos.data(), nmm);
  uint32_t dummy = 0;
  const long nm = oracle_compare_lists(keys.data(), pos.data(), nmm, R.idxKey, R.idxPos, R.nIdx, o.globalMaxFreq, 0, 0, &dummy, &dummy, 0);
  if (nm == 0) return;                                                   // MapRead.h:205-209
  std::vector<uint32_t> qi((size_t)nm), ti((size_t)nm);
  oracle_compare_lists(keys.data(), pos.data(), nmm, R.idxKey, R.idxPos, R.nIdx, o.globalMaxFreq, 0, 0, qi.data(), ti.data(), nm);
  std::vector<uint32_t> mq((size_t)nm), mt((size_t)nm); std::vector<uint64_t> mk((size_t)nm); std::vector<uint8_t> st((size_t)nm);
  for (long i = 0; i < nm; i++) { mq[i] = pos[qi[i]]; mt[i] = R.idxPos[ti[i]]; mk[i] = keys[qi[i]]; }
  oracle_separate_strand(read, R.genome, K, mq.data(), mt.data(), nm, st.data());
  // a5, a7 (Map_lowacc.h:77-153)
  std::vector<int> offs(1, 0); std::vector<uint8_t> cst; std::vector<uint32_t> Q, T; std::vector<int> Ln;
  bool repetitive = false;
  for (int strand = 0; strand < 2; strand++) {
    std::vector<uint32_t> sq, stt; std::vector<uint64_t> sk;
    for (long i = 0; i < nm; i++) if (st[i] == strand) { sq.push_back(mq[i]); stt.push_back(mt[i]); sk.push_back(mk[i]); }
    const long n = (long)sq.size();
    if (n == 0) continue;
    std::vector<uint32_t> oq((size_t)n), ot((size_t)n), cqs((size_t)n), cqe((size_t)n), cts((size_t)n), cte((size_t)n);
    std::vector<long> cs((size_t)n), ce((size_t)n); std::vector<int> cch((size_t)n); std::vector<float> cfr((size_t)n);
    long nclean = 0;
    const long ncl = oracle_clean_matches(sq.data(), stt.data(), sk.data(), n, strand, &o.clean, CH, R.nChrom, oq.data(), ot.data(), &nclean, cs.data(), ce.data(), cqs.data(),
                                          cqe.data(), cts.data(), cte.data(), cch.data(), cfr.data());
    for (long ci = 0; ci < ncl; ci++) {
      const long a = cs[ci], b = ce[ci];
      if (cfr[ci] > 1.0f && cfr[ci] <= 2.0f && b - a >= 500) repetitive = true;           // :86-89
      const int c = cch[ci];
      const uint32_t off = (uint32_t)CH[c];
      const long m = b - a;
      std::vector<uint32_t> tq(oq.begin() + a, oq.begin() + b), tt((size_t)m), eq((size_t)std::max<long>(1, m)), et((size_t)std::max<long>(1, m));
      std::vector<int> el((size_t)std::max<long>(1, m));
      for (long x = 0; x < m; x++) tt[x] = ot[a + x] - off;                               // :118-126
      uint32_t box[4];
      const long ne = oracle_linear_extend(tq.data(), tt.data(), m, strand, K, read, L, R.genome + CH[c], (uint32_t)(CH[c + 1] - CH[c]), eq.data(), et.data(), el.data(), box);
      for (long x = 0; x < ne; x++) { Q.push_back(eq[x]); T.push_back(et[x] + off); Ln.push_back(el[x]); }   // :145-153
      cst.push_back((uint8_t)strand); offs.push_back((int)Q.size());
    }
  }
  if (cst.empty()) return;                                               // :81-85
  const float match_rate = repetitive ? 3.0f : o.initial_anchorbonus;    // :184-185
  out.match_rate = match_rate;
  // a8: primary chains (:186-188)
  const int nF = (int)Q.size(), nC = (int)cst.size();
  oracle_sdp_opts so{match_rate, o.NumAln, o.alnthres, (int)L, o.gapopen, o.gapextend, o.gaproot, o.gapCeiling1, o.gapCeiling2, 0, K};
  const int mc = std::max(1, o.NumAln);
  std::vector<float> fval((size_t)std::max(1, nF)); std::vector<long> fps((size_t)std::max(1, nF)), fpi((size_t)std::max(1, nF)); std::vector<uint8_t> ffl((size_t)std::max(1, nF));
  std::vector<int> coff((size_t)mc + 1); std::vector<uint32_t> cf((size_t)std::max(1, nF)); std::vector<uint8_t> cl((size_t)std::max(1, nF)); std::vector<uint32_t> cbox(4 * (size_t)mc);
  std::vector<float> cv((size_t)mc);
  const int nChains = oracle_sdp_chain(nC, offs.data(), cst.data(), Q.data(), T.data(), Ln.data(), &so, fval.data(), fps.data(), fpi.data(), ffl.data(), mc, coff.data(), cf.data(),
                                       cl.data(), cbox.data(), cv.data());
  if (nChains <= 0) return;                                              // :194-198 (or the reference reads outside an array)
  const std::string rcs = revcomp(read, L);
  const char* fwd = read; const char* rc = rcs.c_str();
  struct QIndex { bool built = false; std::vector<uint64_t> so, bnd; std::vector<uint32_t> tup; } qidx[2];
  auto read_index = [&](int sd) -> QIndex& {                             // :246-250 (both strands in the reference; only the looked-up one matters)
    QIndex& x = qidx[sd];
    if (!x.built) {
      const long nw = ((long)L + o.localIndexWindow - 1) / o.localIndexWindow;
      x.tup.assign((size_t)L + 1, 0); x.bnd.assign((size_t)nw + 1, 0);
      oracle_local_index_seq(sd ? rc : fwd, L, o.localK, o.localW, o.localIndexWindow, o.localMaxFreq, x.tup.data(), (long)L + 1, x.bnd.data());
      x.so.clear();
      for (long p = 0; p < (long)L; p += o.localIndexWindow) x.so.push_back((uint64_t)p);
      x.so.push_back(L);
      if (L == 0) x.so.assign(1, 0);
      x.built = true;
    }
    return x;
  };
  out.unaligned = false;
  for (int p = 0; p < nChains; p++) {
    const int f0 = coff[p], f1 = coff[p + 1], n = f1 - f0;
    // the chain's anchors (trace-back order) with the cluster each lies in
    std::vector<uint32_t> cq((size_t)n), ct((size_t)n); std::vector<int> cln((size_t)n), clOf((size_t)n); std::vector<uint8_t> cstrand((size_t)n), clink((size_t)std::max(1, n));
    for (int i = 0; i < n; i++) {
      const int fr = (int)cf[f0 + i];
      cq[i] = Q[fr]; ct[i] = T[fr]; cln[i] = Ln[fr];
      const int c = (int)(std::upper_bound(offs.begin(), offs.end(), fr) - offs.begin()) - 1;
      clOf[i] = c; cstrand[i] = cst[c];
    }
    for (int i = 0; i + 1 < n; i++) clink[i] = cl[f0 + i];
    // a9 (:189-192, :261-262)
    const int m = std::max(1, n);
    std::vector<uint8_t> keep((size_t)m), lo((size_t)m), spLink((size_t)m), spStrand((size_t)m + 1), sl((size_t)m + 1);
    std::vector<int> spOff((size_t)m + 2), spIdx((size_t)m), spChrom((size_t)m + 1), ciOff((size_t)m + 2), ciIdx((size_t)m);
    std::vector<char> spType((size_t)m + 2); std::vector<uint32_t> spBox(4 * ((size_t)m + 1));
    int nKept = 0, nsl = 0;
    const int nsp = oracle_split_chain(n, cq.data(), ct.data(), cln.data(), cstrand.data(), clOf.data(), clink.data(), CH, R.nChrom + 1, o.splitdist, 1, keep.data(), &nKept, lo.data(),
                                       spOff.data(), spIdx.data(), spLink.data(), spType.data(), spStrand.data(), spChrom.data(), spBox.data(), ciOff.data(), ciIdx.data(), sl.data(),
                                       &nsl);
    if (nsp <= 0) {                                                      // :263-267
      if (p == 0) { out.groups.assign(1, {}); out.unaligned = true; return; }
      break;
    }
    std::vector<uint32_t> kq, kt; std::vector<int> kl, kcl; std::vector<uint8_t> kcs;
    for (int i = 0; i < n; i++) if (keep[i]) { kq.push_back(cq[i]); kt.push_back(ct[i]); kl.push_back(cln[i]); kcl.push_back(clOf[i]); kcs.push_back(cstrand[i]); }
    // a10 (:298)
    std::vector<int> moff(1, 0); std::vector<uint32_t> rmq, rmt, boxes; std::vector<uint8_t> strands; std::vector<int> chroms;
    bool ok = true;
    const oracle_rsc_opts ro{o.window, o.localK, K, o.limitrefine, o.localMaxFreq};
    for (int s = 0; s < nsp && ok; s++) {
      const int sd = spStrand[s];
      QIndex& qx = read_index(sd);
      long cap = 1 << 16;
      for (;;) {
        std::vector<uint32_t> oq((size_t)cap), ot((size_t)cap); uint32_t ob[4] = {0, 0, 0, 0}; float eff = 0;
        const long r = oracle_refine_splitchain((int)kq.size(), kq.data(), kt.data(), kl.data(), kcl.data(), kcs.data(), spOff[s + 1] - spOff[s], spIdx.data() + spOff[s],
                                                spBox.data() + 4 * s, sd, spChrom[s], ciOff[s + 1] - ciOff[s], ciIdx.data() + ciOff[s], CH, R.nChrom, L,
                                                (long)qx.so.size() - 1, qx.so.data(), qx.bnd.data(), qx.tup.data(), R.nWg, R.gSeqOff, R.gBnd, R.gTup, &ro, cap, oq.data(), ot.data(),
                                                ob, &eff);
        if (r < 0) { ok = false; break; }
        if (r <= cap) {
          rmq.insert(rmq.end(), oq.begin(), oq.begin() + r); rmt.insert(rmt.end(), ot.begin(), ot.begin() + r); moff.push_back((int)rmq.size());
          boxes.insert(boxes.end(), ob, ob + 4);
          break;
        }
        cap = r;
      }
      strands.push_back((uint8_t)sd); chroms.push_back(spChrom[s]);
    }
    std::vector<Seg> segs;
    bool reached = false;
    if (ok) {
      // a11 callers (:371)
      const oracle_btwn_opts bo{o.localK, o.localW, o.refineSpaceDist, o.anchorstoosparse, o.match, o.mismatch, o.indel, o.localMaxFreq};
      const long cap = (long)rmq.size() + 4 * (long)L + 1024;
      std::vector<int> oo((size_t)nsp + 1); std::vector<uint32_t> oq((size_t)cap), ot((size_t)cap), ob(4 * (size_t)std::max(1, nsp)); std::vector<uint8_t> orf((size_t)std::max(1, nsp));
      int nrev = 0;
      if (sl.empty()) sl.push_back(0);
      const long nb = oracle_refine_btwn_splitchain(nsp, moff.data(), rmq.data(), rmt.data(), boxes.data(), strands.data(), chroms.data(), sl.data(), fwd, rc, L, R.genome, CH, R.nChrom,
                                                    &bo, cap, oo.data(), oq.data(), ot.data(), ob.data(), orf.data(), &nrev);
      if (nb > 0) {                                                      // SizeRefinedClusters > 0 (:486-491)
        reached = true;
        // MergeChain, second LinearExtend + Trim (:440-476)
        const long cap2 = nb + 8;
        std::vector<int> gm((size_t)nsp + 2), ao((size_t)nsp + 2), al((size_t)cap2), gc((size_t)nsp + 1); std::vector<uint32_t> aq((size_t)cap2), at((size_t)cap2), gb(4 * ((size_t)nsp + 1));
        std::vector<uint8_t> gs((size_t)nsp + 1);
        const int ng = oracle_merge_extend(nsp, oo.data(), oq.data(), ot.data(), ob.data(), strands.data(), chroms.data(), fwd, L, R.genome, CH, o.localK, cap2, gm.data(), ao.data(),
                                           aq.data(), at.data(), al.data(), gb.data(), gs.data(), gc.data());
        // second sparse DP + RemovePairedIndels / RemoveSpuriousAnchors (:529-541)
        std::vector<int> chOff(1, 0), chChrom, chN1; std::vector<uint32_t> uq, ut; std::vector<int> ul; std::vector<uint8_t> chStrand; std::vector<float> chVal;
        for (int g = 0; g < ng; g++) {
          const int a0 = ao[g], a1 = ao[g + 1], na = a1 - a0;
          if (na == 0) continue;
          const int co2[2] = {0, na}; const uint8_t sg = gs[g];
          oracle_sdp_opts s2{o.second_anchorbonus, o.NumAln, o.alnthres, (int)L, o.gapopen, o.gapextend, o.gaproot, o.gapCeiling1, o.gapCeiling2, 1, K};
          std::vector<float> v2((size_t)na); std::vector<long> ps2((size_t)na), pi2((size_t)na); std::vector<uint8_t> fl2((size_t)na), cl2((size_t)na);
          std::vector<int> coff2((size_t)mc + 1); std::vector<uint32_t> cf2((size_t)na), box2(4 * (size_t)mc); std::vector<float> cv2((size_t)mc);
          const int r2 = oracle_sdp_chain(1, co2, &sg, aq.data() + a0, at.data() + a0, al.data() + a0, &s2, v2.data(), ps2.data(), pi2.data(), fl2.data(), mc, coff2.data(), cf2.data(),
                                          cl2.data(), box2.data(), cv2.data());
          if (r2 <= 0) continue;
          const int len2 = coff2[1] - coff2[0];
          std::vector<uint32_t> xq((size_t)len2), xt((size_t)len2); std::vector<int> xl((size_t)len2); std::vector<uint8_t> xs((size_t)len2, sg), kp((size_t)std::max(1, len2)),
              lo2((size_t)std::max(1, len2)), nolink((size_t)std::max(1, len2), 0);
          for (int i = 0; i < len2; i++) { const uint32_t ix = cf2[coff2[0] + i]; xq[i] = aq[a0 + ix]; xt[i] = at[a0 + ix]; xl[i] = al[a0 + ix]; }
          const int ops[2] = {2, 4}; int nl = 0;
          oracle_filter_chain(len2, xq.data(), xt.data(), xl.data(), xs.data(), nolink.data(), 0, ops, 2, kp.data(), lo2.data(), &nl);
          for (int i = 0; i < len2; i++) if (kp[i]) { uq.push_back(xq[i]); ut.push_back(xt[i]); ul.push_back(xl[i]); }
          chOff.push_back((int)uq.size()); chStrand.push_back(sg); chChrom.push_back(gc[g]); chVal.push_back(cv2[0]); chN1.push_back(len2);
        }
        const int nch = (int)chStrand.size();
        if (nch) {
          // a13 (:574-576)
          int lsc = 0;
          for (int c = 1; c < nch; c++) if (chOff[c + 1] - chOff[c] > chOff[lsc + 1] - chOff[lsc]) lsc = c;     // LargestSplitChain: first maximum
          std::vector<int> n0v((size_t)nch, n);                         // chains[p].NumOfAnchors0 (:532)
          const oracle_lra_opts lo_{o.localW, o.localW, o.localMaxFreq, o.match, o.mismatch, o.indel, o.localBand, 1, o.isOnt, o.gapopen, o.gapextend, o.gaproot, o.gapCeiling1,
                                    o.gapCeiling2};
          const int maxSeg = 4 * (int)uq.size() + 8; const long bcap = 4 * ((long)uq.size() + (long)L) + 64;
          std::vector<int> s0((size_t)maxSeg), s1((size_t)maxSeg), s2v((size_t)maxSeg), s3((size_t)maxSeg), s4((size_t)maxSeg), sc((size_t)maxSeg), sbo((size_t)maxSeg + 1),
              blk(3 * (size_t)bcap);
          std::vector<float> sv((size_t)maxSeg);
          if (uq.empty()) { uq.push_back(0); ut.push_back(0); ul.push_back(0); }
          const int ns = oracle_local_refine_alignment(nch, chOff.data(), uq.data(), ut.data(), ul.data(), chStrand.data(), chChrom.data(), chVal.data(), n0v.data(), chN1.data(), lsc, p,
                                                       fwd, rc, L, R.genome, CH, &lo_, maxSeg, s0.data(), s1.data(), s2v.data(), s3.data(), s4.data(), sv.data(), sc.data(), sbo.data(),
                                                       blk.data(), bcap);
          for (int i = 0; i < ns; i++) {
            Seg sg_; sg_.strand = s0[i]; sg_.supp = s1[i]; sg_.secondary = s2v[i]; sg_.n0 = s3[i]; sg_.n1 = s4[i]; sg_.value = sv[i]; sg_.chrom = sc[i];
            sg_.a13.assign(blk.begin() + 3 * sbo[i], blk.begin() + 3 * sbo[i + 1]);
            segs.push_back(std::move(sg_));
          }
        }
      }
    }
    if (!reached) {                                                      // :486-491 (or a stage hit undefined behaviour)
      if (p == 0) { out.groups.assign(1, {}); out.unaligned = true; return; }
      break;
    }
    for (Seg& s : segs) {                                                // a14 (:582)
      const char* sb = s.strand == 0 ? fwd : rc;
      const long nb = (long)s.a13.size() / 3;
      const int* b = s.a13.data();
      const long cap = nb ? (long)0 + 2 * nb + 64 + (b[3 * (nb - 1) + 1] + b[3 * (nb - 1) + 2] - b[1]) + (b[3 * (nb - 1)] + b[3 * (nb - 1) + 2] - b[0]) : 64;
      long sum = 0;
      for (long i = 0; i < nb; i++) sum += b[3 * i + 2];
      s.blocks.assign(3 * (size_t)(cap + sum), 0);
      int stt = 0;
      const uint64_t c0 = CH[s.chrom], clen = CH[s.chrom + 1] - c0;
      const long m = oracle_indel_refine(b, nb, sb, L, R.genome + c0, (long)clen, o.refineBand, o.match, o.mismatch, o.indel, 0, s.blocks.data(), cap + sum, &stt);
      s.blocks.resize(3 * (size_t)std::max<long>(0, m)); s.refine_status = stt;
    }
    if (o.refineBreakpoint)                                              // a15 (:585-595): segments come right to left on the read
      for (size_t si = 1; si < segs.size(); si++) {
        Seg& l = segs[si]; Seg& r = segs[si - 1];
        const int nl = (int)l.blocks.size() / 3, nr = (int)r.blocks.size() / 3;
        std::vector<int> lo3(3 * ((size_t)nl + 502)), ro3(3 * ((size_t)nr + 502)); int nlo = 0, nro = 0;
        const uint64_t lc0 = CH[l.chrom], rc0 = CH[r.chrom];
        const int ret = oracle_refine_breakpoint((int)L, l.blocks.data(), nl, l.strand, l.strand == 0 ? fwd : rc, R.genome + lc0, (int)(CH[l.chrom + 1] - lc0), r.blocks.data(), nr,
                                                 r.strand, r.strand == 0 ? fwd : rc, R.genome + rc0, (int)(CH[r.chrom + 1] - rc0), lo3.data(), &nlo, ro3.data(), &nro);
        if (ret >= 0) { l.blocks.assign(lo3.begin(), lo3.begin() + 3 * nlo); r.blocks.assign(ro3.begin(), ro3.begin() + 3 * nro); }
        l.breakpoint = ret;
      }
    if (o.stats)
      for (Seg& s : segs) {                                              // a16 (:597-599)
        if (s.refine_status != 0 || s.blocks.empty()) continue;
        const long nb = (long)s.blocks.size() / 3;
        long sum = 0;
        for (long i = 0; i < nb; i++) sum += s.blocks[3 * i + 2];
        const long cap = sum * 2 + 4 * nb + 16;
        s.runs.assign((size_t)cap, 0);
        const long nr = oracle_calculate_statistics(s.blocks.data(), nb, s.strand == 0 ? fwd : rc, L, R.genome + CH[s.chrom], R.lut, s.counts, &s.nv, s.runs.data(), cap);
        s.runs.resize((size_t)nr); s.has_stats = 1;
      }
    const bool none = segs.empty();
    out.groups.push_back(std::move(segs));
    if (p == 0 && none) { out.unaligned = true; return; }                // :577-580
  }
}

thread_local std::vector<int32_t> g_flat;

void flatten(const Result& r, std::vector<int32_t>& f) {
  f.clear();
  auto fbits = [](float v) { int32_t b; memcpy(&b, &v, 4); return b; };
  f.push_back(r.unaligned ? 1 : 0); f.push_back((int32_t)r.groups.size()); f.push_back(fbits(r.match_rate));
  for (const auto& g : r.groups) {
    f.push_back((int32_t)g.size());
    for (const Seg& s : g) {
      const int32_t h[14] = {s.strand, s.supp, s.secondary, s.n0, s.n1, s.chrom, fbits(s.value), s.refine_status, s.breakpoint, s.has_stats, fbits(s.nv), (int32_t)(s.a13.size() / 3),
                             (int32_t)(s.blocks.size() / 3), (int32_t)s.runs.size()};
      f.insert(f.end(), h, h + 14);
      for (int i = 0; i < 18; i++) f.push_back((int32_t)s.counts[i]);
      f.insert(f.end(), s.a13.begin(), s.a13.end());
      f.insert(f.end(), s.blocks.begin(), s.blocks.end());
      for (uint32_t x : s.runs) f.push_back((int32_t)x);
    }
  }
}

}  // namespace

// One read.  Returns the number of int32 words of the flattened result (fetch it with oracle_map_read_result from the same thread):
//   [unaligned, n_groups, match_rate bits] then per group [n_segs] and per segment 14 header words {strand, supp, secondary, n0, n1, chrom, value bits,
//   refine_status, breakpoint, has_stats, NV bits, n_a13_blocks, n_blocks, n_runs}, 18 counters, the a13 blocks, the refined blocks, the CIGAR runs.
extern "C" long oracle_map_read_lowacc(const char* read, uint32_t readLen, const char* genome, uint64_t G, const uint64_t* chromPos, int nChrom, const uint64_t* idxKey,
                                       const uint32_t* idxPos, long nIdx, long nWg, const uint64_t* gSeqOff, const uint64_t* gBnd, const uint32_t* gTup, const float* lut,
                                       const oracle_map_opts* o) {
  const Ref R{genome, G, chromPos, nChrom, idxKey, idxPos, nIdx, nWg, gSeqOff, gBnd, gTup, lut};
  Result res;
  map_read(read, readLen, R, *o, res);
  flatten(res, g_flat);
  return (long)g_flat.size();
}
extern "C" void oracle_map_read_result(int32_t* out) { memcpy(out, g_flat.data(), g_flat.size() * 4); }

// reads [first, first + n) of a batch (bases back to back, off[i] .. off[i+1]) on n_threads threads pulling reads from a shared counter.
// Out: seconds of wall time, bases and alignments (SegAlignments) done, a checksum over every refined block and counter (thread-count independent).
extern "C" int oracle_map_reads_lowacc_mt(const char* reads, const uint64_t* off, long first, long n, const char* genome, uint64_t G, const uint64_t* chromPos, int nChrom,
                                          const uint64_t* idxKey, const uint32_t* idxPos, long nIdx, long nWg, const uint64_t* gSeqOff, const uint64_t* gBnd, const uint32_t* gTup,
                                          const float* lut, const oracle_map_opts* o, int n_threads, double* seconds, long* bases, long* n_alignments, uint64_t* checksum) {
  const Ref R{genome, G, chromPos, nChrom, idxKey, idxPos, nIdx, nWg, gSeqOff, gBnd, gTup, lut};
  std::atomic<long> next(0), nal(0), nb(0);
  std::atomic<uint64_t> sum(0);
  const int T = std::max(1, n_threads);
  const auto t0 = std::chrono::steady_clock::now();
  auto work = [&]() {
    Result res;
    for (;;) {
      const long i = next.fetch_add(1);
      if (i >= n) break;
      const long r = first + i;
      const uint32_t L = (uint32_t)(off[r + 1] - off[r]);
      map_read(reads + off[r], L, R, *o, res);
      uint64_t h = 0; long a = 0;
      for (const auto& g : res.groups) for (const Seg& s : g) {
        a++;
        for (int x : s.blocks) h = h * 1099511628211ULL + (uint64_t)(uint32_t)x;
        for (int c = 0; c < 18; c++) h = h * 1099511628211ULL + (uint64_t)s.counts[c];
      }
      sum.fetch_add(h * (uint64_t)(r + 1)); nal.fetch_add(a); nb.fetch_add((long)L);
    }
  };
  std::vector<std::thread> th;
  for (int t = 0; t < T; t++) th.emplace_back(work);
  for (auto& x : th) x.join();
  *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  *bases = nb.load(); *n_alignments = nal.load(); *checksum = sum.load();
  return 0;
}
