// oracle/refine_btwn_space.cpp -- CPU restatement of RefineBtwnSpace (reference: ClusterRefine.h:331-432; called by RefineBtwnClusters_chain
// :433 on the high-accuracy path) for ONE space, up to (not including) the vector insert / SetClusterBoundariesFromMatches it ends with.
// TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's cpu_baseline leg); the product never links it.  PARITY UNPINNED (htslib-dependent
// header); the two RefineSpace calls inside are oracle_refine_space (its pieces pinned, see refine_space.cpp).
#include <math.h>
#include <stdint.h>
#include <algorithm>
#include <vector>

extern "C" long oracle_refine_space(const char* q, int qLen, const char* t, int tLen, uint32_t tSpan, int K, int W, int refineSpaceDiag, int match,
                                    int mismatch, int indel, long maxFreq, uint32_t qAdd, uint32_t tAdd, uint32_t flipLen, uint32_t* outQ,
                                    uint32_t* outT, long cap, float* identity);

// fwd / rc: strands[0] / strands[1]; chrom: genome.seqs[ChromIndex]; read_type: 0 ont, 1 clr, 2 ccs, 3 contig.
// Returns the decision: 0 nothing happens (:371), 1 EndPairs appended (:363-369), 3 EndPairs appended after the reverse strand was tried
// (:415-421, anchorfreq = 1), 2 a RevBtwnCluster is made of the reverse-strand pairs (:422-431, the function returns 1); pairs in out.
extern "C" int oracle_refine_btwn_space(int K, int W, int twoblocks, int read_type, float anchorstoosparse, int match, int mismatch, int indel, long maxFreq,
                                        const char* fwd, const char* rc, uint32_t readLen, const char* chrom, uint32_t qe, uint32_t qs, uint32_t te, uint32_t ts,
                                        int st, uint32_t lrts, uint32_t lrlength, uint32_t* outQ, uint32_t* outT, long cap, long* n_out, float* eff_out,
                                        float* reff_out) {
  if (st == 1) { uint32_t t = qs; qs = readLen - qe; qe = readLen - t; }                               // :336-340
  int refineSpaceDiag = 0;                                                                            // :341-350
  if (read_type == 3 || read_type == 2) refineSpaceDiag = std::min((int)floorf(std::max(100.f, 0.01f * (qe - qs))), 100);
  else if (read_type == 1 || read_type == 0) refineSpaceDiag = std::min((int)floorf(std::max(100.f, 0.15f * (qe - qs))), 1000);
  auto space = [&](int s, uint32_t a_qs, uint32_t a_qe, std::vector<uint32_t>& Q, std::vector<uint32_t>& T) {
    long c = 4L * ((a_qe - a_qs) + (te - ts + lrlength)) + 64;
    float ident;
    while (true) {
      Q.assign(c, 0); T.assign(c, 0);
      long n = oracle_refine_space((s ? rc : fwd) + a_qs, (int)(a_qe - a_qs), chrom + (ts - lrts), (int)(te - ts + lrlength), te - (ts - lrts), K, W, refineSpaceDiag,
                                   match, mismatch, indel, maxFreq, a_qs, ts - lrts, s == 1 ? readLen : 0, Q.data(), T.data(), c, &ident);
      if (n <= c) { Q.resize(n); T.resize(n); return; }
      c = n;
    }
  };
  std::vector<uint32_t> eq, et, rq, rt;
  space(st, qs, qe, eq, et);
  const float eff = ((float)eq.size()) / std::min(qe - qs, te - ts);                                  // :358
  *eff_out = eff; *reff_out = -1.f;
  auto give = [&](std::vector<uint32_t>& Q, std::vector<uint32_t>& T) {
    *n_out = (long)Q.size();
    for (size_t i = 0; i < Q.size() && (long)i < cap; i++) { outQ[i] = Q[i]; outT[i] = T[i]; }
  };
  *n_out = 0;
  if ((eq.size() > 0 && twoblocks) || (eq.size() > 0 && eff >= anchorstoosparse * 2)) { give(eq, et); return 1; }
  if (twoblocks) return 0;
  const int rst = st == 1 ? 0 : 1;
  { uint32_t t = qs; qs = readLen - qe; qe = readLen - t; }                                           // :373-375
  space(rst, qs, qe, rq, rt);
  const float reff = ((float)rq.size()) / std::min(qe - qs, te - ts);
  *reff_out = reff;
  if (eff >= reff) { give(eq, et); return 3; }
  give(rq, rt);
  return 2;
}
