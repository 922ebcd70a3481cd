// oracle/refine_btwn_clusters.cpp -- TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// CPU restatement of the gap seeding between the clusters of the chains of one read on the high-accuracy path:
//   RefineBtwnClusters_chain      ClusterRefine.h:433-614  (Map_highacc.h:513-518: every chain (p, h) of the read in order, on the SHARED
//                                                           RefinedClusters: a cluster's box as one step leaves it is what the next step reads)
//     RefineBtwnSpace             ClusterRefine.h:331-431  (oracle_refine_btwn_space; decision 1 / 3 append the pairs to the cluster and
//                                                           SetClusterBoundariesFromMatches, 3 also sets anchorfreq = 1; decision 2 fills
//                                                           RevBtwnCluster, which MapRead_highacc never reads again)
// Parity status: PARITY UNPINNED -- ClusterRefine.h needs Genome.h (htslib); restated from the source text.
#include "oracle_common.h"
#include <algorithm>
#include <vector>

extern "C" int oracle_refine_btwn_space(int K, int W, int twoblocks, int read_type, float anchorstoosparse, int match, int mismatch, int indel, long maxFreq, const char* fwd,
                                        const char* rc, uint32_t readLen, const char* chrom, uint32_t qe, uint32_t qs, uint32_t te, uint32_t ts, int st, uint32_t lrts,
                                        uint32_t lrlength, uint32_t* outQ, uint32_t* outT, long cap, long* n_out, float* eff_out, float* reff_out);

namespace {
struct Cl { std::vector<uint32_t> q, t; uint32_t qS, qE, tS, tE; int strand, chrom; float freq; int refinespace; };
}

// read_type: 0 clr / raw, 1 ont, 2 ccs, 3 contig (as lra_refine_btwn_space_batch).  Clusters: t relative to the chromosome.  Chains: CSR chainOff over ch
// (cluster indices, in the chain's order: ch[0] is the cluster nearest the read's end).  Out: the clusters' match lists after all chains (CSR outOff
// over outQ / outT, appended pairs behind the old matches), box / freq updated in place, refinespace per cluster.  Returns the total number of matches
// (> cap: nothing written).
extern "C" long oracle_refine_btwn_clusters_chains(int nCl, const int* matchOff, const uint32_t* mq, const uint32_t* mt, uint32_t* box, const uint8_t* strand, const int* chrom,
                                                   float* freq, uint8_t* refinespace, int nChains, const int* chainOff, const int* ch, int K, int W, int read_type,
                                                   float anchorstoosparse, int match, int mismatch, int indel, long maxFreq, const char* fwd, const char* rc, uint32_t readLen,
                                                   const char* genome, const uint64_t* chromPos, long cap, int* outOff, uint32_t* outQ, uint32_t* outT) {
  std::vector<Cl> C((size_t)nCl);
  for (int c = 0; c < nCl; c++) {
    C[c].q.assign(mq + matchOff[c], mq + matchOff[c + 1]); C[c].t.assign(mt + matchOff[c], mt + matchOff[c + 1]);
    C[c].qS = box[4 * c]; C[c].qE = box[4 * c + 1]; C[c].tS = box[4 * c + 2]; C[c].tE = box[4 * c + 3];
    C[c].strand = strand[c]; C[c].chrom = chrom[c]; C[c].freq = freq[c]; C[c].refinespace = 0;
  }
  auto chromLen = [&](int ci) { return (uint32_t)(chromPos[ci + 1] - chromPos[ci]); };
  auto btwn = [&](int twoblocks, Cl& cl, uint32_t qe, uint32_t qs, uint32_t te, uint32_t ts, int st, uint32_t lrts, uint32_t lrlength) {   // RefineBtwnSpace :331-431
    long cp = 4L * ((long)(qe - qs) + (long)(te - ts + lrlength)) + 64, n = 0;
    std::vector<uint32_t> Q((size_t)cp), T((size_t)cp);
    float eff, reff;
    const int d = oracle_refine_btwn_space(K, W, twoblocks, read_type, anchorstoosparse, match, mismatch, indel, maxFreq, fwd, rc, readLen, genome + chromPos[cl.chrom], qe, qs,
                                           te, ts, st, lrts, lrlength, Q.data(), T.data(), cp, &n, &eff, &reff);
    if (d == 1 || d == 3) {
      cl.q.insert(cl.q.end(), Q.begin(), Q.begin() + n); cl.t.insert(cl.t.end(), T.begin(), T.begin() + n);
      cl.qS = cl.q[0]; cl.qE = cl.qS + K; cl.tS = cl.t[0]; cl.tE = cl.tS + K;            // SetClusterBoundariesFromMatches Clustering.h:308-322
      for (size_t i = 1; i < cl.q.size(); i++) {
        cl.tE = std::max(cl.tE, cl.t[i] + (uint32_t)K); cl.tS = std::min(cl.tS, cl.t[i]); cl.qE = std::max(cl.qE, cl.q[i] + (uint32_t)K); cl.qS = std::min(cl.qS, cl.q[i]);
      }
      cl.refinespace = 1;
      if (d == 3) cl.freq = 1.0f;
    }
  };
  const bool contig = read_type == 3;
  const int low_b = contig ? 1000 : 20;                                     // :445-453
  const int upper = contig ? 100000 : 50000;
  for (int x = 0; x < nChains; x++) {
    const int* chn = ch + chainOff[x];
    const int len = chainOff[x + 1] - chainOff[x];
    if (len == 0) continue;                                                 // Map_highacc.h:515
    bool twoblocks = false;
    bool st2 = false;
    for (int c = 1; c < len; c++) {                                         // :454-545
      Cl& cur = C[chn[c]]; Cl& prev = C[chn[c - 1]];
      const uint32_t qs = cur.qE, qe = prev.qS;
      uint32_t te1 = 0, ts1 = 0, te2 = 0, ts2 = 0;
      bool st1;
      if (qe <= qs || cur.chrom != prev.chrom) continue;
      if (contig) twoblocks = false;
      if (cur.strand == prev.strand) {
        twoblocks = false; st1 = cur.strand;
        if (cur.tE <= prev.tS) { ts1 = cur.tE; te1 = prev.tS; }
        else if (cur.tS > prev.tE) { ts1 = prev.tE; te1 = cur.tS; }
        else continue;
      } else if (!contig) {
        st1 = cur.strand; st2 = prev.strand; twoblocks = true;
        const uint32_t gl = chromLen(cur.chrom), d = qe - qs;
        if (cur.tE <= prev.tS) {
          if (st1 == 0) { ts1 = cur.tE; te1 = std::min(gl, ts1 + d); ts2 = prev.tE; te2 = std::min(gl, ts2 + d); }
          else { te1 = cur.tS; ts1 = te1 > d ? te1 - d : 0; te2 = prev.tS; ts2 = te2 > d ? te2 - d : 0; }
        } else if (cur.tS > prev.tE) {
          if (st1 == 0) { ts1 = cur.tE; te1 = std::min(gl, ts1 + d); te2 = cur.tS; ts2 = te2 > d ? te2 - d : 0; }
          else { te1 = cur.tS; ts1 = te1 > d ? te1 - d : 0; te2 = prev.tS; ts2 = te2 > d ? te2 - d : 0; }
        } else continue;
      } else {
        // contig reads with clusters on different strands: the reference falls through with st1 uninitialised and te1 = ts1 = 0 -> `te1 <= ts1` skips
        st1 = false;
      }
      if (te1 <= ts1) continue;
      int SpaceLength = (int)std::max(qe - qs, te1 - ts1);
      if (SpaceLength >= low_b && SpaceLength <= upper) btwn(twoblocks, cur, qe, qs, te1, ts1, st1, 0, 0);
      if (te2 <= ts2) continue;
      SpaceLength = (int)std::max(qe - qs, te2 - ts2);
      if (SpaceLength >= low_b && SpaceLength <= upper) btwn(twoblocks, prev, qe, qs, te2, ts2, st2, 0, 0);
    }
    {                                                                       // :549-579 the read's end
      Cl& rh = C[chn[0]];
      const bool st = rh.strand;
      uint32_t qs = rh.qE, qe = readLen, te = 0, ts = 0;
      if (st == 0) { ts = rh.tE; te = ts + qe - qs; }
      else { te = rh.tS; if (te > qe - qs) ts = te - (qe - qs); else te = 0; }
      if (qe > qs && te > ts) {
        const int SpaceLength = (int)std::max(qe - qs, te - ts);
        if (SpaceLength >= low_b && SpaceLength < upper && te + 500 < chromLen(rh.chrom)) {
          uint32_t lrts = 0, lrlength = 0;
          if (st == 0) { lrts = 0; lrlength = 500; }
          else { if (ts > 500) lrts = 500; lrlength = lrts; }
          btwn(1, rh, qe, qs, te, ts, st, lrts, lrlength);
        }
      }
    }
    {                                                                       // :583-612 the read's start
      Cl& lh = C[chn[len - 1]];
      const bool st = lh.strand;
      uint32_t qs = 0, qe = lh.qS, te, ts;
      if (st == 0) { te = lh.tS; ts = te > qe - qs ? te - (qe - qs) : 0; }
      else { ts = lh.tE; te = ts + (qe - qs); }
      if (qe > qs && te > ts) {
        const int SpaceLength = (int)std::max(qe - qs, te - ts);
        if (SpaceLength >= low_b && SpaceLength < upper && te + 500 < chromLen(lh.chrom)) {
          uint32_t lrts = 0, lrlength = 0;
          if (st == 0) { if (ts > 500) lrts = 500; lrlength = lrts; }
          else { lrts = 0; lrlength = 500; }
          btwn(1, lh, qe, qs, te, ts, st, lrts, lrlength);
        }
      }
    }
  }
  long tot = 0;
  for (int c = 0; c < nCl; c++) tot += (long)C[c].q.size();
  for (int c = 0; c < nCl; c++) {
    box[4 * c] = C[c].qS; box[4 * c + 1] = C[c].qE; box[4 * c + 2] = C[c].tS; box[4 * c + 3] = C[c].tE; freq[c] = C[c].freq; refinespace[c] = (uint8_t)C[c].refinespace;
  }
  if (tot > cap) return tot;
  long at = 0;
  for (int c = 0; c < nCl; c++) {
    outOff[c] = (int)at;
    for (size_t i = 0; i < C[c].q.size(); i++) { outQ[at] = C[c].q[i]; outT[at] = C[c].t[i]; at++; }
  }
  outOff[nCl] = (int)at;
  return tot;
}
