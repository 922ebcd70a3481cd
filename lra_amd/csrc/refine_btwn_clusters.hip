// lra_amd/csrc/refine_btwn_clusters.hip -- SURVEY §8a row a11, the caller on the high-accuracy path: RefineBtwnClusters_chain
// (ClusterRefine.h:433-614, called for every chain (p, h) of a read at Map_highacc.h:513-518) around RefineBtwnSpace (:331-431 =
// lra_refine_btwn_space_batch).  gfx950 only.
// The loop is serial in what it reads: step c of a chain takes the boxes of clusters ch[c] and ch[c-1] as the steps before left them (a refined
// space appends its pairs to the cluster and SetClusterBoundariesFromMatches moves the box), and the chains of a read share the clusters.  So the
// reads advance in lock step: every round, one lane per read walks its own cursor (chain, step) forward to the next step that has something to
// refine and plans its <= 2 RefineBtwnSpace problems (the case analysis of :461-543, then the read's end :549-579 and start :583-612); ONE
// lra_refine_btwn_space_batch runs all problems of the round; one wave per problem folds the kept pairs into the cluster's box.  The pairs stay
// in per-round buffers; at the end every cluster's list = its matches + its appended segments in order.  Rounds = the longest read's number of
// steps (a handful).  Decision 2 of RefineBtwnSpace (a new cluster on the other strand) fills RevBtwnCluster, which MapRead_highacc never reads.
// Algorithmic bytes: 40 B per problem + what RefineSpace reads + 8 B per pair.
#include "common.h"
#include "scan.h"
#include <algorithm>
#include <vector>

namespace {

struct RbcArgs {
  int n_reads; const uint64_t* readChainOff; const uint64_t* chainOff; const uint32_t* ch;
  uint32_t* box; const int32_t* strand; const int32_t* chrom; float* freq; uint8_t* refinespace;
  const uint64_t* read_off; const uint64_t* pos;
  int contig, low_b, upper;
  uint32_t* curChain; uint32_t* curStep;         // per read cursor
  // planned problems: slots 2 r, 2 r + 1
  uint32_t* valid; uint32_t* pCluster; uint32_t* pQs; uint32_t* pQe; uint32_t* pTs; uint32_t* pTe; int32_t* pSt; uint8_t* pTwo; uint32_t* pRead; int32_t* pChrom;
  uint32_t* pLrts; uint32_t* pLrlen;
  uint32_t* active;                               // reads that still have steps
};

__global__ void rbc_plan(RbcArgs a) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= a.n_reads) return;
  a.valid[2 * r] = 0; a.valid[2 * r + 1] = 0;
  const uint64_t c0 = a.readChainOff[r], c1 = a.readChainOff[r + 1];
  uint32_t x = a.curChain[r], step = a.curStep[r];
  const uint32_t readLen = (uint32_t)(a.read_off[r + 1] - a.read_off[r]);
  int np = 0;
  auto emit = [&](uint32_t cl, uint32_t qe, uint32_t qs, uint32_t te, uint32_t ts, int st, int two, uint32_t lrts, uint32_t lrlen) {
    const int s = 2 * r + np;
    a.valid[s] = 1; a.pCluster[s] = cl; a.pQs[s] = qs; a.pQe[s] = qe; a.pTs[s] = ts; a.pTe[s] = te; a.pSt[s] = st; a.pTwo[s] = (uint8_t)two; a.pRead[s] = (uint32_t)r;
    a.pChrom[s] = a.chrom[cl]; a.pLrts[s] = lrts; a.pLrlen[s] = lrlen;
    np++;
  };
  while (c0 + x < c1 && np == 0) {
    const uint64_t e0 = a.chainOff[c0 + x];
    const uint32_t len = (uint32_t)(a.chainOff[c0 + x + 1] - e0);
    if (len == 0) { x++; step = 1; continue; }                           // Map_highacc.h:515
    const uint32_t* chn = a.ch + e0;
    auto B = [&](uint32_t cl, int k) { return a.box[4 * (size_t)cl + k]; };
    auto glen = [&](uint32_t cl) { const int ci = a.chrom[cl]; return (uint32_t)(a.pos[ci + 1] - a.pos[ci]); };
    if (step < len) {                                                     // :454-545, c = step
      const uint32_t cur = chn[step], prev = chn[step - 1];
      step++;
      const uint32_t qs = B(cur, 1), qe = B(prev, 0);
      uint32_t te1 = 0, ts1 = 0, te2 = 0, ts2 = 0;
      int st1 = 0, st2 = 0, two = 0;
      if (qe <= qs || a.chrom[cur] != a.chrom[prev]) continue;
      if (a.strand[cur] == a.strand[prev]) {
        st1 = a.strand[cur];
        if (B(cur, 3) <= B(prev, 2)) { ts1 = B(cur, 3); te1 = B(prev, 2); }
        else if (B(cur, 2) > B(prev, 3)) { ts1 = B(prev, 3); te1 = B(cur, 2); }
        else continue;
      } else if (!a.contig) {
        st1 = a.strand[cur]; st2 = a.strand[prev]; two = 1;
        const uint32_t gl = glen(cur), d = qe - qs;
        if (B(cur, 3) <= B(prev, 2)) {
          if (st1 == 0) { ts1 = B(cur, 3); te1 = min(gl, ts1 + d); ts2 = B(prev, 3); te2 = min(gl, ts2 + d); }
          else { te1 = B(cur, 2); ts1 = te1 > d ? te1 - d : 0; te2 = B(prev, 2); ts2 = te2 > d ? te2 - d : 0; }
        } else if (B(cur, 2) > B(prev, 3)) {
          if (st1 == 0) { ts1 = B(cur, 3); te1 = min(gl, ts1 + d); te2 = B(cur, 2); ts2 = te2 > d ? te2 - d : 0; }
          else { te1 = B(cur, 2); ts1 = te1 > d ? te1 - d : 0; te2 = B(prev, 2); ts2 = te2 > d ? te2 - d : 0; }
        } else continue;
      }
      if (te1 <= ts1) continue;
      int SpaceLength = (int)max(qe - qs, te1 - ts1);
      if (SpaceLength >= a.low_b && SpaceLength <= a.upper) emit(cur, qe, qs, te1, ts1, st1, two, 0, 0);
      if (te2 <= ts2) continue;
      SpaceLength = (int)max(qe - qs, te2 - ts2);
      if (SpaceLength >= a.low_b && SpaceLength <= a.upper) emit(prev, qe, qs, te2, ts2, st2, two, 0, 0);
    } else if (step == len) {                                             // :549-579 the read's end
      step++;
      const uint32_t rh = chn[0];
      const int st = a.strand[rh];
      uint32_t qs = B(rh, 1), qe = readLen, te = 0, ts = 0;
      if (st == 0) { ts = B(rh, 3); te = ts + qe - qs; }
      else { te = B(rh, 2); if (te > qe - qs) ts = te - (qe - qs); else te = 0; }
      if (qe > qs && te > ts) {
        const int SpaceLength = (int)max(qe - qs, te - ts);
        if (SpaceLength >= a.low_b && SpaceLength < a.upper && te + 500 < glen(rh)) {
          uint32_t lrts = 0, lrlen = 0;
          if (st == 0) { lrts = 0; lrlen = 500; } else { if (ts > 500) lrts = 500; lrlen = lrts; }
          emit(rh, qe, qs, te, ts, st, 1, lrts, lrlen);
        }
      }
    } else {                                                              // :583-612 the read's start, then the next chain
      const uint32_t lh = chn[len - 1];
      const int st = a.strand[lh];
      uint32_t qs = 0, qe = B(lh, 0), te, ts;
      if (st == 0) { te = B(lh, 2); ts = te > qe - qs ? te - (qe - qs) : 0; }
      else { ts = B(lh, 3); te = ts + (qe - qs); }
      x++; step = 1;
      if (qe > qs && te > ts) {
        const int SpaceLength = (int)max(qe - qs, te - ts);
        if (SpaceLength >= a.low_b && SpaceLength < a.upper && te + 500 < glen(lh)) {
          uint32_t lrts = 0, lrlen = 0;
          if (st == 0) { if (ts > 500) lrts = 500; lrlen = lrts; } else { lrts = 0; lrlen = 500; }
          emit(lh, qe, qs, te, ts, st, 1, lrts, lrlen);
        }
      }
    }
  }
  a.curChain[r] = x; a.curStep[r] = step;
  a.active[r] = (np > 0 || c0 + x < c1) ? 1u : 0u;
}

// dense problem arrays from the planned slots
__global__ void rbc_compact(uint64_t n_slots, RbcArgs a, const uint64_t* __restrict__ off, uint32_t* dCluster, uint32_t* dQs, uint32_t* dQe, uint32_t* dTs, uint32_t* dTe,
                            int32_t* dSt, uint8_t* dTwo, uint32_t* dRead, int32_t* dChrom, uint32_t* dLrts, uint32_t* dLrlen) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots || !a.valid[s]) return;
  const uint64_t d = off[s];
  dCluster[d] = a.pCluster[s]; dQs[d] = a.pQs[s]; dQe[d] = a.pQe[s]; dTs[d] = a.pTs[s]; dTe[d] = a.pTe[s]; dSt[d] = a.pSt[s]; dTwo[d] = a.pTwo[s]; dRead[d] = a.pRead[s];
  dChrom[d] = a.pChrom[s]; dLrts[d] = a.pLrts[s]; dLrlen[d] = a.pLrlen[s];
}

// one wave per problem: decision 1 / 3 -> the pairs join the cluster: SetClusterBoundariesFromMatches (Clustering.h:308-322; the box of the
// cluster so far is the box of its matches so far), refinespace = 1, anchorfreq = 1 for decision 3 (:414-419).  Problems are applied in planning
// order; the two problems of one read touch different clusters.
__global__ void __launch_bounds__(64) rbc_apply(int n, const uint32_t* __restrict__ dCluster, const int32_t* __restrict__ dec, const uint64_t* __restrict__ pairOff,
                                                const uint32_t* __restrict__ pq, const uint32_t* __restrict__ pt, const uint32_t* __restrict__ nOld, int K, uint32_t* box,
                                                float* freq, uint8_t* refinespace, uint32_t* nAdded) {
  const int p = blockIdx.x;
  if (p >= n) return;
  const int d = dec[p];
  if (d != 1 && d != 3) return;
  const uint64_t b = pairOff[p], e = pairOff[p + 1];
  const uint32_t cl = dCluster[p];
  if (e == b) {                                  // eff >= reff with both empty (:414-419): no pair joins, the flags are still set
    if (threadIdx.x == 0) { refinespace[cl] = 1; if (d == 3) freq[cl] = 1.0f; }
    return;
  }
  uint32_t qS = 0xFFFFFFFFu, qE = 0, tS = 0xFFFFFFFFu, tE = 0;
  for (uint64_t i = b + threadIdx.x; i < e; i += 64) { qS = min(qS, pq[i]); qE = max(qE, pq[i] + (uint32_t)K); tS = min(tS, pt[i]); tE = max(tE, pt[i] + (uint32_t)K); }
  for (int o = 32; o > 0; o >>= 1) { qS = min(qS, __shfl_xor(qS, o)); qE = max(qE, __shfl_xor(qE, o)); tS = min(tS, __shfl_xor(tS, o)); tE = max(tE, __shfl_xor(tE, o)); }
  if (threadIdx.x == 0) {
    if (nOld[cl] + nAdded[cl] > 0) { qS = min(qS, box[4 * cl]); qE = max(qE, box[4 * cl + 1]); tS = min(tS, box[4 * cl + 2]); tE = max(tE, box[4 * cl + 3]); }
    box[4 * cl] = qS; box[4 * cl + 1] = qE; box[4 * cl + 2] = tS; box[4 * cl + 3] = tE;
    refinespace[cl] = 1;
    if (d == 3) freq[cl] = 1.0f;
    nAdded[cl] += (uint32_t)(e - b);
  }
}

__global__ void rbc_counts(uint64_t n, const uint64_t* __restrict__ off, uint32_t* cnt) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) cnt[i] = (uint32_t)(off[i + 1] - off[i]);
}
struct Seg { uint32_t cluster; uint32_t cnt; uint64_t dstOff; const uint32_t* q; const uint32_t* t; };
__global__ void __launch_bounds__(64) rbc_copy_old(uint64_t ncl, const uint64_t* __restrict__ oldOff, const uint32_t* __restrict__ oq, const uint32_t* __restrict__ ot,
                                                   const uint64_t* __restrict__ newOff, uint32_t* nq, uint32_t* nt) {
  for (uint64_t c = blockIdx.x; c < ncl; c += gridDim.x) {
    const uint64_t s = oldOff[c], n = oldOff[c + 1] - s, d = newOff[c];
    for (uint64_t i = threadIdx.x; i < n; i += 64) { nq[d + i] = oq[s + i]; nt[d + i] = ot[s + i]; }
  }
}
__global__ void __launch_bounds__(64) rbc_copy_segs(int nseg, const Seg* __restrict__ segs, uint32_t* nq, uint32_t* nt) {
  const int s = blockIdx.x;
  if (s >= nseg) return;
  const Seg g = segs[s];
  for (uint32_t i = threadIdx.x; i < g.cnt; i += 64) { nq[g.dstOff + i] = g.q[i]; nt[g.dstOff + i] = g.t[i]; }
}

inline size_t szb(size_t n, size_t e) { return (n * e + 255) / 256 * 256; }

}  // namespace

extern "C" int lra_refine_btwn_clusters_batch(lra_ctx* ctx, int n_reads, const uint64_t* d_read_chain_off, uint64_t n_chains, const uint64_t* d_chain_off, const uint32_t* d_ch,
                                              uint64_t n_clusters, const uint64_t* d_match_off, uint64_t n_matches, const uint32_t* d_mq, const uint32_t* d_mt, uint32_t* d_box,
                                              const int32_t* d_strand, const int32_t* d_chrom, float* d_anchorfreq, const uint64_t* d_read_off, const char* d_strands,
                                              uint64_t rc_base, const char* d_genome, const uint64_t* h_chrom_pos, int n_chrom, int K, int W, int read_type, float anchorstoosparse,
                                              int match, int mismatch, int indel, int max_freq, lra_btwn_clusters_result* out) {
  if (!ctx || !out || !h_chrom_pos || n_chrom < 1 || n_reads < 0) return LRA_ERR_INVALID;
  memset(out, 0, sizeof *out);
  out->n_clusters = n_clusters;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t npos = (size_t)n_chrom + 1, nS = 2 * (size_t)n_reads + 2;
  auto grid = [](uint64_t n) { return dim3((unsigned)((n + 255) / 256)); };
  char* w = (char*)lra_ensure(ctx, 91, szb(npos, 8) + szb(n_reads + 1, 4) * 3 + szb(nS, 4) * 20 + szb(nS, 8) + szb(nS, 1) * 2 + szb(n_clusters + 2, 4) * 3 + szb(n_clusters + 2, 1) +
                                           szb(n_clusters + 2, 8) + 16384);
  if (!w) return LRA_ERR_NOMEM;
  auto take = [&](size_t n, size_t e) { char* r = w; w += szb(n, e); return r; };
  uint64_t* dpos = (uint64_t*)take(npos, 8);
  RbcArgs a; memset(&a, 0, sizeof a);
  a.curChain = (uint32_t*)take(n_reads + 1, 4); a.curStep = (uint32_t*)take(n_reads + 1, 4); a.active = (uint32_t*)take(n_reads + 1, 4);
  a.valid = (uint32_t*)take(nS, 4); a.pCluster = (uint32_t*)take(nS, 4); a.pQs = (uint32_t*)take(nS, 4); a.pQe = (uint32_t*)take(nS, 4); a.pTs = (uint32_t*)take(nS, 4);
  a.pTe = (uint32_t*)take(nS, 4); a.pSt = (int32_t*)take(nS, 4); a.pRead = (uint32_t*)take(nS, 4); a.pChrom = (int32_t*)take(nS, 4); a.pLrts = (uint32_t*)take(nS, 4);
  a.pLrlen = (uint32_t*)take(nS, 4);
  uint32_t* dCluster = (uint32_t*)take(nS, 4); uint32_t* dQs = (uint32_t*)take(nS, 4); uint32_t* dQe = (uint32_t*)take(nS, 4); uint32_t* dTs = (uint32_t*)take(nS, 4);
  uint32_t* dTe = (uint32_t*)take(nS, 4); int32_t* dSt = (int32_t*)take(nS, 4); uint32_t* dRead = (uint32_t*)take(nS, 4); int32_t* dChrom = (int32_t*)take(nS, 4);
  uint32_t* dLrts = (uint32_t*)take(nS, 4); uint32_t* dLrlen = (uint32_t*)take(nS, 4);
  uint64_t* slotOff = (uint64_t*)take(nS, 8);
  a.pTwo = (uint8_t*)take(nS, 1); uint8_t* dTwo = (uint8_t*)take(nS, 1);
  uint32_t* nOld = (uint32_t*)take(n_clusters + 2, 4); uint32_t* nAdded = (uint32_t*)take(n_clusters + 2, 4); uint32_t* newCnt = (uint32_t*)take(n_clusters + 2, 4);
  uint8_t* refinespace = (uint8_t*)take(n_clusters + 2, 1);
  uint64_t* newOff = (uint64_t*)take(n_clusters + 2, 8);
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(dpos, h_chrom_pos, npos * 8, hipMemcpyHostToDevice, st));
  LRA_HIP_CHECK(ctx, hipMemsetAsync(a.curChain, 0, (size_t)(n_reads + 1) * 4, st));
  LRA_HIP_CHECK(ctx, hipMemsetAsync(nAdded, 0, (n_clusters + 2) * 4, st));
  LRA_HIP_CHECK(ctx, hipMemsetAsync(refinespace, 0, n_clusters + 2, st));
  if (n_clusters) hipLaunchKernelGGL(rbc_counts, grid(n_clusters), dim3(256), 0, st, n_clusters, d_match_off, nOld);
  {
    std::vector<uint32_t> ones((size_t)n_reads + 1, 1);
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(a.curStep, ones.data(), (size_t)(n_reads + 1) * 4, hipMemcpyHostToDevice, st));
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  }
  a.n_reads = n_reads; a.readChainOff = d_read_chain_off; a.chainOff = d_chain_off; a.ch = d_ch; a.box = d_box; a.strand = d_strand; a.chrom = d_chrom; a.freq = d_anchorfreq;
  a.refinespace = refinespace; a.read_off = d_read_off; a.pos = dpos;
  a.contig = read_type == LRA_READ_CONTIG; a.low_b = a.contig ? 1000 : 20; a.upper = a.contig ? 100000 : 50000;
  (void)n_chains;
  struct HostSeg { uint32_t cluster, cnt; void* q; void* t; };
  std::vector<HostSeg> segs;
  std::vector<void*> roundBufs;
  auto cleanup = [&]() { for (void* p : roundBufs) (void)hipFree(p); roundBufs.clear(); };
  uint64_t n_problems = 0, n_pairs_kept = 0; uint32_t rounds = 0;
  for (;; rounds++) {
    if (n_reads == 0) break;
    lra_time_begin(ctx, "btwn_clusters_plan");
    hipLaunchKernelGGL(rbc_plan, grid((uint64_t)n_reads), dim3(256), 0, st, a);
    lra_time_end(ctx);
    int rc = lra_exclusive_scan<uint32_t>(ctx, (long)(2 * (size_t)n_reads), a.valid, slotOff);
    if (rc) { cleanup(); return rc; }
    uint64_t np = 0;
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(&np, slotOff + 2 * (size_t)n_reads, 8, hipMemcpyDeviceToHost, st));
    // are there reads with steps left?
    std::vector<uint32_t> act((size_t)n_reads);
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(act.data(), a.active, (size_t)n_reads * 4, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (np == 0) {
      bool any = false;
      for (uint32_t v : act) any |= v != 0;
      if (!any) break;
      continue;
    }
    hipLaunchKernelGGL(rbc_compact, grid(2 * (uint64_t)n_reads), dim3(256), 0, st, 2 * (uint64_t)n_reads, a, (const uint64_t*)slotOff, dCluster, dQs, dQe, dTs, dTe, dSt, dTwo, dRead,
                       dChrom, dLrts, dLrlen);
    lra_btwn_space_result br;
    rc = lra_refine_btwn_space_batch(ctx, (int)np, dQs, dQe, dTs, dTe, dSt, dTwo, dRead, dChrom, dLrts, dLrlen, d_read_off, d_strands, rc_base, d_genome, h_chrom_pos, n_chrom, K, W,
                                     read_type, anchorstoosparse, match, mismatch, indel, max_freq, &br);
    if (rc) { cleanup(); return rc; }
    n_problems += np;
    // keep the round's pairs (the stage's buffers are reused next round)
    void* keep = nullptr;
    const size_t pb = szb(br.n_pairs + 1, 4);
    if (hipMalloc(&keep, 2 * pb + 256) != hipSuccess) { cleanup(); return lra_set_err(ctx, LRA_ERR_NOMEM, "btwn clusters: round buffer"); }
    roundBufs.push_back(keep);
    uint32_t* kq = (uint32_t*)keep; uint32_t* kt = (uint32_t*)((char*)keep + pb);
    if (br.n_pairs) {
      LRA_HIP_CHECK(ctx, hipMemcpyAsync(kq, br.d_pair_q, br.n_pairs * 4, hipMemcpyDeviceToDevice, st));
      LRA_HIP_CHECK(ctx, hipMemcpyAsync(kt, br.d_pair_t, br.n_pairs * 4, hipMemcpyDeviceToDevice, st));
    }
    lra_time_begin(ctx, "btwn_clusters_apply");
    hipLaunchKernelGGL(rbc_apply, dim3((unsigned)np), dim3(64), 0, st, (int)np, (const uint32_t*)dCluster, br.d_decision, br.d_pair_off, (const uint32_t*)kq, (const uint32_t*)kt,
                       (const uint32_t*)nOld, K, d_box, d_anchorfreq, refinespace, nAdded);
    lra_time_end(ctx);
    std::vector<int32_t> hdec(np); std::vector<uint64_t> hoff(np + 1); std::vector<uint32_t> hcl(np);
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(hdec.data(), br.d_decision, np * 4, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(hoff.data(), br.d_pair_off, (np + 1) * 8, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(hcl.data(), dCluster, np * 4, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
    for (uint64_t p = 0; p < np; p++)
      if ((hdec[p] == 1 || hdec[p] == 3) && hoff[p + 1] > hoff[p]) {
        segs.push_back(HostSeg{hcl[p], (uint32_t)(hoff[p + 1] - hoff[p]), kq + hoff[p], kt + hoff[p]});
        n_pairs_kept += hoff[p + 1] - hoff[p];
      }
  }
  // the clusters' final lists: old matches, then the appended segments in order
  int rc = LRA_OK;
  if (n_clusters) {
    // (a few segments per read: the offsets are laid out on the host)
    {
      std::vector<uint32_t> old(n_clusters);
      LRA_HIP_CHECK(ctx, hipMemcpy(old.data(), nOld, n_clusters * 4, hipMemcpyDeviceToHost));
      std::vector<uint32_t> add(n_clusters, 0);
      for (auto& s : segs) add[s.cluster] += s.cnt;
      std::vector<uint64_t> off(n_clusters + 1, 0);
      for (uint64_t c = 0; c < n_clusters; c++) off[c + 1] = off[c] + old[c] + add[c];
      const uint64_t total = off[n_clusters];
      char* wo = (char*)lra_ensure(ctx, 92, szb(total + 1, 4) * 2 + szb(segs.size() + 1, sizeof(Seg)) + 4096);
      if (!wo) { cleanup(); return LRA_ERR_NOMEM; }
      uint32_t* nq = (uint32_t*)wo; uint32_t* nt = (uint32_t*)(wo + szb(total + 1, 4)); Seg* dsegs = (Seg*)(wo + 2 * szb(total + 1, 4));
      LRA_HIP_CHECK(ctx, hipMemcpyAsync(newOff, off.data(), (n_clusters + 1) * 8, hipMemcpyHostToDevice, st));
      std::vector<Seg> hs(segs.size());
      std::vector<uint64_t> cursor(n_clusters);
      for (uint64_t c = 0; c < n_clusters; c++) cursor[c] = off[c] + old[c];
      for (size_t i = 0; i < segs.size(); i++) { hs[i] = Seg{segs[i].cluster, segs[i].cnt, cursor[segs[i].cluster], (const uint32_t*)segs[i].q, (const uint32_t*)segs[i].t}; cursor[segs[i].cluster] += segs[i].cnt; }
      if (!hs.empty()) LRA_HIP_CHECK(ctx, hipMemcpyAsync(dsegs, hs.data(), hs.size() * sizeof(Seg), hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(rbc_copy_old, dim3((unsigned)std::min<uint64_t>(n_clusters, (uint64_t)ctx->num_cu * 32)), dim3(64), 0, st, n_clusters, d_match_off, d_mq, d_mt,
                         (const uint64_t*)newOff, nq, nt);
      if (!hs.empty()) hipLaunchKernelGGL(rbc_copy_segs, dim3((unsigned)hs.size()), dim3(64), 0, st, (int)hs.size(), (const Seg*)dsegs, nq, nt);
      LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
      out->n_matches = total; out->d_match_off = newOff; out->d_q = nq; out->d_t = nt;
    }
  }
  (void)n_matches;
  cleanup();
  LRA_HIP_CHECK(ctx, hipGetLastError());
  out->d_refinespace = refinespace; out->n_problems = n_problems; out->n_rounds = rounds; out->n_pairs_added = n_pairs_kept;
  return rc;
}
