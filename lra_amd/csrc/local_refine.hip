// lra_amd/csrc/local_refine.hip -- SURVEY §8a row a13 (low-accuracy path): LocalRefineAlignment (LocalRefineAlignment.h:885-1029, called at
// Map_lowacc.h:576) and RefinedAlignmentbtwnAnchors (:203-550) for every primary chain of a batch.  gfx950 only.
// The walk over a chain is serial only in what it decides, not in what it computes: whether an inversion is tried on a large space depends on
// how many blocks the current alignment holds (:288), everything else is a function of the anchor pair alone.  So all candidate work is
// done up front in batches, and a last pass (one lane per primary chain) walks the chains, takes the reference's decisions with the block
// counts in hand and copies the precomputed pieces into place:
//   1 classify every consecutive anchor pair: nothing between / direct (RefineByLinearAlignment) / large space (RefineBySDP, both sides >= 300)
//   2 lra_between_anchors_batch on the direct pairs
//   3 lra_refine_space_batch_mf on the large spaces, read strand; 4 again on the other strand where the first result is sparse (:287-289 without
//     the block-count clause: a superset of the spaces the reference tries)
//   5 per space: what happens if the inversion is tried (break / forward seeds / inverted seeds); the seed sets that can be used
//   6 per seed set: DiagonalSort + LinearExtend (strand 0, K of the space), the two boundary anchors, TrimOverlappedAnchors (pair version),
//     SparseDP_ForwardOnly (lra_sparse_dp_batch, single-cluster mode), RemovePairedIndels (pair version), the anchors to keep
//   7 lra_between_anchors_batch on the spaces between those anchors (and on whole spaces that got no seed)
//   8 the walk: count, scan, emit (alignments with their blocks, in the reference's order)
// The +,-,+ / typeofaln = 3 pass at the end of the reference function reads fields CalculateStatistics has not filled yet and never fires.
#include "common.h"
#include "scan.h"
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <vector>

#include <cstdio>
#include <cstdlib>
#define LR_DBG(...) do { if (getenv("LRA_DEBUG")) { (void)hipStreamSynchronize(ctx->stream); fprintf(stderr, __VA_ARGS__); fprintf(stderr, " [%s]\n", hipGetErrorString(hipGetLastError())); } } while (0)

namespace {

constexpr uint32_t NONE = 0xFFFFFFFFu;

struct LrArgs {
  // jobs / chains / anchors
  uint64_t nJobs, nChains, nAnch;
  const uint64_t* jobChainOff; const uint32_t* jobRead; const int32_t* jobH;
  const uint64_t* aOff; const int32_t* cStrand; const int32_t* cChrom; const float* cValue; const int32_t* cN0; const int32_t* cN1;
  const uint32_t* AQ; const uint32_t* AT; const int32_t* AL;
  const uint64_t* read_off; uint64_t rcBase; const uint64_t* pos;
  lra_lra_opts o;
  const uint32_t* jobLSC; int minAnchors;   // high-accuracy overload (:552): LSC given by the caller, chains of one anchor are walked too
  // per chain / anchor
  uint32_t* chainJob; uint32_t* anchChain;
  uint8_t* type; uint32_t* cre; uint32_t* nrs; uint32_t* cge; uint32_t* ngs; uint32_t* isDir; uint32_t* isBig; const uint64_t* dirId; const uint64_t* bigId;
  // direct results
  const uint64_t* d1Off; const int32_t* d1Blk;
  // big spaces
  uint64_t nBig;
  uint32_t* bPair; int32_t* bK; int32_t* bW; int32_t* bMf; int32_t* bDiag; float* bMinRatio; int32_t* bMinDist; int32_t* bSv;
  const uint64_t* fOff; const uint32_t* fQ; const uint32_t* fT; const float* fId;
  uint32_t* needRev; const uint64_t* revId; const uint64_t* rOff; const uint32_t* rQ; const uint32_t* rT; const float* rId;
  uint8_t* outcome;                          // if tried: 0 break, 1 forward seeds, 2 inverted seeds
  uint32_t* needJob;                         // [2 nBig]: seed set F (2b) / R (2b+1) is extended
  const uint64_t* jobId;
  // seed-set jobs
  uint64_t nJ;
  uint32_t* jBig; uint8_t* jSet; uint32_t* jCnt; const uint64_t* jPairOff; uint32_t* jq; uint32_t* jt;
  uint64_t* cStart; uint64_t* cEnd; int* cStr0; int* cChr; int* cRd; int* cK;
  const uint32_t* eCount; const uint32_t* eq; const uint32_t* et; const int* el;
  uint32_t* xCnt; const uint64_t* xOff; uint32_t* xq; uint32_t* xt; int32_t* xl;
  // sparse DP result of the jobs + the kept chain
  const uint64_t* sStart; const uint32_t* sLen; const uint32_t* sAnchor; const float* sValue; const uint32_t* sStatus;
  uint32_t* keptCnt; uint32_t* kept;         // kept[xOff[j] + i]: index into the job's anchors, chain order
  int32_t* jStart; int32_t* jEnd;            // btc_start / btc_end
  uint32_t* pCnt; const uint64_t* pOff;      // AOG problems of the second batch: per job, then one per seedless space
  uint32_t* wholeNeed; const uint64_t* wholeId; uint64_t nJobProb;
  uint64_t* p2QB; uint32_t* p2cre; uint32_t* p2nrs; uint64_t* p2TB; uint32_t* p2cge; uint32_t* p2ngs; uint8_t* jFinal;
  const uint64_t* d2Off; const int32_t* d2Blk;
  // output
  uint32_t* nAln; uint32_t* nBlk; const uint64_t* alnOff; const uint64_t* blkOff;
  int32_t* oStrand; int32_t* oSupp; int32_t* oSec; int32_t* oN0; int32_t* oN1; int32_t* oChrom; float* oValue; uint64_t* oBlockOff; int32_t* oBlocks; uint32_t* status;
};

__global__ void lr_maps(LrArgs a) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < a.nJobs) for (uint64_t c = a.jobChainOff[i]; c < a.jobChainOff[i + 1]; c++) a.chainJob[c] = (uint32_t)i;
  if (i < a.nChains) for (uint64_t p = a.aOff[i]; p < a.aOff[i + 1]; p++) a.anchChain[p] = (uint32_t)i;
}

// 1: RefinedAlignmentbtwnAnchors :209-236 for the pair whose `cur` is anchor p
__global__ void lr_classify(LrArgs a) {
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.nAnch) return;
  a.type[p] = 0; a.isDir[p] = 0; a.isBig[p] = 0;
  const uint32_t c = a.anchChain[p];
  const uint64_t a0 = a.aOff[c];
  const int m = (int)(a.aOff[c + 1] - a0), i = (int)(p - a0), str = a.cStrand[c] != 0;
  if (m <= 1) return;
  int nx;
  if (str == 0) { if (i < 1) return; nx = i - 1; } else { if (i > m - 2) return; nx = i + 1; }
  const uint32_t r = a.jobRead[a.chainJob[c]];
  const uint32_t L = (uint32_t)(a.read_off[r + 1] - a.read_off[r]);
  const uint32_t q = a.AQ[p], t = a.AT[p]; const int len = a.AL[p];
  const uint32_t qn = a.AQ[a0 + nx], tn = a.AT[a0 + nx]; const int ln = a.AL[a0 + nx];
  uint32_t cre, nrs, cge, ngs;
  if (str == 0) { cre = q + len; nrs = qn; cge = t + len; ngs = tn; }
  else { cre = L - q; nrs = L - qn - ln; cge = t + len; ngs = tn; }
  a.cre[p] = cre; a.nrs[p] = nrs; a.cge[p] = cge; a.ngs[p] = ngs;
  if (!(cge <= ngs)) { a.type[p] = 3; return; }
  const long long rd = (long long)(uint32_t)(nrs - cre), gd = (long long)(uint32_t)(ngs - cge);
  if (a.o.refineBySDP && min(rd, gd) >= 300) { a.type[p] = 2; a.isBig[p] = 1; }
  else { a.type[p] = 1; a.isDir[p] = 1; }
}

// problems of the first lra_between_anchors_batch
__global__ void lr_direct_gather(LrArgs a, uint64_t* qB, uint32_t* cre, uint32_t* nrs, uint64_t* tB, uint32_t* cge, uint32_t* ngs) {
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.nAnch || !a.isDir[p]) return;
  const uint64_t d = a.dirId[p];
  const uint32_t c = a.anchChain[p];
  const uint32_t r = a.jobRead[a.chainJob[c]];
  qB[d] = (a.cStrand[c] ? a.rcBase : 0) + a.read_off[r]; cre[d] = a.cre[p]; nrs[d] = a.nrs[p];
  tB[d] = a.pos[a.cChrom[c]]; cge[d] = a.cge[p]; ngs[d] = a.ngs[p];
}

// 3: the RefineSpace problem of a large space (:239-283); rev = the other strand (:290-294)
__global__ void lr_big_gather(LrArgs a, int rev, uint64_t* qOff, int32_t* qLen, uint64_t* tOff, int32_t* tLen, uint32_t* tSpan, int32_t* K, int32_t* W, int32_t* diag,
                              uint32_t* qAdd, uint32_t* tAdd, uint32_t* flip, int32_t* mf) {
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.nAnch || !a.isBig[p]) return;
  const uint64_t b = a.bigId[p];
  const uint32_t c = a.anchChain[p];
  const uint32_t r = a.jobRead[a.chainJob[c]];
  const uint32_t L = (uint32_t)(a.read_off[r + 1] - a.read_off[r]);
  const int str = a.cStrand[c] != 0;
  uint32_t cre = a.cre[p], nrs = a.nrs[p];
  const uint32_t cge = a.cge[p], ngs = a.ngs[p];
  uint64_t i;
  int st = str;
  if (!rev) {
    const long long rd = (long long)(uint32_t)(nrs - cre), gd = (long long)(uint32_t)(ngs - cge);
    const int sv = (int)(max(rd, gd) - min(rd, gd));
    int d;
    if (!a.o.isOnt) d = min((int)floorf(fmaxf(80.f, 0.01f * (float)rd)), 500);
    else d = min((int)floorf(fmaxf(100.f, 0.15f * (float)rd)), 2000);
    d = max(2 * sv, d);
    int k, w, f = a.o.localMaxFreq; float mr;
    if (max(rd, gd) < 100) { k = 6; w = 5; mr = (float)(0.5 / 29.5); }
    else if (max(rd, gd) < 500) { k = 9; w = 7; f = 50; mr = (float)(0.5 / 69.1); }
    else { k = 12; w = 7; mr = (float)(0.5 / 140.2); }
    a.bPair[b] = (uint32_t)p; a.bK[b] = k; a.bW[b] = w; a.bMf[b] = f; a.bDiag[b] = d; a.bMinRatio[b] = mr; a.bMinDist[b] = (int)min(rd, gd); a.bSv[b] = sv;
    i = b;
    W[i] = w;
  } else {
    if (!a.needRev[b]) return;
    i = a.revId[b];
    const uint32_t t = cre; cre = L - nrs; nrs = L - t;
    st = !str;
    W[i] = a.o.globalW;
  }
  qOff[i] = (st ? a.rcBase : 0) + a.read_off[r] + cre; qLen[i] = (int32_t)(nrs - cre);
  tOff[i] = a.pos[a.cChrom[c]] + cge; tLen[i] = (int32_t)(ngs - cge); tSpan[i] = ngs - cge;
  K[i] = a.bK[b]; diag[i] = a.bDiag[b]; qAdd[i] = cre; tAdd[i] = cge; flip[i] = 0; mf[i] = a.bMf[b];
}

__global__ void lr_need_rev(LrArgs a) {                                   // :287-289 without the block-count clause
  const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.nBig) return;
  const uint32_t nf = (uint32_t)(a.fOff[b + 1] - a.fOff[b]);
  a.needRev[b] = ((float)nf / (float)a.bMinDist[b]) < a.bMinRatio[b] && (double)a.fId[b] < 0.8;
}

// 5: :295-330 as a function of the two seed counts
__global__ void lr_decide(LrArgs a) {
  const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.nBig) return;
  const uint32_t nf = (uint32_t)(a.fOff[b + 1] - a.fOff[b]);
  uint8_t oc = 1;
  uint32_t nr = 0;
  if (a.needRev[b]) {
    const uint64_t v = a.revId[b];
    nr = (uint32_t)(a.rOff[v + 1] - a.rOff[v]);
    const int minDist = a.bMinDist[b];
    const double driftRate = a.o.isOnt ? (double)0.10f : (double)0.01f;
    if (nf == 0 && nr == 0 && minDist > 500 && (double)a.bSv[b] <= fmax(50.0, minDist * driftRate)) oc = 0;
    else if ((double)a.rId[v] < 0.8 && ((float)nr / (float)minDist) < a.bMinRatio[b]) oc = 0;
    else if (nf >= nr) oc = 1;
    else oc = 2;
  }
  a.outcome[b] = oc;
  a.needJob[2 * b] = nf > 0; a.needJob[2 * b + 1] = (a.needRev[b] && oc == 2);
}

__global__ void lr_jobs(LrArgs a) {
  const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= 2 * a.nBig || !a.needJob[x]) return;
  const uint64_t j = a.jobId[x], b = x >> 1;
  a.jBig[j] = (uint32_t)b; a.jSet[j] = (uint8_t)(x & 1);
  a.jCnt[j] = (x & 1) ? (uint32_t)(a.rOff[a.revId[b] + 1] - a.rOff[a.revId[b]]) : (uint32_t)(a.fOff[b + 1] - a.fOff[b]);
}

// 6a: the seeds of every job, as clusters for the LinearExtend kernel (strand 0, the forward read, K of the space), with DiagonalSort keys
__global__ void __launch_bounds__(64) lr_job_pairs(LrArgs a, uint64_t* key, uint32_t* val) {
  for (uint64_t j = blockIdx.x; j < a.nJ; j += gridDim.x) {
    const uint64_t b = a.jBig[j];
    const int set = a.jSet[j];
    const uint64_t s0 = set ? a.rOff[a.revId[b]] : a.fOff[b];
    const uint32_t* SQ = set ? a.rQ : a.fQ; const uint32_t* ST = set ? a.rT : a.fT;
    const uint64_t o = a.jPairOff[j];
    const uint32_t n = a.jCnt[j];
    for (uint32_t i = threadIdx.x; i < n; i += 64) {
      const uint32_t q = SQ[s0 + i], t = ST[s0 + i];
      a.jq[o + i] = q; a.jt[o + i] = t;
      key[o + i] = ((uint64_t)((long long)q - (long long)t + (1LL << 32)) << 31) | q;
      val[o + i] = (uint32_t)(o + i);
    }
    if (threadIdx.x == 0) {
      const uint32_t c = a.anchChain[a.bPair[b]];
      a.cStart[j] = o; a.cEnd[j] = o + n; a.cStr0[j] = 0; a.cChr[j] = a.cChrom[c]; a.cRd[j] = (int)a.jobRead[a.chainJob[c]]; a.cK[j] = a.bK[b];
    }
  }
}

__global__ void lr_sorted(uint64_t n, const uint32_t* __restrict__ val, const uint32_t* __restrict__ jq, const uint32_t* __restrict__ jt, const uint64_t* __restrict__ key,
                          uint32_t* sq, uint32_t* st) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  (void)key;
  if (i < n) { const uint32_t v = val[i]; sq[i] = jq[v]; st[i] = jt[v]; }
}
__global__ void __launch_bounds__(64) lr_add_coff(LrArgs a, uint32_t* st) {   // the LinearExtend kernel takes genome-wide t
  for (uint64_t j = blockIdx.x; j < a.nJ; j += gridDim.x) {
    const uint32_t coff = (uint32_t)a.pos[a.cChr[j]];
    for (uint64_t i = a.cStart[j] + threadIdx.x; i < a.cEnd[j]; i += 64) st[i] += coff;
  }
}

__global__ void lr_ext_count(LrArgs a) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < a.nJ) a.xCnt[j] = a.eCount[j] + (a.jSet[j] ? 0 : 2);
}

// 6b: the job's anchors: the extended seeds, then (read-strand seeds only) the two boundary anchors :362-367
__global__ void __launch_bounds__(64) lr_ext_gather(LrArgs a) {
  for (uint64_t j = blockIdx.x; j < a.nJ; j += gridDim.x) {
    const uint64_t o = a.xOff[j], s = a.cStart[j];
    const uint32_t n = a.eCount[j];
    const uint32_t coff = (uint32_t)a.pos[a.cChr[j]];
    for (uint32_t i = threadIdx.x; i < n; i += 64) { a.xq[o + i] = a.eq[s + i]; a.xt[o + i] = a.et[s + i] - coff; a.xl[o + i] = a.el[s + i]; }
    if (threadIdx.x == 0 && !a.jSet[j]) {
      const uint64_t p = a.bPair[a.jBig[j]];
      const uint32_t c = a.anchChain[p];
      const uint64_t nx = a.cStrand[c] ? p + 1 : p - 1;
      a.xq[o + n] = a.nrs[p]; a.xt[o + n] = a.ngs[p]; a.xl[o + n] = a.AL[nx];
      a.xq[o + n + 1] = a.cre[p] - (uint32_t)a.AL[p]; a.xt[o + n + 1] = a.cge[p] - (uint32_t)a.AL[p]; a.xl[o + n + 1] = a.AL[p];
    }
  }
}

// 6c + 7a: RemovePairedIndels (pair version, Chain.h:753-811) on the job's chain, btc_start / btc_end (:468-472), the AOG problems of the
// walk :482-497.  One lane per job; PASS 0 counts problems, PASS 1 writes them.
template <int PASS>
__global__ void lr_inner_plan(LrArgs a) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= a.nJ) return;
  const uint64_t o = a.xOff[j];
  const uint32_t ne = a.xCnt[j];
  const uint32_t* XQ = a.xq + o; const uint32_t* XT = a.xt + o; const int32_t* XL = a.xl + o;
  uint32_t* kept = a.kept + o;
  const uint64_t b = a.jBig[j];
  const uint64_t p = a.bPair[b];
  const uint32_t c = a.anchChain[p];
  const uint32_t r = a.jobRead[a.chainJob[c]];
  const uint32_t L = (uint32_t)(a.read_off[r + 1] - a.read_off[r]);
  uint32_t cre = a.cre[p], nrs = a.nrs[p];
  const uint32_t cge = a.cge[p], ngs = a.ngs[p];
  if (a.jSet[j]) { const uint32_t t = cre; cre = L - nrs; nrs = L - t; }  // the inverted walk keeps the swapped read coordinates (:291-293)
  if (PASS == 0) {
    const uint32_t m = a.sStatus[j] ? 0 : a.sLen[j];
    const uint32_t* ch = a.sAnchor + a.sStart[j];
    // remove[] lives in kept[] while it is decided (1 = remove), then kept[] is compacted
    for (uint32_t i = 0; i < m; i++) kept[i] = 0;
    int pSV = 0, pPos = -1, pG = 0; bool have = false;
    for (uint32_t cI = 1; cI < m; cI++) {
      const uint32_t x1 = ch[cI], x0 = ch[cI - 1];
      const int Gap = (int)(((long long)XT[x1] - (long long)XQ[x1]) - ((long long)XT[x0] - (long long)XQ[x0]));
      if (abs(Gap) <= 30) continue;
      const int sv = Gap, g = (int)XT[x1];
      if (have) {
        const int blink = max(abs(sv), abs(pSV));
        const bool ps = sv >= 0, differ = (sv >= 0) != (pSV >= 0);
        const int lim = max(2 * blink, 1000);
        const int dIns = abs(g - pG), dDel = abs(g - sv - pG);
        bool hit = false;
        if (differ && abs(sv + pSV) < 600 && abs(sv) != 0 && pSV != 0) hit = (ps && dIns < lim) || (!ps && dDel < lim);
        else if (differ && sv != 0 && pSV != 0 && ((ps && dIns < 500) || (!ps && dDel < 500))) hit = true;
        else if (!differ && sv != 0 && pSV != 0) hit = (ps && dIns < lim) || (!ps && dDel < lim);
        if (hit) for (int i = pPos; i < (int)cI; i++) if (XL[ch[i]] < 100) kept[i] = 1;
      }
      pSV = sv; pPos = (int)cI; pG = g; have = true;
    }
    uint32_t k = 0;
    for (uint32_t i = 0; i < m; i++) { const bool rm = kept[i] != 0; const uint32_t x = ch[i]; if (!rm) kept[k++] = x; }   // k <= i: in place
    a.keptCnt[j] = k;
    int bs = 0, be = (int)k - 1;
    if (k > 0) { if (kept[k - 1] == ne - 1) be = (int)k - 2; if (kept[0] == ne - 2) bs = 1; }
    a.jStart[j] = bs; a.jEnd[j] = be;
    uint32_t np = be >= bs ? (uint32_t)(be - bs + 1) : 0;
    uint32_t ce = cre, ge = cge;
    for (int btc = be; btc >= bs; btc--) { const uint32_t x = kept[btc]; ce = XQ[x] + (uint32_t)XL[x]; ge = XT[x] + (uint32_t)XL[x]; }
    const bool fin = k > 0 && ngs > ge && nrs > ce;
    a.jFinal[j] = fin;
    a.pCnt[j] = k > 0 ? np + (fin ? 1 : 0) : 0;
    return;
  }
  const uint32_t k = a.keptCnt[j];
  if (k == 0) return;
  uint64_t w = a.pOff[j];
  const uint64_t qB = (a.cStrand[c] ? a.rcBase : 0) + a.read_off[r], tB = a.pos[a.cChrom[c]];   // RefineByLinearAlignment(..., str, ...) whatever the seeds' strand
  uint32_t ce = cre, ge = cge;
  for (int btc = a.jEnd[j]; btc >= a.jStart[j]; btc--) {
    const uint32_t x = kept[btc];
    a.p2QB[w] = qB; a.p2cre[w] = ce; a.p2nrs[w] = XQ[x]; a.p2TB[w] = tB; a.p2cge[w] = ge; a.p2ngs[w] = XT[x]; w++;
    ce = XQ[x] + (uint32_t)XL[x]; ge = XT[x] + (uint32_t)XL[x];
  }
  if (a.jFinal[j]) { a.p2QB[w] = qB; a.p2cre[w] = ce; a.p2nrs[w] = nrs; a.p2TB[w] = tB; a.p2cge[w] = ge; a.p2ngs[w] = ngs; }
}

// spaces whose read-strand search found nothing are aligned whole (:499-502)
__global__ void lr_whole(LrArgs a, int pass) {
  const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.nBig) return;
  const bool need = a.fOff[b + 1] == a.fOff[b];
  if (pass == 0) { a.wholeNeed[b] = need; return; }
  if (!need) return;
  const uint64_t w = a.nJobProb + a.wholeId[b];
  const uint64_t p = a.bPair[b];
  const uint32_t c = a.anchChain[p];
  const uint32_t r = a.jobRead[a.chainJob[c]];
  a.p2QB[w] = (a.cStrand[c] ? a.rcBase : 0) + a.read_off[r]; a.p2cre[w] = a.cre[p]; a.p2nrs[w] = a.nrs[p];
  a.p2TB[w] = a.pos[a.cChrom[c]]; a.p2cge[w] = a.cge[p]; a.p2ngs[w] = a.ngs[p];
}

// 8: LocalRefineAlignment :885-996 for one primary chain per lane.  PASS 0 counts alignments and blocks, PASS 1 writes them.
template <int PASS>
__global__ void lr_walk(LrArgs a) {
  const uint64_t job = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (job >= a.nJobs) return;
  const uint64_t c0 = a.jobChainOff[job], c1 = a.jobChainOff[job + 1];
  const uint32_t r = a.jobRead[job];
  const uint32_t L = (uint32_t)(a.read_off[r + 1] - a.read_off[r]);
  const int h = a.jobH[job];
  uint64_t LSC = 0;                                                       // LargestSplitChain Chain.h:963-971
  if (a.jobLSC) LSC = a.jobLSC[job];
  else for (uint64_t c = c0 + 1; c < c1; c++) if (a.aOff[c + 1] - a.aOff[c] > a.aOff[c0 + LSC + 1] - a.aOff[c0 + LSC]) LSC = c - c0;
  uint32_t nAln = 0; uint64_t nBlk = 0;
  uint64_t ai = PASS ? a.alnOff[job] : 0, bi = PASS ? a.blkOff[job] : 0;   // next alignment / block slot
  uint64_t curBlocks = 0;                                                 // blocks of the alignment being built
  auto open = [&](int strand, int supp, int sec, int n0, int n1, int chrom, float value) {
    if (PASS) { a.oStrand[ai] = strand; a.oSupp[ai] = supp; a.oSec[ai] = sec; a.oN0[ai] = n0; a.oN1[ai] = n1; a.oChrom[ai] = chrom; a.oValue[ai] = value; a.oBlockOff[ai] = bi; }
    ai++; nAln++; curBlocks = 0;
  };
  auto blk = [&](int q, int t, int l) { if (PASS) { a.oBlocks[3 * bi] = q; a.oBlocks[3 * bi + 1] = t; a.oBlocks[3 * bi + 2] = l; } bi++; nBlk++; curBlocks++; };
  auto copy = [&](const uint64_t* off, const int32_t* B, uint64_t id) {
    for (uint64_t k = off[id]; k < off[id + 1]; k++) blk(B[3 * k], B[3 * k + 1], B[3 * k + 2]);
  };
  auto run_job = [&](uint64_t j) {                                        // :468-497 with the precomputed pieces
    const uint32_t* kept = a.kept + a.xOff[j];
    const uint32_t* XQ = a.xq + a.xOff[j]; const uint32_t* XT = a.xt + a.xOff[j]; const int32_t* XL = a.xl + a.xOff[j];
    uint64_t w = a.pOff[j];
    for (int btc = a.jEnd[j]; btc >= a.jStart[j]; btc--) {
      copy(a.d2Off, a.d2Blk, w); w++;
      const uint32_t x = kept[btc];
      blk((int)XQ[x], (int)XT[x], XL[x]);
    }
    if (a.jFinal[j]) copy(a.d2Off, a.d2Blk, w);
  };
  for (uint64_t c = c0; c < c1; c++) {
    const uint64_t a0 = a.aOff[c];
    const int m = (int)(a.aOff[c + 1] - a0);
    if (m < a.minAnchors) continue;
    const int str = a.cStrand[c] != 0, inv_str = !str, chrom = a.cChrom[c];
    const int n0 = a.cN0[c];
    const float val = a.cValue[c];
    open(str, (c - c0) != LSC ? 1 : 0, h > 0 ? 1 : 0, n0, a.cN1[c], chrom, val);
    int last = str ? 0 : m - 1;
    for (int step = 0; step < m - 1; step++) {
      const int fl = str ? step : m - 1 - step;
      const uint64_t p = a0 + fl;
      if (str == 0) blk((int)a.AQ[p], (int)a.AT[p], a.AL[p]); else blk((int)(L - a.AQ[p] - (uint32_t)a.AL[p]), (int)a.AT[p], a.AL[p]);
      const int ty = a.type[p];
      bool inversion = false, brk = false;
      if (ty == 1) copy(a.d1Off, a.d1Blk, a.dirId[p]);
      else if (ty == 2) {
        const uint64_t b = a.bigId[p];
        const bool tried = a.needRev[b] && curBlocks >= 5;
        const int oc = tried ? a.outcome[b] : 1;
        if (oc == 0) brk = true;
        else if (oc == 1) {
          if (a.needJob[2 * b]) { const uint64_t j = a.jobId[2 * b]; if (a.keptCnt[j] == 0) { if (PASS == 0) a.status[job] = LRA_ST_OOB_SLOT; } else run_job(j); }
          else copy(a.d2Off, a.d2Blk, a.nJobProb + a.wholeId[b]);
        } else {
          const uint64_t j = a.jobId[2 * b + 1];
          inversion = true;
          if (a.keptCnt[j] == 0) { if (PASS == 0) a.status[job] = LRA_ST_OOB_SLOT; }
          open(inv_str, 1, 0, (int)a.keptCnt[j], (int)a.keptCnt[j], chrom, a.sValue[j]);    // inv_alignment :474-480
          if (a.keptCnt[j]) run_job(j);
        }
      }
      if (inversion || brk) {                                             // :911-929 / :951-969
        const uint64_t cur = ai - 1;
        const int n1 = str ? fl - last : last - fl;
        if (PASS) { a.oStrand[cur] = inversion ? inv_str : str; a.oN0[cur] = n0; a.oN1[cur] = n1; }
        open(str, 1, 0, n0, 0, chrom, val);
        last = fl;
      }
    }
    const int flEnd = str ? m - 1 : 0;
    if (PASS) { const uint64_t cur = ai - 1; a.oN0[cur] = n0; a.oN1[cur] = str ? flEnd - last : last - flEnd; a.oStrand[cur] = str; }
    const uint64_t pe = a0 + flEnd;
    if (str == 0) blk((int)a.AQ[pe], (int)a.AT[pe], a.AL[pe]); else blk((int)(L - a.AQ[pe] - (uint32_t)a.AL[pe]), (int)a.AT[pe], a.AL[pe]);
  }
  if (PASS == 0) { a.nAln[job] = nAln; a.nBlk[job] = (uint32_t)nBlk; }
}

__global__ void lr_iota(uint64_t n, uint64_t* o, int32_t* z) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= n) { o[i] = i; z[i] = 0; }
}

inline size_t sz(size_t n, size_t e) { return (n * e + 255) / 256 * 256; }

struct Carver {
  char* p;
  template <typename T> T* take(size_t n) { T* r = (T*)p; p += sz(n, sizeof(T)); return r; }
};

}  // namespace

static int local_refine_impl(lra_ctx* ctx, uint64_t n_jobs, const uint64_t* d_job_chain_off, const uint32_t* d_job_read, const int32_t* d_job_h,
                             uint64_t n_chains, const uint64_t* d_chain_anchor_off, const int32_t* d_chain_strand, const int32_t* d_chain_chrom,
                             const float* d_chain_value, const int32_t* d_chain_n0, const int32_t* d_chain_n1, uint64_t n_anchors, const uint32_t* d_q,
                             const uint32_t* d_t, const int32_t* d_len, const uint64_t* d_read_off, const char* d_strands, uint64_t rc_base,
                             const char* d_genome, const uint64_t* h_chrom_pos, int n_chrom, const lra_lra_opts* opts, const uint32_t* d_job_lsc, int min_anchors,
                             lra_alignments_result* out);
extern "C" int lra_local_refine_batch(lra_ctx* ctx, uint64_t n_jobs, const uint64_t* d_job_chain_off, const uint32_t* d_job_read, const int32_t* d_job_h,
                                      uint64_t n_chains, const uint64_t* d_chain_anchor_off, const int32_t* d_chain_strand, const int32_t* d_chain_chrom,
                                      const float* d_chain_value, const int32_t* d_chain_n0, const int32_t* d_chain_n1, uint64_t n_anchors, const uint32_t* d_q,
                                      const uint32_t* d_t, const int32_t* d_len, const uint64_t* d_read_off, const char* d_strands, uint64_t rc_base,
                                      const char* d_genome, const uint64_t* h_chrom_pos, int n_chrom, const lra_lra_opts* opts, lra_alignments_result* out) {
  return local_refine_impl(ctx, n_jobs, d_job_chain_off, d_job_read, d_job_h, n_chains, d_chain_anchor_off, d_chain_strand, d_chain_chrom, d_chain_value, d_chain_n0,
                           d_chain_n1, n_anchors, d_q, d_t, d_len, d_read_off, d_strands, rc_base, d_genome, h_chrom_pos, n_chrom, opts, nullptr, 2, out);
}
// The walk of the high-accuracy overload (LocalRefineAlignment.h:577-766): the same, except that a chain of ONE anchor still makes an alignment
// (`if (ultimatechain.size() == 0) continue`, :579), and Supplymentary is `st != LSC` with the caller's LSC = LargestSplitChain_dist (Map_highacc.h:707).
extern "C" int lra_local_refine_highacc_batch(lra_ctx* ctx, uint64_t n_jobs, const uint64_t* d_job_chain_off, const uint32_t* d_job_read, const int32_t* d_job_h,
                                              const uint32_t* d_job_lsc, uint64_t n_chains, const uint64_t* d_chain_anchor_off, const int32_t* d_chain_strand,
                                              const int32_t* d_chain_chrom, const float* d_chain_value, const int32_t* d_chain_n0, const int32_t* d_chain_n1,
                                              uint64_t n_anchors, const uint32_t* d_q, const uint32_t* d_t, const int32_t* d_len, const uint64_t* d_read_off,
                                              const char* d_strands, uint64_t rc_base, const char* d_genome, const uint64_t* h_chrom_pos, int n_chrom,
                                              const lra_lra_opts* opts, lra_alignments_result* out) {
  if (!d_job_lsc) return LRA_ERR_INVALID;
  return local_refine_impl(ctx, n_jobs, d_job_chain_off, d_job_read, d_job_h, n_chains, d_chain_anchor_off, d_chain_strand, d_chain_chrom, d_chain_value, d_chain_n0,
                           d_chain_n1, n_anchors, d_q, d_t, d_len, d_read_off, d_strands, rc_base, d_genome, h_chrom_pos, n_chrom, opts, d_job_lsc, 1, out);
}
static int local_refine_impl(lra_ctx* ctx, uint64_t n_jobs, const uint64_t* d_job_chain_off, const uint32_t* d_job_read, const int32_t* d_job_h,
                             uint64_t n_chains, const uint64_t* d_chain_anchor_off, const int32_t* d_chain_strand, const int32_t* d_chain_chrom,
                             const float* d_chain_value, const int32_t* d_chain_n0, const int32_t* d_chain_n1, uint64_t n_anchors, const uint32_t* d_q,
                             const uint32_t* d_t, const int32_t* d_len, const uint64_t* d_read_off, const char* d_strands, uint64_t rc_base,
                             const char* d_genome, const uint64_t* h_chrom_pos, int n_chrom, const lra_lra_opts* opts, const uint32_t* d_job_lsc, int min_anchors,
                             lra_alignments_result* out) {
  if (!ctx || !out || !opts || !h_chrom_pos || n_chrom < 1) return LRA_ERR_INVALID;
  memset(out, 0, sizeof *out);
  out->n_jobs = n_jobs;
  if (n_jobs == 0) return LRA_OK;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const uint64_t NJ0 = n_jobs, NC = n_chains, NA = n_anchors;
  const size_t npos = (size_t)n_chrom + 1;
  auto grid = [](uint64_t n) { return dim3((unsigned)((n + 255) / 256)); };
  LrArgs a;
  memset(&a, 0, sizeof a);
  a.nJobs = NJ0; a.nChains = NC; a.nAnch = NA; a.jobChainOff = d_job_chain_off; a.jobRead = d_job_read; a.jobH = d_job_h; a.aOff = d_chain_anchor_off;
  a.cStrand = d_chain_strand; a.cChrom = d_chain_chrom; a.cValue = d_chain_value; a.cN0 = d_chain_n0; a.cN1 = d_chain_n1; a.AQ = d_q; a.AT = d_t; a.AL = d_len;
  a.read_off = d_read_off; a.rcBase = rc_base; a.o = *opts; a.jobLSC = d_job_lsc; a.minAnchors = min_anchors;
  // ---- per chain / anchor state
  char* w0 = (char*)lra_ensure(ctx, 40, sz(npos, 8) + sz(NC + 1, 4) + sz(NA + 1, 4) * 7 + sz(NA + 1, 1) + sz(NA + 2, 8) * 2 + sz(NJ0 + 2, 4) * 3 + sz(NJ0 + 2, 8) * 2 + 4096);
  if (!w0) return LRA_ERR_NOMEM;
  Carver cw{w0};
  uint64_t* dpos = cw.take<uint64_t>(npos);
  a.pos = dpos;
  a.chainJob = cw.take<uint32_t>(NC + 1); a.anchChain = cw.take<uint32_t>(NA + 1);
  a.cre = cw.take<uint32_t>(NA + 1); a.nrs = cw.take<uint32_t>(NA + 1); a.cge = cw.take<uint32_t>(NA + 1); a.ngs = cw.take<uint32_t>(NA + 1);
  a.isDir = cw.take<uint32_t>(NA + 1); a.isBig = cw.take<uint32_t>(NA + 1); a.type = cw.take<uint8_t>(NA + 1);
  uint64_t* dirId = cw.take<uint64_t>(NA + 2); uint64_t* bigId = cw.take<uint64_t>(NA + 2);
  a.dirId = dirId; a.bigId = bigId;
  a.nAln = cw.take<uint32_t>(NJ0 + 2); a.nBlk = cw.take<uint32_t>(NJ0 + 2); a.status = cw.take<uint32_t>(NJ0 + 2);
  uint64_t* alnOff = cw.take<uint64_t>(NJ0 + 2); uint64_t* blkOff = cw.take<uint64_t>(NJ0 + 2);
  a.alnOff = alnOff; a.blkOff = blkOff;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(dpos, h_chrom_pos, npos * 8, hipMemcpyHostToDevice, st));
  LRA_HIP_CHECK(ctx, hipMemsetAsync(a.status, 0, (NJ0 + 2) * 4, st));
  lra_time_begin(ctx, "local_refine");
  hipLaunchKernelGGL(lr_maps, grid(std::max(NJ0, NC)), dim3(256), 0, st, a);
  if (NA) hipLaunchKernelGGL(lr_classify, grid(NA), dim3(256), 0, st, a);
  lra_time_end(ctx);
  { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)NA, a.isDir, dirId); if (rc) return rc; }
  { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)NA, a.isBig, bigId); if (rc) return rc; }
  uint64_t nDir = 0, nBig = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&nDir, dirId + NA, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&nBig, bigId + NA, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  LR_DBG("classified: NA %llu nDir %llu nBig %llu", (unsigned long long)NA, (unsigned long long)nDir, (unsigned long long)nBig);
  a.nBig = nBig; out->n_big = nBig;
  // ---- 2: direct pairs
  uint64_t d1Blocks = 0;
  {
    char* wd = (char*)lra_ensure(ctx, 41, sz(nDir + 1, 8) * 3 + sz(nDir + 1, 4) * 4 + 4096);
    if (!wd) return LRA_ERR_NOMEM;
    Carver c{wd};
    uint64_t* qB = c.take<uint64_t>(nDir + 1); uint64_t* tB = c.take<uint64_t>(nDir + 1); uint64_t* d1Off = c.take<uint64_t>(nDir + 2);
    uint32_t* cre = c.take<uint32_t>(nDir + 1); uint32_t* nrs = c.take<uint32_t>(nDir + 1); uint32_t* cge = c.take<uint32_t>(nDir + 1); uint32_t* ngs = c.take<uint32_t>(nDir + 1);
    a.d1Off = d1Off;
    if (nDir) {
      hipLaunchKernelGGL(lr_direct_gather, grid(NA), dim3(256), 0, st, a, qB, cre, nrs, tB, cge, ngs);
      lra_between_result br;
      { int rc = lra_between_anchors_batch(ctx, (int)nDir, d_strands, qB, cre, nrs, d_genome, tB, cge, ngs, opts->match, opts->mismatch, opts->indel, opts->localBand, 1, &br);
        if (rc) return rc; }
      d1Blocks = br.n_blocks;
      int32_t* d1Blk = (int32_t*)lra_ensure(ctx, 42, (3 * d1Blocks + 3) * 4 + 256);
      if (!d1Blk) return LRA_ERR_NOMEM;
      LRA_HIP_CHECK(ctx, hipMemcpyAsync(d1Off, br.d_block_off, (nDir + 1) * 8, hipMemcpyDeviceToDevice, st));
      LRA_HIP_CHECK(ctx, hipMemcpyAsync(d1Blk, br.d_blocks, 3 * d1Blocks * 4, hipMemcpyDeviceToDevice, st));
      a.d1Blk = d1Blk;
    } else LRA_HIP_CHECK(ctx, hipMemsetAsync(d1Off, 0, 16, st));
  }
  // ---- 3-5: large spaces
  char* wb = (char*)lra_ensure(ctx, 43, sz(nBig + 2, 4) * 12 + sz(2 * nBig + 2, 4) + sz(nBig + 2, 8) * 8 + sz(2 * nBig + 2, 8) + sz(nBig + 2, 1) + sz(nBig + 2, 4) * 8 + 4096);
  if (!wb) return LRA_ERR_NOMEM;
  Carver cb{wb};
  a.bPair = cb.take<uint32_t>(nBig + 2); a.bK = cb.take<int32_t>(nBig + 2); a.bW = cb.take<int32_t>(nBig + 2); a.bMf = cb.take<int32_t>(nBig + 2);
  a.bDiag = cb.take<int32_t>(nBig + 2); a.bMinRatio = cb.take<float>(nBig + 2); a.bMinDist = cb.take<int32_t>(nBig + 2); a.bSv = cb.take<int32_t>(nBig + 2);
  a.needRev = cb.take<uint32_t>(nBig + 2); a.wholeNeed = cb.take<uint32_t>(nBig + 2); float* fId = cb.take<float>(nBig + 2); float* rId = cb.take<float>(nBig + 2);
  a.needJob = cb.take<uint32_t>(2 * nBig + 2);
  uint64_t* fOff = cb.take<uint64_t>(nBig + 2); uint64_t* rOff = cb.take<uint64_t>(nBig + 2); uint64_t* revId = cb.take<uint64_t>(nBig + 2); uint64_t* wholeId = cb.take<uint64_t>(nBig + 2);
  uint64_t* pqOff = cb.take<uint64_t>(nBig + 2); uint64_t* ptOff = cb.take<uint64_t>(nBig + 2);
  uint64_t* jobId = cb.take<uint64_t>(2 * nBig + 2);
  a.outcome = cb.take<uint8_t>(nBig + 2);
  int32_t* pqLen = cb.take<int32_t>(nBig + 2); int32_t* ptLen = cb.take<int32_t>(nBig + 2); uint32_t* ptSpan = cb.take<uint32_t>(nBig + 2); int32_t* pK = cb.take<int32_t>(nBig + 2);
  int32_t* pW = cb.take<int32_t>(nBig + 2); int32_t* pDiag = cb.take<int32_t>(nBig + 2); uint32_t* pqAdd = cb.take<uint32_t>(nBig + 2); uint32_t* ptAdd = cb.take<uint32_t>(nBig + 2);
  // (flip and max_freq share the tail of the block)
  char* wb2 = (char*)lra_ensure(ctx, 44, sz(nBig + 2, 4) * 2 + 1024);
  if (!wb2) return LRA_ERR_NOMEM;
  Carver cb2{wb2};
  uint32_t* pFlip = cb2.take<uint32_t>(nBig + 2); int32_t* pMf = cb2.take<int32_t>(nBig + 2);
  a.fOff = fOff; a.rOff = rOff; a.revId = revId; a.wholeId = wholeId; a.jobId = jobId; a.fId = fId; a.rId = rId;
  uint64_t nF = 0, nR = 0, nRev = 0, nJ = 0;
  if (nBig) {
    hipLaunchKernelGGL(lr_big_gather, grid(NA), dim3(256), 0, st, a, 0, pqOff, pqLen, ptOff, ptLen, ptSpan, pK, pW, pDiag, pqAdd, ptAdd, pFlip, pMf);
    lra_refine_space_result rs;
    { int rc = lra_refine_space_batch_mf(ctx, (int)nBig, d_strands, pqOff, pqLen, d_genome, ptOff, ptLen, ptSpan, pK, pW, pDiag, pqAdd, ptAdd, pFlip, opts->match,
                                         opts->mismatch, opts->indel, pMf, &rs); if (rc) return rc; }
    nF = rs.n_pairs;
    uint32_t* fQT = (uint32_t*)lra_ensure(ctx, 45, sz(nF + 1, 4) * 2 + 512);
    if (!fQT) return LRA_ERR_NOMEM;
    a.fQ = fQT; a.fT = fQT + sz(nF + 1, 4) / 4;
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(fOff, rs.d_pair_off, (nBig + 1) * 8, hipMemcpyDeviceToDevice, st));
    LRA_HIP_CHECK(ctx, hipMemcpyAsync((void*)a.fQ, rs.d_pair_q, nF * 4, hipMemcpyDeviceToDevice, st));
    LRA_HIP_CHECK(ctx, hipMemcpyAsync((void*)a.fT, rs.d_pair_t, nF * 4, hipMemcpyDeviceToDevice, st));
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(fId, rs.d_identity, nBig * 4, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(lr_need_rev, grid(nBig), dim3(256), 0, st, a);
    { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)nBig, a.needRev, revId); if (rc) return rc; }
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(&nRev, revId + nBig, 8, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (nRev) {
      hipLaunchKernelGGL(lr_big_gather, grid(NA), dim3(256), 0, st, a, 1, pqOff, pqLen, ptOff, ptLen, ptSpan, pK, pW, pDiag, pqAdd, ptAdd, pFlip, pMf);
      lra_refine_space_result rr;
      { int rc = lra_refine_space_batch_mf(ctx, (int)nRev, d_strands, pqOff, pqLen, d_genome, ptOff, ptLen, ptSpan, pK, pW, pDiag, pqAdd, ptAdd, pFlip, opts->match,
                                           opts->mismatch, opts->indel, pMf, &rr); if (rc) return rc; }
      nR = rr.n_pairs;
      uint32_t* rQT = (uint32_t*)lra_ensure(ctx, 46, sz(nR + 1, 4) * 2 + 512);
      if (!rQT) return LRA_ERR_NOMEM;
      a.rQ = rQT; a.rT = rQT + sz(nR + 1, 4) / 4;
      LRA_HIP_CHECK(ctx, hipMemcpyAsync(rOff, rr.d_pair_off, (nRev + 1) * 8, hipMemcpyDeviceToDevice, st));
      LRA_HIP_CHECK(ctx, hipMemcpyAsync((void*)a.rQ, rr.d_pair_q, nR * 4, hipMemcpyDeviceToDevice, st));
      LRA_HIP_CHECK(ctx, hipMemcpyAsync((void*)a.rT, rr.d_pair_t, nR * 4, hipMemcpyDeviceToDevice, st));
      LRA_HIP_CHECK(ctx, hipMemcpyAsync(rId, rr.d_identity, nRev * 4, hipMemcpyDeviceToDevice, st));
    }
    hipLaunchKernelGGL(lr_decide, grid(nBig), dim3(256), 0, st, a);
    { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)(2 * nBig), a.needJob, jobId); if (rc) return rc; }
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(&nJ, jobId + 2 * nBig, 8, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  }
  LR_DBG("spaces done: nF %llu nRev %llu nR %llu nJ %llu", (unsigned long long)nF, (unsigned long long)nRev, (unsigned long long)nR, (unsigned long long)nJ);
  a.nJ = nJ; out->n_inner_jobs = nJ;
  // ---- 6: seed-set jobs
  char* wj = (char*)lra_ensure(ctx, 47, sz(nJ + 2, 4) * 12 + sz(nJ + 2, 8) * 7 + sz(nJ + 2, 1) * 2 + 4096);
  if (!wj) return LRA_ERR_NOMEM;
  Carver cj{wj};
  a.jBig = cj.take<uint32_t>(nJ + 2); a.jCnt = cj.take<uint32_t>(nJ + 2); a.xCnt = cj.take<uint32_t>(nJ + 2); a.keptCnt = cj.take<uint32_t>(nJ + 2); a.pCnt = cj.take<uint32_t>(nJ + 2);
  uint32_t* eCount = cj.take<uint32_t>(nJ + 2); a.jStart = cj.take<int32_t>(nJ + 2); a.jEnd = cj.take<int32_t>(nJ + 2);
  a.cStr0 = cj.take<int>(nJ + 2); a.cChr = cj.take<int>(nJ + 2); a.cRd = cj.take<int>(nJ + 2); a.cK = cj.take<int>(nJ + 2);
  uint64_t* jPairOff = cj.take<uint64_t>(nJ + 2); uint64_t* xOff = cj.take<uint64_t>(nJ + 2); uint64_t* pOff = cj.take<uint64_t>(nJ + 2);
  a.cStart = cj.take<uint64_t>(nJ + 2); a.cEnd = cj.take<uint64_t>(nJ + 2); uint64_t* iota = cj.take<uint64_t>(nJ + 2); int32_t* zeros = (int32_t*)cj.take<uint64_t>(nJ + 2);
  a.jSet = cj.take<uint8_t>(nJ + 2); a.jFinal = cj.take<uint8_t>(nJ + 2);
  a.jPairOff = jPairOff; a.xOff = xOff; a.pOff = pOff; a.eCount = eCount;
  uint64_t nJobProb = 0, nWhole = 0;
  lra_chain_result sdp;
  memset(&sdp, 0, sizeof sdp);
  if (nJ) {
    hipLaunchKernelGGL(lr_jobs, grid(2 * nBig), dim3(256), 0, st, a);
    { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)nJ, a.jCnt, jPairOff); if (rc) return rc; }
    uint64_t nJP = 0;
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(&nJP, jPairOff + nJ, 8, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
    char* wp = (char*)lra_ensure(ctx, 48, sz(nJP + 1, 8) * 2 + sz(nJP + 1, 4) * 9 + sz(4 * nJ + 4, 4) + 4096);
    if (!wp) return LRA_ERR_NOMEM;
    Carver cp{wp};
    uint64_t* key = cp.take<uint64_t>(nJP + 1); uint64_t* key2 = cp.take<uint64_t>(nJP + 1);
    uint32_t* val = cp.take<uint32_t>(nJP + 1); uint32_t* val2 = cp.take<uint32_t>(nJP + 1);
    a.jq = cp.take<uint32_t>(nJP + 1); a.jt = cp.take<uint32_t>(nJP + 1); uint32_t* sq = cp.take<uint32_t>(nJP + 1); uint32_t* stt = cp.take<uint32_t>(nJP + 1);
    uint32_t* eq = cp.take<uint32_t>(nJP + 1); uint32_t* et = cp.take<uint32_t>(nJP + 1); int* el = (int*)cp.take<uint32_t>(nJP + 1); uint32_t* ebox = cp.take<uint32_t>(4 * nJ + 4);
    a.eq = eq; a.et = et; a.el = el;
    const unsigned gw = (unsigned)std::min<uint64_t>(nJ, (uint64_t)ctx->num_cu * 32);
    hipLaunchKernelGGL(lr_job_pairs, dim3(gw), dim3(64), 0, st, a, key, val);
    if (nJP) {
      size_t temp_bytes = 0;
      (void)lra_segsort_pairs(ctx, nullptr, temp_bytes, nullptr, nullptr, nullptr, nullptr, (unsigned int)nJP, (unsigned int)nJ, nullptr, nullptr, 0, 64, st);
      void* temp = lra_scratch(ctx, 2, temp_bytes + 256);
      if (!temp) return LRA_ERR_NOMEM;
      hipError_t e = lra_segsort_pairs(ctx, temp, temp_bytes, key, key2, val, val2, (unsigned int)nJP, (unsigned int)nJ, a.cStart, a.cEnd, 0, 64, st);
      if (e != hipSuccess) return lra_set_err(ctx, LRA_ERR_HIP, "segmented sort: %s", hipGetErrorString(e));
      hipLaunchKernelGGL(lr_sorted, grid(nJP), dim3(256), 0, st, nJP, (const uint32_t*)val2, (const uint32_t*)a.jq, (const uint32_t*)a.jt, (const uint64_t*)key2, sq, stt);
      hipLaunchKernelGGL(lr_add_coff, dim3(gw), dim3(64), 0, st, a, stt);
    }
    { int rc = lra_launch_linear_extend(ctx, nJ, 0, a.cStart, a.cEnd, a.cStr0, a.cChr, a.cRd, sq, stt, dpos, (const unsigned char*)d_genome, (const unsigned char*)d_strands,
                                        d_read_off, eq, et, el, eCount, ebox, a.cK); if (rc) return rc; }
    hipLaunchKernelGGL(lr_ext_count, grid(nJ), dim3(256), 0, st, a);
    { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)nJ, a.xCnt, xOff); if (rc) return rc; }
    uint64_t nX = 0;
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(&nX, xOff + nJ, 8, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
    char* wx = (char*)lra_ensure(ctx, 49, sz(nX + 1, 4) * 4 + 1024);
    if (!wx) return LRA_ERR_NOMEM;
    Carver cx{wx};
    a.xq = cx.take<uint32_t>(nX + 1); a.xt = cx.take<uint32_t>(nX + 1); a.xl = cx.take<int32_t>(nX + 1); a.kept = cx.take<uint32_t>(nX + 1);
    hipLaunchKernelGGL(lr_ext_gather, dim3(gw), dim3(64), 0, st, a);
    { int rc = lra_trim_anchor_pairs_batch(ctx, nJ, xOff, nX, a.xq, a.xt, a.xl); if (rc) return rc; }
    hipLaunchKernelGGL(lr_iota, grid(nJ + 1), dim3(256), 0, st, nJ, iota, zeros);
    lra_sdp_opts so;
    memset(&so, 0, sizeof so);
    so.rate = 2.0f; so.NumAln = 1; so.alnthres = 0; so.gapopen = opts->gapopen; so.gapextend = opts->gapextend; so.gaproot = opts->gaproot; so.gapCeiling1 = opts->gapCeiling1;
    so.gapCeiling2 = opts->gapCeiling2; so.mode = LRA_SDP_SINGLE_CLUSTER; so.globalK = 0;
    ctx->sdp_inner = true;
    const int rcS = lra_sparse_dp_batch(ctx, (int)nJ, iota, xOff, a.xCnt, zeros, a.xq, a.xt, a.xl, iota, nullptr, &so, &sdp);
    ctx->sdp_inner = false;
    if (rcS) return rcS;
    a.sStart = sdp.d_chain_start; a.sLen = sdp.d_chain_len; a.sAnchor = sdp.d_chain_anchor; a.sValue = sdp.d_chain_value; a.sStatus = sdp.d_status;
    hipLaunchKernelGGL(lr_inner_plan<0>, grid(nJ), dim3(256), 0, st, a);
    { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)nJ, a.pCnt, pOff); if (rc) return rc; }
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(&nJobProb, pOff + nJ, 8, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  } else LRA_HIP_CHECK(ctx, hipMemsetAsync(pOff, 0, 16, st));
  LR_DBG("jobs done: nJobProb %llu", (unsigned long long)nJobProb);
  a.nJobProb = nJobProb;
  if (nBig) {
    hipLaunchKernelGGL(lr_whole, grid(nBig), dim3(256), 0, st, a, 0);
    { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)nBig, a.wholeNeed, wholeId); if (rc) return rc; }
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(&nWhole, wholeId + nBig, 8, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  }
  // ---- 7: the second AOG batch
  const uint64_t nP2 = nJobProb + nWhole;
  {
    char* w2 = (char*)lra_ensure(ctx, 50, sz(nP2 + 1, 8) * 2 + sz(nP2 + 1, 4) * 4 + 1024);
    if (!w2) return LRA_ERR_NOMEM;
    Carver c2{w2};
    a.p2QB = c2.take<uint64_t>(nP2 + 1); a.p2TB = c2.take<uint64_t>(nP2 + 1); a.p2cre = c2.take<uint32_t>(nP2 + 1); a.p2nrs = c2.take<uint32_t>(nP2 + 1);
    a.p2cge = c2.take<uint32_t>(nP2 + 1); a.p2ngs = c2.take<uint32_t>(nP2 + 1);
    if (nJ) hipLaunchKernelGGL(lr_inner_plan<1>, grid(nJ), dim3(256), 0, st, a);
    if (nWhole) hipLaunchKernelGGL(lr_whole, grid(nBig), dim3(256), 0, st, a, 1);
    if (nP2) {
      lra_between_result b2;
      { int rc = lra_between_anchors_batch(ctx, (int)nP2, d_strands, a.p2QB, a.p2cre, a.p2nrs, d_genome, a.p2TB, a.p2cge, a.p2ngs, opts->match, opts->mismatch, opts->indel,
                                           opts->localBand, 1, &b2); if (rc) return rc; }
      a.d2Off = b2.d_block_off; a.d2Blk = b2.d_blocks;
    }
  }
  LR_DBG("second AOG done: nP2 %llu", (unsigned long long)nP2);
  // ---- 8: the walk
  lra_time_begin(ctx, "local_refine");
  hipLaunchKernelGGL(lr_walk<0>, grid(NJ0), dim3(256), 0, st, a);
  lra_time_end(ctx);
  { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)NJ0, a.nAln, alnOff); if (rc) return rc; }
  { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)NJ0, a.nBlk, blkOff); if (rc) return rc; }
  uint64_t nAln = 0, nBlk = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&nAln, alnOff + NJ0, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&nBlk, blkOff + NJ0, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  LR_DBG("walk counted: nAln %llu nBlk %llu", (unsigned long long)nAln, (unsigned long long)nBlk);
  char* wo = (char*)lra_ensure(ctx, 51, sz(nAln + 2, 4) * 7 + sz(nAln + 2, 8) + sz(3 * nBlk + 3, 4) + 4096);
  if (!wo) return LRA_ERR_NOMEM;
  Carver co{wo};
  a.oStrand = co.take<int32_t>(nAln + 2); a.oSupp = co.take<int32_t>(nAln + 2); a.oSec = co.take<int32_t>(nAln + 2); a.oN0 = co.take<int32_t>(nAln + 2);
  a.oN1 = co.take<int32_t>(nAln + 2); a.oChrom = co.take<int32_t>(nAln + 2); a.oValue = co.take<float>(nAln + 2); a.oBlockOff = co.take<uint64_t>(nAln + 2);
  a.oBlocks = co.take<int32_t>(3 * nBlk + 3);
  lra_time_begin(ctx, "local_refine");
  hipLaunchKernelGGL(lr_walk<1>, grid(NJ0), dim3(256), 0, st, a);
  lra_time_end(ctx);
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(a.oBlockOff + nAln, &nBlk, 8, hipMemcpyHostToDevice, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  LRA_HIP_CHECK(ctx, hipGetLastError());
  out->n_alignments = nAln; out->n_blocks = nBlk; out->d_job_aln_off = alnOff; out->d_strand = a.oStrand; out->d_supp = a.oSupp; out->d_secondary = a.oSec; out->d_n0 = a.oN0;
  out->d_n1 = a.oN1; out->d_chrom = a.oChrom; out->d_value = a.oValue; out->d_block_off = a.oBlockOff; out->d_blocks = a.oBlocks; out->d_status = a.status;
  (void)nR;
  return LRA_OK;
}

// ---- from the second sparse DP to the inputs of lra_local_refine_batch (Map_lowacc.h:530-540) ------------------------------------------------
namespace {

// (the second sparse DP's result has num_aln slots per merged cluster; its one chain is in the first)
__global__ void pi_len(uint64_t ng, int na2, const uint32_t* __restrict__ nChains, const uint32_t* __restrict__ chainLen, const uint32_t* __restrict__ status, uint32_t* len) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < ng) len[g] = (status[g] || nChains[g] == 0) ? 0 : chainLen[g * na2];
}
__global__ void __launch_bounds__(64) pi_gather(uint64_t ng, int na2, const uint64_t* __restrict__ off, const uint64_t* __restrict__ chainStart, const uint32_t* __restrict__ cq,
                                                const uint32_t* __restrict__ ct, const int32_t* __restrict__ cl, const int32_t* __restrict__ gstrand, uint32_t* q, uint32_t* t,
                                                int32_t* l, uint8_t* s) {
  for (uint64_t g = blockIdx.x; g < ng; g += gridDim.x) {
    const uint64_t o = off[g], b = chainStart[g * na2];
    const uint32_t n = (uint32_t)(off[g + 1] - o);
    for (uint32_t i = threadIdx.x; i < n; i += 64) { q[o + i] = cq[b + i]; t[o + i] = ct[b + i]; l[o + i] = cl[b + i]; s[o + i] = (uint8_t)(gstrand[g] != 0); }
  }
}
__global__ void __launch_bounds__(64) pi_kept(uint64_t ng, const uint64_t* __restrict__ off, const uint64_t* __restrict__ koff, const uint8_t* __restrict__ keep,
                                              const uint32_t* __restrict__ q, const uint32_t* __restrict__ t, const int32_t* __restrict__ l, uint32_t* oq, uint32_t* ot, int32_t* ol) {
  const int lane = threadIdx.x;
  for (uint64_t g = blockIdx.x; g < ng; g += gridDim.x) {
    const uint64_t o = off[g];
    const uint32_t n = (uint32_t)(off[g + 1] - o);
    uint64_t w = koff[g];
    for (uint32_t b0 = 0; b0 < n; b0 += 64) {
      const uint32_t i = b0 + lane;
      const bool k = i < n && keep[o + i];
      const unsigned long long m = __ballot(k);
      if (k) { const uint64_t d = w + __popcll(m & ((1ull << lane) - 1)); oq[d] = q[o + i]; ot[d] = t[o + i]; ol[d] = l[o + i]; }
      w += __popcll(m);
    }
  }
}
__global__ void pi_jobs(uint64_t nslots, int numAln, const uint32_t* __restrict__ firstLen, uint32_t* jobRead, int32_t* jobH) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  (void)firstLen;
  if (s < nslots) { jobRead[s] = (uint32_t)(s / numAln); jobH[s] = (int32_t)(s % numAln); }
}
__global__ void pi_chains(uint64_t ng, int na2, const uint32_t* __restrict__ gslot, const uint32_t* __restrict__ firstLen, const uint32_t* __restrict__ len, const float* __restrict__ val,
                          int32_t* n0, int32_t* n1, float* v) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < ng) { n0[g] = firstLen ? (int32_t)firstLen[gslot[g]] : 0; n1[g] = (int32_t)len[g]; v[g] = val[g * na2]; }
}

}  // namespace

// Map_lowacc.h:530-540 + :575: the chains of the per-merged-cluster sparse DP (`second`, single-cluster mode over lra_merge_extend_batch's output)
// go through RemovePairedIndels<UltimateChain> and RemoveSpuriousAnchors (lra_filter_chains_batch ops {2, 4}) and become the ultimatechains of
// their primary chain: job = chain slot of the first sparse DP, chains = its merged clusters in order.  FirstSDPValue = the second DP's best
// value (:2427), NumOfAnchors1 = its chain length (:2430); NumOfAnchors0 = d_slot_n0[slot] (chains[p].NumOfAnchors0 of the first sparse DP, whose
// result arrays the second one has overwritten by now: the caller keeps what it needs; NULL = 0; it only feeds MAPQ).
extern "C" int lra_local_refine_inputs_batch(lra_ctx* ctx, int num_aln, const uint32_t* d_slot_n0, const lra_merge_result* mg, const lra_chain_result* second,
                                             lra_local_refine_inputs* out) {
  if (!ctx || num_aln < 1 || !mg || !second || !out) return LRA_ERR_INVALID;
  memset(out, 0, sizeof *out);
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const uint64_t NG = mg->n_groups, slots = mg->n_slots;
  out->n_jobs = slots; out->n_chains = NG;
  if (slots == 0) return LRA_OK;
  auto grid = [](uint64_t n) { return dim3((unsigned)((n + 255) / 256)); };
  char* w = (char*)lra_ensure(ctx, 52, sz(NG + 2, 4) * 3 + sz(NG + 2, 8) * 2 + sz(NG + 2, 4) + sz(slots + 2, 4) * 2 + 4096);
  if (!w) return LRA_ERR_NOMEM;
  Carver c{w};
  uint32_t* len = c.take<uint32_t>(NG + 2); int32_t* n0 = c.take<int32_t>(NG + 2); int32_t* n1 = c.take<int32_t>(NG + 2); float* val = c.take<float>(NG + 2);
  uint64_t* off = c.take<uint64_t>(NG + 2); uint64_t* koff = c.take<uint64_t>(NG + 2);
  uint32_t* jobRead = c.take<uint32_t>(slots + 2); int32_t* jobH = c.take<int32_t>(slots + 2);
  hipLaunchKernelGGL(pi_jobs, grid(slots), dim3(256), 0, st, slots, num_aln, d_slot_n0, jobRead, jobH);
  out->d_job_chain_off = mg->d_slot_group_off; out->d_job_read = jobRead; out->d_job_h = jobH; out->d_chain_strand = mg->d_strand; out->d_chain_chrom = mg->d_chrom;
  out->d_chain_value = val; out->d_chain_n0 = n0; out->d_chain_n1 = n1; out->d_chain_anchor_off = koff;
  if (NG == 0) { LRA_HIP_CHECK(ctx, hipMemsetAsync(koff, 0, 16, st)); LRA_HIP_CHECK(ctx, hipStreamSynchronize(st)); return LRA_OK; }
  hipLaunchKernelGGL(pi_len, grid(NG), dim3(256), 0, st, NG, (int)second->num_aln, second->d_n_chains, second->d_chain_len, second->d_status, len);
  { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)NG, len, off); if (rc) return rc; }
  uint64_t NA2 = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&NA2, off + NG, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  char* wa = (char*)lra_ensure(ctx, 53, sz(NA2 + 1, 4) * 6 + sz(NA2 + 1, 1) + 4096);
  if (!wa) return LRA_ERR_NOMEM;
  Carver ca{wa};
  uint32_t* q = ca.take<uint32_t>(NA2 + 1); uint32_t* t = ca.take<uint32_t>(NA2 + 1); int32_t* l = ca.take<int32_t>(NA2 + 1);
  uint32_t* oq = ca.take<uint32_t>(NA2 + 1); uint32_t* ot = ca.take<uint32_t>(NA2 + 1); int32_t* ol = ca.take<int32_t>(NA2 + 1); uint8_t* s8 = ca.take<uint8_t>(NA2 + 1);
  const unsigned gw = (unsigned)std::min<uint64_t>(NG, (uint64_t)ctx->num_cu * 32);
  hipLaunchKernelGGL(pi_gather, dim3(gw), dim3(64), 0, st, NG, (int)second->num_aln, (const uint64_t*)off, second->d_chain_start, second->d_chain_q, second->d_chain_t, second->d_chain_alen,
                     mg->d_strand, q, t, l, s8);
  hipLaunchKernelGGL(pi_chains, grid(NG), dim3(256), 0, st, NG, (int)second->num_aln, mg->d_group_slot, d_slot_n0, (const uint32_t*)len, second->d_chain_value, n0, n1, val);
  LR_DBG("inputs: NG %llu NA2 %llu", (unsigned long long)NG, (unsigned long long)NA2);
  lra_filter_result fr;
  const int ops[2] = {2, 4};
  { int rc = lra_filter_chains_batch(ctx, NG, off, NA2, q, t, l, s8, nullptr, ops, 2, &fr); if (rc) return rc; }
  { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)NG, fr.d_n_kept, koff); if (rc) return rc; }
  uint64_t NK = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&NK, koff + NG, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  LR_DBG("filtered: NK %llu", (unsigned long long)NK);
  hipLaunchKernelGGL(pi_kept, dim3(gw), dim3(64), 0, st, NG, (const uint64_t*)off, (const uint64_t*)koff, fr.d_keep, (const uint32_t*)q, (const uint32_t*)t, (const int32_t*)l, oq, ot, ol);
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  LRA_HIP_CHECK(ctx, hipGetLastError());
  out->n_anchors = NK; out->d_q = oq; out->d_t = ot; out->d_len = ol;
  return LRA_OK;
}
