// lra_amd/csrc/refine_btwn.hip -- SURVEY §8a row a11 (callers of RefineSpace on the low-accuracy path): Refine_Btwnsplitchain
// (ChainRefine.h:579-754, called at Map_lowacc.h:362) for every chain of a batch.  gfx950 only.
//   RefineBtwnSpace_AppendCloseCluster   ChainRefine.h:59-121 (append_to_closetcluster :22-56, minGapDifference Clustering.h:532)
//   RefineBtwnSpace                      ClusterRefine.h:327-430
//   Cluster::SetClusterBoundariesFromMatches   Clustering.h:308-322
// Mapping.  Gap c of a chain reads the boxes of refined clusters c-1 and c, which gap c-1 may have moved, so the chains advance in
// lock step: round c plans gap c of every chain that has one (one lane per chain slot: the case analysis of :587-660 -> up to two
// RefineSpace problems, whose arguments all come from the boxes at the start of the iteration), lra_refine_space_batch solves all
// planned problems of the round at once, and an apply pass (one lane per chain slot again) takes the reference's decisions on the
// results in order: dense results are appended as they are, sparse ones are CartesianSort-ed (a segmented radix sort of the packed
// pairs -- only identical pairs tie), cut into runs and each run appended to the nearer cluster.  Two more rounds seed beyond the first
// and beyond the last split chain.  A chain has one to three split chains, so a batch takes three to five rounds.
// Appended matches are not moved while the rounds run: every cluster keeps a linked list of segments of a pair pool, and its box grows
// by min / max (SetClusterBoundariesFromMatches over all matches = the old box extended by the new ones, K being the same); a last
// pass lays every cluster's matches out contiguously, base matches first, segments in order.
// Only the -ONT / -CLR read types: for the others the reference leaves refineSpaceDiag uninitialised (ChainRefine.h:68-71).
#include "common.h"
#include "scan.h"
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <vector>

namespace {

constexpr uint32_t NONE = 0xFFFFFFFFu;

struct Seg { uint64_t off; uint32_t len; uint32_t next; };

struct BtArgs {
  uint64_t n_slots; int numAln;
  const uint32_t* nChains; const uint64_t* chainStart; const uint32_t* nSplit; const uint32_t* spStatus; const uint8_t* spStrand; const int32_t* spChrom;
  const uint8_t* splitLink;
  const uint64_t* read_off; uint64_t rcBase;
  const uint64_t* pos; int npos;
  int K, W; uint32_t RSD; float sparse2;
  int round;                                 // >= 1: gap index; -1: beyond the first split chain; -2: beyond the last one
  // cluster state, indexed like the split arrays
  uint32_t* box; uint32_t* cnt; uint8_t* rspace; uint32_t* segHead; uint32_t* segTail;
  // planned problems, two per slot
  uint32_t* need; const uint64_t* needOff;
  uint64_t* pQoff; int32_t* pQlen; uint64_t* pToff; int32_t* pTlen; uint32_t* pTspan; int32_t* pK; int32_t* pW; int32_t* pDiag; uint32_t* pQadd;
  uint32_t* pTadd; uint32_t* pFlip;
  uint32_t* mSpan; uint8_t* mKind; uint8_t* mTwo; uint8_t* mSt; uint32_t* mX; uint32_t* mXp;
  // results of the round
  const uint64_t* pairOff; const uint64_t* pool; uint64_t poolU, poolS;   // this round's pairs in the pool: as produced / sorted
  Seg* segs; uint32_t* segCount; uint32_t segCap; uint32_t* overflow;
};

__device__ __forceinline__ int space_diag(uint32_t qe, uint32_t qs) {    // ChainRefine.h:70 / ClusterRefine.h:345 (clr, ont)
  return min((int)floorf(fmaxf(100.f, 0.15f * (float)(qe - qs))), 1000);
}

// one RefineSpace call (consider_str = 1) as a problem of the round
__device__ void plan_problem(const BtArgs& a, uint64_t s, int j, uint64_t r, uint32_t readLen, int chrom, uint32_t qe, uint32_t qs, uint32_t te, uint32_t ts,
                             int st, uint32_t lrts, uint32_t lrlength, int kind, int two, uint32_t x, uint32_t xp) {
  if (st == 1) { const uint32_t t = qs; qs = readLen - qe; qe = readLen - t; }
  const uint64_t i = 2 * s + j;
  a.need[i] = 1;
  a.pQoff[i] = (st ? a.rcBase : 0) + a.read_off[r] + qs; a.pQlen[i] = (int32_t)(qe - qs);
  a.pToff[i] = a.pos[chrom] + (ts - lrts); a.pTlen[i] = (int32_t)(te - ts + lrlength); a.pTspan[i] = te - (ts - lrts);
  a.pK[i] = a.K; a.pW[i] = a.W; a.pDiag[i] = space_diag(qe, qs); a.pQadd[i] = qs; a.pTadd[i] = ts - lrts; a.pFlip[i] = st == 1 ? readLen : 0;
  a.mSpan[i] = min(qe - qs, te - ts); a.mKind[i] = (uint8_t)kind; a.mTwo[i] = (uint8_t)two; a.mSt[i] = (uint8_t)st; a.mX[i] = x; a.mXp[i] = xp;
}

__global__ void bt_plan(BtArgs a) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= a.n_slots) return;
  a.need[2 * s] = 0; a.need[2 * s + 1] = 0;
  const uint64_t r = s / a.numAln;
  if ((uint32_t)(s % a.numAln) >= a.nChains[r] || a.spStatus[s]) return;
  const uint32_t nsp = a.nSplit[s];
  const uint64_t cs = a.chainStart[s];
  const uint32_t readLen = (uint32_t)(a.read_off[r + 1] - a.read_off[r]);
  auto glen = [&](int c) { return a.pos[c + 1] - a.pos[c]; };
  if (a.round >= 1) {                                                     // ChainRefine.h:587-682, iteration c = round
    const uint32_t c = (uint32_t)a.round;
    if (c >= nsp) return;
    const uint64_t xc = cs + c, xp = cs + c - 1;
    if (a.cnt[xc] == 0 || a.cnt[xp] == 0) return;
    const uint32_t cqs = a.box[4 * xc], cqe = a.box[4 * xc + 1], cts = a.box[4 * xc + 2], cte = a.box[4 * xc + 3];
    const uint32_t pqs = a.box[4 * xp], pts = a.box[4 * xp + 2], pte = a.box[4 * xp + 3];
    (void)cqs;
    const int cst = a.spStrand[xc], pst = a.spStrand[xp], link = a.splitLink[cs + c - 1];
    const int cchrom = a.spChrom[xc], pchrom = a.spChrom[xp];
    const uint32_t qs = cqe, qe = pqs;
    uint32_t ts1 = 0, te1 = 0, ts2 = 0, te2 = 0;
    if (qe <= qs) return;
    const uint32_t gap = qe - qs;
    int st1 = 0, st2 = 0, two = 0;
    if (cst == pst && link == 0) {
      two = 0; st1 = cst;
      if (cte <= pts) { ts1 = cte; te1 = pts; }
      else if (cts > pte) { ts1 = pte; te1 = cts; }
      else return;
    } else if (cst != pst && link == 1) {
      st1 = cst; st2 = pst; two = 1;
      if (cte <= pts) {
        if (st1 == 0) { ts1 = cte; te1 = ts1 + gap; ts2 = pte; te2 = ts2 + gap; }
        else { te1 = cts; ts1 = te1 > gap ? te1 - gap : 0; te2 = pts; ts2 = te2 > gap ? te2 - gap : 0; }
      } else if (cts > pte) {
        if (st1 == 0) { ts1 = cte; te1 = ts1 + gap; te2 = cts; ts2 = te2 > gap ? te2 - gap : 0; }
        else { te1 = cts; ts1 = te1 > gap ? te1 - gap : 0; te2 = pts; ts2 = te2 > gap ? te2 - gap : 0; }
      } else return;
    } else if (cst == pst && link == 1) {
      st1 = cst; st2 = st1; two = 1;
      if (st1 == 0 && cte > pts) { ts1 = cte; te1 = ts1 + gap; te2 = pts; ts2 = te2 > gap ? te2 - gap : 0; }
      else if (st1 == 1 && cts < pte) { te1 = cts; ts1 = te1 > gap ? te1 - gap : 0; ts2 = pte; te2 = ts2 + gap; }
      else return;
    }
    if (te1 <= ts1) return;                                               // also the strands-differ, link 0 case (ts1 = te1 = 0)
    if (te1 >= glen(cchrom)) return;
    if (max(gap, te1 - ts1) >= 5 * a.RSD) return;
    uint32_t space = max(gap, te1 - ts1);
    if (space >= 20 && space <= a.RSD && cchrom == pchrom)
      plan_problem(a, s, 0, r, readLen, cchrom, qe, qs, te1, ts1, st1, 0, 0, 0, two, (uint32_t)(xc - cs), (uint32_t)(xp - cs));
    if (two) {
      if (te2 <= ts2) return;
      if (te2 >= glen(cchrom)) return;
      if (max(gap, te2 - ts2) >= 5 * a.RSD) return;
      space = max(gap, te2 - ts2);
      if (space >= 20 && space <= a.RSD && cchrom == pchrom)
        plan_problem(a, s, 1, r, readLen, pchrom, qe, qs, te2, ts2, st2, 0, 0, 1, 1, (uint32_t)(xp - cs), (uint32_t)(xp - cs));
    }
    return;
  }
  if (nsp == 0) return;
  if (a.round == -1) {                                                    // :684-724
    const uint64_t x = cs;
    if (a.cnt[x] == 0) return;
    const int st = a.spStrand[x], chrom = a.spChrom[x];
    const uint32_t qs = a.box[4 * x + 1], qe = readLen;
    uint32_t ts = 0, te = 0;
    if (st == 0) { ts = a.box[4 * x + 3]; te = ts + qe - qs; }
    else { te = a.box[4 * x + 2]; if (te > qe - qs) ts = te - (qe - qs); else return; }   // te = 0 there: `te > ts` fails whatever ts holds
    if (!(qe > qs && te > ts)) return;
    const uint32_t space = max(qe - qs, te - ts);
    if (!(space >= 20 && space < a.RSD && (uint64_t)te + 500 < glen(chrom))) return;
    uint32_t lrts = 0, lrlength = 0;
    if (st == 0) { lrts = 0; lrlength = 500; } else { if (ts > 500) lrts = 500; lrlength = lrts; }
    plan_problem(a, s, 0, r, readLen, chrom, qe, qs, te, ts, st, lrts, lrlength, 1, 1, 0, 0);
  } else {                                                                // :725-753
    const uint64_t x = cs + nsp - 1;
    if (a.cnt[x] == 0) return;
    const int st = a.spStrand[x], chrom = a.spChrom[x];
    const uint32_t qs = 0, qe = a.box[4 * x];
    uint32_t ts, te;
    if (st == 0) { te = a.box[4 * x + 2]; ts = te > qe - qs ? te - (qe - qs) : 0; }
    else { ts = a.box[4 * x + 3]; te = ts + (qe - qs); }
    if (!(qe > qs && te > ts)) return;
    const uint32_t space = max(qe - qs, te - ts);
    if (!(space >= 20 && space < a.RSD && (uint64_t)te + 500 < glen(chrom))) return;
    uint32_t lrts = 0, lrlength = 0;
    if (st == 0) { if (ts > 500) lrts = 500; lrlength = lrts; } else { lrts = 0; lrlength = 500; }
    plan_problem(a, s, 0, r, readLen, chrom, qe, qs, te, ts, st, lrts, lrlength, 1, 1, nsp - 1, nsp - 1);
  }
}

// dense problem list of the round
__global__ void bt_compact(BtArgs a, uint64_t* dQoff, int32_t* dQlen, uint64_t* dToff, int32_t* dTlen, uint32_t* dTspan, int32_t* dK, int32_t* dW,
                           int32_t* dDiag, uint32_t* dQadd, uint32_t* dTadd, uint32_t* dFlip) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * a.n_slots || !a.need[i]) return;
  const uint64_t p = a.needOff[i];
  dQoff[p] = a.pQoff[i]; dQlen[p] = a.pQlen[i]; dToff[p] = a.pToff[i]; dTlen[p] = a.pTlen[i]; dTspan[p] = a.pTspan[i]; dK[p] = a.pK[i]; dW[p] = a.pW[i];
  dDiag[p] = a.pDiag[i]; dQadd[p] = a.pQadd[i]; dTadd[p] = a.pTadd[i]; dFlip[p] = a.pFlip[i];
}

__global__ void bt_pack(uint64_t n, const uint32_t* __restrict__ q, const uint32_t* __restrict__ t, uint64_t* out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = ((uint64_t)q[i] << 32) | t[i];
}

// matches [off, off + len) of the pool join cluster x (insert + SetClusterBoundariesFromMatches)
__device__ void append_seg(const BtArgs& a, uint64_t x, uint64_t off, uint32_t len) {
  if (len == 0) return;
  const uint32_t id = atomicAdd(a.segCount, 1u);
  if (id >= a.segCap) { *a.overflow = 1; return; }
  a.segs[id].off = off; a.segs[id].len = len; a.segs[id].next = NONE;
  if (a.segHead[x] == NONE) a.segHead[x] = id; else a.segs[a.segTail[x]].next = id;
  a.segTail[x] = id;
  uint32_t qs = a.box[4 * x], qe = a.box[4 * x + 1], ts = a.box[4 * x + 2], te = a.box[4 * x + 3];
  for (uint32_t i = 0; i < len; i++) {
    const uint64_t v = a.pool[off + i];
    const uint32_t q = (uint32_t)(v >> 32), t = (uint32_t)v;
    qs = min(qs, q); qe = max(qe, q + (uint32_t)a.K); ts = min(ts, t); te = max(te, t + (uint32_t)a.K);
  }
  a.box[4 * x] = qs; a.box[4 * x + 1] = qe; a.box[4 * x + 2] = ts; a.box[4 * x + 3] = te;
  a.cnt[x] += len;
}

__global__ void bt_apply(BtArgs a) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= a.n_slots) return;
  for (int j = 0; j < 2; j++) {
    const uint64_t i = 2 * s + j;
    if (!a.need[i]) continue;
    const uint64_t cs = a.chainStart[s];
    const uint64_t p = a.needOff[i];
    const uint64_t p0 = a.pairOff[p];
    const uint32_t n = (uint32_t)(a.pairOff[p + 1] - p0);
    const float eff = ((float)n) / (float)a.mSpan[i];
    const uint64_t x = cs + a.mX[i], xp = cs + a.mXp[i];
    if (n == 0) continue;
    if (a.mKind[i] == 1) { append_seg(a, x, a.poolU + p0, n); a.rspace[x] = 1; continue; }   // RefineBtwnSpace with twoblocks :348-353
    if (eff >= a.sparse2) { append_seg(a, x, a.poolU + p0, n); a.rspace[x] = 1; continue; }  // ChainRefine.h:79-84
    if (a.mTwo[i]) continue;
    // :87-120 on the CartesianSort-ed pairs (max_pairdist <= 100 and eff >= threshold cannot both hold here: eff < threshold)
    const uint64_t* S = a.pool + a.poolS + p0;
    const int st = a.mSt[i];
    uint32_t start = 0;
    while (start < n) {
      uint32_t end = start + 1;
      while (end < n) {
        const long long dq = llabs((long long)(uint32_t)(S[end] >> 32) - (long long)(uint32_t)(S[end - 1] >> 32));
        const long long dt = llabs((long long)(uint32_t)S[end] - (long long)(uint32_t)S[end - 1]);
        if (min(dq, dt) <= 200) end++; else break;
      }
      if (end - start >= 4) {                                             // append_to_closetcluster :22-56
        uint32_t qS = (uint32_t)(S[start] >> 32), qE = qS + a.K, tS = (uint32_t)S[start], tE = tS + a.K;
        for (uint32_t k = start + 1; k < end; k++) {
          const uint32_t q = (uint32_t)(S[k] >> 32), t = (uint32_t)S[k];
          tE = max(tE, t + (uint32_t)a.K); tS = min(tS, t); qE = max(qE, q + (uint32_t)a.K); qS = min(qS, q);
        }
        const uint32_t cqe = a.box[4 * x + 1], cts = a.box[4 * x + 2], cte = a.box[4 * x + 3];
        const uint32_t pqs = a.box[4 * xp], pts = a.box[4 * xp + 2], pte = a.box[4 * xp + 3];
        int qd = qS >= cqe ? (int)(qS - cqe) : 0, td;
        if (st == 0) td = tS >= cte ? (int)(tS - cte) : 0; else td = cts >= tE ? (int)(cts - tE) : 0;
        const int dist_cur = max(qd, td);
        qd = pqs >= qE ? (int)(pqs - qE) : 0;
        if (st == 0) td = pts >= tE ? (int)(pts - tE) : 0; else td = tS >= pte ? (int)(tS - pte) : 0;
        const int dist_prev = max(qd, td);
        append_seg(a, dist_cur <= dist_prev ? x : xp, a.poolS + p0 + start, end - start);
      }
      start = end;
    }
  }
}

struct LayArgs {
  uint64_t n_slots; int numAln; int K;
  const uint32_t* nChains; const uint64_t* chainStart; const uint32_t* nSplit; const uint32_t* spStatus;
  const uint64_t* baseOff; const uint32_t* baseQ; const uint32_t* baseT;
  const uint32_t* cnt; const uint32_t* box; const uint32_t* segHead; const Seg* segs; const uint64_t* pool;
  const uint64_t* outOff; uint32_t* oq; uint32_t* ot; float* eff;
};

__global__ void __launch_bounds__(64) bt_layout(LayArgs a) {
  const uint64_t s = blockIdx.x;
  const uint64_t r = s / a.numAln;
  if ((uint32_t)(s % a.numAln) >= a.nChains[r] || a.spStatus[s]) return;
  const uint64_t cs = a.chainStart[s];
  const int lane = threadIdx.x;
  for (uint32_t k = 0; k < a.nSplit[s]; k++) {
    const uint64_t x = cs + k;
    uint64_t o = a.outOff[x];
    const uint64_t b0 = a.baseOff[x], b1 = a.baseOff[x + 1];
    for (uint64_t i = b0 + lane; i < b1; i += 64) { a.oq[o + (i - b0)] = a.baseQ[i]; a.ot[o + (i - b0)] = a.baseT[i]; }
    o += b1 - b0;
    for (uint32_t g = a.segHead[x]; g != NONE; g = a.segs[g].next) {
      const Seg sg = a.segs[g];
      for (uint32_t i = lane; i < sg.len; i += 64) { const uint64_t v = a.pool[sg.off + i]; a.oq[o + i] = (uint32_t)(v >> 32); a.ot[o + i] = (uint32_t)v; }
      o += sg.len;
    }
    if (lane == 0) {
      const uint32_t n = a.cnt[x];
      a.eff[x] = n ? ((float)n) / (float)min(a.box[4 * x + 1] - a.box[4 * x], a.box[4 * x + 3] - a.box[4 * x + 2]) : 0.f;   // Clustering.h:321
    }
  }
}

__global__ void bt_init(uint64_t nf, const uint64_t* __restrict__ baseOff, const uint32_t* __restrict__ baseBox, uint32_t* cnt, uint32_t* box,
                        uint32_t* segHead, uint32_t* segTail, uint8_t* rspace) {
  const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (x > nf) return;
  if (x == nf) { cnt[x] = 0; return; }
  cnt[x] = (uint32_t)(baseOff[x + 1] - baseOff[x]);
  box[4 * x] = baseBox[4 * x]; box[4 * x + 1] = baseBox[4 * x + 1]; box[4 * x + 2] = baseBox[4 * x + 2]; box[4 * x + 3] = baseBox[4 * x + 3];
  segHead[x] = NONE; segTail[x] = NONE; rspace[x] = 0;
}

inline size_t sz(size_t n, size_t e) { return (n * e + 255) / 256 * 256; }

// growable buffer whose first `used` bytes survive growth
void* grow_keep(lra_ctx* ctx, int slot, size_t used, size_t need) {
  if (ctx->gbuf[slot] && ctx->gbytes[slot] >= need) return ctx->gbuf[slot];
  const size_t want = need + need / 2 + 4096;
  void* p = nullptr;
  if (hipMalloc(&p, want) != hipSuccess) { lra_set_err(ctx, LRA_ERR_NOMEM, "hipMalloc(%zu) failed", want); return nullptr; }
  if (ctx->gbuf[slot]) {
    if (used) (void)hipMemcpyAsync(p, ctx->gbuf[slot], used, hipMemcpyDeviceToDevice, ctx->stream);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(ctx->gbuf[slot]);
  }
  ctx->gbuf[slot] = p; ctx->gbytes[slot] = want;
  return p;
}

}  // namespace

extern "C" int lra_refine_btwn_splitchain_batch(lra_ctx* ctx, const lra_chain_result* ch, const lra_split_result* sp, const lra_refined_result* rf,
                                                const uint64_t* d_read_off, const char* d_strands, uint64_t rc_base, const char* d_genome,
                                                const uint64_t* h_chrom_pos, int n_chrom, const lra_btwn_opts* opts, lra_btwn_result* out) {
  if (!ctx || !ch || !sp || !rf || !out || !opts || !h_chrom_pos || n_chrom < 1) return LRA_ERR_INVALID;
  memset(out, 0, sizeof *out);
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const uint64_t slots = sp->n_slots, NF = sp->n_frags;
  out->n_frags = NF;
  if (slots == 0) return LRA_OK;
  const size_t npos = (size_t)n_chrom + 1, n2 = 2 * slots;
  auto take = [](char*& p, size_t n, size_t e) { char* r = p; p += sz(n, e); return r; };
  char* w = (char*)lra_ensure(ctx, 35, sz(NF + 2, 4) * 3 + sz(4 * NF + 4, 4) + sz(NF + 2, 1) + sz(NF + 2, 8) + sz(NF + 1, 4) + sz(npos, 8) + sz(n2 + 1, 4) * 12 +
                                           sz(n2 + 2, 8) * 3 + sz(n2 + 1, 1) * 3 + 4096);
  if (!w) return LRA_ERR_NOMEM;
  BtArgs a;
  memset(&a, 0, sizeof a);
  a.n_slots = slots; a.numAln = ch->num_aln; a.nChains = ch->d_n_chains; a.chainStart = ch->d_chain_start; a.nSplit = sp->d_n_split; a.spStatus = sp->d_status;
  a.spStrand = sp->d_sp_strand; a.spChrom = sp->d_sp_chrom; a.splitLink = sp->d_split_link; a.read_off = d_read_off; a.rcBase = rc_base;
  a.K = opts->K; a.W = opts->W; a.RSD = (uint32_t)opts->refineSpaceDist; a.sparse2 = opts->anchorstoosparse * 2;
  a.cnt = (uint32_t*)take(w, NF + 2, 4); a.segHead = (uint32_t*)take(w, NF + 2, 4); a.segTail = (uint32_t*)take(w, NF + 2, 4);
  a.box = (uint32_t*)take(w, 4 * NF + 4, 4); a.rspace = (uint8_t*)take(w, NF + 2, 1);
  uint64_t* outOff = (uint64_t*)take(w, NF + 2, 8); float* eff = (float*)take(w, NF + 1, 4);
  uint64_t* dpos = (uint64_t*)take(w, npos, 8);
  a.pos = dpos; a.npos = (int)npos;
  a.need = (uint32_t*)take(w, n2 + 1, 4); a.pQlen = (int32_t*)take(w, n2 + 1, 4); a.pTlen = (int32_t*)take(w, n2 + 1, 4); a.pTspan = (uint32_t*)take(w, n2 + 1, 4);
  a.pK = (int32_t*)take(w, n2 + 1, 4); a.pW = (int32_t*)take(w, n2 + 1, 4); a.pDiag = (int32_t*)take(w, n2 + 1, 4); a.pQadd = (uint32_t*)take(w, n2 + 1, 4);
  a.pTadd = (uint32_t*)take(w, n2 + 1, 4); a.pFlip = (uint32_t*)take(w, n2 + 1, 4); a.mSpan = (uint32_t*)take(w, n2 + 1, 4); a.mX = (uint32_t*)take(w, n2 + 1, 4);
  uint64_t* needOff = (uint64_t*)take(w, n2 + 2, 8); a.pQoff = (uint64_t*)take(w, n2 + 2, 8); a.pToff = (uint64_t*)take(w, n2 + 2, 8);
  a.mKind = (uint8_t*)take(w, n2 + 1, 1); a.mTwo = (uint8_t*)take(w, n2 + 1, 1); a.mSt = (uint8_t*)take(w, n2 + 1, 1);
  a.needOff = needOff;
  // mXp shares nothing: its own small buffer behind the dense problem arrays (slot 31)
  char* wd = (char*)lra_ensure(ctx, 36, sz(n2 + 1, 4) * 10 + sz(n2 + 1, 8) * 2 + 4096);
  if (!wd) return LRA_ERR_NOMEM;
  a.mXp = (uint32_t*)take(wd, n2 + 1, 4);
  uint64_t* dQoff = (uint64_t*)take(wd, n2 + 1, 8); uint64_t* dToff = (uint64_t*)take(wd, n2 + 1, 8);
  int32_t* dQlen = (int32_t*)take(wd, n2 + 1, 4); int32_t* dTlen = (int32_t*)take(wd, n2 + 1, 4); uint32_t* dTspan = (uint32_t*)take(wd, n2 + 1, 4);
  int32_t* dK = (int32_t*)take(wd, n2 + 1, 4); int32_t* dW = (int32_t*)take(wd, n2 + 1, 4); int32_t* dDiag = (int32_t*)take(wd, n2 + 1, 4);
  uint32_t* dQadd = (uint32_t*)take(wd, n2 + 1, 4); uint32_t* dTadd = (uint32_t*)take(wd, n2 + 1, 4); uint32_t* dFlip = (uint32_t*)take(wd, n2 + 1, 4);
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(dpos, h_chrom_pos, npos * 8, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(bt_init, dim3((unsigned)((NF + 256) / 256)), dim3(256), 0, st, NF, rf->d_match_off, rf->d_box, a.cnt, a.box, a.segHead, a.segTail, a.rspace);
  // the longest chain sets the number of gap rounds
  std::vector<uint32_t> h_nsplit(slots);
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(h_nsplit.data(), sp->d_n_split, slots * 4, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  uint32_t maxSplit = 0;
  for (uint64_t i = 0; i < slots; i++) maxSplit = std::max(maxSplit, h_nsplit[i]);
  const int SEG_SLOT = 32, POOL_SLOT = 33, OUT_SLOT = 34;
  uint32_t* segCtl = (uint32_t*)lra_scratch(ctx, 3, 256);
  if (!segCtl) return LRA_ERR_NOMEM;
  LRA_HIP_CHECK(ctx, hipMemsetAsync(segCtl, 0, 256, st));
  a.segCount = segCtl; a.overflow = segCtl + 1;
  uint64_t poolUsed = 0; uint32_t segUsedBound = 0;
  std::vector<int> rounds;
  for (uint32_t c = 1; c < maxSplit; c++) rounds.push_back((int)c);
  if (maxSplit > 0) { rounds.push_back(-1); rounds.push_back(-2); }
  const unsigned gs = (unsigned)((slots + 255) / 256), g2 = (unsigned)((n2 + 255) / 256);
  uint64_t totalProblems = 0, totalPairs = 0;
  for (int rd : rounds) {
    a.round = rd;
    lra_time_begin(ctx, "btwn_plan");
    hipLaunchKernelGGL(bt_plan, dim3(gs), dim3(256), 0, st, a);
    lra_time_end(ctx);
    { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)n2, a.need, needOff); if (rc) return rc; }
    uint64_t np = 0;
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(&np, needOff + n2, 8, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (np == 0) continue;
    totalProblems += np;
    hipLaunchKernelGGL(bt_compact, dim3(g2), dim3(256), 0, st, a, dQoff, dQlen, dToff, dTlen, dTspan, dK, dW, dDiag, dQadd, dTadd, dFlip);
    lra_refine_space_result rs;
    { int rc = lra_refine_space_batch(ctx, (int)np, d_strands, dQoff, dQlen, d_genome, dToff, dTlen, dTspan, dK, dW, dDiag, dQadd, dTadd, dFlip, opts->match,
                                      opts->mismatch, opts->indel, opts->max_freq, &rs); if (rc) return rc; }
    const uint64_t nP = rs.n_pairs;
    totalPairs += nP;
    uint64_t* pool = (uint64_t*)grow_keep(ctx, POOL_SLOT, poolUsed * 8, (poolUsed + 2 * nP + 2) * 8);
    if (!pool) return LRA_ERR_NOMEM;
    const uint32_t segPrev = segUsedBound;
    segUsedBound += (uint32_t)(nP / 4 + 2 * np + 2);                    // a run has >= 4 pairs; at most one whole-list segment per problem
    Seg* segs = (Seg*)grow_keep(ctx, SEG_SLOT, (size_t)segPrev * sizeof(Seg), (size_t)segUsedBound * sizeof(Seg));
    if (!segs) return LRA_ERR_NOMEM;
    a.pool = pool; a.poolU = poolUsed; a.poolS = poolUsed + nP; a.segs = segs; a.segCap = segUsedBound; a.pairOff = rs.d_pair_off;
    if (nP > 0) {
      lra_time_begin(ctx, "btwn_apply");
      hipLaunchKernelGGL(bt_pack, dim3((unsigned)((nP + 255) / 256)), dim3(256), 0, st, nP, rs.d_pair_q, rs.d_pair_t, pool + poolUsed);
      size_t temp_bytes = 0;
      (void)rocprim::segmented_radix_sort_keys(nullptr, temp_bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (unsigned int)nP, (unsigned int)np, (uint64_t*)nullptr,
                                               (uint64_t*)nullptr, 0, 64, st);
      void* temp = lra_scratch(ctx, 2, temp_bytes + 256);
      if (!temp) { lra_time_end(ctx); return LRA_ERR_NOMEM; }
      hipError_t e = rocprim::segmented_radix_sort_keys(temp, temp_bytes, pool + poolUsed, pool + poolUsed + nP, (unsigned int)nP, (unsigned int)np, rs.d_pair_off,
                                                        rs.d_pair_off + 1, 0, 64, st);
      lra_time_end(ctx);
      if (e != hipSuccess) return lra_set_err(ctx, LRA_ERR_HIP, "segmented sort: %s", hipGetErrorString(e));
    }
    lra_time_begin(ctx, "btwn_apply");
    hipLaunchKernelGGL(bt_apply, dim3(gs), dim3(256), 0, st, a);
    lra_time_end(ctx);
    poolUsed += 2 * nP;
  }
  uint32_t ctl[2] = {0, 0};
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(ctl, segCtl, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  if (ctl[1]) return lra_set_err(ctx, LRA_ERR_HIP, "internal: segment pool overflow");
  // layout
  { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)NF + 1, a.cnt, outOff); if (rc) return rc; }
  uint64_t NM = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&NM, outOff + NF, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  char* wm = (char*)lra_ensure(ctx, OUT_SLOT, sz(NM + 1, 4) * 2 + 1024);
  if (!wm) return LRA_ERR_NOMEM;
  uint32_t* oq = (uint32_t*)take(wm, NM + 1, 4); uint32_t* ot = (uint32_t*)take(wm, NM + 1, 4);
  LayArgs l;
  memset(&l, 0, sizeof l);
  l.n_slots = slots; l.numAln = ch->num_aln; l.K = opts->K; l.nChains = ch->d_n_chains; l.chainStart = ch->d_chain_start; l.nSplit = sp->d_n_split; l.spStatus = sp->d_status;
  l.baseOff = rf->d_match_off; l.baseQ = rf->d_match_q; l.baseT = rf->d_match_t; l.cnt = a.cnt; l.box = a.box; l.segHead = a.segHead;
  l.segs = (const Seg*)ctx->gbuf[SEG_SLOT]; l.pool = (const uint64_t*)ctx->gbuf[POOL_SLOT]; l.outOff = outOff; l.oq = oq; l.ot = ot; l.eff = eff;
  lra_time_begin(ctx, "btwn_apply");
  hipLaunchKernelGGL(bt_layout, dim3((unsigned)slots), dim3(64), 0, st, l);
  lra_time_end(ctx);
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  LRA_HIP_CHECK(ctx, hipGetLastError());
  out->n_matches = NM; out->n_problems = totalProblems; out->n_pairs = totalPairs; out->n_rounds = (uint32_t)rounds.size();
  out->d_match_off = outOff; out->d_match_q = oq; out->d_match_t = ot; out->d_box = a.box; out->d_eff = eff; out->d_refinespace = a.rspace;
  return LRA_OK;
}
