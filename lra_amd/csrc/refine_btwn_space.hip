// lra_amd/csrc/refine_btwn_space.hip -- SURVEY §8a row a11: RefineBtwnSpace (ClusterRefine.h:331-432; called by RefineBtwnClusters_chain :433 on
// the high-accuracy path) for a batch of spaces, up to the vector insert / SetClusterBoundariesFromMatches it ends with.  gfx950 only.
// Per space: RefineSpace on the cluster's strand; if that is enough (two-block spaces take anything, others need an efficiency of
// 2 * anchorstoosparse) its pairs are the answer; otherwise RefineSpace on the other strand, and the denser of the two wins -- the reverse
// one as a new RevBtwnCluster.  Both RefineSpace calls are lra_refine_space_batch over all spaces that need them; the decisions are
// elementwise kernels.
#include "common.h"
#include "scan.h"
#include <algorithm>

namespace {

struct BsArgs {
  int n;
  const uint32_t* qs; const uint32_t* qe; const uint32_t* ts; const uint32_t* te; const int32_t* st; const uint8_t* two; const uint32_t* read;
  const int32_t* chrom; const uint32_t* lrts; const uint32_t* lrlen;
  const uint64_t* read_off; uint64_t rc_base; const uint64_t* pos;
  int K, W, ontClr; float sparse2;
  // problems of the current pass (dense: pass 0 = all spaces, pass 1 = the spaces in idx[])
  const uint32_t* idx;
  uint64_t* pQoff; int32_t* pQlen; uint64_t* pToff; int32_t* pTlen; uint32_t* pTspan; int32_t* pK; int32_t* pW; int32_t* pDiag; uint32_t* pQadd; uint32_t* pTadd;
  uint32_t* pFlip;
  // per space
  uint32_t* span; uint32_t* need; float* eff; float* reff; int32_t* dec;
  const uint64_t* off0; const uint32_t* q0; const uint32_t* t0; const uint64_t* off1; const uint32_t* q1; const uint32_t* t1; const uint64_t* revPos;
  uint32_t* cnt; const uint64_t* outOff; uint32_t* outQ; uint32_t* outT;
};

template <int PASS>
__global__ void bs_plan(BsArgs a, int np) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= np) return;
  const uint32_t i = PASS == 0 ? (uint32_t)p : a.idx[p];
  const uint32_t r = a.read[i];
  const uint32_t readLen = (uint32_t)(a.read_off[r + 1] - a.read_off[r]);
  uint32_t qs = a.qs[i], qe = a.qe[i];
  const uint32_t ts = a.ts[i], te = a.te[i], lrts = a.lrts ? a.lrts[i] : 0, lrlen = a.lrlen ? a.lrlen[i] : 0;
  int st = a.st[i];
  if (st == 1) { const uint32_t t = qs; qs = readLen - qe; qe = readLen - t; }                          // :336-340
  int diag;                                                                                            // :341-350 (from the span on the cluster's strand)
  if (a.ontClr) diag = min((int)floorf(fmaxf(100.f, __fmul_rn(0.15f, (float)(qe - qs)))), 1000);
  else diag = min((int)floorf(fmaxf(100.f, __fmul_rn(0.01f, (float)(qe - qs)))), 100);
  if (PASS == 0) a.span[i] = min(qe - qs, te - ts);
  else { const uint32_t t = qs; qs = readLen - qe; qe = readLen - t; st = st == 1 ? 0 : 1; }            // :372-375
  a.pQoff[p] = (st ? a.rc_base : 0) + a.read_off[r] + qs; a.pQlen[p] = (int32_t)(qe - qs);
  a.pToff[p] = a.pos[a.chrom[i]] + (ts - lrts); a.pTlen[p] = (int32_t)(te - ts + lrlen); a.pTspan[p] = te - (ts - lrts);
  a.pK[p] = a.K; a.pW[p] = a.W; a.pDiag[p] = diag; a.pQadd[p] = qs; a.pTadd[p] = ts - lrts; a.pFlip[p] = st == 1 ? readLen : 0;
}

__global__ void bs_decide0(BsArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const uint32_t c = (uint32_t)(a.off0[i + 1] - a.off0[i]);
  const float eff = __fdiv_rn((float)c, (float)a.span[i]);                                              // :358
  a.eff[i] = eff; a.reff[i] = -1.f;
  int dec = 0; uint32_t need = 0;
  if ((c > 0 && a.two[i]) || (c > 0 && eff >= a.sparse2)) dec = 1;                                      // :360-368
  else if (!a.two[i]) need = 1;                                                                         // :371
  a.dec[i] = dec; a.need[i] = need;
  a.cnt[i] = dec == 1 ? c : 0;
}
__global__ void bs_compact(BsArgs a, uint32_t* idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < a.n && a.need[i]) idx[a.revPos[i]] = (uint32_t)i;
}
__global__ void bs_decide1(BsArgs a, int nrev) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nrev) return;
  const uint32_t i = a.idx[p];
  const uint32_t c0 = (uint32_t)(a.off0[i + 1] - a.off0[i]), c1 = (uint32_t)(a.off1[p + 1] - a.off1[p]);
  const float reff = __fdiv_rn((float)c1, (float)a.span[i]);                                            // :377 (the spans are those of the first call)
  a.reff[i] = reff;
  if (a.eff[i] >= reff) { a.dec[i] = 3; a.cnt[i] = c0; }                                                // :415-421
  else { a.dec[i] = 2; a.cnt[i] = c1; }                                                                 // :422-431
}
__global__ void __launch_bounds__(64) bs_emit(BsArgs a) {
  const int i = blockIdx.x;
  if (i >= a.n) return;
  const int dec = a.dec[i];
  if (dec == 0) return;
  const uint64_t o = a.outOff[i];
  const uint32_t c = a.cnt[i];
  const uint32_t* sq; const uint32_t* stt;
  if (dec == 2) { const uint64_t p = a.revPos[i]; sq = a.q1 + a.off1[p]; stt = a.t1 + a.off1[p]; }
  else { sq = a.q0 + a.off0[i]; stt = a.t0 + a.off0[i]; }
  for (uint32_t x = threadIdx.x; x < c; x += 64) { a.outQ[o + x] = sq[x]; a.outT[o + x] = stt[x]; }
}

}  // namespace

extern "C" int lra_refine_btwn_space_batch(lra_ctx* ctx, int n, const uint32_t* d_qs, const uint32_t* d_qe, const uint32_t* d_ts, const uint32_t* d_te,
                                           const int32_t* d_st, const uint8_t* d_twoblocks, const uint32_t* d_read, const int32_t* d_chrom, const uint32_t* d_lrts,
                                           const uint32_t* d_lrlength, const uint64_t* d_read_off, const char* d_strands, uint64_t rc_base, const char* d_genome,
                                           const uint64_t* h_chrom_pos, int n_chrom, int K, int W, int read_type, float anchorstoosparse, int match, int mismatch,
                                           int indel, int max_freq, lra_btwn_space_result* out) {
  if (!ctx || !out || n < 0 || !h_chrom_pos || n_chrom < 1) return LRA_ERR_INVALID;
  memset(out, 0, sizeof *out);
  out->n = (uint64_t)n;
  if (n == 0) return LRA_OK;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t n1 = (size_t)n + 2;
  char* w = (char*)lra_ensure(ctx, 78, al(n1 * 8) * 6 + al(n1 * 4) * 16 + al(((size_t)n_chrom + 1) * 8) + 4096);
  if (!w) return LRA_ERR_NOMEM;
  auto take = [&](size_t bytes) { char* r = w; w += al(bytes); return r; };
  BsArgs a; memset(&a, 0, sizeof a);
  a.n = n; a.qs = d_qs; a.qe = d_qe; a.ts = d_ts; a.te = d_te; a.st = d_st; a.two = d_twoblocks; a.read = d_read; a.chrom = d_chrom; a.lrts = d_lrts; a.lrlen = d_lrlength;
  a.read_off = d_read_off; a.rc_base = rc_base; a.K = K; a.W = W; a.ontClr = (read_type == LRA_READ_ONT || read_type == LRA_READ_CLR) ? 1 : 0;
  a.sparse2 = anchorstoosparse * 2;
  uint64_t* pos = (uint64_t*)take(((size_t)n_chrom + 1) * 8);
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(pos, h_chrom_pos, ((size_t)n_chrom + 1) * 8, hipMemcpyHostToDevice, st));
  a.pos = pos;
  a.pQoff = (uint64_t*)take(n1 * 8); a.pToff = (uint64_t*)take(n1 * 8);
  uint64_t* off0 = (uint64_t*)take(n1 * 8); uint64_t* revPos = (uint64_t*)take(n1 * 8); uint64_t* outOff = (uint64_t*)take(n1 * 8); uint64_t* off1 = (uint64_t*)take(n1 * 8);
  a.pQlen = (int32_t*)take(n1 * 4); a.pTlen = (int32_t*)take(n1 * 4); a.pTspan = (uint32_t*)take(n1 * 4); a.pK = (int32_t*)take(n1 * 4); a.pW = (int32_t*)take(n1 * 4);
  a.pDiag = (int32_t*)take(n1 * 4); a.pQadd = (uint32_t*)take(n1 * 4); a.pTadd = (uint32_t*)take(n1 * 4); a.pFlip = (uint32_t*)take(n1 * 4);
  a.span = (uint32_t*)take(n1 * 4); a.need = (uint32_t*)take(n1 * 4); a.eff = (float*)take(n1 * 4); a.reff = (float*)take(n1 * 4); a.dec = (int32_t*)take(n1 * 4);
  a.cnt = (uint32_t*)take(n1 * 4); uint32_t* idx = (uint32_t*)take(n1 * 4);
  const unsigned g = (unsigned)((n + 255) / 256);
  // ---- the cluster's strand
  hipLaunchKernelGGL(bs_plan<0>, dim3(g), dim3(256), 0, st, a, n);
  lra_refine_space_result r0;
  int rc = lra_refine_space_batch(ctx, n, d_strands, a.pQoff, a.pQlen, d_genome, a.pToff, a.pTlen, a.pTspan, a.pK, a.pW, a.pDiag, a.pQadd, a.pTadd, a.pFlip, match, mismatch,
                                  indel, max_freq, &r0);
  if (rc) return rc;
  // its pairs live in buffers the second RefineSpace pass reuses: keep them
  uint32_t* keep0 = (uint32_t*)lra_ensure(ctx, 79, al((r0.n_pairs + 1) * 4) * 2 + 512);
  if (!keep0) return LRA_ERR_NOMEM;
  uint32_t* q0 = keep0; uint32_t* t0 = (uint32_t*)((char*)keep0 + al((r0.n_pairs + 1) * 4));
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(off0, r0.d_pair_off, ((size_t)n + 1) * 8, hipMemcpyDeviceToDevice, st));
  if (r0.n_pairs) {
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(q0, r0.d_pair_q, r0.n_pairs * 4, hipMemcpyDeviceToDevice, st));
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(t0, r0.d_pair_t, r0.n_pairs * 4, hipMemcpyDeviceToDevice, st));
  }
  a.off0 = off0; a.q0 = q0; a.t0 = t0;
  hipLaunchKernelGGL(bs_decide0, dim3(g), dim3(256), 0, st, a);
  // ---- the other strand, where the first try was too sparse
  if ((rc = lra_exclusive_scan<uint32_t>(ctx, n, a.need, revPos))) return rc;
  uint64_t nrev = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&nrev, revPos + n, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  a.revPos = revPos;
  lra_refine_space_result r1; memset(&r1, 0, sizeof r1);
  if (nrev) {
    hipLaunchKernelGGL(bs_compact, dim3(g), dim3(256), 0, st, a, idx);
    a.idx = idx;
    hipLaunchKernelGGL(bs_plan<1>, dim3((unsigned)((nrev + 255) / 256)), dim3(256), 0, st, a, (int)nrev);
    if ((rc = lra_refine_space_batch(ctx, (int)nrev, d_strands, a.pQoff, a.pQlen, d_genome, a.pToff, a.pTlen, a.pTspan, a.pK, a.pW, a.pDiag, a.pQadd, a.pTadd, a.pFlip, match,
                                     mismatch, indel, max_freq, &r1))) return rc;
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(off1, r1.d_pair_off, ((size_t)nrev + 1) * 8, hipMemcpyDeviceToDevice, st));
    a.off1 = off1; a.q1 = r1.d_pair_q; a.t1 = r1.d_pair_t;
    hipLaunchKernelGGL(bs_decide1, dim3((unsigned)((nrev + 255) / 256)), dim3(256), 0, st, a, (int)nrev);
  }
  // ---- the pairs each space ends up with
  if ((rc = lra_exclusive_scan<uint32_t>(ctx, n, a.cnt, outOff))) return rc;
  uint64_t total = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&total, outOff + n, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  uint32_t* oq = (uint32_t*)lra_ensure(ctx, 71, al((total + 1) * 4) * 2 + 512);
  if (!oq) return LRA_ERR_NOMEM;
  uint32_t* ot = (uint32_t*)((char*)oq + al((total + 1) * 4));
  a.outOff = outOff; a.outQ = oq; a.outT = ot;
  hipLaunchKernelGGL(bs_emit, dim3((unsigned)n), dim3(64), 0, st, a);
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  LRA_HIP_CHECK(ctx, hipGetLastError());
  out->n_pairs = total; out->n_reverse_tried = nrev; out->d_pair_off = outOff; out->d_pair_q = oq; out->d_pair_t = ot; out->d_decision = a.dec; out->d_eff = a.eff;
  out->d_reff = a.reff;
  return LRA_OK;
}
