// lra_amd/csrc/map_merge.h -- merging the lra_map_result arrays of two passes over (disjoint subsets of) one batch into one result (gfx950 only).
//
// Both drivers run some reads of a batch as a second, small batch: the high-accuracy path the reads that take the REFINEclusters branch
// (mapread_highacc.hip), the low-accuracy path the reads whose chains are far larger than the batch's typical ones, on a second stream beside the
// rest of the batch (mapread.hip).  Job slot s of the merged result comes from slot srcSlot[s] of pass A (bit 63 clear) or of pass B (bit 63 set).
#pragma once
#include "common.h"
#include "scan.h"

namespace lra_merge {

struct PassView {
  const uint64_t* jo; const int32_t* strand; const int32_t* supp; const int32_t* sec; const int32_t* n0; const int32_t* n1; const int32_t* chrom; const float* fval;
  const uint64_t* boff; const int32_t* blocks; const int32_t* rstat; const int32_t* counts; const float* value; const uint64_t* roff; const uint32_t* runs;
  const uint32_t* jstat;
};
constexpr uint64_t FROM_B = 1ull << 63;
inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

inline PassView view_of(const lra_map_result& r) {
  PassView V;
  V.jo = r.n_alignments ? r.d_job_aln_off : nullptr;
  V.strand = r.d_strand; V.supp = r.d_supp; V.sec = r.d_secondary; V.n0 = r.d_n0; V.n1 = r.d_n1; V.chrom = r.d_chrom; V.fval = r.d_first_sdp_value;
  V.boff = r.d_block_off; V.blocks = r.d_blocks; V.rstat = r.d_refine_status; V.counts = r.d_counts; V.value = r.d_value; V.roff = r.d_run_off; V.runs = r.d_runs;
  V.jstat = r.d_job_status;
  return V;
}

static __global__ void k_merge_count(uint64_t S, const uint64_t* __restrict__ srcSlot, PassView A, PassView B, uint32_t* nAln, uint32_t* jstat) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  const uint64_t x = srcSlot[s], j = x & ~FROM_B;
  const PassView& V = (x & FROM_B) ? B : A;
  nAln[s] = V.jo ? (uint32_t)(V.jo[j + 1] - V.jo[j]) : 0;
  jstat[s] = V.jstat ? V.jstat[j] : 0;
}
static __global__ void k_merge_fields(uint64_t S, int na, const uint64_t* __restrict__ srcSlot, PassView A, PassView B, const uint64_t* __restrict__ JO, uint32_t* alnRead,
                                      int32_t* strand, int32_t* supp, int32_t* sec, int32_t* n0, int32_t* n1, int32_t* chrom, float* fval, int32_t* rstat, int32_t* counts,
                                      float* value, uint64_t* srcAln, uint32_t* nb, uint32_t* nr) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  const uint64_t x = srcSlot[s], j = x & ~FROM_B;
  const PassView& V = (x & FROM_B) ? B : A;
  const uint64_t n = JO[s + 1] - JO[s];
  for (uint64_t i = 0; i < n; i++) {
    const uint64_t a = V.jo[j] + i, d = JO[s] + i;
    alnRead[d] = (uint32_t)(s / (uint64_t)na);
    strand[d] = V.strand[a]; supp[d] = V.supp[a]; sec[d] = V.sec[a]; n0[d] = V.n0[a]; n1[d] = V.n1[a]; chrom[d] = V.chrom[a]; fval[d] = V.fval[a];
    rstat[d] = V.rstat ? V.rstat[a] : 0; value[d] = V.value[a];
    for (int k = 0; k < 18; k++) counts[18 * d + k] = V.counts[18 * a + k];
    srcAln[d] = a | (x & FROM_B);
    nb[d] = (uint32_t)(V.boff[a + 1] - V.boff[a]); nr[d] = (uint32_t)(V.roff[a + 1] - V.roff[a]);
  }
}
static __global__ void __launch_bounds__(256) k_merge_payload(uint64_t nA, const uint64_t* __restrict__ srcAln, PassView A, PassView B, const uint64_t* __restrict__ BO,
                                                              const uint64_t* __restrict__ RO, int32_t* blocks, uint32_t* runs) {
  const uint64_t d = blockIdx.x;
  if (d >= nA) return;
  const uint64_t x = srcAln[d], a = x & ~FROM_B;
  const PassView& V = (x & FROM_B) ? B : A;
  const uint64_t b0 = V.boff[a], nbk = V.boff[a + 1] - b0, r0 = V.roff[a], nrn = V.roff[a + 1] - r0;
  for (uint64_t i = threadIdx.x; i < 3 * nbk; i += blockDim.x) blocks[3 * BO[d] + i] = V.blocks[3 * b0 + i];
  for (uint64_t i = threadIdx.x; i < nrn; i += blockDim.x) runs[RO[d] + i] = V.runs[r0 + i];
}
static __global__ void __launch_bounds__(64) k_gather_reads(int n, const uint32_t* __restrict__ pick, const uint64_t* __restrict__ off, const char* __restrict__ seq,
                                                            const uint64_t* __restrict__ newOff, char* out) {
  const int i = blockIdx.x;
  if (i >= n) return;
  const uint64_t a = off[pick[i]], m = off[pick[i] + 1] - a, d = newOff[i];
  for (uint64_t k = threadIdx.x; k < m; k += 64) out[d + k] = seq[a + k];
}

// The per-job and per-alignment arrays of the merged result, carved from the context's buffer `slot` (+ `extra` bytes behind them for the caller: *extra_at).  Sets
// out's n_jobs / n_alignments / n_blocks / n_runs and every d_* pointer except d_job_reached, d_read_status, d_strands / rc_base.  A's and B's arrays may live in any
// context of the device; they must not live in `slot` of ctx.
static inline int merge_passes(lra_ctx* ctx, int slot, uint64_t S, int na, const uint64_t* d_src, PassView A, PassView B, uint64_t nA, uint64_t nBk, uint64_t nRn,
                               size_t extra, char** extra_at, lra_map_result* out) {
  hipStream_t st = ctx->stream;
  auto grid = [](uint64_t n) { return dim3((unsigned)((n + 255) / 256)); };
  char* mg = (char*)lra_ensure(ctx, slot, 2 * al256((S + 2) * 4) + al256((S + 2) * 8) + 12 * al256((nA + 2) * 4) + 3 * al256((nA + 2) * 8) + al256((nA + 1) * 72) + al256((nBk + 1) * 12) +
                                          al256((nRn + 1) * 4) + al256(extra + 8) + 8192);
  if (!mg) return LRA_ERR_NOMEM;
  auto take = [&](size_t bytes) { char* r_ = mg; mg += al256(bytes + 8); return r_; };
  uint32_t* nAln = (uint32_t*)take((S + 1) * 4); uint64_t* JO = (uint64_t*)take((S + 1) * 8); uint32_t* jstat = (uint32_t*)take((S + 1) * 4);
  uint32_t* alnRead = (uint32_t*)take(nA * 4); int32_t* mstrand = (int32_t*)take(nA * 4); int32_t* msupp = (int32_t*)take(nA * 4); int32_t* msec = (int32_t*)take(nA * 4);
  int32_t* mn0 = (int32_t*)take(nA * 4); int32_t* mn1 = (int32_t*)take(nA * 4); int32_t* mchrom = (int32_t*)take(nA * 4); float* mfval = (float*)take(nA * 4);
  int32_t* mrstat = (int32_t*)take(nA * 4); float* mvalue = (float*)take(nA * 4); uint32_t* nb = (uint32_t*)take(nA * 4); uint32_t* nr = (uint32_t*)take(nA * 4);
  uint64_t* srcAln = (uint64_t*)take(nA * 8); uint64_t* BO = (uint64_t*)take((nA + 1) * 8); uint64_t* RO = (uint64_t*)take((nA + 1) * 8);
  int32_t* mcounts = (int32_t*)take(nA * 72); int32_t* mblocks = (int32_t*)take(nBk * 12); uint32_t* mruns = (uint32_t*)take(nRn * 4);
  if (extra_at) *extra_at = take(extra);
  int rc;
  hipLaunchKernelGGL(k_merge_count, grid(S), dim3(256), 0, st, S, d_src, A, B, nAln, jstat);
  if ((rc = lra_exclusive_scan<uint32_t>(ctx, (long)S, nAln, JO))) return rc;
  hipLaunchKernelGGL(k_merge_fields, grid(S), dim3(256), 0, st, S, na, d_src, A, B, (const uint64_t*)JO, alnRead, mstrand, msupp, msec, mn0, mn1, mchrom, mfval, mrstat,
                     mcounts, mvalue, srcAln, nb, nr);
  if (nA) {
    if ((rc = lra_exclusive_scan<uint32_t>(ctx, (long)nA, nb, BO)) || (rc = lra_exclusive_scan<uint32_t>(ctx, (long)nA, nr, RO))) return rc;
    hipLaunchKernelGGL(k_merge_payload, dim3((unsigned)nA), dim3(256), 0, st, nA, (const uint64_t*)srcAln, A, B, (const uint64_t*)BO, (const uint64_t*)RO, mblocks, mruns);
  } else { LRA_HIP_CHECK(ctx, hipMemsetAsync(BO, 0, 8, st)); LRA_HIP_CHECK(ctx, hipMemsetAsync(RO, 0, 8, st)); }
  out->n_jobs = S; out->n_alignments = nA; out->n_blocks = nBk; out->n_runs = nRn;
  out->d_job_aln_off = JO; out->d_job_status = jstat;
  out->d_aln_read = alnRead; out->d_strand = mstrand; out->d_supp = msupp; out->d_secondary = msec; out->d_n0 = mn0; out->d_n1 = mn1; out->d_chrom = mchrom;
  out->d_first_sdp_value = mfval; out->d_block_off = BO; out->d_blocks = mblocks; out->d_refine_status = mrstat; out->d_counts = mcounts; out->d_value = mvalue;
  out->d_run_off = RO; out->d_runs = mruns;
  return LRA_OK;
}

}  // namespace lra_merge
