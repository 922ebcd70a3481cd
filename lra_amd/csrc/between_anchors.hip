// lra_amd/csrc/between_anchors.hip -- SURVEY §8a row a13, its DP leaf: RefineByLinearAlignment (LocalRefineAlignment.h:141-185) =
// SetMatchAndGaps (:93-98) + RefineSubstrings (:127-139) + AlignSubstrings (:100-125), i.e. the AffineOneGapAlign between two
// consecutive anchors of a chain with band  min(2 * |qLen - tLen| + 1, opts.localBand)  and its blocks shifted back to read / chromosome
// coordinates, for a batch of anchor pairs.  gfx950 only.  The DP itself is aog.hip; this file is the per-gap set-up and the compaction
// of the blocks into the order `alignment->blocks.insert(...)` (:156) produces.
#include "common.h"
#include "scan.h"

int lra_aog_launch_device(lra_ctx* ctx, int n, const char* d_qseq, const char* d_tseq, const uint64_t* d_q_off, const int32_t* d_q_len,
                          const uint64_t* d_t_off, const int32_t* d_t_len, const int32_t* d_k, int m, int mm, int indel, int32_t* d_score,
                          int32_t* d_nblocks, int32_t* d_blocks, const uint64_t* d_block_off, int32_t* d_status);

namespace {

struct BaArgs {
  int n;
  const uint64_t* q_base; const uint32_t* qs; const uint32_t* qe; const uint64_t* t_base; const uint32_t* ts; const uint32_t* te;
  int localBand, refineDp;
  uint64_t* q_off; int32_t* q_len; uint64_t* t_off; int32_t* t_len; int32_t* k; uint32_t* cap; uint32_t* status;
  const uint64_t* cap_off; const int32_t* nblocks; const int32_t* blocks; const int32_t* aogStatus;
  uint32_t* cnt; const uint64_t* out_off; int32_t* out;
};

__global__ void ba_setup(BaArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const uint32_t qs = a.qs[i], qe = a.qe[i], ts = a.ts[i], te = a.te[i];
  const int m = (int)min(qe - qs + 1u, te - ts + 1u);                     // Matched / SetMatchAndGaps :88-98 (GenomePos arithmetic)
  uint32_t st = 0;
  int qLen = 0, tLen = 0, k = 1;
  bool run = a.refineDp && m > 0;
  if (run) {
    qLen = (int)(qe - qs); tLen = (int)(te - ts);                         // AlignSubstrings :103-104
    if (qLen < 0 || tLen < 0) { st = LRA_ST_RANGE; run = false; qLen = tLen = 0; }   // std::string of negative length in the reference
    else { const int drift = abs(qLen - tLen); k = min(drift * 2 + 1, a.localBand); }
  }
  a.q_off[i] = a.q_base[i] + qs; a.t_off[i] = a.t_base[i] + ts;
  a.q_len[i] = run ? qLen : -1; a.t_len[i] = run ? tLen : -1; a.k[i] = k;   // q_len = -1: skipped by the compaction below
  a.cap[i] = (uint32_t)(min(qLen, tLen) + 2);                            // skipped gaps run as empty problems and are masked afterwards
  a.status[i] = st;
}

// the AOG batch runs on the gaps that are aligned at all; gaps with q_len = -1 get zero-length problems (no blocks)
__global__ void ba_fix_lengths(int n, int32_t* q_len, int32_t* t_len, uint8_t* live) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  live[i] = q_len[i] >= 0;
  if (q_len[i] < 0) { q_len[i] = 0; t_len[i] = 0; }
}

template <bool EMIT>
__global__ void ba_collect(BaArgs a, const uint8_t* live) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const int nb = live[i] ? a.nblocks[i] : 0;
  if (!EMIT) { a.cnt[i] = (uint32_t)nb; if (live[i] && a.aogStatus[i]) a.status[i] |= (uint32_t)a.aogStatus[i]; return; }
  const int32_t* B = a.blocks + 3 * a.cap_off[i];
  int32_t* O = a.out + 3 * a.out_off[i];
  const int32_t qs = (int32_t)a.qs[i], ts = (int32_t)a.ts[i];
  for (int b = 0; b < nb; b++) { O[3 * b] = B[3 * b] + qs; O[3 * b + 1] = B[3 * b + 1] + ts; O[3 * b + 2] = B[3 * b + 2]; }   // RefineSubstrings :134-138
}

inline size_t sz(size_t n, size_t e) { return (n * e + 255) / 256 * 256; }

}  // namespace

extern "C" int lra_between_anchors_batch(lra_ctx* ctx, int n, const char* d_qseq, const uint64_t* d_q_base, const uint32_t* d_cur_read_end,
                                         const uint32_t* d_next_read_start, const char* d_tseq, const uint64_t* d_t_base, const uint32_t* d_cur_genome_end,
                                         const uint32_t* d_next_genome_start, int match, int mismatch, int indel, int local_band, int refine_dp,
                                         lra_between_result* out) {
  if (!ctx || !out || n < 0) return LRA_ERR_INVALID;
  memset(out, 0, sizeof *out);
  out->n_gaps = (uint64_t)n;
  if (n == 0) return LRA_OK;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t n1 = (size_t)n + 2;
  auto take = [](char*& p, size_t cnt, size_t e) { char* r = p; p += sz(cnt, e); return r; };
  char* w = (char*)lra_ensure(ctx, 18, sz(n1, 8) * 4 + sz(n1, 4) * 8 + sz(n1, 1) + 4096);
  if (!w) return LRA_ERR_NOMEM;
  BaArgs a;
  memset(&a, 0, sizeof a);
  a.n = n; a.q_base = d_q_base; a.qs = d_cur_read_end; a.qe = d_next_read_start; a.t_base = d_t_base; a.ts = d_cur_genome_end; a.te = d_next_genome_start;
  a.localBand = local_band; a.refineDp = refine_dp;
  a.q_off = (uint64_t*)take(w, n1, 8); a.t_off = (uint64_t*)take(w, n1, 8); uint64_t* cap_off = (uint64_t*)take(w, n1, 8); uint64_t* out_off = (uint64_t*)take(w, n1, 8);
  a.q_len = (int32_t*)take(w, n1, 4); a.t_len = (int32_t*)take(w, n1, 4); a.k = (int32_t*)take(w, n1, 4); a.cap = (uint32_t*)take(w, n1, 4);
  a.cnt = (uint32_t*)take(w, n1, 4); int32_t* score = (int32_t*)take(w, n1, 4); int32_t* nblocks = (int32_t*)take(w, n1, 4); int32_t* ast = (int32_t*)take(w, n1, 4);
  uint8_t* live = (uint8_t*)take(w, n1, 1);
  char* r = (char*)lra_ensure(ctx, 19, sz(n1, 8) + sz(n1, 4) * 2 + 4096);
  if (!r) return LRA_ERR_NOMEM;
  uint64_t* offOut = (uint64_t*)take(r, n1, 8); uint32_t* status = (uint32_t*)take(r, n1, 4); int32_t* scoreOut = (int32_t*)take(r, n1, 4);
  a.status = status;
  const unsigned g = (unsigned)((n + 255) / 256);
  lra_time_begin(ctx, "between_anchors");
  hipLaunchKernelGGL(ba_setup, dim3(g), dim3(256), 0, st, a);
  hipLaunchKernelGGL(ba_fix_lengths, dim3(g), dim3(256), 0, st, n, a.q_len, a.t_len, live);
  lra_time_end(ctx);
  { int rc = lra_exclusive_scan<uint32_t>(ctx, n, a.cap, cap_off); if (rc) return rc; }
  uint64_t totalCap = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&totalCap, cap_off + n, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  int32_t* blocks = (int32_t*)lra_ensure(ctx, 20, (3 * totalCap + 3) * 4 + 256);
  if (!blocks) return LRA_ERR_NOMEM;
  // gaps that are not aligned have capacity 0 and zero lengths: AffineOneGapAlign on them must not write; they are masked by `live`
  LRA_HIP_CHECK(ctx, hipMemsetAsync(nblocks, 0, (size_t)n * 4, st));
  {
    int rc = lra_aog_launch_device(ctx, n, d_qseq, d_tseq, a.q_off, a.q_len, a.t_off, a.t_len, a.k, match, mismatch, indel, score, nblocks, blocks, cap_off, ast);
    if (rc) return rc;
  }
  a.cap_off = cap_off; a.nblocks = nblocks; a.blocks = blocks; a.aogStatus = ast;
  lra_time_begin(ctx, "between_anchors");
  hipLaunchKernelGGL(ba_collect<false>, dim3(g), dim3(256), 0, st, a, live);
  lra_time_end(ctx);
  { int rc = lra_exclusive_scan<uint32_t>(ctx, n, a.cnt, out_off); if (rc) return rc; }
  uint64_t nBlocks = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&nBlocks, out_off + n, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  int32_t* outB = (int32_t*)lra_ensure(ctx, 21, (3 * nBlocks + 3) * 4 + 256);
  if (!outB) return LRA_ERR_NOMEM;
  a.out_off = out_off; a.out = outB;
  lra_time_begin(ctx, "between_anchors");
  hipLaunchKernelGGL(ba_collect<true>, dim3(g), dim3(256), 0, st, a, live);
  lra_time_end(ctx);
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(offOut, out_off, ((size_t)n + 1) * 8, hipMemcpyDeviceToDevice, st));
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(scoreOut, score, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  LRA_HIP_CHECK(ctx, hipGetLastError());
  out->n_blocks = nBlocks; out->d_block_off = offOut; out->d_blocks = outB; out->d_status = status; out->d_score = scoreOut;
  return LRA_OK;
}
