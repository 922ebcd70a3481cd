// lra_amd/csrc/scan.h -- exclusive prefix sums (counts -> CSR offsets, off[n] = total) for the
// batch bookkeeping.  Three phases: per-tile reductions, one-block scan of the tile sums, per-tile
// rescans.  Included by the .hip files that need it (kernels are TU-local).
#pragma once
#include "common.h"

namespace { namespace lra_scan_detail {

constexpr int NT = 256;

template <typename CT>
__global__ void __launch_bounds__(NT) tile_sum(long n, long tile, const CT* __restrict__ c, uint64_t* __restrict__ part) {
  __shared__ uint64_t red[NT / 64];
  const long lo = (long)blockIdx.x * tile, hi = (lo + tile < n) ? lo + tile : n;
  uint64_t s = 0;
  for (long i = lo + threadIdx.x; i < hi; i += NT) s += (uint64_t)c[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) { uint64_t t = 0; for (int w = 0; w < NT / 64; w++) t += red[w]; part[blockIdx.x] = t; }
}

// exclusive scan of `nb` tile sums in place, one block; part[nb] = total
static __global__ void __launch_bounds__(1024) part_scan(long nb, uint64_t* part) {
  __shared__ uint64_t sh[1024];
  const int t = threadIdx.x;
  const long per = (nb + 1023) / 1024;
  const long lo = (long)t * per < nb ? (long)t * per : nb, hi = lo + per < nb ? lo + per : nb;
  uint64_t s = 0;
  for (long i = lo; i < hi; i++) s += part[i];
  sh[t] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    uint64_t v = (t >= d) ? sh[t - d] : 0;
    __syncthreads();
    sh[t] += v;
    __syncthreads();
  }
  uint64_t run = (t == 0) ? 0 : sh[t - 1];
  for (long i = lo; i < hi; i++) { uint64_t v = part[i]; part[i] = run; run += v; }
  if (t == 1023) part[nb] = sh[1023];
}

template <typename CT>
__global__ void __launch_bounds__(NT) tile_scan(long n, long tile, const CT* __restrict__ c, const uint64_t* __restrict__ part,
                                               uint64_t* __restrict__ off, long nb) {
  __shared__ uint64_t wsum[NT / 64];
  const long lo = (long)blockIdx.x * tile, hi = (lo + tile < n) ? lo + tile : n;
  uint64_t base = part[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long r = lo; r < hi; r += NT) {
    const long i = r + threadIdx.x;
    const uint64_t v = (i < hi) ? (uint64_t)c[i] : 0;
    uint64_t inc = v;
    for (int d = 1; d < 64; d <<= 1) { uint64_t o = __shfl_up(inc, d); if (lane >= d) inc += o; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint64_t wb = 0, tot = 0;
    for (int w = 0; w < NT / 64; w++) { if (w < wave) wb += wsum[w]; tot += wsum[w]; }
    if (i < hi) off[i] = base + wb + inc - v;
    base += tot;
    __syncthreads();
  }
  if (blockIdx.x == nb - 1 && threadIdx.x == 0) off[n] = part[nb];
}

} }  // namespace lra_scan_detail

// off[0..n] = exclusive prefix sums of counts[0..n) (async on the context's stream)
template <typename CT>
static int lra_exclusive_scan(lra_ctx* ctx, long n, const CT* counts, uint64_t* off) {
  using namespace lra_scan_detail;
  if (n <= 0) { return hipMemsetAsync(off, 0, 8, ctx->stream) == hipSuccess ? LRA_OK : LRA_ERR_HIP; }
  const long max_tiles = 65536;
  long tile = 4096;
  if ((n + tile - 1) / tile > max_tiles) tile = ((n + max_tiles - 1) / max_tiles + NT - 1) / NT * NT;
  const long nb = (n + tile - 1) / tile;
  if (!ctx->scan_tmp) {
    if (hipMalloc((void**)&ctx->scan_tmp, (max_tiles + 2) * sizeof(uint64_t)) != hipSuccess) return lra_set_err(ctx, LRA_ERR_NOMEM, "scan scratch");
  }
  hipLaunchKernelGGL(tile_sum<CT>, dim3((unsigned)nb), dim3(NT), 0, ctx->stream, n, tile, counts, ctx->scan_tmp);
  hipLaunchKernelGGL(part_scan, dim3(1), dim3(1024), 0, ctx->stream, nb, ctx->scan_tmp);
  hipLaunchKernelGGL(tile_scan<CT>, dim3((unsigned)nb), dim3(NT), 0, ctx->stream, n, tile, counts, (const uint64_t*)ctx->scan_tmp, off, nb);
  return LRA_OK;
}
