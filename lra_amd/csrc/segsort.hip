// lra_amd/csrc/segsort.hip -- the path's segmented sorts of (64-bit key, 32-bit value) pairs: DiagonalSort / CartesianSort of a read's (a cluster's, a job's) matches
// (Sorting.h:50-150: any stable order by the packed key), the sparse DP's point orders where the keys of a list are all different (seed.hip: lra_sort_mostly_unique_batch),
// its diagonal order.  gfx950 only.
//
// rocprim::segmented_radix_sort_pairs sorts a segment of a few thousand pairs in ~10 digit passes THROUGH MEMORY (key / value arrays read and written per pass: 10-15 G
// pairs/s on this device, 480 bytes of traffic per pair).  Here a segment of 257 .. 8192 pairs is one workgroup's: loaded once, sorted in LDS (hipcub::BlockRadixSort, the
// same least-significant-digit radix sort -- stable, only the bits [begin_bit, end_bit) compared --, four size classes so that a short segment does not pay for a long one's
// padding), stored once: 21-27 G pairs/s (tools/micro/segsort.hip: identical output on 32768 segments of 640 / 1760 / 3000 / 6000 pairs).  The segments outside that range --
// the many tiny ones (a13's jobs: rocprim's warp sorts serve them well) and the rare ones beyond a workgroup's LDS -- stay with rocprim, which is called on a copy of the
// offsets in which every other segment is empty.  LRA_SEGSORT=0: rocprim for everything (comparisons).
#include "common.h"
#include <cstring>
#include <cstdlib>
#include <rocprim/rocprim.hpp>
#include <hipcub/hipcub.hpp>

namespace {

constexpr int SEG_LO = 256, SEG_HI = 8192;     // a workgroup sorts segments of SEG_LO < n <= SEG_HI pairs

// (also counts the segments of every size class -- (256, 1024], (1024, 2048], (2048, 4096], (4096, 8192] -- so that a class's launch with nothing to sort ends at once)
__global__ void k_mask_offsets(unsigned nseg, const uint64_t* __restrict__ b, const uint64_t* __restrict__ e, uint64_t* mb, uint64_t* me, unsigned* classCnt) {
  const unsigned s = blockIdx.x * blockDim.x + threadIdx.x;
  int cls = -1;
  if (s < nseg) {
    const uint64_t x = b[s], y = e[s], n = y > x ? y - x : 0;
    const bool mine = n > (uint64_t)SEG_LO && n <= (uint64_t)SEG_HI;
    mb[s] = x; me[s] = mine ? x : y;
    if (mine) cls = n <= 1024 ? 0 : n <= 2048 ? 1 : n <= 4096 ? 2 : 3;
  }
  for (int c = 0; c < 4; c++) { const unsigned long long m = __ballot(cls == c); if (m && (threadIdx.x & 63) == 0) atomicAdd(&classCnt[c], (unsigned)__popcll(m)); }
}

template <int NT, int IPT>
__global__ void __launch_bounds__(NT) k_block_sort(unsigned nseg, const uint64_t* __restrict__ b, const uint64_t* __restrict__ e, const uint64_t* __restrict__ kin, uint64_t* kout,
                                                   const uint32_t* __restrict__ vin, uint32_t* vout, int begin_bit, int end_bit, int minLen, const unsigned* __restrict__ classCnt) {
  typedef hipcub::BlockRadixSort<uint64_t, NT, IPT, uint32_t> Sort;
  __shared__ typename Sort::TempStorage tmp;
  if (*classCnt == 0) return;                                              // (no segment of this size class in the call)
  for (unsigned s = blockIdx.x; s < nseg; s += gridDim.x) {
    const uint64_t x = b[s], y = e[s];
    const int n = y > x ? (int)std::min<uint64_t>(y - x, (uint64_t)SEG_HI + 1) : 0;
    if (n > NT * IPT || n <= minLen) continue;                           // (another class's, or rocprim's)
    uint64_t k[IPT]; uint32_t v[IPT];
#pragma unroll
    for (int i = 0; i < IPT; i++) { const int p = (int)threadIdx.x * IPT + i; k[i] = p < n ? kin[x + p] : ~0ull; v[i] = p < n ? vin[x + p] : 0u; }
    Sort(tmp).Sort(k, v, begin_bit, end_bit);                             // (blocked arrangement in and out: position = thread * IPT + item; the padding sorts behind every pair)
#pragma unroll
    for (int i = 0; i < IPT; i++) { const int p = (int)threadIdx.x * IPT + i; if (p < n) { kout[x + p] = k[i]; vout[x + p] = v[i]; } }
    __syncthreads();
  }
}

}  // namespace

// The interface of rocprim::segmented_radix_sort_pairs (temp == nullptr: temp_bytes is set to what the call needs).  As there, the segments must not overlap (each
// workgroup reads its segment whole, then writes it: an overlapping neighbour would read half-written pairs when kin == kout is ever allowed; the entry point below
// rejects in-place calls).
hipError_t lra_segsort_pairs(lra_ctx* ctx, void* temp, size_t& temp_bytes, const uint64_t* kin, uint64_t* kout, const uint32_t* vin, uint32_t* vout, unsigned int total,
                             unsigned int nseg, const uint64_t* b, const uint64_t* e, int begin_bit, int end_bit, hipStream_t st) {
  static const bool off = getenv("LRA_SEGSORT") && getenv("LRA_SEGSORT")[0] == '0';
  size_t rb = 0;
  hipError_t rc = rocprim::segmented_radix_sort_pairs(nullptr, rb, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, total, nseg, (uint64_t*)nullptr,
                                                      (uint64_t*)nullptr, begin_bit, end_bit, st);
  if (rc != hipSuccess) return rc;
  const size_t rbA = (rb + 255) & ~(size_t)255, extra = 2 * ((size_t)nseg + 1) * 8 + 256 + 64;
  if (!temp) { temp_bytes = rbA + extra; return hipSuccess; }
  if (temp_bytes < rbA + extra) return hipErrorInvalidValue;
  if (off || nseg == 0 || total == 0)
    return rocprim::segmented_radix_sort_pairs(temp, rb, kin, kout, vin, vout, total, nseg, b, e, begin_bit, end_bit, st);
  uint64_t* mb = (uint64_t*)((char*)temp + rbA); uint64_t* me = mb + nseg + 1;
  unsigned* cc = (unsigned*)(me + nseg + 1);
  if (hipMemsetAsync(cc, 0, 16, st) != hipSuccess) return hipErrorUnknown;
  hipLaunchKernelGGL(k_mask_offsets, dim3((nseg + 255) / 256), dim3(256), 0, st, nseg, b, e, mb, me, cc);
  const unsigned cu = (unsigned)ctx->num_cu;
  hipLaunchKernelGGL((k_block_sort<256, 4>), dim3(std::min(nseg, cu * 8)), dim3(256), 0, st, nseg, b, e, kin, kout, vin, vout, begin_bit, end_bit, SEG_LO, (const unsigned*)(cc + 0));
  hipLaunchKernelGGL((k_block_sort<256, 8>), dim3(std::min(nseg, cu * 6)), dim3(256), 0, st, nseg, b, e, kin, kout, vin, vout, begin_bit, end_bit, 1024, (const unsigned*)(cc + 1));
  hipLaunchKernelGGL((k_block_sort<512, 8>), dim3(std::min(nseg, cu * 3)), dim3(512), 0, st, nseg, b, e, kin, kout, vin, vout, begin_bit, end_bit, 2048, (const unsigned*)(cc + 2));
  hipLaunchKernelGGL((k_block_sort<1024, 8>), dim3(std::min(nseg, cu)), dim3(1024), 0, st, nseg, b, e, kin, kout, vin, vout, begin_bit, end_bit, 4096, (const unsigned*)(cc + 3));
  return rocprim::segmented_radix_sort_pairs(temp, rb, kin, kout, vin, vout, total, nseg, (const uint64_t*)mb, (const uint64_t*)me, begin_bit, end_bit, st);
}

extern "C" int lra_sort_pairs_batch(lra_ctx* ctx, uint64_t n_pairs, uint64_t n_segments, const uint64_t* d_begin, const uint64_t* d_end, const uint64_t* d_key_in,
                                    uint64_t* d_key_out, const uint32_t* d_val_in, uint32_t* d_val_out, int begin_bit, int end_bit) {
  if (!ctx || begin_bit < 0 || end_bit > 64 || begin_bit >= end_bit) return LRA_ERR_INVALID;
  if (n_pairs >= (1ull << 32) || n_segments >= (1ull << 32)) return lra_set_err(ctx, LRA_ERR_INVALID, "at most 2^32 - 1 pairs and segments per call");
  if (n_pairs == 0 || n_segments == 0) return LRA_OK;
  if (!d_begin || !d_end || !d_key_in || !d_key_out || !d_val_in || !d_val_out || d_key_in == d_key_out || d_val_in == d_val_out) return LRA_ERR_INVALID;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  size_t tb = 0;
  if (lra_segsort_pairs(ctx, nullptr, tb, nullptr, nullptr, nullptr, nullptr, (unsigned int)n_pairs, (unsigned int)n_segments, nullptr, nullptr, begin_bit, end_bit, ctx->stream) != hipSuccess)
    return lra_set_err(ctx, LRA_ERR_HIP, "segmented sort: sizing");
  void* temp = lra_scratch(ctx, 2, tb + 256);
  if (!temp) return LRA_ERR_NOMEM;
  const hipError_t e = lra_segsort_pairs(ctx, temp, tb, d_key_in, d_key_out, d_val_in, d_val_out, (unsigned int)n_pairs, (unsigned int)n_segments, d_begin, d_end, begin_bit, end_bit,
                                         ctx->stream);
  if (e != hipSuccess) return lra_set_err(ctx, LRA_ERR_HIP, "segmented sort: %s", hipGetErrorString(e));
  LRA_HIP_CHECK(ctx, hipGetLastError());
  return LRA_OK;
}
