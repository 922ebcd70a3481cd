// tools/micro/segsort.hip -- diagnostic: rocprim::segmented_radix_sort_pairs against one workgroup per segment with hipcub::BlockRadixSort (the segment's pairs sorted in
// LDS, one trip through memory) on segments like the path's (u64 key, u32 value; lengths around L).  hipcc --offload-arch=gfx950 -O3 -o segsort segsort.hip ; ./segsort
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NT, int IPT>
__global__ void __launch_bounds__(NT) k_block_sort(int nseg, const uint64_t* __restrict__ off, const uint64_t* __restrict__ kin, uint64_t* kout,
                                                   const uint32_t* __restrict__ vin, uint32_t* vout, int begin_bit, int end_bit, int minLen) {
  typedef hipcub::BlockRadixSort<uint64_t, NT, IPT, uint32_t> Sort;
  __shared__ typename Sort::TempStorage tmp;
  for (int s = blockIdx.x; s < nseg; s += gridDim.x) {
    const uint64_t b = off[s]; const int n = (int)(off[s + 1] - b);
    if (n > NT * IPT || n <= minLen) continue;
    uint64_t k[IPT]; uint32_t v[IPT];
#pragma unroll
    for (int i = 0; i < IPT; i++) { const int x = threadIdx.x * IPT + i; k[i] = x < n ? kin[b + x] : ~0ull; v[i] = x < n ? vin[b + x] : 0u; }
    Sort(tmp).Sort(k, v, begin_bit, end_bit);
#pragma unroll
    for (int i = 0; i < IPT; i++) { const int x = threadIdx.x * IPT + i; if (x < n) { kout[b + x] = k[i]; vout[b + x] = v[i]; } }
    __syncthreads();
  }
}

int main() {
  for (int L : {640, 1760, 3000, 6000}) {
    const int nseg = 32768;
    std::mt19937_64 rng(1);
    std::vector<uint64_t> off(nseg + 1, 0);
    for (int i = 0; i < nseg; i++) off[i + 1] = off[i] + (uint64_t)std::max(8.0, L * (0.6 + 0.8 * (rng() % 1000) / 1000.0));
    const uint64_t N = off[nseg];
    std::vector<uint64_t> key(N); std::vector<uint32_t> val(N);
    for (uint64_t i = 0; i < N; i++) { key[i] = ((rng() % 40000) << 32) | ((rng() % 40000) << 1) | (rng() & 1); val[i] = (uint32_t)i; }
    uint64_t *d_off, *d_k, *d_k2, *d_k3; uint32_t *d_v, *d_v2, *d_v3;
    CK(hipMalloc(&d_off, (nseg + 1) * 8)); CK(hipMalloc(&d_k, N * 8)); CK(hipMalloc(&d_k2, N * 8)); CK(hipMalloc(&d_k3, N * 8));
    CK(hipMalloc(&d_v, N * 4)); CK(hipMalloc(&d_v2, N * 4)); CK(hipMalloc(&d_v3, N * 4));
    CK(hipMemcpy(d_off, off.data(), (nseg + 1) * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_k, key.data(), N * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_v, val.data(), N * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int end_bit : {64, 49}) {
      size_t tb = 0;
      (void)rocprim::segmented_radix_sort_pairs(nullptr, tb, d_k, d_k2, d_v, d_v2, (unsigned)N, (unsigned)nseg, d_off, d_off + 1, 0, end_bit, 0);
      void* tmp; CK(hipMalloc(&tmp, tb + 256));
      float msR = 0, msB = 0;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, 0));
        (void)rocprim::segmented_radix_sort_pairs(tmp, tb, d_k, d_k2, d_v, d_v2, (unsigned)N, (unsigned)nseg, d_off, d_off + 1, 0, end_bit, 0);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&msR, e0, e1));
      }
      for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_block_sort<256, 4>), dim3(256 * 8), dim3(256), 0, 0, nseg, d_off, d_k, d_k3, d_v, d_v3, 0, end_bit, 0);
        hipLaunchKernelGGL((k_block_sort<256, 8>), dim3(256 * 6), dim3(256), 0, 0, nseg, d_off, d_k, d_k3, d_v, d_v3, 0, end_bit, 1024);
        hipLaunchKernelGGL((k_block_sort<512, 8>), dim3(256 * 3), dim3(512), 0, 0, nseg, d_off, d_k, d_k3, d_v, d_v3, 0, end_bit, 2048);
        if (getenv("V") && atoi(getenv("V")) == 1) hipLaunchKernelGGL((k_block_sort<512, 16>), dim3(256 * 2), dim3(512), 0, 0, nseg, d_off, d_k, d_k3, d_v, d_v3, 0, end_bit, 4096);
        else if (getenv("V") && atoi(getenv("V")) == 2) { hipLaunchKernelGGL((k_block_sort<768, 8>), dim3(256 * 2), dim3(768), 0, 0, nseg, d_off, d_k, d_k3, d_v, d_v3, 0, end_bit, 4096); hipLaunchKernelGGL((k_block_sort<1024, 8>), dim3(256 * 1), dim3(1024), 0, 0, nseg, d_off, d_k, d_k3, d_v, d_v3, 0, end_bit, 6144); }
        else if (getenv("V") && atoi(getenv("V")) == 3) hipLaunchKernelGGL((k_block_sort<256, 32>), dim3(256 * 2), dim3(256), 0, 0, nseg, d_off, d_k, d_k3, d_v, d_v3, 0, end_bit, 4096);
        else hipLaunchKernelGGL((k_block_sort<1024, 8>), dim3(256 * 1), dim3(1024), 0, 0, nseg, d_off, d_k, d_k3, d_v, d_v3, 0, end_bit, 4096);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&msB, e0, e1));
      }
      std::vector<uint64_t> a(N), b(N); std::vector<uint32_t> va(N), vb(N);
      CK(hipMemcpy(a.data(), d_k2, N * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), d_k3, N * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(va.data(), d_v2, N * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(vb.data(), d_v3, N * 4, hipMemcpyDeviceToHost));
      uint64_t bad = 0, big = 0;
      for (int s = 0; s < nseg; s++) { if (off[s + 1] - off[s] > 8192) { big++; continue; } for (uint64_t i = off[s]; i < off[s + 1]; i++) bad += (a[i] != b[i]) || (va[i] != vb[i]); }
      printf("L %5d  N %.1f M  end_bit %d : rocprim %.2f ms (%.1f G pairs/s)   block sort %.2f ms (%.1f G pairs/s)   mismatches %llu (segments beyond 8192: %llu)\n", L, N / 1e6, end_bit, msR,
             N / msR / 1e6, msB, N / msB / 1e6, (unsigned long long)bad, (unsigned long long)big);
      CK(hipFree(tmp));
    }
    CK(hipFree(d_off)); CK(hipFree(d_k)); CK(hipFree(d_k2)); CK(hipFree(d_k3)); CK(hipFree(d_v)); CK(hipFree(d_v2)); CK(hipFree(d_v3));
  }
  return 0;
}
