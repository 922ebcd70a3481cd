// tools/micro/lds_dma.hip -- where global_load_lds_dwordx4 (gfx950) puts a lane's 16 bytes: LDS base (M0) + lane * 16, lanes that are switched off write nothing.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/micro/lds_dma.hip -o /tmp/lds_dma && /tmp/lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const uint4* __restrict__ src, const int* __restrict__ idx, uint4* out, int nact) {
  __shared__ uint4 buf[3][36];
  const int lane = threadIdx.x;
  for (int i = lane; i < 3 * 36; i += 64) ((uint4*)buf)[i] = make_uint4(0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu);
  __syncthreads();
  if (lane < nact) {
    const uint4* p = src + 3 * idx[lane];
    for (int s = 0; s < 3; s++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + s), (__attribute__((address_space(3))) void*)&buf[s][0], 16, 0, 0);
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int s = 0; s < 3; s++) if (lane < 36) out[s * 64 + lane] = buf[s][lane];
}
int main() {
  const int N = 1000;
  std::vector<uint4> h(3 * N); for (int i = 0; i < 3 * N; i++) h[i] = make_uint4(i, i * 7 + 1, i * 13 + 2, i * 29 + 3);
  std::vector<int> hi(64); for (int i = 0; i < 64; i++) hi[i] = (i * 37 + 11) % N;
  uint4 *d, *o; int* di;
  hipMalloc(&d, h.size() * 16); hipMalloc(&o, 3 * 64 * 16); hipMalloc(&di, 64 * 4);
  hipMemcpy(d, h.data(), h.size() * 16, hipMemcpyHostToDevice); hipMemcpy(di, hi.data(), 256, hipMemcpyHostToDevice);
  int bad = 0;
  for (int nact : {36, 20, 1}) {
    hipMemset(o, 0, 3 * 64 * 16);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, di, o, nact);
    std::vector<uint4> r(3 * 64); hipMemcpy(r.data(), o, r.size() * 16, hipMemcpyDeviceToHost);
    for (int s = 0; s < 3; s++) for (int l = 0; l < 36; l++) {
      const uint4 g = r[s * 64 + l]; const uint4 e = l < nact ? h[3 * hi[l] + s] : make_uint4(0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu);
      if (g.x != e.x || g.y != e.y || g.z != e.z || g.w != e.w) { if (bad < 10) printf("nact %d slot %d lane %d: got %x %x %x %x want %x %x %x %x\n", nact, s, l, g.x, g.y, g.z, g.w, e.x, e.y, e.z, e.w); bad++; }
    }
  }
  printf(bad ? "lds_dma: %d mismatches\n" : "lds_dma: ok (lane * 16 from the base, inactive lanes untouched)\n", bad);
  return bad != 0;
}
