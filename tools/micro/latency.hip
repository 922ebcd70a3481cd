// tools/micro/latency.hip -- diagnostic: dependent-load latency on the device as a function of the footprint the loads are spread over (TLB reach), alone and
// with a second kernel issuing independent random loads beside it.  hipcc --offload-arch=gfx950 -O2 -o latency latency.hip ; ./latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
// a chain through `n` 64-byte lines spread with stride over the buffer: line i holds the index of the next line (a full-period LCG over n = 2^k)
__global__ void k_init(uint64_t* buf, uint64_t n, uint64_t strideWords) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t nx = (i * 6364136223846793005ull + 1442695040888963407ull) & (n - 1);
  buf[i * strideWords] = nx * strideWords;
}
__global__ void k_chase2(const uint64_t* __restrict__ buf, uint64_t n, uint64_t strideWords, int steps, uint64_t* out, long long* cyc) {
  if (threadIdx.x != 0) return;
  uint64_t p = (((uint64_t)blockIdx.x * 2654435761ull) & (n - 1)) * strideWords;
  const long long t0 = wall_clock64();
  for (int s = 0; s < steps; s++) p = __builtin_nontemporal_load(buf + p);
  const long long t1 = wall_clock64();
  out[blockIdx.x] = p; cyc[blockIdx.x] = t1 - t0;
}
// background load: every lane independent random 16-byte loads over the same footprint
__global__ void k_noise(const uint64_t* __restrict__ buf, uint64_t n, uint64_t strideWords, int iters, uint64_t* out) {
  uint64_t x = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
  uint64_t acc = 0;
  for (int i = 0; i < iters; i++) {
    x = x * 6364136223846793005ull + 1442695040888963407ull;
    acc += buf[((x >> 20) & (n - 1)) * strideWords];
  }
  if (acc == 42) out[0] = acc;
}
int main(int argc, char** argv) {
  const double gbs[] = {0.0625, 1, 8, 32, 128, 200};
  size_t freeB = 0, totB = 0; CK(hipMemGetInfo(&freeB, &totB));
  printf("free %.1f GB of %.1f\n", freeB / 1e9, totB / 1e9);
  uint64_t* out; long long* cyc; CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&cyc, 1 << 20));
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  for (double gb : gbs) {
    const size_t bytes = (size_t)(gb * (1ull << 30));
    if (bytes + (2ull << 30) > freeB) continue;
    uint64_t* buf; CK(hipMalloc(&buf, bytes));
    const uint64_t n = 1ull << 22;                       // 4 M lines in the chain, spread over the footprint
    const uint64_t strideWords = bytes / 8 / n;          // >= 2 (64 MB: 16 B apart -> use n smaller)
    uint64_t nn = n, sw = strideWords;
    if (sw < 8) { sw = 8; nn = bytes / 64; }
    hipLaunchKernelGGL(k_init, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, s1, buf, nn, sw);
    CK(hipStreamSynchronize(s1));
    for (int waves : {1, 256, 4096, 8192}) {
      for (int noise = 0; noise < 2; noise++) {
        if (noise) hipLaunchKernelGGL(k_noise, dim3(256 * 16), dim3(256), 0, s2, buf, nn, sw, 3000, out + 4096 * 8);
        const int steps = 2000;
        hipLaunchKernelGGL(k_chase2, dim3(waves), dim3(64), 0, s1, buf, nn, sw, steps, out, cyc);
        CK(hipStreamSynchronize(s1));
        CK(hipStreamSynchronize(s2));
        std::vector<long long> h(waves); CK(hipMemcpy(h.data(), cyc, waves * 8, hipMemcpyDeviceToHost));
        double sum = 0; for (auto v : h) sum += (double)v;
        printf("footprint %7.2f GB  waves %5d  noise %d : %.0f ns per dependent load (wall_clock64 at 100 MHz)\n", gb, waves, noise, sum / waves / steps * 10.0);
      }
    }
    CK(hipFree(buf));
  }
  return 0;
}
