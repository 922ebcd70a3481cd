// lra_amd/csrc/kmer.h -- 2-bit base codes and the bit tricks of the wave-parallel minimizer sketches (seed.hip: reads, index.hip: the genome).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr uint64_t KM_FOR_MASK = ~(1ULL << 63);  // GenomeTuple::for_mask_s, lra.cpp:1008-1012
constexpr uint64_t KM_REV_MASK = (1ULL << 63);

__device__ __forceinline__ int km_code_n(unsigned char c) {  // SeqUtils.h:42 (seqMapN): non-ACGT -> 4
  if (c < 8) return c & 3;
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
  }
}
__device__ __forceinline__ uint64_t km_code2(unsigned char c) {  // SeqUtils.h:7 (seqMap): non-ACGT -> 0
  const int v = km_code_n(c);
  return v > 3 ? 0 : (uint64_t)v;
}
// bit i of x -> bit 2i
__device__ __forceinline__ uint64_t km_spread32(uint64_t x) {
  x = (x | (x << 16)) & 0x0000FFFF0000FFFFULL;
  x = (x | (x << 8)) & 0x00FF00FF00FF00FFULL;
  x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0FULL;
  x = (x | (x << 2)) & 0x3333333333333333ULL;
  x = (x | (x << 1)) & 0x5555555555555555ULL;
  return x;
}
// Canonical GenomeTuple key of the k-mer whose low / high code bits (first base in bit 0) are x0 / x1 (MinCount.h:51-62): the smaller of the
// forward and reverse-complement 2-bit words by masked value, strand in bit 63.
__device__ __forceinline__ uint64_t km_canonical(uint64_t x0, uint64_t x1, int k) {
  const uint64_t mask2k = (k >= 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
  const uint64_t LE = km_spread32(x0) | (km_spread32(x1) << 1);
  const uint64_t rc = (~LE) & mask2k;
  uint64_t v = __brevll(LE);
  v = ((v >> 1) & 0x5555555555555555ULL) | ((v & 0x5555555555555555ULL) << 1);
  const uint64_t fwd = (k >= 32) ? v : (v >> (64 - 2 * k));
  return ((fwd & KM_FOR_MASK) < (rc & KM_FOR_MASK)) ? (fwd & KM_FOR_MASK) : (rc | KM_REV_MASK);
}

}  // namespace
