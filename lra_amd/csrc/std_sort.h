// lra_amd/csrc/std_sort.h -- a faithful libstdc++ std::sort on 64-bit items with an arbitrary comparator, one lane per list.
// For the places where the reference sorts short lists with a comparator that ties distinct elements: the permutation std::sort
// leaves among ties is part of the reference's result.
#pragma once
#include <stdint.h>

namespace lra_std_sort {

// libstdc++ std::sort (bits/stl_algo.h: __introsort_loop -- right part first, depth limit 2*floor(log2 n), heap sort below it --
// then __final_insertion_sort, threshold 16), restated with an explicit stack in the recursion's own order
template <class Less>
__device__ void adjust_heap(uint64_t* v, long first, long hole, long len, uint64_t val, const Less& lt) {
  const long top = hole;
  long child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (lt(v[first + child], v[first + child - 1])) child--;
    v[first + hole] = v[first + child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    v[first + hole] = v[first + child - 1];
    hole = child - 1;
  }
  long parent = (hole - 1) / 2;
  while (hole > top && lt(v[first + parent], val)) { v[first + hole] = v[first + parent]; hole = parent; parent = (hole - 1) / 2; }
  v[first + hole] = val;
}

template <class Less>
__device__ void std_sort(uint64_t* v, long n, const Less& lt) {
  if (n < 2) return;
  long stF[64], stL[64]; int stD[64];
  int sp = 0;
  long first = 0, last = n; int depth = 2 * (63 - __clzll((unsigned long long)n));
  while (true) {
    while (last - first > 16) {
      if (depth == 0) {                                                  // __partial_sort(first, last, last)
        const long len = last - first;
        for (long parent = (len - 2) / 2;; parent--) { adjust_heap(v, first, parent, len, v[first + parent], lt); if (parent == 0) break; }
        long l2 = last;
        while (l2 - first > 1) { --l2; const uint64_t val = v[l2]; v[l2] = v[first]; adjust_heap(v, first, 0, l2 - first, val, lt); }
        break;
      }
      --depth;
      const long x = first + 1, y = first + (last - first) / 2, z = last - 1;
      auto sw = [&](long p, long q) { const uint64_t t = v[p]; v[p] = v[q]; v[q] = t; };
      if (lt(v[x], v[y])) { if (lt(v[y], v[z])) sw(first, y); else if (lt(v[x], v[z])) sw(first, z); else sw(first, x); }
      else if (lt(v[x], v[z])) sw(first, x);
      else if (lt(v[y], v[z])) sw(first, z);
      else sw(first, y);
      long f = first + 1, l = last;
      while (true) {
        while (lt(v[f], v[first])) ++f;
        --l;
        while (lt(v[first], v[l])) --l;
        if (!(f < l)) break;
        sw(f, l);
        ++f;
      }
      // recursion: (f, last) now, (first, f) after it
      stF[sp] = first; stL[sp] = f; stD[sp] = depth; sp++;
      first = f;
    }
    if (sp == 0) break;
    --sp; first = stF[sp]; last = stL[sp]; depth = stD[sp];
  }
  auto ins = [&](long b, long e) {                                       // __insertion_sort
    for (long i = b + 1; i < e; ++i) {
      const uint64_t val = v[i];
      if (lt(val, v[b])) { for (long k = i; k > b; --k) v[k] = v[k - 1]; v[b] = val; }
      else { long j = i; while (lt(val, v[j - 1])) { v[j] = v[j - 1]; --j; } v[j] = val; }
    }
  };
  if (n > 16) {
    ins(0, 16);
    for (long i = 16; i < n; ++i) { const uint64_t val = v[i]; long j = i; while (lt(val, v[j - 1])) { v[j] = v[j - 1]; --j; } v[j] = val; }
  } else ins(0, n);
}


}  // namespace lra_std_sort
