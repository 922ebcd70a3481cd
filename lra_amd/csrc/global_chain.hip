// lra_amd/csrc/global_chain.hip -- GlobalChain over a priority search tree, for a batch of independent fragment sets (gfx950 only).
//
// Replaces   int GlobalChain(vector<T_Fragment>& fragments, vector<int>& optFragmentChainIndices, vector<T_Endpoint>& endpoints)
//            (GlobalChain.h:85-189; Endpoint :10-60, FragmentSetToEndpoints :65-83; PrioritySearchTree.h: CreateTree :65-104, Activate :196-221,
//            FindIndexOfMaxPoint :106-143 / :223-231) -- the chaining component `north_star` names.  lra.cpp does not reach it (the include is commented out,
//            LocalRefineAlignment.h:13); the reference exercises it from TestGlobalChain.cpp only, and so does this library (tests/test_global_chain.py).
// Mapping: one lane per fragment set.  The algorithm is a serial sweep over the sorted endpoints against a tree that is updated on the way, and the reference's
// result depends on its exact order (std::sort's permutation of endpoints with equal (x, y); Activate descending by the ORIGINAL point's key after a swap; the tree
// keyed by y over points sorted by (x, y)), so it is restated literally: libstdc++'s introsort (std_sort.h) on endpoint indices, the tree in pre-order with the
// reference's node numbering (closed form: right child = left child + 2 * (median - start) - 1), explicit stacks for the two recursions.
// Algorithmic bytes: 16 B per fragment in, 8 B out + 4 B per chain element.
#include "common.h"
#include "scan.h"
#include "std_sort.h"

namespace {

struct GcArgs {
  uint64_t n;
  const uint64_t* off; const int32_t* xl; const int32_t* yl; const int32_t* xh; const int32_t* yh; const int32_t* scoreIn;
  int32_t* score; int32_t* prev; int32_t* chain; uint32_t* chainLen;
  // scratch: 2 entries per fragment (endpoints), 4 per fragment (tree nodes)
  uint64_t* order; int32_t* ex; int32_t* ey; int32_t* ef; int32_t* es; uint8_t* eside;
  uint32_t* tLeft; uint32_t* tRight; uint32_t* tMedian; int32_t* tMax; uint8_t* tLeaf;
  uint32_t* stk;
};

__global__ void __launch_bounds__(64) gc_kernel(GcArgs a) {
  const uint64_t pr = (uint64_t)blockIdx.x * 64 + threadIdx.x;
  if (pr >= a.n) return;
  const uint64_t f0 = a.off[pr];
  const int n = (int)(a.off[pr + 1] - f0);
  a.chainLen[pr] = 0;
  if (n == 0) return;
  const int32_t* xl = a.xl + f0; const int32_t* yl = a.yl + f0; const int32_t* xh = a.xh + f0; const int32_t* yh = a.yh + f0;
  int32_t* score = a.score + f0; int32_t* prev = a.prev + f0; int32_t* chain = a.chain + f0;
  uint64_t* order = a.order + 2 * f0;
  int32_t* ex = a.ex + 2 * f0; int32_t* ey = a.ey + 2 * f0; int32_t* ef = a.ef + 2 * f0; int32_t* es = a.es + 2 * f0; uint8_t* eside = a.eside + 2 * f0;
  uint32_t* tL = a.tLeft + 4 * f0; uint32_t* tR = a.tRight + 4 * f0; uint32_t* tMed = a.tMedian + 4 * f0; int32_t* tMax = a.tMax + 4 * f0; uint8_t* tLeaf = a.tLeaf + 4 * f0;
  uint32_t* stk = a.stk + 12 * f0;                                        // 3 words per entry, <= 2 log2(2n) + 2 entries; 12 n words is ample
  for (int i = 0; i < n; i++) { score[i] = a.scoreIn[f0 + i]; prev[i] = -1; }
  const int m = 2 * n;
  // FragmentSetToEndpoints :65-83, then std::sort with Endpoint::LessThan :37-47 (on indices: the same comparisons, the same permutation)
  for (int i = 0; i < m; i++) order[i] = (uint64_t)i;
  auto px = [&](uint64_t e) { return (e & 1) ? xh[e >> 1] : xl[e >> 1]; };
  auto py = [&](uint64_t e) { return (e & 1) ? yh[e >> 1] : yl[e >> 1]; };
  auto lt = [&](uint64_t p, uint64_t q) { const int a_ = px(p), b_ = px(q); return a_ != b_ ? a_ < b_ : py(p) < py(q); };
  lra_std_sort::std_sort(order, (long)m, lt);
  for (int i = 0; i < m; i++) { const uint64_t e = order[i]; ex[i] = px(e); ey[i] = py(e); ef[i] = (int32_t)(e >> 1); eside[i] = (uint8_t)(e & 1); es[i] = 0; }
  // CreateTree :65-104 over [0, m): pre-order numbering
  {
    int sp = 0;
    stk[0] = 0; stk[1] = (uint32_t)m; stk[2] = 0; sp = 1;
    while (sp) {
      sp--;
      const int start = (int)stk[3 * sp], end = (int)stk[3 * sp + 1]; const uint32_t cur = stk[3 * sp + 2];
      const int median = (start + end) / 2;
      tMax[cur] = -1; tL[cur] = 0; tR[cur] = 0;
      if (end - start == 1) { tLeaf[cur] = 1; tMed[cur] = (uint32_t)ey[start]; continue; }
      tLeaf[cur] = 0;
      tMed[cur] = (uint32_t)ey[median - 1];                              // the key the left subtree returns: its last point's
      const uint32_t left = cur + 1, right = left + 2 * (uint32_t)(median - start) - 1;
      tL[cur] = left; tR[cur] = right;
      stk[3 * sp] = (uint32_t)median; stk[3 * sp + 1] = (uint32_t)end; stk[3 * sp + 2] = right; sp++;
      stk[3 * sp] = (uint32_t)start; stk[3 * sp + 1] = (uint32_t)median; stk[3 * sp + 2] = left; sp++;
    }
  }
  // the sweep :117-160
  uint32_t maxEp = 0; bool found = false;
  for (int p = 0; p < m; p++) {
    if (eside[p] == 0) {
      int mi = 0, mv = -1; bool any = false;
      if (tMax[0] != -1) {                                                // FindIndexOfMaxPoint :223-231, :106-143 (left before right)
        const uint32_t maxKey = (uint32_t)ey[p];
        int sp = 0;
        stk[sp++] = 0;
        while (sp) {
          const uint32_t cur = stk[--sp];
          const int ms = tMax[cur];
          if (ms == -1) continue;
          if ((uint32_t)ey[ms] < maxKey) { if (es[ms] > mv) { mv = es[ms]; mi = ms; any = true; } continue; }
          if (tLeaf[cur]) continue;
          if (maxKey <= tMed[cur]) stk[sp++] = tL[cur];
          else { stk[sp++] = tR[cur]; stk[sp++] = tL[cur]; }
        }
      }
      if (any) { prev[ef[p]] = ef[mi]; score[ef[p]] = score[ef[mi]] + score[ef[p]]; }
      else prev[ef[p]] = -1;
    } else {
      es[p] = score[ef[p]];
      {                                                                   // Activate :196-221
        int pointIndex = p;
        const int pointScore = es[p];
        const uint32_t key = (uint32_t)ey[p];
        uint32_t cur = 0;
        while (pointIndex != -1 && tLeaf[cur] == 0) {
          if (tMax[cur] == -1 || es[tMax[cur]] <= pointScore) { const int tmp = tMax[cur]; tMax[cur] = pointIndex; pointIndex = tmp; }
          cur = key <= tMed[cur] ? tL[cur] : tR[cur];
        }
      }
      if (!found || score[ef[maxEp]] < score[ef[p]]) { maxEp = (uint32_t)p; found = true; }
    }
  }
  if (!found) return;
  int k = 0;
  for (int f = ef[maxEp]; f != -1 && k < n; f = prev[f]) chain[k++] = f;
  for (int i = 0, j = k - 1; i < j; i++, j--) { const int32_t t = chain[i]; chain[i] = chain[j]; chain[j] = t; }
  a.chainLen[pr] = (uint32_t)k;
}

inline size_t gsz(size_t n, size_t e) { return (n * e + 255) / 256 * 256; }

}  // namespace

extern "C" int lra_global_chain_batch(lra_ctx* ctx, uint64_t n_sets, const uint64_t* d_off, uint64_t n_fragments, const int32_t* d_xl, const int32_t* d_yl,
                                      const int32_t* d_xh, const int32_t* d_yh, const int32_t* d_score, lra_global_chain_result* out) {
  if (!ctx || !out) return LRA_ERR_INVALID;
  memset(out, 0, sizeof *out);
  out->n_sets = n_sets; out->n_fragments = n_fragments;
  if (n_sets == 0) return LRA_OK;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const uint64_t NF = n_fragments;
  char* w = (char*)lra_ensure(ctx, 176, gsz(NF + 1, 4) * 3 + gsz(n_sets + 1, 4) + gsz(2 * NF + 2, 8) + gsz(2 * NF + 2, 4) * 4 + gsz(2 * NF + 2, 1) + gsz(4 * NF + 4, 4) * 4 +
                                        gsz(4 * NF + 4, 1) + gsz(12 * NF + 16, 4) + 4096);
  if (!w) return LRA_ERR_NOMEM;
  auto take = [&](size_t n, size_t e) { char* r = w; w += gsz(n, e); return r; };
  GcArgs a;
  a.n = n_sets; a.off = d_off; a.xl = d_xl; a.yl = d_yl; a.xh = d_xh; a.yh = d_yh; a.scoreIn = d_score;
  a.score = (int32_t*)take(NF + 1, 4); a.prev = (int32_t*)take(NF + 1, 4); a.chain = (int32_t*)take(NF + 1, 4); a.chainLen = (uint32_t*)take(n_sets + 1, 4);
  a.order = (uint64_t*)take(2 * NF + 2, 8); a.ex = (int32_t*)take(2 * NF + 2, 4); a.ey = (int32_t*)take(2 * NF + 2, 4); a.ef = (int32_t*)take(2 * NF + 2, 4);
  a.es = (int32_t*)take(2 * NF + 2, 4); a.eside = (uint8_t*)take(2 * NF + 2, 1);
  a.tLeft = (uint32_t*)take(4 * NF + 4, 4); a.tRight = (uint32_t*)take(4 * NF + 4, 4); a.tMedian = (uint32_t*)take(4 * NF + 4, 4); a.tMax = (int32_t*)take(4 * NF + 4, 4);
  a.tLeaf = (uint8_t*)take(4 * NF + 4, 1); a.stk = (uint32_t*)take(12 * NF + 16, 4);
  lra_time_begin(ctx, "global_chain");
  hipLaunchKernelGGL(gc_kernel, dim3((unsigned)((n_sets + 63) / 64)), dim3(64), 0, st, a);
  lra_time_end(ctx);
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  LRA_HIP_CHECK(ctx, hipGetLastError());
  out->d_score = a.score; out->d_prev = a.prev; out->d_chain = a.chain; out->d_chain_len = a.chainLen;
  return LRA_OK;
}
