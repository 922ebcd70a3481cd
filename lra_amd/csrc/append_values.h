// lra_amd/csrc/append_values.h -- AppendValues (TupleOps.h:159-195) over the pairs of a batch of CompareLists tasks: what Refine_splitchain (ChainRefine.h:533-539,
// refine_splitchain.hip) and REFINEclusters (ClusterRefine.h:207-215, refine_clusters.hip) apply to every (read window, genome window) task's pairs.  gfx950 only.
#pragma once
#include "common.h"

namespace {

struct AvArgs {
  uint64_t n_tasks;
  const uint64_t* pairOff; const uint32_t* pqi; const uint32_t* pti; const uint32_t* qTup; const uint32_t* gTup;
  const uint32_t* qAdd; const uint32_t* tAdd; const int64_t* mx; const int64_t* mn; const uint32_t* tbox;   // tbox: qs, qe, ts, te per task
  uint32_t* cnt; const uint64_t* outOff; uint32_t* oq; uint32_t* ot;
};

// AppendValues TupleOps.h:159-195.  A WAVE per 64 consecutive tasks: their pairs are one contiguous range of the pair arrays, which the lanes read 64 at a time
// (a lane per task walked its own ~19 pairs: every step of the wave touched 64 lines of each array, 19 GB fetched per pass for 0.75 GB of pairs); a pair finds its task
// among the wave's 65 offsets in LDS, the tasks' parameters come from LDS too.  Pass 1 counts what a task keeps, pass 2 writes it in the pairs' order: a kept pair's
// place = its task's offset + what the task kept in earlier rounds + the kept pairs of its task on lower lanes of this round (a task's pairs are a contiguous lane range).
template <bool EMIT>
__global__ void __launch_bounds__(64) av_filter(AvArgs a) {
  __shared__ uint64_t s_off[65];
  __shared__ uint32_t s_qa[64], s_ta[64], s_box[4 * 64], s_cnt[64];
  __shared__ int64_t s_mx[64], s_mn[64];
  const int lane = threadIdx.x;
  const uint64_t t0 = (uint64_t)blockIdx.x * 64;
  if (t0 >= a.n_tasks) return;
  const int nt = (int)min((uint64_t)64, a.n_tasks - t0);
  if (lane < nt) {
    const uint64_t t = t0 + lane;
    s_off[lane] = a.pairOff[t]; s_qa[lane] = a.qAdd[t]; s_ta[lane] = a.tAdd[t]; s_mx[lane] = a.mx[t]; s_mn[lane] = a.mn[t];
    s_box[4 * lane] = a.tbox[4 * t]; s_box[4 * lane + 1] = a.tbox[4 * t + 1]; s_box[4 * lane + 2] = a.tbox[4 * t + 2]; s_box[4 * lane + 3] = a.tbox[4 * t + 3];
  }
  s_cnt[lane] = 0;
  if (lane == 0) s_off[nt] = a.pairOff[t0 + nt];
  __syncthreads();
  const uint64_t P0 = s_off[0], P1 = s_off[nt];
  const uint32_t* Q = a.qTup;
  const unsigned long long below = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));
  for (uint64_t pb = P0; pb < P1; pb += 64) {
    const uint64_t p = pb + lane;
    const bool in = p < P1;
    int j = 0;
    bool keep = false; uint32_t fp = 0, sp = 0;
    if (in) {
      int lo = 0, hi = nt;                                               // the task of pair p: the last j with s_off[j] <= p (empty tasks are skipped by that rule)
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_off[mid] <= p) lo = mid; else hi = mid; }
      j = lo;
      fp = (Q[a.pqi[p]] >> 20) + s_qa[j]; sp = (a.gTup[a.pti[p]] >> 20) + s_ta[j];
      const int64_t diag = (int64_t)sp - (int64_t)fp;
      keep = diag >= s_mn[j] && diag <= s_mx[j] && fp >= s_box[4 * j] && fp < s_box[4 * j + 1] && sp >= s_box[4 * j + 2] && sp < s_box[4 * j + 3];
    }
    const unsigned long long km = __ballot(keep);
    // the lanes of this round that belong to task j: from max(pb, s_off[j]) on
    const uint64_t jb = in ? s_off[j] : 0;
    const int firstLane = (in && jb > pb) ? (int)(jb - pb) : 0;
    const unsigned long long fromFirst = firstLane == 0 ? ~0ULL : (~0ULL << firstLane);
    const uint32_t before = in ? s_cnt[j] : 0;                            // what the task kept in earlier rounds
    const uint32_t rank = (uint32_t)__popcll(km & below & fromFirst);
    if (EMIT && keep) { const uint64_t o = a.outOff[t0 + j] + before + rank; a.oq[o] = fp; a.ot[o] = sp; }
    __syncthreads();                                                      // (every lane has read its task's count of the earlier rounds)
    // the last lane of a task's range in this round adds the round's kept pairs of that task
    const bool lastOfTask = in && (p + 1 == P1 || p + 1 >= s_off[j + 1] || lane == 63);
    if (lastOfTask) s_cnt[j] = before + rank + (keep ? 1u : 0u);
    __syncthreads();
  }
  if (!EMIT && lane < nt) a.cnt[t0 + lane] = s_cnt[lane];
}


}  // namespace
