// lra_amd/csrc/sdp.hip -- SURVEY §8a row a8: the first sparse dynamic program of the low-accuracy path (SDP#A),
// SparseDP(vector<Cluster>&, vector<UltimateChain>&, ...) (SparseDP.h:2139-2279, called at Map_lowacc.h:188), for a
// whole batch of reads.  gfx950 only.
//
// What the reference does per read: every anchor becomes a start/end point pair per orientation family
// (insertPointsPair :79), the points are std::sort'ed by row and by column, four divide-and-conquer decompositions
// (DivideSubProbBy{Row,Col}{1,2}) build a tree of sub-problems holding the distinct diagonals of the end points of
// one half (Di) and of the start points of the other half (Ei); ProcessPoint (:1015) then walks the points in row
// order: a start point queries every sub-problem on its root-to-leaf paths (Maximization / FindValueInBlock, a
// candidate-list structure over the PWL gap cost, SubRountine.h:270-345), an end point deposits the anchor's value
// in them (PassValueToD*).  The result depends on the order of all these steps (values deposited after a
// candidate was consumed stay invisible, `last` moves backwards, slots are overwritten), so it is reproduced
// literally; what is re-designed is how it is laid out and scheduled:
//
//  * points, sorts: one thread per cluster writes the points; the two std::sorts are the library's libstdc++-exact
//    workgroup sort (seed.hip) on packed 63-bit keys (q,t,ind / t,q,ind) -- the permutation of tied points is part
//    of the result because it fixes the processing order;
//  * decompositions: the sub-problem numbering is internal to the reference (prev_sub is only an index), so the trees
//    are built level by level instead of depth first: per family the points are kept sorted by diagonal and stably
//    partitioned by half at every level (prefix sums), which yields every node's sorted distinct Di/Ei at once;
//    Db/Eb have closed forms (counts of smaller diagonals), and every (point, level) gets its sub-problem and its
//    index in Di/Ei recorded -- the Lower_Bound searches of ProcessPoint/PassValueToD* disappear;
//  * ProcessPoint: one wave per read; lane (family pair, level) < 2 * LV owns the sub-problems of that level, so the sub-problems a
//    point touches advance concurrently and none is ever touched by two lanes; short candidate insertions run per lane, long ones
//    wave-cooperatively (prefetched candidates, iterations that change nothing skipped with a ballot, six-level probes in the
//    boundary search); the ordered `<` update of Value[ii] becomes a (max value, first in order) reduction; stacks / Block lists that
//    fill up double out of a per-read pool;
//  * TraceBack / DecidePrimaryChains: one lane per read after the exact sort of the values.
//
// Roofline: integer/float bookkeeping with dependent loads, HBM-nominal; algorithmic bytes = 44 B per sub-problem entry + 269 B per
// point (visit row + coordinates).  A launch lasts as long as its largest reads: one chunk per batch, largest first.
#include "common.h"
#include <type_traits>
#include "scan.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>
#include <rocprim/rocprim.hpp>

namespace {

constexpr int LV = 18;                    // levels per decomposition (distinct rows / columns per read <= 131072); 2 * LV lanes own them
constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr int MAXALN = 16;

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// Inclusive prefix sum over the wave: four DPP row shifts (a row = 16 lanes; nothing is shifted in across a row's start), then the two row broadcasts (rows 1 and 3 take
// lane 15 of the row before them, rows 2 and 3 lane 31) -- six VALU operations where six __shfl_up are six dependent trips through the LDS crossbar.
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
  (void)lane;
#define LRA_DPP_ADD(ctrl_, rmask_) v += __builtin_amdgcn_update_dpp(0, v, (ctrl_), (rmask_), 0xf, false)
  LRA_DPP_ADD(0x111, 0xf);   // row_shr:1
  LRA_DPP_ADD(0x112, 0xf);   // row_shr:2
  LRA_DPP_ADD(0x114, 0xf);   // row_shr:4
  LRA_DPP_ADD(0x118, 0xf);   // row_shr:8
  LRA_DPP_ADD(0x142, 0xa);   // row_bcast:15
  LRA_DPP_ADD(0x143, 0xc);   // row_bcast:31
#undef LRA_DPP_ADD
  return v;
}

struct PwlTab { long long stops[25]; float slope[25], inter[25]; int c1, c2; };

struct Ent { long long val; int b; float v; };   // one Di / Ei slot: diagonal, Db / Eb, Dv / Ev   (16 bytes)
struct Node {            // one full sub-problem (SubProblem.h:15-37), 48 bytes
  uint32_t dBase;        // entry index of Di[0] within the read's entries; Ei[0] at dBase + nD
  uint32_t nD, nE;
  int32_t last;
  uint32_t sTop, nBlk;   // sizes of S_1 and Block
  uint32_t stkOff, blkOff;   // where they live, in pairs from the read's pair area (stacks, Blocks, then the growth pool)
  uint32_t stkCap, blkCap;   // their current capacities: 2 nD + 4 and 2 (nD + nE) + 8 pairs to begin with, doubled from the pool on demand
  long long eLast;           // Ei[nE - 1]: the boundary diagonal of a candidate that owns the whole tail (nearly every push is (i, nE)), so that a push needs no load
};

// Everything ProcessPoint touches for one read lies in one contiguous block (sections 256-byte aligned): a wave's working set is a
// couple of megabytes in one place instead of six arrays gigabytes apart (TLB reach).
struct ReadArena { uint64_t base; uint32_t entOff, apOff, stkOff, visOff, blkPair, poolPair, poolPairs, edOff; };   // base: device address; byte offsets; nodes at 0
// edOff: one 64-bit word per entry -- for a D entry d the diagonal Ei[Db[d]] (static), which Maximization compares every candidate at (SubRountine.h:292): stored
// beside the entry, the candidate scan is ONE round of independent loads instead of two dependent ones
// The pair area at stkOff holds the candidate stacks, then (from pair index blkPair) the Block lists, then (from poolPair) a pool of
// poolPairs pairs.  Re-inserted candidates (`last` moving backwards) let a stack / Block outgrow any fixed multiple of its sub-problem, so
// they start at 2 nD + 4 / 2 (nD + nE) + 8 pairs and double out of the pool when full; a read that exhausts its pool is re-run with 8x, 64x.

__device__ __host__ inline uint32_t al256(uint64_t x) { return (uint32_t)((x + 255) & ~(uint64_t)255); }
// A read's block is found through an address kept in a table (ReadArena::base).  A pointer made from an integer is a FLAT pointer to the compiler: every access through
// it is a flat_load / flat_store, which counts on the LDS counter as well as on the vector-memory one -- so every LDS read (the gap-cost table inside w(), the slot state)
// waits for all the stores in flight (a stack / Block push is followed by exactly that).  Saying that the address is in global memory gives global_load / global_store.
__device__ __forceinline__ char* arena_ptr(uint64_t addr) { return (char*)(__attribute__((address_space(1))) char*)addr; }

__global__ void k_arena_sizes(int n, int r0, const uint64_t* __restrict__ ptOff, const uint32_t* __restrict__ cntE, const uint32_t* __restrict__ cntN,
                              const uint32_t* __restrict__ cntD, ReadArena* ra, uint64_t* bytes, const uint32_t* __restrict__ order, int shift) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  const int rr = (int)order[b];
  const uint64_t E = cntE[rr], N = cntN[rr], D = cntD[rr], P = ptOff[r0 + rr + 1] - ptOff[r0 + rr];
  ReadArena a;
  a.base = 0;
  uint64_t o = al256(N * sizeof(Node));
  a.entOff = (uint32_t)o; o = al256(o + E * sizeof(Ent));
  a.apOff = (uint32_t)o; o = al256(o + E * 4);
  a.edOff = (uint32_t)o; o = al256(o + E * 8);
  const uint64_t stkPairs = 2 * D + 4 * N + 2, blkPairs = 2 * E + 8 * N + 2, poolPairs = (2 * E + 4096) << shift;
  a.stkOff = (uint32_t)o; a.blkPair = (uint32_t)stkPairs; a.poolPair = (uint32_t)(stkPairs + blkPairs); a.poolPairs = (uint32_t)poolPairs;
  o = al256(o + (stkPairs + blkPairs + poolPairs) * 8);
  a.visOff = (uint32_t)o; o = al256(o + P * 2 * LV * sizeof(uint2));
  ra[rr] = a;
  bytes[b] = o;
}
// Sizes WITHOUT the count pass: entries, sub-problems and D entries per point are narrow distributions (measured over the headline batch: 3.2 .. 9.8 entries and 0.3 .. 1.9
// sub-problems per point, D entries 43 .. 56 % of the entries), so a read's blocks are laid out for fE / fN per point and the emit pass checks every level against them; the rare
// read that outgrows its blocks is counted exactly and built again with the reads whose stacks outgrew theirs (attempt 1 of sdp_run).
__global__ void k_arena_estimate(int n, int r0, const uint64_t* __restrict__ ptOff, float fE, float fN, uint32_t* cntE, uint32_t* cntN, uint32_t* cntD) {
  int rr = blockIdx.x * blockDim.x + threadIdx.x;
  if (rr >= n) return;
  const uint64_t P = ptOff[r0 + rr + 1] - ptOff[r0 + rr];
  const uint64_t E = (uint64_t)(fE * (float)P) + 64, N = (uint64_t)(fN * (float)P) + 64;
  cntE[rr] = (uint32_t)std::min<uint64_t>(E, 0xFFFFFFFFull); cntN[rr] = (uint32_t)std::min<uint64_t>(N, 0xFFFFFFFFull); cntD[rr] = (uint32_t)std::min<uint64_t>(E * 6 / 10 + 64, 0xFFFFFFFFull);
}
__global__ void k_arena_bases(int n, const uint64_t* __restrict__ byteOff, ReadArena* ra, const uint32_t* __restrict__ order, char* arena) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < n) ra[order[b]].base = (uint64_t)(uintptr_t)(arena + byteOff[b]);
}
__global__ void k_visit_clear(const ReadArena* __restrict__ ra, const uint64_t* __restrict__ byteOff, const uint32_t* __restrict__ order) {
  const int b = blockIdx.x;
  const ReadArena A = ra[order[b]];
  uint4* p = (uint4*)(arena_ptr(A.base) + A.visOff);
  const uint64_t n = (byteOff[b + 1] - byteOff[b] - A.visOff) / 16;
  for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) p[i] = make_uint4(NONE, NONE, NONE, NONE);
}

// The distinct rows and columns (GetRowInfo / GetColInfo) of the reads a launch gives a workgroup each, ahead of their build: the maximum picks the sdp_process_wg
// variant, so that the workgroup ProcessPoint launch can follow the large reads' build on its side stream without a word from the host in between.
__global__ void k_big_lines(int r0, const uint32_t* __restrict__ order, const uint64_t* __restrict__ ptOff, const uint32_t* __restrict__ hq, const uint32_t* __restrict__ ht,
                            const uint32_t* __restrict__ h2, uint32_t* maxLines) {
  const int r = r0 + (int)order[blockIdx.x];
  const uint64_t p0 = ptOff[r];
  const int P = (int)(ptOff[r + 1] - p0);
  const uint32_t* q = hq + p0; const uint32_t* t = ht + p0; const uint32_t* c = h2 + p0;
  uint32_t R = 0, C = 0;
  for (int i = threadIdx.x; i < P; i += blockDim.x) { R += (i == 0 || q[i] != q[i - 1]); C += (i == 0 || t[c[i]] != t[c[i - 1]]); }
  for (int o = 32; o > 0; o >>= 1) { R += __shfl_xor(R, o); C += __shfl_xor(C, o); }
  __shared__ uint32_t sR, sC;
  if (threadIdx.x == 0) { sR = 0; sC = 0; }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { atomicAdd(&sR, R); atomicAdd(&sC, C); }
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(maxLines, max(sR, sC));
}

// ---- counting / point generation ------------------------------------------------------------------------------------
__global__ void k_cluster_counts(uint64_t nc, const uint32_t* __restrict__ c_count, uint32_t* fragCnt, uint32_t* ptCnt, int single) {
  uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  if (!c_count) { fragCnt[c] = 1; ptCnt[c] = 4; return; }                  // box mode (SparseDP.h:1959-2018): s1 e1 s2 e2 for every box
  uint32_t n = c_count[c];
  fragCnt[c] = n;
  ptCnt[c] = 2 * n + (single ? 0 : 2 * (n == 0 ? 0 : n == 1 ? 1 : 2));     // SparseDP.h:2159-2166: first and last anchor get the other family's pair too
}

__global__ void k_read_offsets(int n_reads, const uint64_t* __restrict__ cluster_off, const uint64_t* __restrict__ clusFragOff,
                               const uint64_t* __restrict__ clusPtOff, uint64_t* fragOff, uint64_t* ptOff, uint32_t* clusRead,
                               uint32_t* status, uint32_t* nChains) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n_reads) return;
  fragOff[r] = clusFragOff[cluster_off[r]];
  ptOff[r] = clusPtOff[cluster_off[r]];
  if (r < n_reads) {
    for (uint64_t c = cluster_off[r]; c < cluster_off[r + 1]; c++) clusRead[c] = r;
    status[r] = 0; nChains[r] = 0;
  }
}

struct PtArgs {
  uint64_t nc;
  const uint64_t* cluster_off; const uint64_t* c_start; const uint32_t* c_count; const int32_t* c_strand;
  const uint32_t* q; const uint32_t* t; const int32_t* len;
  const uint32_t* clusRead; const uint64_t* clusFragOff; const uint64_t* clusPtOff; const uint64_t* fragOff; const uint64_t* ptOff;
  const float* rate_in; float rate; int single;
  uint32_t* fq; uint32_t* ft; int32_t* flen; uint32_t* fcl; uint32_t* fai; float* fval; uint32_t* fprevNode; uint32_t* fprevInd; uint8_t* fflags;
  uint8_t* used; uint8_t* fstrand;
  const uint32_t* qe; const uint32_t* te; uint32_t* fqe; uint32_t* fte;   // box mode only
  uint64_t* key1; uint32_t* pay1; uint32_t* iq; uint32_t* it; uint8_t* ifl; uint32_t* ifr; uint32_t* ptRead;
};

// anchors -> compact fragment arrays + points in insertion order (SparseDP.h:2152-2169).  Box mode: one thread per cluster (= fragment).  Anchor mode: one WAVE per
// cluster, a lane per anchor -- a merged cluster of a satellite read holds 20 k anchors, and one thread walking them kept the launch at 5-13 ms; an anchor's
// points sit at 2 i (+ 2 behind the first anchor's second pair).
__global__ void k_points(PtArgs a) {
  const bool boxMode = a.qe != nullptr;
  const uint64_t gtid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t c = boxMode ? gtid : (gtid >> 6);
  const uint32_t lane = threadIdx.x & 63;
  if (c >= a.nc) return;
  const uint32_t r = a.clusRead[c];
  const int strand = a.c_strand[c];
  uint64_t g = a.clusFragOff[c], p = a.clusPtOff[c];
  const uint64_t f0 = a.fragOff[r], p0 = a.ptOff[r];
  const uint32_t cl = (uint32_t)(c - a.cluster_off[r]);
  const float rate = a.rate_in ? a.rate_in[r] : a.rate;
  if (a.qe) {                                                          // box mode: the split cluster c is the fragment (SparseDP.h:1959-2018)
    const uint32_t qs = a.q[c], ts = a.t[c], qe = a.qe[c], te = a.te[c];
    const int val = a.len[c];
    a.fq[g] = qs; a.ft[g] = ts; a.fqe[g] = qe; a.fte[g] = te; a.flen[g] = val; a.fcl[g] = cl; a.fai[g] = 0;
    a.fval[g] = val * rate;                                            // Value[ii].val = FragInput[ii].Val*rate (:2084)
    a.fprevNode[g] = NONE; a.fprevInd[g] = NONE; a.fflags[g] = 3; a.used[g] = 0; a.fstrand[g] = (uint8_t)(strand != 0);
    const uint32_t lf = (uint32_t)(g - f0);
    for (int k = 0; k < 4; k++, p++) {                                 // s1 (qs+1,ts+1)  e1 (qe-1,te-1)  s2 (qs+1,te-1)  e2 (qe-1,ts+1)
      const uint8_t ind = (k & 1) ? 0 : 1, inv = k < 2 ? 1 : 0;
      const uint32_t pq = ind ? qs + 1 : qe - 1;
      const uint32_t pt = (k == 0 || k == 3) ? ts + 1 : te - 1;
      a.key1[p] = ((uint64_t)pq << 33) | ((uint64_t)pt << 1) | ind;
      a.pay1[p] = (uint32_t)(p - p0);
      a.iq[p] = pq; a.it[p] = pt; a.ifl[p] = (uint8_t)(ind | (inv << 1)); a.ifr[p] = lf; a.ptRead[p] = r;
    }
    return;
  }
  const uint32_t n = a.c_count[c];
  const uint64_t src = a.c_start[c];
  const uint64_t g0 = g, pc0 = p;
  for (uint32_t i = lane; i < n; i += 64) {
    g = g0 + i; p = pc0 + 2 * (uint64_t)i + ((!a.single && i >= 1) ? 2 : 0);
    const uint32_t q = a.q[src + i], t = a.t[src + i];
    const int len = a.len[src + i];
    const uint32_t lf = (uint32_t)(g - f0);
    a.fq[g] = q; a.ft[g] = t; a.flen[g] = len; a.fcl[g] = cl; a.fai[g] = i;
    a.fval[g] = len * rate;                                            // Value[ii].val = matchesLengths * rate (:2206)
    a.fprevNode[g] = NONE; a.fprevInd[g] = NONE; a.fflags[g] = 3; a.used[g] = 0; a.fstrand[g] = (uint8_t)(strand != 0);
    const bool edge = !a.single && (i == 0 || i == n - 1);                 // the single-cluster SDP (SparseDP.h:2296-2305) inserts one pair only
    for (int rep = 0; rep < (edge ? 2 : 1); rep++) {
      const int pair = (strand == 0) ? rep : 1 - rep;                  // forward cluster: s1/e1 first; reverse: s2/e2 first
      uint32_t sq, st, eq, et;
      if (pair == 0) { sq = q; st = t; eq = q + len; et = t + len; }   // insertPointsPair :79-137
      else { sq = q; st = t + len; eq = q + len; et = t; }
      const uint8_t inv = pair == 0 ? 1 : 0;
      for (int e = 0; e < 2; e++, p++) {
        const uint32_t pq = e ? eq : sq, pt = e ? et : st;
        const uint8_t ind = e ? 0 : 1;
        a.key1[p] = ((uint64_t)pq << 33) | ((uint64_t)pt << 1) | ind; // SortByRowOp: q, t, ind (Sorting.h:226)
        a.pay1[p] = (uint32_t)(p - p0);
        a.iq[p] = pq; a.it[p] = pt; a.ifl[p] = (uint8_t)(ind | (inv << 1)); a.ifr[p] = lf; a.ptRead[p] = r;
      }
    }
  }
}

// after the row sort: gather the point attributes into H1 order, build the column-sort and diagonal-sort keys
__global__ void k_gather(uint64_t np, const uint32_t* __restrict__ ptRead, const uint64_t* __restrict__ ptOff, const uint32_t* __restrict__ pay1,
                         const uint32_t* __restrict__ iq, const uint32_t* __restrict__ it, const uint8_t* __restrict__ ifl,
                         const uint32_t* __restrict__ ifr, uint32_t* hq, uint32_t* ht, uint8_t* hfl, uint32_t* hfr, uint64_t* key2,
                         uint32_t* pay2, uint64_t* key3, uint32_t* pay3) {
  uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= np) return;
  const uint64_t p0 = ptOff[ptRead[p]];
  const uint64_t s = p0 + pay1[p];
  const uint32_t q = iq[s], t = it[s];
  const uint8_t fl = ifl[s];
  hq[p] = q; ht[p] = t; hfl[p] = fl; hfr[p] = ifr[s];
  key2[p] = ((uint64_t)t << 31) | ((uint64_t)q << 1) | (fl & 1);       // SortByColOp: t, q, ind (Sorting.h:241)
  pay2[p] = (uint32_t)(p - p0);
  const int inv = (fl >> 1) & 1, ind = fl & 1;
  const uint64_t cls = (uint64_t)((inv ? 0 : 2) + (ind ? 0 : 1));      // 0: s1, 1: e1, 2: s2, 3: e2
  const uint64_t dg = inv ? (uint64_t)((int64_t)t - (int64_t)q + (1LL << 32)) : (uint64_t)t + q;
  key3[p] = (cls << 40) | dg;
  pay3[p] = (uint32_t)(p - p0);
}

// before a read is re-run with larger stacks: Value[] back to its initial state (SparseDP.h:2206), status cleared
__global__ void k_reset_frags(int r0, const uint32_t* __restrict__ order, const uint64_t* __restrict__ fragOff, const int32_t* __restrict__ flen,
                              const float* __restrict__ rate_in, float rate, float* fval, uint32_t* fprevNode, uint32_t* fprevInd, uint8_t* fflags,
                              uint32_t* status) {
  const int r = r0 + (int)order[blockIdx.x];
  const float rt = rate_in ? rate_in[r] : rate;
  for (uint64_t g = fragOff[r] + threadIdx.x; g < fragOff[r + 1]; g += blockDim.x) { fval[g] = flen[g] * rt; fprevNode[g] = NONE; fprevInd[g] = NONE; fflags[g] = 3; }
  if (threadIdx.x == 0) status[r] = 0;
}

// ---- decompositions ---------------------------------------------------------------------------------------------------
struct BuildArgs {
  int r0, n;                                 // reads [r0, r0 + n)
  const uint32_t* order;                     // block b works on read r0 + order[b] (largest first: the longest waves start first)
  const uint64_t* ptOff;
  const uint32_t* hq; const uint32_t* ht; const uint8_t* hfl; const uint32_t* h2; const uint64_t* key3; const uint32_t* pay3;
  uint32_t* scratch;                         // 34 words per point + 64 per read
  uint32_t* cntEntries; uint32_t* cntNodes; uint32_t* cntD; uint32_t* cntV; uint32_t* cntRC;   // [n] (count pass out; cntRC: max(distinct rows, distinct columns))
  const ReadArena* ra;                       // emit pass: per-read blocks
  uint32_t* status;
  unsigned long long* stat;                  // LRA_SDP_BUILD_STAT: cycles per pass (set-up, A, C, D, E, F, G, family set-up) summed over the launch's reads; null = off
};

// NW = waves per read: 1 (a wave per read) or 16 (a 1024-thread workgroup per LARGE read: every pass over the read's points is spread over the block, the
// wave scans become block scans through LDS; the same arithmetic, the same tables).
template <int NW>
__device__ __forceinline__ int blk_incl_scan(int v, int lane, int wave, int* s_w, int& total) {
  const int inc = wave_incl_scan(v, lane);
  if (NW == 1) { total = __builtin_amdgcn_readlane(inc, 63); return inc; }
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  int pre = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NW; w++) { const int x = s_w[w]; tot += x; if (w < wave) pre += x; }
  __syncthreads();
  total = tot;
  return inc + pre;
}
// LDSV (one wave per read, at most 512 points): the per-element arrays -- 28 bytes per point with 16-bit indices and 32-bit diagonals -- live in the wave's LDS.
// From the scratch arena every level streams them through HBM again (2.6 KB per point and build: 250 GB per step, a third of the step's traffic).
// MODE 2: the same narrow arrays in the arena (reads of 513 .. 16383 points: half the bytes per level, no LDS to run out of).  MODE 0: 32-bit indices, 64-bit diagonals.
// BSTAT (LRA_SDP_BUILD_STAT): cycles per pass -- an instantiation of its own: the kernel spills as it is, and the counter's two scalars more than double what it spills
template <bool EMIT, int NW, int MODE = 0, int OCC = 8, bool BSTAT = false>
__global__ void __launch_bounds__(64 * NW, NW == 1 ? OCC : 1) sdp_build(BuildArgs a) {
  constexpr int NT = 64 * NW;
  constexpr bool LDSV = MODE == 1, NARROW = MODE != 0;
  using IT = typename std::conditional<NARROW, uint16_t, uint32_t>::type; // element -> node / position / line / prefix count
  using DT = typename std::conditional<NARROW, uint32_t, long long>::type; // a diagonal (compared for equality only: 32 bits of it identify it inside one read)
  constexpr IT INONE = (IT)~(IT)0;
  extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
  __shared__ int s_w[4][NW == 1 ? 1 : NW];
  __shared__ unsigned long long s_bt[BSTAT ? 8 : 1];
  unsigned long long btPrev = 0;
  constexpr bool bstat = BSTAT;
  if (bstat) { if (threadIdx.x < 8) s_bt[threadIdx.x] = 0; btPrev = __builtin_amdgcn_s_memtime(); }
#define BTICK(k_) do { if (bstat) { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) s_bt[k_] += t__ - btPrev; btPrev = t__; } } while (0)
  const int rr = (int)a.order[blockIdx.x], r = a.r0 + rr, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto SYNC = [&]() { if (NW == 1) wave_sync(); else __syncthreads(); };
  const uint64_t p0 = a.ptOff[r], pc0 = a.ptOff[a.r0];
  const int P = (int)(a.ptOff[r + 1] - p0);
  if (P == 0) { if (tid == 0) { if (!EMIT) { a.cntEntries[rr] = 0; a.cntNodes[rr] = 0; a.cntD[rr] = 0; } a.cntV[rr] = 0; a.cntRC[rr] = 0; } return; }
  const uint32_t* hq = a.hq + p0; const uint32_t* ht = a.ht + p0; const uint32_t* h2 = a.h2 + p0;
  const uint64_t* key3 = a.key3 + p0; const uint32_t* pay3 = a.pay3 + p0;
  uint32_t* S = a.scratch + 34 * (p0 - pc0) + 64 * (uint64_t)rr;
  const int NCAP = P + 2;
  // the arena's layout (34 words per point); LDSV uses its node tables only
  uint32_t* tbl = S + 8 * P + 2;            // [2][6][NCAP]
  uint32_t* tmp = tbl + 12 * NCAP;          // [8][NCAP]
  IT* eb_ = LDSV ? (IT*)dyn_lds : (IT*)S;
  IT* rowOf = eb_; IT* colOf = rowOf + P;
  IT* lp = colOf + P;                       // [2][P]
  IT* ln = lp + 2 * P;                      // [2][P]: [0] node index of the element, [1] temporary 2k+side
  IT* pf = ln + 2 * P;                      // [P+1]
  IT* ph = pf + P + 1;                      // [P+1]
  // MODE 0: ll and ld behind the tables (word offset 30 P + 42 from S: even, so ld is 8-byte aligned); narrow: ll behind ph, ld behind the tables / behind ll in LDS
  IT* ll = NARROW ? ph + P + 1 : (IT*)(tmp + 8 * NCAP);   // [2][P]  line (row / column index) of the element: travels with it, no gathers per level
  DT* ld = LDSV ? (DT*)(dyn_lds + ((((size_t)(10 * P + 2) * sizeof(IT)) + 7) & ~(size_t)7)) : NARROW ? (DT*)(tmp + 8 * NCAP) : (DT*)(ll + 2 * P);   // [2][P]  its diagonal
  auto LN = [&](int idx) -> uint32_t { const IT v = ln[idx]; return v == INONE ? NONE : (uint32_t)v; };
#define TB(c, f, k) tbl[((c) * 6 + (f)) * NCAP + (k)]
#define TM(f, k) tmp[(f) * NCAP + (k)]
  enum { F_LS, F_LE, F_SB, F_SE, F_EB, F_EE };
  enum { T_C1S, T_C1E, T_ND, T_NE, T_CH0, T_CH1, T_BASE, T_GID };
  // rows (GetRowInfo) and columns (GetColInfo): index of the distinct q / t of every point
  int R = 0, C = 0;
  for (int i0 = 0; i0 < P; i0 += NT) {
    const int i = i0 + tid;
    const int head = (i < P) && (i == 0 || hq[i] != hq[i - 1]);
    int tot; const int inc = blk_incl_scan<NW>(head, lane, wave, s_w[0], tot);
    if (i < P) rowOf[i] = R + inc - 1;
    R += tot;
  }
  for (int i0 = 0; i0 < P; i0 += NT) {
    const int i = i0 + tid;
    const int head = (i < P) && (i == 0 || ht[h2[i]] != ht[h2[i - 1]]);
    int tot; const int inc = blk_incl_scan<NW>(head, lane, wave, s_w[0], tot);
    if (i < P) colOf[h2[i]] = C + inc - 1;
    C += tot;
  }
  // class boundaries in the diagonal-sorted list
  int cOff[5];
  {
    int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int i = tid; i < P; i += NT) { const int cl = (int)(key3[i] >> 40); c0 += cl == 0; c1 += cl == 1; c2 += cl == 2; c3 += cl == 3; }
    for (int o = 32; o > 0; o >>= 1) { c0 += __shfl_xor(c0, o); c1 += __shfl_xor(c1, o); c2 += __shfl_xor(c2, o); c3 += __shfl_xor(c3, o); }
    if (NW > 1) {
      if (lane == 0) { s_w[0][wave] = c0; s_w[1][wave] = c1; s_w[2][wave] = c2; s_w[3][wave] = c3; }
      __syncthreads();
      c0 = c1 = c2 = c3 = 0;
      for (int w = 0; w < NW; w++) { c0 += s_w[0][w]; c1 += s_w[1][w]; c2 += s_w[2][w]; c3 += s_w[3][w]; }
      __syncthreads();
    }
    // (the same number in every lane, but made by shuffles: said to be uniform, the class boundaries live in scalar registers -- as vector registers they are live through the
    // whole kernel and are what the register allocator spills to scratch, to be read back inside every level)
    c0 = __builtin_amdgcn_readfirstlane(c0); c1 = __builtin_amdgcn_readfirstlane(c1); c2 = __builtin_amdgcn_readfirstlane(c2);
    cOff[0] = 0; cOff[1] = c0; cOff[2] = c0 + c1; cOff[3] = c0 + c1 + c2; cOff[4] = P;
  }
  SYNC();
  BTICK(0);
  uint32_t nEntries = 0, nNodesTot = 0, sumD = 0, nVisits = 0;
  Node* nodesR = nullptr; Ent* entR = nullptr; uint32_t* apR = nullptr; int2* stkR = nullptr; uint2* visR = nullptr; long long* edR = nullptr;
  uint32_t blkPair = 0;
  if (EMIT) {
    const ReadArena A = a.ra[rr];
    char* b = arena_ptr(A.base);
    blkPair = A.blkPair;
    nodesR = (Node*)b; entR = (Ent*)(b + A.entOff); apR = (uint32_t*)(b + A.apOff); stkR = (int2*)(b + A.stkOff); visR = (uint2*)(b + A.visOff);
    edR = (long long*)(b + A.edOff);
  }
  bool overflow = false, outgrown = false;
  const uint64_t capE = EMIT ? a.cntEntries[rr] : 0, capN = EMIT ? a.cntNodes[rr] : 0, capD = EMIT ? a.cntD[rr] : 0;
  for (int fam = 0; fam < 4 && !outgrown; fam++) {
    // family switches (DivideSubBy{Row1,Col1,Row2,Col2}.h): R1, C1, R2, C2
    const bool col = fam & 1, back = fam >= 2, desc = (fam == 1 || fam == 2), swapped = (fam == 3);
    const IT* lineOf = col ? colOf : rowOf;
    const int nLines = col ? C : R;
    const int sc = back ? 2 : 0;
    const int nS = cOff[sc + 1] - cOff[sc], nEn = cOff[sc + 2] - cOff[sc + 1], Pf = nS + nEn;
    if (nS == 0 || nEn == 0) continue;
    const int fam2 = fam & 1;
    const int dSide = swapped ? 1 : 0, eSide = swapped ? 0 : 1;
    for (int i = tid; i < Pf; i += NT) {
      const uint32_t pos = pay3[cOff[sc] + i];
      lp[i] = (IT)pos; ln[i] = 0; ll[i] = lineOf[pos];
      ld[i] = (DT)(back ? (long long)ht[pos] + hq[pos] : (long long)ht[pos] - hq[pos]);
    }
    if (tid == 0) { TB(0, F_LS, 0) = 0; TB(0, F_LE, 0) = nLines; TB(0, F_SB, 0) = 0; TB(0, F_SE, 0) = nS; TB(0, F_EB, 0) = nS; TB(0, F_EE, 0) = Pf; }
    int nNodes = 1, cur = 0;
    SYNC();
    BTICK(7);
    for (int level = 0; nNodes > 0 && !outgrown; level++) {
      if (level >= LV) { overflow = true; break; }
      const int nxt = cur ^ 1;
      IT* lpc = lp + cur * P; IT* lpn = lp + nxt * P;
      IT* llc = ll + cur * P; IT* lln = ll + nxt * P;
      DT* ldc = ld + (size_t)cur * P; DT* ldn = ld + (size_t)nxt * P;
      // A: which elements go to the first half of their node's lines; exclusive prefix in pf
      {
        int carry = 0;
        // (UA chunks of the pass at a time: their loads -- node of the element, the node's line range, the element's line -- are all asked for before anything is
        // waited for or stored, so a chunk costs a third of a dependent chain instead of a whole one; the pass is a latency chain per wave, not a stream)
        constexpr int UA = 4;
        for (int i0 = 0; i0 < Pf; i0 += UA * NT) {
          uint32_t kk[UA]; uint32_t ss[UA], ee[UA]; IT lv[UA]; int ff[UA];
#pragma unroll
          for (int u = 0; u < UA; u++) { const int i = i0 + u * NT + tid; kk[u] = i < Pf ? LN(i) : NONE; lv[u] = i < Pf ? llc[i] : (IT)0; }
#pragma unroll
          for (int u = 0; u < UA; u++) { ss[u] = 0; ee[u] = 0; if (kk[u] != NONE) { ss[u] = TB(cur, F_LS, kk[u]); ee[u] = TB(cur, F_LE, kk[u]); } }
#pragma unroll
          for (int u = 0; u < UA; u++) ff[u] = kk[u] == NONE ? 0 : (ee[u] - ss[u] > 1) ? (lv[u] < ((ss[u] + ee[u]) >> 1)) : 1;
#pragma unroll
          for (int u = 0; u < UA; u++) {
            const int i = i0 + u * NT + tid;
            if (i0 + u * NT >= Pf) break;
            int tot; const int inc = blk_incl_scan<NW>(ff[u], lane, wave, s_w[0], tot);
            if (i < Pf) pf[i] = (IT)(carry + inc - ff[u]);
            carry += tot;
          }
        }
        if (tid == 0) pf[Pf] = (IT)carry;
      }
      SYNC();
      BTICK(1);
      for (int k = tid; k < nNodes; k += NT) {
        TM(T_C1S, k) = (uint32_t)pf[TB(cur, F_SE, k)] - (uint32_t)pf[TB(cur, F_SB, k)];
        TM(T_C1E, k) = (uint32_t)pf[TB(cur, F_EE, k)] - (uint32_t)pf[TB(cur, F_EB, k)];
      }
      SYNC();
      // C: stable partition of every node's two segments
      {
        // (a node's two segments are partitioned inside their own ranges, so an element's slot ln[P + .] is written either here (no node) or by the element that moves
        // into it, never both: the chunks of a group may be read before any of them is written)
        constexpr int UC = 2;
        for (int i0 = 0; i0 < Pf; i0 += UC * NT) {
          uint32_t kk[UC], pi0[UC], pi1[UC], sb[UC], c1[UC], psb[UC]; IT vlp[UC], vll[UC]; DT vld[UC];
#pragma unroll
          for (int u = 0; u < UC; u++) {
            const int i = i0 + u * NT + tid;
            kk[u] = NONE; pi0[u] = 0; pi1[u] = 0; vlp[u] = 0; vll[u] = 0; vld[u] = 0;
            if (i < Pf) { kk[u] = LN(i); pi0[u] = (uint32_t)pf[i]; pi1[u] = (uint32_t)pf[i + 1]; vlp[u] = lpc[i]; vll[u] = llc[i]; vld[u] = ldc[i]; }
          }
#pragma unroll
          for (int u = 0; u < UC; u++) {
            const int i = i0 + u * NT + tid;
            const bool isS = i < nS;
            sb[u] = 0; c1[u] = 0;
            if (kk[u] != NONE) { sb[u] = isS ? TB(cur, F_SB, kk[u]) : TB(cur, F_EB, kk[u]); c1[u] = isS ? TM(T_C1S, kk[u]) : TM(T_C1E, kk[u]); }
          }
#pragma unroll
          for (int u = 0; u < UC; u++) psb[u] = kk[u] != NONE ? (uint32_t)pf[sb[u]] : 0;
#pragma unroll
          for (int u = 0; u < UC; u++) {
            const int i = i0 + u * NT + tid;
            if (i >= Pf) continue;
            if (kk[u] == NONE) { ln[P + i] = INONE; continue; }
            const uint32_t rank1 = pi0[u] - psb[u];
            const uint32_t first = pi1[u] - pi0[u];
            const uint32_t np_ = first ? sb[u] + rank1 : sb[u] + c1[u] + ((uint32_t)i - sb[u] - rank1);
            lpn[np_] = vlp[u]; lln[np_] = vll[u]; ldn[np_] = vld[u];
            ln[P + np_] = (IT)(2 * kk[u] + (first ? 0 : 1));
          }
        }
      }
      SYNC();
      BTICK(2);
      // D: heads of the distinct diagonals inside the D segment (ends) / E segment (starts); exclusive prefix in ph
      {
        int carry = 0;
        constexpr int UD = NW == 1 ? 4 : 2;
        for (int j0 = 0; j0 < Pf; j0 += UD * NT) {
          uint32_t k2v[UD], le[UD], ls[UD], bg[UD], c1[UD]; DT d0[UD], d1[UD]; int hd[UD];
#pragma unroll
          for (int u = 0; u < UD; u++) {
            const int j = j0 + u * NT + tid;
            k2v[u] = NONE; d0[u] = 0; d1[u] = 0;
            if (j < Pf) { k2v[u] = LN(P + j); d0[u] = ldn[j]; d1[u] = j > 0 ? ldn[j - 1] : ldn[j]; }
          }
#pragma unroll
          for (int u = 0; u < UD; u++) {
            const int j = j0 + u * NT + tid;
            const bool isS = j < nS;
            le[u] = 0; ls[u] = 0; bg[u] = 0; c1[u] = 0;
            if (k2v[u] != NONE) {
              const uint32_t k = k2v[u] >> 1;
              le[u] = TB(cur, F_LE, k); ls[u] = TB(cur, F_LS, k); bg[u] = isS ? TB(cur, F_SB, k) : TB(cur, F_EB, k); c1[u] = isS ? TM(T_C1S, k) : TM(T_C1E, k);
            }
          }
#pragma unroll
          for (int u = 0; u < UD; u++) {
            const int j = j0 + u * NT + tid;
            hd[u] = 0;
            if (k2v[u] != NONE) {
              const uint32_t side = k2v[u] & 1;
              const bool isS = j < nS;
              const bool leaf = le[u] - ls[u] == 1;
              const bool in = leaf || (int)side == (isS ? eSide : dSide);
              if (in) {
                uint32_t beg = bg[u];
                if (!leaf && side == 1) beg += c1[u];
                hd[u] = ((uint32_t)j == beg) ? 1 : (d0[u] != d1[u]);
              }
            }
          }
#pragma unroll
          for (int u = 0; u < UD; u++) {
            const int j = j0 + u * NT + tid;
            if (j0 + u * NT >= Pf) break;
            int tot; const int inc = blk_incl_scan<NW>(hd[u], lane, wave, s_w[0], tot);
            if (j < Pf) ph[j] = (IT)(carry + inc - hd[u]);
            carry += tot;
          }
        }
        if (tid == 0) ph[Pf] = (IT)carry;
      }
      SYNC();
      BTICK(3);
      // E: per node: sizes, fullness, children, next level's table
      int nNext = 0;
      for (int k0 = 0; k0 < nNodes; k0 += NT) {
        const int k = k0 + tid;
        uint32_t nD = 0, nE = 0, act0 = 0, act1 = 0, full = 0;
        uint32_t ls = 0, le = 0, sb = 0, se = 0, eb = 0, ee = 0, c1S = 0, c1E = 0;
        bool leaf = false;
        if (k < nNodes) {
          ls = TB(cur, F_LS, k); le = TB(cur, F_LE, k); sb = TB(cur, F_SB, k); se = TB(cur, F_SE, k); eb = TB(cur, F_EB, k); ee = TB(cur, F_EE, k);
          c1S = TM(T_C1S, k); c1E = TM(T_C1E, k);
          leaf = le - ls == 1;
          uint32_t dB, dE, eB, eE;
          if (leaf) { dB = eb; dE = ee; eB = sb; eE = se; }
          else {
            dB = dSide == 0 ? eb : eb + c1E; dE = dSide == 0 ? eb + c1E : ee;
            eB = eSide == 0 ? sb : sb + c1S; eE = eSide == 0 ? sb + c1S : se;
          }
          nD = (uint32_t)ph[dE] - (uint32_t)ph[dB]; nE = (uint32_t)ph[eE] - (uint32_t)ph[eB];
          full = nD > 0 && nE > 0;
          if (!leaf) {                                                   // DivideSubProbBy*: which halves are explored
            const bool goD = nD > 0, goE = nE > 0;                       // both empty: none; only Di: D half; only Ei: E half; else both
            if (dSide == 0) { act0 = goD; act1 = goE; } else { act1 = goD; act0 = goE; }
          }
        }
        int totF, totEnt, totD, totC;
        const int incF = blk_incl_scan<NW>((int)full, lane, wave, s_w[0], totF), incEnt = blk_incl_scan<NW>((int)(full ? nD + nE : 0), lane, wave, s_w[1], totEnt),
                  incD = blk_incl_scan<NW>((int)(full ? nD : 0), lane, wave, s_w[2], totD), incC = blk_incl_scan<NW>((int)(act0 + act1), lane, wave, s_w[3], totC);
        // the read's blocks may have been laid out from an estimate (k_arena_estimate): nothing is written past them -- the level is abandoned (the levels before it are
        // complete, and nothing points at this one yet) and the read is built again from exact counts
        if (EMIT && ((uint64_t)nNodesTot + totF > capN || (uint64_t)nEntries + totEnt > capE || (uint64_t)sumD + totD > capD)) { outgrown = true; break; }
        if (k < nNodes) {
          const uint32_t gid = nNodesTot + incF - full, base = nEntries + incEnt - (full ? nD + nE : 0), dpre = sumD + incD - (full ? nD : 0);
          TM(T_ND, k) = nD; TM(T_NE, k) = nE; TM(T_GID, k) = full ? gid : NONE; TM(T_BASE, k) = base;
          uint32_t ci = nNext + incC - (act0 + act1);
          const uint32_t med = (ls + le) >> 1;
          TM(T_CH0, k) = NONE; TM(T_CH1, k) = NONE;
          if (act0) { TM(T_CH0, k) = ci; TB(nxt, F_LS, ci) = ls; TB(nxt, F_LE, ci) = med; TB(nxt, F_SB, ci) = sb; TB(nxt, F_SE, ci) = sb + c1S;
                      TB(nxt, F_EB, ci) = eb; TB(nxt, F_EE, ci) = eb + c1E; ci++; }
          if (act1) { TM(T_CH1, k) = ci; TB(nxt, F_LS, ci) = med; TB(nxt, F_LE, ci) = le; TB(nxt, F_SB, ci) = sb + c1S; TB(nxt, F_SE, ci) = se;
                      TB(nxt, F_EB, ci) = eb + c1E; TB(nxt, F_EE, ci) = ee; }
          if (EMIT && full) {
            Node nd;
            nd.dBase = base; nd.nD = nD; nd.nE = nE; nd.last = -1; nd.sTop = 1; nd.nBlk = 0;
            nd.stkOff = 2 * dpre + 4 * gid; nd.blkOff = blkPair + 2 * base + 8 * gid; nd.stkCap = 2 * nD + 4; nd.blkCap = 2 * (nD + nE) + 8;
            nd.eLast = 0;                                      // (written in F below, by the element that is the head of Ei[nE - 1])
            nodesR[gid] = nd;
            stkR[nd.stkOff] = make_int2(-1, (int)nE + 1);     // dummy pair (DivideSubByRow1.h:470)
          }
        }
        nNodesTot += totF; nEntries += totEnt; sumD += totD; nNext += totC;
      }
      if (outgrown) break;
      SYNC();
      BTICK(4);
      // F: node index of every element for the next level; emit Di / Ei and the visit records
      {
        // (the pass reads ln[P + .], the tables, ph, lpn and the points; it writes ln[.] below P, the entries and the visit rows: nothing it reads)
        constexpr int UF = 2;
        for (int j0 = 0; j0 < Pf; j0 += UF * NT) {
          uint32_t k2v[UF], le[UF], ls[UF], ch[UF], gidv[UF], bg[UF], c1[UF], nNE[UF], nND[UF], bs[UF], p0v[UF], p1v[UF], pbg[UF], posv[UF], tq[UF], tt[UF];
#pragma unroll
          for (int u = 0; u < UF; u++) {
            const int j = j0 + u * NT + tid;
            k2v[u] = NONE; p0v[u] = 0; p1v[u] = 0; posv[u] = 0;
            if (j < Pf) { k2v[u] = LN(P + j); if (EMIT) { p0v[u] = (uint32_t)ph[j]; p1v[u] = (uint32_t)ph[j + 1]; posv[u] = lpn[j]; } }
          }
#pragma unroll
          for (int u = 0; u < UF; u++) {
            const int j = j0 + u * NT + tid;
            const bool isS = j < nS;
            le[u] = 0; ls[u] = 0; ch[u] = 0; gidv[u] = NONE; bg[u] = 0; c1[u] = 0; nNE[u] = 0; nND[u] = 0; bs[u] = 0; tq[u] = 0; tt[u] = 0;
            if (k2v[u] != NONE) {
              const uint32_t k = k2v[u] >> 1, side = k2v[u] & 1;
              le[u] = TB(cur, F_LE, k); ls[u] = TB(cur, F_LS, k); ch[u] = side == 0 ? TM(T_CH0, k) : TM(T_CH1, k); gidv[u] = TM(T_GID, k);
              if (EMIT) {
                bg[u] = isS ? TB(cur, F_SB, k) : TB(cur, F_EB, k); c1[u] = isS ? TM(T_C1S, k) : TM(T_C1E, k);
                nNE[u] = TM(T_NE, k); nND[u] = TM(T_ND, k); bs[u] = TM(T_BASE, k);
                tq[u] = hq[posv[u]]; tt[u] = ht[posv[u]];
              }
            }
          }
#pragma unroll
          for (int u = 0; u < UF; u++) {
            pbg[u] = 0;
            if (EMIT && k2v[u] != NONE) {
              const uint32_t side = k2v[u] & 1;
              const bool leaf = le[u] - ls[u] == 1;
              uint32_t beg = bg[u];
              if (!leaf && side == 1) beg += c1[u];
              bg[u] = beg;
              pbg[u] = (uint32_t)ph[beg];
            }
          }
#pragma unroll
          for (int u = 0; u < UF; u++) {
            const int j = j0 + u * NT + tid;
            if (j >= Pf) continue;
            if (k2v[u] == NONE) { ln[j] = INONE; continue; }
            const uint32_t side = k2v[u] & 1;
            const bool leaf = le[u] - ls[u] == 1;
            ln[j] = (IT)(leaf ? NONE : ch[u]);
            const uint32_t gid = gidv[u];
            const bool isS = j < nS;
            const bool in = leaf || (int)side == (isS ? eSide : dSide);
            if (in && gid != NONE) {
              if (!EMIT) nVisits++;
              else {
                const uint32_t head = p1v[u] - p0v[u];
                const uint32_t grp = p0v[u] - pbg[u] + head - 1;
                const uint32_t n = isS ? nNE[u] : nND[u];
                const uint32_t idx = desc ? n - 1 - grp : grp;
                const uint32_t ent = bs[u] + (isS ? nND[u] + idx : idx);
                const uint32_t pos = posv[u];
                if (head) {
                  const long long dgv = back ? (long long)tt[u] + tq[u] : (long long)tt[u] - tq[u];   // (the element's diagonal, from its point: ldn may hold 32 bits of it)
                  entR[ent].val = dgv;
                  if (isS && idx == n - 1) nodesR[gid].eLast = dgv;
                }
                visR[(uint64_t)pos * (2 * LV) + fam2 * LV + level] = make_uint2(gid, idx);
              }
            }
          }
        }
      }
      SYNC();
      BTICK(5);
      // G: Db / Eb in closed form (Decide_Eb_Db_*), values and back pointers zeroed
      if (EMIT) {
        // (reads: ln[P + .], the tables, ph, the .val fields of the level's entries (written by F, above the barrier); writes: the .b / .v fields, Ei[Db], the back
        // pointers -- so the binary searches of UG chunks run side by side, a probe of each per round)
        constexpr int UG = 2;
        for (int j0 = 0; j0 < Pf; j0 += UG * NT) {
          uint32_t k2v[UG], gidv[UG], le[UG], ls[UG], bg[UG], c1[UG], nDv[UG], nEv[UG], bs[UG], p0v[UG], p1v[UG], pbg[UG];
          bool on[UG];
#pragma unroll
          for (int u = 0; u < UG; u++) {
            const int j = j0 + u * NT + tid;
            k2v[u] = NONE; p0v[u] = 0; p1v[u] = 0;
            if (j < Pf) { k2v[u] = LN(P + j); p0v[u] = (uint32_t)ph[j]; p1v[u] = (uint32_t)ph[j + 1]; }
          }
#pragma unroll
          for (int u = 0; u < UG; u++) {
            const int j = j0 + u * NT + tid;
            const bool isS = j < nS;
            gidv[u] = NONE; le[u] = 0; ls[u] = 0; bg[u] = 0; c1[u] = 0; nDv[u] = 0; nEv[u] = 0; bs[u] = 0;
            if (k2v[u] != NONE && p1v[u] != p0v[u]) {
              const uint32_t k = k2v[u] >> 1;
              gidv[u] = TM(T_GID, k); le[u] = TB(cur, F_LE, k); ls[u] = TB(cur, F_LS, k);
              bg[u] = isS ? TB(cur, F_SB, k) : TB(cur, F_EB, k); c1[u] = isS ? TM(T_C1S, k) : TM(T_C1E, k);
              nDv[u] = TM(T_ND, k); nEv[u] = TM(T_NE, k); bs[u] = TM(T_BASE, k);
            }
          }
#pragma unroll
          for (int u = 0; u < UG; u++) {
            const int j = j0 + u * NT + tid;
            const bool isS = j < nS;
            const uint32_t side = k2v[u] & 1;
            const bool leaf = le[u] - ls[u] == 1;
            on[u] = k2v[u] != NONE && p1v[u] != p0v[u] && gidv[u] != NONE && (leaf || (int)side == (isS ? eSide : dSide));
            pbg[u] = 0;
            if (on[u]) { uint32_t beg = bg[u]; if (!leaf && side == 1) beg += c1[u]; pbg[u] = (uint32_t)ph[beg]; }
          }
          uint32_t entv[UG], mv[UG], lo[UG], cnt[UG]; long long xv[UG]; const Ent* opp[UG];
#pragma unroll
          for (int u = 0; u < UG; u++) {
            const int j = j0 + u * NT + tid;
            const bool isS = j < nS;
            const uint32_t grp = p0v[u] - pbg[u];
            const uint32_t n = isS ? nEv[u] : nDv[u];
            const uint32_t idx = desc ? n - 1 - grp : grp;
            entv[u] = bs[u] + (isS ? nDv[u] + idx : idx);
            opp[u] = entR + bs[u] + (isS ? 0 : nDv[u]);
            mv[u] = isS ? nDv[u] : nEv[u];
            xv[u] = on[u] ? entR[entv[u]].val : 0;
            lo[u] = 0; cnt[u] = on[u] ? mv[u] : 0;
          }
          // D entry: asc  #{Ei < x}   desc #{Ei >= x};   E entry: asc #{Di <= x}   desc #{Di > x}
          // (the lists are sorted and the predicate holds on a prefix: two steps of the bisection per round -- the probe in the middle and the two probes its outcome can lead
          // to are asked for together; a round is a trip to L2 and a level of the decomposition has a dozen of these searches per element group)
          while (true) {
            bool any = false;
#pragma unroll
            for (int u = 0; u < UG; u++) any |= cnt[u] > 0;
            if (!any) break;
            long long vM[UG], vL[UG], vR[UG];
#pragma unroll
            for (int u = 0; u < UG; u++) {
              const uint32_t step = cnt[u] >> 1, it = lo[u] + step, cntT = cnt[u] > 0 ? cnt[u] - step - 1 : 0;
              vM[u] = cnt[u] > 0 ? opp[u][it].val : 0;
              vL[u] = step > 0 ? opp[u][lo[u] + (step >> 1)].val : 0;
              vR[u] = cntT > 0 ? opp[u][it + 1 + (cntT >> 1)].val : 0;
            }
#pragma unroll
            for (int u = 0; u < UG; u++) {
              if (cnt[u] == 0) continue;
              const bool isS = j0 + u * NT + tid < nS;
              auto go = [&](long long v) { return isS ? (desc ? v > xv[u] : v <= xv[u]) : (desc ? v >= xv[u] : v < xv[u]); };
              const uint32_t step = cnt[u] >> 1, it = lo[u] + step;
              if (go(vM[u])) {
                lo[u] = it + 1; cnt[u] -= step + 1;
                if (cnt[u] > 0) { const uint32_t s2 = cnt[u] >> 1; if (go(vR[u])) { lo[u] += s2 + 1; cnt[u] -= s2 + 1; } else cnt[u] = s2; }
              } else {
                cnt[u] = step;
                if (cnt[u] > 0) { const uint32_t s2 = cnt[u] >> 1; if (go(vL[u])) { lo[u] += s2 + 1; cnt[u] -= s2 + 1; } else cnt[u] = s2; }
              }
            }
          }
#pragma unroll
          for (int u = 0; u < UG; u++) {
            if (!on[u]) continue;
            const bool isS = j0 + u * NT + tid < nS;
            const uint32_t ent = entv[u], m = mv[u];
            entR[ent].b = isS ? (int32_t)lo[u] - 1 : (lo[u] == m ? -1 : (int32_t)lo[u]);
            if (!isS) edR[ent] = lo[u] == m ? 0 : opp[u][lo[u]].val;        // Ei[Db[d]]
            // (Ev[] is written but never read by the reference, and Db[Eb + 1] -- which the flush at the end of Maximization tests against the top pair's boundary,
            // :450 -- is never needed: that boundary is n or n + 1, see sdp_process_wg)
            entR[ent].v = 0.f; apR[ent] = 0;
          }
        }
      }
      nNodes = nNext; cur = nxt;
      SYNC();
      BTICK(6);
    }
  }
  if (bstat && tid == 0) { for (int k = 0; k < 8; k++) atomicAdd(a.stat + k, s_bt[k]); atomicAdd(a.stat + 8, (unsigned long long)P); }
  for (int o = 32; o > 0; o >>= 1) nVisits += __shfl_xor(nVisits, o);
  if (NW > 1) {
    if (lane == 0) s_w[0][wave] = (int)nVisits;
    __syncthreads();
    nVisits = 0;
    for (int w = 0; w < NW; w++) nVisits += (uint32_t)s_w[0][w];
  }
  if (tid == 0) {
    if (!EMIT) { a.cntEntries[rr] = nEntries; a.cntNodes[rr] = nNodesTot; a.cntD[rr] = sumD; a.cntV[rr] = nVisits; a.cntRC[rr] = (uint32_t)max(R, C); }
    else { a.cntV[rr] = nEntries; a.cntRC[rr] = (uint32_t)max(R, C); }      // (what the read really has; its rows / columns for the choice of the ProcessPoint kernel)
    if (overflow) atomicOr(&a.status[r], (uint32_t)LRA_ST_RANGE);             // more than 2^(LV-1) distinct rows / columns
    if (outgrown) atomicOr(&a.status[r], (uint32_t)LRA_ST_CAPACITY);
  }
#undef TB
#undef TM
#undef BTICK
}

// ---- ProcessPoint -----------------------------------------------------------------------------------------------------
struct ProcArgs {
  int r0, n;
  const uint32_t* order;
  const uint64_t* ptOff; const uint64_t* fragOff;
  const uint8_t* hfl; const uint32_t* hfr;
  const int32_t* flen; float* fval; uint32_t* fprevNode; uint32_t* fprevInd; uint8_t* fflags;
  const float* rate_in; float rate;
  const ReadArena* ra; uint32_t* poolUsed;
  uint32_t* status;
  PwlTab pwl;
  const short* penTab; int penN;           // -w(|d| + 1) for d < penN (k_pen_table); penN = 0: no table
  int dbg;
  int wgNoRing;          // sdp_process_wg: keep the anchors' words at L2 whatever the spans (LRA_SDP_WG_RING=0: tests of that mode)
  char* wgScratch; const uint64_t* wgOff;   // sdp_process_wg: per large read, the anchors' (best predecessor, contributions) words and the points' ranks
  unsigned long long* stat;                 // sdp_process<true> (LRA_SDP_STAT): 32 counters summed over the launch's waves
};

// w(i, j) = -PWL_w(|j - i| + 1)   (SubRountine.h:101-129).  upper_bound over STOPS[0..24) as a count of constants <= x.
// Every visit of ProcessPoint evaluates this a handful of times and a wave runs them one after the other, so it is written for few instructions: the count of
// stops below 1000 in nine compares, one division between 1000 and 9999, five compares beyond; 32-bit conversions wherever the values fit (the reference's
// (long)(float) and (float)(long) give the same numbers there: both truncate / round the same value).
__device__ __forceinline__ float pwl_w(const float* slope, const float* inter, int c1, int c2, long long i, long long j) {
  const long long x = (j > i ? j - i : i - j) + 1;
  if (x <= 2) return x == 1 ? 0.f : -0.f;
  int b; float xf;
  if (x <= 0x7fffffffLL) {
    const int xs = (int)x;
    if (xs < 1000) b = 1 + (xs >= 5) + (xs >= 10) + (xs >= 20) + (xs >= 40) + (xs >= 80) + (xs >= 100) + (xs >= 200) + (xs >= 300) + (xs >= 500);
    else if (xs < 10000) b = 10 + xs / 1000;                              // stops 1000, 2000, ..., 9000
    else b = 19 + (xs >= 15000) + (xs >= 20000) + (xs >= 30000) + (xs >= 40000) + (xs >= 50000);
    xf = (float)xs;
  } else { b = 24; xf = (float)x; }
  const float f = slope[b - 1] * xf + inter[b - 1];
  if (f > -2.0e9f && f < 2.0e9f) {
    int pen = (int)f;
    if (pen >= c1 && pen < c2) pen = c1;
    else if (pen > c2) pen = c2;
    return -(float)pen;
  }
  long long pen = (long long)f;
  if (pen >= c1 && pen < c2) pen = c1;
  else if (pen > c2) pen = c2;
  return -(float)pen;
}

// The same through a table: -w(i, j) for |j - i| + 1 < n as 16-bit integers in LDS (every penalty is an integer: PWL_w truncates), made once per call by
// k_pen_table with pwl_w itself.  A visit evaluates w ten times and more; from the table that is one LDS read instead of ~50 instructions and two reads.
__device__ __forceinline__ float pwl_w_tab(const short* tab, int n, const float* slope, const float* inter, int c1, int c2, long long i, long long j) {
  const long long d = j > i ? j - i : i - j;
  if (d < (long long)n) { const int xi = (int)d; return xi == 0 ? 0.f : -(float)(int)tab[xi]; }   // (x == 1: w returns +0, SubRountine.h:125)
  return pwl_w(slope, inter, c1, c2, i, j);
}
// a + w(x, e) > b + w(y, e) -- the comparison Maximization / FindBoundary make (SubRountine.h:253, :292, :299) -- with the two table reads issued together (one LDS
// round trip instead of two: the wave walks these one after the other)
__device__ __forceinline__ bool pwl_beats(const short* tab, int n, const float* slope, const float* inter, int c1, int c2, float a, long long x, float b, long long y, long long e) {
  const long long d1 = e > x ? e - x : x - e, d2 = e > y ? e - y : y - e;
  const bool in1 = d1 < (long long)n, in2 = d2 < (long long)n;
  const int x1 = in1 ? (int)d1 : 0, x2 = in2 ? (int)d2 : 0;
  const int p1 = tab[x1], p2 = tab[x2];
  const float w1 = in1 ? (x1 == 0 ? 0.f : -(float)p1) : pwl_w(slope, inter, c1, c2, x, e);
  const float w2 = in2 ? (x2 == 0 ? 0.f : -(float)p2) : pwl_w(slope, inter, c1, c2, y, e);
  return a + w1 > b + w2;
}
__global__ void k_pen_table(PwlTab pw, int n, short* tab, int* bad) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= n) return;
  const float w = pwl_w(pw.slope, pw.inter, pw.c1, pw.c2, 0, (long long)d);      // x = d + 1
  const float p = -w;                                                            // the penalty: an integer
  if (!(p >= 0.f && p <= 32767.f) || (float)(int)p != p) { atomicOr(bad, 1); tab[d] = 0; return; }
  tab[d] = (short)(int)p;
}
constexpr int PEN_TAB_WG = 4096, PEN_TAB_WAVE = 2048;

// The maximum of a float over the 64 lanes, wave-uniform: four row shifts (a row = 16 lanes; lanes with nothing shifted in keep -inf), the two row broadcasts, lane 63.
__device__ __forceinline__ float wave_max_f32(float v) {
  constexpr int NEG_INF = (int)0xff800000u;
  int x = __float_as_int(v);
#define LRA_DPP_MAX(ctrl_, rmask_) x = __float_as_int(fmaxf(__int_as_float(x), __int_as_float(__builtin_amdgcn_update_dpp(NEG_INF, x, (ctrl_), (rmask_), 0xf, false))))
  LRA_DPP_MAX(0x111, 0xf);   // row_shr:1
  LRA_DPP_MAX(0x112, 0xf);   // row_shr:2
  LRA_DPP_MAX(0x114, 0xf);   // row_shr:4
  LRA_DPP_MAX(0x118, 0xf);   // row_shr:8      (lane 15 of a row: the row's maximum)
  LRA_DPP_MAX(0x142, 0xa);   // row_bcast:15   (rows 1 and 3 take the row before them)
  LRA_DPP_MAX(0x143, 0xc);   // row_bcast:31   (rows 2 and 3 take lane 31)
#undef LRA_DPP_MAX
  return __int_as_float(__builtin_amdgcn_readlane(x, 63));
}
__device__ __forceinline__ int rl_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }           // src must be wave-uniform
__device__ __forceinline__ float rl_f(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
__device__ __forceinline__ long long rl_ll(long long v, int src) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(v & 0xffffffffLL), src), hi = (unsigned)__builtin_amdgcn_readlane((int)(v >> 32), src);
  return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ long long shfl_ll(long long v, int src) {
  const int lo = __shfl((int)(v & 0xffffffffLL), src), hi = __shfl((int)(v >> 32), src);
  return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

// Wave-uniform values (every lane computes the same number): moved to scalar registers.  The sparse DP's workgroup kernel is almost entirely uniform control
// (one visit = a serial walk the whole wave follows); left in vector registers its state overflows the 128 a wave of a 1024-thread block may hold, and a reload from
// scratch waits behind every pending store of the walk (vector memory completes in order).  SGPRs spill into VGPR lanes instead: no memory.
__device__ __forceinline__ int u_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t u_u(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ float u_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ long long u_ll(long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffLL)), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int2 u_i2(int2 v) { return make_int2(u_i(v.x), u_i(v.y)); }
__device__ __forceinline__ uint2 u_u2(uint2 v) { return make_uint2(u_u(v.x), u_u(v.y)); }

// The literal binary search  `while (count > 0) { step = count / 2; it = first + step; if (pred(it)) { first = it + 1; count -= step + 1; }
// else count = step; }`  (FindBoundary :245-254, UPPERbound :209-219), six levels per memory round: lane t = 1..63 evaluates the
// predicate at the probe the search would make after taking the decisions spelled by t's bits; the wave then walks the 63 answers.
template <typename Pred>
__device__ __forceinline__ unsigned coop_search(unsigned first, unsigned count, int lane, Pred pred) {
  while (count > 0) {
    unsigned f = first, c = count;
    bool valid = lane >= 1;
    if (valid) {
      const int depth = 31 - __clz(lane);
      for (int d = depth - 1; d >= 0; --d) {
        if (c == 0) { valid = false; break; }
        const unsigned step = c / 2, it = f + step;
        if ((lane >> d) & 1) { f = it + 1; c -= step + 1; } else c = step;
      }
    }
    const bool p = (valid && c > 0) ? pred(f + c / 2) : false;
    const unsigned long long m = __ballot(p);
    unsigned t = 1;
    while (t < 64 && count > 0) {
      const unsigned step = count / 2, it = first + step;
      const unsigned bit = (unsigned)((m >> t) & 1);
      if (bit) { first = it + 1; count -= step + 1; } else count = step;
      t = 2 * t + bit;
    }
  }
  return first;
}

// FindValueInBlock's UPPERbound (:205-221) over Block, the same six levels per round.  The search ends at its right boundary, and the right boundary is the
// position of its most recent probe that came out false (or the end of the list): that probe's lane still holds the pair, so Block[lo].first comes with the
// search instead of costing one more dependent load.  Returns lo; *x = Block[lo].x when lo < count.
__device__ __forceinline__ unsigned coop_upper_block(const int2* B, unsigned count0, int i1, int lane, int* x) {
  unsigned first = 0, count = count0;
  int bx = -1;
  while (count > 0) {
    unsigned f = first, c = count;
    bool valid = lane >= 1;
    if (valid) {
      const int depth = 31 - __clz(lane);
      for (int d = depth - 1; d >= 0; --d) {
        if (c == 0) { valid = false; break; }
        const unsigned step = c / 2, it = f + step;
        if ((lane >> d) & 1) { f = it + 1; c -= step + 1; } else c = step;
      }
    }
    int2 pr = make_int2(0, 0);
    const bool live = valid && c > 0;
    if (live) pr = B[f + c / 2];
    const bool p = live && i1 >= pr.y;
    const unsigned long long m = __ballot(p);
    unsigned t = 1;
    int lastFalse = -1;
    while (t < 64 && count > 0) {
      const unsigned step = count / 2, it = first + step;
      const unsigned bit = (unsigned)((m >> t) & 1);
      if (bit) { first = it + 1; count -= step + 1; } else { count = step; lastFalse = (int)t; }
      t = 2 * t + bit;
    }
    if (lastFalse >= 0) bx = __builtin_amdgcn_readlane(pr.x, lastFalse);
  }
  *x = bx;
  return u_u(first);
}

// a stack / Block that is full moves to twice the room in the read's pool (the old room is abandoned)
__device__ bool grow_pairs(int2* pairs, uint32_t& off, int& cap, int used, uint32_t* poolUsed, uint32_t poolPair, uint32_t poolPairs) {
  const uint32_t ncap = 2u * (uint32_t)cap;
  const uint32_t at = atomicAdd(poolUsed, ncap);
  if (at + ncap > poolPairs) return false;
  int2* dst = pairs + poolPair + at; const int2* src = pairs + off;
  for (int k = 0; k < used && k < cap; k++) dst[k] = src[k];
  off = poolPair + at; cap = (int)ncap;
  return true;
}
__device__ bool coop_grow_pairs(int2* pairs, uint32_t& off, int& cap, int used, uint32_t* poolUsed, uint32_t poolPair, uint32_t poolPairs, int lane) {
  const uint32_t ncap = 2u * (uint32_t)cap;
  uint32_t at = 0;
  if (lane == 0) at = atomicAdd(poolUsed, ncap);
  at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
  if (at + ncap > poolPairs) return false;
  int2* dst = pairs + poolPair + at; const int2* src = pairs + off;
  for (int k = lane; k < used && k < cap; k += 64) dst[k] = src[k];
  wave_sync();
  off = poolPair + at; cap = (int)ncap;
  return true;
}

// One wave per read.  The points are walked in H1 order (ProcessPoint :1015-1171); lane (family pair, level) < 32 owns the
// sub-problem the point touches on that level.  End points (PassValueToD*) are one independent update per lane.  For a start
// point the lanes whose sub-problem has a usable Eb take turns as owner of a wave-cooperative Maximization (:270-345): the
// owner's state is broadcast, all lanes run the (sequential) candidate-list loop in lock step (every lane issues the same stack /
// Block stores, so each sees its own) with the next 64 Di / Dv / Db and
// Ei[Db] prefetched one per lane, and the two binary searches (FindBoundary, FindValueInBlock's UPPERbound) probe six levels
// per memory round.  Value[ii] is then the (max value, first in visit order) reduction the ordered `val < Ev` updates compute.
template <bool STAT>
__global__ void __launch_bounds__(64, 4) sdp_process(ProcArgs a) {
  // STAT (LRA_SDP_STAT): cycles per section of a point's visit and the lengths of its loops, summed over the launch -- an instantiation of its own (the counters' registers)
  __shared__ unsigned long long sT[STAT ? 10 : 1], sC[STAT ? 20 : 1];     // (in LDS: the production kernel's registers are what the counters would take)
  unsigned long long tPrev = 0;
  if (STAT) { if (threadIdx.x < 10) sT[threadIdx.x] = 0; if (threadIdx.x < 20) sC[threadIdx.x] = 0; tPrev = __builtin_amdgcn_s_memtime(); }
#define TICK(k_) do { if (STAT) { if (a.dbg == 2) __builtin_amdgcn_s_waitcnt(0); const unsigned long long t__ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) sT[k_] += t__ - tPrev; tPrev = t__; } } while (0)
#define CNT(k_, x_) do { const unsigned long long x__ = (unsigned long long)(x_); if (threadIdx.x == 0) sC[k_] += x__; } while (0)
#define WMAX(x_) ([&]() { int m__ = (x_); for (int o__ = 32; o__ > 0; o__ >>= 1) m__ = max(m__, __shfl_xor(m__, o__)); return m__; }())
  __shared__ float s_slope[25], s_inter[25];
  __shared__ short s_pen[PEN_TAB_WAVE];
  const int lane = threadIdx.x;
  if (lane < 25) { s_slope[lane] = a.pwl.slope[lane]; s_inter[lane] = a.pwl.inter[lane]; }
  const int penN = min(a.penN, PEN_TAB_WAVE);
  for (int x = lane; x < penN; x += 64) s_pen[x] = a.penTab[x];
  __syncthreads();
  const int c1 = a.pwl.c1, c2 = a.pwl.c2;
#define W(i, j) pwl_w_tab(s_pen, penN, s_slope, s_inter, c1, c2, (i), (j))
#define BEATS(a_, x_, b_, y_, e_) pwl_beats(s_pen, penN, s_slope, s_inter, c1, c2, (a_), (x_), (b_), (y_), (e_))
  const int rr = (int)a.order[blockIdx.x], r = a.r0 + rr;
  if (a.status[r] & LRA_ST_CAPACITY) return;                             // the emit pass gave the read up (it outgrew its estimated blocks): it is built again
  const uint64_t p0 = a.ptOff[r], f0 = a.fragOff[r];
  const int P = (int)(a.ptOff[r + 1] - p0);
  const float rate = a.rate_in ? a.rate_in[r] : a.rate;
  const ReadArena A = a.ra[rr];
  char* ab = arena_ptr(A.base);
  Node* nodes = (Node*)ab;
  Ent* ent = (Ent*)(ab + A.entOff);
  uint32_t* Ap = (uint32_t*)(ab + A.apOff);
  const long long* Ed = (const long long*)(ab + A.edOff);
  int2* pairs = (int2*)(ab + A.stkOff);                                  // stacks, Blocks and the growth pool of this read
  const uint32_t poolPair = A.poolPair, poolPairs = A.poolPairs;
  uint32_t* poolUsed = a.poolUsed + rr;
  const uint2* visR = (const uint2*)(ab + A.visOff);
  const int fam2 = lane < LV ? 0 : 1, level = lane < LV ? lane : lane - LV;   // lanes >= 2 * LV have no visits
  uint32_t bad = 0;
  // per-lane cache of the sub-problem this lane touched last (descriptor, stack top, last Block pair): consecutive points mostly
  // stay in the same sub-problem on the upper levels
  Node cn; cn.dBase = 0; cn.nD = 0; cn.nE = 0; cn.last = -1; cn.sTop = 0; cn.nBlk = 0; cn.stkOff = 0; cn.blkOff = 0; cn.stkCap = 0; cn.blkCap = 0;
  uint32_t cId = NONE;
  int2 cTop = make_int2(0, 0), cLastB = make_int2(0, 0);
  bool cTopOk = false, cDirty = false;
  // The next point's flags and anchor are the same for every lane, and the compiler moves a wave-uniform value to a scalar register where it is MADE: a v_readfirstlane behind
  // the load, i.e. a wait for the load -- one whole memory round trip at the top of every point (a tenth of a point pair's time), with the visit row's load queued behind
  // it.  Read at an index the compiler cannot see through (a zero in a vector register), the two values stay in vector registers while they are in flight and become
  // scalars where they are used, one point later.
  int vz; asm volatile("v_mov_b32_e32 %0, 0" : "=v"(vz));
  uint32_t flN = P > 0 ? a.hfl[p0 + vz] : 0;
  uint32_t lfN = P > 0 ? a.hfr[p0 + vz] : 0;
  // The visit rows run two points ahead, so that at the top of a point the NEXT point's sub-problems are known and their descriptors can be asked for: straight into LDS
  // (global_load_lds_dwordx4: a lane's 16 bytes land at the base + 16 * lane, no vector register is held while the load is in flight -- the registers are what this
  // kernel is short of), three loads for the 48 bytes, two buffers taken in turn.  A lane that moves to another sub-problem finds the descriptor there instead of
  // starting a round trip (every end point's deepest lanes do); never the one the lane is in (newer in its registers than in memory), and one it has left was written back
  // above, ahead of the load.
  static_assert(sizeof(Node) == 48, "a descriptor is fetched as three 16-byte pieces");
  __shared__ uint4 s_node[2][3][2 * LV];
  uint2 vN = make_uint2(NONE, 0), vNN = make_uint2(NONE, 0);
  if (P > 0 && lane < 2 * LV) vN = visR[lane];
  if (P > 1 && lane < 2 * LV) vNN = visR[(uint64_t)(2 * LV) + lane];
  // (the lane is in no sub-problem yet: the first point's descriptors, asked for here)
#define NODE_FETCH(id_, buf_) do { const char* src__ = (const char*)(nodes + (id_)); _Pragma("unroll") for (int w__ = 0; w__ < 3; w__++) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src__ + 16 * w__), (__attribute__((address_space(3))) void*)&s_node[(buf_)][w__][0], 16, 0, 0); } while (0)
  if (vN.x != NONE) NODE_FETCH(vN.x, 0);
  for (int pi = 0; pi < P && !bad; pi++) {
    const uint8_t fl = (uint8_t)u_u(flN);
    const uint32_t lf = u_u(lfN);
    const uint2 v = vN;
    vN = vNN;
    // (whenever a lane moves, the descriptor it moves to was asked for at the point before -- or ahead of the loop --: the only source.  With a second one, a load from
    // memory where the buffer does not hold it, the compiler either folds the two into FLAT loads through a generic pointer or waits for ALL vector memory where the two
    // paths meet.  And the move comes FIRST in the point, the buffer read ahead of the write-back: the compiler waits for all vector memory before it reads what a
    // load wrote to LDS, which costs nothing here -- the point before ended with everything waited for -- and a round trip behind anything asked for earlier in the point)
    bool swd = false;
    if (v.x != NONE && v.x != cId) {
      swd = true;
      const uint4 w0 = s_node[pi & 1][0][lane], w1 = s_node[pi & 1][1][lane], w2 = s_node[pi & 1][2][lane];
      // (the descriptor's changing fields live in this lane's copy while the lane stays in the sub-problem; memory gets them when it leaves: a store per query
      // would be waited for by the next point's loads -- vector memory completes in order)
      if (cDirty) { Node* op = nodes + cId; op->last = cn.last; op->sTop = cn.sTop; op->nBlk = cn.nBlk; op->stkOff = cn.stkOff; op->stkCap = cn.stkCap; op->blkOff = cn.blkOff; op->blkCap = cn.blkCap; cDirty = false; }
      cn.dBase = w0.x; cn.nD = w0.y; cn.nE = w0.z; cn.last = (int32_t)w0.w; cn.sTop = w1.x; cn.nBlk = w1.y; cn.stkOff = w1.z; cn.blkOff = w1.w; cn.stkCap = w2.x; cn.blkCap = w2.y;
      cn.eLast = (long long)(((unsigned long long)w2.w << 32) | w2.z);
      cId = v.x; cTopOk = false;
    }
    if (pi + 1 < P) {                                                    // the rows of the point after next, the next point's flags: in flight while this one is processed
      vNN = make_uint2(NONE, 0);
      if (pi + 2 < P && lane < 2 * LV) vNN = visR[(uint64_t)(pi + 2) * (2 * LV) + lane];
      flN = a.hfl[p0 + pi + 1 + vz]; lfN = a.hfr[p0 + pi + 1 + vz];
    }
    if (vN.x != NONE && vN.x != cId) NODE_FETCH(vN.x, (pi + 1) & 1);      // (lanes >= 2 * LV never have a visit: nothing is written beyond a buffer's 36 slots)
    const int ind = fl & 1, inv = (fl >> 1) & 1;
    const float fvP = a.fval[f0 + lf];                                   // the anchor's value so far (asked for now: the point ends with it)
    if (STAT) { const int sw = __popcll(__ballot(swd)); const int nl = __popcll(__ballot(v.x != NONE)); CNT(ind ? 1 : 0, 1); CNT(ind ? 3 : 2, sw > 0); CNT(ind ? 5 : 4, nl); TICK(ind ? 1 : 0); }
    if (ind == 0) {                                                      // PassValueToD1/D2 (SparseDP.h:140-310)
      if (v.x != NONE) {
        const float val = fvP;
        const uint32_t e = cn.dBase + v.y;
        if (ent[e].v < val) { ent[e].v = val; Ap[e] = lf; }
      }
      TICK(2);
    } else {                                                             // start point (:1025-1060)
      // Every pair on a stack but the dummy at position 0 has the boundary n (see sdp_process_wg): a pair is its D index; `Db >= top.second` never holds,
      // candidates meet the stack at Ei[n - 1] only, FindBoundary never searches.
      // phase 0, every lane for its own sub-problem: Eb[i1], stack top, last Block pair
      const Node& nd = cn;                                                 // (a lane without a visit has need == false below: nothing of nd is used)
      int now = -1;
      long long ei1 = 0;
      const int m = (int)nd.nD, n = (int)nd.nE, i1 = (int)v.y;
      int sTop = (int)nd.sTop, nBlk = (int)nd.nBlk;
      uint32_t stkOff = nd.stkOff, blkOff = nd.blkOff;
      // (stack, Block list, Di, Ei[Db] are addressed from their offsets where they are used: four 64-bit pointers per lane are eight registers)
      int sCap = (int)nd.stkCap, bCap = (int)nd.blkCap;
      const long long eLast = nd.eLast;
      int tx = cTop.x; int2 lastB = cLastB;                               // tx == -1: the dummy
      // sx: the D index of the pair BELOW the top (-1: the dummy is below it; SX_UNK: not known) -- what a pop or the flush would have to read the stack for.  A push makes
      // it known (the top it covers); a pop that leaves two pairs or more above the dummy forgets it.
      constexpr int SX_UNK = -2;
      int sx = cTop.y;
      uint32_t st = 0;
      // what the visit asks memory for first, in ONE round: the query's E entry, the stack top and the last Block pair (when the lane has just come to the sub-problem),
      // the first candidate and the top's D entry (used if the query inserts anything)
      Ent pfD; pfD.val = 0; pfD.b = -1; pfD.v = 0;
      long long pfE = 0;
      Ent pfT; pfT.val = 0; pfT.b = 0; pfT.v = 0;
      bool pfTok = false;
      float pfSv = 0.f; long long pfSd = 0; int pfSx = SX_UNK;            // the D entry of the pair below the top (what the first pop compares with)
      if (v.x != NONE) {
        const Ent e = ent[nd.dBase + nd.nD + v.y];
        int2 sT = make_int2(-1, 0), sS = make_int2(-1, 0), bL = make_int2(0, 0);
        if (!cTopOk) { if (sTop > 1) sT = (pairs + stkOff)[sTop - 1]; if (sTop > 2) sS = (pairs + stkOff)[sTop - 2]; if (nBlk > 0) bL = (pairs + blkOff)[nBlk - 1]; }
        if (nd.last + 1 < m) { pfD = (ent + nd.dBase)[nd.last + 1]; pfE = (Ed + nd.dBase)[nd.last + 1]; }
        if (cTopOk && tx >= 0) { pfT = (ent + nd.dBase)[tx]; pfTok = true; }
        if (cTopOk && sx >= 0) { const Ent es = (ent + nd.dBase)[sx]; pfSv = es.v; pfSd = es.val; pfSx = sx; }
        now = e.b; ei1 = e.val;
        if (!cTopOk) { tx = sTop <= 1 ? -1 : sT.x; sx = sTop <= 2 ? -1 : sS.x; lastB = bL; }
      }
      const bool need = now != -1;
      const int pfTx = pfTok ? tx : SX_UNK;                               // the D index pfT was read for
      const int nBlk0 = nBlk; const uint32_t blkOff0 = blkOff;
      TICK(3);
      // phase 1a, every lane for itself: short insertion runs (most queries advance `now` by a few candidates only) -- the same loop
      // as below, literal and lane-local, all lanes at once
      const int LOCAL_MAX = 4;                                             // (6: 635 ms over 12 launches, 4: 617, 2 / 3: 635, 1: 652, 10: 634, 16: 635)
      const bool small = need && now > nd.last && now - nd.last <= LOCAL_MAX;
      int nIt = 0, nPop = 0;
      if (small) {
        bool topD = false; float tDv = 0; long long tDi = 0;
#define SPUSHL(val_) do { const int2 v__ = (val_); if (sTop >= sCap) { if (!grow_pairs(pairs, stkOff, sCap, sTop, poolUsed, poolPair, poolPairs)) st |= LRA_ST_CAPACITY; } \
                          if (sTop < sCap) (pairs + stkOff)[sTop] = v__; sTop++; } while (0)
#define BPUSHL(val_) do { const int2 v__ = (val_); if (nBlk >= bCap) { if (!grow_pairs(pairs, blkOff, bCap, nBlk, poolUsed, poolPair, poolPairs)) st |= LRA_ST_CAPACITY; } \
                          if (nBlk < bCap) (pairs + blkOff)[nBlk] = v__; nBlk++; lastB = v__; } while (0)
        for (int i = nd.last + 1; i <= now && !st; ++i) {
          if (STAT) nIt++;
          Ent di_ = pfD; long long edb = pfE;
          if (i != nd.last + 1) { di_ = (ent + nd.dBase)[i]; edb = (Ed + nd.dBase)[i]; }
          const int db = di_.b;
          if (db == -1) break;
          const long long di = di_.val; const float dvi = di_.v;
          if (tx == -1) { BPUSHL(make_int2(-1, db)); SPUSHL(make_int2(i, n)); tx = i; sx = -1; tDv = dvi; tDi = di; topD = true; }
          if (!topD) { Ent e = pfT; if (!pfTok) e = (ent + nd.dBase)[tx]; tDv = e.v; tDi = e.val; topD = true; }
          if (BEATS(dvi, di, tDv, tDi, edb)) {
            if (nBlk > 0 && db > lastB.y) BPUSHL(make_int2(tx, db));
            const float sNew = dvi + W(di, eLast);
            int cx = tx; float cDv = tDv; long long cDi = tDi;
            while (sTop > 0) {
              if (cx < 0 || n < 1) { st |= LRA_ST_OOB_SLOT; break; }
              if (!(sNew > cDv + W(cDi, eLast))) break;
              sTop--;
              if (STAT) nPop++;
              if (sTop == 0) { st |= LRA_ST_OOB_SLOT; break; }
              cx = sTop - 1 == 0 ? -1 : sx != SX_UNK ? sx : (pairs + stkOff)[sTop - 1].x;
              sx = sTop - 1 <= 1 ? -1 : SX_UNK;
              if (cx == -1) break;
              if (cx == pfSx) { cDv = pfSv; cDi = pfSd; }
              else { const Ent ce = (ent + nd.dBase)[cx]; cDv = ce.v; cDi = ce.val; }
            }
            if (st) break;
            SPUSHL(make_int2(i, n)); sx = cx; tx = i; tDv = dvi; tDi = di; topD = true;
          }
        }
#undef SPUSHL
#undef BPUSHL
      }
      if (STAT) { const int mi = WMAX(nIt), mp = WMAX(nPop); CNT(6, mi); CNT(7, mp); CNT(8, mi > 0); TICK(4); }
      // phase 1b, one owner at a time, the whole wave: long insertion runs  for (i = last + 1; i <= now; ++i)  of Maximization :275-328
      unsigned long long todo = __ballot(need && now > nd.last && !small);
      if (STAT) CNT(9, __popcll(todo));
      while (todo) {
        const int owner = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const Ent* oD = ent + (uint32_t)rl_i((int)nd.dBase, owner);
        const long long* oEd = Ed + (uint32_t)rl_i((int)nd.dBase, owner);
        const int on = rl_i(n, owner);
        const long long oeLast = rl_ll(eLast, owner);
        const int olast = rl_i(nd.last, owner), onow = rl_i(now, owner);
        int oTop = rl_i(sTop, owner), oBlk = rl_i(nBlk, owner);
        uint32_t oStkOff = (uint32_t)rl_i((int)stkOff, owner), oBlkOff = (uint32_t)rl_i((int)blkOff, owner);
        int2* oS = pairs + oStkOff; int2* oB = pairs + oBlkOff;
        int oSCap = rl_i(sCap, owner), oBCap = rl_i(bCap, owner);
        int otx = rl_i(tx, owner), osx = rl_i(sx, owner); int2 olastB = make_int2(rl_i(lastB.x, owner), rl_i(lastB.y, owner));
        uint32_t ost = 0;
        bool topD = rl_i(pfTok ? 1 : 0, owner) != 0; float tDv = rl_f(pfT.v, owner); long long tDi = rl_ll(pfT.val, owner);   // (the owner's top, asked for above)
#define SPUSH(val_) do { const int2 v__ = (val_); if (oTop >= oSCap) { if (coop_grow_pairs(pairs, oStkOff, oSCap, oTop, poolUsed, poolPair, poolPairs, lane)) oS = pairs + oStkOff; else ost |= LRA_ST_CAPACITY; } \
                         if (oTop < oSCap) oS[oTop] = v__; oTop++; } while (0)
#define BPUSH(val_) do { const int2 v__ = (val_); if (oBlk >= oBCap) { if (coop_grow_pairs(pairs, oBlkOff, oBCap, oBlk, poolUsed, poolPair, poolPairs, lane)) oB = pairs + oBlkOff; else ost |= LRA_ST_CAPACITY; } \
                         if (oBlk < oBCap) oB[oBlk] = v__; oBlk++; olastB = v__; } while (0)
        bool stop = false;
        for (int i0 = olast + 1; i0 <= onow && !stop && !ost; i0 += 64) {
          const int j = i0 + lane;
          Ent dj; dj.val = 0; dj.b = -1; dj.v = 0;
          long long ej = 0;
          if (j <= onow) { dj = oD[j]; ej = oEd[j]; }
          const int nb = min(64, onow - i0 + 1);
          int t = 0;
          while (t < nb && !ost) {
            // iterations that neither stop nor beat the top candidate change nothing: every lane tests its own candidate
            // against the current top and the wave jumps to the first one that does something
            if (otx != -1) {
              if (!topD) { const Ent e = oD[otx]; tDv = e.v; tDi = e.val; topD = true; }
              bool evt = false;
              if (lane >= t && lane < nb) evt = dj.b == -1 || BEATS(dj.v, dj.val, tDv, tDi, ej);
              const unsigned long long em = __ballot(evt);
              if (!em) break;
              t = __ffsll((long long)em) - 1;
            }
            const int i = i0 + t;
            const int db = rl_i(dj.b, t);
            if (db == -1) { stop = true; break; }                         // :277
            const long long di = rl_ll(dj.val, t), edb = rl_ll(ej, t);
            const float dvi = rl_f(dj.v, t);
            bool win = true;                                              // (the ballot's test is the reference's, :405, unless the top was the dummy)
            if (otx == -1) { BPUSH(make_int2(-1, db)); SPUSH(make_int2(i, on)); otx = i; osx = -1; tDv = dvi; tDi = di; topD = true; win = BEATS(dvi, di, tDv, tDi, edb); }   // :389-395
            if (win) {
              if (oBlk > 0 && db > olastB.y) BPUSH(make_int2(otx, db));
              const float sNew = dvi + W(di, oeLast);
              int cx = otx; float cDv = tDv; long long cDi = tDi;
              while (oTop > 0) {                                          // :415-422
                if (cx < 0 || on < 1) { ost |= LRA_ST_OOB_SLOT; break; }
                if (!(sNew > cDv + W(cDi, oeLast))) break;
                oTop--;
                if (oTop == 0) { ost |= LRA_ST_OOB_SLOT; break; }
                cx = oTop - 1 == 0 ? -1 : osx != SX_UNK ? osx : oS[oTop - 1].x;
                osx = oTop - 1 <= 1 ? -1 : SX_UNK;
                if (cx == -1) break;
                const Ent ce = oD[cx]; cDv = ce.v; cDi = ce.val;
              }
              if (ost) break;
              SPUSH(make_int2(i, on)); osx = cx; otx = i; tDv = dvi; tDi = di; topD = true;
            }
            t++;
          }
        }
#undef SPUSH
#undef BPUSH
        if (lane == owner) { sTop = oTop; nBlk = oBlk; tx = otx; sx = osx; lastB = olastB; st |= ost; stkOff = oStkOff; blkOff = oBlkOff; sCap = oSCap; bCap = oBCap; }
      }
      TICK(5);
      // phase 2, every lane for its own sub-problem: the flush of Maximization :438-453 (only its `now == m - 1` branch ever pops), FindValueInBlock :322-333, Ev / Ep
      float ev = -1.f;
      bool got = false;
      int nFl = 0, nSr = 0, nLd = 0, nBs = 0, nCh = 0;
      if (need && !st) {
#define BPUSH2(val_) do { const int2 v__ = (val_); if (nBlk >= bCap) { if (!grow_pairs(pairs, blkOff, bCap, nBlk, poolUsed, poolPair, poolPairs)) st |= LRA_ST_CAPACITY; } \
                          if (nBlk < bCap) (pairs + blkOff)[nBlk] = v__; nBlk++; lastB = v__; } while (0)
        if (now == m - 1) {
          while (sTop > 1 && tx != -1 && !st) {
            if (STAT) nFl++;
            BPUSH2(make_int2(tx, n)); sTop--;
            tx = sTop - 1 == 0 ? -1 : sx != SX_UNK ? sx : (pairs + stkOff)[sTop - 1].x;
            sx = sTop - 1 <= 1 ? -1 : SX_UNK;
          }
        }
#undef BPUSH2
        int i2 = -1;
        if (!st && nBlk > 0) {
          if (i1 >= lastB.y) i2 = tx;                                     // (i1 < top.second always)
          else {
            if (STAT) { nBs = nBlk; nCh = (nBlk != nBlk0 || blkOff != blkOff0) ? 1 : 0; }
            int lo = 0, cnt = nBlk, bx = -1;                              // UPPERbound :205-221, two levels per memory round; the search ends at the position of its most
            while (cnt > 0) {                                             // recent false probe (or at the end): Block[lo].first is that probe's pair, no further load
              if (STAT) nSr++;
              const int step = cnt >> 1, it = lo + step;
              const int cntT = cnt - step - 1, itT = it + 1 + (cntT >> 1), itF = lo + (step >> 1);
              const int2 pM = (pairs + blkOff)[it], pT = cntT > 0 ? (pairs + blkOff)[itT] : make_int2(0, 0), pF = step > 0 ? (pairs + blkOff)[itF] : make_int2(0, 0);
              if (i1 >= pM.y) {
                lo = it + 1; cnt = cntT;
                if (cnt > 0) { const int s2 = cnt >> 1; if (i1 >= pT.y) { lo = itT + 1; cnt -= s2 + 1; } else { cnt = s2; bx = pT.x; } }
              } else {
                cnt = step; bx = pM.x;
                if (cnt > 0) { const int s2 = cnt >> 1; if (i1 >= pF.y) { lo = itF + 1; cnt -= s2 + 1; } else { cnt = s2; bx = pF.x; } }
              }
            }
            if (lo < nBlk) i2 = bx;
          }
        }
        if (st || i2 < 0 || i2 >= m) st |= st ? st : LRA_ST_OOB_SLOT;
        else {
          // (the answer is the stack top more often than not, and when nothing was pushed in this visit its D entry came with the visit's first loads)
          Ent d2;
          if (i2 == pfTx) d2 = pfT;
          else if (i2 == pfSx) { d2.v = pfSv; d2.val = pfSd; d2.b = 0; }
          else { d2 = (ent + nd.dBase)[i2]; if (STAT) nLd = 1; }
          ev = d2.v + W(d2.val, ei1) + rate * a.flen[f0 + lf];            // :1040
          got = true;
          Ap[nd.dBase + nd.nD + i1] = (uint32_t)i2;                       // Ep[i1] (Ev[i1] is never read again)
          cDirty = true;
          cn.last = now; cn.sTop = (uint32_t)sTop; cn.nBlk = (uint32_t)nBlk; cn.stkOff = stkOff; cn.blkOff = blkOff; cn.stkCap = (uint32_t)sCap; cn.blkCap = (uint32_t)bCap;
          cTop = make_int2(tx, sx); cLastB = lastB; cTopOk = true;
        }
      }
      if (STAT) { const int mf = WMAX(nFl), ms = WMAX(nSr), ml = WMAX(nLd), mb = WMAX(nBs), mc = WMAX(nCh), m2 = WMAX(nIt >= 2 ? 1 : 0); CNT(10, mf); CNT(11, ms); CNT(12, ml); CNT(13, ms > 0); CNT(14, mc); CNT(15, mb); CNT(16, m2); TICK(6); }
      const uint32_t myI1 = v.y;
      if (__ballot(st != 0)) { for (int o = 32; o > 0; o >>= 1) st |= __shfl_xor(st, o); }   // (a status is rare: no exchange unless a lane has one)
      bad |= st;
      // Value[ii]: visits apply in the order R family deepest level first, then C family; `val < Ev` keeps the first maximum.  The maximum over the lanes by DPP row
      // shifts / broadcasts (six VALU operations; a butterfly of __shfl_xor is twelve dependent trips through the LDS crossbar, a tenth of a point's time), then the
      // first lane in visit order among those that hold it: within a family a higher lane is a deeper level, and the R family's lanes come first
      if (!bad) {
        const float bvM = wave_max_f32(got ? ev : -__builtin_inff());
        const unsigned long long eq = __ballot(got && ev == bvM);
        const unsigned long long eqR = eq & ((1ull << LV) - 1);
        const int win = eq ? 63 - __clzll((long long)(eqR ? eqR : eq)) : -1;
        if (lane == win) {
          if (fvP < ev) {
            a.fval[f0 + lf] = ev; a.fprevNode[f0 + lf] = v.x; a.fprevInd[f0 + lf] = myI1;
            a.fflags[f0 + lf] = (uint8_t)((fam2 == 0 ? 1 : 0) | (inv ? 2 : 0));   // bit0 prev (row family), bit1 inv
          }
        }
      }
      TICK(7);
    }
    wave_sync();
    TICK(ind ? 9 : 8);
  }
  if (STAT && lane == 0 && a.stat) { for (int k = 0; k < 10; k++) atomicAdd(a.stat + k, sT[k]); for (int k = 0; k < 20; k++) atomicAdd(a.stat + 10 + k, sC[k]); }
  if (cDirty && cId != NONE) { Node* op = nodes + cId; op->last = cn.last; op->sTop = cn.sTop; op->nBlk = cn.nBlk; op->stkOff = cn.stkOff; op->stkCap = cn.stkCap; op->blkOff = cn.blkOff; op->blkCap = cn.blkCap; }
  if (lane == 0 && bad) atomicOr(&a.status[r], bad);
#undef W
#undef BEATS
#undef NODE_FETCH
#undef TICK
#undef CNT
#undef WMAX
}

// ---- the same for LARGE reads: one 1024-thread workgroup per read, the (family pair, level) slots spread over its 16 waves.
// A read from a satellite array gives tens of thousands of anchors on a lattice of tied rows / columns / diagonals; with one wave per read the
// owners of a start point's insertions take turns (phase 1b above) and every turn is a chain of dependent memory round trips: 44 k points took
// 1.2 s, the whole launch waiting for that one wave.  Here wave w owns the slots w and w + 16: every sub-problem still sees exactly the deposits
// and queries it sees above, in the same order (a sub-problem belongs to one slot, a slot to one wave).  The waves do NOT meet at the points:
// each runs through all points for its own slots.  What couples them is Value[] only -- a start point's candidates from all slots are reduced
// to (max value, first in visit order), and an end point deposits its anchor's value.  So a start point's wave folds its slots' candidates
// into one 64-bit word per (anchor, start point) with atomicMax (value bits high, ~visit rank low: the maximum IS the reference's choice) and
// counts itself in; an end point's wave waits until all 16 waves are counted in for the start points of that anchor that precede it (always
// earlier in every wave's sequence, so the wave that is furthest behind never waits), then takes the value.  fval / prev are written once at
// the end.  The critical path is the busiest wave's own work instead of (slowest wave + two barriers + a serial reduction) per point.
// What a visit costs is its chain of dependent memory round trips (a 47 k-point read from a satellite array: 19 per query of a top-level sub-problem, ~0.65 us
// each, 12 us per point).  So the slot keeps, in LDS, what the next visit will ask memory for: beside the stack top its Di / Ei[y - 1] / Dv (Dv dropped when a
// deposit lands on that entry), the entry below the top with its Di / Ei[y - 1] (position 0 is always the dummy pair), and the sub-problem's Ei[nE - 1] (a push
// nearly always owns the whole tail).  A pop then costs one load (the new top's Dv) instead of three dependent ones, a candidate that beats the top without
// popping costs none, the candidate scan is one round (Ei[Db] stored beside the entries by sdp_build), and the query's E entry is in flight one point ahead.
struct SlotState {
  Node cn; uint32_t cId; int dirty;                // dirty: cn's changing fields are newer than the descriptor in memory (written back when the slot leaves the sub-problem)
  int2 cTop, cLastB; int cTopOk;                  // stack top, last Block pair (valid when cTopOk)
  int topInfoOk, topDvOk; float topDv, topWe; long long topDi;   // of cTop: Di[x], w(Di[x], Ei[n - 1]); Dv[x] while no deposit has touched it
  int2 sec; int secOk, secDvOk; float secDv, secWe; long long secDi;    // the pair below the top, with the same
};
constexpr int WG_NW = 16;
constexpr int RING_W = 2048, RING_LEAD = 1024, RING_CHECK = 32;   // window mode of sdp_process_wg

// Which slots a wave owns.  The cost of a slot falls with its level (measured on a 47 k-point read: R0-R3 and C0-C2 ~ 350-400 M cycles each, level 8 ~ 100 M,
// level 13+ ~ 0), so wave w takes row-family level w and column-family level 15 - w (plus the two levels beyond 15): the busiest wave carries ~ 460 M cycles
// instead of ~ 760 M with slots w, w + 16, w + 32.
__device__ __forceinline__ int wg_slot(int wave, int k) {
  static_assert(LV == 18 && WG_NW == 16, "slot table written for 18 levels on 16 waves");
  if (k == 0) return wave;                                   // R level wave
  if (k == 1) return LV + (15 - wave);                       // C level 15 - wave
  return wave == 0 ? 16 : wave == 1 ? 17 : wave == 15 ? LV + 16 : wave == 14 ? LV + 17 : 2 * LV;   // R16, R17, C16, C17; else none
}

// SPW: slots per wave -- 3 in general (36 slots on 16 waves); 2 when the launch's reads have at most 2^15 distinct rows and columns (levels 16 and 17 are empty then:
// every array per slot is a third smaller, which is what the register file is short of)
// DBG (LRA_SDP_DBG): cycle counters per slot and section; a separate instantiation, because the counters' registers are what the production kernel is short of
template <int SPW, bool DBG>
__global__ void __launch_bounds__(64 * WG_NW) sdp_process_wg(ProcArgs a) {
  __shared__ float s_slope[25], s_inter[25];
  __shared__ SlotState ss[2 * LV];
  __shared__ short s_pen[PEN_TAB_WG];
  // the window of anchors in progress (see below): per start point of an anchor the waves' best candidate and how many waves are counted in
  __shared__ unsigned long long r_best[2 * RING_W];
  __shared__ uint32_t r_cnt[2 * RING_W];
  __shared__ int s_pos[WG_NW]; __shared__ uint32_t s_span, s_wtot[WG_NW];
  __shared__ uint32_t s_bad;       // (read with BAD(): a volatile read is a FLAT load that waits for every outstanding store of the wave, at every point)
#define BAD() __hip_atomic_load(&s_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
  const int tid = threadIdx.x, lane = tid & 63, wave = u_i(tid >> 6);
  if (tid < 25) { s_slope[tid] = a.pwl.slope[tid]; s_inter[tid] = a.pwl.inter[tid]; }
  const int penN = min(a.penN, PEN_TAB_WG);
  for (int x = tid; x < penN; x += 64 * WG_NW) s_pen[x] = a.penTab[x];
  if (tid < 2 * LV) {
    SlotState z; memset(&z, 0, sizeof z); z.cn.last = -1; z.cId = NONE;
    ss[tid] = z;
  }
  if (tid == 0) { s_bad = 0; s_span = 0; }
  if (tid < WG_NW) s_pos[tid] = 0;
  for (int x = tid; x < 2 * RING_W; x += 64 * WG_NW) { r_best[x] = 0; r_cnt[x] = 0; }
  __syncthreads();
  const int c1 = a.pwl.c1, c2 = a.pwl.c2;
#define W(i, j) pwl_w_tab(s_pen, penN, s_slope, s_inter, c1, c2, (i), (j))
#define BEATS(a_, x_, b_, y_, e_) pwl_beats(s_pen, penN, s_slope, s_inter, c1, c2, (a_), (x_), (b_), (y_), (e_))
  const int rr = (int)a.order[blockIdx.x], r = a.r0 + rr;
  if (a.status[r] & LRA_ST_CAPACITY) return;                             // (given up by the emit pass, see sdp_process)
  const uint64_t p0 = a.ptOff[r], f0 = a.fragOff[r];
  const int P = (int)(a.ptOff[r + 1] - p0);
  const float rate = a.rate_in ? a.rate_in[r] : a.rate;
  const ReadArena A = a.ra[rr];
  char* ab = arena_ptr(A.base);
  Node* nodes = (Node*)ab;
  Ent* ent = (Ent*)(ab + A.entOff);
  uint32_t* Ap = (uint32_t*)(ab + A.apOff);
  const long long* Ed = (const long long*)(ab + A.edOff);
  int2* pairs = (int2*)(ab + A.stkOff);
  const uint32_t poolPair = A.poolPair, poolPairs = A.poolPairs;
  uint32_t* poolUsed = a.poolUsed + rr;
  const uint2* visR = (const uint2*)(ab + A.visOff);
  // ---- per anchor: best[2] (one word per start point: value bits << 32 | ~visit rank; 0 = no candidate), cnt[2] (waves counted in), nS, sPos[2];
  // per point: pm = how many start points of its anchor precede it (2 bits), and in window mode the anchor's ordinal and whether it has one start point only
  //
  // WINDOW MODE.  An anchor's words are in use from its first start point to its last end point, `span` points at most; a wave cannot pass an end point before all
  // waves are through its anchor's start points, so the waves stay within a few spans of each other wherever it matters and only the anchors of a window of points are
  // in progress at any time.  Their words then live in LDS -- entry (ordinal of the anchor among first start points) mod RING_W -- instead of at L2: counting in and
  // asking whether all are counted in cost an LDS access instead of dependent L2 round trips.  Entries are never cleared: the count of an entry grows by 16 per
  // generation (ordinal / RING_W; an anchor with one start point counts for both), and a candidate carries its generation above its value, so the maximum is the
  // current generation's.  Anchor o + RING_W must not be counted in while anchor o is in progress: a wave that runs ahead where it has no end points of its own
  // to stop at is held RING_LEAD points in front of the slowest (checked every RING_CHECK points); first start points are distinct points, so o + RING_W starts
  // RING_W points after o at least, and RING_W >= RING_LEAD + RING_CHECK + span + 1 keeps them apart.  Reads with longer spans use the words at L2.
  const int F = (int)(a.fragOff[r + 1] - f0);
  char* wsb = a.wgScratch + a.wgOff[blockIdx.x];
  unsigned long long* best = (unsigned long long*)wsb;
  uint32_t* cnt = (uint32_t*)(wsb + 16 * (size_t)F);
  uint32_t* nS = cnt + 2 * (size_t)F;
  uint32_t* sPos = nS + F;
  uint32_t* pm = sPos + 2 * (size_t)F;
  for (int f = tid; f < F; f += 64 * WG_NW) { best[2 * f] = 0; best[2 * f + 1] = 0; cnt[2 * f] = 0; cnt[2 * f + 1] = 0; nS[f] = 0; sPos[2 * f] = 0; sPos[2 * f + 1] = 0; }
  __syncthreads();
  for (int pi = tid; pi < P; pi += 64 * WG_NW) {
    const uint32_t lf = a.hfr[p0 + pi];
    if (a.hfl[p0 + pi] & 1) { const uint32_t k = atomicAdd(&nS[lf], 1u); if (k < 2) sPos[2 * lf + k] = (uint32_t)pi; else atomicOr(&s_bad, (uint32_t)LRA_ST_RANGE); }
    else atomicMax(&cnt[2 * lf], (uint32_t)pi);                            // (for now: the anchor's last end point)
  }
  __syncthreads();
  for (int f = tid; f < F; f += 64 * WG_NW) {
    if (nS[f] == 2 && sPos[2 * f] > sPos[2 * f + 1]) { const uint32_t t = sPos[2 * f]; sPos[2 * f] = sPos[2 * f + 1]; sPos[2 * f + 1] = t; }
    if (nS[f] > 0 && cnt[2 * f] > sPos[2 * f]) atomicMax(&s_span, cnt[2 * f] - sPos[2 * f]);
    cnt[2 * f] = 0;
  }
  __syncthreads();
  const bool ring = a.wgNoRing == 0 && s_span + RING_LEAD + RING_CHECK + 1 <= (uint32_t)RING_W;
  if (ring) {                                                              // ordinals of the anchors, in the order of their first start points -> cnt[2 f]
    const int per = (P + 64 * WG_NW - 1) / (64 * WG_NW), b0 = min(P, tid * per), b1 = min(P, b0 + per);
    uint32_t mine = 0;
    for (int pi = b0; pi < b1; pi++) mine += (a.hfl[p0 + pi] & 1) && sPos[2 * a.hfr[p0 + pi]] == (uint32_t)pi;
    uint32_t inc = mine;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) s_wtot[wave] = inc;
    __syncthreads();
    uint32_t at = inc - mine;
    for (int w = 0; w < wave; w++) at += s_wtot[w];
    for (int pi = b0; pi < b1; pi++) { const uint32_t lf = a.hfr[p0 + pi]; if ((a.hfl[p0 + pi] & 1) && sPos[2 * lf] == (uint32_t)pi) cnt[2 * lf] = at++; }
    __syncthreads();
  }
  for (int pi = tid; pi < P; pi += 64 * WG_NW) {
    const uint32_t lf = a.hfr[p0 + pi];
    uint32_t k = 0;
    for (uint32_t x = 0; x < min(nS[lf], 2u); x++) k += sPos[2 * lf + x] < (uint32_t)pi;
    pm[pi] = ring ? (cnt[2 * lf] << 3) | (nS[lf] == 1 ? 4u : 0u) | k : k;
  }
  __threadfence();
  __syncthreads();
  // LRA_SDP_DBG: cycles per wave spent in each of its slots and waiting at end points (16 words per wave behind pm[], 8-aligned)
  unsigned long long* dbgT = (unsigned long long*)(((uintptr_t)(pm + P) + 7) & ~(uintptr_t)7);
  unsigned long long tRounds = 0, tEvents = 0, tStore = 0, tEvA = 0, tEvB = 0, tEvC = 0, tSwitch = 0, tDep = 0, tPub = 0;   // event loop: choosing the candidate, up to the comparison with the top, the winner's path
  unsigned long long tSlot[4] = {0, 0, 0, 0}, tSec[4] = {0, 0, 0, 0};   // tSec (queries only): set-up, Maximization, flush + Block search, result + state
  static_assert(SPW == 2 || SPW == (2 * LV + WG_NW - 1) / WG_NW, "slots per wave");
  // The rows of the points: this point's are scalars, the next point's too (so that ITS sub-problem descriptors and its anchor's value can be asked for now), and
  // the rows of the point after next are in flight in vector registers.  Memory returns in order: what was asked for at the top of the previous point is there
  // by the time anything of this point has been waited for, so a point starts without a round trip of its own.
  uint2 vN[SPW]; uint32_t flN = P > 0 ? u_u(a.hfl[p0]) : 0, lfN = P > 0 ? u_u(a.hfr[p0]) : 0, rkN = P > 0 ? u_u(pm[0]) : 0;
#pragma unroll
  for (int k = 0; k < SPW; k++) { const int slot = wg_slot(wave, k); vN[k] = (P > 0 && slot < 2 * LV) ? u_u2(visR[slot]) : make_uint2(NONE, 0); }
  uint2 vV[SPW]; uint32_t lfV = 0, rkV = 0; uint8_t flV = 0;   // (a byte stays a byte until it is used: widening one waits for its load)
  //                    // raw (per-lane copies of) the rows of point pi + 1 at the top of point pi
#pragma unroll
  for (int k = 0; k < SPW; k++) vV[k] = make_uint2(NONE, 0);
  if (P > 1) {
    flV = a.hfl[p0 + 1]; lfV = a.hfr[p0 + 1]; rkV = pm[1];
#pragma unroll
    for (int k = 0; k < SPW; k++) { const int slot = wg_slot(wave, k); if (slot < 2 * LV) vV[k] = visR[(uint64_t)(2 * LV) + slot]; }
  }
  uint32_t ndV[SPW], pfId[SPW]; float fvV = 0.f;                          // asked for one point ahead: lane l < 12 holds word l of the descriptor pfId[k]; the anchor's value
#pragma unroll
  for (int k = 0; k < SPW; k++) { ndV[k] = 0; pfId[k] = NONE; }
  if (P > 0) fvV = a.fval[f0 + lfN];
  constexpr int NODE_WORDS = (int)(sizeof(Node) / 4);
  static_assert(sizeof(Node) % 4 == 0 && NODE_WORDS <= 64 && offsetof(SlotState, cn) == 0, "a descriptor is moved a word per lane");
  const unsigned long long tAll0 = DBG ? clock64() : 0;
  for (int pi = 0; pi < P && !BAD(); pi++) {
    const uint32_t fl = flN, lf = lfN, rk = rkN & 3u;
    const uint32_t re = 2 * ((rkN >> 3) & (uint32_t)(RING_W - 1)), gen = (rkN >> 3) / (uint32_t)RING_W + 1, single = (rkN >> 2) & 1u;   // window mode: the anchor's entry
    const float fvC = fvV;
    const unsigned long long tp0 = DBG ? clock64() : 0;
    if (ring && (pi & (RING_CHECK - 1)) == 0 && lane == 0) {             // not further than RING_LEAD points in front of the slowest wave
      __hip_atomic_store(&s_pos[wave], pi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      for (;;) {
        int mn = pi;
        for (int w = 0; w < WG_NW; w++) mn = min(mn, __hip_atomic_load(&s_pos[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (pi - mn <= RING_LEAD || BAD()) break;
        __builtin_amdgcn_s_sleep(8);
      }
    }
    uint2 vv[SPW]; Ent e0[SPW];
#pragma unroll
    for (int k = 0; k < SPW; k++) vv[k] = vN[k];
    const int ind = fl & 1;
    // sub-problem descriptors of this point's visits
    bool act[SPW];
#pragma unroll
    for (int k = 0; k < SPW; k++) {
      const int slot = wg_slot(wave, k);
      act[k] = slot < 2 * LV && vv[k].x != NONE;
      if (act[k] && vv[k].x != u_u(ss[slot].cId)) {
        SlotState& Zs = ss[slot];
        if (lane == 0 && Zs.dirty) { Node* op = nodes + Zs.cId; op->last = Zs.cn.last; op->sTop = Zs.cn.sTop; op->nBlk = Zs.cn.nBlk; op->stkOff = Zs.cn.stkOff; op->stkCap = Zs.cn.stkCap; op->blkOff = Zs.cn.blkOff; op->blkCap = Zs.cn.blkCap; Zs.dirty = 0; }
        uint32_t w = ndV[k];
        if (pfId[k] != vv[k].x && lane < NODE_WORDS) w = ((const uint32_t*)(nodes + vv[k].x))[lane];   // (the first point; otherwise asked for at the previous one)
        if (lane < NODE_WORDS) ((uint32_t*)&Zs)[lane] = w;
        if (lane == 0) { Zs.cId = vv[k].x; Zs.cTopOk = 0; Zs.topInfoOk = 0; Zs.topDvOk = 0; Zs.secOk = 0; Zs.secDvOk = 0; }
      }
    }
    wave_sync();
    if (pi + 1 < P) {                                                    // the next point's rows arrive as scalars; the rows of the one after are asked for
      flN = u_u(flV); lfN = u_u(lfV); rkN = u_u(rkV);
#pragma unroll
      for (int k = 0; k < SPW; k++) vN[k] = u_u2(vV[k]);
      if (pi + 2 < P) {
        flV = a.hfl[p0 + pi + 2]; lfV = a.hfr[p0 + pi + 2]; rkV = pm[pi + 2];
#pragma unroll
        for (int k = 0; k < SPW; k++) { const int slot = wg_slot(wave, k); if (slot < 2 * LV) vV[k] = visR[(uint64_t)(pi + 2) * (2 * LV) + slot]; }
      }
      // ... and what the next point will start with: the descriptors of the sub-problems its slots move to (never the ones the slots are in now: those are newer
      // here than in memory; one a slot has left was written back above or earlier, ahead of this load), and its anchor's value if it is an end point
#pragma unroll
      for (int k = 0; k < SPW; k++) {
        const int slot = wg_slot(wave, k);
        pfId[k] = NONE;
        if (slot < 2 * LV && vN[k].x != NONE && vN[k].x != u_u(ss[slot].cId)) {
          pfId[k] = vN[k].x;
          if (lane < NODE_WORDS) ndV[k] = ((const uint32_t*)(nodes + vN[k].x))[lane];
        }
      }
      if (!(flN & 1)) fvV = a.fval[f0 + lfN];
    }
    if (DBG) { __builtin_amdgcn_s_waitcnt(0); tSwitch += clock64() - tp0; }
    float depVal = 0.f;
    if (!ind) {                                                          // an end point: its anchor's value, once every wave has been through its start points
      bool any = false;
#pragma unroll
      for (int k = 0; k < SPW; k++) any |= act[k];
      if (!any) continue;
      const unsigned long long tw0 = DBG ? clock64() : 0;
      if (lane == 0) {
        const int need = (int)rk;
        depVal = fvC;
        for (int x = 0; x < need; x++) {
          if (ring) {                                                     // (LDS serves a wave's accesses in order: the count, then the candidate)
            if (DBG) { tStore += 1ull << 32; if (__hip_atomic_load(&r_cnt[re + x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (uint32_t)WG_NW * gen) tStore++; }   // polls | not ready at the first
            while (__hip_atomic_load(&r_cnt[re + x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (uint32_t)WG_NW * gen && !BAD()) __builtin_amdgcn_s_sleep(2);
            const unsigned long long key = __hip_atomic_load(&r_best[re + x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const float v = __uint_as_float((uint32_t)(key >> 16));
            if ((uint32_t)(key >> 48) == gen && depVal < v) depVal = v;
          } else {
            if (DBG) { tStore += 1ull << 32; if (__hip_atomic_load(&cnt[2 * lf + x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)WG_NW) tStore++; }
            // (relaxed loads served by L2: an acquire would invalidate the CU's vector cache under all 16 waves at every end point)
            while (__hip_atomic_load(&cnt[2 * lf + x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)WG_NW && !BAD()) __builtin_amdgcn_s_sleep(2);
            const unsigned long long key = __hip_atomic_load(&best[2 * lf + x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float v = __uint_as_float((uint32_t)(key >> 32));
            if (key && depVal < v) depVal = v;
          }
        }
      }
      if (DBG) tSlot[3] += clock64() - tw0;
    }
    if (ind) {
#pragma unroll
      for (int k = 0; k < SPW; k++) {
        e0[k].b = -1; e0[k].val = 0; e0[k].v = 0;
        if (act[k]) { const Node& nd = ss[wg_slot(wave, k)].cn; e0[k] = ent[u_u(nd.dBase) + u_u(nd.nD) + vv[k].y]; }   // (the slots' loads are independent: one round)
      }
    }
    float wBest = -2.f; int wRank = 0;                                   // this wave's best candidate of the point and its visit rank
    // (one copy of the visit's code for all of the wave's slots: the kernel is several times the instruction cache as it is)
#pragma unroll 1
    for (int k = 0; k < SPW; k++) {
      const int slot = wg_slot(wave, k);
      if (slot >= 2 * LV) continue;
      const uint2 v = k == 0 ? vv[0] : k == 1 ? vv[1] : vv[SPW - 1];
      const Ent e0k = k == 0 ? e0[0] : k == 1 ? e0[1] : e0[SPW - 1];
      if (!(k == 0 ? act[0] : k == 1 ? act[1] : act[SPW - 1])) continue;
      const unsigned long long ts0 = DBG ? clock64() : 0;
      SlotState& Z = ss[slot];
      Node nd;
      { const Node& zn = Z.cn; nd.dBase = u_u(zn.dBase); nd.nD = u_u(zn.nD); nd.nE = u_u(zn.nE); nd.last = u_i(zn.last); nd.sTop = u_u(zn.sTop); nd.nBlk = u_u(zn.nBlk); nd.stkOff = u_u(zn.stkOff);
        nd.blkOff = u_u(zn.blkOff); nd.stkCap = u_u(zn.stkCap); nd.blkCap = u_u(zn.blkCap); nd.eLast = u_ll(zn.eLast); }
      if (ind == 0) {                                                    // PassValueToD1/D2
        if (lane == 0) {
          const float val = depVal;
          const uint32_t e = nd.dBase + v.y;
          if (ent[e].v < val) {
            ent[e].v = val; Ap[e] = lf;
            if (Z.cTopOk && Z.cTop.x == (int)v.y) Z.topDvOk = 0;         // the cached Dv of the stack top (of the pair below it) is stale now
            if (Z.cTopOk && Z.secOk && Z.sec.x == (int)v.y) Z.secDvOk = 0;
          }
        }
        if (DBG) { __builtin_amdgcn_s_waitcnt(0); tDep += clock64() - ts0; }
        continue;
      }
      const int now = u_i(e0k.b);
      const long long ei1 = u_ll(e0k.val);
      const bool need = now != -1;
      const int m = (int)nd.nD, n = (int)nd.nE, i1 = (int)v.y;
      int oTop = (int)nd.sTop, oBlk = (int)nd.nBlk;
      uint32_t oStkOff = nd.stkOff, oBlkOff = nd.blkOff;
      int2* oS = pairs + oStkOff; int2* oB = pairs + oBlkOff;
      int oSCap = (int)nd.stkCap, oBCap = (int)nd.blkCap;
      const Ent* oD = ent + nd.dBase;
      const long long* oEd = Ed + nd.dBase;
      const long long eLast = nd.eLast;
      uint32_t ost = 0;
      const int on = n;
      // EVERY pair on the stack but the dummy at position 0 has the boundary n (SubRountine.h:388-434: the first pair is pushed as (i, n); FindBoundary(prev.second,
      // cur.second, ...) searches [n, n) or, below the dummy, returns Ei.size() -- so every later pair is (i, n) too).  Hence: `Db[i] >= top.second` (:398, :450) never
      // holds, a candidate is compared with the stack at Ei[n - 1] only, FindBoundary never searches, and `i1 < top.second` (:326) always holds.  A pair is its D index x;
      // what the slot keeps of the top and of the pair below it: x, Di[x], w(Di[x], Ei[n - 1]) (static) and Dv[x] (dropped when a deposit lands on x).
      int tx = -1, sx = -1; int2 olastB = make_int2(0, 0);               // tx / sx == -1: the dummy
      bool tInfo = false, tDvOk = false, secOk = false, sInfo = false, sDvOk = false;
      float tDv = 0.f, tWe = 0.f, sDv = 0.f, sWe = 0.f; long long tDi = 0, sDi = 0;
      if (need) {
        if (u_i(Z.cTopOk)) {
          tx = u_i(Z.cTop.x); olastB = u_i2(Z.cLastB); tInfo = u_i(Z.topInfoOk) != 0; tDvOk = u_i(Z.topDvOk) != 0; tDv = u_f(Z.topDv); tDi = u_ll(Z.topDi); tWe = u_f(Z.topWe);
          secOk = u_i(Z.secOk) != 0; sx = u_i(Z.sec.x); sInfo = secOk; sDi = u_ll(Z.secDi); sWe = u_f(Z.secWe); sDvOk = u_i(Z.secDvOk) != 0; sDv = u_f(Z.secDv);
        } else if (oTop > 0) { tx = oTop == 1 ? -1 : u_i(oS[oTop - 1].x); olastB = oBlk > 0 ? u_i2(oB[oBlk - 1]) : make_int2(0, 0); }
      }
      // Di, Dv and w(Di, Ei[n - 1]) of a pair read from memory
#define PAIR_INFO(x_, di_, dv_, we_) do { const Ent d__ = oD[(x_)]; (di_) = u_ll(d__.val); (dv_) = u_f(d__.v); (we_) = W((di_), eLast); } while (0)
      // the pair at stack position oTop - 1 after a pop: the remembered second pair, the dummy at position 0, or memory
#define NEXT_DOWN(x_, infoOk_, di_, we_, dv_, dvOk_) do { if (secOk) { (x_) = sx; (di_) = sDi; (we_) = sWe; (infoOk_) = true; (dv_) = sDv; (dvOk_) = sDvOk; secOk = false; sDvOk = false; } \
                                               else if (oTop - 1 == 0) { (x_) = -1; (infoOk_) = true; (dvOk_) = false; } \
                                               else { (x_) = u_i(oS[oTop - 1].x); (infoOk_) = false; (dvOk_) = false; } } while (0)
#define SPUSH(val_) do { const int2 v__ = (val_); if (oTop >= oSCap) { if (coop_grow_pairs(pairs, oStkOff, oSCap, oTop, poolUsed, poolPair, poolPairs, lane)) oS = pairs + oStkOff; else ost |= LRA_ST_CAPACITY; } \
                         if (oTop < oSCap) oS[oTop] = v__; oTop++; } while (0)
#define BPUSH(val_) do { const int2 v__ = (val_); if (oBlk >= oBCap) { if (coop_grow_pairs(pairs, oBlkOff, oBCap, oBlk, poolUsed, poolPair, poolPairs, lane)) oB = pairs + oBlkOff; else ost |= LRA_ST_CAPACITY; } \
                         if (oBlk < oBCap) oB[oBlk] = v__; oBlk++; olastB = v__; } while (0)
      unsigned long long tq = 0;
      if (DBG) { __builtin_amdgcn_s_waitcnt(0); tq = clock64(); tSec[0] += tq - ts0; }
      if (need && now > nd.last) {                                       // Maximization :275-328, the whole wave
        const int olast = nd.last, onow = now;
        bool stop = false;
        for (int i0 = olast + 1; i0 <= onow && !stop && !ost; i0 += 64) {
          const int j = i0 + lane;
          Ent dj; dj.val = 0; dj.b = -1; dj.v = 0;
          long long ej = 0;
          const unsigned long long tl0 = DBG ? clock64() : 0;
          if (j <= onow) { dj = oD[j]; ej = oEd[j]; }                    // Di / Db / Dv and Ei[Db] of 64 candidates: one round
          if (DBG) { __builtin_amdgcn_s_waitcnt(0); tSlot[2] += clock64() - tl0; tRounds++; }
          const int nb = min(64, onow - i0 + 1);
          int t = 0;
          while (t < nb && !ost) {
            const unsigned long long te0 = DBG ? clock64() : 0;
            if (tx != -1) {
              if (!tInfo) { PAIR_INFO(tx, tDi, tDv, tWe); tInfo = true; tDvOk = true; }
              else if (!tDvOk) { tDv = u_f(oD[tx].v); tDvOk = true; }
              bool evt = false;
              if (lane >= t && lane < nb) evt = dj.b == -1 || BEATS(dj.v, dj.val, tDv, tDi, ej);
              const unsigned long long em = __ballot(evt);
              if (!em) break;
              t = __ffsll((long long)em) - 1;
            }
            const int i = i0 + t;
            const int db = rl_i(dj.b, t);
            if (db == -1) { stop = true; break; }
            unsigned long long te1 = 0;
            if (DBG) { tEvents++; te1 = clock64(); tEvA += te1 - te0; }
            const long long di = rl_ll(dj.val, t), edb = rl_ll(ej, t);
            const float dvi = rl_f(dj.v, t);
            bool win = true;                                             // (chosen by the ballot above: it beats the top at Ei[Db[i]] -- unless the top was the dummy)
            if (tx == -1) {                                              // :389-395 (the stack holds the dummy only)
              BPUSH(make_int2(-1, db)); SPUSH(make_int2(i, on));
              sx = -1; secOk = true; sInfo = true; sDvOk = false;
              tx = i; tDv = dvi; tDi = di; tWe = W(di, eLast); tInfo = true; tDvOk = true;
              win = BEATS(dvi, di, tDv, tDi, edb);                        // (a pair against itself, as the reference compares it: never true)
            }
            unsigned long long te2 = 0;
            if (DBG) { te2 = clock64(); tEvB += te2 - te1; }
            if (win) {                                                    // :405
              if (oBlk > 0 && db > olastB.y) BPUSH(make_int2(tx, db));     // (Db[i] < top.second = n always)
              const float wNew = W(di, eLast), sNew = dvi + wNew;         // the candidate at Ei[n - 1]
              int cx = tx; float cDv = tDv, cWe = tWe; long long cDi = tDi; bool cInfo = true, cDvOk = true;
              while (oTop > 0) {                                          // :415-422
                if (cx < 0 || on < 1) { ost |= LRA_ST_OOB_SLOT; break; }
                if (!(sNew > cDv + cWe)) break;
                oTop--;
                if (oTop == 0) { ost |= LRA_ST_OOB_SLOT; break; }
                NEXT_DOWN(cx, cInfo, cDi, cWe, cDv, cDvOk);
                if (cx == -1) break;                                      // the dummy
                if (!cInfo) { PAIR_INFO(cx, cDi, cDv, cWe); cInfo = true; cDvOk = true; }
                else if (!cDvOk) { cDv = u_f(oD[cx].v); cDvOk = true; }
              }
              if (ost) break;
              SPUSH(make_int2(i, on));                                    // FindBoundary: n (see above)
              sx = cx; secOk = cx == -1 || cInfo; sInfo = secOk; sDi = cDi; sWe = cWe; sDv = cDv; sDvOk = cx != -1 && cInfo && cDvOk;
              tx = i; tDv = dvi; tDi = di; tWe = wNew; tInfo = true; tDvOk = true;
            }
            if (DBG) tEvC += clock64() - te2;
            t++;
          }
        }
      }
      // phase 2 (every lane the same values): the flush of Maximization :438-453 (only its `now == m - 1` branch ever pops), FindValueInBlock :322-333 with a
      // wave-cooperative UPPERbound
      float ev = -2.f;
      if (DBG) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t1 = clock64(); tSec[1] += t1 - tq; tq = t1; }
      if (need && !ost) {
        if (now == m - 1) { while (oTop > 1 && tx != -1 && !ost) { BPUSH(make_int2(tx, on)); oTop--; NEXT_DOWN(tx, tInfo, tDi, tWe, tDv, tDvOk); } }
        int i2 = -1;
        if (!ost && oBlk > 0) {
          if (i1 >= olastB.y) i2 = tx;                                    // (i1 < top.second always)
          else {
            int bx;
            const unsigned lo = coop_upper_block(oB, (unsigned)oBlk, i1, lane, &bx);   // UPPERbound :205-221, Block[lo].first with it
            if ((int)lo < oBlk) i2 = bx;
          }
        }
        if (DBG) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t1 = clock64(); tSec[2] += t1 - tq; tq = t1; }
        if (ost || i2 < 0 || i2 >= m) ost |= ost ? ost : LRA_ST_OOB_SLOT;
        else {
          float d2v; long long d2d;
          if (i2 == tx && tInfo && tDvOk) { d2v = tDv; d2d = tDi; }
          else { const Ent d2 = oD[i2]; d2v = u_f(d2.v); d2d = u_ll(d2.val); if (i2 == tx && tInfo) { tDv = d2v; tDvOk = true; } }
          ev = u_f(d2v + W(d2d, ei1) + rate * a.flen[f0 + lf]);
          if (lane == 0) {
            Ap[nd.dBase + nd.nD + i1] = (uint32_t)i2;
            Z.dirty = 1;
            Z.cn.last = now; Z.cn.sTop = (uint32_t)oTop; Z.cn.nBlk = (uint32_t)oBlk; Z.cn.stkOff = oStkOff; Z.cn.blkOff = oBlkOff; Z.cn.stkCap = (uint32_t)oSCap; Z.cn.blkCap = (uint32_t)oBCap;
            Z.cTop = make_int2(tx, tx == -1 ? on + 1 : on); Z.cLastB = olastB; Z.cTopOk = 1;
            Z.topInfoOk = tInfo ? 1 : 0; Z.topDvOk = (tInfo && tDvOk) ? 1 : 0; Z.topDv = tDv; Z.topDi = tDi; Z.topWe = tWe;
            Z.secOk = (secOk && sInfo) ? 1 : 0; Z.sec = make_int2(sx, sx == -1 ? on + 1 : on); Z.secDi = sDi; Z.secWe = sWe; Z.secDv = sDv; Z.secDvOk = (secOk && sInfo && sDvOk) ? 1 : 0;
          }
        }
      }
#undef SPUSH
#undef BPUSH
#undef PAIR_INFO
#undef NEXT_DOWN
      if (ost && lane == 0) atomicOr(&s_bad, ost);
      // Value[ii]: visits apply in the order R family deepest level first, then C family; `val < Ev` keeps the first maximum
      const int vr = (slot / LV) * LV + (LV - 1 - slot % LV);
      if (ev > 0.f && (ev > wBest || (ev == wBest && vr < wRank))) { wBest = ev; wRank = vr; }
      if (DBG) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t1 = clock64(); if (k == 0) tSlot[0] += t1 - ts0; else if (k == 1) tSlot[1] += t1 - ts0; else tSlot[2] += t1 - ts0; tSec[3] += t1 - tq; }
    }
    wave_sync();
    const unsigned long long tb0 = DBG ? clock64() : 0;
    if (ind && lane == 0) {
      const int x = (int)rk;                                              // which start point of the anchor this is
      if (ring) {
        if (wBest > 0.f) (void)__hip_atomic_fetch_max(&r_best[re + x], ((unsigned long long)gen << 48) | ((unsigned long long)__float_as_uint(wBest) << 16) | (unsigned long long)(0xFFFFu - (uint32_t)wRank),
                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t was = __hip_atomic_fetch_add(&r_cnt[re + x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (single) (void)__hip_atomic_fetch_add(&r_cnt[re + 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (was + 1 == (uint32_t)WG_NW * gen) {                            // the last wave in: the anchor's candidate of this start point, for the pass at the end
          const unsigned long long key = __hip_atomic_load(&r_best[re + x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if ((uint32_t)(key >> 48) == gen) best[2 * lf + x] = ((key >> 16) & 0xFFFFFFFFull) << 32 | (unsigned long long)(0xFFFFFFFFu - (0xFFFFu - (uint32_t)(key & 0xFFFFu)));
        }
      } else {
      // the candidate is at L2 before the wave counts itself in: the count's operand depends on the max's return value
      uint32_t one = 1u;
      if (wBest > 0.f) {
        const unsigned long long was = __hip_atomic_fetch_max(&best[2 * lf + x], ((unsigned long long)__float_as_uint(wBest) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)wRank),
                                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        one += (uint32_t)(was == 0xFFFFFFFFFFFFFFFFull);                   // never true: a key's low word is below 2^32 - 1 only ... (value bits of a finite float are not all ones)
      }
      (void)__hip_atomic_fetch_add(&cnt[2 * lf + x], one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (DBG) { __builtin_amdgcn_s_waitcnt(0); tPub += clock64() - tb0; }
  }
  if (DBG && lane == 0) {
    unsigned long long* o = dbgT + 16 * wave;
    o[0] = clock64() - tAll0; o[1] = tSlot[3]; o[2] = tSwitch; o[3] = tDep; o[4] = tPub; o[5] = tSec[0]; o[6] = tSec[1]; o[7] = tSec[2]; o[8] = tSec[3];
    o[9] = tEvA; o[10] = tEvB; o[11] = tEvC; o[12] = (tRounds << 32) | tEvents; o[13] = tStore; o[14] = tSlot[0] + tSlot[1] + tSlot[2]; o[15] = 0;
  }
  __syncthreads();
  // Value[], prev: per anchor the start points in order, `val < Ev` (strict) at each
  if (!s_bad) {
    for (int f = tid; f < F; f += 64 * WG_NW) {
      float val = a.fval[f0 + f];
      int win = -1; unsigned long long wkey = 0;
      for (uint32_t x = 0; x < min(nS[f], 2u); x++) {
        const unsigned long long key = best[2 * f + x];
        const float v = __uint_as_float((uint32_t)(key >> 32));
        if (key && val < v) { val = v; win = (int)x; wkey = key; }
      }
      if (win >= 0) {
        const int vr = (int)(0xFFFFFFFFu - (uint32_t)wkey);
        const int slot = (vr / LV) * LV + (LV - 1 - vr % LV);
        const uint32_t pi = sPos[2 * f + win];
        const uint2 v = visR[(uint64_t)pi * (2 * LV) + slot];
        a.fval[f0 + f] = val; a.fprevNode[f0 + f] = v.x; a.fprevInd[f0 + f] = v.y;
        a.fflags[f0 + f] = (uint8_t)((slot < LV ? 1 : 0) | (((a.hfl[p0 + pi] >> 1) & 1) ? 2 : 0));
      }
    }
  }
  if (tid == 0 && s_bad) atomicOr(&a.status[r], (uint32_t)s_bad);
#undef BAD
#undef W
#undef BEATS
}

// ---- value order, TraceBack, DecidePrimaryChains ------------------------------------------------------------------------
__global__ void k_valkeys(uint64_t f0, uint64_t n, const float* __restrict__ fval, const uint32_t* __restrict__ fragRead,
                          const uint64_t* __restrict__ fragOff, uint64_t* okey, uint32_t* opay) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t g = f0 + i;
  okey[g] = (uint64_t)(0xFFFFFFFFu - __float_as_uint(fval[g]));         // Fragment_valueOrder: value descending (values are >= 0)
  opay[g] = (uint32_t)(g - fragOff[fragRead[g]]);
}

// TraceBack's step Dp[Ep[prev_ind]] of sub-problem prev_sub, resolved for every fragment at once (the chain walk then chases one pointer
// per anchor instead of four dependent loads); taken after ProcessPoint has finished, as the reference's trace back reads it
__global__ void k_pred(uint64_t f0, uint64_t n, int r0, const uint32_t* __restrict__ fragRead, const uint32_t* __restrict__ fprevNode,
                       const uint32_t* __restrict__ fprevInd, const uint32_t* __restrict__ status, const ReadArena* __restrict__ ra, uint32_t* fpred) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t g = f0 + i;
  const uint32_t r = fragRead[g];
  uint32_t pred = NONE;
  const uint32_t pn = fprevNode[g], pi = fprevInd[g];
  if (!status[r] && pn != NONE && pi != NONE) {
    const ReadArena A = ra[(int)r - r0];
    const Node* nodesR = (const Node*)arena_ptr(A.base);
    const uint32_t* apR = (const uint32_t*)(arena_ptr(A.base) + A.apOff);
    const Node nd = nodesR[pn];
    pred = apR[nd.dBase + apR[nd.dBase + nd.nD + pi]];
  }
  fpred[g] = pred;
}

__global__ void k_frag_read(int n_reads, const uint64_t* __restrict__ fragOff, uint32_t* fragRead) {
  int r = blockIdx.x;
  for (uint64_t g = fragOff[r] + threadIdx.x; g < fragOff[r + 1]; g += blockDim.x) fragRead[g] = r;
}

struct TraceArgs {
  int r0, n, numAln, single; float alnthres;
  int boxes, globalK; const uint32_t* fqe; const uint32_t* fte; const int32_t* numAnchors; int32_t* chainNum;   // box mode (DecidePrimaryChains :1587)
  const uint64_t* fragOff; const uint64_t* read_off;
  const uint32_t* fq; const uint32_t* ft; const int32_t* flen; const uint32_t* fcl; const uint32_t* fai;
  const float* fval; const uint32_t* fpred; const uint8_t* fflags; const uint32_t* opay;
  uint8_t* used;
  const ReadArena* ra;
  uint32_t* nChains; uint64_t* chainStart; uint32_t* chainLen; uint32_t* chainBox; float* chainValue;
  uint32_t* ccl; uint32_t* can; uint8_t* clink; uint32_t* cq; uint32_t* ct; int32_t* clen; uint8_t* cstrand; const uint8_t* fstrand;
  const uint32_t* status;
};

constexpr int TRACE_LANES = 16;           // a serial walk with dependent loads per read: fewer lanes per wave, more waves
__global__ void __launch_bounds__(64) sdp_trace(TraceArgs a) {
  if (threadIdx.x >= TRACE_LANES) return;
  const int rr = blockIdx.x * TRACE_LANES + threadIdx.x;
  if (rr >= a.n) return;
  const int r = a.r0 + rr;
  const uint64_t f0 = a.fragOff[r];
  const int total = (int)(a.fragOff[r + 1] - f0);
  a.nChains[r] = 0;
  if (total == 0 || a.status[r]) return;
  if (a.single) {                                                        // SparseDP.h:2417-2434: first anchor of maximal value, plain TraceBack :1521
    float maxv = 0; uint32_t i = 0;
    for (int l = 0; l < total; l++) if (a.fval[f0 + l] > maxv) { maxv = a.fval[f0 + l]; i = l; }
    uint32_t len = 0;
    a.ccl[f0] = i; len = 1;
    uint32_t nx;
    while ((nx = a.fpred[f0 + i]) != NONE && len < (uint32_t)total) {
      a.clink[f0 + len - 1] = (a.fflags[f0 + i] & 2) ? 0 : 1;
      i = nx;
      a.ccl[f0 + len] = i; len++;
    }
    a.clink[f0 + len - 1] = 0;
    const int slot = r * a.numAln;
    a.chainStart[slot] = f0; a.chainLen[slot] = len; a.chainValue[slot] = maxv;
    a.chainBox[4 * slot] = 0; a.chainBox[4 * slot + 1] = 0; a.chainBox[4 * slot + 2] = 0; a.chainBox[4 * slot + 3] = 0;
    for (uint64_t k = f0; k < f0 + len; k++) {
      const uint32_t lf = a.ccl[k];
      a.cq[k] = a.fq[f0 + lf]; a.ct[k] = a.ft[f0 + lf]; a.clen[k] = a.flen[f0 + lf]; a.cstrand[k] = a.fstrand[f0 + lf];
      a.can[k] = a.fai[f0 + lf]; a.ccl[k] = a.fcl[f0 + lf];
    }
    a.nChains[r] = 1;
    return;
  }
  const int readLen = (int)(a.read_off[r + 1] - a.read_off[r]);
  const float best = a.fval[f0 + a.opay[f0]];
  const float thres = a.boxes ? fmaxf(a.alnthres * best, best - (float)(130 * a.globalK)) : a.alnthres * best;   // :1592 / :1663
  int nCh = 0, fv = 0;
  uint64_t out = f0;                                                     // chains are written back to back into the read's fragment range
  uint32_t c0TS = 0, c0TE = 0;
  while ((a.boxes || nCh < a.numAln) && fv < total && a.fval[f0 + a.opay[f0 + fv]] >= thres) {
    uint32_t i = a.opay[f0 + fv];
    const float firstVal = a.fval[f0 + i];
    // TraceBack with `used` (:1351-1438); the chain is written at out.. and rolled back if it runs into a used anchor
    uint32_t len = 0;
    bool abandoned = false;
    if (a.used[f0 + i] == 0) {
      a.ccl[out] = i; len = 1; a.used[f0 + i] = 1;
      uint32_t nx;
      while ((nx = a.fpred[f0 + i]) != NONE) {                          // Dp[Ep[prev_ind]] of sub-problem prev_sub (k_pred)
        if (a.used[f0 + nx] == 0) { a.clink[out + len - 1] = (a.fflags[f0 + i] & 2) ? 0 : 1; i = nx; }
        else { abandoned = true; break; }
        a.ccl[out + len] = i; len++; a.used[f0 + i] = 1;                 // (the reference tests used[i] again here: it has just seen it clear)
      }
      if (abandoned) { for (uint32_t k = 0; k < len; k++) a.used[f0 + a.ccl[out + k]] = 0; len = 0; }
    }
    if (len != 0 && a.boxes) {                                           // :1607-1650
      uint32_t f = a.ccl[out], l = a.ccl[out + len - 1];
      uint32_t QEnd = a.fqe[f0 + f], TEnd = a.fte[f0 + f], QStart = a.fq[f0 + l], TStart = a.ft[f0 + l];
      int na = 0;
      for (uint32_t k = 0; k < len; k++) {
        f = a.ccl[out + k];
        QEnd = max(QEnd, a.fqe[f0 + f]); TEnd = max(TEnd, a.fte[f0 + f]);
        QStart = min(QStart, a.fq[f0 + f]); TStart = min(TStart, a.ft[f0 + f]);
        if (a.numAnchors) na += a.numAnchors[f0 + f];                    // ComputeNumOfAnchors :1577
      }
      if ((double)((float)(QEnd - QStart) / readLen) > 0.005) {
        if (nCh >= a.numAln) break;
        const int slot = r * a.numAln + nCh;
        a.chainStart[slot] = out; a.chainLen[slot] = len; a.chainValue[slot] = firstVal; a.chainNum[slot] = na;
        a.chainBox[4 * slot] = QStart; a.chainBox[4 * slot + 1] = QEnd; a.chainBox[4 * slot + 2] = TStart; a.chainBox[4 * slot + 3] = TEnd;
        a.clink[out + len - 1] = 0;
        nCh++;
        out += len;
      } else break;
    } else if (len != 0) {
      uint32_t f = a.ccl[out], l = a.ccl[out + len - 1];
      uint32_t QEnd = a.fq[f0 + f] + a.flen[f0 + f], QStart = a.fq[f0 + l], TEnd = a.ft[f0 + f] + a.flen[f0 + f], TStart = a.ft[f0 + l];
      for (uint32_t k = 0; k < len; k++) {
        f = a.ccl[out + k];
        QEnd = max(QEnd, a.fq[f0 + f] + (uint32_t)a.flen[f0 + f]);
        QStart = min(QStart, a.fq[f0 + f]);
        TStart = min(TStart, a.ft[f0 + f]);
        TEnd = min(TEnd, a.ft[f0 + f] + (uint32_t)a.flen[f0 + f]);       // min, as the reference has it (:1694)
      }
      if (len >= 3 && QEnd > QStart && (double)((float)(QEnd - QStart) / readLen) > 0.005 && QEnd - QStart >= 200) {
        bool push = false;
        if (nCh == 0) push = true;
        else {                                                           // chains[0].OverlapsOnT(TStart, TEnd, 0.05f)  Chain.h:261
          int ovp = 0;
          if (TStart >= c0TS && TStart < c0TE) ovp = (int)(min(TEnd, c0TE) - TStart);
          else if (TEnd > c0TS && TEnd <= c0TE) ovp = (int)(TEnd - max(TStart, c0TS));
          else if (TStart < c0TS && TEnd > c0TE) ovp = (int)(c0TE - c0TS);
          const float denomA = (float)(c0TE - c0TS);
          push = (ovp / denomA <= 0.05f);
        }
        if (push) {
          const int slot = r * a.numAln + nCh;
          a.chainStart[slot] = out; a.chainLen[slot] = len; a.chainValue[slot] = firstVal;
          a.chainBox[4 * slot] = QStart; a.chainBox[4 * slot + 1] = QEnd; a.chainBox[4 * slot + 2] = TStart; a.chainBox[4 * slot + 3] = TEnd;
          a.clink[out + len - 1] = 0;
          if (nCh == 0) { c0TS = TStart; c0TE = TEnd; }
          nCh++;
          out += len;
        }
      } else break;
    }
    fv++;
  }
  // local fragment index -> (cluster, anchor)
  for (uint64_t k = f0; k < out; k++) {
    const uint32_t lf = a.ccl[k];
    a.cq[k] = a.fq[f0 + lf]; a.ct[k] = a.ft[f0 + lf]; a.clen[k] = a.flen[f0 + lf]; a.cstrand[k] = a.fstrand[f0 + lf];
    a.can[k] = a.fai[f0 + lf]; a.ccl[k] = a.fcl[f0 + lf];
  }
  a.nChains[r] = (uint32_t)nCh;
}

inline size_t sz(size_t n, size_t elem) { return (n * elem + 255) / 256 * 256; }

// Box mode (d_qe != null): clusters are the fragments; d_c_start / d_c_count are null, d_q/d_t/d_qe/d_te/d_len(=Val)/d_c_strand are per box.
// The one-wave-per-read builds of a launch: reads [from, to) of `order` (largest first).  Those of at most 512 points keep their element arrays in LDS
// (three sizes of LDS request, so that small reads do not pay for large ones' occupancy); the rest work from the scratch arena.
// from how many points on a read gets a workgroup (the workgroup kernels' per-point latency is half the wave kernel's, at four times its wave slots): a launch is as long
// as its largest reads' chains.  LRA_SDP_BIG_POINTS overrides all, LRA_SDP_BIG_POINTS_A the first sparse DP's (mode 0) alone
static long sdp_big_points(const lra_ctx* ctx, int mode) {
  if (const char* e = getenv("LRA_SDP_BIG_POINTS")) return atol(e);
  if (ctx->sdp_inner) return 1500;
  if (mode == 0) { if (const char* e = getenv("LRA_SDP_BIG_POINTS_A")) return atol(e); return 2500; }   // (its largest reads have ~5000 points: the top few hundred as workgroups, 93 -> 83 ms)
  if (const char* e = getenv("LRA_SDP_BIG_POINTS_2")) return atol(e);
  // Two-stage batches: the other half of another batch fills what a long tail leaves idle, and a wave per read costs a third of the device time per point that a
  // workgroup per read does -- so only the reads that would make the wave launch far longer than everything beside it stay workgroup jobs (measured, two-stage step:
  // 6000: 980 ms, 9000: 954, 12000: 943, 14000: 933, 16000+: up again; each with at most one job per CU, see maxBig)
  return ctx->pipelined ? 14000 : 6000;
}
template <bool EMIT>
static void launch_small_builds(lra_ctx* ctx, const BuildArgs& ba, const uint32_t* d_order, const std::vector<uint32_t>& h_order, const uint64_t* h_pt, int from, int to) {
  static const bool noLds = getenv("LRA_SDP_BUILD_NOLDS") != nullptr;
  hipStream_t st = ctx->stream;
  auto pts = [&](int i) { return (long)(h_pt[h_order[i] + 1] - h_pt[h_order[i]]); };
  int at = from;
  // (measured: at 28 KB -- up to 1024 points -- five waves per CU are slower from LDS than 32 from the arena; up to 768 points is a wash)
  const long caps[3] = {512, 256, 128};
  int cut[4];                                                            // [from, cut0): arena;  [cut0, cut1): <= 512;  [cut1, cut2): <= 256;  [cut2, to): <= 128
  for (int c = 0; c < 3; c++) { while (at < to && (noLds || pts(at) > caps[c])) at++; cut[c] = at; }
  cut[3] = to;
  if (cut[0] > from) {                                                   // arena: 16-bit indices below 16384 points (2 x node index + side must fit)
    int mid = from;
    while (mid < cut[0] && pts(mid) >= 16384) mid++;
    if (mid > from) { BuildArgs bb = ba; bb.order = d_order + from; hipLaunchKernelGGL((sdp_build<EMIT, 1, 0>), dim3(mid - from), dim3(64), 0, st, bb); }
    if (cut[0] > mid) {
      BuildArgs bb = ba; bb.order = d_order + mid;
      // waves per SIMD the register budget is set for: 8; beside another batch's half (two-stage batches) 6 -- fewer, fatter waves leave the other half's launches room
      // (two-stage step 952 -> 937 ms; in the one call 8 is the faster one)
      static const int occEnv = getenv("LRA_SDP_BUILD_OCC") ? atoi(getenv("LRA_SDP_BUILD_OCC")) : 0;
      const int occ = occEnv ? occEnv : ctx->pipelined ? 6 : 8;
      if (bb.stat) { if (occ == 6) hipLaunchKernelGGL((sdp_build<EMIT, 1, 2, 6, true>), dim3(cut[0] - mid), dim3(64), 0, st, bb); else hipLaunchKernelGGL((sdp_build<EMIT, 1, 2, 8, true>), dim3(cut[0] - mid), dim3(64), 0, st, bb); }
      else if (occ == 4) hipLaunchKernelGGL((sdp_build<EMIT, 1, 2, 4>), dim3(cut[0] - mid), dim3(64), 0, st, bb);
      else if (occ == 5) hipLaunchKernelGGL((sdp_build<EMIT, 1, 2, 5>), dim3(cut[0] - mid), dim3(64), 0, st, bb);
      else if (occ == 6) hipLaunchKernelGGL((sdp_build<EMIT, 1, 2, 6>), dim3(cut[0] - mid), dim3(64), 0, st, bb);
      else hipLaunchKernelGGL((sdp_build<EMIT, 1, 2>), dim3(cut[0] - mid), dim3(64), 0, st, bb);
    }
  }
  for (int c = 0; c < 3; c++) {
    const int n = cut[c + 1] - cut[c];
    if (n <= 0) continue;
    BuildArgs bb = ba; bb.order = d_order + cut[c];
    hipLaunchKernelGGL((sdp_build<EMIT, 1, 1>), dim3(n), dim3(64), (size_t)(28 * caps[c] + 32), st, bb);
  }
}

int sdp_run(lra_ctx* ctx, int n_reads, const uint64_t* d_cluster_off, const uint64_t* d_c_start, const uint32_t* d_c_count,
            const int32_t* d_c_strand, const uint32_t* d_q, const uint32_t* d_t, const int32_t* d_len, const uint64_t* d_read_off,
            const float* d_rate, const lra_sdp_opts* opts, lra_chain_result* out, const uint32_t* d_qe, const uint32_t* d_te,
            const int32_t* d_num_anchors) {
  const bool boxes = d_qe != nullptr;
  if (!ctx || !out || !opts || n_reads < 0) return LRA_ERR_INVALID;
  if (opts->NumAln < 1 || opts->NumAln > MAXALN) return lra_set_err(ctx, LRA_ERR_INVALID, "NumAln must be 1..%d", MAXALN);
  memset(out, 0, sizeof *out);
  out->n_reads = n_reads; out->num_aln = opts->NumAln;
  if (n_reads == 0) return LRA_OK;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t n1 = (size_t)n_reads + 1;
  // InitPWL on the host (SubRountine.h:43-99), host libm as in the reference
  PwlTab pw;
  {
    static const long long stv[25] = {0,    5,    10,   20,   40,   80,    100,   200,   300,   500,   1000,  2000, 3000,
                                      4000, 5000, 6000, 7000, 8000, 9000, 15000, 20000, 30000, 40000, 50000, 100000};
    float intercept = opts->gapopen, scalar = opts->gapextend, root = opts->gaproot, vals[25];
    for (int i = 0; i < 25; i++) { pw.stops[i] = stv[i]; pw.slope[i] = 0; pw.inter[i] = 0; }
    vals[0] = 0;
    for (int i = 1; i < 25; i++) { if (i <= 2) intercept = 0; vals[i] = intercept + scalar * std::pow((float)stv[i], 1 / root); }
    for (int i = 0; i < 24; i++) {
      float slope = (vals[i + 1] - vals[i]) / (stv[i + 1] - stv[i]);
      if (stv[i] <= 10) { pw.slope[i] = 0; pw.inter[i] = 0; }
      else { pw.slope[i] = slope; pw.inter[i] = vals[i] - stv[i] * slope + intercept; }
    }
    pw.c1 = opts->gapCeiling1; pw.c2 = opts->gapCeiling2;
  }
  // -w as a table for small distances (the kernels copy it to LDS); left out when a penalty does not fit 16 bits
  short* d_penTab = (short*)lra_ensure(ctx, 190, PEN_TAB_WG * sizeof(short) + 64);
  int penN = 0;
  if (d_penTab) {
    int* d_bad = (int*)(d_penTab + PEN_TAB_WG);
    LRA_HIP_CHECK(ctx, hipMemsetAsync(d_bad, 0, 4, st));
    hipLaunchKernelGGL(k_pen_table, dim3(PEN_TAB_WG / 256), dim3(256), 0, st, pw, PEN_TAB_WG, d_penTab, d_bad);
    int h_bad = 1;
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(&h_bad, d_bad, 4, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (!h_bad) penN = PEN_TAB_WG;
  }
  std::vector<uint64_t> h_off(n1);
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(h_off.data(), d_cluster_off, n1 * 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  const uint64_t NC = h_off[n_reads];
  // ---- batch-level buffers: clusters and reads
  const size_t nslot = (size_t)n_reads * opts->NumAln;
  size_t needA = sz(NC + 1, 4) * 3 + sz(NC + 2, 8) * 2 + sz(n1, 8) * 2 + sz(n1, 4) * 2 + sz(nslot, 8) + sz(nslot, 4) * 3 + sz(4 * nslot, 4) + 4096;
  char* wa = (char*)lra_ensure(ctx, 7, needA);
  if (!wa) return LRA_ERR_NOMEM;
  auto take = [](char*& w, size_t n, size_t e) { char* p = w; w += sz(n, e); return p; };
  uint32_t* clusFragCnt = (uint32_t*)take(wa, NC + 1, 4); uint32_t* clusPtCnt = (uint32_t*)take(wa, NC + 1, 4); uint32_t* clusRead = (uint32_t*)take(wa, NC + 1, 4);
  uint64_t* clusFragOff = (uint64_t*)take(wa, NC + 2, 8); uint64_t* clusPtOff = (uint64_t*)take(wa, NC + 2, 8);
  uint64_t* fragOff = (uint64_t*)take(wa, n1, 8); uint64_t* ptOff = (uint64_t*)take(wa, n1, 8);
  uint32_t* status = (uint32_t*)take(wa, n1, 4); uint32_t* nChains = (uint32_t*)take(wa, n1, 4);
  uint64_t* chainStart = (uint64_t*)take(wa, nslot, 8); uint32_t* chainLen = (uint32_t*)take(wa, nslot, 4); float* chainValue = (float*)take(wa, nslot, 4);
  uint32_t* chainBox = (uint32_t*)take(wa, 4 * nslot, 4);
  int32_t* chainNum = (int32_t*)take(wa, nslot, 4);
  LRA_HIP_CHECK(ctx, hipMemsetAsync(chainLen, 0, nslot * 4, st));
  if (NC > 0) {
    lra_time_begin(ctx, ctx->sdp_inner ? "sdp_inner_points" : "sdp_points");
    hipLaunchKernelGGL(k_cluster_counts, dim3((unsigned)((NC + 255) / 256)), dim3(256), 0, st, NC, boxes ? nullptr : d_c_count, clusFragCnt, clusPtCnt,
                       opts->mode == LRA_SDP_SINGLE_CLUSTER);
    lra_time_end(ctx);
  }
  { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)NC, clusFragCnt, clusFragOff); if (rc) return rc; }
  { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)NC, clusPtCnt, clusPtOff); if (rc) return rc; }
  hipLaunchKernelGGL(k_read_offsets, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, st, n_reads, d_cluster_off, clusFragOff, clusPtOff, fragOff, ptOff,
                     clusRead, status, nChains);
  std::vector<uint64_t> h_frag(n1), h_pt(n1);
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(h_frag.data(), fragOff, n1 * 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(h_pt.data(), ptOff, n1 * 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  const uint64_t NF = h_frag[n_reads], NP = h_pt[n_reads];
  out->n_frags = NF; out->n_points = NP;
  // ---- fragments
  size_t needF = sz(NF + 1, 4) * 16 + sz(2 * NF + 2, 4) + sz(NF + 1, 1) * 5 + sz(NF + 1, 8) * 2 + 4096;
  char* wf = (char*)lra_ensure(ctx, 8, needF);
  if (!wf) return LRA_ERR_NOMEM;
  uint32_t* fq = (uint32_t*)take(wf, NF + 1, 4); uint32_t* ft = (uint32_t*)take(wf, NF + 1, 4); int32_t* flen = (int32_t*)take(wf, NF + 1, 4);
  uint32_t* fcl = (uint32_t*)take(wf, NF + 1, 4); uint32_t* fai = (uint32_t*)take(wf, NF + 1, 4); float* fval = (float*)take(wf, NF + 1, 4);
  uint32_t* fprevNode = (uint32_t*)take(wf, NF + 1, 4); uint32_t* fprevInd = (uint32_t*)take(wf, NF + 1, 4);
  uint32_t* ccl = (uint32_t*)take(wf, NF + 1, 4); uint32_t* can = (uint32_t*)take(wf, NF + 1, 4);
  uint8_t* fflags = (uint8_t*)take(wf, NF + 1, 1); uint8_t* used = (uint8_t*)take(wf, NF + 1, 1); uint8_t* clink = (uint8_t*)take(wf, NF + 1, 1);
  uint8_t* fstrand = (uint8_t*)take(wf, NF + 1, 1); uint8_t* cstrand = (uint8_t*)take(wf, NF + 1, 1);
  uint32_t* cq = (uint32_t*)take(wf, NF + 1, 4); uint32_t* ct = (uint32_t*)take(wf, NF + 1, 4); int32_t* clen = (int32_t*)take(wf, NF + 1, 4);
  uint64_t* okey = (uint64_t*)take(wf, NF + 1, 8);
  uint32_t* fqe = (uint32_t*)take(wf, NF + 1, 4); uint32_t* fte = (uint32_t*)take(wf, NF + 1, 4);
  // ---- points
  size_t needP = sz(NP + 1, 8) * 3 + sz(NP + 1, 4) * 11 + sz(NP + 1, 1) * 2 + sz(NF + 1, 4) * 2 + 4096;
  char* wp = (char*)lra_ensure(ctx, 9, needP);
  if (!wp) return LRA_ERR_NOMEM;
  uint64_t* key1 = (uint64_t*)take(wp, NP + 1, 8); uint64_t* key2 = (uint64_t*)take(wp, NP + 1, 8); uint64_t* key3 = (uint64_t*)take(wp, NP + 1, 8);
  uint32_t* pay1 = (uint32_t*)take(wp, NP + 1, 4); uint32_t* pay2 = (uint32_t*)take(wp, NP + 1, 4); uint32_t* pay3 = (uint32_t*)take(wp, NP + 1, 4);
  uint32_t* iq = (uint32_t*)take(wp, NP + 1, 4); uint32_t* it = (uint32_t*)take(wp, NP + 1, 4); uint32_t* ifr = (uint32_t*)take(wp, NP + 1, 4);
  uint32_t* ptRead = (uint32_t*)take(wp, NP + 1, 4);
  uint32_t* hq = (uint32_t*)take(wp, NP + 1, 4); uint32_t* ht = (uint32_t*)take(wp, NP + 1, 4); uint32_t* hfr = (uint32_t*)take(wp, NP + 1, 4);
  uint32_t* spare = (uint32_t*)take(wp, NP + 1, 4);          // TraceBack's predecessor per fragment (k_pred)
  uint8_t* ifl = (uint8_t*)take(wp, NP + 1, 1); uint8_t* hfl = (uint8_t*)take(wp, NP + 1, 1);
  uint32_t* opay = (uint32_t*)take(wp, NF + 1, 4); uint32_t* fragRead = (uint32_t*)take(wp, NF + 1, 4);
  out->d_n_chains = nChains; out->d_chain_start = chainStart; out->d_chain_len = chainLen; out->d_chain_box = chainBox; out->d_chain_value = chainValue;
  out->d_chain_cluster = ccl; out->d_chain_anchor = can; out->d_chain_link = clink; out->d_chain_q = cq; out->d_chain_t = ct; out->d_chain_alen = clen;
  out->d_chain_strand = cstrand; out->d_frag_off = fragOff; out->d_frag_val = fval; out->d_status = status;
  out->d_chain_num_anchors = boxes ? chainNum : nullptr;
  if (NF == 0) { LRA_HIP_CHECK(ctx, hipStreamSynchronize(st)); return LRA_OK; }
  {
    PtArgs pa;
    pa.nc = NC; pa.cluster_off = d_cluster_off; pa.c_start = d_c_start; pa.c_count = d_c_count; pa.c_strand = d_c_strand; pa.q = d_q; pa.t = d_t; pa.len = d_len;
    pa.clusRead = clusRead; pa.clusFragOff = clusFragOff; pa.clusPtOff = clusPtOff; pa.fragOff = fragOff; pa.ptOff = ptOff; pa.rate_in = d_rate; pa.rate = opts->rate; pa.single = opts->mode == LRA_SDP_SINGLE_CLUSTER;
    pa.fq = fq; pa.ft = ft; pa.flen = flen; pa.fcl = fcl; pa.fai = fai; pa.fval = fval; pa.fprevNode = fprevNode; pa.fprevInd = fprevInd; pa.fflags = fflags; pa.used = used; pa.fstrand = fstrand;
    pa.qe = d_qe; pa.te = d_te; pa.fqe = fqe; pa.fte = fte;
    pa.key1 = key1; pa.pay1 = pay1; pa.iq = iq; pa.it = it; pa.ifl = ifl; pa.ifr = ifr; pa.ptRead = ptRead;
    lra_time_begin(ctx, ctx->sdp_inner ? "sdp_inner_points" : "sdp_points");
    if (pa.qe) hipLaunchKernelGGL(k_points, dim3((unsigned)((NC + 127) / 128)), dim3(128), 0, st, pa);
    else hipLaunchKernelGGL(k_points, dim3((unsigned)((NC + 3) / 4)), dim3(256), 0, st, pa);
    hipLaunchKernelGGL(k_frag_read, dim3(n_reads), dim3(64), 0, st, n_reads, fragOff, fragRead);
    lra_time_end(ctx);
  }
  if (const char* dumpPath = getenv("LRA_SDP_DUMP")) {                     // analysis hook (tools/sdp_case_stats.py): the inputs of the call's largest jobs + all job sizes
    static int callNo = 0;
    const int topK = getenv("LRA_SDP_DUMP_TOP") ? atoi(getenv("LRA_SDP_DUMP_TOP")) : 8;
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
    std::vector<uint32_t> idx(n_reads);
    for (int i = 0; i < n_reads; i++) idx[i] = (uint32_t)i;
    std::partial_sort(idx.begin(), idx.begin() + std::min(topK, n_reads), idx.end(), [&](uint32_t x, uint32_t y) { return h_frag[x + 1] - h_frag[x] > h_frag[y + 1] - h_frag[y]; });
    std::string pth = std::string(dumpPath) + ".call" + std::to_string(callNo) + (ctx->sdp_inner ? "i" : "") + ".bin";
    if (FILE* f = fopen(pth.c_str(), "wb")) {
      for (int k = 0; k < std::min(topK, n_reads); k++) {
        const uint32_t r = idx[k];
        const uint64_t a0 = h_frag[r], n = h_frag[r + 1] - a0;
        if (n == 0) continue;
        std::vector<uint32_t> q(n), t(n), cl(n); std::vector<int32_t> ln(n); std::vector<uint8_t> sd(n);
        (void)hipMemcpy(q.data(), fq + a0, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(t.data(), ft + a0, n * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(ln.data(), flen + a0, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(cl.data(), fcl + a0, n * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(sd.data(), fstrand + a0, n, hipMemcpyDeviceToHost);
        std::vector<int32_t> coff; std::vector<uint8_t> cst;
        for (uint64_t i = 0; i < n; i++) if (i == 0 || cl[i] != cl[i - 1]) { coff.push_back((int32_t)i); cst.push_back(sd[i]); }
        coff.push_back((int32_t)n);
        int hdr[4] = {opts->mode, (int)cst.size(), (int)n, 30000};
        float rate = opts->rate;
        fwrite(hdr, 4, 4, f); fwrite(&rate, 4, 1, f); fwrite(coff.data(), 4, coff.size(), f); fwrite(cst.data(), 1, cst.size(), f);
        fwrite(q.data(), 4, n, f); fwrite(t.data(), 4, n, f); fwrite(ln.data(), 4, n, f);
      }
      fclose(f);
    }
    pth = std::string(dumpPath) + ".call" + std::to_string(callNo) + (ctx->sdp_inner ? "i" : "") + ".sizes";
    if (FILE* f = fopen(pth.c_str(), "wb")) { fwrite(h_pt.data(), 8, n1, f); fclose(f); }
    callNo++;
  }
  struct Retag { lra_ctx* c; Retag(lra_ctx* x) : c(x) { c->sort_tag = c->sdp_inner ? "sdp_inner_sort" : "sdp_sort"; c->sort_fb_tag = c->sdp_inner ? "sdp_inner_sort_fallback" : "sdp_sort_fallback"; c->sort_short = true; } ~Retag() { c->sort_tag = "sort"; c->sort_fb_tag = "sort_fallback"; c->sort_short = false; } } retag(ctx);
  // the two point orders: (q, t, ind) / (t, q, ind) keys repeat only where two anchors share a corner, so the radix path takes nearly all lists
  { int rc = lra_sort_mostly_unique_batch(ctx, n_reads, ptOff, NP, key1, pay1, key3, pay3, 64); if (rc) return rc; }   // sort(H1, SortByRowOp)  :2171
  lra_time_begin(ctx, ctx->sdp_inner ? "sdp_inner_points" : "sdp_points");
  hipLaunchKernelGGL(k_gather, dim3((unsigned)((NP + 255) / 256)), dim3(256), 0, st, NP, ptRead, ptOff, pay1, iq, it, ifl, ifr, hq, ht, hfl, hfr, key2, pay2,
                     key3, pay3);
  lra_time_end(ctx);
  { int rc = lra_sort_mostly_unique_batch(ctx, n_reads, ptOff, NP, key2, pay2, key1, pay1, 63); if (rc) return rc; }   // sort(H2, SortByColOp)  :2174
  // diagonal order per point class: any sorted order serves (ties are the same diagonal), so this one is a segmented radix sort -- the
  // exact introsort degenerates on the long runs of equal diagonals.  Sorted into the (now free) key1 / pay1 buffers.
  {
    size_t temp_bytes = 0;
    (void)lra_segsort_pairs(ctx, nullptr, temp_bytes, nullptr, nullptr, nullptr, nullptr, (unsigned int)NP, (unsigned int)n_reads, nullptr, nullptr, 0, 42, st);
    void* temp = lra_scratch(ctx, 2, temp_bytes + 256);
    if (!temp) return LRA_ERR_NOMEM;
    lra_time_begin(ctx, ctx->sdp_inner ? "sdp_inner_sort" : "sdp_sort");
    hipError_t e = lra_segsort_pairs(ctx, temp, temp_bytes, key3, key1, pay3, pay1, (unsigned int)NP, (unsigned int)n_reads, ptOff, ptOff + 1, 0, 42, st);
    lra_time_end(ctx);
    if (e != hipSuccess) return lra_set_err(ctx, LRA_ERR_HIP, "segmented sort: %s", hipGetErrorString(e));
  }
  LRA_HIP_CHECK(ctx, hipGetLastError());
  // ---- chunks of reads: decompositions, ProcessPoint, trace
  // one chunk if it fits: the kernels' duration is set by the longest read once the chip is no longer full, so few large launches
  // beat many small ones (32768 reads, sdp_process: 4 chunks 420 ms, 2 chunks 255 ms, 1 chunk 187 ms)
  uint64_t chunkPts = 96ull << 20;            // ~75 GB of arenas per chunk: a 32768-read batch of 30 kb reads (80 M points) is one chunk
  if (const char* e = getenv("LRA_SDP_CHUNK_MPOINTS")) { const long v = atol(e); if (v > 0) chunkPts = (uint64_t)v << 20; }   // tuning knob
  uint64_t totalEntries = 0;
  for (int r0 = 0; r0 < n_reads;) {
    int r1 = r0 + 1;
    while (r1 < n_reads && h_pt[r1 + 1] - h_pt[r0] <= chunkPts) r1++;
    const int nr = r1 - r0;
    const uint64_t cp = h_pt[r1] - h_pt[r0];
    if (cp == 0) { r0 = r1; continue; }
    const size_t nr1 = (size_t)nr + 1;
    size_t needS = sz(34 * cp + 64 * (size_t)nr + 64, 4) + sz(nr1, 4) * 8 + sz(nr1 + 1, 8) * 3 + sz(nr1, sizeof(ReadArena)) + 4096;
    char* ws = (char*)lra_ensure(ctx, 10, needS);
    if (!ws) return LRA_ERR_NOMEM;
    uint32_t* scratch = (uint32_t*)take(ws, 34 * cp + 64 * (size_t)nr + 64, 4);
    uint32_t* cntE = (uint32_t*)take(ws, nr1, 4); uint32_t* cntN = (uint32_t*)take(ws, nr1, 4); uint32_t* cntD = (uint32_t*)take(ws, nr1, 4);
    uint32_t* cntV = (uint32_t*)take(ws, nr1, 4); uint32_t* cntRC = (uint32_t*)take(ws, nr1, 4); uint32_t* order = (uint32_t*)take(ws, nr1, 4); uint32_t* order2 = (uint32_t*)take(ws, nr1, 4); uint32_t* poolUsed = (uint32_t*)take(ws, nr1, 4);
    std::vector<uint32_t> h_orderAll, h_prev;
    {
      // largest first, equal sizes in read order: a counting sort by size (a13's inner sparse DP orders 170 k jobs: 11 ms of std::sort with the device idle)
      std::vector<uint32_t> h_order(nr);
      uint64_t maxP = 0;
      for (int i = 0; i < nr; i++) maxP = std::max<uint64_t>(maxP, h_pt[r0 + i + 1] - h_pt[r0 + i]);
      if (maxP <= (uint64_t(1) << 24)) {
        std::vector<uint32_t> at(maxP + 2, 0);
        for (int i = 0; i < nr; i++) at[maxP - (h_pt[r0 + i + 1] - h_pt[r0 + i]) + 1]++;
        for (uint64_t v = 1; v <= maxP + 1; v++) at[v] += at[v - 1];
        for (int i = 0; i < nr; i++) h_order[at[maxP - (h_pt[r0 + i + 1] - h_pt[r0 + i])]++] = (uint32_t)i;
      } else {
        for (int i = 0; i < nr; i++) h_order[i] = (uint32_t)i;
        std::sort(h_order.begin(), h_order.end(), [&](uint32_t x, uint32_t y) {
          const uint64_t px = h_pt[r0 + x + 1] - h_pt[r0 + x], py = h_pt[r0 + y + 1] - h_pt[r0 + y];
          return px != py ? px > py : x < y;
        });
      }
      LRA_HIP_CHECK(ctx, hipMemcpyAsync(order, h_order.data(), (size_t)nr * 4, hipMemcpyHostToDevice, st));
      LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
      h_orderAll = h_order;
    }
    uint64_t* entOff = (uint64_t*)take(ws, nr1 + 1, 8); uint64_t* bytes = (uint64_t*)take(ws, nr1 + 1, 8); uint64_t* byteOff = (uint64_t*)take(ws, nr1 + 1, 8);
    ReadArena* ra = (ReadArena*)take(ws, nr1, sizeof(ReadArena));
    BuildArgs ba;
    memset(&ba, 0, sizeof ba);
    ba.r0 = r0; ba.n = nr; ba.ptOff = ptOff; ba.hq = hq; ba.ht = ht; ba.hfl = hfl; ba.h2 = pay2; ba.key3 = key1; ba.pay3 = pay1; ba.scratch = scratch;
    ba.cntEntries = cntE; ba.cntNodes = cntN; ba.cntD = cntD; ba.cntV = cntV; ba.cntRC = cntRC; ba.status = status; ba.order = order; ba.stat = nullptr;
    static const bool buildStat = getenv("LRA_SDP_BUILD_STAT") != nullptr;
    if (buildStat) { (void)hipMalloc((void**)&ba.stat, 16 * 8); (void)hipMemsetAsync(ba.stat, 0, 16 * 8, st); }
    // the count pass: the same divide as the emit pass, for the sizes of a read's blocks -- run for everything only on request (LRA_SDP_ONEPASS=0, LRA_SDP_RATIOS);
    // otherwise the blocks are laid out from k_arena_estimate and only the reads that outgrow them are counted (attempt 1 below)
    const bool onePassEnv = !(getenv("LRA_SDP_ONEPASS") && getenv("LRA_SDP_ONEPASS")[0] == '0');
    const bool onePass = onePassEnv && !getenv("LRA_SDP_RATIOS");
    auto count_pass = [&](const uint32_t* d_ord, const std::vector<uint32_t>& h_ord, int n) {
      lra_time_begin(ctx, ctx->sdp_inner ? "sdp_inner_build_count" : "sdp_build_count");
      // reads ordered largest first: the large ones get a 1024-thread workgroup each, beside the wave-per-read launch
      const long big_pts = sdp_big_points(ctx, opts->mode);
      int nb0 = 0;
      while (nb0 < n && (long)(h_pt[r0 + h_ord[nb0] + 1] - h_pt[r0 + h_ord[nb0]]) >= big_pts) nb0++;
      const bool forked = nb0 > 0 && n > nb0;
      BuildArgs bc = ba; bc.order = d_ord; bc.n = n;
      if (nb0 > 0) hipLaunchKernelGGL((sdp_build<false, 16>), dim3(nb0), dim3(1024), 0, forked ? lra_side_fork(ctx) : st, bc);
      if (n > nb0) launch_small_builds<false>(ctx, bc, d_ord, h_ord, h_pt.data() + r0, nb0, n);
      if (forked) lra_side_join(ctx);
      lra_time_end(ctx);
    };
    uint64_t totE = 0;
    uint32_t maxRC = 0;                                                    // over the chunk's reads: which sdp_process_wg variant serves its large reads
    auto read_rc = [&]() -> int {                                          // ... and the entries the reads have (the count pass's cntE, or what the emit pass found: cntV)
      std::vector<uint32_t> h_rc(nr), h_e(nr);
      LRA_HIP_CHECK(ctx, hipMemcpyAsync(h_rc.data(), cntRC, (size_t)nr * 4, hipMemcpyDeviceToHost, st));
      LRA_HIP_CHECK(ctx, hipMemcpyAsync(h_e.data(), onePass ? cntV : cntE, (size_t)nr * 4, hipMemcpyDeviceToHost, st));
      LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
      maxRC = 0; totE = 0;
      for (uint32_t v : h_rc) maxRC = std::max(maxRC, v);
      for (uint32_t v : h_e) totE += v;
      return LRA_OK;
    };
    if (!onePass) { count_pass(order, h_orderAll, nr); int rc = read_rc(); if (rc) return rc; }
    else {
      float fE = opts->mode == 0 ? 10.0f : 8.5f, fN = 2.0f;
      if (const char* e = getenv("LRA_SDP_ESTIMATE")) { float x = 0, y = 0; if (sscanf(e, "%f,%f", &x, &y) == 2 && x > 0 && y > 0) { fE = x; fN = y; } }   // (tests: estimates that many reads outgrow)
      hipLaunchKernelGGL(k_arena_estimate, dim3((nr + 255) / 256), dim3(256), 0, st, nr, r0, ptOff, fE, fN, cntE, cntN, cntD);
    }
    if (getenv("LRA_SDP_RATIOS")) {                                        // analysis: entries / nodes / D entries per point over the chunk's reads
      std::vector<uint32_t> hE(nr), hN(nr), hD(nr);
      (void)hipMemcpy(hE.data(), cntE, (size_t)nr * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(hN.data(), cntN, (size_t)nr * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(hD.data(), cntD, (size_t)nr * 4, hipMemcpyDeviceToHost);
      std::vector<double> rE, rN, rD; double sE = 0, sP = 0;
      for (int i = 0; i < nr; i++) { const double P = (double)(h_pt[r0 + i + 1] - h_pt[r0 + i]); if (P < 64) continue; rE.push_back(hE[i] / P); rN.push_back(hN[i] / P); rD.push_back(hD[i] / (double)std::max(1u, hE[i])); sE += hE[i]; sP += P; }
      auto pr = [&](const char* nm, std::vector<double>& v) { if (v.empty()) return; std::sort(v.begin(), v.end()); fprintf(stderr, "[sdp ratios] %s: min %.2f p50 %.2f p90 %.2f p99 %.2f p99.9 %.2f max %.2f\n", nm, v[0], v[v.size() / 2], v[v.size() * 9 / 10], v[v.size() * 99 / 100], v[(size_t)(v.size() * 0.999)], v.back()); };
      fprintf(stderr, "[sdp ratios] mode %d inner %d reads %d: entries per point overall %.2f\n", opts->mode, (int)ctx->sdp_inner, nr, sE / std::max(1.0, sP));
      pr("entries / point", rE); pr("nodes / point", rN); pr("D entries / entries", rD);
    }
    // attempt 0: all reads of the chunk; attempts 1, 2: the reads whose candidate stack / Block outgrew its slots, with 8x / 64x the slots
    std::vector<uint32_t> h_status(nr), h_sub;
    uint32_t* subOrder = order;
    int nsub = nr;
    for (int att = 0; att < 3 && nsub > 0; att++) {
      const int shift = 3 * att;
      const int slot = att == 0 ? 12 : 21 + att;
      if (onePass && att == 1) count_pass(subOrder, h_prev, nsub);         // (exact sizes for the reads that come back: some outgrew their estimated blocks)
      static const bool dbg = getenv("LRA_SDP_DBG") != nullptr;
      // which reads of this attempt get a workgroup each (they are ordered by their number of points, largest first)
      const std::vector<uint32_t>& ordAtt = att == 0 ? h_orderAll : h_prev;
      int nbig = 0;
      {
        const long big_pts = sdp_big_points(ctx, opts->mode);   // (tests lower it to run small reads through the workgroup kernels)
        // ... but no more of them than the device runs side by side (a workgroup holds 16 wave slots for a per-point latency a third of the wave kernel's, at 3.5 times its
        // wave-time per point): beyond that the large reads queue up behind each other, and the ones further down the order are better off as one wave each
        static const int maxBigEnv = getenv("LRA_SDP_MAX_BIG") ? std::max(0, atoi(getenv("LRA_SDP_MAX_BIG"))) : -1;
        // (two-stage batches: one round of workgroup jobs -- one per CU, all resident at once; 192: 1026 ms, 256: 933, 320: 955)
        const int maxBigAll = maxBigEnv >= 0 ? maxBigEnv : ctx->pipelined ? ctx->num_cu : (1 << 30);
        static const int maxBigA = getenv("LRA_SDP_MAX_BIG_A") ? std::max(0, atoi(getenv("LRA_SDP_MAX_BIG_A"))) : 512;
        const int maxBig = (opts->mode == 0 && !ctx->sdp_inner) ? std::min(maxBigAll, maxBigA) : maxBigAll;
        while (nbig < nsub && nbig < maxBig && (long)(h_pt[r0 + ordAtt[nbig] + 1] - h_pt[r0 + ordAtt[nbig]]) >= big_pts) nbig++;
      }
      const bool forked = nbig > 0 && nsub > nbig;
      // LARGE READS FIRST (attempt 0, LRA_SDP_BIG_FIRST=0 switches it off).  The stage is as long as the chain build -> ProcessPoint of its largest read, and that read is one
      // of the workgroup jobs: their build goes first, on the side stream, with nothing of this stage beside it; their ProcessPoint launch follows it on that stream at
      // once -- what it needs from the host (the per-anchor words' offsets, the kernel variant) is made ready before the builds: the variant from the large reads' own rows /
      // columns (k_big_lines) instead of the emit pass's counts --; the small reads' builds start behind the large reads' build and their wave-per-read ProcessPoint launch
      // behind those, beside the workgroup launch as before.  What used to be  max(builds) + max(ProcessPoint launches)  is  build_large + max(wg, builds_small + wave).
      const bool bigFirstEnv = !(getenv("LRA_SDP_BIG_FIRST") && getenv("LRA_SDP_BIG_FIRST")[0] == '0');   // (read per call: the tests run both orders in one process)
      const bool early = bigFirstEnv && att == 0 && onePass && forked && !dbg;
      // the large reads' per-anchor words (see sdp_process_wg)
      char* wsc = nullptr; uint64_t* dwoff = nullptr; uint32_t* d_maxLines = nullptr;
      std::vector<uint64_t> woff((size_t)nbig + 1, 0);
      if (nbig > 0) {
        for (int i = 0; i < nbig; i++) {
          const uint64_t rdx = (uint64_t)r0 + ordAtt[i];
          const uint64_t Fr = h_frag[rdx + 1] - h_frag[rdx], Pr = h_pt[rdx + 1] - h_pt[rdx];
          woff[i + 1] = woff[i] + ((36 * Fr + 4 * Pr + 64 + 8 + 256 * 8 + 255) & ~(uint64_t)255);
        }
        wsc = (char*)lra_ensure(ctx, 177, woff[nbig] + 256);
        dwoff = (uint64_t*)lra_ensure(ctx, 178, ((size_t)nbig + 4) * 8);
        if (!wsc || !dwoff) return LRA_ERR_NOMEM;
        d_maxLines = (uint32_t*)(dwoff + nbig + 2);
        LRA_HIP_CHECK(ctx, hipMemcpyAsync(dwoff, woff.data(), ((size_t)nbig + 1) * 8, hipMemcpyHostToDevice, st));   // (woff lives until the stream has been waited for, below)
        if (early) {
          LRA_HIP_CHECK(ctx, hipMemsetAsync(d_maxLines, 0, 4, st));
          hipLaunchKernelGGL(k_big_lines, dim3(nbig), dim3(256), 0, st, r0, (const uint32_t*)subOrder, ptOff, (const uint32_t*)hq, (const uint32_t*)ht, (const uint32_t*)pay2, d_maxLines);
        }
      }
      hipLaunchKernelGGL(k_arena_sizes, dim3((nsub + 255) / 256), dim3(256), 0, st, nsub, r0, ptOff, cntE, cntN, cntD, ra, bytes, subOrder, shift);
      { int rc = lra_exclusive_scan<uint64_t>(ctx, nsub, bytes, byteOff); if (rc) { (void)hipStreamSynchronize(st); return rc; } }   // (the copy out of woff may still be queued)
      uint64_t totB = 0; uint32_t bigLines = 0;
      LRA_HIP_CHECK(ctx, hipMemcpyAsync(&totB, byteOff + nsub, 8, hipMemcpyDeviceToHost, st));
      if (early) LRA_HIP_CHECK(ctx, hipMemcpyAsync(&bigLines, d_maxLines, 4, hipMemcpyDeviceToHost, st));
      LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
      char* arena = (char*)lra_ensure(ctx, slot, totB + 4096);
      if (!arena) return LRA_ERR_NOMEM;
      hipLaunchKernelGGL(k_arena_bases, dim3((nsub + 255) / 256), dim3(256), 0, st, nsub, byteOff, ra, subOrder, arena);
      ProcArgs pa;
      pa.wgScratch = wsc; pa.wgOff = dwoff; pa.dbg = (dbg && nbig > 0) ? 1 : 0; { const char* e = getenv("LRA_SDP_WG_RING"); pa.wgNoRing = (e && e[0] == '0') ? 1 : 0; }
      uint64_t dbgOff0 = 0; char* dbgBase = nullptr;
      if (nbig > 0) { dbgOff0 = woff[0] + 36 * (h_frag[(uint64_t)r0 + ordAtt[0] + 1] - h_frag[(uint64_t)r0 + ordAtt[0]]) + 4 * (h_pt[(uint64_t)r0 + ordAtt[0] + 1] - h_pt[(uint64_t)r0 + ordAtt[0]]); dbgBase = wsc; }
      pa.r0 = r0; pa.n = nsub; pa.order = subOrder; pa.ptOff = ptOff; pa.fragOff = fragOff; pa.hfl = hfl; pa.hfr = hfr; pa.flen = flen; pa.fval = fval;
      pa.fprevNode = fprevNode; pa.fprevInd = fprevInd; pa.fflags = fflags; pa.rate_in = d_rate; pa.rate = opts->rate; pa.ra = ra;
      pa.status = status; pa.pwl = pw; pa.poolUsed = poolUsed; pa.penTab = d_penTab; pa.penN = penN;
      // (levels used = ceil(log2(lines)) + 1: up to 2^15 distinct rows and columns stay within levels 0..15)
      auto launch_wg = [&](hipStream_t ws, uint32_t lines) {
        if (lines <= 32768) { if (dbg) hipLaunchKernelGGL((sdp_process_wg<2, true>), dim3(nbig), dim3(64 * WG_NW), 0, ws, pa); else hipLaunchKernelGGL((sdp_process_wg<2, false>), dim3(nbig), dim3(64 * WG_NW), 0, ws, pa); }
        else { if (dbg) hipLaunchKernelGGL((sdp_process_wg<3, true>), dim3(nbig), dim3(64 * WG_NW), 0, ws, pa); else hipLaunchKernelGGL((sdp_process_wg<3, false>), dim3(nbig), dim3(64 * WG_NW), 0, ws, pa); }
      };
      if (early) LRA_HIP_CHECK(ctx, hipMemsetAsync(poolUsed, 0, (size_t)nr * 4, st));   // (in front of the fork: the workgroup launch uses its reads' pools)
      lra_time_begin(ctx, ctx->sdp_inner ? "sdp_inner_build" : "sdp_build");
      hipLaunchKernelGGL(k_visit_clear, dim3(nsub), dim3(256), 0, st, ra, byteOff, subOrder);
      ba.ra = ra; ba.order = subOrder;
      if (early) {
        hipStream_t ws = lra_side_fork(ctx);
        if (ws == st) {                                                  // (no side stream: the old order, everything on the one stream)
          hipLaunchKernelGGL((sdp_build<true, 16>), dim3(nbig), dim3(1024), 0, st, ba);
          launch_small_builds<true>(ctx, ba, subOrder, ordAtt, h_pt.data() + r0, nbig, nsub);
          launch_wg(st, bigLines);
        } else {
          hipLaunchKernelGGL((sdp_build<true, 16>), dim3(nbig), dim3(1024), 0, ws, ba);
          // the small reads' builds behind the large reads' build (an event of its own: the join event is the end of the workgroup ProcessPoint launch)
          if (!ctx->ev_mid) (void)hipEventCreateWithFlags(&ctx->ev_mid, hipEventDisableTiming);
          if (ctx->ev_mid) { (void)hipEventRecord(ctx->ev_mid, ws); (void)hipStreamWaitEvent(st, ctx->ev_mid, 0); }
          lra_time_begin(ctx, ctx->sdp_inner ? "sdp_inner_process_wg" : "sdp_process_wg", ws);
          launch_wg(ws, bigLines);
          lra_time_end(ctx, ws);
          launch_small_builds<true>(ctx, ba, subOrder, ordAtt, h_pt.data() + r0, nbig, nsub);
        }
      } else {
        if (nbig > 0) hipLaunchKernelGGL((sdp_build<true, 16>), dim3(nbig), dim3(1024), 0, forked ? lra_side_fork(ctx) : st, ba);
        if (nsub > nbig) launch_small_builds<true>(ctx, ba, subOrder, ordAtt, h_pt.data() + r0, nbig, nsub);
        if (forked) lra_side_join(ctx);
      }
      lra_time_end(ctx);
      if (onePass && att == 0) { int rc = read_rc(); if (rc) return rc; }   // (rows / columns of the reads, known after the emit pass only)
      if (att > 0)
        hipLaunchKernelGGL(k_reset_frags, dim3(nsub), dim3(64), 0, st, r0, subOrder, fragOff, flen, d_rate, opts->rate, fval, fprevNode, fprevInd, fflags, status);
      if (!early) LRA_HIP_CHECK(ctx, hipMemsetAsync(poolUsed, 0, (size_t)nr * 4, st));
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (dbg) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, st); }
      lra_time_begin(ctx, ctx->sdp_inner ? "sdp_inner_process" : "sdp_process");
      // the few large reads (a workgroup each) run beside the many small ones (a wave each) instead of in front of them
      if (nbig > 0 && !early) launch_wg(forked ? lra_side_fork(ctx) : st, maxRC);
      static const int statEnv = getenv("LRA_SDP_STAT") ? atoi(getenv("LRA_SDP_STAT")) : 0;
      unsigned long long* d_stat = nullptr;
      if (nsub > nbig) {
        ProcArgs pb = pa; pb.order = subOrder + nbig; pb.n = nsub - nbig; pb.stat = nullptr;
        if (statEnv > 0) {
          (void)hipMalloc((void**)&d_stat, 40 * 8); (void)hipMemsetAsync(d_stat, 0, 40 * 8, st);
          pb.stat = d_stat; pb.dbg = statEnv;
          hipLaunchKernelGGL(sdp_process<true>, dim3(nsub - nbig), dim3(64), 0, st, pb);
        } else hipLaunchKernelGGL(sdp_process<false>, dim3(nsub - nbig), dim3(64), 0, st, pb);
      }
      if (forked) lra_side_join(ctx);
      lra_time_end(ctx);
      if (d_stat) {
        unsigned long long hs[40];
        (void)hipStreamSynchronize(st); (void)hipMemcpy(hs, d_stat, sizeof hs, hipMemcpyDeviceToHost); (void)hipFree(d_stat);
        const double ne = (double)std::max<unsigned long long>(hs[10], 1), ns = (double)std::max<unsigned long long>(hs[11], 1);
        fprintf(stderr, "[sdp-stat] mode %d inner %d reads %d: end points %llu (switch %.2f, %.1f lanes) cycles: switch %.0f deposit %.0f sync %.0f | start points %llu (switch %.2f, %.1f lanes) cycles: switch %.0f "
                "first %.0f small %.0f coop %.0f flush+search %.0f result %.0f sync %.0f | per start point: small iters (max lane) %.2f pops %.2f with-small %.2f coop owners %.3f flush %.2f search rounds %.2f | answer's entry read %.2f, a lane searches %.2f (its Block list changed in the visit %.2f, longest list %.1f), a lane with two candidates or more %.2f\n",
                opts->mode, (int)ctx->sdp_inner, nsub - nbig, hs[10], hs[12] / ne, hs[14] / ne, hs[0] / ne, hs[2] / ne, hs[8] / ne, hs[11], hs[13] / ns, hs[15] / ns, hs[1] / ns, hs[3] / ns, hs[4] / ns, hs[5] / ns,
                hs[6] / ns, hs[7] / ns, hs[9] / ns, hs[16] / ns, hs[17] / ns, hs[18] / ns, hs[19] / ns, hs[20] / ns, hs[21] / ns, hs[22] / ns, hs[23] / ns, hs[24] / ns, hs[25] / std::max(1.0, (double)hs[23]), hs[26] / ns);
      }
      if (dbg) {
        (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        uint64_t mx = 0, tot = 0;
        for (int i = 0; i < nr; i++) { const uint64_t p = h_pt[r0 + i + 1] - h_pt[r0 + i]; mx = std::max(mx, p); tot += p; }
        fprintf(stderr, "[sdp] mode %d inner %d att %d reads %d (of %d) points total %llu max %llu  process %.1f ms\n", opts->mode, (int)ctx->sdp_inner, att, nsub, nr,
                (unsigned long long)tot, (unsigned long long)mx, ms);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        if (dbgBase && nbig > 0) {                                         // the largest read's waves: cycles in each slot and waiting at end points
          std::vector<unsigned long long> tw(256);
          const uint64_t o8 = (dbgOff0 + 7) & ~(uint64_t)7;
          (void)hipMemcpy(tw.data(), dbgBase + o8, 256 * 8, hipMemcpyDeviceToHost);
          fprintf(stderr, "[sdp]   per wave, M cycles: all | waiting  switch  deposits  publish | queries: set-up  maximization (choose, to-compare, compare+win)  flush+search  result | scan rounds  events | polls  not ready at the first\n");
          for (int w = 0; w < 16; w++) {
            const unsigned long long* o = tw.data() + 16 * w;
            fprintf(stderr, "[sdp]   wave %2d: %6.1f | %6.1f %6.1f %6.1f %6.1f | %6.1f %6.1f (%5.1f %5.1f %5.1f) %6.1f %6.1f | %6llu %6llu | %6llu %6llu\n", w, o[0] * 1e-6, o[1] * 1e-6, o[2] * 1e-6, o[3] * 1e-6, o[4] * 1e-6,
                    o[5] * 1e-6, o[6] * 1e-6, o[9] * 1e-6, o[10] * 1e-6, o[11] * 1e-6, o[7] * 1e-6, o[8] * 1e-6, o[12] >> 32, o[12] & 0xffffffffULL, o[13] >> 32, o[13] & 0xffffffffULL);
          }
        }
      }
      if (att == 2) break;
      LRA_HIP_CHECK(ctx, hipMemcpyAsync(h_status.data(), status + r0, (size_t)nr * 4, hipMemcpyDeviceToHost, st));
      LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
      h_sub.clear();
      for (int i = 0; i < nsub; i++) { const uint32_t rr = att == 0 ? h_orderAll[i] : h_prev[i]; if (h_status[rr] & LRA_ST_CAPACITY) h_sub.push_back(rr); }
      nsub = (int)h_sub.size();
      if (nsub == 0) break;
      h_prev = h_sub;
      LRA_HIP_CHECK(ctx, hipMemcpyAsync(order2, h_sub.data(), (size_t)nsub * 4, hipMemcpyHostToDevice, st));
      LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
      subOrder = order2;
    }
    if (ba.stat) {
      unsigned long long hb[16];
      (void)hipStreamSynchronize(st); (void)hipMemcpy(hb, ba.stat, sizeof hb, hipMemcpyDeviceToHost); (void)hipFree(ba.stat); ba.stat = nullptr;
      const double pts = (double)std::max<unsigned long long>(hb[8], 1);
      fprintf(stderr, "[sdp-build-stat] mode %d inner %d reads %d; the wave-per-read builds of 513 .. 16383 points, %llu points; cycles per point: set-up %.1f family %.1f | per level pass A %.1f C %.1f D %.1f E %.1f F %.1f G %.1f\n", opts->mode, (int)ctx->sdp_inner, nr,
              hb[8], hb[0] / pts, hb[7] / pts, hb[1] / pts, hb[2] / pts, hb[3] / pts, hb[4] / pts, hb[5] / pts, hb[6] / pts);
    }
    if (onePass) { int rc = read_rc(); if (rc) return rc; }                 // (entries as emitted last, re-built reads included)
    totalEntries += totE;
    const uint64_t cf0 = h_frag[r0], cfn = h_frag[r1] - h_frag[r0];
    if (cfn > 0) {
      lra_time_begin(ctx, ctx->sdp_inner ? "sdp_inner_trace" : "sdp_trace");
      // The value order (Fragment_valueOrder::Sort) is what DecidePrimaryChains walks; the single-cluster drivers (SparseDP.h:2417-2434, SparseDP_Forward.h) take the first
      // anchor of maximal value and nothing else of it: sdp_trace finds that anchor by a scan, so the keys and the exact sort are left out there (the second sparse DP of a
      // batch: 29 M values through the libstdc++-exact sort for nothing, 23 ms of the back half's chain)
      const bool needOrder = opts->mode != LRA_SDP_SINGLE_CLUSTER;
      if (needOrder) hipLaunchKernelGGL(k_valkeys, dim3((unsigned)((cfn + 255) / 256)), dim3(256), 0, st, cf0, cfn, fval, fragRead, fragOff, okey, opay);
      hipLaunchKernelGGL(k_pred, dim3((unsigned)((cfn + 255) / 256)), dim3(256), 0, st, cf0, cfn, r0, (const uint32_t*)fragRead, (const uint32_t*)fprevNode,
                         (const uint32_t*)fprevInd, (const uint32_t*)status, (const ReadArena*)ra, spare);
      lra_time_end(ctx);
      if (needOrder) { int rc = lra_sort_minimizers_batch(ctx, nr, fragOff + r0, okey, opay); if (rc) return rc; }   // Fragment_valueOrder::Sort (Fragment_Info.h:88)
      TraceArgs ta;
      ta.r0 = r0; ta.n = nr; ta.numAln = opts->NumAln; ta.single = opts->mode == LRA_SDP_SINGLE_CLUSTER; ta.alnthres = opts->alnthres; ta.fragOff = fragOff; ta.read_off = d_read_off; ta.fq = fq; ta.ft = ft;
      ta.flen = flen; ta.fcl = fcl; ta.fai = fai; ta.fval = fval; ta.fpred = spare; ta.fflags = fflags; ta.opay = opay; ta.used = used;
      ta.ra = ra; ta.nChains = nChains; ta.chainStart = chainStart; ta.chainLen = chainLen;
      ta.chainBox = chainBox; ta.chainValue = chainValue; ta.ccl = ccl; ta.can = can; ta.clink = clink; ta.status = status;
      ta.cq = cq; ta.ct = ct; ta.clen = clen; ta.cstrand = cstrand; ta.fstrand = fstrand;
      ta.boxes = boxes; ta.globalK = opts->globalK; ta.fqe = fqe; ta.fte = fte; ta.numAnchors = d_num_anchors; ta.chainNum = chainNum;
      lra_time_begin(ctx, ctx->sdp_inner ? "sdp_inner_trace" : "sdp_trace");
      hipLaunchKernelGGL(sdp_trace, dim3((nr + TRACE_LANES - 1) / TRACE_LANES), dim3(64), 0, st, ta);
      lra_time_end(ctx);
    }
    LRA_HIP_CHECK(ctx, hipGetLastError());
    r0 = r1;
  }
  out->n_subproblem_entries = totalEntries;
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  LRA_HIP_CHECK(ctx, hipGetLastError());
  return LRA_OK;
}

}  // namespace

extern "C" int lra_sparse_dp_batch(lra_ctx* ctx, int n_reads, const uint64_t* d_cluster_off, const uint64_t* d_c_start, const uint32_t* d_c_count,
                                   const int32_t* d_c_strand, const uint32_t* d_q, const uint32_t* d_t, const int32_t* d_len,
                                   const uint64_t* d_read_off, const float* d_rate, const lra_sdp_opts* opts, lra_chain_result* out) {
  if (opts && opts->mode != LRA_SDP_CLUSTERS && opts->mode != LRA_SDP_SINGLE_CLUSTER) return lra_set_err(ctx, LRA_ERR_INVALID, "mode must be 0 or 1");
  return sdp_run(ctx, n_reads, d_cluster_off, d_c_start, d_c_count, d_c_strand, d_q, d_t, d_len, d_read_off, d_rate, opts, out, nullptr, nullptr, nullptr);
}

extern "C" int lra_sparse_dp_boxes_batch(lra_ctx* ctx, int n_reads, const uint64_t* d_box_off, const uint32_t* d_qs, const uint32_t* d_qe,
                                         const uint32_t* d_ts, const uint32_t* d_te, const int32_t* d_strand, const int32_t* d_val,
                                         const int32_t* d_num_anchors, const uint64_t* d_read_off, const float* d_rate, const lra_sdp_opts* opts,
                                         lra_chain_result* out) {
  if (!opts || !d_qe || !d_te) return LRA_ERR_INVALID;
  lra_sdp_opts o = *opts;
  o.mode = LRA_SDP_CLUSTERS;                                             // chain decision of the box mode is selected by d_qe
  return sdp_run(ctx, n_reads, d_box_off, nullptr, nullptr, d_strand, d_qs, d_ts, d_val, d_read_off, d_rate, &o, out, d_qe, d_te, d_num_anchors);
}
