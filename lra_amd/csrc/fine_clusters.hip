// lra_amd/csrc/fine_clusters.hip -- SURVEY §8a row a5, the high-accuracy half: MatchesToFineClusters (Clustering.h:1555-1680) behind
// lra_clean_matches_batch (DiagonalSort / AntiDiagonalSort + CleanOffDiagonal: the rough clusters).  gfx950 only.
//   CartesianSort of every rough cluster          Sorting.h:157            one segmented radix sort of (q << 32 | t) (distinct pairs: one order)
//   SplitRoughClustersWithGaps                    Clustering.h:1358-1432   (CloseToPreviousCluster :1333, MergeTwoClusters :1351)
//   StoreFineClusters                             Clustering.h:892-1331    (DiagonalDifference :503, minGapDifference :532,
//                                                                           SetClusterBoundariesFromMatches :308, Cluster::CHROMIndex :327)
// Mapping.  The two functions are serial state machines over a read's few hundred matches whose output order and `pop_back` pairs
// (:1301-1306 may drop the cluster pushed BEFORE the current one, of either strand) tie all rough clusters of a read together: one lane per
// read (16 active lanes per wave, like the other one-lane-per-read kernels) walks strand 0 then strand 1 -- split table and index lists in a
// per-read slab sized by its cleaned matches (a match belongs to at most one split cluster and to at most one fine cluster, so the read's own
// match range is capacity enough for every list) -- then one scan + one gather make the clusters CSR over the batch.
// Algorithmic bytes: 8 B per cleaned match in, 8 B per fine-cluster match out, 28 B per cluster.
#include "common.h"
#include "scan.h"
#include <rocprim/rocprim.hpp>

namespace {

__device__ int fc_header_find(const uint64_t* pos, int npos, uint64_t query) {   // Genome.h:20-32
  if (npos > 0 && query == pos[0]) return 0;
  int lo = 0, hi = npos;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (pos[mid] < query) lo = mid + 1; else hi = mid; }
  if (lo < npos && pos[lo] == query) return lo;
  return lo - 1;
}

struct FcArgs {
  int n_reads; const uint64_t* cluster_off; const uint64_t* c_start; const uint64_t* c_end; const int32_t* c_strand; const float* c_freq;
  const uint32_t* c_box;             // rough boxes qStart, qEnd, tStart, tEnd as 4 arrays of n_clusters
  const uint32_t* bq0; const uint32_t* bq1; const uint32_t* bt0; const uint32_t* bt1;
  const uint64_t* key;               // sorted (q << 32 | t) per rough cluster, at the cleaned-match positions
  const uint64_t* read_moff;         // [n_reads+1] first cleaned match of every read
  const uint64_t* pos; int npos;
  lra_fine_opts o;
  // per-read slabs at read_moff[r]: 8 int32 work lists, the split index list, the output matches; tables with one slot per match
  int32_t* work; uint32_t* idx; uint32_t* oq; uint32_t* ot;
  int32_t* spOff; int32_t* spLen; uint32_t* spT; float* spFreq; uint32_t* spBox;   // split table (spBox: qStart qEnd tStart tEnd)
  uint32_t* clOff; uint32_t* clLen; uint32_t* clBox; int32_t* clStrand; int32_t* clChrom; float* clFreq;   // fine clusters of the read, relative to its slab
  uint32_t* nCl; uint32_t* nMt; uint32_t* status;
};

#define MQ(i) ((uint32_t)(a.key[m0 + (i)] >> 32))
#define MT_(i) ((uint32_t)(a.key[m0 + (i)] & 0xFFFFFFFFu))

__device__ __forceinline__ long fc_diag_diff(uint32_t aq, uint32_t at, uint32_t bq, uint32_t bt, int strand) {
  if (strand == 0) return ((long)at - (long)aq) - ((long)bt - (long)bq);
  return (long)(uint32_t)(aq + at) - (long)(uint32_t)(bq + bt);
}
__device__ __forceinline__ long fc_labs(long x) { return x < 0 ? -x : x; }
__device__ __forceinline__ long fc_min_gap(uint32_t aq, uint32_t at, uint32_t bq, uint32_t bt) {
  return min(fc_labs((long)bq - (long)aq), fc_labs((long)bt - (long)at));
}

constexpr int FC_LANES = 16;

__global__ void __launch_bounds__(64) fc_kernel(FcArgs a) {
  if (threadIdx.x >= FC_LANES) return;
  const int r = blockIdx.x * FC_LANES + threadIdx.x;
  if (r >= a.n_reads) return;
  const uint64_t m0 = a.read_moff[r];
  const int N = (int)(a.read_moff[r + 1] - m0);
  const lra_fine_opts o = a.o;
  const int K = o.globalK;
  int32_t* W = a.work + 8 * m0;
  int32_t* match_num = W; int32_t* pos_start = W + N; int32_t* Start = W + 2 * (size_t)N; int32_t* End = W + 3 * (size_t)N; int32_t* AddOrNot = W + 4 * (size_t)N;
  int32_t* Stretch = W + 5 * (size_t)N; int32_t* CIdx = W + 6 * (size_t)N;
  uint32_t* idx = a.idx + m0;
  uint32_t* oq = a.oq + m0; uint32_t* ot = a.ot + m0;
  int32_t* spOff = a.spOff + m0; int32_t* spLen = a.spLen + m0; uint32_t* spT = a.spT + m0; float* spFreq = a.spFreq + m0; uint32_t* spBox = a.spBox + 4 * m0;
  uint32_t* clOff = a.clOff + m0; uint32_t* clLen = a.clLen + m0; uint32_t* clBox = a.clBox + 4 * m0; int32_t* clStrand = a.clStrand + m0; int32_t* clChrom = a.clChrom + m0;
  float* clFreq = a.clFreq + m0;
  int ncl = 0; uint32_t nout = 0; uint32_t st = 0;
  // a cluster under construction = clusters.back(): matches oq/ot[clOff[ncl-1] .. nout)
  auto set_bounds = [&](int c) {                                          // :308-322
    const uint32_t b = clOff[c], e = b + clLen[c];
    uint32_t qS = oq[b], qE = qS + K, tS = ot[b], tE = tS + K;
    for (uint32_t i = b + 1; i < e; i++) { tE = max(tE, ot[i] + (uint32_t)K); tS = min(tS, ot[i]); qE = max(qE, oq[i] + (uint32_t)K); qS = min(qS, oq[i]); }
    clBox[4 * c] = qS; clBox[4 * c + 1] = qE; clBox[4 * c + 2] = tS; clBox[4 * c + 3] = tE;
  };
  auto chrom_index = [&](int c) -> bool {                                 // Cluster::CHROMIndex :327-337
    if (clLen[c] == 0) return true;
    const int x = fc_header_find(a.pos, a.npos, (uint64_t)clBox[4 * c + 2] + 1), y = fc_header_find(a.pos, a.npos, clBox[4 * c + 3]);
    if (x != y) return true;
    clChrom[c] = x;
    return false;
  };
  auto pop_back = [&]() { ncl--; nout = clOff[ncl]; };
  auto push_match = [&](uint32_t q, uint32_t t) { oq[nout] = q; ot[nout] = t; nout++; clLen[ncl - 1]++; };
  auto push_cluster = [&](int strand) { clOff[ncl] = nout; clLen[ncl] = 0; clStrand[ncl] = strand; clChrom[ncl] = 0; clFreq[ncl] = 0; ncl++; };
  for (int strand = 0; strand < 2 && !st; strand++) {
    // ---- SplitRoughClustersWithGaps over this strand's rough clusters, in order
    int nsp = 0; uint32_t nidx = 0;
    for (uint64_t c = a.cluster_off[r]; c < a.cluster_off[r + 1]; c++) {
      if (a.c_strand[c] != strand) continue;
      const int s0 = (int)(a.c_start[c] - m0), e0 = (int)(a.c_end[c] - m0);
      if (e0 - s0 == 0) continue;
      const float freq = a.c_freq[c];
      if (freq >= 10.0f) {                                                // :1364-1370
        spOff[nsp] = (int)nidx; spLen[nsp] = e0 - s0; spT[nsp] = a.bt0[c]; spFreq[nsp] = freq;
        spBox[4 * nsp] = a.bq0[c]; spBox[4 * nsp + 1] = a.bq1[c]; spBox[4 * nsp + 2] = a.bt0[c]; spBox[4 * nsp + 3] = a.bt1[c];
        for (int q = s0; q < e0; q++) idx[nidx++] = (uint32_t)q;
        nsp++;
        continue;
      }
      const int cur_s = nsp;
      int split_cs = s0;
      uint32_t sq0 = MQ(split_cs), st0 = MT_(split_cs), sq1 = sq0 + K, st1 = st0 + K;
      auto merge_or_push = [&](int e) {
        bool close = false;
        if (nsp > cur_s) {                                                // CloseToPreviousCluster :1333-1343
          const uint32_t* B = spBox + 4 * (nsp - 1);
          const long aDiff = fc_labs((long)sq0 - (long)B[1]);
          const long bDiff = strand == 0 ? fc_labs((long)st0 - (long)B[3]) : fc_labs((long)B[2] - (long)st1);
          long aDiag, bDiag;
          if (strand == 0) { aDiag = (long)B[3] - (long)B[1]; bDiag = (long)st0 - (long)sq0; }
          else { aDiag = (long)B[1] + (long)B[2]; bDiag = (long)sq0 + (long)st1; }
          close = min(aDiff, bDiff) <= o.RoughClustermaxGap && fc_labs(aDiag - bDiag) < o.maxDiag;
        }
        if (close) {                                                      // MergeTwoClusters :1351-1355
          uint32_t* B = spBox + 4 * (nsp - 1);
          B[0] = min(B[0], sq0); B[1] = max(B[1], sq1); B[2] = min(B[2], st0); B[3] = max(B[3], st1);
          for (int q = split_cs; q < e; q++) idx[nidx++] = (uint32_t)q;
          spLen[nsp - 1] += e - split_cs;
          spT[nsp - 1] = B[2];
        } else {
          spOff[nsp] = (int)nidx; spLen[nsp] = e - split_cs; spT[nsp] = st0; spFreq[nsp] = freq;
          spBox[4 * nsp] = sq0; spBox[4 * nsp + 1] = sq1; spBox[4 * nsp + 2] = st0; spBox[4 * nsp + 3] = st1;
          for (int q = split_cs; q < e; q++) idx[nidx++] = (uint32_t)q;
          nsp++;
        }
      };
      for (int i = s0 + 1; i < e0; i++) {
        const long gap = fc_min_gap(MQ(i), MT_(i), MQ(i - 1), MT_(i - 1));
        if (gap > o.RoughClustermaxGap) {
          if (i - split_cs >= o.minClusterSize) merge_or_push(i);
          sq0 = MQ(i); st0 = MT_(i); sq1 = sq0 + K; st1 = st0 + K; split_cs = i;
        } else {
          sq0 = min(sq0, MQ(i)); st0 = min(st0, MT_(i)); sq1 = max(sq1, MQ(i) + (uint32_t)K); st1 = max(st1, MT_(i) + (uint32_t)K);
        }
      }
      if (e0 - split_cs >= o.minClusterSize) merge_or_push(e0);
    }
    // ---- StoreFineClusters for every split cluster of the strand
    for (int sc = 0; sc < nsp && !st; sc++) {
      const uint32_t* smi = idx + spOff[sc];
      const int n = spLen[sc];
      const float anchorfreq = spFreq[sc];
      const int ri = fc_header_find(a.pos, a.npos, spT[sc]);
#define SQ(i) MQ(smi[i])
#define ST(i) MT_(smi[i])
      if (n == 1) continue;
      if (fabsf(anchorfreq - 1.0f) <= 0.005) {                            // :900-942 (the reference compares in double: same decision for these values)
        push_cluster(strand);
        for (int i = 0; i < n; i++) push_match(SQ(i), ST(i));
        set_bounds(ncl - 1);
        clChrom[ncl - 1] = ri; clFreq[ncl - 1] = 1.0f;
        if (chrom_index(ncl - 1)) pop_back();
        continue;
      }
      int nm = 0, oc = 1, us = 0;                                          // :948-965
      for (int i = 1; i < n; i++) {
        if (SQ(i) == SQ(i - 1)) oc++;
        else { match_num[nm] = oc; pos_start[nm] = us; nm++; us = i; oc = 1; }
        if (i == n - 1) { match_num[nm] = oc; pos_start[nm] = us; nm++; }
      }
      int u_start = 0, u_end = 0, u_maxstart = 0, u_maxend = 0, max_pos = 0, ns = 0;
      if (nm == 1) { u_maxstart = 0; u_maxend = 1; Start[0] = 0; End[0] = 1; ns = 1; }
      else {
        int k = 0;
        while (k < nm - 1) {
          while (k < nm - 1 && match_num[k] != 1) k++;
          u_start = k; u_end = k + 1;
          while (k < nm - 1 && match_num[k + 1] == match_num[k] &&
                 fc_labs(fc_diag_diff(SQ(pos_start[k + 1]), ST(pos_start[k + 1]), SQ(pos_start[k]), ST(pos_start[k]), strand)) < o.maxDiag &&
                 fc_min_gap(SQ(pos_start[k + 1]), ST(pos_start[k + 1]), SQ(pos_start[k]), ST(pos_start[k])) <= o.maxGap) { u_end = k + 2; k++; }
          Start[ns] = u_start; End[ns] = u_end; ns++;
          k++;
          if ((u_maxstart == 0 && u_maxend == 0) || (u_maxend - u_maxstart < u_end - u_start)) { u_maxstart = u_start; u_maxend = u_end; max_pos = ns - 1; }
        }
      }
      if (u_maxstart == 0 && u_maxend == 0) continue;
      int c_s = pos_start[u_maxstart], c_e = pos_start[u_maxend - 1] + 1;
      if (!(c_e - c_s >= o.minUniqueStretchNum && (long)SQ(c_e - 1) + K - (long)SQ(c_s) >= o.minUniqueStretchDist)) continue;
      push_cluster(strand);
      for (int i = 0; i < ns; i++) AddOrNot[i] = 0;
      if (c_e - c_s == n) {
        for (int i = c_s; i < c_e; i++) push_match(SQ(i), ST(i));
        clFreq[ncl - 1] = anchorfreq;
        AddOrNot[0] = 1;
      } else {
        // StretchOfOne as an array in ASCENDING stretch order (the reference's list read backwards): backward picks are collected in
        // descending order in Stretch[0..nb), forward picks in ascending order in CIdx[0..nf) for a moment, then laid out
        int nb = 0, nf = 0;
        auto near_ = [&](int i_m, int pa) {
          const long g = fc_min_gap(SQ(i_m), ST(i_m), SQ(pa), ST(pa));
          return (fc_labs(fc_diag_diff(SQ(i_m), ST(i_m), SQ(pa), ST(pa), strand)) <= o.maxDiag && g <= o.maxGap) || g <= o.maxGap / 2;
        };
        int prev_anchor = c_s;
        Stretch[nb++] = max_pos; AddOrNot[max_pos] = 1;
        for (int i = max_pos - 1; i >= 0; i--) {
          const int i_m = pos_start[End[i] - 1];
          if (near_(i_m, prev_anchor)) { Stretch[nb++] = i; AddOrNot[i] = 1; prev_anchor = pos_start[Start[i]]; }
        }
        prev_anchor = c_e - 1;
        for (int i = max_pos + 1; i < ns; i++) {
          const int i_m = pos_start[Start[i]];
          if (near_(i_m, prev_anchor)) { CIdx[nf++] = i; AddOrNot[i] = 1; prev_anchor = pos_start[End[i] - 1]; }
        }
        // ascending: reverse of Stretch[0..nb) followed by CIdx[0..nf); keep it in Stretch
        for (int x = 0; x < nb / 2; x++) { const int t = Stretch[x]; Stretch[x] = Stretch[nb - 1 - x]; Stretch[nb - 1 - x] = t; }
        for (int x = 0; x < nf; x++) Stretch[nb + x] = CIdx[x];
        const int nst = nb + nf;
        int prev_stretch = -1, p_s = 0, p_e = 0;
        for (int z = 0; z < nst; z++) {                                    // :1103-1163
          const int it = Stretch[z];
          c_s = pos_start[Start[it]]; c_e = pos_start[End[it] - 1] + 1;
          if (z == 0) { p_s = it == 0 ? 0 : pos_start[End[it - 1]]; p_e = pos_start[Start[it]]; }
          else { p_s = pos_start[End[prev_stretch]]; p_e = pos_start[Start[it]]; }
          prev_stretch = it;
          int prev_match = c_s, nci = 0;
          for (int si = p_e - 1; si >= p_s; si--)
            if (fc_labs(fc_diag_diff(SQ(si), ST(si), SQ(prev_match), ST(prev_match), strand)) < o.maxDiag) { CIdx[nci++] = si; prev_match = si; }
          for (int x = nci - 1; x >= 0; x--) push_match(SQ(CIdx[x]), ST(CIdx[x]));
          for (int si = c_s; si < c_e; si++) push_match(SQ(si), ST(si));
          if (z == nst - 1) {
            p_s = pos_start[End[it] - 1] + 1;
            p_e = (it == ns - 1) ? n : pos_start[Start[it + 1]];
            prev_match = c_e - 1;
            for (int si = p_s; si < p_e; si++)
              if (fc_labs(fc_diag_diff(SQ(si), ST(si), SQ(prev_match), ST(prev_match), strand)) < o.maxDiag) { push_match(SQ(si), ST(si)); prev_match = si; }
          }
        }
        clFreq[ncl - 1] = anchorfreq;
      }
      set_bounds(ncl - 1);
      clChrom[ncl - 1] = ri;
      {
        const int c = ncl - 1;
        const long qs = clBox[4 * c], qe = clBox[4 * c + 1], ts = clBox[4 * c + 2], te = clBox[4 * c + 3];
        if (chrom_index(c)) pop_back();
        else if ((long)clLen[c] <= o.minClusterSize) pop_back();
        else if (qe == qs) pop_back();
        else if (te - ts >= 5 * (qe - qs)) pop_back();
      }
      for (int ar = 0; ar < ns && !st; ar++) {                             // :1297-1323
        if (!AddOrNot[ar] && End[ar] - Start[ar] >= 15) {
          push_cluster(strand);
          for (int i = pos_start[Start[ar]]; i < pos_start[End[ar] - 1] + 1; i++) push_match(SQ(i), ST(i));
          set_bounds(ncl - 1);
          clChrom[ncl - 1] = ri; clFreq[ncl - 1] = anchorfreq;
          if (chrom_index(ncl - 1)) pop_back();
          if (ncl == 0) { st |= LRA_ST_OOB_SLOT; break; }                  // the reference reads clusters.back() of an empty vector
          const int c = ncl - 1;
          const long qs = clBox[4 * c], qe = clBox[4 * c + 1], ts = clBox[4 * c + 2], te = clBox[4 * c + 3];
          if (qe - qs == 0) { st |= LRA_ST_OOB_SLOT; break; }
          if ((te - ts) / (qe - qs) >= 5) pop_back();
        }
      }
#undef SQ
#undef ST
    }
  }
  a.nCl[r] = st ? 0 : (uint32_t)ncl; a.nMt[r] = st ? 0 : nout; a.status[r] = st;
}

__global__ void fc_keys(uint64_t n, const uint32_t* __restrict__ q, const uint32_t* __restrict__ t, uint64_t* key) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) key[i] = ((uint64_t)q[i] << 32) | t[i];
}
__global__ void fc_read_moff(int n_reads, uint64_t n_matches, const uint64_t* __restrict__ cluster_off, const uint64_t* __restrict__ c_start, uint64_t n_clusters,
                             uint64_t* moff) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n_reads) return;
  const uint64_t c = cluster_off[r];
  moff[r] = c < n_clusters ? c_start[c] : n_matches;                       // the cleaned matches are laid out read by read, cluster by cluster
}
// dense output: clusters CSR by read, matches CSR by cluster
__global__ void __launch_bounds__(64) fc_gather(FcArgs a, const uint64_t* __restrict__ clBase, const uint64_t* __restrict__ mtBase, uint64_t* outOff, uint32_t* outQ,
                                                uint32_t* outT, uint32_t* outBox, int32_t* outStrand, int32_t* outChrom, float* outFreq) {
  const int r = blockIdx.x;
  if (r >= a.n_reads) return;
  const uint64_t m0 = a.read_moff[r];
  const uint32_t ncl = a.nCl[r];
  const uint64_t cb = clBase[r], mb = mtBase[r];
  for (uint32_t c = threadIdx.x; c < ncl; c += 64) {
    outOff[cb + c] = mb + a.clOff[m0 + c];
    for (int k = 0; k < 4; k++) outBox[4 * (cb + c) + k] = a.clBox[4 * (m0 + c) + k];
    outStrand[cb + c] = a.clStrand[m0 + c]; outChrom[cb + c] = a.clChrom[m0 + c]; outFreq[cb + c] = a.clFreq[m0 + c];
  }
  for (uint32_t i = threadIdx.x; i < a.nMt[r]; i += 64) { outQ[mb + i] = a.oq[m0 + i]; outT[mb + i] = a.ot[m0 + i]; }
}

inline size_t sz(size_t n, size_t e) { return (n * e + 255) / 256 * 256; }

}  // namespace

extern "C" int lra_fine_clusters_batch(lra_ctx* ctx, const lra_cluster_result* rough, const lra_fine_opts* opts, const uint64_t* h_chrom_pos, int n_chrom,
                                       lra_fine_result* out) {
  if (!ctx || !rough || !opts || !out || !h_chrom_pos || n_chrom < 1) return LRA_ERR_INVALID;
  memset(out, 0, sizeof *out);
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int nR = rough->n_reads;
  const uint64_t NC = rough->n_clusters, NM = rough->n_matches;
  out->n_reads = nR;
  const size_t npos = (size_t)n_chrom + 1;
  char* w = (char*)lra_ensure(ctx, 86, sz(NM + 1, 8) * 2 + sz(nR + 2, 8) * 3 + sz(npos, 8) + sz(8 * NM + 8, 4) + sz(NM + 1, 4) * 12 + sz(4 * NM + 4, 4) * 2 + sz(nR + 1, 4) * 3 + 8192);
  if (!w) return LRA_ERR_NOMEM;
  auto take = [&](size_t n, size_t e) { char* r = w; w += sz(n, e); return r; };
  uint64_t* key = (uint64_t*)take(NM + 1, 8); uint64_t* key2 = (uint64_t*)take(NM + 1, 8);
  uint64_t* moff = (uint64_t*)take(nR + 2, 8); uint64_t* clBase = (uint64_t*)take(nR + 2, 8); uint64_t* mtBase = (uint64_t*)take(nR + 2, 8);
  uint64_t* dpos = (uint64_t*)take(npos, 8);
  FcArgs a; memset(&a, 0, sizeof a);
  a.work = (int32_t*)take(8 * NM + 8, 4);
  a.idx = (uint32_t*)take(NM + 1, 4); a.oq = (uint32_t*)take(NM + 1, 4); a.ot = (uint32_t*)take(NM + 1, 4);
  a.spOff = (int32_t*)take(NM + 1, 4); a.spLen = (int32_t*)take(NM + 1, 4); a.spT = (uint32_t*)take(NM + 1, 4); a.spFreq = (float*)take(NM + 1, 4);
  a.clOff = (uint32_t*)take(NM + 1, 4); a.clLen = (uint32_t*)take(NM + 1, 4); a.clStrand = (int32_t*)take(NM + 1, 4); a.clChrom = (int32_t*)take(NM + 1, 4);
  a.clFreq = (float*)take(NM + 1, 4);
  a.spBox = (uint32_t*)take(4 * NM + 4, 4); a.clBox = (uint32_t*)take(4 * NM + 4, 4);
  a.nCl = (uint32_t*)take(nR + 1, 4); a.nMt = (uint32_t*)take(nR + 1, 4); a.status = (uint32_t*)take(nR + 1, 4);
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(dpos, h_chrom_pos, npos * 8, hipMemcpyHostToDevice, st));
  if (nR == 0) return LRA_OK;
  auto grid = [](uint64_t n) { return dim3((unsigned)((n + 255) / 256)); };
  lra_time_begin(ctx, "fine_clusters");
  hipLaunchKernelGGL(fc_read_moff, grid((uint64_t)nR + 1), dim3(256), 0, st, nR, NM, rough->d_cluster_off, rough->d_c_start, NC, moff);
  const uint64_t* skey = key;
  if (NM && NC) {
    hipLaunchKernelGGL(fc_keys, grid(NM), dim3(256), 0, st, NM, rough->d_cl_qpos, rough->d_cl_tpos, key);
    size_t tb = 0;
    (void)rocprim::segmented_radix_sort_keys(nullptr, tb, key, key2, (unsigned int)NM, (unsigned int)NC, rough->d_c_start, rough->d_c_end, 0, 64, st);
    void* tmp = lra_scratch(ctx, 2, tb + 256);
    if (!tmp) { lra_time_end(ctx); return LRA_ERR_NOMEM; }
    // (matches outside every rough cluster do not exist: the clusters tile the cleaned matches; positions no segment covers are copied through)
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(key2, key, NM * 8, hipMemcpyDeviceToDevice, st));
    hipError_t e = rocprim::segmented_radix_sort_keys(tmp, tb, key, key2, (unsigned int)NM, (unsigned int)NC, rough->d_c_start, rough->d_c_end, 0, 64, st);
    if (e != hipSuccess) { lra_time_end(ctx); return lra_set_err(ctx, LRA_ERR_HIP, "segmented sort: %s", hipGetErrorString(e)); }
    skey = key2;
  }
  a.n_reads = nR; a.cluster_off = rough->d_cluster_off; a.c_start = rough->d_c_start; a.c_end = rough->d_c_end; a.c_strand = rough->d_c_strand; a.c_freq = rough->d_c_anchorfreq;
  a.bq0 = rough->d_c_qStart; a.bq1 = rough->d_c_qEnd; a.bt0 = rough->d_c_tStart; a.bt1 = rough->d_c_tEnd;
  a.key = skey; a.read_moff = moff; a.pos = dpos; a.npos = (int)npos; a.o = *opts;
  hipLaunchKernelGGL(fc_kernel, dim3((nR + FC_LANES - 1) / FC_LANES), dim3(64), 0, st, a);
  lra_time_end(ctx);
  int rc;
  if ((rc = lra_exclusive_scan<uint32_t>(ctx, nR, a.nCl, clBase)) || (rc = lra_exclusive_scan<uint32_t>(ctx, nR, a.nMt, mtBase))) return rc;
  uint64_t nCl = 0, nMt = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&nCl, clBase + nR, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&nMt, mtBase + nR, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  char* wo = (char*)lra_ensure(ctx, 87, sz(nCl + 2, 8) + sz(nMt + 1, 4) * 2 + sz(4 * nCl + 4, 4) + sz(nCl + 1, 4) * 3 + 4096);
  if (!wo) return LRA_ERR_NOMEM;
  auto take2 = [&](size_t n, size_t e) { char* r = wo; wo += sz(n, e); return r; };
  uint64_t* outOff = (uint64_t*)take2(nCl + 2, 8); uint32_t* outQ = (uint32_t*)take2(nMt + 1, 4); uint32_t* outT = (uint32_t*)take2(nMt + 1, 4);
  uint32_t* outBox = (uint32_t*)take2(4 * nCl + 4, 4); int32_t* outStrand = (int32_t*)take2(nCl + 1, 4); int32_t* outChrom = (int32_t*)take2(nCl + 1, 4);
  float* outFreq = (float*)take2(nCl + 1, 4);
  hipLaunchKernelGGL(fc_gather, dim3(nR), dim3(64), 0, st, a, (const uint64_t*)clBase, (const uint64_t*)mtBase, outOff, outQ, outT, outBox, outStrand, outChrom, outFreq);
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(outOff + nCl, &nMt, 8, hipMemcpyHostToDevice, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  LRA_HIP_CHECK(ctx, hipGetLastError());
  out->n_clusters = nCl; out->n_matches = nMt; out->d_cluster_off = clBase; out->d_match_off = outOff; out->d_q = outQ; out->d_t = outT; out->d_box = outBox;
  out->d_strand = outStrand; out->d_chrom = outChrom; out->d_anchorfreq = outFreq; out->d_status = a.status;
  return LRA_OK;
}
