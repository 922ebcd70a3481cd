// lra_amd/csrc/linear_extend_clusters.hip -- SURVEY §8a row a7, the cluster version of LinearExtend (high-accuracy path).  gfx950 only.
//   LinearExtend(vector<Cluster*> clusters, vector<Cluster>& extCluster, vector<Tup>& chain, ..., skiprepetitive, K)   LinearExtend.h:136-352
//   CheckOverlap :88-101, Checkbp :50-85, DecideCoordinates :105-128, LinearExtend_chain :783-792 (+ TrimOverlappedAnchors :574-649, already
//   lra_trim_overlapped_anchors_batch), called at Map_highacc.h:580 for every chain.
// Mapping.  (1) DiagonalSort / AntiDiagonalSort of every refined cluster's matches (:201-210, in place as in the reference) = one segmented radix
// sort of ((diagonal key) << 32 | q) with t as payload.  (2) One lane per chain element (16 active lanes per wave): the walk over the sorted matches
// with its m / n jumps at anchors that touch the neighbours' box coordinates is serial; an element emits at most as many anchors as its cluster
// has matches, so it writes into a slab of that size and the slabs are compacted afterwards.  (3) TrimOverlappedAnchors on the compacted lists.
// Algorithmic bytes: 8 B per match in (+ the bases Checkbp compares), 13 B per anchor out.
#include "common.h"
#include "scan.h"
#include <rocprim/rocprim.hpp>

namespace {

struct LecArgs {
  uint64_t n_items; const uint32_t* itemCluster; const int32_t* itemPrev; const int32_t* itemNext; const uint32_t* itemRead;
  const uint64_t* matchOff; const uint32_t* mq; const uint32_t* mt; const uint32_t* box; const int32_t* strand; const int32_t* chrom; const float* freq;
  const unsigned char* seq; const uint64_t* read_off; const unsigned char* genome; const uint64_t* pos;
  int skiprepetitive, K;
  const uint64_t* slabOff;       // [n_items+1] prefix of the clusters' match counts
  uint32_t* eq; uint32_t* et; int32_t* elen; uint8_t* eovl; uint32_t* cnt; uint32_t* novl; uint32_t* obox; int32_t* ostrand; int32_t* ochrom; float* ofreq;
};

constexpr int LEC_LANES = 16;

__global__ void __launch_bounds__(64) lec_kernel(LecArgs a) {
  if (threadIdx.x >= LEC_LANES) return;
  const uint64_t it = (uint64_t)blockIdx.x * LEC_LANES + threadIdx.x;
  if (it >= a.n_items) return;
  const uint32_t cm = a.itemCluster[it];
  const uint64_t m0 = a.matchOff[cm];
  const long n = (long)(a.matchOff[cm + 1] - m0);
  const uint64_t o0 = a.slabOff[it];
  uint32_t* eq = a.eq + o0; uint32_t* et = a.et + o0; int32_t* elen = a.elen + o0; uint8_t* eovl = a.eovl + o0;
  // extCluster[c + start] keeps the Cluster() defaults when the refined cluster is empty (:145): strand -1, nothing else set
  a.ostrand[it] = -1; a.ochrom[it] = 0; a.ofreq[it] = 0; a.cnt[it] = 0; a.novl[it] = 0;
  a.obox[4 * it] = 0; a.obox[4 * it + 1] = 0; a.obox[4 * it + 2] = 0; a.obox[4 * it + 3] = 0;
  if (n == 0) return;
  const int K = a.K, strand = a.strand[cm], ci = a.chrom[cm];
  const float freq = a.freq[cm];
  const uint32_t* q = a.mq + m0; const uint32_t* t = a.mt + m0;
  const uint32_t r = a.itemRead[it];
  const unsigned char* read = a.seq + a.read_off[r];
  const uint32_t readLen = (uint32_t)(a.read_off[r + 1] - a.read_off[r]);
  const unsigned char* chr = a.genome + a.pos[ci];
  const uint32_t chromLen = (uint32_t)(a.pos[ci + 1] - a.pos[ci]);
  uint32_t setV[8]; int setT[8]; int ns = 0;
  const uint32_t qsb = a.box[4 * cm], qeb = a.box[4 * cm + 1], tsb = a.box[4 * cm + 2], teb = a.box[4 * cm + 3];
  if (a.skiprepetitive && freq <= 1.1f) {                                 // :161-192
    for (int side = 0; side < 2; side++) {
      const int nb = side == 0 ? a.itemPrev[it] : a.itemNext[it];
      if (nb < 0) continue;
      const uint32_t* B = a.box + 4 * (size_t)nb;
      if (B[0] > qsb && B[0] < qeb) { setV[ns] = B[0]; setT[ns++] = 0; }
      if (B[1] > qsb && B[1] < qeb) { setV[ns] = B[1]; setT[ns++] = 0; }
      if (B[2] > tsb && B[2] < teb) { setV[ns] = B[2]; setT[ns++] = 1; }
      if (B[3] > tsb && B[3] < teb) { setV[ns] = B[3]; setT[ns++] = 1; }
    }
  }
  auto ovp = [&](long i) -> bool {                                        // CheckOverlap :88-101
    for (int s = 0; s < ns; s++) {
      if (setT[s] == 0 && setV[s] >= q[i] && setV[s] < q[i] + (uint32_t)K) return true;
      if (setT[s] == 1 && setV[s] >= t[i] && setV[s] < t[i] + (uint32_t)K) return true;
    }
    return false;
  };
  uint32_t ne = 0, nov = 0;
  auto push = [&](uint32_t x, uint32_t y, int l, int o) { eq[ne] = x; et[ne] = y; elen[ne] = l; eovl[ne] = (uint8_t)o; ne++; nov += o; };
  long i = 1, m = 0;
  bool chm = true;
  while (i < n) {                                                         // :218-329
    if (chm) {
      if (ovp(m)) { push(q[m], t[m], K, 1); m = i; i++; chm = true; continue; }
      chm = false;
    }
    if (ovp(i)) {
      push(q[m], strand == 0 ? t[m] : t[i - 1], (int)(q[i - 1] + K - q[m]), 0);
      push(q[i], t[i], K, 1);
      m = i + 1; i = m + 1; chm = true;
      continue;
    }
    long curDiag, nextDiag;
    if (strand == 0) { curDiag = (long)q[i - 1] - (long)t[i - 1]; nextDiag = (long)q[i] - (long)t[i]; }
    else { curDiag = (long)q[i - 1] + (long)t[i - 1]; nextDiag = (long)q[i] + (long)t[i]; }
    if (curDiag == nextDiag) {
      if (q[i] < q[i - 1] + (uint32_t)K) i++;
      else {
        uint32_t curQ = q[i - 1] + K, curT;                               // Checkbp :50-85
        const uint32_t nextQ = q[i];
        if (strand == 0) {
          curT = min(chromLen, t[i - 1] + (uint32_t)K);
          const uint32_t nextT = min(chromLen, t[i]);
          while (curQ < readLen && curT < chromLen && nextQ > curQ && nextT > curT && chr[curT] == read[curQ]) { curQ++; curT++; }
        } else {
          curT = min(chromLen - 1, t[i - 1] - 1);
          const uint32_t nextT = min(chromLen - 1, t[i] + (uint32_t)K - 1);
          while (curQ < readLen && nextQ > curQ && nextT < curT && chr[curT] == read[curQ]) { curQ++; curT--; }
        }
        const uint32_t qe = curQ, te = curT;
        if (strand == 0 && qe == q[i] && te == t[i]) i++;
        else if (strand == 1 && qe == q[i] && te == t[i] + (uint32_t)K - 1) i++;
        else { push(q[m], strand == 0 ? t[m] : te + 1, (int)(qe - q[m]), 0); m = i; i++; }
      }
    } else { push(q[m], strand == 0 ? t[m] : t[i - 1], (int)(q[i - 1] + K - q[m]), 0); m = i; i++; }
    chm = false;
  }
  if (i == n) push(q[m], strand == 0 ? t[m] : t[i - 1], (int)(q[i - 1] + K - q[m]), 0);
  a.cnt[it] = ne; a.novl[it] = nov;
  a.ochrom[it] = ci; a.ofreq[it] = freq;                                   // :157-158
  if (ne) {                                                               // DecideCoordinates :105-128
    uint32_t qS = eq[0], qE = eq[0] + elen[0], tS = et[0], tE = et[0] + elen[0];
    for (uint32_t x = 1; x < ne; x++) { qS = min(qS, eq[x]); qE = max(qE, eq[x] + (uint32_t)elen[x]); tS = min(tS, et[x]); tE = max(tE, et[x] + (uint32_t)elen[x]); }
    a.obox[4 * it] = qS; a.obox[4 * it + 1] = qE; a.obox[4 * it + 2] = tS; a.obox[4 * it + 3] = tE;
    a.ostrand[it] = strand;
  }
}

__global__ void lec_keys(uint64_t ncl, const uint64_t* __restrict__ off, const int32_t* __restrict__ strand, const uint32_t* __restrict__ q, const uint32_t* __restrict__ t,
                         uint64_t* key) {
  // one wave per cluster
  for (uint64_t c = blockIdx.x; c < ncl; c += gridDim.x) {
    const int s = strand[c];
    for (uint64_t i = off[c] + threadIdx.x; i < off[c + 1]; i += blockDim.x) {
      const uint32_t qq = q[i], tt = t[i];
      const uint64_t d = s == 0 ? (uint64_t)((long long)qq - (long long)tt + (1LL << 32)) : (uint64_t)(uint32_t)(qq + tt);   // Sorting.h:36-47, :74-88 (32-bit sum)
      key[i] = (d << 31) | qq;                                            // q < 2^31; d < 2^33
    }
  }
}
__global__ void lec_unkey(uint64_t n, const uint64_t* __restrict__ key, uint32_t* q) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) q[i] = (uint32_t)(key[i] & 0x7FFFFFFFu);
}
__global__ void lec_item_sizes(uint64_t n, const uint32_t* __restrict__ itemCluster, const uint64_t* __restrict__ matchOff, uint32_t* sz) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const uint32_t c = itemCluster[i]; sz[i] = (uint32_t)(matchOff[c + 1] - matchOff[c]); }
}
__global__ void __launch_bounds__(64) lec_gather(uint64_t n_items, const uint64_t* __restrict__ slabOff, const uint64_t* __restrict__ outOff, const uint32_t* __restrict__ eq,
                                                 const uint32_t* __restrict__ et, const int32_t* __restrict__ elen, const uint8_t* __restrict__ eovl, uint32_t* oq, uint32_t* ot,
                                                 int32_t* ol, uint8_t* oo) {
  const uint64_t it = blockIdx.x;
  if (it >= n_items) return;
  const uint64_t s = slabOff[it], d = outOff[it], n = outOff[it + 1] - d;
  for (uint64_t x = threadIdx.x; x < n; x += 64) { oq[d + x] = eq[s + x]; ot[d + x] = et[s + x]; ol[d + x] = elen[s + x]; oo[d + x] = eovl[s + x]; }
}

inline size_t szb(size_t n, size_t e) { return (n * e + 255) / 256 * 256; }

}  // namespace

extern "C" int lra_linear_extend_clusters_batch(lra_ctx* ctx, uint64_t n_items, const uint32_t* d_item_cluster, const int32_t* d_item_prev, const int32_t* d_item_next,
                                                const uint32_t* d_item_read, uint64_t n_clusters, const uint64_t* d_match_off, uint64_t n_matches, uint32_t* d_mq, uint32_t* d_mt,
                                                const uint32_t* d_box, const int32_t* d_strand, const int32_t* d_chrom, const float* d_anchorfreq, const char* d_seq,
                                                const uint64_t* d_read_off, const char* d_genome, const uint64_t* h_chrom_pos, int n_chrom, int skiprepetitive, int K, int trim,
                                                lra_ext_clusters_result* out) {
  if (!ctx || !out || !h_chrom_pos || n_chrom < 1 || K < 1) return LRA_ERR_INVALID;
  memset(out, 0, sizeof *out);
  out->n_items = n_items;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t npos = (size_t)n_chrom + 1;
  auto grid = [](uint64_t n) { return dim3((unsigned)((n + 255) / 256)); };
  char* w = (char*)lra_ensure(ctx, 88, szb(n_matches + 1, 8) * 2 + szb(n_matches + 1, 4) + szb(npos, 8) + szb(n_items + 2, 8) * 2 + szb(n_items + 1, 4) * 6 + szb(4 * n_items + 4, 4) + 8192);
  if (!w) return LRA_ERR_NOMEM;
  auto take = [&](size_t n, size_t e) { char* r = w; w += szb(n, e); return r; };
  uint64_t* key = (uint64_t*)take(n_matches + 1, 8); uint64_t* key2 = (uint64_t*)take(n_matches + 1, 8); uint32_t* t2 = (uint32_t*)take(n_matches + 1, 4);
  uint64_t* dpos = (uint64_t*)take(npos, 8); uint64_t* slabOff = (uint64_t*)take(n_items + 2, 8); uint64_t* outOff = (uint64_t*)take(n_items + 2, 8);
  uint32_t* isz = (uint32_t*)take(n_items + 1, 4); uint32_t* cnt = (uint32_t*)take(n_items + 1, 4); uint32_t* novl = (uint32_t*)take(n_items + 1, 4);
  int32_t* ostrand = (int32_t*)take(n_items + 1, 4); int32_t* ochrom = (int32_t*)take(n_items + 1, 4); float* ofreq = (float*)take(n_items + 1, 4);
  uint32_t* obox = (uint32_t*)take(4 * n_items + 4, 4);
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(dpos, h_chrom_pos, npos * 8, hipMemcpyHostToDevice, st));
  out->d_box = obox; out->d_strand = ostrand; out->d_chrom = ochrom; out->d_anchorfreq = ofreq; out->d_anchor_off = outOff;
  if (n_items == 0) { LRA_HIP_CHECK(ctx, hipMemsetAsync(outOff, 0, 16, st)); LRA_HIP_CHECK(ctx, hipStreamSynchronize(st)); return LRA_OK; }
  lra_time_begin(ctx, "linear_extend_clusters");
  if (n_matches && n_clusters) {                                          // (1) the in-place (anti-)diagonal sort of every refined cluster
    hipLaunchKernelGGL(lec_keys, dim3((unsigned)std::min<uint64_t>(n_clusters, (uint64_t)ctx->num_cu * 32)), dim3(64), 0, st, n_clusters, d_match_off, d_strand, d_mq, d_mt, key);
    size_t tb = 0;
    (void)rocprim::segmented_radix_sort_pairs(nullptr, tb, key, key2, d_mt, t2, (unsigned int)n_matches, (unsigned int)n_clusters, d_match_off, d_match_off + 1, 0, 64, st);
    void* tmp = lra_scratch(ctx, 2, tb + 256);
    if (!tmp) { lra_time_end(ctx); return LRA_ERR_NOMEM; }
    hipError_t e = rocprim::segmented_radix_sort_pairs(tmp, tb, key, key2, d_mt, t2, (unsigned int)n_matches, (unsigned int)n_clusters, d_match_off, d_match_off + 1, 0, 64, st);
    if (e != hipSuccess) { lra_time_end(ctx); return lra_set_err(ctx, LRA_ERR_HIP, "segmented sort: %s", hipGetErrorString(e)); }
    hipLaunchKernelGGL(lec_unkey, grid(n_matches), dim3(256), 0, st, n_matches, (const uint64_t*)key2, d_mq);
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(d_mt, t2, n_matches * 4, hipMemcpyDeviceToDevice, st));
  }
  hipLaunchKernelGGL(lec_item_sizes, grid(n_items), dim3(256), 0, st, n_items, d_item_cluster, d_match_off, isz);
  lra_time_end(ctx);
  int rc = lra_exclusive_scan<uint32_t>(ctx, (long)n_items, isz, slabOff);
  if (rc) return rc;
  uint64_t slabTot = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&slabTot, slabOff + n_items, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  char* ws = (char*)lra_ensure(ctx, 89, szb(slabTot + 1, 4) * 3 + szb(slabTot + 1, 1) + 4096);
  if (!ws) return LRA_ERR_NOMEM;
  auto take2 = [&](size_t n, size_t e) { char* r = ws; ws += szb(n, e); return r; };
  LecArgs a; memset(&a, 0, sizeof a);
  a.eq = (uint32_t*)take2(slabTot + 1, 4); a.et = (uint32_t*)take2(slabTot + 1, 4); a.elen = (int32_t*)take2(slabTot + 1, 4); a.eovl = (uint8_t*)take2(slabTot + 1, 1);
  a.n_items = n_items; a.itemCluster = d_item_cluster; a.itemPrev = d_item_prev; a.itemNext = d_item_next; a.itemRead = d_item_read;
  a.matchOff = d_match_off; a.mq = d_mq; a.mt = d_mt; a.box = d_box; a.strand = d_strand; a.chrom = d_chrom; a.freq = d_anchorfreq;
  a.seq = (const unsigned char*)d_seq; a.read_off = d_read_off; a.genome = (const unsigned char*)d_genome; a.pos = dpos; a.skiprepetitive = skiprepetitive; a.K = K;
  a.slabOff = slabOff; a.cnt = cnt; a.novl = novl; a.obox = obox; a.ostrand = ostrand; a.ochrom = ochrom; a.ofreq = ofreq;
  lra_time_begin(ctx, "linear_extend_clusters");
  hipLaunchKernelGGL(lec_kernel, dim3((unsigned)((n_items + LEC_LANES - 1) / LEC_LANES)), dim3(64), 0, st, a);
  lra_time_end(ctx);
  if ((rc = lra_exclusive_scan<uint32_t>(ctx, (long)n_items, cnt, outOff))) return rc;
  uint64_t nA = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&nA, outOff + n_items, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  char* wo = (char*)lra_ensure(ctx, 90, szb(nA + 1, 4) * 3 + szb(nA + 1, 1) + 4096);
  if (!wo) return LRA_ERR_NOMEM;
  auto take3 = [&](size_t n, size_t e) { char* r = wo; wo += szb(n, e); return r; };
  uint32_t* oq = (uint32_t*)take3(nA + 1, 4); uint32_t* ot = (uint32_t*)take3(nA + 1, 4); int32_t* ol = (int32_t*)take3(nA + 1, 4); uint8_t* oo = (uint8_t*)take3(nA + 1, 1);
  hipLaunchKernelGGL(lec_gather, dim3((unsigned)n_items), dim3(64), 0, st, n_items, (const uint64_t*)slabOff, (const uint64_t*)outOff, a.eq, a.et, a.elen, a.eovl, oq, ot, ol, oo);
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  LRA_HIP_CHECK(ctx, hipGetLastError());
  out->n_anchors = nA; out->d_q = oq; out->d_t = ot; out->d_len = ol; out->d_overlap = oo;
  if (trim && nA) { rc = lra_trim_overlapped_anchors_batch(ctx, n_items, outOff, nA, ostrand, oq, ot, ol); if (rc) return rc; }   // LinearExtend_chain :791
  return LRA_OK;
}
