// lra_amd/csrc/stats.hip -- a16: alignment statistics and CIGAR runs for a batch of alignments (gfx950).
//
// Replaces Alignment::CalculateStatistics (reference: Alignment.h:513-531): CreateAlignmentStrings
// (:247-331) + AlignStringsToCigar (:414-504, opts.showmm).  The reference materialises three
// alignment strings and re-parses them; here the column stream is generated on the fly from the
// blocks: one wave per alignment, 64 aligned pairs per step compared at once (ballot of the
// mismatch flags), runs cut out of the 64-bit masks with count-trailing-zeros, and the counters and
// the float value accumulated run by run in the reference's order (so the float additions round
// identically; the log table is the caller's, i.e. the host libm's logf, LogLookUpTable.h:9-15).
#include "common.h"
#include "scan.h"
#include <algorithm>
#include <stdlib.h>

namespace {

__device__ __forceinline__ int code2(unsigned char c) {   // seqMap (SeqUtils.h:7): non-ACGT -> 0
  switch (c) {
    case 1: case 5: case 'C': case 'c': return 1;
    case 2: case 6: case 'G': case 'g': return 2;
    case 3: case 7: case 'T': case 't': return 3;
    default: return 0;
  }
}

struct StatArgs {
  int n_aln;
  const int32_t* blocks; const uint64_t* block_off;
  const unsigned char* qseq; const uint64_t* q_off; const int32_t* q_len;
  const unsigned char* tseq; const uint64_t* t_off;
  const float* lut;
  int32_t* counts; float* value; uint32_t* n_runs;
  const uint64_t* cap_off; uint32_t* runs;      // runs at capacity offsets (<= aligned columns per alignment)
  const int32_t* only;                          // stats_kernel: when set, only the alignments flagged here (the ones stats_starts leaves to the serial walk)
  int32_t* serial; uint32_t* cols;              // stats_starts: alignment left to the serial walk / its number of columns
  const uint64_t* run_off; uint32_t* dst;       // stats_runs: the compact run lists
};

// Value bookkeeping: '=' / 'X' runs and gaps <= 20 add integers; as long as no long gap (a log-table
// term, :462/:491) has been added the float `value` is an exact integer, so those contributions can
// be summed in any order (ival) -- whole 64-column chunks at a time.  From the first long gap on
// every run is applied to the float in the reference's order.
// GW lanes per alignment (64 / GW alignments per wave): a noisy read's blocks are ~16 columns, so a whole wave per alignment leaves three quarters of its lanes idle
// on almost every block, and the kernel's time is (alignments / resident groups) x one alignment's chain of ~1800 dependent block steps.
template <int GW>
__global__ void __launch_bounds__(64) stats_kernel(StatArgs A) {
  const int lane = threadIdx.x & (GW - 1), gbase = threadIdx.x - lane;      // lane: inside the alignment's group
  const unsigned long long below = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));
  const unsigned long long gmask = ~0ULL >> (64 - GW);
  auto BAL = [&](bool x) -> unsigned long long { return (__ballot(x) >> gbase) & gmask; };
  constexpr int GPW = 64 / GW;
  for (int a = blockIdx.x * GPW + (int)threadIdx.x / GW; a < A.n_aln; a += gridDim.x * GPW) {
    if (A.only && !A.only[a]) continue;
    const long nb = (long)(A.block_off[a + 1] - A.block_off[a]);
    const int32_t* B = A.blocks + 3 * A.block_off[a];
    const unsigned char* R = A.qseq + A.q_off[a];
    const unsigned char* G = A.tseq + A.t_off[a];
    uint32_t* out = A.runs + A.cap_off[a];
    int nm = 0, nmm = 0, nD = 0, nI = 0, tdel = 0, tins = 0, sD = 0, mD = 0, lD = 0, sI = 0, mI = 0, lI = 0;
    long ival = 0; float value = 0; bool frac = false;
    uint32_t nr = 0;
    int curType = -1; long curLen = 0;
    auto addv = [&](long x) { if (!frac) ival += x; else value += (float)x; };   // x integer: +len or -len
    auto close_run = [&]() {                                             // one CIGAR run (:419-501)
      if (curType < 0 || curLen == 0) return;
      const long len = curLen;
      if (lane == 0) out[nr] = (uint32_t)(len << 4) | (uint32_t)curType;
      nr++;
      if (curType == 0) { nm += (int)len; addv(len); }
      else if (curType == 1) { nmm += (int)len; addv(-len); }
      else {
        const bool small = len <= 20;
        if (curType == 3) {                                              // 'D' :447-470
          tdel += (int)len; nD++;
          if (len <= 10) sD++;
          if (len > 10 && len < 50) mD++; else if (len > 50) lD++;
        } else {                                                         // 'I' :472-499
          tins += (int)len; nI++;
          if (len <= 10) sI++;
          if (len > 10 && len < 50) mI++; else if (len > 50) lI++;
          if (small) sI++;
        }
        if (small) addv(-len);
        else {
          if (!frac) { value = (float)ival; frac = true; }
          float pen;
          if (len <= 10001) pen = -3.0f * A.lut[(int)((len - 1) / 5)] - 1;
          else if (len <= 100001) pen = -1000;
          else pen = -2000;
          value += pen;
        }
      }
    };
    auto feed = [&](int type, long len) {                                // append `len` columns of one type
      if (len <= 0) return;
      if (type == curType) curLen += len;
      else { close_run(); curType = type; curLen = len; }
    };
    // the first 64 columns of the next block are fetched while the current block is worked on (blocks are ~16 bp on noisy reads, so
    // without this every block costs a full dependent-load latency)
    auto pairs = [&](long q, long t, long len, bool usePre, unsigned long long pm) {   // aligned pairs: '=' / 'X' by base code
      for (long off = 0; off < len; off += GW) {
        const int cnt = (int)min((long)GW, len - off);
        const unsigned long long valid = cnt >= 64 ? ~0ULL : ((1ULL << cnt) - 1);
        unsigned long long mx;
        if (off == 0 && usePre) mx = pm & valid;
        else {
          bool x = false;
          if (lane < cnt) x = code2(R[q + off + lane]) != code2(G[t + off + lane]);
          mx = BAL(x) & valid;
        }
        // run starts inside the chunk: column c > 0 whose kind differs from column c-1
        const unsigned long long starts = ((mx ^ (mx << 1)) & valid) & ~1ULL;
        if (!starts) { feed((int)(mx & 1ULL), cnt); continue; }          // the whole chunk is one kind
        const int firstStart = __ffsll((long long)starts) - 1, lastStart = 63 - __clzll((long long)starts);
        feed((int)(mx & 1ULL), firstStart);                              // head: continues the open run
        close_run();
        // interior runs [start_i, start_{i+1}): every run-start lane writes its own run
        const int nInner = __popcll(starts) - 1;
        if ((starts >> lane) & 1ULL && lane != lastStart) {
          const unsigned long long later = starts & ~(below | (1ULL << lane));
          const int nxt = __ffsll((long long)later) - 1;
          out[nr + __popcll(starts & below)] = (uint32_t)((nxt - lane) << 4) | (uint32_t)((mx >> lane) & 1ULL);
        }
        if (nInner > 0) {
          const unsigned long long inner = ((1ULL << lastStart) - 1) & ~((1ULL << firstStart) - 1);   // columns [firstStart, lastStart)
          const int nx = __popcll(mx & inner), ne = __popcll(inner) - nx;
          nm += ne; nmm += nx;
          if (!frac) ival += ne - nx;
          else {                                                          // rare: apply in order
            unsigned long long st2 = starts & ~(1ULL << lastStart);
            while (st2) {
              const int c0 = __ffsll((long long)st2) - 1; st2 &= st2 - 1;
              const unsigned long long later = starts & ~((1ULL << c0) | ((1ULL << c0) - 1));
              const int c1 = __ffsll((long long)later) - 1;
              value += ((mx >> c0) & 1ULL) ? -(float)(c1 - c0) : (float)(c1 - c0);
            }
          }
          nr += nInner;
        }
        curType = (int)((mx >> lastStart) & 1ULL); curLen = cnt - lastStart;   // tail: stays open
      }
    };
    if (nb > 0) {
      long q = B[0], t = B[1];
      // block table: 64 blocks per load, handed out by shuffles.  Each lane also compares the first 64 columns of its block, so the
      // per-block loop below touches no memory for blocks of up to 64 columns (a noisy read's blocks are ~16 bp: one dependent HBM
      // round trip per block otherwise)
      int tq = 0, tt = 0, tl = 0; unsigned long long tm = 0; long tbase = -GW;
      auto blk = [&](long b, int& bq, int& bt, int& bl, unsigned long long& bm) {
        if (b >= tbase + GW || b < tbase) {
          tbase = b;
          const long i = b + lane;
          tm = 0;
          if (i < nb) {
            tq = B[3 * i]; tt = B[3 * i + 1]; tl = B[3 * i + 2];
            const int c1 = min(tl, GW);
            const unsigned char* rp = R + tq; const unsigned char* gp = G + tt;
#pragma unroll 8
            for (int c = 0; c < c1; c++) tm |= (unsigned long long)(code2(rp[c]) != code2(gp[c])) << c;
          }
        }
        const int k = gbase + (int)(b - tbase);
        bq = __shfl(tq, k); bt = __shfl(tt, k); bl = __shfl(tl, k); bm = __shfl(tm, k);
      };
      int cq, ct, cl; unsigned long long cm;
      blk(0, cq, ct, cl, cm);
      for (long b = 0; b < nb; b++) {                                    // :261-330
        const long L = cl;
        int nq = 0, nt = 0, nl = 0; unsigned long long nm2 = 0;
        const bool hasNext = b + 1 < nb;
        if (hasNext) blk(b + 1, nq, nt, nl, nm2);
        pairs(q, t, L, q == cq && t == ct, cm);
        q += L; t += L;
        if (!hasNext) continue;
        long qg = (long)nq - cq - L, tg = (long)nt - ct - L;
        if (qg > 0 || tg > 0) {
          const long common = qg > tg ? tg : qg;
          tg -= common; qg -= common;
          feed(2, qg); q += max(qg, 0L);
          feed(3, tg); t += max(tg, 0L);
          if (common > 0) { pairs(q, t, common, false, 0); q += common; t += common; }
        }
        cq = nq; ct = nt; cl = nl; cm = nm2;
      }
      close_run();
    }
    if (lane == 0) {
      int32_t* o = A.counts + 18 * (long)a;
      o[0] = nm; o[1] = nmm; o[2] = nD; o[3] = nI; o[4] = tdel; o[5] = tins; o[6] = sD; o[7] = mD; o[8] = lD; o[9] = sI; o[10] = mI; o[11] = lI;
      if (nb > 0) {
        const long last = nb - 1;
        o[12] = B[0]; o[13] = A.q_len[a] - B[3 * last] - B[3 * last + 2];
        o[14] = B[0]; o[15] = B[3 * last] + B[3 * last + 2]; o[16] = B[1]; o[17] = B[3 * last + 1] + B[3 * last + 2];
      } else { for (int x = 12; x < 18; x++) o[x] = 0; }
      A.value[a] = frac ? value : (float)ival;
      A.n_runs[a] = nr;
    }
  }
}


// ---- the parallel form.  stats_kernel walks an alignment's blocks one after the other (~1800 dependent steps of ~500 wave instructions for four alignments per wave:
// 45 ms per batch, all of it instruction issue).  But the column stream is a concatenation of per-block pieces -- [L pairs][insertion][deletion][`common` pairs], all
// given by the block and its successor as long as consecutive blocks do not overlap -- and the CIGAR is its run-length encoding, so:
//   stats_starts  a wave per alignment, a LANE per block: every lane walks its own block's columns and notes where a run starts (a column whose kind differs from the
//                 column before it; the kind of the column before a block is the kind of its predecessor's last column, which that lane can tell without walking);
//                 positions and run indices come from wave prefix sums carried from one group of 64 blocks to the next.
//   stats_runs    a wave per alignment, a lane per RUN: length = distance to the next start, the twelve counters as wave reductions, `value` as an integer sum up to the
//                 first gap longer than 20 and from there on as the reference's chain of float additions (:462, :491 make it a non-integer), 64 runs per step.
// Where blocks overlap the walk's q / t leave the blocks' coordinates (:261-330 advance them by what was consumed): they are prefix sums of per-block advances,
// taken the same way.  Alignments with gaps or blocks of 2^27 and more keep the serial walk.
struct BlockPiece { int cq, ct, L, ga, gb, common, bl; };   // L: columns of the block's own pairs (max(bl, 0)); bl: what the walk adds to q / t after them (the block's length as given)
__device__ __forceinline__ int col_kind(const unsigned char* R, const unsigned char* G, long q, long t) { return code2(R[q]) != code2(G[t]) ? 1 : 0; }
// seqMap without the switch: letters by their low five bits (A 1, C 3, G 7, T 20, either case), the raw codes 0..7 by their low two bits, everything else 0
__device__ __forceinline__ uint32_t code2b(uint32_t c) {
  constexpr unsigned long long T = (1ULL << 6) | (2ULL << 14) | (3ULL << 40);
  const uint32_t letter = (uint32_t)(T >> (2 * (c & 31u))) & 3u;
  return (c & 0xC0u) == 0x40u ? letter : (c < 8u ? (c & 3u) : 0u);
}
// bit c of the result: column c of the piece (c < m <= 64) is a mismatch.  Eight bases per load where eight are left (the tail byte by byte: nothing is read past the piece).
__device__ __forceinline__ unsigned long long kinds64(const unsigned char* rp, const unsigned char* gp, int m) {
  unsigned long long mx = 0;
  int c = 0;
  for (; c + 8 <= m; c += 8) {
    unsigned long long r8, g8;
    __builtin_memcpy(&r8, rp + c, 8); __builtin_memcpy(&g8, gp + c, 8);
    uint32_t bits = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) bits |= (uint32_t)(code2b((uint32_t)(r8 >> (8 * j)) & 255u) != code2b((uint32_t)(g8 >> (8 * j)) & 255u)) << j;
    mx |= (unsigned long long)bits << c;
  }
  for (; c < m; c++) mx |= (unsigned long long)(code2b(rp[c]) != code2b(gp[c])) << c;
  return mx;
}

// The run starts of one block's columns.  m1 / m2: the kinds of the first 64 columns of its two runs of pairs (computed once, used by the counting and the writing call).
template <bool WRITE>
__device__ __forceinline__ uint32_t walk_piece(const BlockPiece& P, const unsigned char* R, const unsigned char* G, int prevKind, uint32_t pos, uint32_t* out,
                                               unsigned long long m1, unsigned long long m2) {
  int cur = prevKind; uint32_t k = 0;
  auto pairs = [&](const unsigned char* rp, const unsigned char* gp, int n, unsigned long long first) {
    for (int c0 = 0; c0 < n; c0 += 64) {
      const int m = n - c0 < 64 ? n - c0 : 64;
      const unsigned long long mx = (c0 == 0) ? first : kinds64(rp + c0, gp + c0, m);
      const unsigned long long valid = m >= 64 ? ~0ULL : ((1ULL << m) - 1);
      unsigned long long st = (mx ^ ((mx << 1) | (unsigned long long)(cur == 1))) & valid;
      if (cur != 0 && cur != 1) st |= 1ULL;                              // after a gap / at the alignment's start the first column starts a run
      if (WRITE) { unsigned long long x = st; uint32_t kk = k; while (x) { const int c = __ffsll((long long)x) - 1; x &= x - 1; out[kk++] = ((pos + (uint32_t)(c0 + c)) << 2) | (uint32_t)((mx >> c) & 1ULL); } }
      k += (uint32_t)__popcll(st);
      cur = (int)((mx >> (m - 1)) & 1ULL);
    }
    pos += (uint32_t)n;
  };
  pairs(R + P.cq, G + P.ct, P.L, m1);
  if (P.ga > 0) { if (cur != 2) { if (WRITE) out[k] = (pos << 2) | 2u; k++; cur = 2; } pos += (uint32_t)P.ga; }
  if (P.gb > 0) { if (cur != 3) { if (WRITE) out[k] = (pos << 2) | 3u; k++; cur = 3; } pos += (uint32_t)P.gb; }
  if (P.common > 0) pairs(R + P.cq + P.bl + P.ga, G + P.ct + P.bl + P.gb, P.common, m2);
  return k;
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
  for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(v, d); if (lane >= d) v += y; }
  return v;
}

__global__ void __launch_bounds__(64) stats_starts(StatArgs A) {
  const int lane = threadIdx.x;
  for (int a = blockIdx.x; a < A.n_aln; a += gridDim.x) {
    const long nb = (long)(A.block_off[a + 1] - A.block_off[a]);
    const int32_t* B = A.blocks + 3 * A.block_off[a];
    const unsigned char* R = A.qseq + A.q_off[a];
    const unsigned char* G = A.tseq + A.t_off[a];
    uint32_t* out = A.runs + A.cap_off[a];
    uint32_t carryPos = 0, carryIdx = 0; int carryKind = -1; bool bad = false;
    long carryQ = nb > 0 ? (long)B[0] : 0, carryT = nb > 0 ? (long)B[1] : 0;     // the walk's own q / t (:261-330): they leave the blocks' coordinates where blocks overlap
    for (long base = 0; base < nb; base += 64) {
      const long i = base + lane; const bool valid = i < nb;
      BlockPiece P = {0, 0, 0, 0, 0, 0, 0}; bool irr = false;
      long advq = 0, advt = 0;
      if (valid) {
        const int bq = B[3 * i], bt = B[3 * i + 1], bl = B[3 * i + 2];
        P.L = bl > 0 ? bl : 0; P.bl = bl;
        advq = bl; advt = bl;
        if (i + 1 < nb) {
          long qg = (long)B[3 * i + 3] - bq - bl, tg = (long)B[3 * i + 4] - bt - bl;
          if (qg >= (1L << 27) || tg >= (1L << 27) || qg <= -(1L << 27) || tg <= -(1L << 27) || bl >= (1 << 27)) irr = true;
          else if (qg > 0 || tg > 0) {
            const long c = qg > tg ? tg : qg;
            tg -= c; qg -= c;
            P.ga = qg > 0 ? (int)qg : 0; P.gb = tg > 0 ? (int)tg : 0; P.common = c > 0 ? (int)c : 0;
            advq += P.ga + P.common; advt += P.gb + P.common;
          }
        }
      }
      if (__ballot(irr) != 0ULL) { bad = true; break; }
      { long iq = advq, it = advt;                                         // this block's q / t: the sums of the advances before it
        for (int d = 1; d < 64; d <<= 1) { const long yq = __shfl_up(iq, d), yt = __shfl_up(it, d); if (lane >= d) { iq += yq; it += yt; } }
        const long q0 = carryQ + iq - advq, t0 = carryT + it - advt;
        P.cq = (int)q0; P.ct = (int)t0;
        carryQ += __shfl(iq, 63); carryT += __shfl(it, 63);
      }
      unsigned long long m1 = 0, m2 = 0;
      if (valid) {
        m1 = kinds64(R + P.cq, G + P.ct, P.L < 64 ? P.L : 64);
        if (P.common > 0) m2 = kinds64(R + P.cq + P.bl + P.ga, G + P.ct + P.bl + P.gb, P.common < 64 ? P.common : 64);
      }
      int lastKind = -1;                                                   // kind of the block's last column (-1: it has none)
      if (valid) {
        if (P.common > 0) lastKind = P.common <= 64 ? (int)((m2 >> (P.common - 1)) & 1ULL) : col_kind(R, G, (long)P.cq + P.bl + P.ga + P.common - 1, (long)P.ct + P.bl + P.gb + P.common - 1);
        else if (P.gb > 0) lastKind = 3;
        else if (P.ga > 0) lastKind = 2;
        else if (P.L > 0) lastKind = P.L <= 64 ? (int)((m1 >> (P.L - 1)) & 1ULL) : col_kind(R, G, (long)P.cq + P.L - 1, (long)P.ct + P.L - 1);
      }
      int lk = lastKind;                                                   // the last column at or before this block
      if (lane == 0 && lk < 0) lk = carryKind;
      for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(lk, d); if (lane >= d && lk < 0) lk = y; }
      int prevKind = __shfl_up(lk, 1);
      if (lane == 0) prevKind = carryKind;
      const uint32_t ncols = valid ? (uint32_t)(P.L + P.ga + P.gb + P.common) : 0u;
      const uint32_t ns = valid ? walk_piece<false>(P, R, G, prevKind, 0, nullptr, m1, m2) : 0u;
      const uint32_t iPos = wave_incl_scan(ncols, lane), iIdx = wave_incl_scan(ns, lane);
      if (valid && ns) walk_piece<true>(P, R, G, prevKind, carryPos + iPos - ncols, out + (carryIdx + iIdx - ns), m1, m2);
      carryPos += __shfl(iPos, 63); carryIdx += __shfl(iIdx, 63);
      carryKind = __shfl(lk, 63);
    }
    if (lane == 0) {
      A.serial[a] = bad ? 1 : 0;
      if (!bad) { A.n_runs[a] = carryIdx; A.cols[a] = carryPos; }
    }
  }
}

__device__ __forceinline__ int wave_sum_i(int v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ __forceinline__ long wave_sum_l(long v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }

__global__ void __launch_bounds__(64) stats_runs(StatArgs A) {
  const int lane = threadIdx.x;
  for (int a = blockIdx.x; a < A.n_aln; a += gridDim.x) {
    const uint32_t nr = A.n_runs[a];
    const uint32_t* src = A.runs + A.cap_off[a];
    uint32_t* dst = A.dst + A.run_off[a];
    if (A.serial[a]) {                                                   // the serial walk has written the runs, the counters and the value
      for (uint32_t x = lane; x < nr; x += 64) dst[x] = src[x];
      continue;
    }
    const uint32_t total = A.cols[a];
    int nm = 0, nmm = 0, nD = 0, nI = 0, tdel = 0, tins = 0, sD = 0, mD = 0, lD = 0, sI = 0, mI = 0, lI = 0;
    long ival = 0; float value = 0; bool frac = false;
    for (uint32_t base = 0; base < nr; base += 64) {
      const uint32_t i = base + lane; const bool valid = i < nr;
      int type = 0; long len = 0;
      if (valid) {
        const uint32_t w = src[i], wn = (i + 1 < nr) ? src[i + 1] : (total << 2);
        type = (int)(w & 3u); len = (long)((wn >> 2) - (w >> 2));
        dst[i] = (uint32_t)(len << 4) | (uint32_t)type;
      }
      bool isLong = false; long x = 0; float pen = 0;
      if (valid) {                                                       // one CIGAR run (:419-501)
        if (type == 0) { nm += (int)len; x = len; }
        else if (type == 1) { nmm += (int)len; x = -len; }
        else {
          const bool small = len <= 20;
          if (type == 3) {                                               // 'D' :447-470
            tdel += (int)len; nD++;
            if (len <= 10) sD++;
            if (len > 10 && len < 50) mD++; else if (len > 50) lD++;
          } else {                                                       // 'I' :472-499
            tins += (int)len; nI++;
            if (len <= 10) sI++;
            if (len > 10 && len < 50) mI++; else if (len > 50) lI++;
            if (small) sI++;
          }
          if (small) x = -len;
          else {
            isLong = true;
            if (len <= 10001) pen = -3.0f * A.lut[(int)((len - 1) / 5)] - 1;
            else if (len <= 100001) pen = -1000;
            else pen = -2000;
          }
        }
      }
      const int cnt = (nr - base >= 64) ? 64 : (int)(nr - base);
      int from = 0;                                                      // lanes from here on go through the float
      if (!frac) {
        const unsigned long long m = __ballot(isLong);
        if (m == 0ULL) { ival += x; from = 64; }
        else {
          from = __ffsll((long long)m) - 1;
          if (lane < from) ival += x;
          value = (float)wave_sum_l(ival); frac = true;
        }
      }
      if (from < cnt) {
        const float add = isLong ? pen : (float)x;
        for (int l = from; l < cnt; l++) value += __shfl(add, l);
      }
    }
    nm = wave_sum_i(nm); nmm = wave_sum_i(nmm); nD = wave_sum_i(nD); nI = wave_sum_i(nI); tdel = wave_sum_i(tdel); tins = wave_sum_i(tins);
    sD = wave_sum_i(sD); mD = wave_sum_i(mD); lD = wave_sum_i(lD); sI = wave_sum_i(sI); mI = wave_sum_i(mI); lI = wave_sum_i(lI);
    if (!frac) ival = wave_sum_l(ival);
    if (lane == 0) {
      const long nb = (long)(A.block_off[a + 1] - A.block_off[a]);
      const int32_t* B = A.blocks + 3 * A.block_off[a];
      int32_t* o = A.counts + 18 * (long)a;
      o[0] = nm; o[1] = nmm; o[2] = nD; o[3] = nI; o[4] = tdel; o[5] = tins; o[6] = sD; o[7] = mD; o[8] = lD; o[9] = sI; o[10] = mI; o[11] = lI;
      if (nb > 0) {
        const long last = nb - 1;
        o[12] = B[0]; o[13] = A.q_len[a] - B[3 * last] - B[3 * last + 2];
        o[14] = B[0]; o[15] = B[3 * last] + B[3 * last + 2]; o[16] = B[1]; o[17] = B[3 * last + 1] + B[3 * last + 2];
      } else { for (int x2 = 12; x2 < 18; x2++) o[x2] = 0; }
      A.value[a] = frac ? value : (float)ival;
    }
  }
}

// capacity per alignment = aligned columns + 2 (every run has >= 1 column)
__global__ void stats_capacity(int n_aln, const int32_t* blocks, const uint64_t* block_off, uint64_t* cap) {
  int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n_aln) return;
  const uint64_t b0 = block_off[a], b1 = block_off[a + 1];
  if (b1 == b0) { cap[a] = 1; return; }
  const int32_t* f = blocks + 3 * b0; const int32_t* l = blocks + 3 * (b1 - 1);
  const long qs = (long)l[0] + l[2] - f[0], ts = (long)l[1] + l[2] - f[1];
  cap[a] = (uint64_t)(max(qs, 0L) + max(ts, 0L) + 2);
}

__global__ void stats_fill(int n, int32_t* p, int v) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }

__global__ void __launch_bounds__(64) stats_compact(int n_aln, const uint64_t* cap_off, const uint64_t* run_off, const uint32_t* src, uint32_t* dst) {
  for (int a = blockIdx.x; a < n_aln; a += gridDim.x) {
    const uint64_t s = cap_off[a], d = run_off[a], n = run_off[a + 1] - d;
    for (uint64_t x = threadIdx.x; x < n; x += 64) dst[d + x] = src[s + x];
  }
}

}  // namespace

extern "C" int lra_calculate_statistics_batch(lra_ctx* ctx, int n_aln, const int32_t* d_blocks, const uint64_t* d_block_off,
                                              const char* d_qseq, const uint64_t* d_q_off, const int32_t* d_q_len, const char* d_tseq,
                                              const uint64_t* d_t_off, const float* h_lookup, int n_lookup, lra_stats_result* out) {
  if (!ctx || !out || n_aln < 0 || !h_lookup || n_lookup < 2001) return LRA_ERR_INVALID;
  memset(out, 0, sizeof(*out));
  out->n_aln = n_aln;
  if (n_aln == 0) return LRA_OK;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t nA = (size_t)n_aln;
  auto sz = [](size_t n, size_t e) { return (n * e + 255) & ~(size_t)255; };
  char* w = (char*)lra_scratch(ctx, 2, sz(18 * nA, 4) + sz(nA, 4) * 4 + sz(nA + 1, 8) * 2 + sz(nA, 8) + sz((size_t)n_lookup, 4) + 4096);
  if (!w) return LRA_ERR_NOMEM;
  StatArgs A;
  A.n_aln = n_aln; A.blocks = d_blocks; A.block_off = d_block_off;
  A.qseq = (const unsigned char*)d_qseq; A.q_off = d_q_off; A.q_len = d_q_len; A.tseq = (const unsigned char*)d_tseq; A.t_off = d_t_off;
  A.counts = (int32_t*)w; w += sz(18 * nA, 4);
  A.value = (float*)w; w += sz(nA, 4);
  A.n_runs = (uint32_t*)w; w += sz(nA, 4);
  A.serial = (int32_t*)w; w += sz(nA, 4);
  A.cols = (uint32_t*)w; w += sz(nA, 4);
  uint64_t* run_off = (uint64_t*)w; w += sz(nA + 1, 8);
  uint64_t* cap_off = (uint64_t*)w; w += sz(nA + 1, 8);
  uint64_t* cap = (uint64_t*)w; w += sz(nA, 8);
  float* lut = (float*)w;
  A.lut = lut; A.cap_off = cap_off;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(lut, h_lookup, (size_t)n_lookup * 4, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(stats_capacity, dim3((n_aln + 255) / 256), dim3(256), 0, st, n_aln, d_blocks, d_block_off, cap);
  if (lra_exclusive_scan<uint64_t>(ctx, (long)n_aln, cap, cap_off)) return LRA_ERR_HIP;
  uint64_t total_cap = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&total_cap, cap_off + n_aln, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  uint32_t* tmp = (uint32_t*)lra_ensure(ctx, 12, (total_cap + 1) * 4);     // shares the buffer of the sparse DP's arena / the refine stage's temporaries (all dead here)
  if (!tmp) return LRA_ERR_NOMEM;
  A.runs = tmp;
  const int grid = n_aln < ctx->num_cu * 32 ? n_aln : ctx->num_cu * 32;
  static const int statGw = getenv("LRA_STATS_GW") ? atoi(getenv("LRA_STATS_GW")) : 16;
  static const bool serialOnly = getenv("LRA_STATS_SERIAL") != nullptr;     // the one-alignment-after-the-other walk for everything (kept for comparison)
  A.only = nullptr;
  lra_time_begin(ctx, "stats");
  if (serialOnly) {
    if (statGw == 64) hipLaunchKernelGGL(stats_kernel<64>, dim3(grid), dim3(64), 0, st, A);
    else if (statGw == 32) hipLaunchKernelGGL(stats_kernel<32>, dim3(std::min((n_aln + 1) / 2, ctx->num_cu * 32)), dim3(64), 0, st, A);
    else hipLaunchKernelGGL(stats_kernel<16>, dim3(std::min((n_aln + 3) / 4, ctx->num_cu * 32)), dim3(64), 0, st, A);
    hipLaunchKernelGGL(stats_fill, dim3((n_aln + 255) / 256), dim3(256), 0, st, n_aln, A.serial, 1);
  } else {
    hipLaunchKernelGGL(stats_starts, dim3(n_aln), dim3(64), 0, st, A);
    A.only = A.serial;
    hipLaunchKernelGGL(stats_kernel<64>, dim3(grid), dim3(64), 0, st, A);   // the alignments with overlapping / empty blocks (normally none)
  }
  lra_time_end(ctx);
  if (getenv("LRA_STATS_DBG")) {
    std::vector<int32_t> h(nA); std::vector<uint64_t> bo(nA + 1);
    (void)hipMemcpyAsync(h.data(), A.serial, nA * 4, hipMemcpyDeviceToHost, st); (void)hipMemcpyAsync(bo.data(), d_block_off, (nA + 1) * 8, hipMemcpyDeviceToHost, st); (void)hipStreamSynchronize(st);
    long ns = 0, nbmax = 0; for (size_t i = 0; i < nA; i++) if (h[i]) { ns++; nbmax = std::max(nbmax, (long)(bo[i + 1] - bo[i])); }
    fprintf(stderr, "[stats] %d alignments, %ld left to the serial walk (largest: %ld blocks)\n", n_aln, ns, nbmax);
  }
  if (lra_exclusive_scan<uint32_t>(ctx, (long)n_aln, A.n_runs, run_off)) return LRA_ERR_HIP;
  uint64_t total = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&total, run_off + n_aln, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  uint32_t* runs = (uint32_t*)lra_scratch(ctx, 3, (total + 1) * 4);
  if (!runs) return LRA_ERR_NOMEM;
  A.run_off = run_off; A.dst = runs;
  lra_time_begin(ctx, "stats_cigar");
  hipLaunchKernelGGL(stats_runs, dim3(n_aln), dim3(64), 0, st, A);
  lra_time_end(ctx);
  LRA_HIP_CHECK(ctx, hipGetLastError());
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  out->n_runs = total; out->d_counts = A.counts; out->d_value = A.value; out->d_run_off = run_off; out->d_runs = runs;
  return LRA_OK;
}
