// lra_amd/csrc/index.hip -- SURVEY §8 row f1: `lra index` for the global minimizer index on the device, and the .mms / .gli files.  gfx950 only.
//
//   StoreIndex                      MMIndex.h:286-400   (per sequence StoreMinimizers<GenomeTuple,Tuple> MinCount.h:8-179, sort by masked key :314,
//                                                        frequency filter :331-352, CountSort :258-283, <= NumOfminimizersPerWindow per
//                                                        globalWinsize window :359-376, RemoveFrequent :88-98)
//   WriteIndex / ReadIndex          MMIndex.h:402-424   (+ Header::Write / Read  Genome.h:59-84)
//   LocalIndex::Write / Read        MMIndex.h:138-173
//
// Mapping.
//  (1) gsketch: a chromosome is cut into chunks of 4096 k-mer positions, one wave per chunk, 64 positions per tile exactly like the read
//      sketch of seed.hip.  The only serial state of StoreMinimizers is the position of the active minimizer; wherever the minimum of the
//      current window is unique (or the incoming k-mer is strictly smaller than the previous window's minimum) that position does not depend
//      on the history, so a chunk starts 192 positions early with a guessed state and a flag "state is exact" that such a position sets.  A
//      chunk that would have to emit before its state is exact (long runs of tied k-mers: low-complexity sequence) is redone by one lane
//      (gsketch_serial) that backs off further and further, down to the start of the chromosome, until it is exact.  Windows with a non-ACGT
//      byte emit nothing (:109-132): valid(p) <=> no such byte in [p-w+1, p+k-1], and nothing at all once the bases after the last such byte
//      are too few for `nextValidWindowStart < seqLen - windowSpan` (:117) -- only the very last position can see that case.
//  (2) stable radix sort by masked key (rocPRIM).  libstdc++'s std::sort leaves equal keys in an order only a literal introsort of the whole
//      200 M-entry array reproduces; the entries and their key order are the reference's, the order INSIDE a run of equal keys is emission
//      order here.  Which entries survive (3) depends on that order only when one window holds two candidates with the same key.
//  (3) frequency of every key run, drop runs longer than globalMaxFreq; thinning = for every window the first NumOfminimizersPerWindow
//      candidates in CountSort's order (frequency ascending, sorted position descending): candidates are laid out in descending sorted
//      position and stably radix-sorted by (window, frequency); a candidate survives iff the one NumOfminimizersPerWindow places before it
//      belongs to another window.
// Algorithmic bytes: genome bytes once + 12 B per minimizer x (emit + 2 sort passes + compaction).
#include "common.h"
#include "seed_state.h"
#include "scan.h"
#include "kmer.h"
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <stdio.h>
#include <string>
#include <vector>

int lra_seed_install_index(lra_ctx* ctx, uint64_t* d_key, uint32_t* d_pos, uint64_t n);   // seed.hip: adopts the arrays, builds the bucket directory

namespace {

constexpr int GCH = 4096;       // k-mer positions per chunk
constexpr int GWARM = 192;      // positions a chunk starts early (3 tiles)
constexpr uint32_t UNK = 0xFFFFFFFFu;

struct GSketchArgs {
  const unsigned char* genome; const uint64_t* chrom_pos; int n_chrom;
  const uint64_t* chunk_first;   // [n_chrom+1] first chunk of every chromosome
  uint64_t n_chunks; int k, w;
  const uint64_t* out_off; uint64_t* out_key; uint32_t* out_pos; uint32_t* counts; uint32_t* flags;
};

__device__ __forceinline__ int chrom_of_chunk(const uint64_t* chunk_first, int n_chrom, uint64_t chunk) {
  int lo = 0, hi = n_chrom;                      // last c with chunk_first[c] <= chunk
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (chunk_first[mid] <= chunk) lo = mid; else hi = mid; }
  return lo;
}
// highest set bit with index <= e of the 128-bit mask (hi:lo), -1 if none
__device__ __forceinline__ int last_set_le(unsigned long long lo, unsigned long long hi, int e) {
  if (e >= 64) {
    const int eh = e - 64;
    const unsigned long long m = (eh >= 63) ? hi : (hi & ((2ULL << eh) - 1));
    if (m) return 127 - __clzll((long long)m);
    return lo ? 63 - __clzll((long long)lo) : -1;
  }
  const unsigned long long m = (e >= 63) ? lo : (lo & ((2ULL << e) - 1));
  return m ? 63 - __clzll((long long)m) : -1;
}

template <bool EMIT>
__global__ void __launch_bounds__(64) gsketch_kernel(GSketchArgs a) {
  __shared__ uint64_t kbuf[128];
  const int lane = threadIdx.x;
  const int k = a.k, w = a.w;
  const uint64_t kbits = (k >= 32) ? 0xFFFFFFFFULL : ((1ULL << k) - 1);
  const unsigned long long below = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));
  const int span = w + k - 1;
  for (uint64_t chunk = blockIdx.x; chunk < a.n_chunks; chunk += gridDim.x) {
    if (EMIT && a.flags[chunk]) continue;
    const int c = chrom_of_chunk(a.chunk_first, a.n_chrom, chunk);
    const uint64_t S = a.chrom_pos[c];
    const uint32_t seqLen = (uint32_t)(a.chrom_pos[c + 1] - S);
    const unsigned char* seq = a.genome + S;
    const uint32_t nk = seqLen - k + 1;
    const uint32_t c0 = (uint32_t)(chunk - a.chunk_first[c]) * GCH, c1 = min(c0 + (uint32_t)GCH, nk);
    const uint32_t P0 = min((uint32_t)(2 * w), nk);
    const bool first = c0 == 0;
    const uint32_t B0 = first ? 0 : c0 - GWARM;
    const uint32_t statFrom = first ? P0 : B0 + w - 1;       // first position whose whole window is in hand
    const uint32_t liveFrom = first ? P0 : B0 + w;
    uint64_t* okey = EMIT ? a.out_key + a.out_off[chunk] : nullptr;
    uint32_t* opos = EMIT ? a.out_pos + a.out_off[chunk] : nullptr;
    uint32_t nout = 0;
    bool flagged = false;
    uint64_t carry_m = 0; uint32_t carry_act = 0; bool carry_ex = true;
    long lastN = -1;                                         // last non-ACGT position seen in earlier tiles
    for (uint32_t B = B0; B < c1; B += 64) {
      const uint32_t p = B + lane;
      int c0b = 0, c1b = 0; bool n0 = false, n1 = false;
      if (p < seqLen) { const int cc = km_code_n(seq[p]); n0 = cc > 3; c0b = cc & 3; if (n0) c0b = 0; }
      if (lane < k - 1 && p + 64 < seqLen) { const int cc = km_code_n(seq[p + 64]); n1 = cc > 3; c1b = n1 ? 0 : cc; }
      const unsigned long long b0 = __ballot(c0b & 1), b1 = __ballot(c0b & 2), t0 = __ballot(c1b & 1), t1 = __ballot(c1b & 2);
      const unsigned long long nb0 = __ballot(n0), nb1 = __ballot(n1);
      uint64_t x0 = b0 >> lane, x1 = b1 >> lane;
      if (lane) { x0 |= t0 << (64 - lane); x1 |= t1 << (64 - lane); }
      const uint64_t key = km_canonical(x0 & kbits, x1 & kbits, k);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      kbuf[p & 127] = key;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      // valid(q) for a position q of this tile: no non-ACGT byte in [q-w+1, q+k-1]; the last position of the sequence is also invalid when the
      // bases behind the last such byte are exactly one window (MinCount.h:117: `<`)
      auto valid_at = [&](uint32_t q) -> bool {
        const int ln = last_set_le(nb0, nb1, (int)(q - B) + k - 1);
        const long LN = ln >= 0 ? (long)B + ln : lastN;
        if (LN < 0) return true;
        if (LN > (long)q - w) return false;
        return !(q == nk - 1 && LN == (long)seqLen - span - 1);
      };
      if (first && B == 0) {
        // literal replay of positions 0 .. P0-1 on uniform values (the first window is chosen with the UNMASKED comparison, :91)
        uint64_t actT = kbuf[0]; uint32_t actP = 0;
        for (uint32_t q = 1; q < (uint32_t)w && q < nk; q++) { const uint64_t cc = kbuf[q]; if (cc < actT) { actT = cc; actP = q; } }
        if (last_set_le(nb0, nb1, span - 1) < 0) {                                      // :100-102 the first window holds no N
          if (lane == 0 && EMIT) { okey[nout] = actT; opos[nout] = (uint32_t)(S + actP); }
          nout++;
        }
        for (uint32_t q = w; q < P0; q++) {
          const uint64_t cc = kbuf[q];
          bool em = false;
          if (q - w >= actP) {
            uint64_t bt = 0; uint32_t bp = 0;
            for (int j = 0; j < w; j++) {
              const uint32_t x = q - ((q - (uint32_t)j) % (uint32_t)w);
              const uint64_t t = kbuf[x];
              if (j == 0 || (t & KM_FOR_MASK) < (bt & KM_FOR_MASK)) { bt = t; bp = x; }
            }
            actT = bt; actP = bp; em = true;
          } else if ((cc & KM_FOR_MASK) < (actT & KM_FOR_MASK)) { actT = cc; actP = q; em = true; }
          if (em && valid_at(q)) { if (lane == 0 && EMIT) { okey[nout] = actT; opos[nout] = (uint32_t)(S + actP); } nout++; }
        }
        carry_m = actT & KM_FOR_MASK; carry_act = actP; carry_ex = true;
      }
      const bool stat = p >= statFrom && p < nk;
      const bool live = p >= liveFrom && p < nk;
      uint64_t bk = key & KM_FOR_MASK; uint32_t bpos = p; int br = (int)(p % (uint32_t)w), cnt = 1;
      if (stat && !(first && p < P0)) {
        int rx = br;
        for (int d = 1; d < w; d++) {
          rx = (rx == 0) ? w - 1 : rx - 1;
          const uint32_t x = p - d;
          const uint64_t kx = kbuf[x & 127] & KM_FOR_MASK;
          if (kx < bk) { bk = kx; bpos = x; br = rx; cnt = 1; }
          else if (kx == bk) { cnt++; if (rx < br) { br = rx; bpos = x; } }
        }
      }
      uint64_t mprev = __shfl_up(bk, 1);
      const bool prevHas = lane > 0 && (p - 1) >= statFrom;
      if (!prevHas) mprev = carry_m;
      const bool strictNew = live && (key & KM_FOR_MASK) < mprev;
      uint32_t A; bool ex;
      if (!live) { A = bpos; ex = stat && cnt == 1; }                       // the chunk's first stat position: a guess, exact if the minimum is unique
      else if (cnt == 1 || strictNew) { A = strictNew ? p : bpos; ex = true; }
      else { A = UNK; ex = false; }
      unsigned long long unk = __ballot(live && A == UNK);
      while (unk) {
        const int l = __ffsll((long long)unk) - 1;
        unk &= unk - 1;
        uint32_t pa = __shfl(A, (l + 63) & 63);
        int pe = __shfl((int)ex, (l + 63) & 63);
        const uint32_t pl = B + l;
        if (l == 0 || pl - 1 < statFrom) { pa = carry_act; pe = carry_ex; }
        const uint32_t posR = __shfl(bpos, l);
        const uint32_t an = (pa == pl - w) ? posR : pa;
        if (lane == l) { A = an; ex = pe != 0; }
      }
      uint32_t prevA = __shfl_up(A, 1);
      int prevEx = __shfl_up((int)ex, 1);
      if (!prevHas) { prevA = carry_act; prevEx = carry_ex; }
      const bool inRange = live && p >= c0 && p < c1;
      const bool val = inRange && valid_at(p);
      const bool em = val && (prevA == p - (uint32_t)w || strictNew);
      if (__ballot(val && !strictNew && !prevEx)) flagged = true;           // would have to decide on a guessed state
      const unsigned long long me = __ballot(em);
      if (EMIT && em) {
        const uint32_t o = nout + __popcll(me & below);
        okey[o] = kbuf[A & 127]; opos[o] = (uint32_t)(S + A);
      }
      nout += __popcll(me);
      const unsigned long long ml = __ballot(stat);
      if (ml) {
        const int last = 63 - __clzll((long long)ml);
        carry_m = __shfl(bk, last); carry_act = __shfl(A, last); carry_ex = __shfl((int)ex, last) != 0;
      }
      if (nb0) lastN = (long)B + (63 - __clzll((long long)nb0));
      __builtin_amdgcn_wave_barrier();
    }
    if (!EMIT && lane == 0) { a.counts[chunk] = flagged ? 0 : nout; a.flags[chunk] = flagged ? 1 : 0; }
  }
}

// One lane per flagged chunk: the literal state machine, started `back` positions early with a guessed active minimizer; exact once a window
// with a unique minimum or a strictly smaller incoming k-mer is seen.  If a valid position of the chunk comes before that, back off 8x further;
// from the start of the chromosome the walk is the reference's own.
template <bool EMIT>
__global__ void __launch_bounds__(64) gsketch_serial_kernel(GSketchArgs a, const uint32_t* __restrict__ list, uint32_t n_list) {
  const uint32_t li = blockIdx.x * 64 + threadIdx.x;
  if (li >= n_list) return;
  const uint64_t chunk = list[li];
  const int k = a.k, w = a.w, span = w + k - 1;
  const int c = chrom_of_chunk(a.chunk_first, a.n_chrom, chunk);
  const uint64_t S = a.chrom_pos[c];
  const uint32_t seqLen = (uint32_t)(a.chrom_pos[c + 1] - S);
  const unsigned char* seq = a.genome + S;
  const uint32_t nk = seqLen - k + 1;
  const uint32_t c0 = (uint32_t)(chunk - a.chunk_first[c]) * GCH, c1 = min(c0 + (uint32_t)GCH, nk);
  const uint64_t kmask = (k >= 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
  uint64_t* okey = EMIT ? a.out_key + a.out_off[chunk] : nullptr;
  uint32_t* opos = EMIT ? a.out_pos + a.out_off[chunk] : nullptr;
  uint64_t ringT[32]; uint32_t ringP[32];
  for (uint64_t back = (uint64_t)GWARM * 8;; back *= 8) {
    const uint32_t s0 = (back >= c0) ? 0 : (uint32_t)(c0 - back);
    uint32_t nout = 0;
    uint64_t cur = 0, rc = 0;
    for (int x = 0; x < k; x++) cur = (cur << 2) + km_code2(seq[s0 + x]);
    { uint64_t t = cur; for (int i = 0; i < k; i++) { rc = (rc << 2) + ((~t) & 3ULL); t >>= 2; } }
    auto canon = [&]() -> uint64_t { return ((cur & KM_FOR_MASK) < (rc & KM_FOR_MASK)) ? (cur & KM_FOR_MASK) : (rc | KM_REV_MASK); };
    auto shift = [&](uint32_t at) {
      const uint64_t cc = km_code2(seq[at]);
      cur = ((cur << 2) & kmask) + cc;
      rc = (rc >> 2) + (((~cc) & 3ULL) << (2 * ((uint64_t)k - 1)));
    };
    long lastN = -1;
    for (uint32_t x = s0; x < s0 + (uint32_t)span && x < seqLen; x++) if (km_code_n(seq[x]) > 3) lastN = x;
    uint64_t actT = canon(); uint32_t actP = s0;
    ringT[s0 % w] = actT; ringP[s0 % w] = s0;
    uint32_t p;
    bool exact = s0 == 0;
    for (p = s0 + 1; p < s0 + (uint32_t)w && p < nk; p++) {
      shift(p + k - 1);
      const uint64_t cc = canon();
      if (s0 == 0 ? (cc < actT) : ((cc & KM_FOR_MASK) < (actT & KM_FOR_MASK))) { actT = cc; actP = p; }   // start of the sequence: unmasked (:91)
      ringT[p % w] = cc; ringP[p % w] = p;
    }
    if (s0 != 0) {
      // guessed state: the ring-order minimum of the first full window; exact if that minimum is unique
      uint64_t bt = ringT[0]; uint32_t bp = ringP[0]; int cnt = 1;
      for (int j = 1; j < w; j++) {
        const uint64_t t = ringT[j];
        if ((t & KM_FOR_MASK) < (bt & KM_FOR_MASK)) { bt = t; bp = ringP[j]; cnt = 1; }
        else if ((t & KM_FOR_MASK) == (bt & KM_FOR_MASK)) cnt++;
      }
      actT = bt; actP = bp; exact = cnt == 1;
    } else if (lastN < 0 && c0 == 0) {                                     // :100-102 the first minimizer, only when the first window holds no N
      if (EMIT) { okey[nout] = actT; opos[nout] = (uint32_t)(S + actP); }
      nout++;
    }
    bool failed = false;
    for (p = s0 + w; p < c1; p++) {
      const uint32_t e = p + k - 1;
      if (km_code_n(seq[e]) > 3) lastN = e;
      shift(e);
      const uint64_t cc = canon();
      ringT[p % w] = cc; ringP[p % w] = p;
      const bool val = p >= c0 && (lastN < 0 || (lastN <= (long)p - w && !(p == nk - 1 && lastN == (long)seqLen - span - 1)));
      bool em = false;
      if ((cc & KM_FOR_MASK) < (actT & KM_FOR_MASK) && !(p - w >= actP)) { actT = cc; actP = p; em = true; exact = true; }
      else {
        if (val && !exact) { failed = true; break; }
        if (p - w >= actP) {
          actT = ringT[0]; actP = ringP[0];
          for (int j = 1; j < w; j++) if ((ringT[j] & KM_FOR_MASK) < (actT & KM_FOR_MASK)) { actT = ringT[j]; actP = ringP[j]; }
          em = true;
        }
      }
      if (!exact) {                                   // a window with a unique minimum pins the state
        int cnt = 0; const uint64_t am = actT & KM_FOR_MASK; bool lower = false;
        for (int j = 0; j < w; j++) { const uint64_t t = ringT[j] & KM_FOR_MASK; if (t == am) cnt++; if (t < am) lower = true; }
        if (!lower && cnt == 1) exact = true;
      }
      if (em && val) { if (EMIT) { okey[nout] = actT; opos[nout] = (uint32_t)(S + actP); } nout++; }
    }
    if (failed && s0 != 0) continue;
    if (!EMIT) a.counts[chunk] = nout;
    return;
  }
}

__global__ void k_flag_list(uint64_t n, const uint32_t* __restrict__ flags, const uint64_t* __restrict__ off, uint32_t* __restrict__ list) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flags[i]) list[off[i]] = (uint32_t)i;
}

// ---- frequency filter + window thinning
__global__ void k_run_heads(uint64_t n, const uint64_t* __restrict__ key, uint32_t* __restrict__ head) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) head[i] = (i == 0 || (key[i] & KM_FOR_MASK) != (key[i - 1] & KM_FOR_MASK)) ? 1u : 0u;
}
__global__ void k_run_starts(uint64_t n, const uint32_t* __restrict__ head, const uint64_t* __restrict__ run_id, uint32_t* __restrict__ run_start, uint64_t n_runs) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && head[i]) run_start[run_id[i]] = (uint32_t)i;
  if (i == 0) run_start[n_runs] = (uint32_t)n;
}
// freq[i] = length of i's run; cand[i] = 1 when the run is at most maxFreq long
__global__ void k_freq(uint64_t n, const uint32_t* __restrict__ head, const uint64_t* __restrict__ run_id, const uint32_t* __restrict__ run_start, int maxFreq,
                       uint32_t* __restrict__ freq, uint32_t* __restrict__ cand) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t r = run_id[i] + head[i] - 1;                              // exclusive scan of heads -> id of i's run
  const uint32_t f = run_start[r + 1] - run_start[r];
  freq[i] = f; cand[i] = f <= (uint32_t)maxFreq ? 1u : 0u;
}
// candidates in DESCENDING sorted position: slot U-1-rank
__global__ void k_cand_emit(uint64_t n, const uint32_t* __restrict__ cand, const uint64_t* __restrict__ rank, uint64_t U, const uint32_t* __restrict__ pos,
                            const uint32_t* __restrict__ freq, uint32_t winsize, int fbits, uint64_t* __restrict__ wkey, uint32_t* __restrict__ widx) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !cand[i]) return;
  const uint64_t j = U - 1 - rank[i];
  wkey[j] = ((uint64_t)(pos[i] / winsize) << fbits) | (uint64_t)freq[i];
  widx[j] = (uint32_t)i;
}
__global__ void k_thin(uint64_t U, const uint64_t* __restrict__ wkey, const uint32_t* __restrict__ widx, int fbits, int nPerWin, uint64_t sz, uint32_t* __restrict__ keep,
                       uint32_t* __restrict__ oob) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= U) return;
  const uint64_t win = wkey[j] >> fbits;
  if (win >= sz) *oob = 1;                                                 // the reference indexes winCount out of range here (MMIndex.h:366)
  if (j < (uint64_t)nPerWin || (wkey[j - nPerWin] >> fbits) != win) keep[widx[j]] = 1;
}
__global__ void k_compact(uint64_t n, const uint32_t* __restrict__ keep, const uint64_t* __restrict__ off, const uint64_t* __restrict__ key, const uint32_t* __restrict__ pos,
                          uint64_t* __restrict__ okey, uint32_t* __restrict__ opos) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && keep[i]) { okey[off[i]] = key[i]; opos[off[i]] = pos[i]; }
}

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  template <typename T> T* get(size_t n) { if (p) { (void)hipFree(p); p = nullptr; } if (hipMalloc(&p, n * sizeof(T) + 256) != hipSuccess) p = nullptr; return (T*)p; }
  void* release() { void* r = p; p = nullptr; return r; }
};

dim3 grid1(uint64_t n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

extern "C" int lra_ctx_build_global_index(lra_ctx* ctx, const uint64_t* h_chrom_pos, int n_chrom, int k, int w, int maxFreq, int winsize, int nPerWin,
                                          uint64_t* n_minimizers, uint64_t* n_index, int* status) {
  if (!ctx || !h_chrom_pos || n_chrom < 1 || k < 1 || k > 31 || w < 1 || w > 32 || maxFreq < 1 || winsize < 1 || nPerWin < 1) return LRA_ERR_INVALID;
  if (!ctx->seed || !ctx->seed->genome) return lra_set_err(ctx, LRA_ERR_INVALID, "load the genome first");
  if (h_chrom_pos[n_chrom] != ctx->seed->genome_len || h_chrom_pos[n_chrom] >= (1ULL << 32)) return lra_set_err(ctx, LRA_ERR_INVALID, "chromosome table does not cover the genome (or >= 4 G bases)");
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  if (status) *status = 0;
  const int span = w + k - 1;
  std::vector<uint64_t> chunk_first(n_chrom + 1, 0);
  for (int c = 0; c < n_chrom; c++) {
    const uint64_t len = h_chrom_pos[c + 1] - h_chrom_pos[c];
    const uint64_t nk = (len >= (uint64_t)k && len > (uint64_t)span) ? len - k + 1 : 0;    // MinCount.h:12, :26-41
    chunk_first[c + 1] = chunk_first[c] + (nk + GCH - 1) / GCH;
  }
  const uint64_t n_chunks = chunk_first[n_chrom];
  DevBuf b_cp, b_cf, b_cnt, b_flag, b_off, b_list;
  uint64_t* d_cp = b_cp.get<uint64_t>(n_chrom + 1); uint64_t* d_cf = b_cf.get<uint64_t>(n_chrom + 1);
  uint32_t* d_cnt = b_cnt.get<uint32_t>(n_chunks + 1); uint32_t* d_flag = b_flag.get<uint32_t>(n_chunks + 1); uint64_t* d_off = b_off.get<uint64_t>(n_chunks + 2);
  if (!d_cp || !d_cf || !d_cnt || !d_flag || !d_off) return lra_set_err(ctx, LRA_ERR_NOMEM, "index build: chunk tables");
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(d_cp, h_chrom_pos, (size_t)(n_chrom + 1) * 8, hipMemcpyHostToDevice, st));
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(d_cf, chunk_first.data(), (size_t)(n_chrom + 1) * 8, hipMemcpyHostToDevice, st));
  GSketchArgs a{}; a.genome = ctx->seed->genome; a.chrom_pos = d_cp; a.n_chrom = n_chrom; a.chunk_first = d_cf; a.n_chunks = n_chunks; a.k = k; a.w = w;
  a.counts = d_cnt; a.flags = d_flag;
  uint64_t N = 0;
  DevBuf b_key, b_pos;
  uint64_t* mkey = nullptr; uint32_t* mpos = nullptr;
  if (n_chunks) {
    const unsigned gw = (unsigned)std::min<uint64_t>(n_chunks, (uint64_t)ctx->num_cu * 64);
    lra_time_begin(ctx, "gsketch_count");
    hipLaunchKernelGGL(gsketch_kernel<false>, dim3(gw), dim3(64), 0, st, a);
    lra_time_end(ctx);
    // the chunks that need the serial walk
    int rc = lra_exclusive_scan<uint32_t>(ctx, (long)n_chunks, d_flag, d_off);
    if (rc) return rc;
    uint64_t n_flag = 0;
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(&n_flag, d_off + n_chunks, 8, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
    uint32_t* d_list = nullptr;
    if (n_flag) {
      d_list = b_list.get<uint32_t>(n_flag + 1);
      if (!d_list) return lra_set_err(ctx, LRA_ERR_NOMEM, "index build: flagged chunks");
      hipLaunchKernelGGL(k_flag_list, grid1(n_chunks), dim3(256), 0, st, n_chunks, d_flag, d_off, d_list);
      lra_time_begin(ctx, "gsketch_serial");
      hipLaunchKernelGGL(gsketch_serial_kernel<false>, dim3((unsigned)((n_flag + 63) / 64)), dim3(64), 0, st, a, d_list, (uint32_t)n_flag);
      lra_time_end(ctx);
    }
    if ((rc = lra_exclusive_scan<uint32_t>(ctx, (long)n_chunks, d_cnt, d_off))) return rc;
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(&N, d_off + n_chunks, 8, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (N >= (1ULL << 32)) return lra_set_err(ctx, LRA_ERR_INVALID, "more than 2^32 minimizers");
    mkey = b_key.get<uint64_t>(N + 1); mpos = b_pos.get<uint32_t>(N + 1);
    if (!mkey || !mpos) return lra_set_err(ctx, LRA_ERR_NOMEM, "index build: minimizers");
    a.out_off = d_off; a.out_key = mkey; a.out_pos = mpos;
    lra_time_begin(ctx, "gsketch_emit");
    hipLaunchKernelGGL(gsketch_kernel<true>, dim3(gw), dim3(64), 0, st, a);
    lra_time_end(ctx);
    if (n_flag) hipLaunchKernelGGL(gsketch_serial_kernel<true>, dim3((unsigned)((n_flag + 63) / 64)), dim3(64), 0, st, a, d_list, (uint32_t)n_flag);
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
    LRA_HIP_CHECK(ctx, hipGetLastError());
  }
  if (n_minimizers) *n_minimizers = N;
  if (N == 0) { if (n_index) *n_index = 0; return lra_seed_install_index(ctx, nullptr, nullptr, 0); }
  // (2) stable sort by masked key
  DevBuf b_key2, b_pos2, b_tmp;
  uint64_t* skey = b_key2.get<uint64_t>(N + 1); uint32_t* spos = b_pos2.get<uint32_t>(N + 1);
  if (!skey || !spos) return lra_set_err(ctx, LRA_ERR_NOMEM, "index build: sort buffers");
  {
    size_t tb = 0;
    LRA_HIP_CHECK(ctx, rocprim::radix_sort_pairs(nullptr, tb, mkey, skey, mpos, spos, (size_t)N, 0, 2 * k, st));
    void* tmp = b_tmp.get<char>(tb + 256);
    if (!tmp) return lra_set_err(ctx, LRA_ERR_NOMEM, "index build: sort scratch");
    lra_time_begin(ctx, "gindex_sort");
    LRA_HIP_CHECK(ctx, rocprim::radix_sort_pairs(tmp, tb, mkey, skey, mpos, spos, (size_t)N, 0, 2 * k, st));   // the strand bit (63) is not part of the order
    lra_time_end(ctx);
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  }
  b_key.get<char>(0); b_pos.get<char>(0);                                  // the unsorted copies are dead
  // (3) frequency per key run
  DevBuf b_head, b_rid, b_rs, b_freq, b_cand, b_rank;
  uint32_t* head = b_head.get<uint32_t>(N + 1); uint64_t* rid = b_rid.get<uint64_t>(N + 2);
  if (!head || !rid) return lra_set_err(ctx, LRA_ERR_NOMEM, "index build: runs");
  hipLaunchKernelGGL(k_run_heads, grid1(N), dim3(256), 0, st, N, skey, head);
  int rc = lra_exclusive_scan<uint32_t>(ctx, (long)N, head, rid);
  if (rc) return rc;
  uint64_t n_runs = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&n_runs, rid + N, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  uint32_t* run_start = b_rs.get<uint32_t>(n_runs + 2); uint32_t* freq = b_freq.get<uint32_t>(N + 1); uint32_t* cand = b_cand.get<uint32_t>(N + 1);
  uint64_t* rank = b_rank.get<uint64_t>(N + 2);
  if (!run_start || !freq || !cand || !rank) return lra_set_err(ctx, LRA_ERR_NOMEM, "index build: frequencies");
  hipLaunchKernelGGL(k_run_starts, grid1(N), dim3(256), 0, st, N, head, rid, run_start, n_runs);
  hipLaunchKernelGGL(k_freq, grid1(N), dim3(256), 0, st, N, head, rid, run_start, maxFreq, freq, cand);
  if ((rc = lra_exclusive_scan<uint32_t>(ctx, (long)N, cand, rank))) return rc;
  uint64_t U = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&U, rank + N, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  b_rs.get<char>(0); b_rid.get<char>(0);
  uint32_t* keep = head;                                                  // reuse
  LRA_HIP_CHECK(ctx, hipMemsetAsync(keep, 0, (N + 1) * 4, st));
  uint64_t n_out = 0;
  if (U) {
    int fbits = 1;
    while ((1 << fbits) <= maxFreq) fbits++;
    const uint64_t G = h_chrom_pos[n_chrom];
    uint64_t sz = G / (uint64_t)winsize;                                   // MMIndex.h:359-360
    if (G / (uint64_t)winsize % (uint64_t)winsize > 0) sz += 1;
    int wbits = 1;
    while (wbits < 40 && ((G / (uint64_t)winsize + 1) >> wbits)) wbits++;
    DevBuf b_wk, b_wi, b_wk2, b_wi2, b_t2, b_oob;
    uint64_t* wkey = b_wk.get<uint64_t>(U + 1); uint32_t* widx = b_wi.get<uint32_t>(U + 1);
    uint64_t* wkey2 = b_wk2.get<uint64_t>(U + 1); uint32_t* widx2 = b_wi2.get<uint32_t>(U + 1);
    uint32_t* oob = b_oob.get<uint32_t>(4);
    if (!wkey || !widx || !wkey2 || !widx2 || !oob) return lra_set_err(ctx, LRA_ERR_NOMEM, "index build: thinning");
    LRA_HIP_CHECK(ctx, hipMemsetAsync(oob, 0, 4, st));
    hipLaunchKernelGGL(k_cand_emit, grid1(N), dim3(256), 0, st, N, cand, rank, U, spos, freq, (uint32_t)winsize, fbits, wkey, widx);
    size_t tb = 0;
    LRA_HIP_CHECK(ctx, rocprim::radix_sort_pairs(nullptr, tb, wkey, wkey2, widx, widx2, (size_t)U, 0, fbits + wbits, st));
    void* tmp = b_t2.get<char>(tb + 256);
    if (!tmp) return lra_set_err(ctx, LRA_ERR_NOMEM, "index build: sort scratch");
    lra_time_begin(ctx, "gindex_window_sort");
    LRA_HIP_CHECK(ctx, rocprim::radix_sort_pairs(tmp, tb, wkey, wkey2, widx, widx2, (size_t)U, 0, fbits + wbits, st));
    lra_time_end(ctx);
    hipLaunchKernelGGL(k_thin, grid1(U), dim3(256), 0, st, U, wkey2, widx2, fbits, nPerWin, sz, keep, oob);
    uint32_t h_oob = 0;
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(&h_oob, oob, 4, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (h_oob && status) *status = LRA_ST_OOB_SLOT;
  }
  if ((rc = lra_exclusive_scan<uint32_t>(ctx, (long)N, keep, rank))) return rc;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&n_out, rank + N, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  DevBuf b_ok, b_op;
  uint64_t* okey = b_ok.get<uint64_t>(n_out + 2); uint32_t* opos = b_op.get<uint32_t>(n_out + 2);
  if (!okey || !opos) return lra_set_err(ctx, LRA_ERR_NOMEM, "index build: output");
  hipLaunchKernelGGL(k_compact, grid1(N), dim3(256), 0, st, N, keep, rank, skey, spos, okey, opos);
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  LRA_HIP_CHECK(ctx, hipGetLastError());
  if (n_index) *n_index = n_out;
  rc = lra_seed_install_index(ctx, okey, opos, n_out);
  if (rc == LRA_OK) { b_ok.release(); b_op.release(); }
  return rc;
}

// ---------------------------------------------------------------------------------------------------------------- files (host)
namespace {
struct File {
  FILE* f;
  explicit File(const char* path, const char* mode) : f(fopen(path, mode)) {}
  ~File() { if (f) fclose(f); }
  bool wr(const void* p, size_t n) { return n == 0 || fwrite(p, 1, n, f) == n; }
  bool rd(void* p, size_t n) { return n == 0 || fread(p, 1, n, f) == n; }
};
}  // namespace

// WriteIndex (MMIndex.h:416-424): int64 n; int32 globalK; Header (int32 nChrom; per chromosome int32 nameLen + bytes; uint64 pos[nChrom+1]); then n
// GenomeTuples of 16 bytes (uint64 t, uint32 pos, 4 bytes of padding -- written as zeros here; the reference writes whatever the vector holds).
extern "C" int lra_write_mms(const char* path, int globalK, const char* const* chrom_names, const uint64_t* chrom_pos, int n_chrom, const uint64_t* key,
                             const uint32_t* pos, uint64_t n) {
  if (!path || !chrom_names || !chrom_pos || n_chrom < 0 || (n && (!key || !pos))) return LRA_ERR_INVALID;
  File f(path, "wb");
  if (!f.f) return LRA_ERR_INVALID;
  const int64_t len = (int64_t)n; const int32_t K = globalK, nc = n_chrom;
  bool ok = f.wr(&len, 8) && f.wr(&K, 4) && f.wr(&nc, 4);
  for (int i = 0; ok && i < n_chrom; i++) { const int32_t l = (int32_t)strlen(chrom_names[i]); ok = f.wr(&l, 4) && f.wr(chrom_names[i], (size_t)l); }
  ok = ok && f.wr(chrom_pos, (size_t)(n_chrom + 1) * 8);
  std::vector<uint64_t> buf;
  const uint64_t step = 1 << 20;
  for (uint64_t i = 0; ok && i < n; i += step) {
    const uint64_t m = std::min(step, n - i);
    buf.assign(2 * m, 0);
    for (uint64_t j = 0; j < m; j++) { buf[2 * j] = key[i + j]; buf[2 * j + 1] = (uint64_t)pos[i + j]; }
    ok = f.wr(buf.data(), (size_t)m * 16);
  }
  return ok ? LRA_OK : LRA_ERR_INVALID;
}

// ReadIndex (MMIndex.h:402-414).  Two-call convention: with key == NULL only *n, *globalK, *n_chrom and *names_len (bytes needed for the names,
// each NUL-terminated) are returned; the second call fills key[n], pos[n], chrom_pos[n_chrom + 1] and names.
extern "C" int lra_read_mms(const char* path, int* globalK, uint64_t* n, int* n_chrom, uint64_t* names_len, char* names, uint64_t* chrom_pos, uint64_t* key,
                            uint32_t* pos) {
  if (!path || !n || !n_chrom) return LRA_ERR_INVALID;
  // the filling call (key != NULL) takes what the sizing call returned in *n / *n_chrom / *names_len as the capacities of the caller's buffers: a file that has
  // changed in between (or is corrupt) is refused instead of written past them
  const bool fill = key != nullptr;
  const uint64_t capN = *n; const int capC = *n_chrom; const uint64_t capNames = names_len ? *names_len : 0;
  File f(path, "rb");
  if (!f.f) return LRA_ERR_INVALID;
  int64_t len = 0; int32_t K = 0, nc = 0;
  if (!f.rd(&len, 8) || !f.rd(&K, 4) || !f.rd(&nc, 4) || len < 0 || nc < 0 || nc > (1 << 24)) return LRA_ERR_INVALID;
  if (fill && ((uint64_t)len > capN || nc > capC)) return LRA_ERR_INVALID;
  if (globalK) *globalK = K;
  *n = (uint64_t)len; *n_chrom = nc;
  uint64_t nl = 0;
  std::string all;
  for (int i = 0; i < nc; i++) {
    int32_t l = 0;
    if (!f.rd(&l, 4) || l < 0 || l > (1 << 20)) return LRA_ERR_INVALID;   // (a sequence name of a megabyte is a corrupt file)
    std::string s((size_t)l, '\0');
    if (!f.rd(&s[0], (size_t)l)) return LRA_ERR_INVALID;
    all += s; all.push_back('\0'); nl += (uint64_t)l + 1;
  }
  if (names_len) *names_len = nl;
  if (!key) return LRA_OK;
  if (!pos || !chrom_pos || !names || !names_len || all.size() > capNames) return LRA_ERR_INVALID;
  memcpy(names, all.data(), all.size());
  if (!f.rd(chrom_pos, (size_t)(nc + 1) * 8)) return LRA_ERR_INVALID;
  std::vector<uint64_t> buf;
  const uint64_t step = 1 << 20;
  for (uint64_t i = 0; i < (uint64_t)len; i += step) {
    const uint64_t m = std::min(step, (uint64_t)len - i);
    buf.resize(2 * m);
    if (!f.rd(buf.data(), (size_t)m * 16)) return LRA_ERR_INVALID;
    for (uint64_t j = 0; j < m; j++) { key[i + j] = buf[2 * j]; pos[i + j] = (uint32_t)buf[2 * j + 1]; }
  }
  return LRA_OK;
}

// LocalIndex::Write (MMIndex.h:138-151): int32 k, w, localIndexWindow, nRegions (= n_windows + 1); uint64 seqOffsets[nRegions]; uint64
// tupleBoundaries[nRegions]; uint64 nMin; LocalTuple[nMin] (uint32: t in the low 20 bits, pos in the high 12, TupleOps.h:20-47).
extern "C" int lra_write_gli(const char* path, int k, int w, int window, uint64_t n_windows, const uint64_t* seq_offsets, const uint64_t* tuple_bnd,
                             const uint32_t* tuples) {
  if (!path || !seq_offsets || !tuple_bnd) return LRA_ERR_INVALID;
  File f(path, "wb");
  if (!f.f) return LRA_ERR_INVALID;
  const int32_t h[4] = {k, w, window, (int32_t)(n_windows + 1)};
  const uint64_t nMin = tuple_bnd[n_windows];
  const bool ok = f.wr(h, 16) && f.wr(seq_offsets, (size_t)(n_windows + 1) * 8) && f.wr(tuple_bnd, (size_t)(n_windows + 1) * 8) && f.wr(&nMin, 8) &&
                  f.wr(tuples, (size_t)nMin * 4);
  return ok ? LRA_OK : LRA_ERR_INVALID;
}

// LocalIndex::Read (MMIndex.h:153-173).  Two-call convention: with seq_offsets == NULL only k, w, window, n_windows and n_tuples come back.
extern "C" int lra_read_gli(const char* path, int* k, int* w, int* window, uint64_t* n_windows, uint64_t* n_tuples, uint64_t* seq_offsets, uint64_t* tuple_bnd,
                            uint32_t* tuples) {
  if (!path || !n_windows || !n_tuples) return LRA_ERR_INVALID;
  File f(path, "rb");
  if (!f.f) return LRA_ERR_INVALID;
  int32_t h[4];
  if (!f.rd(h, 16) || h[3] < 1) return LRA_ERR_INVALID;
  if (k) *k = h[0]; if (w) *w = h[1]; if (window) *window = h[2];
  const uint64_t nr = (uint64_t)h[3];
  const uint64_t capW = *n_windows, capT = *n_tuples;                    // filling call: the sizing call's results = the capacities of the caller's buffers
  if (seq_offsets && nr - 1 > capW) return LRA_ERR_INVALID;
  *n_windows = nr - 1;
  if (!seq_offsets) {
    if (fseek(f.f, (long)(16 + nr * 16), SEEK_SET) != 0) return LRA_ERR_INVALID;
    uint64_t nMin = 0;
    if (!f.rd(&nMin, 8)) return LRA_ERR_INVALID;
    *n_tuples = nMin;
    return LRA_OK;
  }
  if (!tuple_bnd || !tuples) return LRA_ERR_INVALID;
  uint64_t nMin = 0;
  if (!f.rd(seq_offsets, (size_t)nr * 8) || !f.rd(tuple_bnd, (size_t)nr * 8) || !f.rd(&nMin, 8) || nMin > capT || !f.rd(tuples, (size_t)nMin * 4)) return LRA_ERR_INVALID;
  *n_tuples = nMin;
  return LRA_OK;
}
