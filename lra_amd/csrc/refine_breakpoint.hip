// lra_amd/csrc/refine_breakpoint.hip -- SURVEY §8a row a15: RefineBreakpoint (RefineBreakpoint.h:210-466) for a batch of junctions between
// two adjacent segments of split alignments.  gfx950 only.
//   RSdp :150-195 (full DP, match 2 / mismatch -2 / gap -4, first-maximum arrow order diag, left, down), FindMax :197 (first maximum in
//   row-major order), StoreQScoreVect :118, TraceBack :91, PathToBlocks :49, PrependBlocks :6, AppendBlocks :29.
// Mapping: one wave per junction.  The two (span + 1) x (tLen + 1) matrices (span < 500) live in HBM; the fill runs over anti-diagonals,
// 64 cells at a time; the maximum is a wave reduction on (score, smallest index); the merge of the two local alignments, the trace
// backs and the block glue are short serial walks done by lane 0.  Rare path (split alignments only): sized for correctness, not speed.
// Algorithmic bytes: 5 B per DP cell written + read once.
#include "common.h"
#include "scan.h"

namespace {

enum { LEFT = 1, DOWN = 2, DIAG = 3 };

struct RbArgs {
  int n;
  const int32_t* read_len; const char* seq; const char* genome;
  const int32_t* lBlocks; const uint64_t* lOff; const int32_t* lStrand; const uint64_t* lReadOff; const uint64_t* lChromOff; const int32_t* lChromLen;
  const int32_t* rBlocks; const uint64_t* rOff; const int32_t* rStrand; const uint64_t* rReadOff; const uint64_t* rChromOff; const int32_t* rChromLen;
  // per junction plan (written by rb_plan)
  int32_t* span; int32_t* ltLen; int32_t* rtLen; uint8_t* lPrefix; uint8_t* rPrefix; uint64_t* lqAt; uint64_t* ltAt; uint64_t* rqAt; uint64_t* rtAt;
  uint32_t* cells; const uint64_t* cellOff; int32_t* score; uint8_t* path; int32_t* tb;      // matrices: left at cellOff[2j], right at cellOff[2j+1]; tb: 2 * 1002 per junction
  int32_t* lOut; int32_t* rOut; const uint64_t* lOutOff; const uint64_t* rOutOff; int32_t* nLOut; int32_t* nROut; uint32_t* status;
};

// which junctions are refined, the four substrings (start address and direction), matrix sizes   (:216-352)
__global__ void rb_plan(RbArgs a) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= a.n) return;
  const int nL = (int)(a.lOff[j + 1] - a.lOff[j]), nR = (int)(a.rOff[j + 1] - a.rOff[j]);
  const int32_t* L = a.lBlocks + 3 * a.lOff[j]; const int32_t* R = a.rBlocks + 3 * a.rOff[j];
  const int lqs = nL ? L[0] : 0, lqe = nL ? L[3 * (nL - 1)] + L[3 * (nL - 1) + 2] : 0, lts = nL ? L[1] : 0, lte = nL ? L[3 * (nL - 1) + 1] + L[3 * (nL - 1) + 2] : 0;
  const int rqs = nR ? R[0] : 0, rqe = nR ? R[3 * (nR - 1)] + R[3 * (nR - 1) + 2] : 0, rts = nR ? R[1] : 0, rte = nR ? R[3 * (nR - 1) + 1] + R[3 * (nR - 1) + 2] : 0;
  const int readLen = a.read_len[j];
  const int flqe = a.lStrand[j] == 0 ? lqe : readLen - lqs;
  const int frqs = a.rStrand[j] == 0 ? rqs : readLen - rqe;
  uint32_t st = 0;
  int span = 0, ltLen = 0, rtLen = 0;
  a.cells[2 * j] = 0; a.cells[2 * j + 1] = 0;
  if (frqs > flqe && frqs - flqe < 500) {
    span = frqs - flqe;
    if (a.lStrand[j] == 0) {
      if (lqe + span > readLen || a.lChromLen[j] - lte < 0) st = LRA_ST_OOB_SLOT;
      a.lqAt[j] = a.lReadOff[j] + lqe; ltLen = min(a.lChromLen[j] - lte, span); a.ltAt[j] = a.lChromOff[j] + lte; a.lPrefix[j] = 0;
    } else {
      if (lqs - span < 0) st = LRA_ST_OOB_SLOT;
      const int tS = max(0, lts - span);
      ltLen = lts - tS;
      a.lqAt[j] = a.lReadOff[j] + (lqs - span); a.ltAt[j] = a.lChromOff[j] + tS; a.lPrefix[j] = 1;   // both strings are read backwards
    }
    if (a.rStrand[j] == 0) {
      if (rqs - span < 0) st = LRA_ST_OOB_SLOT;
      rtLen = min(rts, span);
      a.rqAt[j] = a.rReadOff[j] + (rqs - span); a.rtAt[j] = a.rChromOff[j] + (rts - rtLen); a.rPrefix[j] = 1;
    } else {
      if (rqe + span > readLen) st = LRA_ST_OOB_SLOT;
      rtLen = span;
      if (rte + span >= a.rChromLen[j]) rtLen = a.rChromLen[j] - rte;
      if (rtLen < 0) st = LRA_ST_OOB_SLOT;
      a.rqAt[j] = a.rReadOff[j] + rqe; a.rtAt[j] = a.rChromOff[j] + rte; a.rPrefix[j] = 0;
    }
    if (st) span = 0;
    else { a.cells[2 * j] = (uint32_t)((span + 1) * (ltLen + 1)); a.cells[2 * j + 1] = (uint32_t)((span + 1) * (rtLen + 1)); }
  }
  a.span[j] = span; a.ltLen[j] = ltLen; a.rtLen[j] = rtLen; a.status[j] = st;
}

// RSdp over anti-diagonals; q / t are read forwards or (prefix extension) backwards
__device__ void rsdp_wave(const char* q, int qs, const char* t, int ts, bool rev, int32_t* score, uint8_t* path, int lane) {
  const int row = qs + 1;
  for (int i = lane; i < (qs + 1) * (ts + 1); i += 64) { score[i] = 0; path[i] = 0xFF; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  for (int i = 1 + lane; i < qs + 1; i += 64) { path[i] = LEFT; score[i] = -4 * i; }
  for (int i = 1 + lane; i < ts + 1; i += 64) { path[row * i] = DOWN; score[row * i] = -4 * i; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  for (int d = 0; d <= qs + ts - 2; d++) {                               // cells (i, j) of the strings with i + j = d
    const int ilo = max(0, d - (qs - 1)), ihi = min(ts - 1, d);
    for (int i = ilo + lane; i <= ihi; i += 64) {
      const int j = d - i;
      const char qc = rev ? q[qs - 1 - j] : q[j], tc = rev ? t[ts - 1 - i] : t[i];
      const int diagScore = score[i * row + j] + (qc == tc ? 2 : -2);
      const int leftScore = score[(i + 1) * row + j] - 4, downScore = score[i * row + (j + 1)] - 4;
      const int mx = max(diagScore, max(leftScore, downScore));
      score[(i + 1) * row + (j + 1)] = mx;
      path[(i + 1) * row + (j + 1)] = mx == diagScore ? DIAG : mx == leftScore ? LEFT : DOWN;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
}

__device__ void find_max_wave(const int32_t* score, int cells, int row, int lane, int& q, int& t) {   // FindMax :197-207
  int best = INT_MIN, bi = 0x7fffffff;
  for (int i = lane; i < cells; i += 64) if (score[i] > best) { best = score[i]; bi = i; }   // per lane: first index of its maximum (indices ascend)
  for (int o = 32; o > 0; o >>= 1) {
    const int ob = __shfl_xor(best, o), oi = __shfl_xor(bi, o);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  t = bi / row - 1; q = bi % row - 1;
}

__device__ int store_q(const int32_t* score, const uint8_t* path, int q, int t, int r, int32_t* qv, int32_t* index) {   // :118-146 (qv, index: r - 1 entries, zeroed)
  for (int x = 0; x < r - 1; x++) { qv[x] = 0; index[x] = 0; }
  int i = (t + 1) * r + q + 1;
  q++; t++;
  while (i > 0) {
    const int p = path[i];
    if (p == DIAG || p == LEFT) { qv[q - 1] = score[i]; index[q - 1] = i; }
    if (p == DIAG) { q--; t--; }
    if (p == LEFT) q--;
    if (p == DOWN) t--;
    i = t * r + q;
  }
  return 0;
}

// TraceBack :91-116 into tb[] back to front (so tb[0..n) ends up in forward order), returns n
__device__ int trace_back(const uint8_t* path, int q, int t, int r, int32_t* tb, int cap) {
  q++; t++;
  int i = t * r + q, n = 0;
  while ((q > 0 || t > 0) && n < cap) {
    const int p = path[i];
    if (p == DIAG) { q--; t--; tb[n++] = DIAG; }
    if (p == LEFT) { q--; tb[n++] = LEFT; }
    if (p == DOWN) { t--; tb[n++] = DOWN; }
    i = t * r + q;
  }
  for (int x = 0, y = n - 1; x < y; x++, y--) { const int v = tb[x]; tb[x] = tb[y]; tb[y] = v; }   // reverse(tb)
  return n;
}

// PathToBlocks :49-82 into out (triples), returns the number of blocks
__device__ int path_to_blocks(const int32_t* path, int n, int32_t* out, int qAdd, int tAdd) {
  int i = 0, q = 0, t = 0, nb = 0;
  while (i < n && path[i] != DIAG && (path[i] == LEFT || path[i] == DOWN)) { if (path[i] == LEFT) q++; if (path[i] == DOWN) t++; i++; }
  while (i < n) {
    const int qs = q, ts = t;
    while (i < n && path[i] == DIAG) { q++; t++; i++; }
    while (i < n && (path[i] == LEFT || path[i] == DOWN)) { if (path[i] == LEFT) q++; if (path[i] == DOWN) t++; i++; }
    const int match = min(q - qs, t - ts);
    if (match > 0) { out[3 * nb] = qs + qAdd; out[3 * nb + 1] = ts + tAdd; out[3 * nb + 2] = match; nb++; }
  }
  return nb;
}

// PrependBlocks / AppendBlocks (:6-46): dest (nd blocks) + src (ns blocks, in `src`) -> out; returns the new count
__device__ int glue(const int32_t* dest, int nd, int32_t* src, int ns, bool prepend, int32_t* out) {
  int n = 0;
  auto put = [&](int q, int t, int l) { out[3 * n] = q; out[3 * n + 1] = t; out[3 * n + 2] = l; n++; };
  if (ns == 0) { for (int b = 0; b < nd; b++) put(dest[3 * b], dest[3 * b + 1], dest[3 * b + 2]); return n; }
  if (nd == 0) { for (int b = 0; b < ns; b++) put(src[3 * b], src[3 * b + 1], src[3 * b + 2]); return n; }
  if (prepend) {
    const int last = ns - 1;
    int d0q = dest[0], d0t = dest[1], d0l = dest[2];
    if (src[3 * last + 1] + src[3 * last + 2] == d0t && src[3 * last] + src[3 * last + 2] == d0q) { d0t -= src[3 * last + 2]; d0q -= src[3 * last + 2]; d0l += src[3 * last + 2]; ns = last; }
    for (int b = 0; b < ns; b++) put(src[3 * b], src[3 * b + 1], src[3 * b + 2]);
    put(d0q, d0t, d0l);
    for (int b = 1; b < nd; b++) put(dest[3 * b], dest[3 * b + 1], dest[3 * b + 2]);
  } else {
    const int last = nd - 1;
    int srcStart = 0, ll = dest[3 * last + 2];
    if (dest[3 * last + 1] + ll == src[1] && dest[3 * last] + ll == src[0]) { ll += src[2]; srcStart = 1; }
    for (int b = 0; b < last; b++) put(dest[3 * b], dest[3 * b + 1], dest[3 * b + 2]);
    put(dest[3 * last], dest[3 * last + 1], ll);
    for (int b = srcStart; b < ns; b++) put(src[3 * b], src[3 * b + 1], src[3 * b + 2]);
  }
  return n;
}

__global__ void __launch_bounds__(64) rb_kernel(RbArgs a) {
  const int j = blockIdx.x, lane = threadIdx.x;
  const int nL = (int)(a.lOff[j + 1] - a.lOff[j]), nR = (int)(a.rOff[j + 1] - a.rOff[j]);
  const int32_t* L = a.lBlocks + 3 * a.lOff[j]; const int32_t* R = a.rBlocks + 3 * a.rOff[j];
  int32_t* lOut = a.lOut + 3 * a.lOutOff[j]; int32_t* rOut = a.rOut + 3 * a.rOutOff[j];
  const int span = a.span[j];
  if (span == 0) {                                                       // left alone (or flagged)
    for (int x = lane; x < 3 * nL; x += 64) lOut[x] = L[x];
    for (int x = lane; x < 3 * nR; x += 64) rOut[x] = R[x];
    if (lane == 0) { a.nLOut[j] = nL; a.nROut[j] = nR; }
    return;
  }
  const int ltLen = a.ltLen[j], rtLen = a.rtLen[j], row = span + 1;
  const bool lPrefix = a.lPrefix[j], rPrefix = a.rPrefix[j];
  int32_t* lScore = a.score + a.cellOff[2 * j]; uint8_t* lPath = a.path + a.cellOff[2 * j];
  int32_t* rScore = a.score + a.cellOff[2 * j + 1]; uint8_t* rPath = a.path + a.cellOff[2 * j + 1];
  rsdp_wave(a.seq + a.lqAt[j], span, a.genome + a.ltAt[j], ltLen, lPrefix, lScore, lPath, lane);
  rsdp_wave(a.seq + a.rqAt[j], span, a.genome + a.rtAt[j], rtLen, rPrefix, rScore, rPath, lane);
  int mlq, mlt, mrq, mrt;
  find_max_wave(lScore, row * (ltLen + 1), row, lane, mlq, mlt);
  find_max_wave(rScore, row * (rtLen + 1), row, lane, mrq, mrt);
  if (lane != 0) return;
  int32_t* tb = a.tb + (size_t)j * 4 * 1002;                              // [0,1002) left path / q scores, [1002,2004) right, [2004,4008) blocks scratch
  if (!(mlq < span - mrq)) {                                             // :362-384
    int32_t* lqS = tb; int32_t* lqI = tb + 501; int32_t* rqS = tb + 1002; int32_t* rqI = tb + 1503;
    store_q(lScore, lPath, mlq, mlt, row, lqS, lqI);
    store_q(rScore, rPath, mrq, mrt, row, rqS, rqI);
    int maxScore = 0, maxL = 0, maxR = 0;
    for (int i = 0; i < span; i++)
      if (lqS[i] + rqS[span - i - 1] > maxScore) { maxScore = lqS[i] + rqS[span - i - 1]; maxL = i; maxR = span - i - 1; }
    mlq = maxL; mlt = lqI[maxL] / row - 1; mrq = maxR; mrt = rqI[maxR] / row - 1;
  }
  const int lqs = nL ? L[0] : 0, lqe = nL ? L[3 * (nL - 1)] + L[3 * (nL - 1) + 2] : 0, lts = nL ? L[1] : 0, lte = nL ? L[3 * (nL - 1) + 1] + L[3 * (nL - 1) + 2] : 0;
  const int rqs = nR ? R[0] : 0, rqe = nR ? R[3 * (nR - 1)] + R[3 * (nR - 1) + 2] : 0, rts = nR ? R[1] : 0, rte = nR ? R[3 * (nR - 1) + 1] + R[3 * (nR - 1) + 2] : 0;
  int32_t* blk = tb + 2004;
  {
    int n = trace_back(lPath, mlq, mlt, row, tb, 1002);
    if (lPrefix) for (int x = 0, y = n - 1; x < y; x++, y--) { const int v = tb[x]; tb[x] = tb[y]; tb[y] = v; }
    const int nb = path_to_blocks(tb, n, blk, lPrefix ? lqs - mlq - 1 : lqe, lPrefix ? lts - mlt - 1 : lte);
    a.nLOut[j] = glue(L, nL, blk, nb, lPrefix, lOut);
  }
  {
    int n = trace_back(rPath, mrq, mrt, row, tb + 1002, 1002);
    if (rPrefix) for (int x = 0, y = n - 1; x < y; x++, y--) { const int v = tb[1002 + x]; tb[1002 + x] = tb[1002 + y]; tb[1002 + y] = v; }
    const int nb = path_to_blocks(tb + 1002, n, blk, rPrefix ? rqs - mrq - 1 : rqe, rPrefix ? rts - mrt - 1 : rte);
    a.nROut[j] = glue(R, nR, blk, nb, rPrefix, rOut);
  }
  a.status[j] |= 0x10000u;                                               // refined
}

__global__ void rb_caps(int n, const uint64_t* lOff, const uint64_t* rOff, uint32_t* lCap, uint32_t* rCap) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  lCap[j] = (uint32_t)(lOff[j + 1] - lOff[j]) + 502; rCap[j] = (uint32_t)(rOff[j + 1] - rOff[j]) + 502;
}

inline size_t sz(size_t n, size_t e) { return (n * e + 255) / 256 * 256; }

}  // namespace

extern "C" int lra_refine_breakpoint_batch(lra_ctx* ctx, int n, const int32_t* d_read_len, const char* d_seq, const char* d_genome, const int32_t* d_l_blocks,
                                           const uint64_t* d_l_off, const int32_t* d_l_strand, const uint64_t* d_l_read_off, const uint64_t* d_l_chrom_off,
                                           const int32_t* d_l_chrom_len, const int32_t* d_r_blocks, const uint64_t* d_r_off, const int32_t* d_r_strand,
                                           const uint64_t* d_r_read_off, const uint64_t* d_r_chrom_off, const int32_t* d_r_chrom_len,
                                           lra_breakpoint_result* out) {
  if (!ctx || !out || n < 0) return LRA_ERR_INVALID;
  memset(out, 0, sizeof *out);
  out->n_junctions = (uint64_t)n;
  if (n == 0) return LRA_OK;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t n1 = (size_t)n + 2;
  auto take = [](char*& p, size_t c, size_t e) { char* r = p; p += sz(c, e); return r; };
  char* w = (char*)lra_ensure(ctx, 18, sz(n1, 4) * 7 + sz(2 * n1, 4) + sz(n1, 1) * 2 + sz(n1, 8) * 6 + sz(2 * n1 + 2, 8) + sz(n1 * 4 * 1002, 4) + 4096);
  if (!w) return LRA_ERR_NOMEM;
  RbArgs a;
  memset(&a, 0, sizeof a);
  a.n = n; a.read_len = d_read_len; a.seq = d_seq; a.genome = d_genome; a.lBlocks = d_l_blocks; a.lOff = d_l_off; a.lStrand = d_l_strand; a.lReadOff = d_l_read_off;
  a.lChromOff = d_l_chrom_off; a.lChromLen = d_l_chrom_len; a.rBlocks = d_r_blocks; a.rOff = d_r_off; a.rStrand = d_r_strand; a.rReadOff = d_r_read_off;
  a.rChromOff = d_r_chrom_off; a.rChromLen = d_r_chrom_len;
  a.span = (int32_t*)take(w, n1, 4); a.ltLen = (int32_t*)take(w, n1, 4); a.rtLen = (int32_t*)take(w, n1, 4);
  uint32_t* lCap = (uint32_t*)take(w, n1, 4); uint32_t* rCap = (uint32_t*)take(w, n1, 4); int32_t* nLOut = (int32_t*)take(w, n1, 4); int32_t* nROut = (int32_t*)take(w, n1, 4);
  a.cells = (uint32_t*)take(w, 2 * n1, 4); a.lPrefix = (uint8_t*)take(w, n1, 1); a.rPrefix = (uint8_t*)take(w, n1, 1);
  a.lqAt = (uint64_t*)take(w, n1, 8); a.ltAt = (uint64_t*)take(w, n1, 8); a.rqAt = (uint64_t*)take(w, n1, 8); a.rtAt = (uint64_t*)take(w, n1, 8);
  uint64_t* lOutOff = (uint64_t*)take(w, n1, 8); uint64_t* rOutOff = (uint64_t*)take(w, n1, 8); uint64_t* cellOff = (uint64_t*)take(w, 2 * n1 + 2, 8);
  a.tb = (int32_t*)take(w, n1 * 4 * 1002, 4);
  uint32_t* status = (uint32_t*)lra_ensure(ctx, 19, sz(n1, 4) + 256);
  if (!status) return LRA_ERR_NOMEM;
  a.status = status; a.nLOut = nLOut; a.nROut = nROut;
  const unsigned g = (unsigned)((n + 255) / 256);
  lra_time_begin(ctx, "refine_breakpoint");
  hipLaunchKernelGGL(rb_plan, dim3(g), dim3(256), 0, st, a);
  hipLaunchKernelGGL(rb_caps, dim3(g), dim3(256), 0, st, n, d_l_off, d_r_off, lCap, rCap);
  lra_time_end(ctx);
  { int rc = lra_exclusive_scan<uint32_t>(ctx, 2 * n, a.cells, cellOff); if (rc) return rc; }
  { int rc = lra_exclusive_scan<uint32_t>(ctx, n, lCap, lOutOff); if (rc) return rc; }
  { int rc = lra_exclusive_scan<uint32_t>(ctx, n, rCap, rOutOff); if (rc) return rc; }
  uint64_t tot[3];
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&tot[0], cellOff + 2 * n, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&tot[1], lOutOff + n, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&tot[2], rOutOff + n, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  char* m = (char*)lra_ensure(ctx, 20, sz(tot[0] + 1, 4) + sz(tot[0] + 1, 1) + 4096);
  if (!m) return LRA_ERR_NOMEM;
  a.score = (int32_t*)take(m, tot[0] + 1, 4); a.path = (uint8_t*)take(m, tot[0] + 1, 1); a.cellOff = cellOff;
  char* r = (char*)lra_ensure(ctx, 21, sz(3 * tot[1] + 3, 4) + sz(3 * tot[2] + 3, 4) + sz(n1, 8) * 2 + sz(n1, 4) * 2 + 4096);
  if (!r) return LRA_ERR_NOMEM;
  a.lOut = (int32_t*)take(r, 3 * tot[1] + 3, 4); a.rOut = (int32_t*)take(r, 3 * tot[2] + 3, 4);
  uint64_t* lOffOut = (uint64_t*)take(r, n1, 8); uint64_t* rOffOut = (uint64_t*)take(r, n1, 8); int32_t* nLO = (int32_t*)take(r, n1, 4); int32_t* nRO = (int32_t*)take(r, n1, 4);
  a.lOutOff = lOutOff; a.rOutOff = rOutOff; a.nLOut = nLO; a.nROut = nRO;
  lra_time_begin(ctx, "refine_breakpoint");
  hipLaunchKernelGGL(rb_kernel, dim3(n), dim3(64), 0, st, a);
  lra_time_end(ctx);
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(lOffOut, lOutOff, ((size_t)n + 1) * 8, hipMemcpyDeviceToDevice, st));
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(rOffOut, rOutOff, ((size_t)n + 1) * 8, hipMemcpyDeviceToDevice, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  LRA_HIP_CHECK(ctx, hipGetLastError());
  out->d_l_blocks = a.lOut; out->d_l_off = lOffOut; out->d_l_n = nLO; out->d_r_blocks = a.rOut; out->d_r_off = rOffOut; out->d_r_n = nRO; out->d_status = status;
  return LRA_OK;
}
