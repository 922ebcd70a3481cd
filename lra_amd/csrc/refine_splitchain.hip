// lra_amd/csrc/refine_splitchain.hip -- SURVEY §8a row a10 (low-accuracy path): Refine_splitchain (ChainRefine.h:384-576, called at
// Map_lowacc.h:294) for every split chain of a batch.  gfx950 only.
//   SwapStrand(read, opts, cluster, K)          ClusterRefine.h:24-31      LocalIndex::LookupIndex                 MMIndex.h:175-190
//   GenomeHeader::GetNextOffset / Find          Genome.h:43-47, :20-32     AppendValues<LocalPairs>                TupleOps.h:159-195
//   CompareLists<LocalTuple,SmallTuple>         CompareLists.h:9 (lra_local_compare_batch)
//   Cluster::SetClusterBoundariesFromMatches    Clustering.h:308-322
// Mapping.  (1) rsc_tasks, one lane per chain slot: walks the genome local-index windows under each of its split chains exactly as the
// reference does (the anchor cursor matchStart runs on across windows, so this is a serial scan per split chain), and turns every
// (genome window, read window) the reference intersects into a task: two tuple ranges + the AppendValues box.  Run twice: count, then
// emit at the scanned offsets.  (2) lra_local_compare_batch intersects all tasks.  (3) rsc_filter, one lane per task, applies
// AppendValues' diagonal / box test, again count then emit, so the matches of a split chain come out contiguous and in the reference's
// order.  (4) rsc_finish, one wave per chain slot: SwapStrand on reverse split chains, box and refineEffiency by wave reduction.
// UNDEFINED BEHAVIOUR IN THE REFERENCE: with opts.limitrefine (the default) the per-window upper diagonal bound starts from an
// uninitialised variable (ChainRefine.h:468, `miniMaxDiag = miniMaxDiag;`).  What the reference binary does with it was measured
// (SURVEY.md H2): the stale stack slot holds a pointer-sized value that only grows, i.e. there is NO upper diagonal bound; only
// miniMinDiag - 100 filters (AppendValues, TupleOps.h:168).  That is what is implemented: the upper bound is 2^60.
// Algorithmic bytes: 21 B per chain anchor, 36 B per task, 8 B per candidate pair in, 8 B per kept match out.
#include "common.h"
#include "append_values.h"
#include "scan.h"
#include <vector>

namespace {

struct RscArgs {
  uint64_t n_slots; int numAln;
  const uint32_t* nChains; const uint64_t* chainStart;
  const uint32_t* cq; const uint32_t* ct; const int32_t* clen; const uint32_t* ccl; const uint8_t* cstrand;
  const uint32_t* nSplit; const uint32_t* spBeg; const uint32_t* spLen; const uint32_t* spIdx; const uint8_t* spStrand; const int32_t* spChrom;
  const uint32_t* spBox; const uint32_t* ciBeg; const uint32_t* ciLen; const uint32_t* ciIdx; const uint32_t* fidx; const uint32_t* spStatus;
  const uint64_t* read_off;
  const uint64_t* pos; int npos;
  const uint64_t* qWinOff; const uint64_t* qBnd; uint64_t n_reads;   // read index: sequences [0, n_reads) forward, [n_reads, 2 n_reads) reverse
  const uint64_t* gSeqOff; uint64_t nWg; const uint64_t* gBnd;
  int window, smallK, K, limitrefine, lwin;
  // per split (indexed like the split arrays)
  uint32_t* taskCnt; const uint64_t* taskOff; uint32_t* status;
  // per task
  uint64_t* qLo; uint64_t* qHi; uint64_t* tLo; uint64_t* tHi; uint32_t* qAdd; uint32_t* tAdd; int64_t* mx; int64_t* mn;
};

__device__ int header_find(const uint64_t* pos, int npos, uint64_t query, bool& ub) {   // Genome.h:20-32
  if (npos > 0 && query == pos[0]) return 0;
  int lo = 0, cnt = npos;
  while (cnt > 0) { const int step = cnt >> 1; if (pos[lo + step] < query) { lo += step + 1; cnt -= step + 1; } else cnt = step; }
  if (lo == npos) { ub = true; return lo - 1; }
  if (query == pos[lo]) return lo;
  return lo - 1;
}

__device__ long lookup_index(const uint64_t* so, long n, uint64_t pos, bool& ub) {      // MMIndex.h:175-190, n = seqOffsets.size()
  if (n == 0) return 0;
  long lo = 0, cnt = n;
  while (cnt > 0) { const long step = cnt >> 1; if (so[lo + step] < pos) { lo += step + 1; cnt -= step + 1; } else cnt = step; }
  if (lo == n) { ub = true; return lo - 1; }
  if (so[lo] != pos) return lo - 1;
  return lo;
}

template <bool EMIT>
__global__ void __launch_bounds__(64) rsc_tasks(RscArgs a) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= a.n_slots) return;
  const uint64_t r = s / a.numAln;
  if ((uint32_t)(s % a.numAln) >= a.nChains[r]) return;
  const uint64_t cs = a.chainStart[s];
  const uint32_t readLen = (uint32_t)(a.read_off[r + 1] - a.read_off[r]);
  const long nWq = ((long)readLen + a.lwin - 1) / a.lwin;
  const uint32_t nsp = a.spStatus[s] ? 0 : a.nSplit[s];
  for (uint32_t k = 0; k < nsp; k++) {
    const uint64_t x = cs + k;
    const int m = (int)a.spLen[x];
    uint32_t ntask = 0;
    uint64_t to = EMIT ? a.taskOff[x] : 0;
    if (!EMIT) { a.taskCnt[x] = 0; a.status[x] = 0; }
    else if (a.status[x]) continue;
    if (m == 0) continue;
    const uint32_t* idx = a.spIdx + cs + a.spBeg[x];
    const uint32_t* ci = a.ciIdx + cs + a.ciBeg[x];
    const int nci = (int)a.ciLen[x];
    const int Strand = a.spStrand[x];
    const uint32_t chromOffset = (uint32_t)a.pos[a.spChrom[x]];
    auto flipped = [&](uint32_t c) { for (int j = 0; j < nci; j++) if (ci[j] == c) return true; return false; };
    auto AN = [&](int i) { return cs + a.fidx[cs + idx[i]]; };
    auto tS = [&](int i) { const uint64_t p = AN(i); return flipped(a.ccl[p]) ? a.ct[p] - chromOffset : a.ct[p]; };
    auto qS = [&](int i) { const uint64_t p = AN(i); return (a.cstrand[p] == 1 && flipped(a.ccl[p])) ? readLen - (a.cq[p] + (uint32_t)a.K) : a.cq[p]; };
    auto qE = [&](int i) { return qS(i) + (uint32_t)a.clen[AN(i)]; };
    const uint32_t QStart = a.spBox[4 * x], QEnd = a.spBox[4 * x + 1], TStart = a.spBox[4 * x + 2], TEnd = a.spBox[4 * x + 3];
    bool ub = false;
    const int fi = header_find(a.pos, a.npos, TEnd, ub);
    if (ub || fi + 1 >= a.npos) { if (!EMIT) a.status[x] = LRA_ST_OOB_SLOT; continue; }
    const uint32_t chromEndOffset = (uint32_t)a.pos[fi + 1];
    int64_t maxDN = (int64_t)tS(0) - (int64_t)qS(0), minDN = maxDN;
    if (!a.limitrefine)
      for (int db = 0; db < m; db++) { const int64_t d = (int64_t)tS(db) - (int64_t)qS(db); maxDN = max(maxDN, d); minDN = min(minDN, d); }
    const int64_t maxDiagNum = maxDN + 50, minDiagNum = minDN - 50;
    const uint32_t wts = (TStart >= chromOffset + (uint32_t)a.window) ? TStart - a.window : chromOffset;
    const uint32_t wte = (TEnd + (uint32_t)a.window < chromEndOffset) ? TEnd + a.window : chromEndOffset;
    const long ls = lookup_index(a.gSeqOff, (long)a.nWg + 1, wts, ub), le = lookup_index(a.gSeqOff, (long)a.nWg + 1, wte, ub);
    if (ub) { if (!EMIT) a.status[x] = LRA_ST_OOB_SLOT; continue; }
    const uint32_t qStart = Strand == 0 ? QStart : readLen - QEnd, qEnd = Strand == 0 ? QEnd : readLen - QStart;
    const uint64_t w0 = a.qWinOff[(Strand ? a.n_reads : 0) + r];
    int matchStart = 0, matchEnd = 0;
    bool bad = false;
    for (long lsi = ls; lsi <= le && !bad; lsi++) {
      // lsi == nWg happens next to the end of the genome; the reference then reads seqOffsets one past its end, and whatever it finds
      // the window holds no anchor (every tStart is <= the chromosome length), so it adds nothing
      if (lsi + 1 > (long)a.nWg) continue;
      if (a.gSeqOff[lsi] < chromOffset || a.gSeqOff[lsi + 1] < chromOffset) continue;
      const uint32_t gStart = (uint32_t)(a.gSeqOff[lsi] - chromOffset), gEnd = (uint32_t)(a.gSeqOff[lsi + 1] - 1 - chromOffset);
      if (gStart >= gEnd) continue;
      while (matchStart < m && tS(matchStart) <= gStart) matchStart++;
      matchEnd = matchStart;
      while (matchEnd < m && tS(matchEnd) < gEnd) matchEnd++;
      if (matchStart >= m) continue;
      if (matchEnd == matchStart) continue;
      uint32_t readStart = qS(matchStart), readEnd = qS(matchEnd - 1);
      int64_t miniMin = (int64_t)tS(matchStart) - (int64_t)qS(matchStart);
      const int64_t miniMax = (int64_t)1 << 60;                            // no upper bound: see the header
      for (int mi = matchStart; mi < matchEnd; mi++) {
        const uint32_t q0 = qS(mi), q1 = q0 + (uint32_t)a.clen[AN(mi)];
        if (q0 < readStart) readStart = q0;
        if (q1 > readEnd) readEnd = q1;
        const int64_t d = (int64_t)tS(mi) - (int64_t)q0;
        miniMin = min(miniMin, d);
      }
      if (readStart == readEnd) { if (lsi > ls && readStart > 0) readStart = 0; }   // prev_readEnd is 0 here (:452, :461)
      miniMin -= 100;
      const uint32_t sow = 500;
      if (lsi == ls) readStart = (readStart < sow) ? 0 : readStart - sow;
      if (lsi == le) readEnd = (readEnd + sow > readLen) ? readLen : readEnd + sow;
      if (readStart > readEnd) continue;
      // LookupIndex on the read's own index: windows of lwin bases, seqOffsets = 0, lwin, .., readLen
      const uint32_t e2 = min(readEnd, readLen - 1);
      const long qi0 = readStart == readLen ? nWq : (long)(readStart / (uint32_t)a.lwin);
      const long qi1 = (long)(e2 / (uint32_t)a.lwin);
      if (readStart > readLen) { bad = true; break; }
      for (long qi = qi0; qi <= qi1; ++qi) {
        if (EMIT) {
          a.qLo[to] = a.qBnd[w0 + qi]; a.qHi[to] = a.qBnd[w0 + qi + 1];
          a.tLo[to] = a.gBnd[lsi]; a.tHi[to] = a.gBnd[lsi + 1];
          a.qAdd[to] = (uint32_t)min((long)readLen, qi * (long)a.lwin); a.tAdd[to] = gStart;
          a.mx[to] = a.limitrefine ? miniMax : maxDiagNum; a.mn[to] = a.limitrefine ? miniMin : minDiagNum;
          to++;
        }
        ntask++;
      }
    }
    (void)qE; (void)qStart; (void)qEnd;
    if (bad) { if (!EMIT) { a.status[x] = LRA_ST_OOB_SLOT; a.taskCnt[x] = 0; } continue; }
    if (!EMIT) a.taskCnt[x] = ntask;
  }
}

typedef AvArgs FilterArgs;                                              // (append_values.h: the filter both refinement drivers share)

// the AppendValues box of each task, from its split chain (ChainRefine.h:533-534, :537-539)
__global__ void rsc_task_box(uint64_t n_slots, int numAln, const uint32_t* nChains, const uint64_t* chainStart, const uint32_t* nSplit,
                             const uint32_t* spStatus, const uint64_t* taskOff, const uint32_t* spBox, const uint8_t* spStrand, const int32_t* spChrom,
                             const uint64_t* pos, const uint64_t* read_off, uint32_t* tbox) {
  const uint64_t s = (uint64_t)blockIdx.x;
  const uint64_t r = s / numAln;
  if ((uint32_t)(s % numAln) >= nChains[r] || spStatus[s]) return;
  const uint64_t cs = chainStart[s];
  const uint32_t readLen = (uint32_t)(read_off[r + 1] - read_off[r]);
  for (uint32_t k = 0; k < nSplit[s]; k++) {
    const uint64_t x = cs + k;
    const uint32_t chromOffset = (uint32_t)pos[spChrom[x]];
    const uint32_t QStart = spBox[4 * x], QEnd = spBox[4 * x + 1], TStart = spBox[4 * x + 2], TEnd = spBox[4 * x + 3];
    const int st = spStrand[x];
    const uint32_t qs = st == 0 ? QStart : readLen - QEnd, qe = st == 0 ? QEnd : readLen - QStart;
    for (uint64_t t = taskOff[x] + threadIdx.x; t < taskOff[x + 1]; t += blockDim.x) {
      tbox[4 * t] = qs; tbox[4 * t + 1] = qe; tbox[4 * t + 2] = TStart - chromOffset; tbox[4 * t + 3] = TEnd - chromOffset;
    }
  }
}

struct FinishArgs {
  uint64_t n_slots; int numAln; int smallK;
  const uint32_t* nChains; const uint64_t* chainStart; const uint32_t* nSplit; const uint32_t* spStatus; const uint8_t* spStrand;
  const uint64_t* taskOff; const uint64_t* outOff; const uint64_t* read_off;
  uint64_t* matchOff; uint32_t* oq; const uint32_t* ot; uint32_t* box; float* eff; uint64_t nf;
};

// matchOff over the whole split index space (an empty range everywhere but at real split chains)
__global__ void rsc_match_off(FinishArgs a) {
  const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (x > a.nf) return;
  a.matchOff[x] = a.outOff[a.taskOff[x]];
}

// ChainRefine.h:561-574: SwapStrand(read, smallOpts, refined, smallK) for reverse split chains, SetClusterBoundariesFromMatches,
// refineEffiency.  One wave per chain slot.
__global__ void __launch_bounds__(64) rsc_finish(FinishArgs a) {
  const uint64_t s = blockIdx.x;
  const uint64_t r = s / a.numAln;
  if ((uint32_t)(s % a.numAln) >= a.nChains[r] || a.spStatus[s]) return;
  const uint64_t cs = a.chainStart[s];
  const uint32_t readLen = (uint32_t)(a.read_off[r + 1] - a.read_off[r]);
  const int lane = threadIdx.x;
  for (uint32_t k = 0; k < a.nSplit[s]; k++) {
    const uint64_t x = cs + k;
    const uint64_t m0 = a.matchOff[x], m1 = a.matchOff[x + 1];
    if (m1 == m0) { if (lane == 0) { a.box[4 * x] = 0; a.box[4 * x + 1] = 0; a.box[4 * x + 2] = 0; a.box[4 * x + 3] = 0; a.eff[x] = 0; } continue; }
    const bool rev = a.spStrand[x] == 1;
    uint32_t qmin = 0xFFFFFFFFu, qmax = 0, tmin = 0xFFFFFFFFu, tmax = 0;
    for (uint64_t i = m0 + lane; i < m1; i += 64) {
      uint32_t q = a.oq[i];
      if (rev) { q = readLen - (q + (uint32_t)a.smallK); a.oq[i] = q; }
      const uint32_t t = a.ot[i];
      qmin = min(qmin, q); qmax = max(qmax, q + (uint32_t)a.smallK); tmin = min(tmin, t); tmax = max(tmax, t + (uint32_t)a.smallK);
    }
    for (int o = 32; o > 0; o >>= 1) {
      qmin = min(qmin, __shfl_xor(qmin, o)); qmax = max(qmax, __shfl_xor(qmax, o));
      tmin = min(tmin, __shfl_xor(tmin, o)); tmax = max(tmax, __shfl_xor(tmax, o));
    }
    if (lane == 0) {
      a.box[4 * x] = qmin; a.box[4 * x + 1] = qmax; a.box[4 * x + 2] = tmin; a.box[4 * x + 3] = tmax;
      a.eff[x] = ((float)(m1 - m0)) / (float)min(qmax - qmin, tmax - tmin);
    }
  }
}

inline size_t sz(size_t n, size_t e) { return (n * e + 255) / 256 * 256; }

}  // namespace

extern "C" int lra_refine_splitchain_batch(lra_ctx* ctx, const lra_chain_result* ch, const lra_split_result* sp, const uint64_t* d_read_off,
                                           const uint64_t* h_chrom_pos, int n_chrom, const lra_local_index_result* read_index,
                                           uint64_t n_g_windows, const uint64_t* d_g_seq_off, const uint64_t* d_g_tuple_bnd,
                                           const uint32_t* d_g_tuples, const lra_rsc_opts* opts, lra_refined_result* out) {
  if (!ctx || !ch || !sp || !out || !opts || !h_chrom_pos || n_chrom < 1 || !read_index) return LRA_ERR_INVALID;
  if (opts->local_window <= 0) return lra_set_err(ctx, LRA_ERR_INVALID, "local_window must be positive");
  if (read_index->n_seqs != 2 * ch->n_reads) return lra_set_err(ctx, LRA_ERR_INVALID, "read_index must hold the reads forward, then reverse-complemented");
  memset(out, 0, sizeof *out);
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const uint64_t slots = sp->n_slots, NF = sp->n_frags;
  out->n_frags = NF;
  if (slots == 0) return LRA_OK;
  const size_t npos = (size_t)n_chrom + 1;
  auto take = [](char*& p, size_t n, size_t e) { char* r = p; p += sz(n, e); return r; };
  char* w = (char*)lra_ensure(ctx, 27, sz(NF + 2, 4) * 2 + sz(NF + 2, 8) * 2 + sz(4 * NF + 4, 4) + sz(NF + 1, 4) + sz(npos, 8) + 4096);
  if (!w) return LRA_ERR_NOMEM;
  RscArgs a;
  memset(&a, 0, sizeof a);
  a.n_slots = slots; a.numAln = ch->num_aln; a.nChains = ch->d_n_chains; a.chainStart = ch->d_chain_start;
  a.cq = ch->d_chain_q; a.ct = ch->d_chain_t; a.clen = ch->d_chain_alen; a.ccl = ch->d_chain_cluster; a.cstrand = ch->d_chain_strand;
  a.nSplit = sp->d_n_split; a.spBeg = sp->d_sp_beg; a.spLen = sp->d_sp_len; a.spIdx = sp->d_sp_idx; a.spStrand = sp->d_sp_strand; a.spChrom = sp->d_sp_chrom;
  a.spBox = sp->d_sp_box; a.ciBeg = sp->d_ci_beg; a.ciLen = sp->d_ci_len; a.ciIdx = sp->d_ci_idx; a.fidx = sp->d_fidx; a.spStatus = sp->d_status;
  a.read_off = d_read_off;
  a.qWinOff = read_index->d_win_off; a.qBnd = read_index->d_tuple_bnd; a.n_reads = (uint64_t)ch->n_reads;
  a.gSeqOff = d_g_seq_off; a.nWg = n_g_windows; a.gBnd = d_g_tuple_bnd;
  a.window = opts->window; a.smallK = opts->smallK; a.K = opts->K; a.limitrefine = opts->limitrefine; a.lwin = opts->local_window;
  a.taskCnt = (uint32_t*)take(w, NF + 2, 4); a.status = (uint32_t*)take(w, NF + 2, 4);
  uint64_t* taskOff = (uint64_t*)take(w, NF + 2, 8); uint64_t* matchOff = (uint64_t*)take(w, NF + 2, 8);
  uint32_t* box = (uint32_t*)take(w, 4 * NF + 4, 4); float* eff = (float*)take(w, NF + 1, 4);
  uint64_t* dpos = (uint64_t*)take(w, npos, 8);
  a.pos = dpos; a.npos = (int)npos; a.taskOff = taskOff;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(dpos, h_chrom_pos, npos * 8, hipMemcpyHostToDevice, st));
  LRA_HIP_CHECK(ctx, hipMemsetAsync(a.taskCnt, 0, (NF + 2) * 4, st));
  LRA_HIP_CHECK(ctx, hipMemsetAsync(a.status, 0, (NF + 2) * 4, st));
  const unsigned gs = (unsigned)((slots + 63) / 64);
  lra_time_begin(ctx, "rsc_tasks");
  hipLaunchKernelGGL(rsc_tasks<false>, dim3(gs), dim3(64), 0, st, a);
  lra_time_end(ctx);
  { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)NF + 1, a.taskCnt, taskOff); if (rc) return rc; }
  uint64_t NT = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(&NT, taskOff + NF + 1, 8, hipMemcpyDeviceToHost, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  out->n_tasks = NT;
  out->d_match_off = matchOff; out->d_box = box; out->d_eff = eff; out->d_status = a.status;
  char* wt = (char*)lra_ensure(ctx, 28, sz(NT + 2, 8) * 7 + sz(NT + 1, 4) * 3 + sz(4 * NT + 4, 4) + 4096);
  if (!wt) return LRA_ERR_NOMEM;
  a.qLo = (uint64_t*)take(wt, NT + 2, 8); a.qHi = (uint64_t*)take(wt, NT + 2, 8); a.tLo = (uint64_t*)take(wt, NT + 2, 8); a.tHi = (uint64_t*)take(wt, NT + 2, 8);
  a.mx = (int64_t*)take(wt, NT + 2, 8); a.mn = (int64_t*)take(wt, NT + 2, 8); uint64_t* outOff = (uint64_t*)take(wt, NT + 2, 8);
  a.qAdd = (uint32_t*)take(wt, NT + 1, 4); a.tAdd = (uint32_t*)take(wt, NT + 1, 4);
  uint32_t* passCnt = (uint32_t*)take(wt, NT + 1, 4); uint32_t* tbox = (uint32_t*)take(wt, 4 * NT + 4, 4);
  FinishArgs f;
  memset(&f, 0, sizeof f);
  f.n_slots = slots; f.numAln = ch->num_aln; f.smallK = opts->smallK; f.nChains = ch->d_n_chains; f.chainStart = ch->d_chain_start; f.nSplit = sp->d_n_split;
  f.spStatus = sp->d_status; f.spStrand = sp->d_sp_strand; f.taskOff = taskOff; f.outOff = outOff; f.read_off = d_read_off; f.matchOff = matchOff;
  f.box = box; f.eff = eff; f.nf = NF;
  uint64_t NM = 0;
  uint32_t* oq = nullptr; uint32_t* ot = nullptr;
  if (NT > 0) {
    lra_time_begin(ctx, "rsc_tasks");
    hipLaunchKernelGGL(rsc_tasks<true>, dim3(gs), dim3(64), 0, st, a);
    hipLaunchKernelGGL(rsc_task_box, dim3((unsigned)slots), dim3(64), 0, st, slots, ch->num_aln, ch->d_n_chains, ch->d_chain_start, sp->d_n_split, sp->d_status,
                       (const uint64_t*)taskOff, sp->d_sp_box, sp->d_sp_strand, sp->d_sp_chrom, (const uint64_t*)dpos, d_read_off, tbox);
    lra_time_end(ctx);
    lra_local_pairs_result pr;                                           // CompareLists (maxDiagNum = minDiagNum = 0: no band inside, :531)
    { int rc = lra_local_compare_batch(ctx, NT, read_index->d_tuples, a.qLo, a.qHi, d_g_tuples, a.tLo, a.tHi, opts->max_freq, nullptr, nullptr, &pr); if (rc) return rc; }
    out->n_pairs = pr.n_pairs;
    FilterArgs fa;
    memset(&fa, 0, sizeof fa);
    fa.n_tasks = NT; fa.pairOff = pr.d_pair_off; fa.pqi = pr.d_pair_qi; fa.pti = pr.d_pair_ti; fa.qTup = read_index->d_tuples; fa.gTup = d_g_tuples;
    fa.qAdd = a.qAdd; fa.tAdd = a.tAdd; fa.mx = a.mx; fa.mn = a.mn; fa.tbox = tbox; fa.cnt = passCnt; fa.outOff = outOff;
    const unsigned gt = (unsigned)((NT + 63) / 64);
    lra_time_begin(ctx, "rsc_filter");
    hipLaunchKernelGGL(av_filter<false>, dim3(gt), dim3(64), 0, st, fa);
    lra_time_end(ctx);
    { int rc = lra_exclusive_scan<uint32_t>(ctx, (long)NT, passCnt, outOff); if (rc) return rc; }
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(&NM, outOff + NT, 8, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
    char* wm = (char*)lra_ensure(ctx, 29, sz(NM + 1, 4) * 2 + 1024);
    if (!wm) return LRA_ERR_NOMEM;
    oq = (uint32_t*)take(wm, NM + 1, 4); ot = (uint32_t*)take(wm, NM + 1, 4);
    fa.oq = oq; fa.ot = ot;
    lra_time_begin(ctx, "rsc_filter");
    hipLaunchKernelGGL(av_filter<true>, dim3(gt), dim3(64), 0, st, fa);
    lra_time_end(ctx);
  } else {
    LRA_HIP_CHECK(ctx, hipMemsetAsync(outOff, 0, 16, st));
  }
  out->n_matches = NM; out->d_match_q = oq; out->d_match_t = ot;
  out->d_task_q_lo = a.qLo; out->d_task_q_hi = a.qHi; out->d_task_t_lo = a.tLo; out->d_task_t_hi = a.tHi;
  f.oq = oq; f.ot = ot;
  lra_time_begin(ctx, "rsc_filter");
  hipLaunchKernelGGL(rsc_match_off, dim3((unsigned)((NF + 2 + 255) / 256)), dim3(256), 0, st, f);
  hipLaunchKernelGGL(rsc_finish, dim3((unsigned)slots), dim3(64), 0, st, f);
  lra_time_end(ctx);
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  LRA_HIP_CHECK(ctx, hipGetLastError());
  return LRA_OK;
}
