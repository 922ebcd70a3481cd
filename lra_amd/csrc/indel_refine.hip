// lra_amd/csrc/indel_refine.hip -- batched banded 3-state affine indel refinement (gfx950).
//
// Replaces IndelRefineAlignment (reference: IndelRefine.h:53-784) for a batch of
// alignments.  The reference walks an alignment's gapless blocks, groups runs of blocks
// separated by small gaps into SEGMENTS (:132-211), builds for each segment a per-target-row
// query window [qS,qE] (:220-333), fills a match/deletion/insertion affine DP inside the
// windows (:383-622), traces it back (:626-674) and turns the path into blocks (:713-745);
// very short segments go to AffineOneGapAlign instead (:344-357).
//
// Pipeline on the GPU (all device-resident, CSR arrays; sizes come back to the host between
// phases only to size the next buffers):
//   ir_segment  one lane per alignment : the block-grouping state machine -> segment
//               descriptors + an ordered item list (pass-through block | segment)
//   ir_band     one wave per segment   : the row-window construction, event by event (the
//               k look-ahead / look-back updates of one event are done by k lanes; a
//               128-row ring in LDS holds the rows still being modified), then suffix-min /
//               prefix-max / prefix-sum scans over the rows (:318-328)
//   ir_fill     one wave per segment   : rows are swept top to bottom, ONE LANE PER CELL OF
//               THE ROW; the previous row stays in registers (cross-lane permutes line the
//               two windows up); the in-row insertion recurrence is solved in closed form with
//               one prefix-max scan (see "row recurrence" below), so a row costs O(log width)
//               dependent steps; one byte of trace-back state per cell goes to HBM
//               (coalesced, the only HBM traffic that scales with cells)
//   ir_trace    one wave per segment   : serial walk by lane 0 over 64-row chunks of arrows
//               staged in LDS; run twice (count blocks, then emit them back to front)
//   ir_gather   one wave per item      : assemble the refined block list of every alignment
//
// Row recurrence.  With g = indel (< 0), go = 2g+1, extension 0 (IndelRefine.h:338-340), cell
// q of a row takes  M = max(V, M[q-1]+g, I),  I = max(M[q-1]+go, I[q-1]),  where V collects the
// candidates that only depend on the previous row (match, single-base deletion, deletion
// close).  Because go+go <= go, g+g <= go and g+go <= go, chains of horizontal moves never beat
// one direct move, and the row's left boundary injects BAD through the extension chain, so
//     M[q] = max(BAD, V[q], V[q-1]+g, go + max_{q'<q} V[q']),   I[q] = max(BAD, go + max_{q'<q} V[q'])
// exactly (integers, no rounding).  The arrows are then chosen by the reference's equality
// cascade (:583-616) from the true candidate values, so ties resolve identically.
#include "common.h"
#include <algorithm>

int lra_aog_launch_device(lra_ctx* ctx, int n, const char* d_qseq, const char* d_tseq, const uint64_t* d_q_off,
                          const int32_t* d_q_len, const uint64_t* d_t_off, const int32_t* d_t_len, const int32_t* d_k,
                          int m, int mm, int indel, int32_t* d_score, int32_t* d_nblocks, int32_t* d_blocks,
                          const uint64_t* d_block_off, int32_t* d_status);

namespace {

constexpr int BAD = -999999999;       // IndelRefine.h:368
constexpr int NEG = -2000000000;      // "no candidate" sentinel, below every reachable score
enum { C_DIAG = 0, C_LEFT = 1, C_DOWN = 2, C_BOUND = 3, C_DELCLOSE = 4, C_INSCLOSE = 5, C_DONE = 6 };
constexpr int RING = 128;

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

struct IRArgs {
  int n_aln;
  const int32_t* blocks_in; const uint64_t* block_off;
  const char* qseq; const uint64_t* q_off; const int32_t* q_len;
  const char* tseq; const uint64_t* t_off; const int64_t* t_len;
  int k, match, mismatch, indel, endAlign;
  // per alignment
  uint32_t* a_nseg; uint32_t* a_nitem; int32_t* a_status;
  const uint64_t* seg_off; const uint64_t* item_off;
  int32_t* ab;               // augmented block triples, alignment a at 3*(block_off[a] + 2a)
  // per segment
  int32_t* s_aln; int32_t* s_kind; int32_t* s_qStart; int32_t* s_tStart; int32_t* s_qEnd; int32_t* s_tEnd;
  int32_t* s_b0; int32_t* s_b1; int32_t* s_first; /* 3 per seg */ int32_t* s_lastLen; uint64_t* s_rows; /* tLen (0 for AOG) */
  uint32_t* s_isAog;
  // per item
  int32_t* i_kind; int32_t* i_data; /* 3 per item: pass block, or (seg,_,_) */
};

// ---------------------------------------------------------------------------------- segments
// The grouping loop of IndelRefine.h:79-211 / :761-767.  Only blocks[startBlock] and
// blocks[endBlock] are ever modified by the reference, and only the "alt" remainder survives
// an iteration, so one override record replaces the in-place edits.
template <bool EMIT>
__global__ void __launch_bounds__(64) ir_segment(IRArgs A) {
  const int a = blockIdx.x * 64 + threadIdx.x;
  if (a >= A.n_aln) return;
  const long nIn = (long)(A.block_off[a + 1] - A.block_off[a]);
  const int32_t* bin = A.blocks_in + 3 * A.block_off[a];
  int32_t* ab = A.ab + 3 * (A.block_off[a] + 2 * (uint64_t)a);
  const int k = A.k, maxGap = k - 1;
  uint32_t nseg = 0, nitem = 0;
  const uint64_t so = EMIT ? A.seg_off[a] : 0, io = EMIT ? A.item_off[a] : 0;
  auto pass = [&](long q, long t, long l) {
    if (EMIT) { A.i_kind[io + nitem] = 0; int32_t* d = A.i_data + 3 * (io + nitem); d[0] = (int)q; d[1] = (int)t; d[2] = (int)l; }
    nitem++;
  };
  if (nIn <= 1) {                                                       // :79
    for (long i = 0; i < nIn; i++) pass(bin[3 * i], bin[3 * i + 1], bin[3 * i + 2]);
    if (!EMIT) { A.a_nseg[a] = 0; A.a_nitem[a] = nitem; A.a_status[a] = 0; }
    return;
  }
  int addStart = 0, addEnd = 0;
  long sQ = 0, sT = 0, sL = 0, eQ = 0, eT = 0, eL = 0;
  if (A.endAlign) {                                                     // :89-130
    long qS0 = bin[0], tS0 = bin[1];
    int minStart = (int)min(qS0, tS0);
    if (minStart < 40) { sQ = qS0 - minStart; sT = tS0 - minStart; sL = minStart; addStart = 1; }
    long qAE = (long)bin[3 * (nIn - 1)] + bin[3 * (nIn - 1) + 2], tAE = (long)bin[3 * (nIn - 1) + 1] + bin[3 * (nIn - 1) + 2];
    int minEnd = (int)min((long)A.q_len[a] - qAE, (long)A.t_len[a] - tAE);
    if (minEnd < 40) { eQ = qAE; eT = tAE; eL = minEnd; addEnd = 1; }
  }
  const long nB = nIn + addStart + addEnd;
  auto getRaw = [&](long i, long& q, long& t, long& l) {
    if (addStart && i == 0) { q = sQ; t = sT; l = sL; }
    else if (addEnd && i == nB - 1) { q = eQ; t = eT; l = eL; }
    else { const int32_t* p = bin + 3 * (i - addStart); q = p[0]; t = p[1]; l = p[2]; }
  };
  if (EMIT)
    for (long i = 0; i < nB; i++) { long q, t, l; getRaw(i, q, t, l); ab[3 * i] = (int)q; ab[3 * i + 1] = (int)t; ab[3 * i + 2] = (int)l; }
  long ovIdx = -1, ovQ = 0, ovT = 0, ovL = 0;                           // surviving "alt" block
  auto get = [&](long i, long& q, long& t, long& l) {
    if (i == ovIdx) { q = ovQ; t = ovT; l = ovL; } else getRaw(i, q, t, l);
  };
  long startBlock = 0, endBlock = 0;
  while (endBlock < nB) {                                               // :132
    long q0, t0, l0;
    get(startBlock, q0, t0, l0);
    long qStart = q0, tStart = t0;
    long qPos = q0 + l0, tPos = t0 + l0;
    int tGap = 0, qGap = 0;
    long nq, nt, nl;
    if (endBlock < nB - 1) { get(endBlock + 1, nq, nt, nl); tGap = (int)(nt - tPos); qGap = (int)(nq - qPos); }
    long el = l0, eq = q0, et = t0;                                     // blocks[endBlock]
    while (endBlock < nB - 1 && qGap < maxGap && tGap < maxGap && (startBlock == endBlock || el < 100)) {   // :148-162
      endBlock++;
      get(endBlock, eq, et, el);
      qPos = eq + el; tPos = et + el;
      if (endBlock + 1 < nB - 1) { get(endBlock + 1, nq, nt, nl); tGap = (int)(nt - tPos); qGap = (int)(nq - qPos); }
    }
    bool usedAlt = false;
    long altQ = 0, altT = 0, altL = 0;
    if (endBlock == startBlock) {
      pass(q0, t0, l0);                                                 // :170-173
    } else {
      long fq = q0, ft = t0, fl = l0;                                   // first block as the segment sees it
      if (l0 > maxGap) {                                                // :178-196
        long advanced = l0 - maxGap;
        pass(q0, t0, advanced);
        fq = q0 + advanced; ft = t0 + advanced; fl = maxGap;
        qStart += advanced; tStart += advanced;
      }
      long ll = el;
      if (el > maxGap) {                                                // :198-211
        usedAlt = true;
        altQ = eq + maxGap; altT = et + maxGap; altL = el - maxGap;
        ll = maxGap;
        qPos = eq + maxGap; tPos = et + maxGap;
      }
      const long qEnd = eq + ll, tEnd = et + ll;
      const long tLen = tPos - tStart;
      const long tSeqLen = tEnd - tStart, qSeqLen = qEnd - qStart;
      const bool aog = (tSeqLen < k || qSeqLen < k);                    // :344
      if (EMIT) {
        const uint64_t s = so + nseg;
        A.s_aln[s] = a; A.s_kind[s] = aog ? 1 : 0;
        A.s_qStart[s] = (int)qStart; A.s_tStart[s] = (int)tStart; A.s_qEnd[s] = (int)qEnd; A.s_tEnd[s] = (int)tEnd;
        A.s_b0[s] = (int)startBlock; A.s_b1[s] = (int)endBlock;
        A.s_first[3 * s] = (int)fq; A.s_first[3 * s + 1] = (int)ft; A.s_first[3 * s + 2] = (int)fl;
        A.s_lastLen[s] = (int)ll;
        A.s_rows[s] = aog ? 0 : (uint64_t)(tLen > 0 ? tLen : 0);
        A.s_isAog[s] = aog ? 1u : 0u;
        A.i_kind[io + nitem] = 1; A.i_data[3 * (io + nitem)] = (int)(nseg);   // segment index local to the alignment
      }
      nitem++; nseg++;
    }
    if (!usedAlt) endBlock++;                                           // :761-766
    else { ovIdx = endBlock; ovQ = altQ; ovT = altT; ovL = altL; }
    startBlock = endBlock;
  }
  if (!EMIT) { A.a_nseg[a] = nseg; A.a_nitem[a] = nitem; A.a_status[a] = 0; }
}

// ---------------------------------------------------------------------------------- band
struct BandArgs {
  uint64_t n_seg;
  const int32_t* s_aln; const int32_t* s_kind; const int32_t* s_qStart; const int32_t* s_qEnd;
  const int32_t* s_b0; const int32_t* s_b1; const int32_t* s_first; const int32_t* s_lastLen;
  const uint64_t* s_rows; const uint64_t* s_row_off;
  const int32_t* ab; const uint64_t* block_off;
  int k;
  int32_t* rowS; int32_t* rowE; uint32_t* rowC;     // per row: window start / end (absolute q), cell offset in segment
  uint64_t* s_cells; int32_t* s_status;
};

__global__ void __launch_bounds__(64) ir_band(BandArgs B) {
  __shared__ int ringS[RING], ringE[RING];
  const int lane = threadIdx.x;
  for (uint64_t s = blockIdx.x; s < B.n_seg; s += gridDim.x) {
    if (B.s_kind[s] != 0) { if (lane == 0) { B.s_cells[s] = 0; B.s_status[s] = 0; } continue; }
    const long tLen = (long)B.s_rows[s];
    const int a = B.s_aln[s];
    const int32_t* ab = B.ab + 3 * (B.block_off[a] + 2 * (uint64_t)a);
    const long b0 = B.s_b0[s], b1 = B.s_b1[s];
    const long qStart = B.s_qStart[s], qEnd = B.s_qEnd[s];
    const int k = B.k;
    int32_t* gS = B.rowS + B.s_row_off[s];
    int32_t* gE = B.rowE + B.s_row_off[s];
    uint32_t* gC = B.rowC + B.s_row_off[s];
    int status = 0;
    for (int x = lane; x < RING; x += 64) { ringS[x] = -1; ringE[x] = -1; }
    wave_sync();
    auto blk = [&](long b, long& q, long& t, long& l) {
      if (b == b0) { q = B.s_first[3 * s]; t = B.s_first[3 * s + 1]; l = B.s_first[3 * s + 2]; }
      else { q = ab[3 * b]; t = ab[3 * b + 1]; l = (b == b1) ? B.s_lastLen[s] : ab[3 * b + 2]; }
    };
    auto flush = [&](long r) {   // row r can no longer change: write it out and recycle its slot
      if (lane == 0 && r >= 0 && r < tLen) { gS[r] = ringS[r & (RING - 1)]; gE[r] = ringE[r & (RING - 1)]; ringS[r & (RING - 1)] = -1; ringE[r & (RING - 1)] = -1; }
    };
    long q, t, l;
    blk(b0, q, t, l);
    long tOff = 0;
    for (long b = b0; b <= b1 && !(status & 1); b++) {                  // :232-315
      long bq, bt, bl;
      blk(b, bq, bt, bl);
      int bqGap = 0, btGap = 0;
      long blockLength = bl;
      if (b < b1) {
        long nq2, nt2, nl2;
        blk(b + 1, nq2, nt2, nl2);
        bqGap = (int)(nq2 - (bq + bl)); btGap = (int)(nt2 - (bt + bl));
        if (bqGap > 0 && btGap > 0) { int c = min(bqGap, btGap); bqGap -= c; btGap -= c; blockLength += c; }
      }
      for (long bi = 0; bi < blockLength; bi++) {                       // :252-283
        if (tOff >= tLen) { status |= 1; break; }
        const int slot = (int)(tOff & (RING - 1));
        if (lane == 0) {
          int lo = (int)max(q - k, qStart);
          ringS[slot] = (ringS[slot] == -1) ? lo : min(ringS[slot], lo);
          if (ringE[slot] == -1 || ringE[slot] < q + k) ringE[slot] = (int)min(qEnd - 1, q + k);
        }
        wave_sync();
        if (lane < k) {
          if (tOff - lane >= 0) { int sl = (int)((tOff - lane) & (RING - 1)); if (ringE[sl] < q) ringE[sl] = (int)q; }
          if (tOff + lane < tLen) { int sl = (int)((tOff + lane) & (RING - 1)); if (ringS[sl] == -1 || ringS[sl] > q) ringS[sl] = (int)q; }
        }
        wave_sync();
        tOff++; q++; t++;
        flush(tOff - k);
      }
      if (bqGap > btGap) {                                              // :287-305
        for (int qi = 0; qi < bqGap; qi++, q++) {
          if (lane < k) {
            if (tOff - lane >= 0 && tOff - lane < tLen) { int sl = (int)((tOff - lane) & (RING - 1)); if (ringE[sl] < q) ringE[sl] = (int)q; }
            if (tOff + lane < tLen) { int sl = (int)((tOff + lane) & (RING - 1)); if (ringS[sl] == 0 || ringS[sl] > q) ringS[sl] = (int)q; }   // (sic) == 0
          }
          wave_sync();
        }
      }
      if (btGap > bqGap) {                                              // :306-314
        for (int ti = 0; ti < btGap; ti++) {
          if (tOff >= tLen) { status |= 1; break; }
          if (lane == 0) { int sl = (int)(tOff & (RING - 1)); ringS[sl] = (int)max(q - k, qStart); ringE[sl] = (int)min(qEnd - 1, q + k); }
          wave_sync();
          tOff++; t++;
          flush(tOff - k);
        }
      }
    }
    wave_sync();
    for (long r = max(0L, tOff - k + 1); r < tLen; r++) flush(r);
    wave_sync();
    // ---- :318-322 suffix minimum of qS (back to front, 64 rows at a time)
    int carry = 0x7fffffff;
    for (long base = ((tLen - 1) / 64) * 64; base >= 0; base -= 64) {
      long r = base + lane;
      int v = (r < tLen) ? gS[r] : 0x7fffffff;
      for (int d = 1; d < 64; d <<= 1) { int o = __shfl_down(v, d); if (lane + d < 64) v = min(v, o); }
      v = min(v, carry);
      if (r < tLen) gS[r] = v;
      carry = __shfl(v, 0);
    }
    // ---- :323-328 prefix maximum of qE, row lengths, cell offsets
    int carryE = -0x7fffffff;
    unsigned long long cells = 0;
    for (long base = 0; base < tLen; base += 64) {
      long r = base + lane;
      int e = (r < tLen) ? gE[r] : -0x7fffffff;
      for (int d = 1; d < 64; d <<= 1) { int o = __shfl_up(e, d); if (lane >= d) e = max(e, o); }
      e = max(e, carryE);
      carryE = __shfl(e, 63);
      int len = 0;
      if (r < tLen) {
        gE[r] = e;
        len = e - gS[r] + 1;
        if (len < 1 || len > 64 || gS[r] < 0) status |= (len > 64 ? 4 : 1);
      }
      unsigned int incl = (unsigned int)max(len, 0);
      for (int d = 1; d < 64; d <<= 1) { unsigned int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
      if (r < tLen) gC[r] = (uint32_t)(cells + incl - (unsigned int)max(len, 0));
      cells += __shfl(incl, 63);
    }
    for (int off = 32; off > 0; off >>= 1) status |= __shfl_xor(status, off);
    if (tOff != tLen) status |= 1;
    if (lane == 0) { B.s_cells[s] = (status ? 0 : cells); B.s_status[s] = status; }
    wave_sync();
  }
}

// ---------------------------------------------------------------------------------- fill
struct FillArgs {
  uint64_t n_seg;
  const int32_t* s_aln; const int32_t* s_kind; const int32_t* s_tStart; const uint64_t* s_rows; const uint64_t* s_row_off;
  const uint64_t* s_cells; const uint64_t* s_cell_off; const int32_t* s_status;
  const int32_t* rowS; const int32_t* rowE; const uint32_t* rowC;
  const char* qseq; const uint64_t* q_off; const char* tseq; const uint64_t* t_off;
  int match, mismatch, g;
  unsigned char* path;
};

__device__ __forceinline__ bool is_bound(long row, int c, int len) { return c == len - 1 || (row > 0 && c == 0); }

__global__ void __launch_bounds__(64) ir_fill(FillArgs F) {
  const int lane = threadIdx.x;
  const int g = F.g, go = 2 * F.g + 1;
  for (uint64_t s = blockIdx.x; s < F.n_seg; s += gridDim.x) {
    if (F.s_kind[s] != 0 || F.s_status[s] != 0) continue;
    const long tLen = (long)F.s_rows[s];
    const int a = F.s_aln[s];
    const int32_t* gS = F.rowS + F.s_row_off[s];
    const int32_t* gE = F.rowE + F.s_row_off[s];
    const uint32_t* gC = F.rowC + F.s_row_off[s];
    const unsigned char* qb = (const unsigned char*)F.qseq + F.q_off[a];
    const unsigned char* tb = (const unsigned char*)F.tseq + F.t_off[a] + F.s_tStart[s];
    unsigned char* P = F.path + F.s_cell_off[s];
    // 64-row chunk of per-row data, one row per lane
    int cS = 0, cE = 0; unsigned int cC = 0; int cT = 0;
    auto load_chunk = [&](long base) {
      long r = base + lane;
      if (r < tLen) { cS = gS[r]; cE = gE[r]; cC = gC[r]; cT = tb[r]; }
    };
    load_chunk(0);
    // ---- row 0 (:407-431)
    int S = __shfl(cS, 0), E = __shfl(cE, 0);
    unsigned int C0 = __shfl(cC, 0);
    int len = E - S + 1;
    int prevM, prevD;
    {
      const bool last = (lane == len - 1) && (tLen > 1);
      prevM = last ? BAD : (lane == 0 ? 0 : lane * g);
      prevD = BAD;
      int code = last ? C_BOUND : (lane == 0 ? C_DONE : C_LEFT);
      if (lane < len) P[C0 + lane] = (unsigned char)code;
    }
    int prevS = S, prevLen = len;
    for (long ti = 1; ti < tLen; ti++) {                                // :438-622
      if ((ti & 63) == 0) load_chunk(ti);
      const int src = (int)(ti & 63);
      S = __shfl(cS, src); E = __shfl(cE, src);
      const unsigned int C = __shfl(cC, src);
      const int tch = __shfl(cT, src);
      len = E - S + 1;
      const int off = S - prevS;
      const bool lastRow = (ti == tLen - 1);
      const int c = lane;
      const bool interior = c >= 1 && (lastRow ? c <= len - 1 : c <= len - 2);
      const int srcA = c + off, srcD = srcA - 1;
      const bool aboveIn = srcA <= prevLen - 1;                          // qE[ti-1] >= q   (:491,:548,:567)
      const int aM = __shfl(prevM, srcA & 63), aD = __shfl(prevD, srcA & 63), dM = __shfl(prevM, srcD & 63);
      const bool okA = aboveIn && !is_bound(ti - 1, srcA, prevLen);
      const bool okD = aboveIn && srcD >= 0 && !is_bound(ti - 1, srcD, prevLen);
      const int dOpen = okA ? aM + go : BAD, dExt = okA ? aD : BAD;      // :491-502 (gapExtend = 0)
      const int Dv = max(dOpen, dExt);
      const int delOpen = (Dv == dOpen) ? 1 : 0;                         // :504-516
      int qch = 0;
      if (interior) qch = qb[S + c];
      const int mS = okD ? dM + (tch == qch ? F.match : F.mismatch) : BAD;   // :548-563
      const int dS = okA ? aM + g : BAD;                                     // :567-574
      const int V = interior ? max(mS, max(dS, Dv)) : NEG;
      int W = V;                                                         // inclusive prefix max of V
      for (int d = 1; d < 64; d <<= 1) { int o = __shfl_up(W, d); if (lane >= d) W = max(W, o); }
      int Wm1 = __shfl_up(W, 1), Vm1 = __shfl_up(V, 1);
      if (lane == 0) { Wm1 = NEG; Vm1 = NEG; }
      const int Iv = max(BAD, go + Wm1);
      int M = max(max(BAD, V), max(Vm1 + g, go + Wm1));
      if (!interior) M = BAD;
      int Mleft = __shfl_up(M, 1);
      if (c <= 1) Mleft = BAD;                                           // the row's left boundary cell (:413-418)
      const int iOpen = Mleft + go;                                      // :523
      const int insOpen = (Iv == iOpen) ? 1 : 0;                         // :528-540
      const int iS = Mleft + g;                                          // :565
      int code;
      if (!interior) code = C_BOUND;
      else if (M == mS) code = C_DIAG;                                   // :583-616
      else if (M == iS) code = C_LEFT;
      else if (M == dS) code = C_DOWN;
      else if (M == Dv) code = C_DELCLOSE;
      else code = C_INSCLOSE;
      if (c < len) P[C + c] = (unsigned char)(code | (delOpen << 3) | (insOpen << 4));
      prevM = M;
      prevD = interior ? Dv : BAD;
      prevS = S; prevLen = len;
    }
  }
}

// ---------------------------------------------------------------------------------- trace
struct TraceArgs {
  uint64_t n_seg;
  const int32_t* s_kind; const int32_t* s_qStart; const int32_t* s_tStart; const uint64_t* s_rows; const uint64_t* s_row_off;
  const uint64_t* s_cells; const uint64_t* s_cell_off; int32_t* s_status;
  const int32_t* rowS; const int32_t* rowE; const uint32_t* rowC;
  const unsigned char* path;
  uint32_t* s_nblk; uint32_t* s_nq; uint32_t* s_nt;   // count pass outputs: blocks, q consumed, t consumed
  const uint64_t* s_out_off; int32_t* out_blocks;     // emit pass
};

template <bool EMIT>
__global__ void __launch_bounds__(64) ir_trace(TraceArgs T) {
  __shared__ int lS[64], lLen[64];
  __shared__ unsigned int lC[64];
  __shared__ unsigned char lP[64 * 64];
  const int lane = threadIdx.x;
  for (uint64_t s = blockIdx.x; s < T.n_seg; s += gridDim.x) {
    if (T.s_kind[s] != 0) continue;
    if (T.s_status[s] != 0) { if (!EMIT && lane == 0) { T.s_nblk[s] = 0; T.s_nq[s] = 0; T.s_nt[s] = 0; } continue; }
    const long tLen = (long)T.s_rows[s];
    const int32_t* gS = T.rowS + T.s_row_off[s];
    const int32_t* gE = T.rowE + T.s_row_off[s];
    const uint32_t* gC = T.rowC + T.s_row_off[s];
    const unsigned char* P = T.path + T.s_cell_off[s];
    // walk state (meaningful on lane 0, broadcast at chunk boundaries): row ti, absolute read
    // position qa of the current cell, current matrix
    int ti = (int)(tLen - 1);
    int qa = gE[tLen - 1];                    // last cell of the matrix (:629)
    int mat = 0;                              // 0 match, 1 del, 2 ins
    int done = 0, bad = 0;
    // block assembly back to front: positions are forward coordinates
    unsigned int nblk = 0, nD = 0, nL = 0, nU = 0;
    long q = 0, t = 0;
    long outIdx = 0;
    int32_t* ob = nullptr;
    if (EMIT) {
      q = (long)T.s_qStart[s] + T.s_nq[s]; t = (long)T.s_tStart[s] + T.s_nt[s];
      outIdx = (long)T.s_nblk[s] - 1;
      ob = T.out_blocks + 3 * T.s_out_off[s];
    }
    int curKind = -1, pending = 0;
    long dlen = 0;
    long steps = 0;
    const long step_cap = 4 * (long)T.s_cells[s] + 64;
    auto emit_block = [&]() {
      if (EMIT) { if (outIdx >= 0) { ob[3 * outIdx] = (int)q; ob[3 * outIdx + 1] = (int)t; ob[3 * outIdx + 2] = (int)dlen; } outIdx--; }
      nblk++;
    };
    // The forward parse (:718-745) is: [diag run][one run of left OR of down] -> one block.  Seen
    // back to front: every gap run closes the block whose diag run (possibly empty) precedes it,
    // and a trailing diag run is a block of its own.
    auto op = [&](int kind) {                 // kind: 0 diag, 1 left, 2 down
      if (kind != curKind) {
        if (curKind == -1 && kind == 0) { pending = 1; dlen = 0; }
        if (kind != 0) {
          if (pending) emit_block();
          pending = 1; dlen = 0;
        }
        curKind = kind;
      }
      if (kind == 0) { dlen++; q--; t--; nD++; }
      else if (kind == 1) { q--; nL++; }
      else { t--; nU++; }
    };
    while (!done) {
      // stage rows [lo, hi] = [max(0, ti-63), ti]
      const int hi = ti, lo = max(0, ti - 63);
      wave_sync();
      {
        int r = lo + lane;
        if (r <= hi) { lS[lane] = gS[r]; lC[lane] = gC[r]; lLen[lane] = gE[r] - gS[r] + 1; }
      }
      wave_sync();
      const unsigned int cbase = lC[0];
      const unsigned int cend = lC[hi - lo] + (unsigned int)lLen[hi - lo];
      for (unsigned int x = cbase + lane; x < cend; x += 64) lP[x - cbase] = P[x];
      wave_sync();
      if (lane == 0) {
        while (ti >= lo) {
          const int ri = ti - lo;
          const int c = qa - lS[ri];
          if (c < 0 || c >= lLen[ri]) { bad = 1; done = 1; break; }
          if (ti == 0 && c == 0) { done = 1; break; }                   // flat index 0 (:631)
          if (++steps > step_cap) { bad = 1; done = 1; break; }
          const unsigned char pb = lP[lC[ri] - cbase + c];
          if (mat == 0) {                                               // :632-648
            const int code = pb & 7;
            if (code == C_DELCLOSE) mat = 1;
            else if (code == C_INSCLOSE) mat = 2;
            else if (code == C_DIAG) { op(0); ti--; qa--; }
            else if (code == C_LEFT) { op(1); qa--; }
            else if (code == C_DOWN) { op(2); ti--; }
            else { bad = 1; done = 1; break; }                          // boundary arrow: endless loop in the reference
          } else if (mat == 1) {                                        // :649-659
            op(2);
            mat = ((pb >> 3) & 1) ? 0 : 1;
            ti--;
          } else {                                                      // :660-671
            op(1);
            mat = ((pb >> 4) & 1) ? 0 : 2;
            qa--;
          }
        }
        if (ti < 0) { bad = 1; done = 1; }
      }
      ti = __shfl(ti, 0); qa = __shfl(qa, 0); done = __shfl(done, 0);
    }
    if (lane == 0) {
      op(0);                                                            // the aligned first base (:674)
      if (pending) emit_block();
      if (bad) T.s_status[s] |= 2;
      if (!EMIT) { T.s_nblk[s] = bad ? 0 : nblk; T.s_nq[s] = nD + nL; T.s_nt[s] = nD + nU; }
    }
    wave_sync();
  }
}

// ---------------------------------------------------------------------------------- gather
struct GatherArgs {
  uint64_t n_item;
  const int32_t* i_kind; const int32_t* i_data; const int32_t* i_aln; const uint64_t* i_out_off;
  const uint64_t* seg_off;            // per alignment
  const int32_t* s_kind; const int32_t* s_qStart; const int32_t* s_tStart; const uint32_t* s_aog_idx;
  const int32_t* aog_blocks; const uint64_t* aog_block_off; const int32_t* aog_nblocks;
  int32_t* out_blocks;
};

// per item: number of output blocks
__global__ void ir_item_counts(uint64_t n_item, const int32_t* i_kind, const int32_t* i_data, const int32_t* i_aln, const uint64_t* seg_off,
                               const int32_t* s_kind, const uint32_t* s_nblk, const uint32_t* s_aog_idx, const int32_t* aog_nblocks,
                               uint32_t* i_count) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_item) return;
  if (i_kind[i] == 0) { i_count[i] = 1; return; }
  uint64_t s = seg_off[i_aln[i]] + (uint64_t)i_data[3 * i];
  i_count[i] = (s_kind[s] == 0) ? s_nblk[s] : (uint32_t)aog_nblocks[s_aog_idx[s]];
}

__global__ void __launch_bounds__(64) ir_gather(GatherArgs G) {
  const int lane = threadIdx.x;
  for (uint64_t i = blockIdx.x; i < G.n_item; i += gridDim.x) {
    int32_t* out = G.out_blocks + 3 * G.i_out_off[i];
    if (G.i_kind[i] == 0) {
      if (lane < 3) out[lane] = G.i_data[3 * i + lane];
      continue;
    }
    uint64_t s = G.seg_off[G.i_aln[i]] + (uint64_t)G.i_data[3 * i];
    if (G.s_kind[s] == 0) continue;                                     // DP segments were written in place by ir_trace
    const uint32_t p = G.s_aog_idx[s];
    const int32_t* src = G.aog_blocks + 3 * G.aog_block_off[p];
    const int n = G.aog_nblocks[p];
    const int qs = G.s_qStart[s], ts = G.s_tStart[s];
    for (int x = lane; x < n; x += 64) {                                // :352-356
      out[3 * x] = src[3 * x] + qs; out[3 * x + 1] = src[3 * x + 1] + ts; out[3 * x + 2] = src[3 * x + 2];
    }
  }
}

// out offsets per alignment (= of its first item) and per DP segment (= of its item)
__global__ void ir_finalize_offsets(int n_aln, uint64_t n_item, const uint64_t* item_off, const uint64_t* i_out_off, const int32_t* i_kind,
                                    const int32_t* i_data, const int32_t* i_aln, const uint64_t* seg_off, uint64_t* s_out_off,
                                    uint64_t* out_block_off) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= (uint64_t)n_aln) out_block_off[i] = i_out_off[i < (uint64_t)n_aln ? item_off[i] : n_item];
  if (i < n_item && i_kind[i] == 1) s_out_off[seg_off[i_aln[i]] + (uint64_t)i_data[3 * i]] = i_out_off[i];
}

__global__ void ir_aln_status(int n_aln, const uint64_t* seg_off, const int32_t* s_kind, const int32_t* s_status, const uint32_t* s_aog_idx,
                              const int32_t* p_status, int32_t* a_status) {
  int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n_aln) return;
  int st = 0;
  for (uint64_t s = seg_off[a]; s < seg_off[a + 1]; s++) {
    if (s_kind[s] == 0) { int x = s_status[s]; if (x & 1) st |= LRA_ST_OOB_SLOT; if (x & 2) st |= LRA_ST_NO_TERMINATION; if (x & 4) st |= LRA_ST_RANGE; }
    else st |= p_status[s_aog_idx[s]];
  }
  a_status[a] = st;
}

// item -> alignment map, AOG problem descriptors
__global__ void ir_item_aln(int n_aln, const uint64_t* item_off, int32_t* i_aln) {
  int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n_aln) return;
  for (uint64_t i = item_off[a]; i < item_off[a + 1]; i++) i_aln[i] = a;
}

__global__ void ir_aog_setup(uint64_t n_seg, const int32_t* s_kind, const int32_t* s_aln, const uint64_t* s_aog_off /* exclusive scan of isAog */,
                             const int32_t* s_qStart, const int32_t* s_tStart, const int32_t* s_qEnd, const int32_t* s_tEnd,
                             const uint64_t* q_off, const uint64_t* t_off, int k, uint32_t* s_aog_idx, uint64_t* p_q_off, int32_t* p_q_len,
                             uint64_t* p_t_off, int32_t* p_t_len, int32_t* p_k, uint32_t* p_cap) {
  uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_seg) return;
  if (s_kind[s] != 1) { s_aog_idx[s] = 0; return; }
  uint32_t p = (uint32_t)s_aog_off[s];
  s_aog_idx[s] = p;
  int a = s_aln[s];
  p_q_off[p] = q_off[a] + (uint64_t)s_qStart[s]; p_q_len[p] = s_qEnd[s] - s_qStart[s];
  p_t_off[p] = t_off[a] + (uint64_t)s_tStart[s]; p_t_len[p] = s_tEnd[s] - s_tStart[s];
  p_k[p] = k;
  p_cap[p] = (uint32_t)(min(p_q_len[p], p_t_len[p]) + 1);
}

template <typename CT>
__global__ void __launch_bounds__(1024) scan_kernel(long n, const CT* __restrict__ counts, uint64_t* __restrict__ off) {
  __shared__ uint64_t part[1024];
  const int t = threadIdx.x;
  const long per = (n + 1023) / 1024;
  const long lo = min((long)t * per, n), hi = min(lo + per, n);
  uint64_t s = 0;
  for (long i = lo; i < hi; i++) s += (uint64_t)counts[i];
  part[t] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    uint64_t v = (t >= d) ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  uint64_t run = (t == 0) ? 0 : part[t - 1];
  for (long i = lo; i < hi; i++) { off[i] = run; run += (uint64_t)counts[i]; }
  if (t == 1023) off[n] = part[1023];
}

// simple device bump allocator over one scratch arena (slot 2), 256-byte aligned
struct Arena {
  char* base; size_t cap, used;
  template <typename T> T* get(size_t n) {
    size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
    if (used + bytes > cap) return nullptr;
    T* p = (T*)(base + used);
    used += bytes;
    return p;
  }
};

}  // namespace

template <typename CT>
static void scan(lra_ctx* ctx, long n, const CT* c, uint64_t* off) {
  hipLaunchKernelGGL(scan_kernel<CT>, dim3(1), dim3(1024), 0, ctx->stream, n, c, off);
}

static int d2h(lra_ctx* ctx, void* dst, const void* src, size_t bytes) {
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return LRA_OK;
}

extern "C" int lra_indel_refine_batch(lra_ctx* ctx, int n_aln, const int32_t* d_blocks_in, const uint64_t* d_block_off, uint64_t n_blocks_in,
                                      const char* d_qseq, const uint64_t* d_q_off, const int32_t* d_q_len, const char* d_tseq,
                                      const uint64_t* d_t_off, const int64_t* d_t_len, int refine_band, int match, int mismatch,
                                      int indel, int end_align, lra_refine_result* out) {
  if (!ctx || !out || n_aln < 0) return LRA_ERR_INVALID;
  if (refine_band < 2 || refine_band > 64) return lra_set_err(ctx, LRA_ERR_INVALID, "refine_band must be 2..64");
  if (indel >= 0) return lra_set_err(ctx, LRA_ERR_INVALID, "indel score must be negative");
  memset(out, 0, sizeof(*out));
  if (n_aln == 0) return LRA_OK;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t nA = (size_t)n_aln, nBk = (size_t)n_blocks_in + 2 * nA;   // augmented blocks
  // ---- arena A (slot 2): everything whose size is bounded by the input
  size_t needA = 0;
  auto add = [&](size_t n, size_t sz) { needA += ((n * sz + 255) & ~(size_t)255); };
  add(nA, 4); add(nA, 4); add(nA, 4); add(nA + 1, 8); add(nA + 1, 8);                 // a_nseg a_nitem a_status seg_off item_off
  add(3 * nBk, 4);                                                                    // ab
  const size_t capSeg = nBk, capItem = 2 * nBk + nA;
  add(capSeg, 4); add(capSeg, 4); add(capSeg, 4); add(capSeg, 4); add(capSeg, 4); add(capSeg, 4); add(capSeg, 4); add(capSeg, 4);
  add(3 * capSeg, 4); add(capSeg, 4); add(capSeg, 8); add(capSeg, 4);                 // first lastLen rows isAog
  add(capSeg + 1, 8); add(capSeg + 1, 8); add(capSeg, 8); add(capSeg + 1, 8); add(capSeg, 4);   // row_off aog_off cells cell_off status
  add(capSeg, 4); add(capSeg, 4); add(capSeg, 4); add(capSeg, 4); add(capSeg + 1, 8);           // nblk nq nt aog_idx out_off(seg)
  add(capItem, 4); add(3 * capItem, 4); add(capItem, 4); add(capItem, 4); add(capItem + 1, 8);  // i_kind i_data i_aln i_count i_out_off
  add(capSeg, 8); add(capSeg, 4); add(capSeg, 8); add(capSeg, 4); add(capSeg, 4); add(capSeg, 4); add(capSeg + 1, 8);   // aog problem arrays
  add(capSeg, 4); add(capSeg, 4); add(capSeg, 4);                                               // aog score nblocks status
  add(nA + 1, 8);                                                                               // out block_off per alignment
  char* baseA = (char*)lra_scratch(ctx, 2, needA + 4096);
  if (!baseA) return LRA_ERR_NOMEM;
  Arena ar{baseA, needA + 4096, 0};
  IRArgs A;
  A.n_aln = n_aln; A.blocks_in = d_blocks_in; A.block_off = d_block_off;
  A.qseq = d_qseq; A.q_off = d_q_off; A.q_len = d_q_len; A.tseq = d_tseq; A.t_off = d_t_off; A.t_len = d_t_len;
  A.k = refine_band; A.match = match; A.mismatch = mismatch; A.indel = indel; A.endAlign = end_align;
  A.a_nseg = ar.get<uint32_t>(nA); A.a_nitem = ar.get<uint32_t>(nA); A.a_status = ar.get<int32_t>(nA);
  uint64_t* seg_off = ar.get<uint64_t>(nA + 1); uint64_t* item_off = ar.get<uint64_t>(nA + 1);
  A.seg_off = seg_off; A.item_off = item_off;
  A.ab = ar.get<int32_t>(3 * nBk);
  A.s_aln = ar.get<int32_t>(capSeg); A.s_kind = ar.get<int32_t>(capSeg); A.s_qStart = ar.get<int32_t>(capSeg); A.s_tStart = ar.get<int32_t>(capSeg);
  A.s_qEnd = ar.get<int32_t>(capSeg); A.s_tEnd = ar.get<int32_t>(capSeg); A.s_b0 = ar.get<int32_t>(capSeg); A.s_b1 = ar.get<int32_t>(capSeg);
  A.s_first = ar.get<int32_t>(3 * capSeg); A.s_lastLen = ar.get<int32_t>(capSeg); A.s_rows = ar.get<uint64_t>(capSeg); A.s_isAog = ar.get<uint32_t>(capSeg);
  uint64_t* s_row_off = ar.get<uint64_t>(capSeg + 1); uint64_t* s_aog_off = ar.get<uint64_t>(capSeg + 1);
  uint64_t* s_cells = ar.get<uint64_t>(capSeg); uint64_t* s_cell_off = ar.get<uint64_t>(capSeg + 1); int32_t* s_status = ar.get<int32_t>(capSeg);
  uint32_t* s_nblk = ar.get<uint32_t>(capSeg); uint32_t* s_nq = ar.get<uint32_t>(capSeg); uint32_t* s_nt = ar.get<uint32_t>(capSeg);
  uint32_t* s_aog_idx = ar.get<uint32_t>(capSeg); uint64_t* s_out_off = ar.get<uint64_t>(capSeg + 1);
  A.i_kind = ar.get<int32_t>(capItem); A.i_data = ar.get<int32_t>(3 * capItem);
  int32_t* i_aln = ar.get<int32_t>(capItem); uint32_t* i_count = ar.get<uint32_t>(capItem); uint64_t* i_out_off = ar.get<uint64_t>(capItem + 1);
  uint64_t* p_q_off = ar.get<uint64_t>(capSeg); int32_t* p_q_len = ar.get<int32_t>(capSeg); uint64_t* p_t_off = ar.get<uint64_t>(capSeg);
  int32_t* p_t_len = ar.get<int32_t>(capSeg); int32_t* p_k = ar.get<int32_t>(capSeg); uint32_t* p_cap = ar.get<uint32_t>(capSeg);
  uint64_t* p_block_off = ar.get<uint64_t>(capSeg + 1);
  int32_t* p_score = ar.get<int32_t>(capSeg); int32_t* p_nblocks = ar.get<int32_t>(capSeg); int32_t* p_status = ar.get<int32_t>(capSeg);
  uint64_t* out_block_off = ar.get<uint64_t>(nA + 1);
  if (!out_block_off) return lra_set_err(ctx, LRA_ERR_NOMEM, "arena accounting");

  const int nbA = (n_aln + 63) / 64;
  // ---- segments: count, scan, emit
  lra_time_begin(ctx, "ir_segment");
  hipLaunchKernelGGL(ir_segment<false>, dim3(nbA), dim3(64), 0, st, A);
  scan(ctx, (long)n_aln, A.a_nseg, seg_off);
  scan(ctx, (long)n_aln, A.a_nitem, item_off);
  hipLaunchKernelGGL(ir_segment<true>, dim3(nbA), dim3(64), 0, st, A);
  lra_time_end(ctx);
  uint64_t n_seg = 0, n_item = 0;
  if (d2h(ctx, &n_seg, seg_off + n_aln, 8) || d2h(ctx, &n_item, item_off + n_aln, 8)) return LRA_ERR_HIP;
  if (n_seg > capSeg || n_item > capItem) return lra_set_err(ctx, LRA_ERR_INVALID, "segment accounting");
  hipLaunchKernelGGL(ir_item_aln, dim3((n_aln + 255) / 256), dim3(256), 0, st, n_aln, item_off, i_aln);
  uint64_t n_rows = 0, n_aog = 0, n_cells = 0;
  int32_t* rowS = nullptr; int32_t* rowE = nullptr; uint32_t* rowC = nullptr;
  if (n_seg) {
    scan(ctx, (long)n_seg, A.s_rows, s_row_off);
    scan(ctx, (long)n_seg, A.s_isAog, s_aog_off);
    if (d2h(ctx, &n_rows, s_row_off + n_seg, 8) || d2h(ctx, &n_aog, s_aog_off + n_seg, 8)) return LRA_ERR_HIP;
    // ---- arena B (slot 3): rows, then cells
    size_t rowBytes = ((n_rows * 4 + 255) & ~(size_t)255);
    char* baseB = (char*)lra_scratch(ctx, 3, 3 * rowBytes + 4096);
    if (!baseB) return LRA_ERR_NOMEM;
    rowS = (int32_t*)baseB; rowE = (int32_t*)(baseB + rowBytes); rowC = (uint32_t*)(baseB + 2 * rowBytes);
    BandArgs B;
    B.n_seg = n_seg; B.s_aln = A.s_aln; B.s_kind = A.s_kind; B.s_qStart = A.s_qStart; B.s_qEnd = A.s_qEnd; B.s_b0 = A.s_b0; B.s_b1 = A.s_b1;
    B.s_first = A.s_first; B.s_lastLen = A.s_lastLen; B.s_rows = A.s_rows; B.s_row_off = s_row_off; B.ab = A.ab; B.block_off = d_block_off;
    B.k = refine_band; B.rowS = rowS; B.rowE = rowE; B.rowC = rowC; B.s_cells = s_cells; B.s_status = s_status;
    const unsigned gridW = (unsigned)std::min<uint64_t>(n_seg, (uint64_t)ctx->num_cu * 32);
    lra_time_begin(ctx, "ir_band");
    hipLaunchKernelGGL(ir_band, dim3(gridW), dim3(64), 0, st, B);
    lra_time_end(ctx);
    scan(ctx, (long)n_seg, s_cells, s_cell_off);
    if (d2h(ctx, &n_cells, s_cell_off + n_seg, 8)) return LRA_ERR_HIP;
    // path bytes live after the row arrays; re-fetch the arena in case it must grow
    size_t needB = 3 * rowBytes + n_cells + 4096;
    if (needB > ctx->scratch_bytes[3]) {
      // grow while keeping the row arrays: allocate new, copy, free old
      void* nb = nullptr;
      size_t want = needB + needB / 4;
      if (hipMalloc(&nb, want) != hipSuccess) return lra_set_err(ctx, LRA_ERR_NOMEM, "trace-back arena (%zu bytes)", want);
      LRA_HIP_CHECK(ctx, hipMemcpyAsync(nb, baseB, 3 * rowBytes, hipMemcpyDeviceToDevice, st));
      LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
      (void)hipFree(ctx->scratch[3]);
      ctx->scratch[3] = nb; ctx->scratch_bytes[3] = want;
      baseB = (char*)nb;
      rowS = (int32_t*)baseB; rowE = (int32_t*)(baseB + rowBytes); rowC = (uint32_t*)(baseB + 2 * rowBytes);
    }
    unsigned char* path = (unsigned char*)(baseB + 3 * rowBytes);
    FillArgs F;
    F.n_seg = n_seg; F.s_aln = A.s_aln; F.s_kind = A.s_kind; F.s_tStart = A.s_tStart; F.s_rows = A.s_rows; F.s_row_off = s_row_off;
    F.s_cells = s_cells; F.s_cell_off = s_cell_off; F.s_status = s_status; F.rowS = rowS; F.rowE = rowE; F.rowC = rowC;
    F.qseq = d_qseq; F.q_off = d_q_off; F.tseq = d_tseq; F.t_off = d_t_off; F.match = match; F.mismatch = mismatch; F.g = indel; F.path = path;
    lra_time_begin(ctx, "ir_fill");
    hipLaunchKernelGGL(ir_fill, dim3(gridW), dim3(64), 0, st, F);
    lra_time_end(ctx);
    TraceArgs T;
    T.n_seg = n_seg; T.s_kind = A.s_kind; T.s_qStart = A.s_qStart; T.s_tStart = A.s_tStart; T.s_rows = A.s_rows; T.s_row_off = s_row_off;
    T.s_cells = s_cells; T.s_cell_off = s_cell_off; T.s_status = s_status; T.rowS = rowS; T.rowE = rowE; T.rowC = rowC; T.path = path;
    T.s_nblk = s_nblk; T.s_nq = s_nq; T.s_nt = s_nt; T.s_out_off = nullptr; T.out_blocks = nullptr;
    lra_time_begin(ctx, "ir_trace_count");
    hipLaunchKernelGGL(ir_trace<false>, dim3(gridW), dim3(64), 0, st, T);
    lra_time_end(ctx);
    // ---- short segments -> AffineOneGapAlign (:344-357)
    int32_t* aog_blocks = nullptr;
    if (n_aog) {
      hipLaunchKernelGGL(ir_aog_setup, dim3((unsigned)((n_seg + 255) / 256)), dim3(256), 0, st, n_seg, A.s_kind, A.s_aln, s_aog_off, A.s_qStart,
                         A.s_tStart, A.s_qEnd, A.s_tEnd, d_q_off, d_t_off, refine_band, s_aog_idx, p_q_off, p_q_len, p_t_off, p_t_len, p_k, p_cap);
      scan(ctx, (long)n_aog, p_cap, p_block_off);
      uint64_t aog_cap = 0;
      if (d2h(ctx, &aog_cap, p_block_off + n_aog, 8)) return LRA_ERR_HIP;
      // AOG uses scratch slots 0/1; its blocks go to the tail of arena A's slot? use a dedicated hipMalloc-backed buffer in ctx
      if (aog_cap * 12 + 64 > ctx->aux_bytes) {
        if (ctx->aux) (void)hipFree(ctx->aux);
        ctx->aux = nullptr; ctx->aux_bytes = 0;
        size_t want = aog_cap * 12 + aog_cap * 3 + 4096;
        if (hipMalloc(&ctx->aux, want) != hipSuccess) return lra_set_err(ctx, LRA_ERR_NOMEM, "aog block buffer");
        ctx->aux_bytes = want;
      }
      aog_blocks = (int32_t*)ctx->aux;
      int rc = lra_aog_launch_device(ctx, (int)n_aog, d_qseq, d_tseq, p_q_off, p_q_len, p_t_off, p_t_len, p_k, match, mismatch, indel, p_score,
                                     p_nblocks, aog_blocks, p_block_off, p_status);
      if (rc) return rc;
    }
    // ---- output sizing
    hipLaunchKernelGGL(ir_item_counts, dim3((unsigned)((n_item + 255) / 256)), dim3(256), 0, st, n_item, A.i_kind, A.i_data, i_aln, seg_off, A.s_kind,
                       s_nblk, s_aog_idx, p_nblocks, i_count);
  } else {
    hipLaunchKernelGGL(ir_item_counts, dim3((unsigned)((n_item + 255) / 256)), dim3(256), 0, st, n_item, A.i_kind, A.i_data, i_aln, seg_off, A.s_kind,
                       s_nblk, s_aog_idx, p_nblocks, i_count);
  }
  scan(ctx, (long)n_item, i_count, i_out_off);
  uint64_t n_out = 0;
  if (d2h(ctx, &n_out, i_out_off + n_item, 8)) return LRA_ERR_HIP;
  if ((n_out + 1) * 12 > ctx->out_bytes) {
    if (ctx->out_buf) (void)hipFree(ctx->out_buf);
    ctx->out_buf = nullptr; ctx->out_bytes = 0;
    size_t want = (n_out + 1) * 12 + (n_out + 1) * 3 + 4096;
    if (hipMalloc(&ctx->out_buf, want) != hipSuccess) return lra_set_err(ctx, LRA_ERR_NOMEM, "refined block buffer");
    ctx->out_bytes = want;
  }
  int32_t* out_blocks = (int32_t*)ctx->out_buf;
  // per-alignment offsets = out offset of its first item; per-segment out offset = its item's
  hipLaunchKernelGGL(ir_finalize_offsets, dim3((unsigned)((std::max<uint64_t>(n_item, nA + 1) + 255) / 256)), dim3(256), 0, st, n_aln, n_item, item_off,
                     i_out_off, A.i_kind, A.i_data, i_aln, seg_off, s_out_off, out_block_off);
  if (n_seg) {
    TraceArgs T;
    T.n_seg = n_seg; T.s_kind = A.s_kind; T.s_qStart = A.s_qStart; T.s_tStart = A.s_tStart; T.s_rows = A.s_rows; T.s_row_off = s_row_off;
    T.s_cells = s_cells; T.s_cell_off = s_cell_off; T.s_status = s_status; T.rowS = rowS; T.rowE = rowE; T.rowC = rowC;
    T.path = (unsigned char*)((char*)ctx->scratch[3] + 3 * ((n_rows * 4 + 255) & ~(size_t)255));
    T.s_nblk = s_nblk; T.s_nq = s_nq; T.s_nt = s_nt; T.s_out_off = s_out_off; T.out_blocks = out_blocks;
    const unsigned gridW = (unsigned)std::min<uint64_t>(n_seg, (uint64_t)ctx->num_cu * 32);
    lra_time_begin(ctx, "ir_trace_emit");
    hipLaunchKernelGGL(ir_trace<true>, dim3(gridW), dim3(64), 0, st, T);
    lra_time_end(ctx);
  }
  GatherArgs G;
  G.n_item = n_item; G.i_kind = A.i_kind; G.i_data = A.i_data; G.i_aln = i_aln; G.i_out_off = i_out_off; G.seg_off = seg_off;
  G.s_kind = A.s_kind; G.s_qStart = A.s_qStart; G.s_tStart = A.s_tStart; G.s_aog_idx = s_aog_idx;
  G.aog_blocks = (const int32_t*)ctx->aux; G.aog_block_off = p_block_off; G.aog_nblocks = p_nblocks; G.out_blocks = out_blocks;
  if (n_item) hipLaunchKernelGGL(ir_gather, dim3((unsigned)std::min<uint64_t>(n_item, 65535)), dim3(64), 0, st, G);
  // per-alignment status = OR over its segments
  hipLaunchKernelGGL(ir_aln_status, dim3((n_aln + 255) / 256), dim3(256), 0, st, n_aln, seg_off, A.s_kind, s_status, s_aog_idx, p_status, A.a_status);
  LRA_HIP_CHECK(ctx, hipGetLastError());
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  out->n_aln = n_aln; out->n_blocks = n_out; out->n_segments = n_seg; out->n_cells = n_cells; out->n_rows = n_rows; out->n_aog = n_aog;
  out->d_block_off = out_block_off; out->d_blocks = out_blocks; out->d_status = A.a_status;
  return LRA_OK;
}
