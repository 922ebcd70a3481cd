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
#include "scan.h"
#include <algorithm>
#include <stdlib.h>

int lra_aog_launch_device(lra_ctx* ctx, int n, const char* d_qseq, const char* d_tseq, const uint64_t* d_q_off,
                          const int32_t* d_q_len, const uint64_t* d_t_off, const int32_t* d_t_len, const int32_t* d_k,
                          int m, int mm, int indel, int32_t* d_score, int32_t* d_nblocks, int32_t* d_blocks,
                          const uint64_t* d_block_off, int32_t* d_status);

namespace {

constexpr int BAD = -999999999;       // IndelRefine.h:368
constexpr int NEG = -2000000000;      // "no candidate" sentinel, below every reachable score
enum { C_DIAG = 0, C_LEFT = 1, C_DOWN = 2, C_BOUND = 3, C_DELCLOSE = 4, C_INSCLOSE = 5, C_DONE = 6 };

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// The same for LDS traffic only (a wave's LDS instructions execute in order: nothing has to be waited for -- in particular not the wave's outstanding global stores,
// which wave_sync() drains: a store-to-acknowledge round trip per call)
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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
constexpr int SEG_LANES = 16;             // one lane per alignment walks its blocks: fewer lanes per wave, more waves
template <bool EMIT>
__global__ void __launch_bounds__(64) ir_segment(IRArgs A) {
  if (threadIdx.x >= SEG_LANES) return;
  const int a = blockIdx.x * SEG_LANES + threadIdx.x;
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
struct __attribute__((aligned(16))) Row { int S, E; unsigned int C; int T; };   // window [S,E], cell offset in segment, target base

constexpr int CB = 32;                // blocks per band chunk
constexpr int MAXW = 1024;            // widest row the fill kernels take (rows of more than 64 cells: ir_fill_wide; refineBand 50 of -CONTIG gives ~100-200)

struct BandArgs {
  uint64_t n_seg, n_task;
  const int32_t* s_aln; const int32_t* s_kind; const int32_t* s_qStart; const int32_t* s_qEnd; const int32_t* s_tStart;
  const int32_t* s_b0; const int32_t* s_b1; const int32_t* s_first; const int32_t* s_lastLen;
  const uint64_t* s_rows; const uint64_t* s_row_off;
  const int32_t* ab; const uint64_t* block_off;
  const char* tseq; const uint64_t* t_off;
  int k;
  Row* rows;
  uint64_t* s_cells; int32_t* s_status; int32_t* s_width; uint64_t* s_tmpcap;
  uint32_t* s_nchunk; const uint64_t* chunk_off; uint32_t* task_seg; int32_t* chunkT; int32_t* chunkQ;
};

// The block view of a segment: block b0 is the (possibly trimmed) first block, block b1 carries
// the trimmed last length (IndelRefine.h:178-211).
struct SegBlocks {
  const int32_t* ab; long b0, b1; int fq, ft, fl, lastLen;
  __device__ __forceinline__ void get(long b, long& q, long& t, long& l) const {
    if (b == b0) { q = fq; t = ft; l = fl; }
    else { q = ab[3 * b]; t = ab[3 * b + 1]; l = (b == b1) ? lastLen : ab[3 * b + 2]; }
  }
};
__device__ __forceinline__ SegBlocks seg_blocks(const BandArgs& B, uint64_t s) {
  SegBlocks v;
  const int a = B.s_aln[s];
  v.ab = B.ab + 3 * (B.block_off[a] + 2 * (uint64_t)a);
  v.b0 = B.s_b0[s]; v.b1 = B.s_b1[s];
  v.fq = B.s_first[3 * s]; v.ft = B.s_first[3 * s + 1]; v.fl = B.s_first[3 * s + 2]; v.lastLen = B.s_lastLen[s];
  return v;
}
// What one iteration of the block loop (:232-315) consumes: the gaps to the next block after the
// shared diagonal part `c` is moved into the block (:243-250), the target rows and the query
// positions the iteration advances by.
struct BlockStep { int bqGap, btGap; long blockLength, rows, advq; };
__device__ __forceinline__ BlockStep block_step(const SegBlocks& sb, long b, long bq, long bt, long bl, long& nq2, long& nt2, long& nl2) {
  BlockStep st; st.bqGap = 0; st.btGap = 0; st.blockLength = bl; nq2 = nt2 = nl2 = 0;
  if (b < sb.b1) {
    sb.get(b + 1, nq2, nt2, nl2);
    st.bqGap = (int)(nq2 - (bq + bl)); st.btGap = (int)(nt2 - (bt + bl));
    if (st.bqGap > 0 && st.btGap > 0) { int c = min(st.bqGap, st.btGap); st.bqGap -= c; st.btGap -= c; st.blockLength += c; }
  }
  const long body = st.blockLength > 0 ? st.blockLength : 0;
  st.rows = body + (st.btGap > st.bqGap && st.btGap > 0 ? st.btGap : 0);
  st.advq = body + (st.bqGap > st.btGap && st.bqGap > 0 ? st.bqGap : 0);
  return st;
}

// chunks per segment: CB blocks each (0 for the AffineOneGapAlign segments)
__global__ void ir_band_nchunk(BandArgs B) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= B.n_seg) return;
  B.s_nchunk[s] = (B.s_kind[s] != 0) ? 0u : (uint32_t)((B.s_b1[s] - B.s_b0[s] + 1 + CB - 1) / CB);
}
__global__ void ir_band_tasks(BandArgs B) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= B.n_seg) return;
  const uint64_t lo = B.chunk_off[s], hi = B.chunk_off[s + 1];
  for (uint64_t x = lo; x < hi; x++) B.task_seg[x] = (uint32_t)s;
  if (hi - lo <= 1) B.s_status[s] = 0;
}

// One WAVE per multi-chunk segment: the row / query position at which every chunk of CB blocks
// starts (a prefix sum over the block steps), and whether the replay would run off the segment
// (tOff reaches tLen early, or ends short of it, :253,:308 -> status 1): every step consumes
// rows >= 0, so that happens exactly when the steps do not add up to tLen.
__global__ void __launch_bounds__(64) ir_band_prep(BandArgs B) {
  const uint64_t s = blockIdx.x;
  const int lane = threadIdx.x;
  const uint64_t co = B.chunk_off[s];
  if (B.chunk_off[s + 1] - co <= 1) return;
  const SegBlocks sb = seg_blocks(B, s);
  const long nb = sb.b1 - sb.b0 + 1;
  long long carryT = 0, carryQ = 0;
  for (long base = 0; base < nb; base += 64) {
    const long i = base + lane;
    long long r = 0, q = 0;
    if (i < nb) {
      long bq, bt, bl, nq2, nt2, nl2;
      sb.get(sb.b0 + i, bq, bt, bl);
      const BlockStep st = block_step(sb, sb.b0 + i, bq, bt, bl, nq2, nt2, nl2);
      r = st.rows; q = st.advq;
    }
    long long ir = r, iq = q;
    for (int d = 1; d < 64; d <<= 1) {
      long long orr = __shfl_up(ir, d), oq = __shfl_up(iq, d);
      if (lane >= d) { ir += orr; iq += oq; }
    }
    if (i < nb && i > 0 && (i % CB) == 0) {
      B.chunkT[co + i / CB] = (int32_t)(carryT + ir - r);
      B.chunkQ[co + i / CB] = (int32_t)(sb.fq + carryQ + iq - q);
    }
    carryT += __shfl(ir, 63); carryQ += __shfl(iq, 63);
  }
  if (lane == 0) B.s_status[s] = (carryT != (long long)B.s_rows[s]) ? 1 : 0;
}

// One LANE per chunk of CB blocks: the row-window construction of IndelRefine.h:220-333 replayed
// event by event.  Only the 2k rows around the current target row can still change; they live in a
// per-lane ring in LDS (lane-interleaved, conflict-free), finished rows stream out to HBM.  An
// event at row t touches rows t-k+1 .. t+k-1 only and rows do not interact, so a chunk that starts
// at row T0 reproduces the state of rows >= T0-k+1 exactly by replaying, on an empty ring, from
// any block that starts at or before row T0-2k+1; it walks the previous chunk's blocks
// arithmetically up to that block, and owns (writes) the rows it retires from T0-k+1 on.
template <int RS>
__global__ void __launch_bounds__(64) ir_band_chunk(BandArgs B) {
  __shared__ int ringS[RS * 64], ringE[RS * 64];
  const int lane = threadIdx.x;
  const uint64_t x = (uint64_t)blockIdx.x * 64 + lane;
  if (x >= B.n_task) return;
  const uint64_t s = B.task_seg[x];
  const uint64_t co = B.chunk_off[s];
  const long c = (long)(x - co), nch = (long)(B.chunk_off[s + 1] - co);
  const bool multi = nch > 1, lastChunk = (c == nch - 1);
  if (multi && B.s_status[s] != 0) return;
  const long tLen = (long)B.s_rows[s];
  const SegBlocks sb = seg_blocks(B, s);
  const long qStart = B.s_qStart[s], qEnd = B.s_qEnd[s];
  const int k = B.k;
  Row* rows = B.rows + B.s_row_off[s];
  int status = 0;
#define RG(arr, r) arr[(int)((r) & (RS - 1)) * 64 + lane]
  for (int z = 0; z < RS; z++) { ringS[z * 64 + lane] = -1; ringE[z * 64 + lane] = -1; }
  const long bFirst = sb.b0 + c * CB, bLast = min(sb.b1, bFirst + CB - 1);
  const long T0 = c > 0 ? B.chunkT[co + c] : 0;
  long from = c;
  while (from > 0) { from--; if (from == 0 || T0 - B.chunkT[co + from] >= 2 * (long)k - 1) break; }
  const long ownLo = c > 0 ? T0 - k + 1 : 0, thr = T0 - 2 * (long)k + 1;
  long q = from > 0 ? B.chunkQ[co + from] : sb.fq;
  long tOff = from > 0 ? B.chunkT[co + from] : 0;
  auto flush = [&](long r) {   // row r can no longer change: write it out and recycle its slot
    if (r >= 0 && r < tLen) {
      if (r >= ownLo) { int2 v; v.x = RG(ringS, r); v.y = RG(ringE, r); *(int2*)&rows[r] = v; }
      RG(ringS, r) = -1; RG(ringE, r) = -1;
    }
  };
  long bq, bt, bl;
  sb.get(sb.b0 + from * CB, bq, bt, bl);
  bool live = (c == 0);
  int eMin = -1;                                                        // lower bound of qE over the k rows behind tOff (-1: not known, the literal loop runs)
  long sHi = -1;                                                        // the last row whose qS has been set
  for (long b = sb.b0 + from * CB; b <= bLast && !(status & 1); b++) {  // :232-315
    long nq2, nt2, nl2;
    const BlockStep st = block_step(sb, b, bq, bt, bl, nq2, nt2, nl2);
    if (!live) {
      if (b < bFirst && tOff + st.rows <= thr) { tOff += st.rows; q += st.advq; bq = nq2; bt = nt2; bl = nl2; continue; }
      live = true;
    }
    for (long bi = 0; bi < st.blockLength; bi++) {                      // :252-283
      if (tOff >= tLen) { status |= 1; break; }
      int newE;
      {
        int lo = (int)max(q - k, qStart);
        int cs = RG(ringS, tOff), ce = RG(ringE, tOff);
        RG(ringS, tOff) = (cs == -1) ? lo : min(cs, lo);
        newE = ce;
        if (ce == -1 || ce < q + k) { newE = (int)min(qEnd - 1, q + k); RG(ringE, tOff) = newE; }
      }
      // The reference's loop over ki = 0 .. k - 1 (:262-281) raises qE of the rows tOff - ki to q and sets qS of the rows tOff + ki to q where it is unset or larger.
      // q never decreases (it is a running counter), so: (E) nothing changes while q <= the smallest qE of the k rows behind -- eMin is a lower bound of that
      // minimum (a row that enters the window can only lower it, a row that leaves it is ignored), and the literal loop runs, and makes it exact, only when q gets
      // past it: once per ~k bases inside a block instead of k reads per base; (S) a row whose qS has been set holds a value <= q for good (every value written is
      // the q or q - k of its time), so only the rows beyond sHi, the last row set, are still unset (-1), and they take q without being read.
      if (newE < eMin) eMin = newE;
      if (q > eMin) {
        int m = 0x7fffffff;
        for (int ki = 0; ki < k; ki++)
          if (tOff - ki >= 0) { int e = RG(ringE, tOff - ki); if (e < q) { e = (int)q; RG(ringE, tOff - ki) = e; } m = min(m, e); }
        eMin = m;
      }
      if (sHi < tOff) sHi = tOff;
      for (long r = sHi + 1; r <= tOff + k - 1 && r < tLen; r++) RG(ringS, r) = (int)q;
      if (sHi < tOff + k - 1) sHi = tOff + k - 1;
      tOff++; q++;
      flush(tOff - k);
    }
    if (st.bqGap > st.btGap) {                                          // :287-305
      for (int qi = 0; qi < st.bqGap; qi++, q++)
        for (int ki = 0; ki < k; ki++) {
          if (tOff - ki >= 0 && tOff - ki < tLen) { if (RG(ringE, tOff - ki) < q) RG(ringE, tOff - ki) = (int)q; }
          if (tOff + ki < tLen) { int v = RG(ringS, tOff + ki); if (v == 0 || v > q) RG(ringS, tOff + ki) = (int)q; }   // (sic) == 0
        }
    }
    if (st.btGap > st.bqGap) {                                          // :306-314
      for (int ti = 0; ti < st.btGap; ti++) {
        if (tOff >= tLen) { status |= 1; break; }
        const int e = (int)min(qEnd - 1, q + k);
        RG(ringS, tOff) = (int)max(q - k, qStart); RG(ringE, tOff) = e;
        if (e < eMin) eMin = e;
        if (sHi < tOff) sHi = tOff;
        tOff++;
        flush(tOff - k);
      }
    }
    bq = nq2; bt = nt2; bl = nl2;
  }
  if (lastChunk) for (long r = max(0L, tOff - k + 1); r < tLen; r++) flush(r);
#undef RG
  if (!multi) { if (tOff != tLen) status |= 1; B.s_status[s] = status; }
}

// One WAVE per segment: the suffix minimum of qS (:318-322), then the prefix maximum of qE, the
// window checks and the cell offsets (:323-328, + the target base of the row), 64 rows a step.
__global__ void __launch_bounds__(64) ir_band_scan(BandArgs B) {
  const uint64_t s = blockIdx.x;
  const int lane = threadIdx.x;
  if (B.s_kind[s] != 0) { if (lane == 0) { B.s_cells[s] = 0; B.s_status[s] = 0; B.s_width[s] = 0; B.s_tmpcap[s] = 0; } return; }
  int status = B.s_status[s];
  const long tLen = (long)B.s_rows[s];
  Row* rows = B.rows + B.s_row_off[s];
  unsigned long long cells = 0;
  int width = 0;
  if (!status && tLen > 0) {
    int run = 0x7fffffff;
    for (long base = ((tLen - 1) / 64) * 64; base >= 0; base -= 64) {
      const long r = base + lane;
      const int own = (r < tLen) ? rows[r].S : 0x7fffffff;
      int v = own;
      for (int d = 1; d < 64; d <<= 1) { int o = __shfl_down(v, d); if (lane + d < 64) v = min(v, o); }
      v = min(v, run);
      if (r < tLen && v != own) rows[r].S = v;
      run = __shfl(v, 0);
    }
    const int a = B.s_aln[s];
    const unsigned char* tb = (const unsigned char*)B.tseq + B.t_off[a] + B.s_tStart[s];
    int runE = -0x7fffffff;
    for (long base = 0; base < tLen; base += 64) {
      const long r = base + lane;
      Row w; w.S = 0; w.E = -0x7fffffff; w.C = 0; w.T = 0;
      if (r < tLen) w = rows[r];
      int e = w.E;
      for (int d = 1; d < 64; d <<= 1) { int o = __shfl_up(e, d); if (lane >= d) e = max(e, o); }
      e = max(e, runE);
      runE = __shfl(e, 63);
      int len = (r < tLen) ? e - w.S + 1 : 0;
      if (r < tLen) {
        if (len < 1 || w.S < 0) status |= 1;
        else if (len > MAXW) status |= 4;
        width = max(width, len);
      }
      const unsigned long long l = (unsigned long long)max(len, 0);
      unsigned long long inc = l;
      for (int d = 1; d < 64; d <<= 1) { unsigned long long o = __shfl_up(inc, d); if (lane >= d) inc += o; }
      if (r < tLen) { w.E = e; w.C = (unsigned int)(cells + inc - l); w.T = tb[r]; rows[r] = w; }
      cells += __shfl(inc, 63);
    }
    for (int o = 32; o > 0; o >>= 1) { status |= __shfl_xor(status, o); width = max(width, __shfl_xor(width, o)); }
  }
  if (lane == 0) {
    B.s_cells[s] = status ? 0 : cells;
    B.s_status[s] = status;
    B.s_width[s] = status ? 0 : width;
    B.s_tmpcap[s] = status ? 0 : (uint64_t)(tLen + ((long)B.s_qEnd[s] - B.s_qStart[s]) + 2);
  }
}

// Work lists of ir_fill: width class (16 / 32 / 64 lanes per segment by the segment's widest row; class 3 = wider rows, ir_fill_wide) x length bucket (log2 of the row
// count, longest first), so that the lane groups of a wave sweep segments of similar length and the long ones start first.  Bin =
// class * 32 + (31 - log2 rows).  EMIT = false counts the bins, EMIT = true places the segments (cursor = bin start offsets).
constexpr int FILL_BINS = 128;
template <bool EMIT>
__global__ void ir_classify(uint64_t n_seg, const int32_t* __restrict__ s_kind, const int32_t* __restrict__ s_status, const int32_t* __restrict__ s_width,
                            const uint64_t* __restrict__ s_rows, int* bins, uint32_t* list) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  int bin = -1;
  if (s < n_seg && s_kind[s] == 0 && s_status[s] == 0) {
    const int w = s_width[s];
    const uint32_t r = (uint32_t)min((unsigned long long)s_rows[s], 0x7fffffffULL);
    bin = (w <= 16 ? 0 : w <= 32 ? 1 : w <= 64 ? 2 : 3) * 32 + (r ? __clz(r) : 31);
  }
  const unsigned long long below = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));
  unsigned long long todo = __ballot(bin >= 0);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int b = __shfl(bin, leader);
    const unsigned long long m = __ballot(bin == b);
    int base = 0;
    if (lane == leader) base = atomicAdd(&bins[b], __popcll(m));
    base = __shfl(base, leader);
    if (EMIT && bin == b) list[base + __popcll(m & below)] = (uint32_t)s;
    todo &= ~m;
  }
}

// ---------------------------------------------------------------------------------- fill
struct FillArgs {
  uint64_t n_seg;
  const int32_t* s_aln; const uint64_t* s_rows; const uint64_t* s_row_off; const uint64_t* s_cell_off;
  const Row* rows;
  const char* qseq; const uint64_t* q_off; const int32_t* q_len;
  int match, mismatch, g;
  unsigned char* path;
  const uint32_t* list; int* cursor;      // work list (class-major, long segments first); cursor[cls] = next entry, class cls ends at cursor[4 + cls]
};

__device__ __forceinline__ bool is_bound(long row, int c, int len) { return c == len - 1 || (row > 0 && c == 0); }

// Cross-lane moves on the VALU (DPP), no LDS round trip.  shr1: lane i takes lane i-1 of the wave
// (lane 0 takes `fill`).  scan_max<G>: inclusive prefix maximum inside aligned groups of G lanes
// (row_shr inside the 16-lane rows, then row_bcast 15 / 31 carry the row totals forward).
__device__ __forceinline__ int shr1(int x, int fill) { return __builtin_amdgcn_update_dpp(fill, x, 0x138, 0xf, 0xf, false); }
template <int G>
__device__ __forceinline__ int scan_max(int w) {
  w = max(w, __builtin_amdgcn_update_dpp(NEG, w, 0x111, 0xf, 0xf, false));
  w = max(w, __builtin_amdgcn_update_dpp(NEG, w, 0x112, 0xf, 0xf, false));
  w = max(w, __builtin_amdgcn_update_dpp(NEG, w, 0x114, 0xf, 0xf, false));
  w = max(w, __builtin_amdgcn_update_dpp(NEG, w, 0x118, 0xf, 0xf, false));
  if (G >= 32) w = max(w, __builtin_amdgcn_update_dpp(NEG, w, 0x142, 0xa, 0xf, false));
  if (G >= 64) w = max(w, __builtin_amdgcn_update_dpp(NEG, w, 0x143, 0xc, 0xf, false));
  return w;
}

// G lanes per segment (G = 16, 32 or 64 by the segment's widest row), 64/G segments per wave.
// Each group sweeps its own segment row by row: one lane per cell, previous row in registers.
// The query bases of the rows come from a sliding register window (2G bases + G prefetched), the
// row descriptors from a register chunk of G rows; the only per-row memory traffic is the row of
// arrows going out.
template <int G>
__global__ void __launch_bounds__(64) ir_fill(FillArgs F) {
  constexpr int GP = 64 / G;
  constexpr int CLS = (G == 16) ? 0 : (G == 32) ? 1 : 2;
  const int lane = threadIdx.x;
  const int c = lane % G, gbase = lane - c;
  const int g = F.g, go = 2 * F.g + 1;
  const long listEnd = F.cursor[4 + CLS];
  const uint32_t* list = F.list;
  // per-group state (identical on the G lanes of a group)
  long ti = -1, tLen = 0;
  const Row* rows = nullptr; const unsigned char* qb = nullptr; unsigned char* P = nullptr;
  long chunkBase = 0, qLast = 0;
  Row chunk; chunk.S = chunk.E = chunk.T = 0; chunk.C = 0;
  int W0 = 0, qlo = 0, qhi = 0, qnext = 0;
  int prevM = BAD, prevD = BAD, prevS = 0, prevLen = 0;
  bool done = false;
  auto ldq = [&](long idx) -> int { return qb[idx < qLast ? idx : qLast]; };   // past the read: its last base (the reference reads out of range there)
  while (true) {
    if (!done && ti < 0) {
      long x = 0;
      if (c == 0) x = atomicAdd(&F.cursor[CLS], 1);                       // the group's next segment: whichever is next in the list
      x = __shfl(x, gbase);
      if (x < listEnd) {
        const uint64_t s = list[x];
        const int a = F.s_aln[s];
        tLen = (long)F.s_rows[s];
        rows = F.rows + F.s_row_off[s];
        qb = (const unsigned char*)F.qseq + F.q_off[a];
        qLast = (long)F.q_len[a] - 1;
        P = F.path + F.s_cell_off[s];
        ti = 0; chunkBase = 0;
        if (c < tLen) chunk = rows[c];
        W0 = rows[0].S;
        qlo = ldq((long)W0 + c); qhi = ldq((long)W0 + G + c); qnext = ldq((long)W0 + 2 * G + c);
      } else done = true;
    }
    if (__ballot(!done) == 0ULL) break;
    if (!done && ti - chunkBase == G) { chunkBase = ti; if (ti + c < tLen) chunk = rows[ti + c]; }
    const int src = gbase + (int)((ti - chunkBase) & (G - 1));
    const int S = __shfl(chunk.S, src), E = __shfl(chunk.E, src), tch = __shfl(chunk.T, src);
    const unsigned int C = __shfl(chunk.C, src);
    while (!done && S - W0 >= G) { W0 += G; qlo = qhi; qhi = qnext; qnext = ldq((long)W0 + 2 * G + c); }
    const int j = S - W0 + c;                                           // 0 .. 2G-1
    const int q1 = __shfl(qlo, gbase + (j & (G - 1))), q2 = __shfl(qhi, gbase + (j & (G - 1)));
    const int qch = (j < G) ? q1 : q2;
    const int len = E - S + 1;
    const bool lastRow = (ti == tLen - 1);
    const int off = S - prevS;
    const bool interior = c >= 1 && (lastRow ? c <= len - 1 : c <= len - 2);
    const int srcA = c + off, srcD = srcA - 1;
    const bool aboveIn = srcA <= prevLen - 1;                            // qE[ti-1] >= q   (:491,:548,:567)
    const int aM = __shfl(prevM, gbase + (srcA & (G - 1))), aD = __shfl(prevD, gbase + (srcA & (G - 1)));
    const int dM = shr1(aM, BAD);                                        // prevM[srcD]: only read by interior cells (c >= 1)
    const bool okA = aboveIn && !is_bound(ti - 1, srcA, prevLen);
    const bool okD = aboveIn && srcD >= 0 && !is_bound(ti - 1, srcD, prevLen);
    const int dOpen = okA ? aM + go : BAD, dExt = okA ? aD : BAD;        // :491-502 (gapExtend = 0)
    const int Dv = max(dOpen, dExt);
    const int delOpen = (Dv == dOpen) ? 1 : 0;                           // :504-516
    const int mS = okD ? dM + (tch == qch ? F.match : F.mismatch) : BAD; // :548-563
    const int dS = okA ? aM + g : BAD;                                   // :567-574
    const int V = interior ? max(mS, max(dS, Dv)) : NEG;
    const int W = scan_max<G>(V);                                        // inclusive prefix max of V inside the group
    int Wm1 = shr1(W, NEG), Vm1 = shr1(V, NEG);
    if (c == 0) { Wm1 = NEG; Vm1 = NEG; }
    const int Iv = max(BAD, go + Wm1);
    int M = max(max(BAD, V), max(Vm1 + g, go + Wm1));
    if (!interior) M = BAD;
    int Mleft = shr1(M, BAD);
    if (c <= 1) Mleft = BAD;                                             // the row's left boundary cell (:413-418)
    const int iOpen = Mleft + go;                                        // :523
    const int insOpen = (Iv == iOpen) ? 1 : 0;                           // :528-540
    const int iS = Mleft + g;                                            // :565
    int code;
    if (!interior) code = C_BOUND;
    else if (M == mS) code = C_DIAG;                                     // :583-616
    else if (M == iS) code = C_LEFT;
    else if (M == dS) code = C_DOWN;
    else if (M == Dv) code = C_DELCLOSE;
    else code = C_INSCLOSE;
    int outM = M, outD = interior ? Dv : BAD;
    unsigned char outB = (unsigned char)(code | (delOpen << 3) | (insOpen << 4));
    if (ti == 0) {                                                       // :407-431 first row
      const bool last0 = (c == len - 1) && (tLen > 1);
      outM = last0 ? BAD : (c == 0 ? 0 : c * g);
      outD = BAD;
      outB = (unsigned char)(last0 ? C_BOUND : (c == 0 ? C_DONE : C_LEFT));
    }
    if (!done) {
      if (c < len) P[C + c] = outB;
      prevM = outM; prevD = outD; prevS = S; prevLen = len;
      ti++;
      if (ti == tLen) ti = -1;
    }
  }
}


// Rows of more than 64 cells (refineBand 50: -CONTIG): one wave per segment, a row is swept in pieces of 64 cells; the previous row's M and D
// live in LDS (double buffered), the prefix maximum, V and M of a piece's last cell are carried into the next piece.  Same recurrence, same
// arrows as ir_fill.
__global__ void __launch_bounds__(64) ir_fill_wide(FillArgs F) {
  __shared__ int sM[2][MAXW + 64], sD[2][MAXW + 64];
  const int lane = threadIdx.x;
  const int g = F.g, go = 2 * F.g + 1;
  const long listEnd = F.cursor[4 + 3];
  while (true) {
    long x = 0;
    if (lane == 0) x = atomicAdd(&F.cursor[3], 1);
    // (lane 0's value as a SCALAR: through a shuffle it is a vector value to the compiler, and with it the segment, its rows, every loop bound below)
    x = (long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)x >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(x & 0xffffffffL)));
    if (x >= listEnd) break;
    const uint64_t s = F.list[x];
    const int a = F.s_aln[s];
    const long tLen = (long)F.s_rows[s];
    const Row* rows = F.rows + F.s_row_off[s];
    const unsigned char* qb = (const unsigned char*)F.qseq + F.q_off[a];
    const long qLast = (long)F.q_len[a] - 1;
    unsigned char* P = F.path + F.s_cell_off[s];
    int prevS = 0, prevLen = 0;
    for (long ti = 0; ti < tLen; ti++) {
      const Row rw = rows[ti];
      const int S = rw.S, len = rw.E - rw.S + 1, tch = rw.T;
      const unsigned int C = rw.C;
      const bool lastRow = (ti == tLen - 1);
      const int off = S - prevS;
      const int cur = (int)(ti & 1), prv = cur ^ 1;
      int carryW = NEG, carryV = NEG, carryM = BAD;
      for (int base = 0; base < len; base += 64) {
        const int c = base + lane;
        int outM, outD; unsigned char outB;
        if (ti == 0) {                                                     // :407-431 first row
          const bool last0 = (c == len - 1) && (tLen > 1);
          outM = last0 ? BAD : (c == 0 ? 0 : c * g);
          outD = BAD;
          outB = (unsigned char)(last0 ? C_BOUND : (c == 0 ? C_DONE : C_LEFT));
        } else {
          const long qi = (long)S + c;
          const int qch = qb[qi < qLast ? qi : qLast];
          const bool interior = c >= 1 && (lastRow ? c <= len - 1 : c <= len - 2);
          const int srcA = c + off, srcD = srcA - 1;
          const bool aboveIn = srcA <= prevLen - 1;
          const bool inA = srcA >= 0 && srcA < prevLen, inD = srcD >= 0 && srcD < prevLen;
          const int aM = inA ? sM[prv][srcA] : BAD, aD = inA ? sD[prv][srcA] : BAD;
          const int dM = inD ? sM[prv][srcD] : BAD;
          const bool okA = aboveIn && !is_bound(ti - 1, srcA, prevLen);
          const bool okD = aboveIn && srcD >= 0 && !is_bound(ti - 1, srcD, prevLen);
          const int dOpen = okA ? aM + go : BAD, dExt = okA ? aD : BAD;
          const int Dv = max(dOpen, dExt);
          const int delOpen = (Dv == dOpen) ? 1 : 0;
          const int mS = okD ? dM + (tch == qch ? F.match : F.mismatch) : BAD;
          const int dS = okA ? aM + g : BAD;
          const int V = interior ? max(mS, max(dS, Dv)) : NEG;
          const int W = max(scan_max<64>(V), carryW);
          int Wm1 = shr1(W, NEG), Vm1 = shr1(V, NEG);
          if (lane == 0) { Wm1 = carryW; Vm1 = carryV; }
          if (c == 0) { Wm1 = NEG; Vm1 = NEG; }
          const int Iv = max(BAD, go + Wm1);
          int M = max(max(BAD, V), max(Vm1 + g, go + Wm1));
          if (!interior) M = BAD;
          int Mleft = shr1(M, BAD);
          if (lane == 0) Mleft = carryM;
          if (c <= 1) Mleft = BAD;
          const int iOpen = Mleft + go;
          const int insOpen = (Iv == iOpen) ? 1 : 0;
          const int iS = Mleft + g;
          int code;
          if (!interior) code = C_BOUND;
          else if (M == mS) code = C_DIAG;
          else if (M == iS) code = C_LEFT;
          else if (M == dS) code = C_DOWN;
          else if (M == Dv) code = C_DELCLOSE;
          else code = C_INSCLOSE;
          outM = M; outD = interior ? Dv : BAD;
          outB = (unsigned char)(code | (delOpen << 3) | (insOpen << 4));
          carryW = __shfl(W, 63); carryV = __shfl(V, 63); carryM = __shfl(M, 63);
        }
        if (c < len) { P[C + c] = outB; sM[cur][c] = outM; sD[cur][c] = outD; }
      }
      wave_sync_lds();                                                       // (the rows meet in sM / sD only; the arrows are written and never read here)
      prevS = S; prevLen = len;
    }
  }
}

// ---------------------------------------------------------------------------------- trace
// Segments whose widest row has 17 .. 32 cells, on 16 lanes (four segments per wave instead of two): a read's rows are 15 cells wide (2 refineBand + 1) except around
// its indels, so with 32 lanes per segment half of them idle on almost every row.  A row of more than 16 cells is done in two pieces of 16, the second one taking over
// the first's running prefix maximum and its last cell's V / M (what the shifts by one cell read); the previous row's M and D are two registers per lane.
__global__ void __launch_bounds__(64) ir_fill_16x2(FillArgs F) {
  constexpr int G = 16;
  const int lane = threadIdx.x;
  const int c = lane % G, gbase = lane - c;
  const int g = F.g, go = 2 * F.g + 1;
  const long listEnd = F.cursor[4 + 1];
  const uint32_t* list = F.list;
  long ti = -1, tLen = 0;
  const Row* rows = nullptr; const unsigned char* qb = nullptr; unsigned char* P = nullptr;
  long chunkBase = 0, qLast = 0;
  Row chunk; chunk.S = chunk.E = chunk.T = 0; chunk.C = 0;
  int W0 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;                               // query bases W0 + c, + 16, + 32, + 48
  int pM[2] = {BAD, BAD}, pD[2] = {BAD, BAD}, prevS = 0, prevLen = 0;
  bool done = false;
  auto ldq = [&](long idx) -> int { return qb[idx < qLast ? idx : qLast]; };
  while (true) {
    if (!done && ti < 0) {
      long x = 0;
      if (c == 0) x = atomicAdd(&F.cursor[1], 1);
      x = __shfl(x, gbase);
      if (x < listEnd) {
        const uint64_t s = list[x];
        const int a = F.s_aln[s];
        tLen = (long)F.s_rows[s];
        rows = F.rows + F.s_row_off[s];
        qb = (const unsigned char*)F.qseq + F.q_off[a];
        qLast = (long)F.q_len[a] - 1;
        P = F.path + F.s_cell_off[s];
        ti = 0; chunkBase = 0;
        if (c < tLen) chunk = rows[c];
        W0 = rows[0].S;
        q0 = ldq((long)W0 + c); q1 = ldq((long)W0 + G + c); q2 = ldq((long)W0 + 2 * G + c); q3 = ldq((long)W0 + 3 * G + c);
      } else done = true;
    }
    if (__ballot(!done) == 0ULL) break;
    if (!done && ti - chunkBase == G) { chunkBase = ti; if (ti + c < tLen) chunk = rows[ti + c]; }
    const int src = gbase + (int)((ti - chunkBase) & (G - 1));
    const int S = __shfl(chunk.S, src), E = __shfl(chunk.E, src), tch = __shfl(chunk.T, src);
    const unsigned int C = __shfl(chunk.C, src);
    while (!done && S - W0 >= G) { W0 += G; q0 = q1; q1 = q2; q2 = q3; q3 = ldq((long)W0 + 3 * G + c); }
    const int len = E - S + 1;
    const bool lastRow = (ti == tLen - 1);
    const int off = S - prevS;
    int nM[2] = {BAD, BAD}, nD[2] = {BAD, BAD};
    int carryW = NEG, carryV = NEG, carryM = BAD;
    const int np = (!done && len > G) ? 2 : 1;
    for (int p = 0; p < np; p++) {
      const int cc = c + G * p;
      // the query base of cell cc: position S + cc = W0 + j, j in [0, 4 G)
      const int j = S - W0 + cc;
      const int l = gbase + (j & (G - 1));
      const int qa = __shfl(q0, l), qbb = __shfl(q1, l), qc = __shfl(q2, l), qd = __shfl(q3, l);
      const int qch = (j < G) ? qa : (j < 2 * G) ? qbb : (j < 3 * G) ? qc : qd;
      const bool interior = cc >= 1 && (lastRow ? cc <= len - 1 : cc <= len - 2);
      const int srcA = cc + off, srcD = srcA - 1;
      const bool aboveIn = srcA <= prevLen - 1;                            // qE[ti-1] >= q   (:491,:548,:567)
      // the previous row's cells srcA and srcA - 1 (pieces 0 / 1 of it)
      const int la = gbase + (srcA & (G - 1)), ld = gbase + (srcD & (G - 1));
      const int a0 = __shfl(pM[0], la), a1 = __shfl(pM[1], la), b0 = __shfl(pD[0], la), b1 = __shfl(pD[1], la);
      const int d0 = __shfl(pM[0], ld), d1 = __shfl(pM[1], ld);
      const int aM = (srcA & G) ? a1 : a0, aD = (srcA & G) ? b1 : b0;
      const int dM = (srcD & G) ? d1 : d0;
      const bool okA = aboveIn && !is_bound(ti - 1, srcA, prevLen);
      const bool okD = aboveIn && srcD >= 0 && !is_bound(ti - 1, srcD, prevLen);
      const int dOpen = okA ? aM + go : BAD, dExt = okA ? aD : BAD;        // :491-502 (gapExtend = 0)
      const int Dv = max(dOpen, dExt);
      const int delOpen = (Dv == dOpen) ? 1 : 0;                           // :504-516
      const int mS = okD ? dM + (tch == qch ? F.match : F.mismatch) : BAD; // :548-563
      const int dS = okA ? aM + g : BAD;                                   // :567-574
      const int V = interior ? max(mS, max(dS, Dv)) : NEG;
      const int W = max(scan_max<G>(V), carryW);                           // inclusive prefix max of V over the row so far
      int Wm1 = shr1(W, NEG), Vm1 = shr1(V, NEG);
      if (c == 0) { Wm1 = carryW; Vm1 = carryV; }                          // (piece 0: NEG, NEG)
      const int Iv = max(BAD, go + Wm1);
      int M = max(max(BAD, V), max(Vm1 + g, go + Wm1));
      if (!interior) M = BAD;
      int Mleft = shr1(M, BAD);
      if (c == 0) Mleft = carryM;
      if (cc <= 1) Mleft = BAD;                                            // the row's left boundary cell (:413-418)
      const int iOpen = Mleft + go;                                        // :523
      const int insOpen = (Iv == iOpen) ? 1 : 0;                           // :528-540
      const int iS = Mleft + g;                                            // :565
      int code;
      if (!interior) code = C_BOUND;
      else if (M == mS) code = C_DIAG;                                     // :583-616
      else if (M == iS) code = C_LEFT;
      else if (M == dS) code = C_DOWN;
      else if (M == Dv) code = C_DELCLOSE;
      else code = C_INSCLOSE;
      int outM = M, outD = interior ? Dv : BAD;
      unsigned char outB = (unsigned char)(code | (delOpen << 3) | (insOpen << 4));
      if (ti == 0) {                                                       // :407-431 first row
        const bool last0 = (cc == len - 1) && (tLen > 1);
        outM = last0 ? BAD : (cc == 0 ? 0 : cc * g);
        outD = BAD;
        outB = (unsigned char)(last0 ? C_BOUND : (cc == 0 ? C_DONE : C_LEFT));
      }
      if (!done && cc < len) P[C + cc] = outB;
      nM[p] = outM; nD[p] = outD;
      carryW = __shfl(W, gbase + G - 1); carryV = __shfl(V, gbase + G - 1); carryM = __shfl(M, gbase + G - 1);
    }
    if (!done) {
      pM[0] = nM[0]; pM[1] = nM[1]; pD[0] = nD[0]; pD[1] = nD[1]; prevS = S; prevLen = len;
      ti++;
      if (ti == tLen) ti = -1;
    }
  }
}

// ---------------------------------------------------------------------------------- fill, anti-diagonal form
// The same recurrence swept along anti-diagonals instead of rows: a LANE per row, eight rows of a segment in flight on the eight lanes of a group (lane of row i =
// i mod 8), every lane two cells of its row per step.  A cell needs three cells of the row above and its own left neighbour: the left neighbour is the lane's own last
// cell -- the row's chain M[q-1], I[q-1] is walked serially, literally, no prefix-maximum scan --, and the row above is one lane to the left, two cells per step like
// this one.  Row i starts  1 + ceil(off / 2)  steps behind row i - 1 (off = the shift of its window, qS[i] - qS[i-1] >= 0): then the three cells above are exactly
// among the four cells the lane to the left made in the last two steps (which two it takes depends on the parity of off), and they come over with seven DPP moves
// (row_ror:2 -- the two groups of a 16-lane DPP row are its even and its odd lanes, so "one lane to the left, cyclically inside the group" is one rotation).
// Nothing but the arrows and the lanes' next row descriptors / query bases touches memory; a row step of the row-wise kernels costs ~190 wave instructions for 60
// cells (a prefix-maximum scan, thirteen cross-lane reads), a step here ~150 for ~120.
// A lane is free for row i + 8 when that row has to start if  ceil(len_i / 2) <= sum over the eight rows between of (1 + ceil(off / 2)), which is >= 8: always for
// rows of at most 16 cells; a segment whose wider rows break it is given up on the spot (nothing of it is used) and listed for ir_fill_16x2.
__device__ __forceinline__ int ror2(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x122, 0xf, 0xf, false); }   // lane l takes lane (l - 2) mod 16 of its DPP row

__global__ void __launch_bounds__(64) ir_fill_diag(FillArgs F, int cls, uint32_t* retry, int* retryCursor) {
  enum { ST_DONE = 0, ST_WAIT = 1, ST_RUN = 2 };
  const int lane = threadIdx.x;
  const int p = (lane & 15) >> 1;                                          // position in the group = row index mod 8
  const int leader = lane & 0x31;                                          // the group's lane with p == 0
  const unsigned long long gmask = (0x5555ULL << (lane & 1)) << (lane & 48);
  const int g = F.g, go = 2 * F.g + 1, match = F.match, mismatch = F.mismatch;
  const long listEnd = F.cursor[4 + cls];
  // group state (identical on the eight lanes of a group)
  bool segOn = false, listDone = false;
  int tLen = 0, t = 0;
  uint32_t segId = 0;
  const Row* rows = nullptr; const unsigned char* qb = nullptr; unsigned char* P = nullptr;
  long qLast = 0;
  // lane state
  int st = ST_DONE, row = 0, c = 0;
  int pubRow = -1, pubT0 = 0;                                              // the row this lane started last, and when
  Row cur; cur.S = cur.E = cur.T = 0; cur.C = 0;
  int upS = 0, upLen = 0;                                                  // window start / length of the row above `cur`
  Row nx = cur; int nxUpS = 0, nxUpLen = 0;                                // the same for the lane's row after this one (asked for when a row starts)
  int Wrun = NEG, Vprev = NEG, Mprev = BAD;
  int h1M0 = BAD, h1D0 = BAD, h1M1 = BAD, h1D1 = BAD, h2M0 = BAD, h2M1 = BAD, h2D1 = BAD;   // this lane's cells of the last step (both) and of the step before
  int qn0 = 0, qn1 = 0;                                                    // the query bases of the lane's next two cells
  while (true) {
    // ---- a group without a segment takes the next one of the list
    const bool need = !segOn && !listDone;
    if (__ballot(need)) {
      if (need) {
        long x = 0;
        if (p == 0) x = atomicAdd(&F.cursor[cls], 1);
        x = __shfl(x, leader);
        if (x < listEnd) {
          segId = F.list[x];
          const int a = F.s_aln[segId];
          tLen = (int)F.s_rows[segId];
          rows = F.rows + F.s_row_off[segId];
          qb = (const unsigned char*)F.qseq + F.q_off[a];
          qLast = (long)F.q_len[a] - 1;
          P = F.path + F.s_cell_off[segId];
          segOn = true; t = 0; pubRow = -1; pubT0 = 0;
          row = p;
          if (row < tLen) {
            st = ST_WAIT;
            cur = rows[row];
            if (row > 0) { const Row u = rows[row - 1]; upS = u.S; upLen = u.E - u.S + 1; } else { upS = cur.S; upLen = 0; }
            c = 0;
            const long q0i = (long)cur.S, q1i = q0i + 1;
            qn0 = qb[q0i < qLast ? q0i : qLast]; qn1 = qb[q1i < qLast ? q1i : qLast];
          } else st = ST_DONE;
        } else listDone = true;
      }
    }
    if (__ballot(segOn) == 0ULL) break;
    // ---- what the lane to the left has: its row, that row's first step, its cells of the last two steps
    const int nbRow = ror2(pubRow), nbT0 = ror2(pubT0);
    const int n1M0 = ror2(h1M0), n1D0 = ror2(h1D0), n1M1 = ror2(h1M1), n1D1 = ror2(h1D1), n2M0 = ror2(h2M0), n2M1 = ror2(h2M1), n2D1 = ror2(h2D1);
    const int off = cur.S - upS;
    bool conflict = false;
    if (st == ST_WAIT) {
      bool go_ = false;
      if (row == 0) go_ = true;
      else if (nbRow == row - 1) {
        const int treq = nbT0 + 1 + ((off + 1) >> 1);
        go_ = t == treq; conflict = t > treq;
      }
      if (go_) {
        st = ST_RUN; c = 0; Wrun = NEG; Vprev = NEG; Mprev = BAD; pubRow = row; pubT0 = t;
        if (row + 8 < tLen) { nx = rows[row + 8]; const Row u = rows[row + 7]; nxUpS = u.S; nxUpLen = u.E - u.S + 1; }
      }
    }
    int o0M = BAD, o0D = BAD, o1M = BAD, o1D = BAD;
    if (st == ST_RUN) {
      const int len = cur.E - cur.S + 1, tch = cur.T;
      const bool lastRow = row == tLen - 1, upNotFirst = row >= 2;
      const bool odd = off & 1;
      const int aM0 = odd ? n2M1 : n1M0, aD0 = odd ? n2D1 : n1D0, dM0 = odd ? n2M0 : n2M1;
      const int aM1 = odd ? n1M0 : n1M1, aD1 = odd ? n1D0 : n1D1, dM1 = aM0;
#define IR_CELL(cc_, aM_, aD_, dM_, qch_, oM_, oD_) do {                                                                                     \
        const int cc = (cc_);                                                                                                                 \
        const bool interior = cc >= 1 && (lastRow ? cc <= len - 1 : cc <= len - 2);                                                           \
        const int srcA = cc + off, srcD = srcA - 1;                                                                                           \
        const bool aboveIn = srcA <= upLen - 1;                              /* qE[ti-1] >= q   (:491,:548,:567) */                            \
        const bool okA = aboveIn && !(srcA == upLen - 1 || (upNotFirst && srcA == 0));                                                         \
        const bool okD = aboveIn && srcD >= 0 && !(srcD == upLen - 1 || (upNotFirst && srcD == 0));                                            \
        const int dOpen = okA ? (aM_) + go : BAD, dExt = okA ? (aD_) : BAD;  /* :491-502 (gapExtend = 0) */                                    \
        const int Dv = max(dOpen, dExt);                                                                                                      \
        const int delOpen = (Dv == dOpen) ? 1 : 0;                           /* :504-516 */                                                    \
        const int mS = okD ? (dM_) + (tch == (qch_) ? match : mismatch) : BAD;   /* :548-563 */                                                \
        const int dS = okA ? (aM_) + g : BAD;                                /* :567-574 */                                                    \
        const int V = interior ? max(mS, max(dS, Dv)) : NEG;                                                                                  \
        const int Wm1 = Wrun, Vm1 = Vprev;                                   /* the row's prefix maximum of V and V itself, one cell to the left (cell 0: NEG) */ \
        const int Iv = max(BAD, go + Wm1);                                                                                                    \
        int M = max(max(BAD, V), max(Vm1 + g, go + Wm1));                                                                                     \
        if (!interior) M = BAD;                                                                                                               \
        int Mleft = Mprev;                                                                                                                    \
        if (cc <= 1) Mleft = BAD;                                            /* the row's left boundary cell (:413-418) */                     \
        const int iOpen = Mleft + go;                                        /* :523 */                                                        \
        const int insOpen = (Iv == iOpen) ? 1 : 0;                           /* :528-540 */                                                    \
        const int iS = Mleft + g;                                            /* :565 */                                                        \
        int code;                                                                                                                             \
        if (!interior) code = C_BOUND;                                                                                                        \
        else if (M == mS) code = C_DIAG;                                     /* :583-616 */                                                    \
        else if (M == iS) code = C_LEFT;                                                                                                      \
        else if (M == dS) code = C_DOWN;                                                                                                      \
        else if (M == Dv) code = C_DELCLOSE;                                                                                                  \
        else code = C_INSCLOSE;                                                                                                               \
        int outM = M, outD = interior ? Dv : BAD;                                                                                             \
        unsigned char outB = (unsigned char)(code | (delOpen << 3) | (insOpen << 4));                                                         \
        if (row == 0) {                                                      /* :407-431 first row */                                          \
          const bool last0 = (cc == len - 1) && (tLen > 1);                                                                                   \
          outM = last0 ? BAD : (cc == 0 ? 0 : cc * g);                                                                                        \
          outD = BAD;                                                                                                                         \
          outB = (unsigned char)(last0 ? C_BOUND : (cc == 0 ? C_DONE : C_LEFT));                                                              \
        }                                                                                                                                     \
        P[cur.C + cc] = outB;                                                                                                                 \
        Wrun = max(Wrun, V); Vprev = V; Mprev = M;                                                                                            \
        (oM_) = outM; (oD_) = outD;                                                                                                           \
      } while (0)
      IR_CELL(c, aM0, aD0, dM0, qn0, o0M, o0D);
      if (c + 1 < len) IR_CELL(c + 1, aM1, aD1, dM1, qn1, o1M, o1D);
#undef IR_CELL
      c += 2;
      if (c >= len) {                                                      // the row is through: the lane's next row is eight further on
        row += 8;
        if (row < tLen) { st = ST_WAIT; cur = nx; upS = nxUpS; upLen = nxUpLen; c = 0; }
        else st = ST_DONE;
      }
    }
    h2M0 = h1M0; h2M1 = h1M1; h2D1 = h1D1;
    h1M0 = o0M; h1D0 = o0D; h1M1 = o1M; h1D1 = o1D;
    // the bases of the lane's next two cells (a waiting lane: its row's first two)
    if (st != ST_DONE) {
      const long q0i = (long)cur.S + c, q1i = q0i + 1;
      qn0 = qb[q0i < qLast ? q0i : qLast]; qn1 = qb[q1i < qLast ? q1i : qLast];   // past the read: its last base (the reference reads out of range there)
    }
    t++;
    // ---- the group's segment is through (or given up)
    const unsigned long long cf = __ballot(conflict), live = __ballot(segOn && st != ST_DONE);
    if (segOn) {
      if (cf & gmask) {
        if (p == 0) retry[atomicAdd(&retryCursor[5], 1)] = segId;
        segOn = false; st = ST_DONE;
      } else if (!(live & gmask)) segOn = false;
    }
  }
}

// LRA_IR_DIAG_CHECK: the arrows of two fills of the same segments, byte for byte (a block per list entry)
__global__ void ir_fill_compare(const uint32_t* __restrict__ list, long first, long n, const uint64_t* __restrict__ s_cell_off, const uint64_t* __restrict__ s_cells,
                                const unsigned char* __restrict__ a, const unsigned char* __restrict__ b, unsigned long long* out) {
  const long e = first + blockIdx.x;
  if (e >= first + n) return;
  const uint32_t s = list[e];
  const uint64_t o = s_cell_off[s], m = s_cells[s];
  unsigned long long bad = 0, firstBad = ~0ULL;
  for (uint64_t i = threadIdx.x; i < m; i += blockDim.x) if (a[o + i] != b[o + i]) { bad++; if (i < firstBad) firstBad = i; }
  if (bad) { atomicAdd(&out[0], bad); atomicAdd(&out[1], 1ULL); atomicMin(&out[2], ((unsigned long long)s << 32) | (firstBad & 0xffffffffULL)); }
}

struct TraceArgs {
  uint64_t n_seg;
  const int32_t* s_kind; const int32_t* s_tStart; const uint64_t* s_rows; const uint64_t* s_row_off;
  const uint64_t* s_cells; const uint64_t* s_cell_off; int32_t* s_status;
  const Row* rows;
  const unsigned char* path;
  uint32_t* s_nblk;
  const uint64_t* s_tmp_off; int32_t* tmp_blocks;     // blocks in walk order (back to front), forward coordinates
};

// One LANE per segment: the trace back of IndelRefine.h:626-674 and the path -> blocks parse of
// :718-745 fused.  Walking back from the last cell, the forward position before the ops consumed so
// far is known from the cell coordinates, so blocks come out (in reverse order) with their final
// coordinates in one pass.  The forward parse is: [diag run][one run of left OR of down] -> one
// block; seen back to front every gap run closes the block whose (possibly empty) diag run precedes
// it, and a trailing diag run is a block of its own.
// The walk (:629-674) is a chain of dependent one-byte loads, a row apart each, ~27 k of them for a 30 kb read: one lane per segment walks at HBM latency.  Here a WAVE
// owns a segment: the rows' descriptors and path bytes of a window of rows (as many as fit 4 KB of path, at most 128: more waves per CU hide more of the walk's LDS latency than longer windows save in refills) are staged into LDS with coalesced loads,
// and while the walk is in the match state the 64 lanes look at the next 64 cells down the current diagonal at once -- a run of diagonal arrows (the common case: a
// 10 % error read has an indel every ~14 bases) is one round.  Everything else -- the gap states, left / down arrows, the end tests -- is the reference's step, from LDS.
constexpr int TW_ROWS = 128, TW_PATH = 4096;
__global__ void __launch_bounds__(64) ir_trace_wave(TraceArgs T) {
  __shared__ Row s_rows[TW_ROWS];
  __shared__ __attribute__((aligned(16))) unsigned char s_path[TW_PATH + 16];
  const int lane = threadIdx.x;
  const uint64_t s = blockIdx.x;
  if (s >= T.n_seg) return;
  if (T.s_kind[s] != 0) return;
  if (T.s_status[s] != 0) { if (lane == 0) T.s_nblk[s] = 0; return; }
  const long tLen = (long)T.s_rows[s];
  const Row* rows = T.rows + T.s_row_off[s];
  const unsigned char* P = T.path + T.s_cell_off[s];
  int32_t* ob = T.tmp_blocks + 3 * T.s_tmp_off[s];
  const long cap = (long)(T.s_tmp_off[s + 1] - T.s_tmp_off[s]);
  long ti = tLen - 1;
  long rBase = 0, wLo = 1, wHi = 0; unsigned pBase = 0;                   // the staged window: rows [wLo, wHi] (descriptors from rBase on), path bytes from pBase on
  auto stage = [&](long at) {
    wave_sync();
    wHi = at; rBase = max(0L, at - TW_ROWS + 1);
    for (long x = lane; x <= at - rBase; x += 64) s_rows[x] = rows[rBase + x];
    wave_sync();
    const Row last = s_rows[at - rBase];
    const unsigned end = last.C + (unsigned)max(last.E - last.S + 1, 0);
    // the first row of the window: the smallest w with end - rows[w].C <= TW_PATH (C grows with the row)
    long lo = rBase, hi = at;
    while (lo < hi) { const long mid = (lo + hi) >> 1; if (end - s_rows[mid - rBase].C <= (unsigned)TW_PATH) hi = mid; else lo = mid + 1; }
    wLo = lo; pBase = s_rows[lo - rBase].C;
    const unsigned nbytes = min(end - pBase, (unsigned)TW_PATH);
    // 16 bytes per lane and load: s_path[x] holds the byte at (16-byte aligned address at or below the window's first byte) + x; the bytes in front of the window
    // belong to earlier rows / segments of the same buffer, the last, partial 16 bytes go one by one
    const unsigned char* start = P + pBase;
    const unsigned delta = (unsigned)((uintptr_t)start & 15);
    const uint4* a16 = (const uint4*)(start - delta);
    const unsigned full = (delta + nbytes) >> 4;
    for (unsigned x = lane; x < full; x += 64) ((uint4*)s_path)[x] = a16[x];
    for (unsigned x = (full << 4) + lane; x < delta + nbytes; x += 64) s_path[x] = (start - delta)[x];
    pBase -= delta;                                                       // (modulo 2^32: only differences with a row's C are used)
    wave_sync();
  };
  stage(ti);
  Row rw = s_rows[ti - rBase];
  int qa = rw.E;                              // last cell of the matrix (:629)
  int mat = 0;                                // 0 match, 1 del, 2 ins
  int bad = 0;
  long nblk = 0;
  long q = (long)rw.E + 1, t = (long)T.s_tStart[s] + tLen;   // forward position after the whole path
  int curKind = -1, pending = 0;
  long dlen = 0;
  long steps = 0;
  const long step_cap = 4 * (long)T.s_cells[s] + 64;
  auto emit_block = [&]() {
    if (nblk < cap) { if (lane == 0) { ob[3 * nblk] = (int)q; ob[3 * nblk + 1] = (int)t; ob[3 * nblk + 2] = (int)dlen; } } else bad = 1;
    nblk++;
  };
  auto op = [&](int kind, long times) {       // kind: 0 diag, 1 left, 2 down
    if (kind != curKind) {
      if (curKind == -1 && kind == 0) { pending = 1; dlen = 0; }
      if (kind != 0) {
        if (pending) emit_block();
        pending = 1; dlen = 0;
      }
      curKind = kind;
    }
    if (kind == 0) { dlen += times; q -= times; t -= times; }
    else if (kind == 1) q -= times;
    else t -= times;
  };
  while (true) {
    if (ti < wLo || ti > wHi) stage(ti);
    rw = s_rows[ti - rBase];
    if (mat == 0) {
      // lane l: the cell l steps down the diagonal; a diagonal arrow there (not in row 0: the step after it would leave the matrix) extends the run
      const long tl = ti - lane;
      bool fast = false;
      if (tl >= wLo && tl >= 1) {
        const Row r = s_rows[tl - rBase];
        const int c = (qa - lane) - r.S;
        if (c >= 0 && c <= r.E - r.S) fast = (s_path[r.C - pBase + (unsigned)c] & 7) == C_DIAG;
      }
      const unsigned long long nf = ~__ballot(fast);
      const int f = nf ? __ffsll((long long)nf) - 1 : 64;
      if (f > 0) {
        steps += f;
        if (steps > step_cap) { bad = 1; break; }
        op(0, f); ti -= f; qa -= f;
        continue;
      }
    }
    const int c = qa - rw.S;
    if (c < 0 || c > rw.E - rw.S) { bad = 1; break; }
    if (ti == 0 && c == 0) break;                                       // flat index 0 (:631)
    if (++steps > step_cap) { bad = 1; break; }
    const unsigned char pb = s_path[rw.C - pBase + (unsigned)c];
    long nti = ti;
    if (mat == 0) {                                                     // :632-648
      const int code = pb & 7;
      if (code == C_DELCLOSE) mat = 1;
      else if (code == C_INSCLOSE) mat = 2;
      else if (code == C_DIAG) { op(0, 1); nti = ti - 1; qa--; }
      else if (code == C_LEFT) { op(1, 1); qa--; }
      else if (code == C_DOWN) { op(2, 1); nti = ti - 1; }
      else { bad = 1; break; }                                          // boundary arrow: endless loop in the reference
    } else if (mat == 1) {                                              // :649-659
      op(2, 1);
      mat = ((pb >> 3) & 1) ? 0 : 1;
      nti = ti - 1;
    } else {                                                            // :660-671
      op(1, 1);
      mat = ((pb >> 4) & 1) ? 0 : 2;
      qa--;
    }
    if (nti != ti) {
      if (nti < 0) { bad = 1; break; }
      ti = nti;
    }
  }
  if (!bad) {
    op(0, 1);                                                           // the aligned first base (:674)
    if (pending) emit_block();
  }
  if (lane == 0) {
    if (bad) T.s_status[s] |= 2;
    T.s_nblk[s] = bad ? 0 : (uint32_t)nblk;
  }
}

// ---------------------------------------------------------------------------------- gather
struct GatherArgs {
  uint64_t n_item;
  const int32_t* i_kind; const int32_t* i_data; const int32_t* i_aln; const uint64_t* i_out_off;
  const uint64_t* seg_off;            // per alignment
  const int32_t* s_kind; const int32_t* s_qStart; const int32_t* s_tStart; const uint32_t* s_aog_idx;
  const int32_t* aog_blocks; const uint64_t* aog_block_off; const int32_t* aog_nblocks;
  const int32_t* tmp_blocks; const uint64_t* s_tmp_off; const uint32_t* s_nblk;
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
    if (G.s_kind[s] == 0) {                                             // DP segment: the walk left its blocks back to front
      const int32_t* src = G.tmp_blocks + 3 * G.s_tmp_off[s];
      const int n = (int)G.s_nblk[s];
      for (int x = lane; x < n; x += 64) {
        const int y = n - 1 - x;
        out[3 * x] = src[3 * y]; out[3 * x + 1] = src[3 * y + 1]; out[3 * x + 2] = src[3 * y + 2];
      }
      continue;
    }
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
  (void)lra_exclusive_scan<CT>(ctx, n, c, off);
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
  // ---- per-alignment arrays (they outlive the call: status, block offsets), then the segment / item counts
  const size_t perAln = ((nA * 4 + 255) & ~(size_t)255) * 3 + (((nA + 1) * 8 + 255) & ~(size_t)255) * 3;
  char* baseP = (char*)lra_ensure(ctx, 70, perAln + 4096);
  if (!baseP) return LRA_ERR_NOMEM;
  Arena ap{baseP, perAln + 4096, 0};
  IRArgs A;
  memset(&A, 0, sizeof A);
  A.n_aln = n_aln; A.blocks_in = d_blocks_in; A.block_off = d_block_off;
  A.qseq = d_qseq; A.q_off = d_q_off; A.q_len = d_q_len; A.tseq = d_tseq; A.t_off = d_t_off; A.t_len = d_t_len;
  A.k = refine_band; A.match = match; A.mismatch = mismatch; A.indel = indel; A.endAlign = end_align;
  A.a_nseg = ap.get<uint32_t>(nA); A.a_nitem = ap.get<uint32_t>(nA); A.a_status = ap.get<int32_t>(nA);
  uint64_t* seg_off = ap.get<uint64_t>(nA + 1); uint64_t* item_off = ap.get<uint64_t>(nA + 1); uint64_t* out_block_off = ap.get<uint64_t>(nA + 1);
  if (!out_block_off) return lra_set_err(ctx, LRA_ERR_NOMEM, "arena accounting");
  A.seg_off = seg_off; A.item_off = item_off;
  const int nbA = (n_aln + SEG_LANES - 1) / SEG_LANES;
  lra_time_begin(ctx, "ir_segment");
  hipLaunchKernelGGL(ir_segment<false>, dim3(nbA), dim3(64), 0, st, A);
  lra_time_end(ctx);
  scan(ctx, (long)n_aln, A.a_nseg, seg_off);
  scan(ctx, (long)n_aln, A.a_nitem, item_off);
  uint64_t n_seg = 0, n_item = 0;
  if (d2h(ctx, &n_seg, seg_off + n_aln, 8) || d2h(ctx, &n_item, item_off + n_aln, 8)) return LRA_ERR_HIP;
  // ---- arena A (scratch slot 2): the per-segment / per-item arrays, sized by the counts
  const size_t capSeg = (size_t)n_seg + 1, capItem = (size_t)n_item + 1;
  size_t needA = 0;
  auto add = [&](size_t n, size_t sz) { needA += ((n * sz + 255) & ~(size_t)255); };
  add(3 * nBk, 4);
  for (int i = 0; i < 8; i++) add(capSeg, 4);
  add(3 * capSeg, 4); add(capSeg, 4); add(capSeg, 8); add(capSeg, 4);
  add(capSeg + 1, 8); add(capSeg + 1, 8); add(capSeg, 8); add(capSeg + 1, 8); add(capSeg, 4);
  add(capSeg, 4); add(capSeg, 4); add(capSeg, 8); add(capSeg + 1, 8); add(capSeg, 4); add(capSeg + 1, 8);
  add(3 * capSeg, 4); add(FILL_BINS + 16, 4);
  add(capItem, 4); add(3 * capItem, 4); add(capItem, 4); add(capItem, 4); add(capItem + 1, 8);
  add(capSeg, 8); add(capSeg, 4); add(capSeg, 8); add(capSeg, 4); add(capSeg, 4); add(capSeg, 4); add(capSeg + 1, 8);
  add(capSeg, 4); add(capSeg, 4); add(capSeg, 4);
  add(capSeg, 4); add(capSeg + 1, 8);
  char* baseA = (char*)lra_scratch(ctx, 2, needA + 4096);
  if (!baseA) return LRA_ERR_NOMEM;
  Arena ar{baseA, needA + 4096, 0};
  A.ab = ar.get<int32_t>(3 * nBk);
  A.s_aln = ar.get<int32_t>(capSeg); A.s_kind = ar.get<int32_t>(capSeg); A.s_qStart = ar.get<int32_t>(capSeg); A.s_tStart = ar.get<int32_t>(capSeg);
  A.s_qEnd = ar.get<int32_t>(capSeg); A.s_tEnd = ar.get<int32_t>(capSeg); A.s_b0 = ar.get<int32_t>(capSeg); A.s_b1 = ar.get<int32_t>(capSeg);
  A.s_first = ar.get<int32_t>(3 * capSeg); A.s_lastLen = ar.get<int32_t>(capSeg); A.s_rows = ar.get<uint64_t>(capSeg); A.s_isAog = ar.get<uint32_t>(capSeg);
  uint64_t* s_row_off = ar.get<uint64_t>(capSeg + 1); uint64_t* s_aog_off = ar.get<uint64_t>(capSeg + 1);
  uint64_t* s_cells = ar.get<uint64_t>(capSeg); uint64_t* s_cell_off = ar.get<uint64_t>(capSeg + 1); int32_t* s_status = ar.get<int32_t>(capSeg);
  uint32_t* s_nblk = ar.get<uint32_t>(capSeg); int32_t* s_width = ar.get<int32_t>(capSeg); uint64_t* s_tmpcap = ar.get<uint64_t>(capSeg);
  uint64_t* s_tmp_off = ar.get<uint64_t>(capSeg + 1); uint32_t* s_aog_idx = ar.get<uint32_t>(capSeg); uint64_t* s_out_off = ar.get<uint64_t>(capSeg + 1);
  uint32_t* fill_lists = ar.get<uint32_t>(3 * capSeg); int* fill_counts = ar.get<int>(FILL_BINS + 16);
  A.i_kind = ar.get<int32_t>(capItem); A.i_data = ar.get<int32_t>(3 * capItem);
  int32_t* i_aln = ar.get<int32_t>(capItem); uint32_t* i_count = ar.get<uint32_t>(capItem); uint64_t* i_out_off = ar.get<uint64_t>(capItem + 1);
  uint64_t* p_q_off = ar.get<uint64_t>(capSeg); int32_t* p_q_len = ar.get<int32_t>(capSeg); uint64_t* p_t_off = ar.get<uint64_t>(capSeg);
  int32_t* p_t_len = ar.get<int32_t>(capSeg); int32_t* p_k = ar.get<int32_t>(capSeg); uint32_t* p_cap = ar.get<uint32_t>(capSeg);
  uint64_t* p_block_off = ar.get<uint64_t>(capSeg + 1);
  int32_t* p_score = ar.get<int32_t>(capSeg); int32_t* p_nblocks = ar.get<int32_t>(capSeg); int32_t* p_status = ar.get<int32_t>(capSeg);
  uint32_t* s_nchunk = ar.get<uint32_t>(capSeg); uint64_t* chunk_off = ar.get<uint64_t>(capSeg + 1);
  if (!chunk_off) return lra_set_err(ctx, LRA_ERR_NOMEM, "arena accounting");
  // ---- segments: emit
  lra_time_begin(ctx, "ir_segment");
  hipLaunchKernelGGL(ir_segment<true>, dim3(nbA), dim3(64), 0, st, A);
  lra_time_end(ctx);
  hipLaunchKernelGGL(ir_item_aln, dim3((n_aln + 255) / 256), dim3(256), 0, st, n_aln, item_off, i_aln);
  uint64_t n_rows = 0, n_aog = 0, n_cells = 0;
  const int32_t* tmp_blocks = nullptr;
  if (n_seg) {
    scan(ctx, (long)n_seg, A.s_rows, s_row_off);
    scan(ctx, (long)n_seg, A.s_isAog, s_aog_off);
    if (d2h(ctx, &n_rows, s_row_off + n_seg, 8) || d2h(ctx, &n_aog, s_aog_off + n_seg, 8)) return LRA_ERR_HIP;
    Row* rows = (Row*)lra_ensure(ctx, 0, (n_rows + 1) * sizeof(Row));
    if (!rows) return LRA_ERR_NOMEM;
    BandArgs B;
    B.n_seg = n_seg; B.s_aln = A.s_aln; B.s_kind = A.s_kind; B.s_qStart = A.s_qStart; B.s_qEnd = A.s_qEnd; B.s_tStart = A.s_tStart;
    B.s_b0 = A.s_b0; B.s_b1 = A.s_b1; B.s_first = A.s_first; B.s_lastLen = A.s_lastLen; B.s_rows = A.s_rows; B.s_row_off = s_row_off;
    B.ab = A.ab; B.block_off = d_block_off; B.tseq = d_tseq; B.t_off = d_t_off; B.k = refine_band; B.rows = rows;
    B.s_cells = s_cells; B.s_status = s_status; B.s_width = s_width; B.s_tmpcap = s_tmpcap;
    const unsigned gridL = (unsigned)((n_seg + 63) / 64);
    const unsigned gridS = (unsigned)((n_seg + 255) / 256);
    lra_time_begin(ctx, "ir_band");
    B.s_nchunk = s_nchunk; B.chunk_off = chunk_off; B.n_task = 0; B.task_seg = nullptr; B.chunkT = nullptr; B.chunkQ = nullptr;
    hipLaunchKernelGGL(ir_band_nchunk, dim3(gridS), dim3(256), 0, st, B);
    scan(ctx, (long)n_seg, s_nchunk, chunk_off);
    uint64_t n_task = 0;
    if (d2h(ctx, &n_task, chunk_off + n_seg, 8)) return LRA_ERR_HIP;
    uint32_t* task_seg = (uint32_t*)lra_ensure(ctx, 54, (n_task + 1) * 4);
    int32_t* chunk_tq = (int32_t*)lra_ensure(ctx, 55, (n_task + 1) * 8);
    if (!task_seg || !chunk_tq) return LRA_ERR_NOMEM;
    B.n_task = n_task; B.task_seg = task_seg; B.chunkT = chunk_tq; B.chunkQ = chunk_tq + n_task + 1;
    hipLaunchKernelGGL(ir_band_tasks, dim3(gridS), dim3(256), 0, st, B);
    hipLaunchKernelGGL(ir_band_prep, dim3((unsigned)n_seg), dim3(64), 0, st, B);
    if (n_task) {
      const unsigned gridT = (unsigned)((n_task + 63) / 64);
      if (refine_band <= 8) hipLaunchKernelGGL(ir_band_chunk<16>, dim3(gridT), dim3(64), 0, st, B);
      else if (refine_band <= 32) hipLaunchKernelGGL(ir_band_chunk<64>, dim3(gridT), dim3(64), 0, st, B);
      else hipLaunchKernelGGL(ir_band_chunk<128>, dim3(gridT), dim3(64), 0, st, B);
    }
    hipLaunchKernelGGL(ir_band_scan, dim3((unsigned)n_seg), dim3(64), 0, st, B);
    lra_time_end(ctx);
    scan(ctx, (long)n_seg, s_cells, s_cell_off);
    scan(ctx, (long)n_seg, s_tmpcap, s_tmp_off);
    uint64_t n_tmp = 0;
    if (d2h(ctx, &n_cells, s_cell_off + n_seg, 8) || d2h(ctx, &n_tmp, s_tmp_off + n_seg, 8)) return LRA_ERR_HIP;
    // The arrows (1 B per cell) and the walk-order blocks are this stage's two large temporaries; they live in the buffer of the sparse DP's arena (slot 12):
    // on the path the arena is dead by now (the last sparse DP ran inside LocalRefineAlignment) and this stage is over before the next batch's first one.
    const size_t pathBytes = (n_cells + 256 + 255) & ~(size_t)255;
    unsigned char* path = (unsigned char*)lra_ensure(ctx, 12, pathBytes + (n_tmp + 1) * 12 + 256);
    if (!path) return LRA_ERR_NOMEM;
    int32_t* tmpb = (int32_t*)(path + pathBytes);
    tmp_blocks = tmpb;
    int h_bins[FILL_BINS], h_start[FILL_BINS], h_cursor[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    LRA_HIP_CHECK(ctx, hipMemsetAsync(fill_counts, 0, FILL_BINS * 4, st));
    hipLaunchKernelGGL(ir_classify<false>, dim3((unsigned)((n_seg + 255) / 256)), dim3(256), 0, st, n_seg, A.s_kind, s_status, s_width, A.s_rows, fill_counts, fill_lists);
    if (d2h(ctx, h_bins, fill_counts, FILL_BINS * 4)) return LRA_ERR_HIP;
    { int run = 0; for (int b = 0; b < FILL_BINS; b++) { h_start[b] = run; run += h_bins[b]; if ((b & 31) == 0) h_cursor[b / 32] = h_start[b]; if ((b & 31) == 31) h_cursor[4 + b / 32] = run; } }
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(fill_counts, h_start, FILL_BINS * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(ir_classify<true>, dim3((unsigned)((n_seg + 255) / 256)), dim3(256), 0, st, n_seg, A.s_kind, s_status, s_width, A.s_rows, fill_counts, fill_lists);
    int* fill_cursor = fill_counts + FILL_BINS;
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(fill_cursor, h_cursor, 32, hipMemcpyHostToDevice, st));
    FillArgs F;
    F.n_seg = n_seg; F.s_aln = A.s_aln; F.s_rows = A.s_rows; F.s_row_off = s_row_off; F.s_cell_off = s_cell_off; F.rows = rows;
    F.qseq = d_qseq; F.q_off = d_q_off; F.q_len = d_q_len; F.match = match; F.mismatch = mismatch; F.g = indel; F.path = path;
    F.list = fill_lists; F.cursor = fill_cursor;
    const unsigned cap_grid = (unsigned)ctx->num_cu * 32;
    if (getenv("LRA_IR_DBG")) {
      const int hc[3] = {h_cursor[4] - h_cursor[0], h_cursor[5] - h_cursor[1], h_cursor[6] - h_cursor[2]};
      fprintf(stderr, "[ir] n_seg %llu n_rows %llu n_cells %llu n_task %llu fill classes 16/32/64: %d %d %d\n", (unsigned long long)n_seg,
              (unsigned long long)n_rows, (unsigned long long)n_cells, (unsigned long long)n_task, hc[0], hc[1], hc[2]);
    }
    lra_time_begin(ctx, "ir_fill");
    const uint64_t n16 = (uint64_t)(h_cursor[4] - h_cursor[0]), n32 = (uint64_t)(h_cursor[5] - h_cursor[1]), n64 = (uint64_t)(h_cursor[6] - h_cursor[2]);
    // a kernel's time is its longest segment's row chain, whatever the class: the classes side by side (the widest class of segments, usually the bulk, on the context's stream)
    static const bool no16x2 = getenv("LRA_IR_NO16X2") != nullptr;
    const uint64_t nWide = (uint64_t)(h_cursor[7] - h_cursor[3]);
    // classes 0 and 1 (rows of at most 32 cells: nearly every segment of a read): the anti-diagonal kernel, eight segments per wave; a class-1 segment whose wide rows
    // do not fit its schedule is listed by it and redone by ir_fill_16x2 behind it (the list's count never leaves the device).  LRA_IR_DIAG=1 switches it on.
    static const bool diag = getenv("LRA_IR_DIAG") && atoi(getenv("LRA_IR_DIAG")) == 1;   // (off: measured slower so far -- half its lanes idle, a memory wait per step)
    uint32_t* retry_list = fill_lists + capSeg; int* retry_cursor = fill_counts + FILL_BINS + 8;
    if (diag) LRA_HIP_CHECK(ctx, hipMemsetAsync(retry_cursor, 0, 32, st));
    if (n16 && !diag) hipLaunchKernelGGL(ir_fill<16>, dim3((unsigned)std::min<uint64_t>((n16 + 3) / 4, cap_grid)), dim3(64), 0, lra_side_fork(ctx, 1), F);
    if (n16 && diag) hipLaunchKernelGGL(ir_fill_diag, dim3((unsigned)std::min<uint64_t>((n16 + 7) / 8, cap_grid)), dim3(64), 0, lra_side_fork(ctx, 1), F, 0, retry_list, retry_cursor);
    if (n64) hipLaunchKernelGGL(ir_fill<64>, dim3((unsigned)std::min<uint64_t>(n64, cap_grid)), dim3(64), 0, lra_side_fork(ctx, 2), F);
    if (nWide) hipLaunchKernelGGL(ir_fill_wide, dim3((unsigned)std::min<uint64_t>(nWide, cap_grid)), dim3(64), 0, lra_side_fork(ctx, 3), F);
    if (n32 && no16x2 && !diag) hipLaunchKernelGGL(ir_fill<32>, dim3((unsigned)std::min<uint64_t>((n32 + 1) / 2, cap_grid)), dim3(64), 0, st, F);
    if (n32 && !no16x2 && !diag) hipLaunchKernelGGL(ir_fill_16x2, dim3((unsigned)std::min<uint64_t>((n32 + 3) / 4, cap_grid)), dim3(64), 0, st, F);
    if (n32 && diag) {
      hipLaunchKernelGGL(ir_fill_diag, dim3((unsigned)std::min<uint64_t>((n32 + 7) / 8, cap_grid)), dim3(64), 0, st, F, 1, retry_list, retry_cursor);
      FillArgs F2 = F; F2.list = retry_list; F2.cursor = retry_cursor;
      hipLaunchKernelGGL(ir_fill_16x2, dim3((unsigned)std::min<uint64_t>((n32 + 3) / 4, 2048)), dim3(64), 0, st, F2);
    }
    if (n16) lra_side_join(ctx, 1);
    if (n64) lra_side_join(ctx, 2);
    if (nWide) lra_side_join(ctx, 3);
    lra_time_end(ctx);
    if (diag && getenv("LRA_IR_DIAG_CHECK")) {                             // debugging: the row-wise kernels' arrows of the same segments beside the anti-diagonal kernel's
      int h_retry[8];
      if (d2h(ctx, h_retry, retry_cursor, 32)) return LRA_ERR_HIP;
      unsigned char* path2 = nullptr; unsigned long long* d_bad = nullptr;
      LRA_HIP_CHECK(ctx, hipMalloc((void**)&path2, n_cells + 512));
      LRA_HIP_CHECK(ctx, hipMalloc((void**)&d_bad, 64));
      const unsigned long long init[3] = {0, 0, ~0ULL};
      LRA_HIP_CHECK(ctx, hipMemcpyAsync(d_bad, init, 24, hipMemcpyHostToDevice, st));
      LRA_HIP_CHECK(ctx, hipMemcpyAsync(fill_cursor, h_cursor, 32, hipMemcpyHostToDevice, st));
      FillArgs F3 = F; F3.path = path2;
      if (n16) hipLaunchKernelGGL(ir_fill<16>, dim3((unsigned)std::min<uint64_t>((n16 + 3) / 4, cap_grid)), dim3(64), 0, st, F3);
      if (n32) hipLaunchKernelGGL(ir_fill_16x2, dim3((unsigned)std::min<uint64_t>((n32 + 3) / 4, cap_grid)), dim3(64), 0, st, F3);
      if (n16) hipLaunchKernelGGL(ir_fill_compare, dim3((unsigned)n16), dim3(256), 0, st, (const uint32_t*)fill_lists, (long)h_cursor[0], (long)n16, s_cell_off, s_cells, path, path2, d_bad);
      if (n32) hipLaunchKernelGGL(ir_fill_compare, dim3((unsigned)n32), dim3(256), 0, st, (const uint32_t*)fill_lists, (long)h_cursor[1], (long)n32, s_cell_off, s_cells, path, path2, d_bad);
      unsigned long long h_bad[3];
      if (d2h(ctx, h_bad, d_bad, 24)) return LRA_ERR_HIP;
      fprintf(stderr, "[ir diag check] segments %llu + %llu, redone row-wise %d; cells that differ %llu in %llu segments%s", (unsigned long long)n16, (unsigned long long)n32,
              h_retry[5], h_bad[0], h_bad[1], h_bad[1] ? "" : "\n");
      if (h_bad[1]) fprintf(stderr, "; first: segment %llu cell %llu\n", h_bad[2] >> 32, h_bad[2] & 0xffffffffULL);
      (void)hipFree(path2); (void)hipFree(d_bad);
    }
    TraceArgs T;
    T.n_seg = n_seg; T.s_kind = A.s_kind; T.s_tStart = A.s_tStart; T.s_rows = A.s_rows; T.s_row_off = s_row_off;
    T.s_cells = s_cells; T.s_cell_off = s_cell_off; T.s_status = s_status; T.rows = rows; T.path = path;
    T.s_nblk = s_nblk; T.s_tmp_off = s_tmp_off; T.tmp_blocks = tmpb;
    lra_time_begin(ctx, "ir_trace");
    hipLaunchKernelGGL(ir_trace_wave, dim3((unsigned)n_seg), dim3(64), 0, st, T);
    lra_time_end(ctx);
    // ---- short segments -> AffineOneGapAlign (:344-357)
    if (n_aog) {
      hipLaunchKernelGGL(ir_aog_setup, dim3((unsigned)((n_seg + 255) / 256)), dim3(256), 0, st, n_seg, A.s_kind, A.s_aln, s_aog_off, A.s_qStart,
                         A.s_tStart, A.s_qEnd, A.s_tEnd, d_q_off, d_t_off, refine_band, s_aog_idx, p_q_off, p_q_len, p_t_off, p_t_len, p_k, p_cap);
      scan(ctx, (long)n_aog, p_cap, p_block_off);
      uint64_t aog_cap = 0;
      if (d2h(ctx, &aog_cap, p_block_off + n_aog, 8)) return LRA_ERR_HIP;
      if (aog_cap * 12 + 64 > ctx->aux_bytes) {
        if (ctx->aux) (void)hipFree(ctx->aux);
        ctx->aux = nullptr; ctx->aux_bytes = 0;
        size_t want = aog_cap * 12 + aog_cap * 3 + 4096;
        if (hipMalloc(&ctx->aux, want) != hipSuccess) return lra_set_err(ctx, LRA_ERR_NOMEM, "aog block buffer");
        ctx->aux_bytes = want;
      }
      int rc = lra_aog_launch_device(ctx, (int)n_aog, d_qseq, d_tseq, p_q_off, p_q_len, p_t_off, p_t_len, p_k, match, mismatch, indel, p_score,
                                     p_nblocks, (int32_t*)ctx->aux, p_block_off, p_status);
      if (rc) return rc;
    }
  }
  // ---- output sizing and assembly
  hipLaunchKernelGGL(ir_item_counts, dim3((unsigned)((n_item + 255) / 256)), dim3(256), 0, st, n_item, A.i_kind, A.i_data, i_aln, seg_off, A.s_kind,
                     s_nblk, s_aog_idx, p_nblocks, i_count);
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
  hipLaunchKernelGGL(ir_finalize_offsets, dim3((unsigned)((std::max<uint64_t>(n_item, nA + 1) + 255) / 256)), dim3(256), 0, st, n_aln, n_item, item_off,
                     i_out_off, A.i_kind, A.i_data, i_aln, seg_off, s_out_off, out_block_off);
  GatherArgs G;
  G.n_item = n_item; G.i_kind = A.i_kind; G.i_data = A.i_data; G.i_aln = i_aln; G.i_out_off = i_out_off; G.seg_off = seg_off;
  G.s_kind = A.s_kind; G.s_qStart = A.s_qStart; G.s_tStart = A.s_tStart; G.s_aog_idx = s_aog_idx;
  G.aog_blocks = (const int32_t*)ctx->aux; G.aog_block_off = p_block_off; G.aog_nblocks = p_nblocks;
  G.tmp_blocks = tmp_blocks; G.s_tmp_off = s_tmp_off; G.s_nblk = s_nblk; G.out_blocks = out_blocks;
  lra_time_begin(ctx, "ir_gather");
  if (n_item) hipLaunchKernelGGL(ir_gather, dim3((unsigned)std::min<uint64_t>(n_item, 65535 * 4)), dim3(64), 0, st, G);
  lra_time_end(ctx);
  hipLaunchKernelGGL(ir_aln_status, dim3((n_aln + 255) / 256), dim3(256), 0, st, n_aln, seg_off, A.s_kind, s_status, s_aog_idx, p_status, A.a_status);
  LRA_HIP_CHECK(ctx, hipGetLastError());
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  out->n_aln = n_aln; out->n_blocks = n_out; out->n_segments = n_seg; out->n_cells = n_cells; out->n_rows = n_rows; out->n_aog = n_aog;
  out->d_block_off = out_block_off; out->d_blocks = out_blocks; out->d_status = A.a_status;
  return LRA_OK;
}
