// oracle/refine_space.cpp -- TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// CPU restatement of RefineSpace (ClusterRefine.h:242-325), the gap-seeding function of SURVEY §8a row a11: anchors inside one
// (read span x genome span) gap.  Both spans < 1000: AffineOneGapAlign with k = 30, exact K-mers every K bases of its blocks,
// identity = matching bases / min(span).  Otherwise: non-canonical (W,K) minimizers of both spans as GenomeTuples
// (StoreMinimizers_noncanonical<GenomeTuple,Tuple>, MinCount.h:182-338), std::sort, CompareLists<GenomeTuple,Tuple> with the
// diagonal band [min(0,diag2) - refineSpaceDiag, max(0,diag2) + refineSpaceDiag] (CompareLists.h:9, Global = false, canonical = false).
// Parity status: the pieces it calls are PINNED where the reference compiles (oracle_affine_one_gap_align, oracle_compare_lists,
// std::sort); the minimizer sketch and RefineSpace's own glue are PARITY UNPINNED (MinCount.h, ClusterRefine.h need htslib).
#include "oracle_common.h"
#include <algorithm>
#include <vector>

extern "C" int oracle_affine_one_gap_align(const char* q, int qLen, const char* t, int tLen, int m, int mm, int indel, int k, int* blocks, int cap,
                                           int* nBlocks, int* status);
extern "C" void oracle_sort_minimizers(uint64_t* keys, uint32_t* poss, long n);
extern "C" long oracle_compare_lists(const uint64_t* qk, const uint32_t* qp, long nq, const uint64_t* tk, const uint32_t* tp, long nt, long maxFreq,
                                     int64_t maxDiag, int64_t minDiag, uint32_t* out_qi, uint32_t* out_ti, long cap);

// StoreMinimizers_noncanonical<GenomeTuple,Tuple>(seq, seqLen, k, w, out, Global = false)   MinCount.h:182-338
extern "C" long oracle_store_minimizers_noncanonical64(const char* seq, uint32_t seqLen, int k, int w, uint64_t* keys, uint32_t* pos, long cap) {
  long n = 0;
  auto emit = [&](uint64_t t, uint32_t p) { if (n < cap) { keys[n] = t; pos[n] = p; } n++; };
  if (seqLen < (uint32_t)k) return 0;
  const int span = w + k - 1;
  if (seqLen < (uint32_t)span) return 0;
  uint64_t mask = 0;
  for (int i = 0; i < k; i++) { mask <<= 2; mask += 3; }                // InitMask TupleOps.h:95
  long nvStart = 0, nvEnd = 0;
  bool valid = false;
  auto find_valid = [&]() -> bool {
    valid = false;
    while ((uint32_t)nvStart < seqLen - (uint32_t)span && !valid) {
      valid = true;
      for (long x = nvStart; valid && x < nvStart + span; x++)
        if (oracle_code_n((unsigned char)seq[x]) > 3) { nvStart = x + 1; valid = false; }
    }
    return valid;
  };
  if (!find_valid()) return 0;
  nvEnd = nvStart + span;
  uint64_t cur = 0;
  for (int p = 0; p < k; p++) { cur <<= 2; cur += (uint64_t)oracle_code((unsigned char)seq[p]); }     // StoreTuple :104
  auto shift = [&](uint32_t at) { cur = ((cur << 2) & mask) + (uint64_t)oracle_code((unsigned char)seq[at]); };   // ShiftOne :114
  const uint64_t FM = 0x7FFFFFFFFFFFFFFFULL;                            // GenomeTuple::for_mask_s
  std::vector<uint64_t> ringT(w); std::vector<uint32_t> ringP(w);
  uint64_t actT = cur; uint32_t actP = 0;
  ringT[0] = actT; ringP[0] = 0;
  uint32_t p;
  for (p = 1; p < (uint32_t)w && p < seqLen - k + 1; p++) {
    shift(p + k - 1);
    const uint64_t c = cur & FM;
    if (c < actT) { actT = c; actP = p; }
    ringT[p % w] = c; ringP[p % w] = p;
  }
  if (nvEnd == span) emit(actT, actP);
  for (p = w; p < seqLen - k + 1; p++) {
    shift(p + k - 1);
    const uint64_t c = cur & FM;
    if (nvEnd == (long)(p + k - 1)) {
      if (oracle_code_n((unsigned char)seq[p + k - 1]) <= 3) nvEnd++;
      else {
        nvStart = p + k;
        if (!find_valid()) return n;
        nvEnd = nvStart + span;
      }
    }
    ringT[p % w] = c; ringP[p % w] = p;
    if (p - w >= actP) {
      actT = ringT[0]; actP = ringP[0];
      for (int j = 1; j < w; j++) if ((ringT[j] & FM) < (actT & FM)) { actT = ringT[j]; actP = ringP[j]; }
      if (nvEnd == (long)(p + k)) emit(actT, actP);
    } else if ((c & FM) < (actT & FM)) {
      actT = c; actP = p;
      if (nvEnd == (long)(p + k)) emit(actT, actP);
    }
  }
  return n;
}

// One gap.  q = strands[st] + qs (qLen = qe - qs), t = genome.seqs[chrom] + (ts - lrts) (tLen = te - ts + lrlength),
// tSpan = te - (ts - lrts) (GenomePos arithmetic).  Output pairs already carry "+= qs" / "+= ts - lrts" (qAdd, tAdd) and, if
// flipLen != 0 (consider_str and st == 1), first.pos = flipLen - first.pos - K.  Returns the number of pairs; *identity as the reference.
extern "C" long oracle_refine_space(const char* q, int qLen, const char* t, int tLen, uint32_t tSpan, int K, int W, int refineSpaceDiag, int match,
                                    int mismatch, int indel, long maxFreq, uint32_t qAdd, uint32_t tAdd, uint32_t flipLen, uint32_t* outQ,
                                    uint32_t* outT, long cap, float* identity) {
  int64_t diag1 = 0, diag2 = (int64_t)tSpan - (int64_t)(uint32_t)qLen;
  int64_t minDiagNum = std::min(diag1, diag2) - refineSpaceDiag, maxDiagNum = std::max(diag1, diag2) + refineSpaceDiag;
  *identity = -1;
  long n = 0;
  auto emit = [&](uint32_t a, uint32_t b) { if (n < cap) { outQ[n] = a; outT[n] = b; } n++; };
  if (qLen < 1000 && tLen < 1000) {
    int bcap = qLen + tLen + 8;
    std::vector<int> blocks(3 * (size_t)bcap);
    int nb = 0, st = 0;
    oracle_affine_one_gap_align(q, qLen, t, tLen, match, mismatch, indel, 30, blocks.data(), bcap, &nb, &st);
    int nMatch = 0;
    for (int b = 0; b < nb; b++) {
      const int bq = blocks[3 * b], bt = blocks[3 * b + 1], bl = blocks[3 * b + 2];
      for (int x = 0; x < bl; x++) if (q[bq + x] == t[bt + x]) nMatch++;
      if (bl > K)
        for (int bp = 0; bp + K < bl; bp += K) {
          bool mis = false;
          for (int x = 0; x < K; x++) if (t[bt + bp + x] != q[bq + bp + x]) { mis = true; break; }
          if (!mis) emit((uint32_t)(bq + bp), (uint32_t)(bt + bp));
        }
    }
    *identity = nMatch / (float)std::min(qLen, tLen);
  } else {
    std::vector<uint64_t> tk((size_t)tLen + 1), qk((size_t)qLen + 1);
    std::vector<uint32_t> tp((size_t)tLen + 1), qp((size_t)qLen + 1);
    long nt = oracle_store_minimizers_noncanonical64(t, (uint32_t)tLen, K, W, tk.data(), tp.data(), tLen + 1);
    oracle_sort_minimizers(tk.data(), tp.data(), nt);
    long nq = oracle_store_minimizers_noncanonical64(q, (uint32_t)qLen, K, W, qk.data(), qp.data(), qLen + 1);
    oracle_sort_minimizers(qk.data(), qp.data(), nq);
    long pc = 3 * (nq + 1) * 64 + 1024;
    std::vector<uint32_t> qi(pc), ti(pc);
    long np = oracle_compare_lists(qk.data(), qp.data(), nq, tk.data(), tp.data(), nt, maxFreq, maxDiagNum, minDiagNum, qi.data(), ti.data(), pc);
    if (np > pc) { qi.resize(np); ti.resize(np); np = oracle_compare_lists(qk.data(), qp.data(), nq, tk.data(), tp.data(), nt, maxFreq, maxDiagNum, minDiagNum, qi.data(), ti.data(), np); }
    for (long i = 0; i < np; i++) emit(qp[qi[i]], tp[ti[i]]);
  }
  const long m = std::min(n, cap);
  for (long i = 0; i < m; i++) {                                         // :313-323
    outQ[i] += qAdd; outT[i] += tAdd;
    if (flipLen) outQ[i] = flipLen - outQ[i] - (uint32_t)K;
  }
  return n;
}

// RefineByLinearAlignment (LocalRefineAlignment.h:141-185): the blocks appended to alignment->blocks for one anchor pair.
// q = strands[str], t = genome.seqs[chromIndex] (whole sequences); returns the number of blocks, -1 if a span is negative
// (std::string of negative length in the reference).
extern "C" int oracle_between_anchors(const char* q, const char* t, uint32_t curReadEnd, uint32_t nextReadStart, uint32_t curGenomeEnd,
                                      uint32_t nextGenomeStart, int match, int mismatch, int indel, int localBand, int refineDp, int* blocks, int cap,
                                      int* score) {
  const int m = (int)std::min(nextReadStart - curReadEnd + 1u, nextGenomeStart - curGenomeEnd + 1u);   // SetMatchAndGaps :93-98
  *score = 0;
  if (!(m > 0) || !refineDp) return 0;
  const int qLen = (int)(nextReadStart - curReadEnd), tLen = (int)(nextGenomeStart - curGenomeEnd);      // AlignSubstrings :103-104
  if (qLen < 0 || tLen < 0) return -1;
  const int drift = std::abs(qLen - tLen);
  int nb = 0, st = 0;
  *score = oracle_affine_one_gap_align(q + curReadEnd, qLen, t + curGenomeEnd, tLen, match, mismatch, indel, std::min(drift * 2 + 1, localBand), blocks,
                                       cap, &nb, &st);
  for (int b = 0; b < nb && b < cap; b++) { blocks[3 * b] += (int)curReadEnd; blocks[3 * b + 1] += (int)curGenomeEnd; }   // RefineSubstrings :134-138
  return nb;
}
