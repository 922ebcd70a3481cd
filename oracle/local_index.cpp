// oracle/local_index.cpp -- TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// CPU restatement of the tier-2 (local) minimizer machinery:
//   StoreMinimizers_noncanonical<LocalTuple,SmallTuple>   MinCount.h:182-338
//   LocalIndex::IndexSeq (per-window lists: sort + RemoveFrequent)   MMIndex.h:200-245, :69-84
//   CompareLists<LocalTuple,SmallTuple> (Global == false)            CompareLists.h:9-146
//   AppendValues (diagonal band + box filter)                        TupleOps.h:159-195
// A LocalTuple is the 32-bit word  t | pos << 20  (TupleOps.h:20-25: `t:20, pos:12`, first field in the
// low bits); all comparisons are on the 20-bit t.
// Parity status: CompareLists<LocalTuple> and AppendValues are PINNED (reference templates compiled in
// place, oracle/ref_harness/comparelists_ref.cpp mode 1, tests/golden/local_compare_golden.json);
// StoreMinimizers_noncanonical / IndexSeq are PARITY UNPINNED (MinCount.h, MMIndex.h need htslib).
#include "oracle_common.h"
#include <algorithm>
#include <vector>

namespace {
const uint32_t TMASK = 0xFFFFF;
inline uint32_t T_(uint32_t v) { return v & TMASK; }
inline uint32_t P_(uint32_t v) { return v >> 20; }
}

// non-canonical (w,k)-minimizers of seq[0,seqLen) as LocalTuples (pos relative to seq); returns count.
extern "C" long oracle_store_minimizers_noncanonical(const char* seq, uint32_t seqLen, int k, int w, uint32_t* out, long cap) {
  long n = 0;
  auto emit = [&](uint32_t t, uint32_t p) { if (n < cap) out[n] = (t & TMASK) | ((p & 0xFFF) << 20); n++; };
  if (seqLen < (uint32_t)k) return 0;                                   // :186
  const int span = w + k - 1;
  if (seqLen < (uint32_t)span) return 0;                                // :199
  const uint32_t kmask = (k >= 16) ? 0xFFFFFFFFu : ((1u << (2 * k)) - 1);   // InitMask on a 20-bit field
  long nvStart = 0, nvEnd = 0;
  bool valid = false;
  auto find_valid = [&]() -> bool {                                     // :200-214 / :293-307
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
  uint32_t cur = 0;
  for (int p = 0; p < k; p++) cur = ((cur << 2) + (uint32_t)oracle_code((unsigned char)seq[p])) & TMASK;   // StoreTuple into t:20
  auto shift = [&](uint32_t at) { cur = ((((cur << 2) & TMASK) & kmask) + (uint32_t)oracle_code((unsigned char)seq[at])) & TMASK; };
  std::vector<uint32_t> ringT(w), ringP(w);
  uint32_t actT = cur, actP = 0;
  ringT[0] = actT; ringP[0] = 0;
  uint32_t p;
  for (p = 1; p < (uint32_t)w && p < seqLen - k + 1; p++) {             // :251-270
    shift(p + k - 1);
    if (cur < actT) { actT = cur; actP = p; }
    ringT[p % w] = cur; ringP[p % w] = p;
  }
  if (nvEnd == span) emit(actT, actP);                                  // :271-273
  for (p = w; p < seqLen - k + 1; p++) {                                // :276-337
    shift(p + k - 1);
    if (nvEnd == (long)(p + k - 1)) {
      if (oracle_code_n((unsigned char)seq[p + k - 1]) <= 3) nvEnd++;
      else {
        nvStart = p + k;
        if (!find_valid()) return n;
        nvEnd = nvStart + span;
      }
    }
    ringT[p % w] = cur; ringP[p % w] = p;
    if (p - w >= actP) {
      actT = ringT[0]; actP = ringP[0];
      for (int j = 1; j < w; j++) if (ringT[j] < actT) { actT = ringT[j]; actP = ringP[j]; }
      if (nvEnd == (long)(p + k)) emit(actT, actP);
    } else if (cur < actT) {
      actT = cur; actP = p;
      if (nvEnd == (long)(p + k)) emit(actT, actP);
    }
  }
  return n;
}

// LocalIndex::IndexSeq: per window of `window` bases: minimizers, sorted by t (std::sort), runs of >= maxFreq
// equal t removed.  tuples (cap) + boundaries[nWindows+1].  Returns the number of windows.
extern "C" long oracle_local_index_seq(const char* seq, long seqLen, int k, int w, int window, int maxFreq, uint32_t* tuples, long cap,
                                       uint64_t* boundaries) {
  long nIndex = seqLen / window + (seqLen % window != 0 ? 1 : 0);       // :201-206
  long total = 0;
  boundaries[0] = 0;
  long seqPos = 0;
  std::vector<uint32_t> loc(window + 8);
  for (long i = 0; i < nIndex; i++) {
    const long len = std::min(seqLen, seqPos + window) - seqPos;
    long n = oracle_store_minimizers_noncanonical(seq + seqPos, (uint32_t)len, k, w, loc.data(), (long)loc.size());
    std::sort(loc.begin(), loc.begin() + n, [](uint32_t a, uint32_t b) { return T_(a) < T_(b); });   // :219 (LocalTuple::operator<)
    long c = 0, x = 0;                                                  // RemoveFrequent :69-84
    while (x < n) {
      long ne = x;
      while (ne < n && T_(loc[ne]) == T_(loc[x])) ne++;
      if (ne - x < maxFreq) for (long y = x; y < ne; y++) loc[c++] = loc[y];
      x = ne;
    }
    for (long y = 0; y < c; y++) { if (total < cap) tuples[total] = loc[y]; total++; }
    seqPos += std::min((long)window, seqLen - seqPos);
    boundaries[i + 1] = (uint64_t)total;
  }
  return nIndex;
}

// CompareLists<LocalTuple,SmallTuple>(qBegin,qEnd,tBegin,tEnd,result,opts,Global=false,maxDiagNum,minDiagNum)
extern "C" long oracle_compare_lists_local(const uint32_t* q, long nq, const uint32_t* t, long nt, long maxFreq, int64_t maxDiag,
                                           int64_t minDiag, uint32_t* out_qi, uint32_t* out_ti, long cap) {
  long n = 0;
  if (nq == 0 || nt == 0) return 0;
  auto Q = [&](long i) { return T_(q[i]); };
  auto T = [&](long i) { return T_(t[i]); };
  auto emit = [&](long qi, long ti) {
    if (maxDiag != 0 && minDiag != 0) {
      int64_t d = (int64_t)P_(t[ti]) - (int64_t)P_(q[qi]);
      if (!(d <= maxDiag && d >= minDiag)) return;
    }
    if (n < cap) { out_qi[n] = (uint32_t)qi; out_ti[n] = (uint32_t)ti; }
    n++;
  };
  long qs = 0, qe = nq - 1, ts = 0, te = nt;
  do {
    while (qs <= qe && Q(qs) < T(ts)) qs++;
    if (qs >= qe) return n;
    const uint32_t startGap = (Q(qs) - T(ts)) & TMASK;                  // gaps live in a 20-bit field
    while (qe > qs && te > ts && Q(qe) > T(te - 1)) qe--;
    const uint32_t endGap = (T(te - 1) - Q(qe)) & TMASK;
    if (startGap == 0 || startGap > endGap) {
      const long tsOrig = ts, qsOrig = qs;
      long lo = ts, hi = te;
      while (lo < hi) { long mid = lo + (hi - lo) / 2; if (T(mid) < Q(qs)) lo = mid + 1; else hi = mid; }
      ts = lo;
      if (ts < te && T(ts) == Q(qs)) {
        long tsi = ts;
        while (tsi != te && Q(qs) == T(tsi)) tsi++;
        const long qsStart = qs;
        while (qs < qe && Q(qs + 1) == Q(qs)) qs++;
        for (long ti = ts; ti != tsi; ti++)
          if (qs - qsStart < maxFreq)
            for (long qi = qsStart; qi <= qs; qi++) emit(qi, ti);
      }
      while (ts < te && T(ts) == T(tsOrig)) ts++;
      while (qs < qe && Q(qs) == Q(qsOrig)) qs++;
    } else {
      if (te != nt && T(te - 1) == Q(qe)) {
      } else {
        long lo = ts, hi = te;
        while (lo < hi) { long mid = lo + (hi - lo) / 2; if (!(Q(qe) < T(mid))) lo = mid + 1; else hi = mid; }
        te = lo;
      }
      const long teStart = te;
      long tei = te;
      while (tei > ts && T(tei - 1) == Q(qe)) tei--;
      if (tei < teStart && teStart > 0) {
        const long qeStart = qe;
        while (qe > qs && Q(qe) == Q(qe - 1)) qe--;
        for (long ti = tei; ti < teStart; ti++)
          if (qeStart - qe < maxFreq)
            for (long qi = qe; qi <= qeStart; qi++) emit(qi, ti);
      }
      te = tei;
    }
  } while (qs < qe && ts < te);
  return n;
}
