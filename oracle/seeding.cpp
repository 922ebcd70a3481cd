// oracle/seeding.cpp -- TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// CPU restatement of the tier-1 seeding stages of MapRead (reference: MapRead.h:169-203):
//   a1  StoreMinimizers<GenomeTuple,Tuple>   MinCount.h:8-179   (+ TupleOps.h:104-138)
//   a2  std::sort(readmm)                    MapRead.h:185      (GenomeTuple::operator< TupleOps.h:76)
//   a3  CompareLists<GenomeTuple,Tuple>      CompareLists.h:9-146
//   a4  SeparateMatchesByStrand              MapRead.h:109-150
// Parity status:
//   a3 PINNED  -- bit-exact against the reference template compiled from
//                 /root/reference/CompareLists.h (oracle/ref_harness/comparelists_ref.cpp),
//                 golden in tests/golden/comparelists_golden.json.
//   a2 PINNED  -- it IS libstdc++'s std::sort with the reference's comparator (the
//                 permutation of equal keys is an implementation property of libstdc++).
//   a1, a4 PARITY UNPINNED -- MinCount.h and MapRead.h include htslib/kseq.h (htslib is
//                 not in this image and is not vendored by the reference), so they
//                 cannot be compiled here; restated from the source text only.
#include "oracle_common.h"
#include <algorithm>
#include <vector>
#include <string.h>

namespace {
const uint64_t FOR_MASK = ~(1ULL << 63);  // lra.cpp:1008-1012 (InitStatic)
const uint64_t REV_MASK = (1ULL << 63);

struct GTup {  // TupleOps.h:68-89
  uint64_t t;
  uint32_t pos;
  bool operator<(const GTup& b) const { return (t & FOR_MASK) < (b.t & FOR_MASK); }
};
}  // namespace

// ---- a1 -------------------------------------------------------------------------------
// canonical (w,k)-minimizers of seq[0,seqLen); returns the number produced (only the first
// cap are written).  Mirrors the control flow of MinCount.h:8-179 including: the first
// window's UNMASKED comparison (:91), the ring-index (p % w) order of the re-scan (:148-154)
// and the N-window bookkeeping (:27-41,:109-132).
extern "C" long oracle_store_minimizers(const char* seq, uint32_t seqLen, int k, int w, uint64_t* keys,
                                        uint32_t* poss, long cap) {
  long n = 0;
  auto emit = [&](uint64_t t, uint32_t p) { if (n < cap) { keys[n] = t; poss[n] = p; } n++; };
  if (seqLen < (uint32_t)k) return 0;                                  // :12
  const int span = w + k - 1;                                          // :17
  if (seqLen < (uint32_t)span) return 0;                               // :26
  const uint64_t kmask = (k >= 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);  // InitMask TupleOps.h:95
  long nvStart = 0, nvEnd = 0;
  bool valid = false;
  auto find_valid = [&]() -> bool {                                    // :27-41 / :117-131
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
  uint64_t cur = 0, rc = 0;
  for (int p = 0; p < k; p++) cur = (cur << 2) + (uint64_t)oracle_code((unsigned char)seq[p]);  // StoreTuple :104
  {                                                                    // TupleRC :125-138
    uint64_t a = cur;
    for (int i = 0; i < k; i++) { rc = (rc << 2) + ((~a) & 3ULL); a >>= 2; }
  }
  auto canon = [&]() -> uint64_t {                                     // :60-61,:89-90,:144-145
    return ((cur & FOR_MASK) < (rc & FOR_MASK)) ? (cur & FOR_MASK) : (rc | REV_MASK);
  };
  auto shift = [&](uint32_t at) {                                      // ShiftOne/ShiftOneRC :114-123
    uint64_t c = (uint64_t)oracle_code((unsigned char)seq[at]);
    cur = ((cur << 2) & kmask) + c;
    rc = (rc >> 2) + (((~c) & 3ULL) << (2 * ((uint64_t)k - 1)));
  };
  std::vector<uint64_t> ringT(w);
  std::vector<uint32_t> ringP(w);
  uint64_t actT = canon();
  uint32_t actP = 0;
  ringT[0] = actT; ringP[0] = 0;
  uint32_t p;
  for (p = 1; p < (uint32_t)w && p < seqLen - k + 1; p++) {            // :77-96
    shift(p + k - 1);
    uint64_t c = canon();
    if (c < actT) { actT = c; actP = p; }                              // unmasked (:91)
    ringT[p % w] = c; ringP[p % w] = p;
  }
  if (nvEnd == span) emit(actT, actP);                                 // :100-102
  for (p = w; p < seqLen - k + 1; p++) {                               // :105-178
    if (nvEnd == (long)(p + k - 1)) {
      if (oracle_code_n((unsigned char)seq[p + k - 1]) <= 3) nvEnd++;
      else {
        nvStart = p + k;
        if (!find_valid()) return n;
        nvEnd = nvStart + span;
      }
    }
    shift(p + k - 1);
    uint64_t c = canon();
    ringT[p % w] = c; ringP[p % w] = p;
    if (p - w >= actP) {                                               // active left the window
      actT = ringT[0]; actP = ringP[0];
      for (int j = 1; j < w; j++)
        if ((ringT[j] & FOR_MASK) < (actT & FOR_MASK)) { actT = ringT[j]; actP = ringP[j]; }
      if (nvEnd == (long)(p + k)) emit(actT, actP);
    } else if ((c & FOR_MASK) < (actT & FOR_MASK)) {
      actT = c; actP = p;
      if (nvEnd == (long)(p + k)) emit(actT, actP);
    }
  }
  return n;
}

// ---- a2 -------------------------------------------------------------------------------
extern "C" void oracle_sort_minimizers(uint64_t* keys, uint32_t* poss, long n) {
  std::vector<GTup> v(n);
  for (long i = 0; i < n; i++) { v[i].t = keys[i]; v[i].pos = poss[i]; }
  std::sort(v.begin(), v.end());                                       // MapRead.h:185
  for (long i = 0; i < n; i++) { keys[i] = v[i].t; poss[i] = v[i].pos; }
}

// ---- a3 -------------------------------------------------------------------------------
// Two-ended galloping intersection.  Emits (query index, target index) pairs in the
// reference's discovery order.  mask = comparison mask (FOR_MASK for the global index).
extern "C" long oracle_compare_lists(const uint64_t* qk, const uint32_t* qp, long nq, const uint64_t* tk,
                                     const uint32_t* tp, long nt, long maxFreq, int64_t maxDiag, int64_t minDiag,
                                     uint32_t* out_qi, uint32_t* out_ti, long cap) {
  long n = 0;
  if (nq == 0 || nt == 0) return 0;                                    // :27-30
  const uint64_t M = FOR_MASK;
  auto Q = [&](long i) { return qk[i] & M; };
  auto T = [&](long i) { return tk[i] & M; };
  auto emit = [&](long qi, long ti) {                                  // :87-97 / :127-137
    if (maxDiag != 0 && minDiag != 0) {
      int64_t d = (int64_t)tp[ti] - (int64_t)qp[qi];
      if (!(d <= maxDiag && d >= minDiag)) return;
    }
    if (n < cap) { out_qi[n] = (uint32_t)qi; out_ti[n] = (uint32_t)ti; }
    n++;
  };
  long qs = 0, qe = nq - 1;                                            // qe inclusive (:23)
  long ts = 0, te = nt;                                                // te exclusive (:24)
  do {
    while (qs <= qe && Q(qs) < T(ts)) qs++;                            // :47-49
    if (qs >= qe) return n;                                            // :51-53
    uint64_t startGap = Q(qs) - T(ts);                                 // :55-57
    while (qe > qs && te > ts && Q(qe) > T(te - 1)) qe--;              // :63-65
    uint64_t endGap = T(te - 1) - Q(qe);                               // :67
    if (startGap == 0 || (startGap & M) > (endGap & M)) {              // :69 (operator> is masked)
      long tsOrig = ts, qsOrig = qs;
      // lower_bound over [ts,te) by masked key (:76)
      long lo = ts, hi = te;
      while (lo < hi) { long mid = lo + (hi - lo) / 2; if (T(mid) < Q(qs)) lo = mid + 1; else hi = mid; }
      ts = lo;
      if (ts < te && T(ts) == Q(qs)) {                                 // :78 (ts==te reads past the range in the reference; no effect)
        uint32_t tsStart = (uint32_t)ts, tsi = (uint32_t)ts;
        while ((long)tsi != te && Q(qs) == T(tsi)) tsi++;
        uint32_t qsStart = (uint32_t)qs;
        while (qs < qe && Q(qs + 1) == Q(qs)) qs++;
        for (uint32_t ti = tsStart; ti != tsi; ti++)
          if (qs - (long)qsStart < maxFreq)
            for (uint32_t qi = qsStart; (long)qi <= qs; qi++) emit(qi, ti);
      }
      while (ts < te && tk[ts] == tk[tsOrig]) ts++;                    // :101 raw compare
      while (qs < qe && qk[qs] == qk[qsOrig]) qs++;                    // :102 raw compare
    } else {
      if (te != nt && T(te - 1) == Q(qe)) {                            // :112-114
      } else {                                                         // upper_bound (:116-118)
        long lo = ts, hi = te;
        while (lo < hi) { long mid = lo + (hi - lo) / 2; if (!(Q(qe) < T(mid))) lo = mid + 1; else hi = mid; }
        te = lo;
      }
      uint32_t teStart = (uint32_t)te, tei = (uint32_t)te;
      while ((long)tei > ts && T(tei - 1) == Q(qe)) tei--;
      if (tei < teStart && teStart > 0) {
        uint32_t qeStart = (uint32_t)qe;
        while (qe > qs && Q(qe) == Q(qe - 1)) qe--;
        for (uint32_t ti = tei; ti < teStart; ti++)
          if ((long)qeStart - qe < maxFreq)
            for (uint32_t qi = (uint32_t)qe; qi <= qeStart; qi++) emit(qi, ti);
      }
      te = tei;
    }
  } while (qs < qe && ts < te);
  return n;
}

// ---- a4 -------------------------------------------------------------------------------
// strand[i] = 0 if the k read bytes at qpos equal the k genome bytes at tpos, else 1.
// (strncmp over k bytes; neither buffer holds NUL inside a k-mer.)
extern "C" long oracle_separate_strand(const char* read, const char* genome, int k, const uint32_t* qpos,
                                       const uint32_t* tpos, long n, uint8_t* strand) {
  long nf = 0;
  for (long i = 0; i < n; i++) {
    bool same = strncmp(read + qpos[i], genome + tpos[i], k) == 0;
    strand[i] = same ? 0 : 1;
    nf += same;
  }
  return nf;
}

// ---- test input generator: McIlroy's "antiquicksort" adversary run against libstdc++'s
// std::sort, yielding a key assignment that drives introsort to its depth limit (so the
// heap-sort fall-back is exercised).  vals_out[i] = key of element i.
namespace {
struct Adversary {
  std::vector<int> val; int gas, nsolid = 0, candidate = 0;
  bool less(int x, int y) {
    if (val[x] == gas && val[y] == gas) { if (x == candidate) val[x] = nsolid++; else val[y] = nsolid++; }
    if (val[x] == gas) candidate = x; else if (val[y] == gas) candidate = y;
    return val[x] < val[y];
  }
};
}
extern "C" void oracle_antiqsort(int n, uint64_t* vals_out) {
  Adversary a;
  a.gas = n;
  a.val.assign(n, n);
  std::vector<int> ptr(n);
  for (int i = 0; i < n; i++) ptr[i] = i;
  std::sort(ptr.begin(), ptr.end(), [&](int x, int y) { return a.less(x, y); });
  for (int i = 0; i < n; i++) vals_out[i] = (uint64_t)a.val[i];
}

// The k-mer stepping of oracle_store_minimizers on its own (StoreTuple + TupleRC for the first k-mer, ShiftOne / ShiftOneRC after, the canonical key of
// MinCount.h:60-61) and CreateRC (SeqUtils.h:151-158): PINNED to the reference's TupleOps.h / SeqUtils.h (ref_harness/tuple_ops_ref.cpp ->
// tests/golden/tuple_ops_golden.json).  out: 3 words per k-mer (forward, reverse complement, key); rc_out: the reverse complement of seq.
extern "C" long oracle_kmer_stream(const char* seq, long n, int k, uint64_t* out, char* rc_out) {
  static const char* RC = nullptr;
  static char table[256];
  if (!RC) {
    memset(table, 'N', 256);
    table[(int)'A'] = 'T'; table[(int)'C'] = 'G'; table[(int)'G'] = 'C'; table[(int)'T'] = 'A';
    table[(int)'a'] = 't'; table[(int)'c'] = 'g'; table[(int)'g'] = 'c'; table[(int)'t'] = 'a'; table[(int)'n'] = 'n';
    RC = table;
  }
  for (long i = 0; i < n; i++) rc_out[n - i - 1] = RC[(unsigned char)seq[i]];
  if (n < k) return 0;
  const uint64_t kmask = (k >= 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
  uint64_t cur = 0, rc = 0;
  for (int p = 0; p < k; p++) cur = (cur << 2) + (uint64_t)oracle_code((unsigned char)seq[p]);
  { uint64_t a = cur; for (int i = 0; i < k; i++) { rc = (rc << 2) + ((~a) & 3ULL); a >>= 2; } }
  long m = 0;
  for (long p = 0; p + k <= n; p++) {
    if (p > 0) {
      const uint64_t c = (uint64_t)oracle_code((unsigned char)seq[p + k - 1]);
      cur = ((cur << 2) & kmask) + c;
      rc = (rc >> 2) + (((~c) & 3ULL) << (2 * ((uint64_t)k - 1)));
    }
    out[3 * m] = cur; out[3 * m + 1] = rc; out[3 * m + 2] = ((cur & FOR_MASK) < (rc & FOR_MASK)) ? (cur & FOR_MASK) : (rc | REV_MASK);
    m++;
  }
  return m;
}
