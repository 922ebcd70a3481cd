// oracle/sdp.cpp -- TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// CPU restatement of the first sparse dynamic program of the low-accuracy path ("SDP#A"):
//   SparseDP(vector<Cluster>&, vector<UltimateChain>&, ...)      SparseDP.h:2139-2279
//   insertPointsPair                                               SparseDP.h:79-137
//   SortByRowOp / SortByColOp / Lower_Bound                        Sorting.h:226-319
//   GetRowInfo / GetColInfo                                        DivideSubByRow1.h:28-49, DivideSubByCol1.h:32-53
//   ScanPoints_* / Decide_Eb_Db_* / DivideSubProbBy{Row,Col}{1,2}  DivideSubBy{Row1,Col1,Row2,Col2}.h
//   InitPWL / PWL_w / w                                            SubRountine.h:43-121
//   UPPERbound / FindValueInBlock / FindBoundary / Maximization    SubRountine.h:205-458
//   PassValueToD1 / PassValueToD2                                  SparseDP.h:140-310
//   ProcessPoint<Cluster> (the overload SDP#A selects)             SparseDP.h:1015-1171
//   TraceBack (with `used`)                                        SparseDP.h:1351-1438
//   DecidePrimaryChains (pure-match version)                       SparseDP.h:1658-1760
//   UltimateChain::OverlapsOnT                                     Chain.h:261-276
//
// The four decompositions differ only in a handful of switches, restated here as one routine with a
// family descriptor:
//   family  points in      diagonal   inv  arrays      lower bound        halves
//   R1      row order H1   t - q      1    ascending   forward            Di <- [start,med) ends,  Ei <- [med,end) starts
//   C1      col order H2   t - q      1    descending  reverse, then --t  same
//   R2      row order H1   t + q      0    descending  reverse, then --t  same
//   C2      col order H2   t + q      0    ascending   forward            Ei <- [start,med) starts, Di <- [med,end) ends; second half recursed first
//
// Parity status: COMPONENTS PINNED, GLUE UNPINNED.  SparseDP.h itself includes Clustering.h -> Genome.h ->
// htslib and cannot be compiled here.  The pieces that compile from the reference's own headers are
// compiled in place by oracle/ref_harness/sdp_parts_ref.cpp and this file is checked against them
// (tests/golden/sdp_parts_golden.json): the point sorts, row/column tables, all four decompositions
// (every Di/Ei/Db/Eb array and every SS_A/SS_B list), InitPWL/PWL_w/w, and Maximization +
// FindValueInBlock driven with random value updates.  ProcessPoint, PassValueToD*, TraceBack and
// DecidePrimaryChains are restated from the source text.
#include "oracle_common.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <set>
#include <string>
#include <utility>
#include <vector>

#ifndef SDP_STAT
#define SDP_STAT(...)                      // tools/sdp_case_stats.cpp defines it to count the steps of Maximization / FindValueInBlock per level
#endif

namespace {

long g_off_boundary = 0;      // pushes of a stack pair whose boundary is not Ei.size() (see maximization)

typedef std::pair<long, long> LPair;

struct Pt {                 // Point.h:9-22
  uint32_t q, t;            // se.first, se.second
  bool orient, ind, inv;
  uint32_t frag;
  int cluster;
};

struct Info {               // Info.h:14-30
  uint32_t pstart, pend, rc;
  std::vector<uint32_t> A[2], B[2];   // [0] = SS_A1/SS_B1, [1] = SS_A2/SS_B2
};

struct Sub {                // SubProblem.h:15-37
  uint32_t num = 0, now = 0;
  long last = -1;
  std::vector<long> Di, Ei, Eb, Db;
  std::vector<float> Dv, Ev;
  std::vector<uint32_t> Dp, Ep;
  std::vector<LPair> Block, S;        // S = S_1, a stack (back() is top())
};

struct Family { bool col, back, desc, swapped; int inv; };
const Family FAM[4] = {
    {false, false, false, false, 1},   // R1
    {true, false, true, false, 1},     // C1
    {false, true, true, false, 0},     // R2
    {true, true, false, true, 0},      // C2
};

// ---- PWL gap cost -------------------------------------------------------------------------------------
struct Pwl {
  enum { NUMPWL = 25 };
  long STOPS[NUMPWL];
  float INTER[NUMPWL], SLOPE[NUMPWL];
  int c1, c2;
  void init(float intercept, float scalar, float root, int g1, int g2) {   // SubRountine.h:43-99
    static const long st[NUMPWL] = {0,    5,    10,   20,   40,   80,    100,   200,   300,   500,   1000,  2000, 3000,
                                    4000, 5000, 6000, 7000, 8000, 9000, 15000, 20000, 30000, 40000, 50000, 100000};
    c1 = g1; c2 = g2;
    float vals[NUMPWL];
    for (int i = 0; i < NUMPWL; i++) { STOPS[i] = st[i]; INTER[i] = 0; SLOPE[i] = 0; }
    vals[0] = 0;
    for (int i = 1; i < NUMPWL; i++) {
      if (i <= 2) intercept = 0;                         // :83 -- the parameter stays 0 from here on
      vals[i] = intercept + scalar * std::pow((float)STOPS[i], 1 / root);
    }
    for (int i = 0; i < NUMPWL - 1; i++) {
      float slope = (vals[i + 1] - vals[i]) / (STOPS[i + 1] - STOPS[i]);
      if (STOPS[i] <= 10) { SLOPE[i] = 0; INTER[i] = 0; }
      else { SLOPE[i] = slope; INTER[i] = vals[i] - STOPS[i] * slope + intercept; }
    }
  }
  float pwl(long x) const {                              // PWL_w :101-121 (minX forced to 2)
    long penalty;
    if (x <= 2) penalty = 0;
    else {
      int bound = (int)(std::upper_bound(&STOPS[0], &STOPS[NUMPWL - 1], x) - &STOPS[0]);
      penalty = (long)(SLOPE[bound - 1] * x + INTER[bound - 1]);
      if (penalty >= c1 && penalty < c2) penalty = c1;
      else if (penalty > c2) penalty = c2;
    }
    return (float)penalty;
  }
  float w(long i, long j) const {                        // w :123-129 (everything after the first return is dead)
    long x = std::labs(j - i) + 1;
    if (x == 1) return 0;
    return -pwl(x);
  }
};

// ---- Lower_Bound (Sorting.h:303-319) over the index sequence 0..n-1, forward or reversed ----------------
// returns the POSITION in the traversed sequence; pos2idx converts.
inline size_t lower_pos(const std::vector<long>& arr, long val, bool rev) {
  size_t n = arr.size(), first = 0, count = n;
  while (count > 0) {
    size_t step = count / 2, it = first + step;
    size_t idx = rev ? n - 1 - it : it;
    if (arr[idx] < val) { first = it + 1; count -= step + 1; }
    else count = step;
  }
  return first;
}
inline size_t pos2idx(size_t n, size_t pos, bool rev) { return rev ? n - 1 - pos : pos; }

void decide_eb_db(Sub& s, bool desc) {                  // Decide_Eb_Db_{R1,C1,R2,C2}
  size_t h = s.Ei.size();
  for (size_t d = 0; d < s.Di.size(); d++) {
    size_t p = lower_pos(s.Ei, s.Di[d], desc);
    size_t idx;
    if (!desc) { if (p == h) break; idx = p; }
    else { if (p == 0) break; idx = pos2idx(h, p - 1, true); }
    s.Db[d] = (long)idx;
    s.Eb[idx] = (long)d;
  }
  unsigned int cur = (unsigned int)-1;
  for (size_t e = 0; e < s.Eb.size(); e++) {
    if (s.Eb[e] == -1 && cur == (unsigned int)-1) continue;
    else if (s.Eb[e] != -1) cur = (unsigned int)s.Eb[e];
    else s.Eb[e] = cur;
  }
}

struct Ctx {
  std::vector<Pt> H1;
  std::vector<uint32_t> H2;
  std::vector<Info> Row, Col;
  std::vector<Sub> subs[4];
  Pwl pwl;
};

inline const Pt& pt_at(const Ctx& c, const Family& f, uint32_t j) { return f.col ? c.H1[c.H2[j]] : c.H1[j]; }
inline long diag_of(const Pt& p, bool back) { return back ? (long)p.t + (long)p.q : (long)p.t - (long)p.q; }

// ScanPoints_*(…, Bi, s, e, DE, n): one of Di / Ei from the points of rows/cols [s,e)
void scan_one(Ctx& c, int fam, std::vector<long>& out, uint32_t s, uint32_t e, bool DE, uint32_t n) {
  const Family& f = FAM[fam];
  std::vector<Info>& V = f.col ? c.Col : c.Row;
  std::set<long> idx;
  for (uint32_t i = s; i < e; i++) {
    unsigned count = 0;
    for (uint32_t j = V[i].pstart; j < V[i].pend; j++) {
      const Pt& p = pt_at(c, f, j);
      if (p.ind == DE && (int)p.inv == f.inv) { idx.insert(diag_of(p, f.back)); ++count; }
    }
    if (count != 0) { if (DE) V[i].B[f.back].push_back(n); else V[i].A[f.back].push_back(n); }
  }
  if (!f.desc) for (auto it = idx.begin(); it != idx.end(); ++it) out.push_back(*it);
  else for (auto it = idx.rbegin(); it != idx.rend(); ++it) out.push_back(*it);
}
// leaf version: Ei and Di from the same row/col
void scan_leaf(Ctx& c, int fam, std::vector<long>& Ei, std::vector<long>& Di, uint32_t s, uint32_t e, uint32_t n) {
  const Family& f = FAM[fam];
  std::vector<Info>& V = f.col ? c.Col : c.Row;
  std::set<long> i1, i2;
  for (uint32_t i = s; i < e; i++) {
    unsigned c1 = 0, c2 = 0;
    for (uint32_t j = V[i].pstart; j < V[i].pend; j++) {
      const Pt& p = pt_at(c, f, j);
      if (p.ind == 1 && (int)p.inv == f.inv) { i1.insert(diag_of(p, f.back)); ++c1; }
      else if (p.ind == 0 && (int)p.inv == f.inv) { i2.insert(diag_of(p, f.back)); ++c2; }
    }
    if (c1 != 0 && c2 != 0) { V[i].B[f.back].push_back(n); V[i].A[f.back].push_back(n); }
  }
  if (!f.desc) { for (long v : i1) Ei.push_back(v); for (long v : i2) Di.push_back(v); }
  else { for (auto it = i1.rbegin(); it != i1.rend(); ++it) Ei.push_back(*it); for (auto it = i2.rbegin(); it != i2.rend(); ++it) Di.push_back(*it); }
}

void finish_sub(Sub& s, bool desc) {                    // the "non-leaf case" block shared by all four files
  size_t l = s.Di.size(), h = s.Ei.size();
  s.Eb.assign(h, -1); s.Db.assign(l, -1);
  decide_eb_db(s, desc);
  s.Dv.assign(l, 0); s.Dp.assign(l, 0); s.Ev.assign(h, 0); s.Ep.assign(h, 0);
  s.S.push_back(LPair(-1, (long)h + 1));
}

void divide(Ctx& c, int fam, uint32_t start, uint32_t end, uint32_t& n) {
  const Family& f = FAM[fam];
  std::vector<Sub>& S = c.subs[fam];
  S.emplace_back(); S.back().num = n;
  size_t me = S.size() - 1;
  if (end == start + 1) {
    scan_leaf(c, fam, S[me].Ei, S[me].Di, start, end, n);
    if (!S[me].Ei.empty() && !S[me].Di.empty()) finish_sub(S[me], f.desc);
    else { S.pop_back(); --n; }
    return;
  }
  uint32_t med = (start + end) / 2;
  if (!f.swapped) { scan_one(c, fam, S[me].Di, start, med, false, n); scan_one(c, fam, S[me].Ei, med, end, true, n); }
  else { scan_one(c, fam, S[me].Ei, start, med, true, n); scan_one(c, fam, S[me].Di, med, end, false, n); }
  bool eE = S[me].Ei.empty(), dE = S[me].Di.empty();
  // halves: "D side" = where Di came from, "E side" = where Ei came from
  uint32_t dS = f.swapped ? med : start, dEnd = f.swapped ? end : med;
  uint32_t eS = f.swapped ? start : med, eEnd = f.swapped ? med : end;
  if (eE && dE) { S.pop_back(); --n; }
  else if (eE && !dE) { ++n; divide(c, fam, dS, dEnd, n); }
  else if (!eE && dE) { ++n; divide(c, fam, eS, eEnd, n); }
  else {
    finish_sub(S[me], f.desc);
    ++n; divide(c, fam, dS, dEnd, n);       // R1/C1/R2: first half; C2: second half
    ++n; divide(c, fam, eS, eEnd, n);
  }
}

// ---- maximisation structure ---------------------------------------------------------------------------
size_t upper_block(const std::vector<LPair>& B, unsigned int val) {     // UPPERbound :205-221
  size_t first = 0, count = B.size();
  while (count > 0) {
    size_t step = count / 2, it = first + step;
    if ((long)val >= B[it].second) { first = it + 1; count -= step + 1; }
    else count = step;
  }
  return first;
}

// returns false on the reference's undefined behaviour (dereferencing Block.end() / index -1)
bool find_value_in_block(const Sub& s, unsigned int i1, unsigned int& i2) {   // FindValueInBlock :224-236
  if (s.Block.empty() || s.S.empty()) return false;
  if ((long)i1 >= s.Block.back().second && (long)i1 < s.S.back().second) { i2 = (unsigned int)s.S.back().first; SDP_STAT(fast, s, 1); }
  else {
    SDP_STAT(bsearch, s, (long)s.Block.size());
    size_t it = upper_block(s.Block, i1);
    if (it == s.Block.size()) return false;
    i2 = (unsigned int)s.Block[it].first;
  }
  return i2 < s.Di.size();
}

unsigned int find_boundary(unsigned int first, unsigned int last, unsigned int a, unsigned int b, const Sub& s, const Pwl& P) {   // :239-263
  if (b != (unsigned int)-1) {
    unsigned int count = last - first;
    while (count > 0) {
      unsigned int step = count / 2, it = first + step;
      if (s.Dv[a] + P.w(s.Di[a], s.Ei[it]) > s.Dv[b] + P.w(s.Di[b], s.Ei[it])) { first = it + 1; count -= step + 1; }
      else count = step;
    }
  } else first = (unsigned int)s.Ei.size();
  return first;
}

void maximization(Sub& s, const Pwl& P) {               // Maximization :270-345 (now/last are the sub's members)
  unsigned int m = (unsigned int)s.Di.size(), n = (unsigned int)s.Ei.size();
  unsigned int now = s.now;
  SDP_STAT(query, s, (long)now - s.last);
  for (unsigned int i = (unsigned int)(s.last + 1); i <= now; ++i) {
    SDP_STAT(iter, s, 1);
    if (s.Db[i] == -1) break;
    if (s.S.back().second == (long)n + 1) {
      s.Block.push_back(LPair(-1, s.Db[i]));
      s.S.push_back(LPair(i, n));
    }
    while (s.Db[i] >= s.S.back().second) { s.Block.push_back(s.S.back()); s.S.pop_back(); SDP_STAT(pop1, s, 1); }
    long l = s.S.back().first;
    if (s.Dv[i] + P.w(s.Di[i], s.Ei[s.Db[i]]) > s.Dv[l] + P.w(s.Di[l], s.Ei[s.Db[i]])) {
      if (s.Db[i] < s.S.back().second && !s.Block.empty() && s.Db[i] > s.Block.back().second) s.Block.push_back(LPair(s.S.back().first, s.Db[i]));
      SDP_STAT(win, s, 1);
      LPair cur = s.S.back(), prev = s.S.back();
      while (!s.S.empty() && s.Dv[i] + P.w(s.Di[i], s.Ei[cur.second - 1]) > s.Dv[cur.first] + P.w(s.Di[cur.first], s.Ei[cur.second - 1])) {
        s.S.pop_back();
        SDP_STAT(pop2, s, 1);
        prev = cur;
        cur = s.S.back();
        if (cur.second == (long)n + 1) break;
      }
      SDP_STAT(fb, s, cur.first == -1 ? 0 : (long)cur.second - (long)prev.second);
      unsigned int h = find_boundary((unsigned int)prev.second, (unsigned int)cur.second, i, (unsigned int)cur.first, s, P);
      if (h != n) g_off_boundary++;                        // (never: every pair but the dummy is (i, n) -- what the HIP kernels are built on; tests/test_sdp.py checks the counter)
      s.S.push_back(LPair(i, h));
    }
  }
  if (now == m - 1) {
    while (s.S.back().second != (long)n + 1) { s.Block.push_back(s.S.back()); s.S.pop_back(); SDP_STAT(pop3, s, 1); }
  } else {
    while (s.Db[now + 1] >= s.S.back().second) { s.Block.push_back(s.S.back()); s.S.pop_back(); SDP_STAT(pop3, s, 1); }
  }
  s.last = now;
}

// ---- per-fragment state (Fragment_Info.h:8-46) ----------------------------------------------------------
struct Frag {
  float val = 0;
  int cluster = 0;
  long prev_sub = -1, prev_ind = -1;
  bool prev = 1, inv = 1, orient = 1;
  std::vector<uint32_t> A[4], B[4];        // SS_A_/SS_B_ for R1, C1, R2, C2
};

void insert_pair(std::vector<Pt>& H1, uint32_t frag, uint32_t qs, uint32_t ts, int len, int cluster, int pair, int strand) {   // :79-137
  Pt s, e;
  s.frag = e.frag = frag; s.cluster = e.cluster = cluster; s.orient = e.orient = strand;
  s.ind = 1; e.ind = 0;
  if (pair == 0) { s.inv = e.inv = 1; s.q = qs; s.t = ts; e.q = qs + len; e.t = ts + len; }
  else { s.inv = e.inv = 0; s.q = qs; s.t = ts + len; e.q = qs + len; e.t = ts; }
  H1.push_back(s); H1.push_back(e);
}

struct RowLess {                                         // SortByRowOp Sorting.h:226-239
  bool operator()(const Pt& a, const Pt& b) const {
    if (a.q != b.q) return a.q < b.q;
    else if (a.t != b.t) return a.t < b.t;
    else return a.ind < b.ind;
  }
};
struct ColLess {                                         // SortByColOp :241-257
  const std::vector<Pt>* H;
  bool operator()(uint32_t a, uint32_t b) const {
    const Pt &x = (*H)[a], &y = (*H)[b];
    if (x.t != y.t) return x.t < y.t;
    else if (x.q != y.q) return x.q < y.q;
    else return x.ind < y.ind;
  }
};

void row_info(const std::vector<Pt>& H1, std::vector<Info>& M) {          // GetRowInfo
  uint32_t row = H1[0].q, ps = 0, pe = 1;
  for (uint32_t i = 0; i < H1.size(); ++i) {
    if (row == H1[i].q) pe = i + 1;
    else { Info p; p.pstart = ps; p.pend = pe; p.rc = row; M.push_back(p); ps = i; pe = i + 1; row = H1[i].q; }
    if (i == H1.size() - 1) { Info p; p.pstart = ps; p.pend = pe; p.rc = row; M.push_back(p); }
  }
}
void col_info(const std::vector<Pt>& H1, const std::vector<uint32_t>& H2, std::vector<Info>& M) {   // GetColInfo
  uint32_t col = H1[H2[0]].t, ps = 0, pe = 1;
  for (uint32_t i = 0; i < H2.size(); ++i) {
    if (col == H1[H2[i]].t) pe = i + 1;
    else { Info p; p.pstart = ps; p.pend = pe; p.rc = col; M.push_back(p); ps = i; pe = i + 1; col = H1[H2[i]].t; }
    if (i == H2.size() - 1) { Info p; p.pstart = ps; p.pend = pe; p.rc = col; M.push_back(p); }
  }
}

// sorts + tables + the four decompositions (SparseDP.h:2171-2193)
void build(Ctx& c) {
  std::sort(c.H1.begin(), c.H1.end(), RowLess());
  c.H2.resize(c.H1.size());
  std::iota(c.H2.begin(), c.H2.end(), 0);
  ColLess cl; cl.H = &c.H1;
  std::sort(c.H2.begin(), c.H2.end(), cl);
  row_info(c.H1, c.Row);
  col_info(c.H1, c.H2, c.Col);
  for (int fam = 0; fam < 4; fam++) {
    uint32_t n = 0;
    divide(c, fam, 0, (uint32_t)(FAM[fam].col ? c.Col.size() : c.Row.size()), n);
  }
}

void dump_vec(std::string& o, const char* name, const std::vector<long>& v) {
  o += name; o += ':';
  char b[32];
  for (long x : v) { snprintf(b, sizeof b, "%ld,", x); o += b; }
  o += ';';
}
void dump_vec(std::string& o, const char* name, const std::vector<uint32_t>& v) {
  o += name; o += ':';
  char b[32];
  for (uint32_t x : v) { snprintf(b, sizeof b, "%u,", x); o += b; }
  o += ';';
}

uint64_t fnv1a(const std::string& s) {
  uint64_t h = 1469598103934665603ULL;
  for (unsigned char ch : s) { h ^= ch; h *= 1099511628211ULL; }
  return h;
}

}  // namespace

extern "C" long oracle_sdp_off_boundary_pushes(int reset) { const long v = g_off_boundary; if (reset) g_off_boundary = 0; return v; }

// ---- test hooks for the component pinning (same canonical text as oracle/ref_harness/sdp_parts_ref.cpp) ----
// points: q,t,ind,inv (frag/cluster/orient play no part in sorting or decomposition; frag is used to make the
// H1 permutation visible).  Returns the length of the canonical text; writes FNV-1a of it to *hash and, if buf
// is non-null, up to cap bytes of the text.
extern "C" long oracle_sdp_divide_dump(long nPts, const uint32_t* q, const uint32_t* t, const uint8_t* ind, const uint8_t* inv,
                                       uint64_t* hash, char* buf, long cap) {
  Ctx c;
  for (long i = 0; i < nPts; i++) {
    Pt p; p.q = q[i]; p.t = t[i]; p.ind = ind[i]; p.inv = inv[i]; p.orient = 0; p.frag = (uint32_t)i; p.cluster = 0;
    c.H1.push_back(p);
  }
  std::string o;
  if (nPts > 0) {
    build(c);
    std::vector<uint32_t> perm;
    for (auto& p : c.H1) perm.push_back(p.frag);
    dump_vec(o, "H1", perm);
    dump_vec(o, "H2", c.H2);
    o += '\n';
    for (int rc = 0; rc < 2; rc++) {
      std::vector<Info>& V = rc ? c.Col : c.Row;
      for (size_t i = 0; i < V.size(); i++) {
        char b[96];
        snprintf(b, sizeof b, "%c%zu:%u,%u,%u;", rc ? 'C' : 'R', i, V[i].pstart, V[i].pend, V[i].rc);
        o += b;
        dump_vec(o, "A1", V[i].A[0]); dump_vec(o, "B1", V[i].B[0]); dump_vec(o, "A2", V[i].A[1]); dump_vec(o, "B2", V[i].B[1]);
        o += '\n';
      }
    }
    for (int fam = 0; fam < 4; fam++) {
      for (size_t i = 0; i < c.subs[fam].size(); i++) {
        Sub& s = c.subs[fam][i];
        char b[64];
        snprintf(b, sizeof b, "F%d.%zu:num=%u;", fam, i, s.num);
        o += b;
        dump_vec(o, "Di", s.Di); dump_vec(o, "Ei", s.Ei); dump_vec(o, "Db", s.Db); dump_vec(o, "Eb", s.Eb);
        snprintf(b, sizeof b, "S=%zu;", s.S.size());
        o += b;
        o += '\n';
      }
    }
  }
  if (hash) *hash = fnv1a(o);
  if (buf && cap > 0) { long m = std::min<long>(cap - 1, (long)o.size()); memcpy(buf, o.data(), m); buf[m] = 0; }
  return (long)o.size();
}

// PWL_w(x) and w(0, x - 1) for each x after InitPWL(intercept, scalar, root, g1, g2); also the tables.
extern "C" void oracle_sdp_pwl(float intercept, float scalar, float root, int g1, int g2, long n, const long* x, float* pwlOut, float* wOut,
                               float* slope25, float* inter25) {
  Pwl P; P.init(intercept, scalar, root, g1, g2);
  for (long i = 0; i < n; i++) { pwlOut[i] = P.pwl(x[i]); wOut[i] = P.w(0, x[i] - 1); }
  if (slope25) memcpy(slope25, P.SLOPE, sizeof P.SLOPE);
  if (inter25) memcpy(inter25, P.INTER, sizeof P.INTER);
}

// Maximization driver for pinning: one ascending sub-problem (Di, Ei ascending, Db/Eb by Decide_Eb_Db_R1), a script of
// operations  op[k] = 0: Dv[a[k]] = max(Dv, v[k]) (PassValueToD1's update)   op[k] = 1: query start index a[k] (skipped if
// Eb == -1): now = Eb, Maximization, FindValueInBlock -> i2 appended to out (or -2 on undefined behaviour).
// Returns the number of outputs; the final Block is appended to blockOut as pairs.
extern "C" long oracle_sdp_maximization_script(long nD, const long* Di, long nE, const long* Ei, long nOps, const int* op, const long* a,
                                               const float* v, float intercept, float scalar, float root, int g1, int g2, long* out,
                                               long* blockOut, long* nBlockOut) {
  Pwl P; P.init(intercept, scalar, root, g1, g2);
  Sub s;
  s.Di.assign(Di, Di + nD); s.Ei.assign(Ei, Ei + nE);
  finish_sub(s, false);
  long no = 0;
  for (long k = 0; k < nOps; k++) {
    if (op[k] == 0) { if (s.Dv[a[k]] < v[k]) { s.Dv[a[k]] = v[k]; s.Dp[a[k]] = (uint32_t)k; } }
    else {
      long i1 = a[k];
      if (s.Eb[i1] == -1) { out[no++] = -1; continue; }
      s.now = (uint32_t)s.Eb[i1];
      maximization(s, P);
      unsigned int i2;
      if (!find_value_in_block(s, (unsigned int)i1, i2)) out[no++] = -2;
      else out[no++] = i2;
    }
  }
  long nb = 0;
  for (auto& b : s.Block) { blockOut[2 * nb] = b.first; blockOut[2 * nb + 1] = b.second; nb++; }
  *nBlockOut = nb;
  return no;
}

struct oracle_sdp_opts {
  float rate;          // match_rate (Map_lowacc.h:185-186)
  int NumAln;          // Options::NumAln
  float alnthres;      // Options::alnthres
  int readLen;         // read.length
  float gapopen, gapextend, gaproot;   // InitPWL arguments (lra.cpp:648)
  int gapCeiling1, gapCeiling2;
  int mode;            // 0: SDP#A (:2139);  1: the single-cluster SparseDP (:2287-2438): one point pair per anchor, first maximum, plain TraceBack
                       // 2: the high-accuracy SparseDP over split-cluster boxes (:1956-2135), through oracle_sdp_chain_boxes only
  int globalK;         // Options::globalK (mode 2: the value threshold of DecidePrimaryChains :1592)
};

// SDP#A over the extended clusters of one read.  Fragments are the clusters' matches, concatenated in cluster order
// (global fragment index = MatchStart[cluster] + i, SparseDP.h:2144-2150).
// Outputs: per fragment val / prev_sub / prev_ind / flags (bit0 prev, bit1 inv); chains as CSR (chainOff[nChains+1]) of global
// fragment indices in trace-back order (last anchor first) with link bits (chainLink[chainOff[c] + s], s < len-1), boxes
// (QStart,QEnd,TStart,TEnd) and FirstSDPValue.  Returns the number of chains, or -1 on undefined behaviour in the reference.
static int sdp_chain_impl(int nClusters, const int* clusterOff, const uint8_t* clusterStrand, const uint32_t* q, const uint32_t* t,
                          const int* len, const oracle_sdp_opts* o, float* fragVal, long* fragPrevSub, long* fragPrevInd,
                          uint8_t* fragFlags, int maxChains, int* chainOff, uint32_t* chainFrag, uint8_t* chainLink, uint32_t* chainBox,
                          float* chainValue, const uint32_t* qe, const uint32_t* te, const int* numAnchors, int* chainNumAnchors) {
  int total = nClusters > 0 ? (o->mode == 2 ? nClusters : clusterOff[nClusters]) : 0;
  chainOff[0] = 0;
  if (total == 0) return 0;
  if (const char* dump = getenv("ORACLE_SDP_DUMP")) {                      // analysis hook (tools/sdp_case_stats.py): the inputs of every call, appended
    if (o->mode != 2) {
      FILE* f = fopen(dump, "ab");
      if (f) {
        int hdr[4] = {o->mode, nClusters, total, o->readLen};
        fwrite(hdr, 4, 4, f); fwrite(&o->rate, 4, 1, f);
        fwrite(clusterOff, 4, nClusters + 1, f); fwrite(clusterStrand, 1, nClusters, f);
        fwrite(q, 4, total, f); fwrite(t, 4, total, f); fwrite(len, 4, total, f);
        fclose(f);
      }
    }
  }
  Ctx c;
  c.pwl.init(o->gapopen, o->gapextend, o->gaproot, o->gapCeiling1, o->gapCeiling2);
  if (o->mode == 2) {                                                       // :1959-2018: four points per box, s1 e1 s2 e2
    for (int i = 0; i < total; i++) {
      const int orient = clusterStrand[i] == 0 ? 1 : 0;
      Pt p;
      p.frag = i; p.cluster = i; p.orient = orient;
      p.ind = 1; p.inv = 1; p.q = q[i] + 1; p.t = t[i] + 1; c.H1.push_back(p);
      p.ind = 0; p.inv = 1; p.q = qe[i] - 1; p.t = te[i] - 1; c.H1.push_back(p);
      p.ind = 1; p.inv = 0; p.q = q[i] + 1; p.t = te[i] - 1; c.H1.push_back(p);
      p.ind = 0; p.inv = 0; p.q = qe[i] - 1; p.t = t[i] + 1; c.H1.push_back(p);
    }
  }
  for (int cm = 0; o->mode != 2 && cm < nClusters; cm++) {                  // :2152-2169
    int ms = clusterOff[cm], sz = clusterOff[cm + 1] - ms;
    for (int i = 0; i < sz; i++) {
      int g = ms + i;
      bool edge = o->mode == 0 && (i == 0 || i == sz - 1);
      if (clusterStrand[cm] == 0) {
        insert_pair(c.H1, g, q[g], t[g], len[g], cm, 0, 1);
        if (edge) insert_pair(c.H1, g, q[g], t[g], len[g], cm, 1, 1);
      } else {
        insert_pair(c.H1, g, q[g], t[g], len[g], cm, 1, 0);
        if (edge) insert_pair(c.H1, g, q[g], t[g], len[g], cm, 0, 0);
      }
    }
  }
  build(c);
  std::vector<Frag> V(total);
  // SS lists per fragment (:2198-2262); family index: 0 R1, 1 C1, 2 R2, 3 C2
  for (int rc = 0; rc < 2; rc++) {
    std::vector<Info>& T = rc ? c.Col : c.Row;
    for (size_t ti = 0; ti < T.size(); ti++)
      for (uint32_t tt = T[ti].pstart; tt < T[ti].pend; tt++) {
        const Pt& p = rc ? c.H1[c.H2[tt]] : c.H1[tt];
        Frag& f = V[p.frag];
        int fam = (p.inv ? 0 : 2) + rc;
        int k = p.inv ? 0 : 1;
        if (p.ind == 1) {
          f.B[fam] = T[ti].B[k];
          f.val = len[p.frag] * o->rate;
          f.cluster = p.cluster;
          f.orient = p.orient;
        } else f.A[fam] = T[ti].A[k];
      }
  }
  // ProcessPoint :1015-1171
  for (size_t i = 0; i < c.H1.size(); i++) {
    const Pt& p = c.H1[i];
    long fd = (long)p.t - (long)p.q, bd = (long)p.t + (long)p.q;
    uint32_t ii = p.frag;
    Frag& F = V[ii];
    for (int rc = 0; rc < 2; rc++) {
      int fam = (p.inv ? 0 : 2) + rc;
      const Family& fm = FAM[fam];
      long dg = fm.back ? bd : fd;
      std::vector<Sub>& S = c.subs[fam];
      if (p.ind == 1) {
        const std::vector<uint32_t>& L = F.B[fam];
        for (size_t k = 0; k < L.size(); k++) {
          Sub& s = S[L[L.size() - 1 - k]];
          if (s.Di.empty()) continue;
          size_t pos = lower_pos(s.Ei, dg, fm.desc);
          if (pos >= s.Ei.size()) return -1;
          unsigned int i1 = (unsigned int)pos2idx(s.Ei.size(), pos, fm.desc);
          if (s.Eb[i1] == -1) continue;
          s.now = (uint32_t)s.Eb[i1];
          maximization(s, c.pwl);
          s.last = s.Eb[i1];
          unsigned int i2;
          if (!find_value_in_block(s, i1, i2)) return -1;
          s.Ev[i1] = s.Dv[i2] + c.pwl.w(s.Di[i2], s.Ei[i1]) + o->rate * len[ii];
          s.Ep[i1] = i2;
          if (F.val < s.Ev[i1]) {
            F.val = s.Ev[i1];
            F.prev_sub = s.num;
            F.prev_ind = i1;
            F.prev = (rc == 0);
            F.inv = fm.inv;
          }
        }
      } else {                                                         // PassValueToD1/D2 :140-310
        const std::vector<uint32_t>& L = F.A[fam];
        for (size_t k = 0; k < L.size(); k++) {
          Sub& s = S[L[L.size() - 1 - k]];
          if (s.Ei.empty()) continue;
          size_t pos = lower_pos(s.Di, dg, fm.desc);
          if (pos >= s.Di.size()) return -1;
          size_t d = pos2idx(s.Di.size(), pos, fm.desc);
          if (s.Dv[d] < F.val) { s.Dv[d] = F.val; s.Dp[d] = ii; }
        }
      }
    }
  }
  for (int i = 0; i < total; i++) {
    if (fragVal) fragVal[i] = V[i].val;
    if (fragPrevSub) fragPrevSub[i] = V[i].prev_sub;
    if (fragPrevInd) fragPrevInd[i] = V[i].prev_ind;
    if (fragFlags) fragFlags[i] = (uint8_t)((V[i].prev ? 1 : 0) | (V[i].inv ? 2 : 0));
  }
  if (o->mode == 1) {                                                     // :2417-2434
    float max_value = 0; unsigned int max_pos = 0;
    for (int l = 0; l < total; l++) if (V[l].val > max_value) { max_value = V[l].val; max_pos = l; }
    unsigned int i = max_pos;                                             // TraceBack :1521-1565
    long ps = V[i].prev_sub, pi = V[i].prev_ind;
    int n = 0;
    chainFrag[n++] = i;
    while (ps != -1 && pi != -1 && n < total) {
      int fam = (V[i].inv ? 0 : 2) + (V[i].prev ? 0 : 1);
      Sub& s = c.subs[fam][ps];
      chainLink[n - 1] = V[i].inv ? 0 : 1;
      i = s.Dp[s.Ep[pi]];
      ps = V[i].prev_sub; pi = V[i].prev_ind;
      chainFrag[n++] = i;
    }
    chainLink[n - 1] = 0;
    chainOff[1] = n; chainValue[0] = max_value;
    chainBox[0] = chainBox[1] = chainBox[2] = chainBox[3] = 0;
    return 1;
  }
  // DecidePrimaryChains :1658-1760 (modes 0) / :1587-1655 (mode 2)
  std::vector<int> order(total);
  std::iota(order.begin(), order.end(), 0);
  {
    std::vector<float> fv(total);
    for (int i = 0; i < total; i++) fv[i] = V[i].val;
    std::sort(order.begin(), order.end(), [&fv](int a, int b) { return fv[a] > fv[b]; });   // Fragment_valueOrder::Sort
  }
  std::vector<bool> used(total, 0);
  if (o->mode == 2) {
    const float best = V[order[0]].val;
    const float value_thres = std::max(o->alnthres * best, best - 130 * o->globalK);
    int nChains = 0, fv = 0;
    while (fv < total && V[order[fv]].val >= value_thres) {
      unsigned int i = order[fv];
      std::vector<unsigned int> chain;
      std::vector<bool> link;
      long ps = V[i].prev_sub, pi = V[i].prev_ind;                        // TraceBack with `used` :1351-1438
      if (used[i] == 0) {
        chain.push_back(i); used[i] = 1;
        auto abandon = [&]() { for (unsigned int x : chain) used[x] = 0; chain.clear(); link.clear(); };
        while (ps != -1 && pi != -1) {
          int fam = (V[i].inv ? 0 : 2) + (V[i].prev ? 0 : 1);
          Sub& s = c.subs[fam][ps];
          unsigned int nx = s.Dp[s.Ep[pi]];
          if (used[nx] == 0) { link.push_back(V[i].inv ? 0 : 1); i = nx; }
          else { abandon(); break; }
          ps = V[i].prev_sub; pi = V[i].prev_ind;
          if (used[i] == 0) { chain.push_back(i); used[i] = 1; }
          else { abandon(); break; }
        }
      }
      if (!chain.empty()) {
        uint32_t qEnd = qe[chain[0]], tEnd = te[chain[0]], qStart = q[chain.back()], tStart = t[chain.back()];
        for (size_t k = 0; k < chain.size(); k++) {
          qEnd = std::max(qe[chain[k]], qEnd); tEnd = std::max(te[chain[k]], tEnd);
          qStart = std::min(q[chain[k]], qStart); tStart = std::min(t[chain[k]], tStart);
        }
        if (((float)(qEnd - qStart) / o->readLen) > 0.005) {
          if (nChains >= o->NumAln || nChains >= maxChains) break;       // :1636-1645 (the first one opens Primary_chains[0])
          int na = 0;
          for (unsigned int x : chain) na += numAnchors ? numAnchors[x] : 0;   // ComputeNumOfAnchors :1577
          int off = chainOff[nChains];
          for (size_t k = 0; k < chain.size(); k++) chainFrag[off + k] = chain[k];
          for (size_t k = 0; k < link.size(); k++) chainLink[off + k] = link[k];
          if (chain.size() > link.size()) chainLink[off + link.size()] = 0;
          chainBox[4 * nChains] = qStart; chainBox[4 * nChains + 1] = qEnd; chainBox[4 * nChains + 2] = tStart; chainBox[4 * nChains + 3] = tEnd;
          chainValue[nChains] = V[order[fv]].val;
          if (chainNumAnchors) chainNumAnchors[nChains] = na;
          nChains++;
          chainOff[nChains] = off + (int)chain.size();
        } else break;
      }
      fv++;
    }
    return nChains;
  }
  float thres = o->alnthres * V[order[0]].val;
  int nChains = 0, fv = 0;
  uint32_t c0TS = 0, c0TE = 0;
  while (nChains < o->NumAln && nChains < maxChains && fv < total && V[order[fv]].val >= thres) {
    unsigned int i = order[fv];
    std::vector<unsigned int> chain;
    std::vector<bool> link;
    // TraceBack with `used` :1351-1438
    long ps = V[i].prev_sub, pi = V[i].prev_ind;
    if (used[i] == 0) {
      chain.push_back(i); used[i] = 1;
      auto abandon = [&]() { for (unsigned int x : chain) used[x] = 0; chain.clear(); link.clear(); };
      while (ps != -1 && pi != -1) {
        int fam = (V[i].inv ? 0 : 2) + (V[i].prev ? 0 : 1);
        Sub& s = c.subs[fam][ps];
        unsigned int ind = s.Ep[pi];
        unsigned int nx = s.Dp[ind];
        if (used[nx] == 0) { link.push_back(V[i].inv ? 0 : 1); i = nx; }
        else { abandon(); break; }
        ps = V[i].prev_sub; pi = V[i].prev_ind;
        if (used[i] == 0) { chain.push_back(i); used[i] = 1; }
        else { abandon(); break; }
      }
    }
    if (!chain.empty()) {
      int f = chain[0], l = chain.back();
      uint32_t QEnd = q[f] + len[f], QStart = q[l], TEnd = t[f] + len[f], TStart = t[l];
      for (size_t k = 0; k < chain.size(); k++) {
        f = chain[k];
        QEnd = std::max(QEnd, q[f] + (uint32_t)len[f]);
        QStart = std::min(QStart, q[f]);
        TStart = std::min(TStart, t[f]);
        TEnd = std::min(TEnd, t[f] + (uint32_t)len[f]);               // min, as the reference has it (:1694)
      }
      if (chain.size() >= 3 && QEnd > QStart && ((float)(QEnd - QStart) / o->readLen) > 0.005 && QEnd - QStart >= 200) {
        bool push = false;
        if (nChains == 0) push = true;
        else if (nChains < o->NumAln) {
          // chains[0].OverlapsOnT(TStart, TEnd, 0.05f)  Chain.h:261-276
          int ovp = 0;
          if (TStart >= c0TS && TStart < c0TE) ovp = std::min(TEnd, c0TE) - TStart;
          else if (TEnd > c0TS && TEnd <= c0TE) ovp = TEnd - std::max(TStart, c0TS);
          else if (TStart < c0TS && TEnd > c0TE) ovp = c0TE - c0TS;
          float denomA = c0TE - c0TS;
          push = (ovp / denomA <= 0.05f);
        } else break;
        if (push) {
          int off = chainOff[nChains];
          for (size_t k = 0; k < chain.size(); k++) chainFrag[off + k] = chain[k];
          for (size_t k = 0; k < link.size(); k++) chainLink[off + k] = link[k];
          if (chain.size() > link.size()) chainLink[off + link.size()] = 0;
          chainBox[4 * nChains] = QStart; chainBox[4 * nChains + 1] = QEnd; chainBox[4 * nChains + 2] = TStart; chainBox[4 * nChains + 3] = TEnd;
          chainValue[nChains] = V[order[fv]].val;
          if (nChains == 0) { c0TS = TStart; c0TE = TEnd; }
          nChains++;
          chainOff[nChains] = off + (int)chain.size();
        }
      } else break;
    }
    fv++;
  }
  return nChains;
}

extern "C" int oracle_sdp_chain(int nClusters, const int* clusterOff, const uint8_t* clusterStrand, const uint32_t* q, const uint32_t* t,
                                const int* len, const oracle_sdp_opts* o, float* fragVal, long* fragPrevSub, long* fragPrevInd,
                                uint8_t* fragFlags, int maxChains, int* chainOff, uint32_t* chainFrag, uint8_t* chainLink, uint32_t* chainBox,
                                float* chainValue) {
  if (o->mode == 2) return -1;
  return sdp_chain_impl(nClusters, clusterOff, clusterStrand, q, t, len, o, fragVal, fragPrevSub, fragPrevInd, fragFlags, maxChains, chainOff,
                        chainFrag, chainLink, chainBox, chainValue, nullptr, nullptr, nullptr, nullptr);
}

// The high-accuracy SparseDP (SparseDP.h:1956-2135, Map_highacc.h:229) over one read's split clusters: box i = (qs, qe, ts, te,
// strand, Val, NumofAnchors0); o->rate is the caller's `rate` (:227-228), o->mode is taken as 2.  Chains come back as for
// oracle_sdp_chain (indices are box indices), plus Num_Anchors per chain.
extern "C" int oracle_sdp_chain_boxes(int nBoxes, const uint32_t* qs, const uint32_t* qe, const uint32_t* ts, const uint32_t* te,
                                      const uint8_t* strand, const int* val, const int* numAnchors, const oracle_sdp_opts* o, float* fragVal,
                                      long* fragPrevSub, long* fragPrevInd, uint8_t* fragFlags, int maxChains, int* chainOff, uint32_t* chainFrag,
                                      uint8_t* chainLink, uint32_t* chainBox, float* chainValue, int* chainNumAnchors) {
  oracle_sdp_opts o2 = *o;
  o2.mode = 2;
  return sdp_chain_impl(nBoxes, nullptr, strand, qs, ts, val, &o2, fragVal, fragPrevSub, fragPrevInd, fragFlags, maxChains, chainOff, chainFrag,
                        chainLink, chainBox, chainValue, qe, te, numAnchors, chainNumAnchors);
}
