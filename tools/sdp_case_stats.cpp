// tools/sdp_case_stats.cpp -- analysis only (not product, not test): runs the oracle's sparse DP (oracle/sdp.cpp, included as is) over the jobs of a
// dump written by LRA_SDP_DUMP / ORACLE_SDP_DUMP and counts, per family and sub-problem size class, the steps ProcessPoint takes: queries, candidates
// walked by Maximization, candidates that beat the stack top, pops, FindBoundary / Block binary searches and their lengths.
//   g++ -O2 -std=c++17 -I oracle tools/sdp_case_stats.cpp -o /tmp/sdp_case_stats && /tmp/sdp_case_stats dump.bin [case]
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <unordered_map>
#include <vector>

struct StatRow { long subs, ent, query, qback, span, iter, pop1, win, pop2, fb, fbLog, pop3, fast, bs, bsLog, dep; };
static StatRow g_stat[4][24];
static std::unordered_map<const void*, int> g_tag;    // sub -> fam * 32 + size class
static inline int lg(long x) { int l = 0; while (x > 1) { x >>= 1; l++; } return l; }
template <class S> static inline StatRow& row(const S& s) { auto it = g_tag.find((const void*)&s); int k = it == g_tag.end() ? 0 : it->second; return g_stat[k >> 5][k & 31]; }
template <class S> static inline void stat_query(const S& s, long v) { StatRow& r = row(s); r.query++; if (v > 0) r.span += v; else if (v < 0) r.qback++; }
template <class S> static inline void stat_iter(const S& s, long v) { row(s).iter += v; }
template <class S> static inline void stat_pop1(const S& s, long v) { row(s).pop1 += v; }
static long g_popHist[8], g_depthHist[8], g_curPops = -1;
template <class S> static inline void stat_win(const S& s, long v) { row(s).win += v; if (g_curPops >= 0) g_popHist[g_curPops > 7 ? 7 : g_curPops]++; g_curPops = 0; long d = (long)s.S.size(); int b = 0; while (d > 1 && b < 7) { d >>= 1; b++; } g_depthHist[b]++; }
template <class S> static inline void stat_pop2(const S& s, long v) { row(s).pop2 += v; g_curPops += v; }
template <class S> static inline void stat_fb(const S& s, long v) { StatRow& r = row(s); r.fb++; r.fbLog += lg(v + 1); }
template <class S> static inline void stat_pop3(const S& s, long v) { row(s).pop3 += v; }
template <class S> static inline void stat_fast(const S& s, long v) { row(s).fast += v; }
template <class S> static inline void stat_bsearch(const S& s, long v) { StatRow& r = row(s); r.bs++; r.bsLog += lg(v + 1); }
#define SDP_STAT(kind, s, v) stat_##kind(s, v)
#include "sdp.cpp"

struct Case { int mode, nc, total, rl; float rate; std::vector<int> off; std::vector<uint8_t> st; std::vector<uint32_t> q, t; std::vector<int> len; };

int main(int argc, char** argv) {
  if (argc < 2) return 1;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 1;
  std::vector<Case> cases;
  for (;;) {
    int hdr[4];
    if (fread(hdr, 4, 4, f) != 4) break;
    Case c; c.mode = hdr[0]; c.nc = hdr[1]; c.total = hdr[2]; c.rl = hdr[3];
    fread(&c.rate, 4, 1, f);
    c.off.resize(c.nc + 1); c.st.resize(c.nc); c.q.resize(c.total); c.t.resize(c.total); c.len.resize(c.total);
    fread(c.off.data(), 4, c.nc + 1, f); fread(c.st.data(), 1, c.nc, f); fread(c.q.data(), 4, c.total, f); fread(c.t.data(), 4, c.total, f); fread(c.len.data(), 4, c.total, f);
    cases.push_back(c);
  }
  fclose(f);
  const int which = argc > 2 ? atoi(argv[2]) : 0;
  if (which >= (int)cases.size()) return 1;
  const Case& c = cases[which];
  printf("case %d: mode %d clusters %d anchors %d\n", which, c.mode, c.nc, c.total);
  // the oracle's own steps, with the subs tagged after build(): replicate the head of sdp_chain_impl
  Ctx cx;
  cx.pwl.init(4.0f, 10.0f, 1.5f, 1500, 3000);            // -ONT: gapopen 4 gapextend 10 gaproot 1.5 ceilings 1500 / 3000
  for (int cm = 0; cm < c.nc; cm++) {
    int ms = c.off[cm], sz = c.off[cm + 1] - ms;
    for (int i = 0; i < sz; i++) {
      int g = ms + i;
      bool edge = c.mode == 0 && (i == 0 || i == sz - 1);
      if (c.st[cm] == 0) { insert_pair(cx.H1, g, c.q[g], c.t[g], c.len[g], cm, 0, 1); if (edge) insert_pair(cx.H1, g, c.q[g], c.t[g], c.len[g], cm, 1, 1); }
      else { insert_pair(cx.H1, g, c.q[g], c.t[g], c.len[g], cm, 1, 0); if (edge) insert_pair(cx.H1, g, c.q[g], c.t[g], c.len[g], cm, 0, 0); }
    }
  }
  auto t0 = std::chrono::steady_clock::now();
  build(cx);
  auto t1 = std::chrono::steady_clock::now();
  long nsub = 0, nent = 0;
  for (int fam = 0; fam < 4; fam++)
    for (auto& s : cx.subs[fam]) {
      const int cls = lg((long)(s.Di.size() + s.Ei.size()));
      g_tag[(const void*)&s] = fam * 32 + cls;
      g_stat[fam][cls].subs++; g_stat[fam][cls].ent += (long)(s.Di.size() + s.Ei.size());
      nsub++; nent += (long)(s.Di.size() + s.Ei.size());
    }
  printf("points %zu rows %zu cols %zu subs %ld entries %ld  build %.1f ms\n", cx.H1.size(), cx.Row.size(), cx.Col.size(), nsub, nent,
         std::chrono::duration<double, std::milli>(t1 - t0).count());
  // ProcessPoint, as sdp_chain_impl has it (values only)
  const int total = c.total;
  std::vector<Frag> V(total);
  for (int rc = 0; rc < 2; rc++) {
    std::vector<Info>& T = rc ? cx.Col : cx.Row;
    for (size_t ti = 0; ti < T.size(); ti++)
      for (uint32_t tt = T[ti].pstart; tt < T[ti].pend; tt++) {
        const Pt& p = rc ? cx.H1[cx.H2[tt]] : cx.H1[tt];
        Frag& fr = V[p.frag];
        int fam = (p.inv ? 0 : 2) + rc;
        int k = p.inv ? 0 : 1;
        if (p.ind == 1) { fr.B[fam] = T[ti].B[k]; fr.val = c.len[p.frag] * c.rate; }
        else fr.A[fam] = T[ti].A[k];
      }
  }
  auto t2 = std::chrono::steady_clock::now();
  long nVisQ = 0, nVisD = 0;
  for (size_t i = 0; i < cx.H1.size(); i++) {
    const Pt& p = cx.H1[i];
    long fd = (long)p.t - (long)p.q, bd = (long)p.t + (long)p.q;
    uint32_t ii = p.frag;
    Frag& F = V[ii];
    for (int rc = 0; rc < 2; rc++) {
      int fam = (p.inv ? 0 : 2) + rc;
      const Family& fm = FAM[fam];
      long dg = fm.back ? bd : fd;
      std::vector<Sub>& S = cx.subs[fam];
      if (p.ind == 1) {
        const std::vector<uint32_t>& L = F.B[fam];
        for (size_t k = 0; k < L.size(); k++) {
          Sub& s = S[L[L.size() - 1 - k]];
          if (s.Di.empty()) continue;
          size_t pos = lower_pos(s.Ei, dg, fm.desc);
          unsigned int i1 = (unsigned int)pos2idx(s.Ei.size(), pos, fm.desc);
          if (s.Eb[i1] == -1) continue;
          nVisQ++;
          s.now = (uint32_t)s.Eb[i1];
          maximization(s, cx.pwl);
          s.last = s.Eb[i1];
          unsigned int i2;
          if (!find_value_in_block(s, i1, i2)) { printf("UB\n"); return 2; }
          s.Ev[i1] = s.Dv[i2] + cx.pwl.w(s.Di[i2], s.Ei[i1]) + c.rate * c.len[ii];
          s.Ep[i1] = i2;
          if (F.val < s.Ev[i1]) { F.val = s.Ev[i1]; F.prev_sub = s.num; F.prev_ind = i1; F.prev = (rc == 0); F.inv = fm.inv; }
        }
      } else {
        const std::vector<uint32_t>& L = F.A[fam];
        for (size_t k = 0; k < L.size(); k++) {
          Sub& s = S[L[L.size() - 1 - k]];
          if (s.Ei.empty()) continue;
          size_t pos = lower_pos(s.Di, dg, fm.desc);
          size_t d = pos2idx(s.Di.size(), pos, fm.desc);
          nVisD++;
          row(s).dep++;
          if (s.Dv[d] < F.val) { s.Dv[d] = F.val; s.Dp[d] = ii; }
        }
      }
    }
  }
  auto t3 = std::chrono::steady_clock::now();
  printf("ProcessPoint %.1f ms: %ld query visits, %ld deposit visits (%.1f visits per point)\n", std::chrono::duration<double, std::milli>(t3 - t2).count(), nVisQ, nVisD,
         (double)(nVisQ + nVisD) / cx.H1.size());
  {                                                       // is Block sorted by .second (then any upper-bound search gives the literal one's answer)?
    long sortedSubs = 0, unsortedSubs = 0, inv = 0, tot = 0; long unsortedTop = 0;
    for (int fam = 0; fam < 4; fam++) for (auto& s : cx.subs[fam]) {
      long v = 0;
      for (size_t k = 1; k < s.Block.size(); k++) v += s.Block[k].second < s.Block[k - 1].second;
      tot += (long)s.Block.size(); inv += v;
      if (v) { unsortedSubs++; if (s.Di.size() + s.Ei.size() >= 256) unsortedTop++; } else sortedSubs++;
    }
    printf("Block order: %ld subs sorted, %ld not (%ld of them with >= 256 entries); %ld descents in %ld pairs\n", sortedSubs, unsortedSubs, unsortedTop, inv, tot);
  }
  long maxStack = 0, maxBlock = 0, sumBlock = 0;
  for (int fam = 0; fam < 4; fam++) for (auto& s : cx.subs[fam]) { maxBlock = std::max<long>(maxBlock, s.Block.size()); sumBlock += s.Block.size(); maxStack = std::max<long>(maxStack, s.S.size()); }
  printf("Block: max %ld total %ld; final stack max %ld\n", maxBlock, sumBlock, maxStack);
  printf("pops per win event (0,1,..,7+):"); for (int i = 0; i < 8; i++) printf(" %ld", g_popHist[i]); printf("\nstack depth at win (log2 buckets):"); for (int i = 0; i < 8; i++) printf(" %ld", g_depthHist[i]); printf("\n");
  printf("fam cls   subs     ent    dep   query  qback     span     iter    win   pop1   pop2   pop3     fb fbLog/fb   fast     bs bsLog/bs\n");
  for (int fam = 0; fam < 4; fam++)
    for (int cl = 23; cl >= 0; cl--) {
      const StatRow& r = g_stat[fam][cl];
      if (!r.subs) continue;
      printf("%s %3d %6ld %7ld %6ld %7ld %6ld %8ld %8ld %6ld %6ld %6ld %6ld %6ld %8.1f %6ld %6ld %8.1f\n", fam == 0 ? "R1" : fam == 1 ? "C1" : fam == 2 ? "R2" : "C2", cl, r.subs, r.ent, r.dep, r.query,
             r.qback, r.span, r.iter, r.win, r.pop1, r.pop2, r.pop3, r.fb, r.fb ? (double)r.fbLog / r.fb : 0.0, r.fast, r.bs, r.bs ? (double)r.bsLog / r.bs : 0.0);
    }
  return 0;
}
