// oracle/ref_harness/sdp_parts_ref.cpp -- driver around the pieces of the REFERENCE's sparse-DP engine that compile
// without Clustering.h/Genome.h (htslib): Sorting.h (SortByRowOp, SortByColOp, Lower_Bound), Point.h, Info.h,
// SubProblem.h, DivideSubBy{Row1,Col1,Row2,Col2}.h (GetRowInfo, GetColInfo, ScanPoints_*, Decide_Eb_Db_*,
// DivideSubProbBy*), SubRountine.h (InitPWL, PWL_w, w, Maximization, FindValueInBlock, FindBoundary, UPPERbound).
// Compiled from /root/reference in place.  The driver below only feeds inputs and prints state; the order of calls in
// "D" is the one SparseDP.h:2171-2193 uses.
//
// stdin commands:
//   D n  q t ind inv (x n)                      -> canonical text of H1/H2 permutations, row/col tables, 4 decompositions, then "END"
//   P intercept scalar root g1 g2 n x (x n)     -> n lines "pwl_bits w_bits", then 25 lines "slope_bits inter_bits"
//   M intercept scalar root g1 g2 nD Di.. nE Ei.. nOps (op a v_bits) ..  -> one line of outputs, one line of the final Block
#include <assert.h>
#include <string.h>
#include <iostream>
#include <vector>
#include <string>
#include <set>
#include <map>
#include <stack>
#include <cmath>
#include <numeric>
#include <algorithm>
using namespace std;
#include "Options.h"
#include "SubProblem.h"
#include "Sorting.h"
#include "SubRountine.h"
#include "Fragment_Info.h"
#include "Info.h"
#include "DivideSubByRow1.h"
#include "DivideSubByCol1.h"
#include "DivideSubByRow2.h"
#include "DivideSubByCol2.h"
#include "Point.h"

template <typename T> static void dumpv(const char* name, const vector<T>& v) {
  cout << name << ':';
  for (size_t i = 0; i < v.size(); i++) cout << (long)v[i] << ',';
  cout << ';';
}
static unsigned bits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static float unbits(unsigned u) { float f; memcpy(&f, &u, 4); return f; }

int main() {
  string cmd;
  Options opts;
  vector<float> lut;
  while (cin >> cmd) {
    if (cmd == "D") {
      int n; cin >> n;
      vector<Point> H1;
      for (int i = 0; i < n; i++) {
        unsigned q, t; int ind, inv; cin >> q >> t >> ind >> inv;
        Point p; p.se.first = q; p.se.second = t; p.ind = ind; p.inv = inv; p.frag_num = i; p.clusterNum = 0; p.orient = 0;
        H1.push_back(p);
      }
      if (n > 0) {
        sort(H1.begin(), H1.end(), SortByRowOp<Point>());
        vector<unsigned int> H2(H1.size());
        iota(H2.begin(), H2.end(), 0);
        sort(H2.begin(), H2.end(), SortByColOp<Point, unsigned int>(H1));
        vector<info> Row, Col;
        GetRowInfo(H1, Row);
        GetColInfo(H1, H2, Col);
        unsigned int n1 = 0, m1 = 0, n2 = 0, m2 = 0;
        StackOfSubProblems S[4];
        int e0 = 0, e1 = 0, e2 = 0, e3 = 0;
        DivideSubProbByRow1(H1, Row, 0, Row.size(), n1, S[0], e0);
        DivideSubProbByCol1(H1, H2, Col, 0, Col.size(), m1, S[1], e1);
        DivideSubProbByRow2(H1, Row, 0, Row.size(), n2, S[2], e2);
        DivideSubProbByCol2(H1, H2, Col, 0, Col.size(), m2, S[3], e3);
        vector<unsigned int> perm;
        for (size_t i = 0; i < H1.size(); i++) perm.push_back(H1[i].frag_num);
        dumpv("H1", perm); dumpv("H2", H2); cout << '\n';
        for (int rc = 0; rc < 2; rc++) {
          vector<info>& V = rc ? Col : Row;
          for (size_t i = 0; i < V.size(); i++) {
            cout << (rc ? 'C' : 'R') << i << ':' << V[i].pstart << ',' << V[i].pend << ',' << V[i].rc_num << ';';
            dumpv("A1", V[i].SS_A1); dumpv("B1", V[i].SS_B1); dumpv("A2", V[i].SS_A2); dumpv("B2", V[i].SS_B2);
            cout << '\n';
          }
        }
        for (int f = 0; f < 4; f++)
          for (int i = 0; i < S[f].size(); i++) {
            Subproblem& s = S[f][i];
            cout << 'F' << f << '.' << i << ":num=" << s.num << ';';
            dumpv("Di", s.Di); dumpv("Ei", s.Ei); dumpv("Db", s.Db); dumpv("Eb", s.Eb);
            cout << "S=" << s.S_1.size() << ";\n";
          }
      }
      cout << "END\n";
    } else if (cmd == "P") {
      float a, b, c; int g1, g2, n;
      cin >> a >> b >> c >> g1 >> g2 >> n;
      InitPWL(a, b, c, g1, g2);
      bool step = 0;
      for (int i = 0; i < n; i++) {
        long x; cin >> x;
        cout << bits(PWL_w(x, 0)) << ' ' << bits(w(0, x - 1, lut, opts, step)) << '\n';
      }
      for (int i = 0; i < NUMPWL; i++) cout << bits(SLOPE[i]) << ' ' << bits(INTER[i]) << '\n';
    } else if (cmd == "M") {
      float a, b, c; int g1, g2;
      cin >> a >> b >> c >> g1 >> g2;
      InitPWL(a, b, c, g1, g2);
      Subproblem s(0);
      int nD, nE, nOps;
      cin >> nD; s.Di.resize(nD); for (int i = 0; i < nD; i++) cin >> s.Di[i];
      cin >> nE; s.Ei.resize(nE); for (int i = 0; i < nE; i++) cin >> s.Ei[i];
      // the "non-leaf case" initialisation of DivideSubByRow1.h:458-471
      s.E.assign(nE, 0); iota(s.E.begin(), s.E.end(), 0);
      s.Eb.assign(nE, -1); s.Db.assign(nD, -1);
      Decide_Eb_Db_R1(s.Di, s.Ei, s.Db, s.Eb, s.E);
      s.Dv.assign(nD, 0); s.Dp.assign(nD, 0); s.D.assign(nD, 0); iota(s.D.begin(), s.D.end(), 0);
      s.Ev.assign(nE, 0); s.Ep.assign(nE, 0);
      s.S_1.push(make_pair((long)-1, (long)nE + 1));
      cin >> nOps;
      bool step = 0;
      for (int k = 0; k < nOps; k++) {
        int op; long x; unsigned vb; cin >> op >> x >> vb;
        if (op == 0) { float v = unbits(vb); if (s.Dv[x] < v) { s.Dv[x] = v; s.Dp[x] = k; } }
        else {
          if (s.Eb[x] == -1) { cout << -1 << ' '; continue; }
          s.now = s.Eb[x];
          Maximization(s.now, s.last, s.Di, s.Ei, s.Dv, s.Db, s.Block, s.S_1, lut, opts, step);
          s.last = s.Eb[x];
          unsigned int i1 = x, i2;
          long fd = s.Ei[x];
          FindValueInBlock(fd, s.S_1, s.Ei, s.Block, i1, i2);
          cout << i2 << ' ';
        }
      }
      cout << '\n';
      for (size_t i = 0; i < s.Block.size(); i++) cout << s.Block[i].first << ' ' << s.Block[i].second << ' ';
      cout << '\n';
    }
  }
  return 0;
}
