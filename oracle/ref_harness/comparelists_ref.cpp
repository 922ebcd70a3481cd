// oracle/ref_harness/comparelists_ref.cpp -- driver around the REFERENCE's own
// CompareLists<GenomeTuple,Tuple> (CompareLists.h:9, compiled from /root/reference in
// place).  Also runs the reference's std::sort(readmm) (MapRead.h:185) when asked.
//
// stdin : cases:  nq nt maxFreq sortq   then nq lines "t pos", then nt lines "t pos"
// stdout: per case: one line "n qi0 ti0 qi1 ti1 ..."  where qi/ti identify the emitted
//         tuples by their (unique) pos fields mapped back to list indices;
//         if sortq: a second line with the sorted query order as original indices.
#include <string>
#include <vector>
#include <iostream>
#include <cassert>
#include <algorithm>
#include <map>
using namespace std;
#include "Options.h"
#include "TupleOps.h"
#include "CompareLists.h"

// mode 1 (first token "L"): CompareLists<LocalTuple,SmallTuple> with Global=false and a diagonal band:
//   L nq nt maxFreq maxDiag minDiag   then nq + nt lines "t pos"   ->  "n qi0 ti0 ..." (list indices)
static void local_mode() {
  long nq, nt, maxFreq, maxDiag, minDiag;
  cin >> nq >> nt >> maxFreq >> maxDiag >> minDiag;
  vector<LocalTuple> q(nq), t(nt);
  for (long i = 0; i < nq; i++) { unsigned a, b; cin >> a >> b; q[i].t = a; q[i].pos = b; }
  for (long i = 0; i < nt; i++) { unsigned a, b; cin >> a >> b; t[i].t = a; t[i].pos = b; }
  Options opts;
  opts.localMaxFreq = maxFreq;
  vector<pair<LocalTuple, LocalTuple> > res;
  // identify emitted tuples by address-free keys: (t,pos) may repeat, so tag pos with the index
  CompareLists<LocalTuple, SmallTuple>(q.begin(), q.end(), t.begin(), t.end(), res, opts, false, maxDiag, minDiag, false);
  cout << res.size();
  for (auto& r : res) cout << " " << r.first.t << " " << r.first.pos << " " << r.second.t << " " << r.second.pos;
  cout << "\n";
}

int main() {
  LocalTuple::for_mask_s = 0xFFFFF;   // as InitStatic (lra.cpp:1013-1017)
  LocalTuple::rev_mask_s = 0;
  Tuple mask = 1;
  GenomeTuple::for_mask_s = ~(mask << 63);   // as InitStatic (lra.cpp:1008-1012)
  GenomeTuple::rev_mask_s = (mask << 63);
  long nq, nt, maxFreq; int sortq;
  string tok;
  while (cin >> tok) {
    if (tok == "L") { local_mode(); continue; }
    nq = stol(tok);
    cin >> nt >> maxFreq >> sortq;
    vector<GenomeTuple> q(nq), t(nt);
    for (long i = 0; i < nq; i++) { unsigned long long a; unsigned int b; cin >> a >> b; q[i].t = a; q[i].pos = b; }
    for (long i = 0; i < nt; i++) { unsigned long long a; unsigned int b; cin >> a >> b; t[i].t = a; t[i].pos = b; }
    map<unsigned int, long> origq;
    for (long i = 0; i < nq; i++) origq[q[i].pos] = i;
    if (sortq) sort(q.begin(), q.end());
    map<unsigned int, long> qidx, tidx;
    for (long i = 0; i < nq; i++) qidx[q[i].pos] = i;
    for (long i = 0; i < nt; i++) tidx[t[i].pos] = i;
    Options opts;
    opts.globalMaxFreq = maxFreq;
    vector<pair<GenomeTuple, GenomeTuple> > res;
    CompareLists<GenomeTuple, Tuple>(q, t, res, opts, true);
    cout << res.size();
    for (auto& r : res) cout << " " << qidx[r.first.pos] << " " << tidx[r.second.pos];
    cout << "\n";
    if (sortq) {
      for (long i = 0; i < nq; i++) cout << (i ? " " : "") << origq[q[i].pos];
      cout << "\n";
    }
  }
  return 0;
}
