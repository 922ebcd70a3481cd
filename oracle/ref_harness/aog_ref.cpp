// oracle/ref_harness/aog_ref.cpp -- driver around the REFERENCE's own
// AffineOneGapAlign (compiled from /root/reference/AffineOneGapAlign.h where it
// lies; nothing of the reference is copied here).  Built only when
// /root/reference exists, output goes to oracle/_ref/aog_ref (git-ignored).
//
// stdin : one case per line:  <q|-> <t|-> m mm indel k      ("-" = empty string)
// stdout: one line per case:  score nblocks q0 t0 l0 q1 t1 l1 ...
#include <string>
#include <vector>
#include <iostream>
#include <cassert>
#include <algorithm>
#include <iomanip>
using namespace std;
#include "AffineOneGapAlign.h"

int main() {
  string q, t;
  int m, mm, indel, k;
  AffineAlignBuffers buf;
  while (cin >> q >> t >> m >> mm >> indel >> k) {
    if (q == "-") q = "";
    if (t == "-") t = "";
    Alignment aln;
    int s = AffineOneGapAlign(q, (int)q.size(), t, (int)t.size(), m, mm, indel, k, aln, buf);
    cout << s << " " << aln.blocks.size();
    for (auto& b : aln.blocks) cout << " " << b.qPos << " " << b.tPos << " " << b.length;
    cout << "\n";
  }
  return 0;
}
