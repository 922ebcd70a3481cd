// oracle/ref_harness/stats_ref.cpp -- driver around the REFERENCE's own Alignment::CalculateStatistics
// (Alignment.h:513-531 -> CreateAlignmentStrings :247, AlignStringsToCigar :414), compiled from
// /root/reference in place.
//
// stdin : cases:  read genome nblocks  q0 t0 l0 q1 t1 l1 ...     (sequences without spaces)
// stdout: per case one line:
//   cigar nm nmm nins ndel tdel tins nSmallDel nMedDel nLargeDel nSmallIns nMedIns nLargeIns value(hex float bits) preClip sufClip qStart qEnd tStart tEnd
#include <string>
#include <vector>
#include <iostream>
#include <cassert>
#include <algorithm>
#include <iomanip>
#include <cstring>
using namespace std;
#include "Alignment.h"
#include "LogLookUpTable.h"

int main() {
  vector<float> lut;
  CreateLookUpTable(lut);
  string r, g;
  int nb;
  Options opts;
  while (cin >> r >> g >> nb) {
    Alignment a;
    a.read = (char*)r.c_str(); a.genome = (char*)g.c_str();
    a.readLen = r.size(); a.genomeLen = g.size();
    for (int i = 0; i < nb; i++) { unsigned q, t, l; cin >> q >> t >> l; a.blocks.push_back(Block(q, t, l)); }
    a.CalculateStatistics(opts, NULL, lut);
    unsigned bits; float v = a.value; memcpy(&bits, &v, 4);
    cout << (a.cigar.empty() ? "*" : a.cigar) << " " << a.nm << " " << a.nmm << " " << a.nins << " " << a.ndel << " " << a.tdel << " " << a.tins << " "
         << a.nSmallDel << " " << a.nMedDel << " " << a.nLargeDel << " " << a.nSmallIns << " " << a.nMedIns << " " << a.nLargeIns << " "
         << bits << " " << a.preClip << " " << a.sufClip << " " << a.qStart << " " << a.qEnd << " " << a.tStart << " " << a.tEnd << "\n";
  }
  return 0;
}
