// oracle/ref_harness/order_ref.cpp -- driver around the REFERENCE's own SegAlignmentGroup::SetFromSegAlignment (Alignment.h:944-983) and
// AlignmentsOrder::Update (:1021-1046), compiled from /root/reference in place.
//
// stdin, one case per line:  nGroups  { nSeg  { valueBits N0 N1 qStart qEnd tStart tEnd nm nmm ndel nins strand supp isSecondary typeofaln flag } x nSeg } x nGroups
// stdout, one line per case: per group "G isSecondary valueBits N0 N1 qStart qEnd tStart tEnd nm nmm ndel nins" then per segment
//   "S flag supp isSecondary typeofaln", then "I" and the order (AlignmentsOrder::index)
#include <string>
#include <vector>
#include <iostream>
#include <sstream>
#include <cassert>
#include <algorithm>
#include <iomanip>
#include <cstring>
using namespace std;
#include "Alignment.h"

int main() {
  int nGroups;
  while (cin >> nGroups) {
    vector<vector<Alignment>> store(nGroups);
    vector<SegAlignmentGroup> alignments(nGroups);
    for (int g = 0; g < nGroups; g++) {
      int nSeg; cin >> nSeg;
      store[g].resize(nSeg);
      for (int s = 0; s < nSeg; s++) {
        Alignment& a = store[g][s];
        unsigned vb, qs, qe, ts, te, flag; int supp, sec;
        cin >> vb >> a.NumOfAnchors0 >> a.NumOfAnchors1 >> qs >> qe >> ts >> te >> a.nm >> a.nmm >> a.ndel >> a.nins >> a.strand >> supp >> sec >> a.typeofaln >> flag;
        memcpy(&a.value, &vb, 4);
        a.qStart = qs; a.qEnd = qe; a.tStart = ts; a.tEnd = te; a.Supplymentary = supp; a.ISsecondary = sec; a.flag = flag;
      }
      for (int s = 0; s < nSeg; s++) alignments[g].SegAlignment.push_back(&store[g][s]);
    }
    Options opts;
    for (int g = 0; g < nGroups; g++) alignments[g].SetFromSegAlignment(opts);
    AlignmentsOrder order(&alignments);
    if (nGroups > 0) order.Update(&alignments);
    for (int g = 0; g < nGroups; g++) {
      SegAlignmentGroup& G = alignments[g];
      unsigned vb; memcpy(&vb, &G.value, 4);
      cout << "G " << (int)G.ISsecondary << " " << vb << " " << G.NumOfAnchors0 << " " << G.NumOfAnchors1 << " " << G.qStart << " " << G.qEnd << " " << G.tStart << " "
           << G.tEnd << " " << G.nm << " " << G.nmm << " " << G.ndel << " " << G.nins << " ";
      for (size_t s = 0; s < G.SegAlignment.size(); s++) {
        Alignment* a = G.SegAlignment[s];
        cout << "S " << a->flag << " " << (int)a->Supplymentary << " " << (int)a->ISsecondary << " " << a->typeofaln << " ";
      }
    }
    cout << "I";
    for (int i = 0; i < order.size(); i++) cout << " " << order.index[i];
    cout << "\n";
  }
  return 0;
}
