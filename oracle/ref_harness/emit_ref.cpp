// oracle/ref_harness/emit_ref.cpp -- driver around the REFERENCE's own record emitters Alignment::PrintSAM / SimplePrintSAM /
// PrintPAF / PrintBed (Alignment.h:591-905), compiled from /root/reference in place.
//
// stdin, one case per line (all strings without white space; qual "NULL" = null pointer; passthrough "-" = none):
//   mode hardClip passthrough nGroup as  { name read qual chrom cigar readLen genomeLen flag strand mapqv supp typeofaln qStart qEnd tStart tEnd
//   preClip sufClip nm nmm nins ndel tdel tins nSmallDel nMedDel nLargeDel nSmallIns nMedIns nLargeIns valueBits order N0 N1 runtime nBlocks
//   firstBlockQPos lastBlockQEnd } x nGroup
//   mode: S PrintSAM, s SimplePrintSAM, P PrintPAF with CIGAR, p PrintPAF, B PrintBed
// stdout: the record (one line each)
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
  string mode;
  while (cin >> mode) {
    int hardClip, nGroup, as; string pass;
    cin >> hardClip >> pass >> nGroup >> as;
    Options opts;
    opts.hardClip = hardClip; opts.printMD = false; opts.passthroughtag = pass != "-";
    vector<Alignment> alns(nGroup);
    vector<string> reads(nGroup), quals(nGroup);
    vector<Alignment*> group;
    for (int g = 0; g < nGroup; g++) {
      Alignment& a = alns[g];
      string name, chrom, cigar;
      unsigned vb, flag, mapqv, qs, qe, ts, te, genomeLen, fb, lb;
      int supp, nBlocks;
      cin >> name >> reads[g] >> quals[g] >> chrom >> cigar >> a.readLen >> genomeLen >> flag >> a.strand >> mapqv >> supp >> a.typeofaln >> qs >> qe >> ts >> te
          >> a.preClip >> a.sufClip >> a.nm >> a.nmm >> a.nins >> a.ndel >> a.tdel >> a.tins >> a.nSmallDel >> a.nMedDel >> a.nLargeDel >> a.nSmallIns
          >> a.nMedIns >> a.nLargeIns >> vb >> a.order >> a.NumOfAnchors0 >> a.NumOfAnchors1 >> a.runtime >> nBlocks >> fb >> lb;
      a.readName = name; a.chrom = chrom; a.cigar = cigar == "-" ? "" : cigar; a.genomeLen = genomeLen; a.flag = flag; a.mapqv = (unsigned char)mapqv;
      a.Supplymentary = supp; a.qStart = qs; a.qEnd = qe; a.tStart = ts; a.tEnd = te;
      memcpy(&a.value, &vb, 4);
      a.read = (char*)reads[g].c_str();
      a.qual = quals[g] == "NULL" ? NULL : (char*)quals[g].c_str();
      a.prepared = true;
      for (int b = 0; b < nBlocks; b++) {
        if (b == nBlocks - 1) a.blocks.push_back(Block(nBlocks == 1 ? fb : lb - 1, 0, nBlocks == 1 ? lb - fb : 1));
        else a.blocks.push_back(Block(b == 0 ? fb : fb + b, 0, 1));
      }
      group.push_back(&a);
    }
    char* pt = pass == "-" ? NULL : (char*)pass.c_str();
    if (mode == "S") alns[as].PrintSAM(cout, opts, group, as, pt);
    else if (mode == "s") alns[as].SimplePrintSAM(cout, opts, pt);
    else if (mode == "P") alns[as].PrintPAF(cout, true);
    else if (mode == "p") alns[as].PrintPAF(cout, false);
    else alns[as].PrintBed(cout);
  }
  return 0;
}
