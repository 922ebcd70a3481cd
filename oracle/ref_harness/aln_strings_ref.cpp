// oracle/ref_harness/aln_strings_ref.cpp -- driver around the REFERENCE's own Alignment::CreateAlignmentStrings (Alignment.h:247-331),
// AlignmentStringsToMD (:204-245) and PrintPairwise (:564-589), compiled from /root/reference in place.
//
// stdin, one case per line:  readName chrom read text nBlocks { qPos tPos length } x nBlocks
// stdout per case: "Q <queryString>", "A <alignString with ' ' shown as '_'>", "T <refString>", "R <refLen>", "M <md>", the PrintPairwise text,
// then a line "@@END"
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
  string name, chrom, read, text; int nb;
  while (cin >> name >> chrom >> read >> text >> nb) {
    Alignment a;
    a.readName = name; a.chrom = chrom; a.readLen = read.size(); a.genomeLen = text.size();
    a.read = (char*)read.c_str(); a.genome = (char*)text.c_str();
    for (int b = 0; b < nb; b++) { int q, t, l; cin >> q >> t >> l; a.blocks.push_back(Block(q, t, l)); }
    a.CreateAlignmentStrings(a.read, a.genome, a.queryString, a.alignString, a.refString);
    string md;
    a.AlignmentStringsToMD(a.queryString, a.refString, md);
    string al = a.alignString; for (size_t i = 0; i < al.size(); i++) if (al[i] == ' ') al[i] = '_';
    cout << "Q " << a.queryString << "\nA " << al << "\nT " << a.refString << "\nR " << a.refLen << "\nM " << md << "\n";
    a.prepared = true;
    a.PrintPairwise(cout);
    cout << "@@END\n";
  }
  return 0;
}
