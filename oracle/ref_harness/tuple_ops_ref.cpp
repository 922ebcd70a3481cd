// oracle/ref_harness/tuple_ops_ref.cpp -- TEST INFRASTRUCTURE ONLY.
// The k-mer primitives StoreMinimizers (MinCount.h:8-179, itself behind an htslib include) is made of, from the reference's own headers compiled in place:
// seqMap / seqMapN (SeqUtils.h:5-75), StoreTuple, ShiftOne, ShiftOneRC, TupleRC (TupleOps.h:104-138), GenomeTuple::operator< (TupleOps.h:76), CreateRC
// (SeqUtils.h:151-158).  stdin: number of cases, then per case "k sequence"; stdout per case: the forward and reverse-complement codes of every k-mer as
// StoreMinimizers steps them (StoreTuple + TupleRC for the first, ShiftOne / ShiftOneRC after), the canonical key it would store, and CreateRC of the sequence.
#include <vector>
#include <string>
#include <iostream>
#include <cstdio>
#include <cstring>
using namespace std;
#include "SeqUtils.h"
#include "TupleOps.h"

int main() {
  Tuple for_mask_s = ~(((Tuple)1) << 63), rev_mask_s = ((Tuple)1) << 63;
  int P;
  if (scanf("%d", &P) != 1) return 1;
  static char buf[1 << 20];
  for (int c = 0; c < P; c++) {
    int k;
    if (scanf("%d %1048000s", &k, buf) != 2) return 1;
    const int n = (int)strlen(buf);
    GenomeTuple cur, curRC, mask;
    mask.t = 0;
    for (int i = 0; i < k; i++) { mask.t <<= 2; mask.t += 3; }                     // InitMask TupleOps.h:95-101
    StoreTuple(buf, 0, k, cur);
    TupleRC(cur, curRC, k);
    printf("%d", n - k + 1);
    for (int p = 0; p + k <= n; p++) {
      if (p > 0) { ShiftOne(buf, p + k - 1, mask, cur); ShiftOneRC(buf, p + k - 1, k, curRC); }
      Tuple key = ((cur.t & for_mask_s) < (curRC.t & for_mask_s)) ? (cur.t & for_mask_s) : (curRC.t | rev_mask_s);   // MinCount.h:60-61
      printf(" %llu %llu %llu", (unsigned long long)cur.t, (unsigned long long)curRC.t, (unsigned long long)key);
    }
    char* rc = NULL;
    CreateRC(buf, n, rc);
    printf("\n%.*s\n", n, rc);
    delete[] rc;
  }
  return 0;
}
