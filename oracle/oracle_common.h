// oracle/oracle_common.h
//
// TEST INFRASTRUCTURE ONLY.  This directory holds a CPU restatement of the
// reference's (ChaissonLab/LRA) per-read alignment hot path.  It exists so
// that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can
// check / time the HIP product path against it.  Nothing under lra_amd/ may
// include, link, import or execute anything from this directory.
#pragma once
#include <stdint.h>
#include <stddef.h>

// ASCII -> 2-bit code, non-ACGT -> 4   (reference: SeqUtils.h:42-75 seqMapN;
// indices 0..7 map to 0,1,2,3,0,1,2,3 exactly as the reference table does).
static inline int oracle_code_n(unsigned char c) {
  if (c < 8) return c & 3;
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
  }
}
// ASCII -> 2-bit code, non-ACGT -> 0   (reference: SeqUtils.h:7-40 seqMap)
static inline int oracle_code(unsigned char c) {
  int v = oracle_code_n(c);
  return v > 3 ? 0 : v;
}
