// oracle/mapqv.cpp -- TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// CPU restatement of SimpleMapQV (Mapping_ultility.h:497-595) on plain arrays.  PARITY UNPINNED: Mapping_ultility.h includes Genome.h
// (htslib).  (SetFromSegAlignment / AlignmentsOrder are pinned to the reference directly: ref_harness/order_ref.cpp.)
#include "oracle_common.h"
#include <cmath>

// Alignments in AlignmentsOrder order: alignment r has segments [segOff[r], segOff[r+1]); per alignment value / NumOfAnchors0 (the
// SegAlignmentGroup's), per segment N0, nm, nmm, ndel, nins, value.  Out: mapqv per segment.
extern "C" void oracle_simple_mapqv(int len, const int* segOff, const float* gValue, const int* gN0, const int* N0, const int* nm, const int* nmm, const int* ndel,
                                    const int* nins, const float* value, int bypass, int isClr, int isOnt, int globalK, int* mapqv) {
  float q_coef;
  if (bypass && isClr) q_coef = 4.0f;
  else if (bypass && isOnt) q_coef = 30.0f;
  else q_coef = 1.0f;
  for (int r = 0; r < len; r++) {
    if (r == 0 && len == 1) {
      for (int s = segOff[r + 1] - 1; s >= segOff[r]; s--) {
        float pen_cm_1;
        if (!bypass) { pen_cm_1 = (N0[s] > 20 ? 1.0f : 0.05f) * N0[s]; pen_cm_1 = (N0[s] >= 5 ? 1.0f : 0.1f) * pen_cm_1; }
        else { pen_cm_1 = (N0[s] > 10 ? 1.0f : 0.05f) * N0[s]; pen_cm_1 = (N0[s] >= 5 ? 1.0f : 0.02f) * pen_cm_1; }
        float identity;
        if (nmm[s] + ndel[s] + nins[s] == 0) identity = 1.0f;
        else identity = ((float)nm[s]) / (nmm[s] + ndel[s] + nins[s]);
        identity = (identity < 1 ? identity : 1);
        float l = (value[s] > 3 ? logf(value[s] / globalK) : 0);
        long mapq;
        if (!bypass) mapq = (int)(pen_cm_1 * q_coef * l * identity);
        else mapq = (int)(pen_cm_1 * q_coef * identity);
        mapq = mapq > 0 ? mapq : 0;
        mapqv[s] = (unsigned char)(mapq < 60 ? mapq : 60);
        if (r == 0 && len == 2 && mapqv[s] == 0) mapqv[s] = 1;
      }
    } else if (r == 0 && len > 1) {
      float x = gValue[r + 1] / gValue[r];
      float y = 1.0f;
      for (int s = segOff[r + 1] - 1; s >= segOff[r]; s--) {
        float pen_cm_1;
        if (!bypass) { pen_cm_1 = (N0[s] > 20 ? 1.0f : 0.05f) * N0[s]; pen_cm_1 = (N0[s] >= 5 ? 1.0f : 0.1f) * pen_cm_1; }
        else {
          y = ((float)gN0[r]) / ((float)gN0[r + 1]);
          pen_cm_1 = (N0[s] > 10 ? 1.0f : 0.05f) * N0[s]; pen_cm_1 = (N0[s] >= 5 ? 1.0f : 0.02f) * pen_cm_1;
        }
        float identity;
        if (nmm[s] + ndel[s] + nins[s] == 0) identity = 1.0f;
        else identity = ((float)nm[s]) / (nmm[s] + ndel[s] + nins[s]);
        float l = (value[s] > 3 ? logf(value[s] / globalK) : 0);
        identity = (identity < 1 ? identity : 1);
        long mapq;
        if (x >= 0.990f) mapq = (int)(pen_cm_1 * (1.0f - x) * y * identity);
        else if (!bypass) mapq = (int)(pen_cm_1 * q_coef * (1.0f - x) * l * y * identity);
        else mapq = (int)(pen_cm_1 * q_coef * (1.0f - x) * y * identity);
        mapq -= (int)(4.343f * logf(len) + .499f);
        mapq = mapq > 0 ? mapq : 0;
        mapqv[s] = (unsigned char)(mapq < 60 ? mapq : 60);
        if (r == 0 && len == 2 && mapqv[s] == 0) mapqv[s] = 1;
      }
    } else {
      for (int s = segOff[r + 1] - 1; s >= segOff[r]; s--) mapqv[s] = 0;
    }
  }
}
