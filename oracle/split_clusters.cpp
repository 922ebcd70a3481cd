// oracle/split_clusters.cpp -- TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// CPU restatement of the high-accuracy path's cluster cutting (Map_highacc.h:153-155):
//   IntervalSet (line through a cluster's box, mixed q/t comparator)   SplitClusters.h:18-60
//   SplitClusters                                                        SplitClusters.h:63-171
//   DecideSplitClustersValue                                             SplitClusters.h:176-249  (CartesianLowerBound Sorting.h:171-178)
// Parity status: PARITY UNPINNED -- SplitClusters.h needs Cluster (Clustering.h -> Genome.h -> htslib); restated from the source text.
//
// The double -> GenomePos conversions of the reference are undefined for negative values; here (and in the kernel) they go through
// int64_t, which is what x86-64 g++ emits for them.
#include "oracle_common.h"
#include <algorithm>
#include <cmath>
#include <set>
#include <utility>
#include <vector>

namespace {

struct Box { uint32_t qs, qe, ts, te; int strand; };

struct IntervalSet {                                                     // SplitClusters.h:18-60
  double slope, intercept;
  bool strand;
  std::vector<std::pair<uint32_t, bool>> Set;
  explicit IntervalSet(const Box& c) {
    slope = (double)((int64_t)c.te - (int64_t)c.ts) / ((int64_t)c.qe - (int64_t)c.qs);
    if (c.strand == 0) {
      intercept = ((double)((int64_t)c.qe * c.ts - (int64_t)c.qs * c.te)) / ((int64_t)c.qe - (int64_t)c.qs);
    } else {
      slope = -1 * slope;
      intercept = (double)((int64_t)c.qs * c.ts - (int64_t)c.qe * c.te) / ((int64_t)c.qs - (int64_t)c.qe);
    }
    strand = c.strand;
  }
  int operator()(const std::pair<uint32_t, bool>& a, const std::pair<uint32_t, bool>& b) {
    if (a.second == b.second && a.second == 0) return a.first < b.first;
    else if (a.second == b.second && a.second == 1) {
      if (strand == 0) return a.first < b.first;
      else return a.first > b.first;
    } else if (a.second == 0 && b.second == 1) {
      if (strand == 0) return a.first * slope + intercept < (double)b.first;
      else return a.first * slope + intercept > (double)b.first;
    } else {
      if (strand == 0) return (double)a.first < b.first * slope + intercept;
      else return (double)a.first > b.first * slope + intercept;
    }
  }
};

inline uint32_t to_pos(double x) { return (uint32_t)(int64_t)x; }

}  // namespace

// One read.  In: n cluster boxes (qs, qe, ts, te, strand 0 = forward), anchorfreq, the q positions of every cluster's matches
// (matches are CartesianSort-ed: ascending q) as CSR matchOff[n+1] / matchQ, contig = (opts.readType == Options::contig),
// K = opts.globalK.  Out: clusterVal[n] (Cluster::Val of the originals), clusterSplit[n]; the split clusters in push order
// (out* arrays of capacity maxOut): box, strand, coarse (index of the original), Val, NumofAnchors0.
// Returns the number of split clusters, or -(needed) when maxOut is too small.
extern "C" int oracle_split_clusters(int n, const uint32_t* qs, const uint32_t* qe, const uint32_t* ts, const uint32_t* te, const uint8_t* strand,
                                     const float* anchorfreq, const int* matchOff, const uint32_t* matchQ, int contig, int K, int* clusterVal,
                                     uint8_t* clusterSplit, int maxOut, uint32_t* oqs, uint32_t* oqe, uint32_t* ots, uint32_t* ote, uint8_t* ostrand,
                                     int* ocoarse, int* oval, int* onum) {
  struct Out { Box b; int coarse; int Val = 0; int NumofAnchors0 = 0; };
  std::vector<Out> sp;
  std::vector<Box> cl(n);
  for (int m = 0; m < n; m++) cl[m] = Box{qs[m], qe[m], ts[m], te[m], (int)strand[m]};
  auto push = [&](uint32_t a, uint32_t b, uint32_t c, uint32_t d, int s, int m) { Out o; o.b = Box{a, b, c, d, s}; o.coarse = m; sp.push_back(o); };
  std::set<uint32_t> qSet, tSet;
  std::vector<uint8_t> split(n);
  for (int m = 0; m < n; m++) {                                           // :69-98
    const uint32_t span = std::max(cl[m].te - cl[m].ts, cl[m].qe - cl[m].qs);
    if (contig && (anchorfreq[m] <= 3.0f || (anchorfreq[m] <= 5.0f && span <= 2000))) split[m] = 1;
    else if (contig) { split[m] = 0; push(cl[m].qs, cl[m].qe, cl[m].ts, cl[m].te, cl[m].strand, m); }
    else split[m] = 1;
    if (split[m]) { qSet.insert(cl[m].qs); qSet.insert(cl[m].qe); tSet.insert(cl[m].ts); tSet.insert(cl[m].te); }
  }
  for (int m = 0; m < n; m++) {                                           // :103-170
    if (clusterSplit) clusterSplit[m] = split[m];
    if (split[m] == 0) continue;
    const Box& c = cl[m];
    IntervalSet itl(c);
    for (auto it = qSet.upper_bound(c.qs), ie = qSet.lower_bound(c.qe); it != ie; ++it) itl.Set.push_back(std::make_pair(*it, false));
    for (auto it = tSet.upper_bound(c.ts), ie = tSet.lower_bound(c.te); it != ie; ++it) itl.Set.push_back(std::make_pair(*it, true));
    std::sort(itl.Set.begin(), itl.Set.end(), itl);
    std::pair<uint32_t, uint32_t> prev = c.strand == 0 ? std::make_pair(c.qs, c.ts) : std::make_pair(c.qs, c.te);
    for (auto it = itl.Set.begin(); it < itl.Set.end(); ++it) {
      if (it->second == 0) {                                              // cut on a q coordinate
        const uint32_t t = to_pos(std::ceil(itl.slope * it->first + itl.intercept));
        if (prev.first < it->first) {
          if (c.strand == 0 && it->first >= prev.first + 3 && t >= prev.second + 3) push(prev.first, it->first, prev.second, t, c.strand, m);
          else if (c.strand == 1 && it->first >= prev.first + 3 && prev.second >= t + 3) push(prev.first, it->first, t, prev.second, c.strand, m);
        } else continue;
        prev = std::make_pair(it->first, t);
      } else {                                                            // cut on a t coordinate
        const uint32_t q = to_pos(std::ceil((it->first - itl.intercept) / itl.slope));
        if (prev.first < q) {
          if (c.strand == 0 && q >= prev.first + 3 && it->first >= prev.second + 3) push(prev.first, q, prev.second, it->first, c.strand, m);
          else if (c.strand == 1 && q >= prev.first + 3 && prev.second >= it->first + 3) push(prev.first, q, it->first, prev.second, c.strand, m);
        } else continue;
        prev = std::make_pair(q, it->first);
      }
    }
    if (prev.first < c.qe) {
      if (c.strand == 0 && c.qe >= prev.first + 3 && c.te >= prev.second + 3) push(prev.first, c.qe, prev.second, c.te, c.strand, m);
      else if (c.strand == 1 && c.qe >= prev.first + 3 && prev.second >= c.ts + 3) push(prev.first, c.qe, c.ts, prev.second, c.strand, m);
    }
  }
  // DecideSplitClustersValue :176-249
  std::vector<int> Val(n, 0);
  if (!sp.empty()) {
    for (int m = 0; m < n; m++) {
      const int b = matchOff[m], sz = matchOff[m + 1] - b;
      if (sz == 0) continue;
      uint32_t cur_len = matchQ[b], MatNum = 0;
      for (int i = 0; i < sz; i++) {
        if (cur_len > matchQ[b + i]) MatNum += matchQ[b + i] + K - cur_len;
        else MatNum += K;
        cur_len = matchQ[b + i] + K;
      }
      Val[m] = (int)MatNum;
    }
    for (size_t m = 0; m < sp.size(); m++) {
      const int ic = sp[m].coarse;
      const float pika = (float)std::min(sp[m].b.qe - sp[m].b.qs, sp[m].b.te - sp[m].b.ts) /
                         (float)std::min(cl[ic].qe - cl[ic].qs, cl[ic].te - cl[ic].ts);
      sp[m].Val = (int)((int)Val[ic] * pika);                             // float product truncated into the int member
    }
    size_t m = 0, k = 1;
    int ic_m = sp[0].coarse, ic_n = 0;
    if (sp.size() > k) ic_n = sp[k].coarse;
    int matchS = 0, matchE = 0;
    while (k < sp.size()) {
      if (ic_m == ic_n) {
        const uint32_t* b = matchQ + matchOff[ic_n];
        const uint32_t* e = matchQ + matchOff[ic_n + 1];
        matchE = (int)(std::lower_bound(b, e, sp[k].b.qs) - b);          // CartesianLowerBound: second.pos of the query is 0
        sp[m].NumofAnchors0 = matchE - matchS;
        matchS = matchE;
      } else {
        matchE = matchOff[ic_m + 1] - matchOff[ic_m];
        sp[m].NumofAnchors0 = matchE - matchS;
        matchS = 0;
      }
      m = k; ic_m = ic_n; k++;
      if (k < sp.size()) ic_n = sp[k].coarse;
    }
    sp[k - 1].NumofAnchors0 = (matchOff[ic_m + 1] - matchOff[ic_m]) - matchS;
  }
  for (int m = 0; m < n; m++) if (clusterVal) clusterVal[m] = Val[m];
  if ((int)sp.size() > maxOut) return -(int)sp.size();
  for (size_t i = 0; i < sp.size(); i++) {
    oqs[i] = sp[i].b.qs; oqe[i] = sp[i].b.qe; ots[i] = sp[i].b.ts; ote[i] = sp[i].b.te; ostrand[i] = (uint8_t)sp[i].b.strand;
    ocoarse[i] = sp[i].coarse; oval[i] = sp[i].Val; onum[i] = sp[i].NumofAnchors0;
  }
  return (int)sp.size();
}
