// oracle/switchindex.cpp -- CPU restatement of switchindex (reference: Mapping_ultility.h:39-168, called at Map_highacc.h:274) for ONE
// chain of a Primary_chain.  TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's cpu_baseline leg); the product never links it.
// Parity status: PARITY UNPINNED (Mapping_ultility.h includes the htslib-dependent headers); restated from the source.
#include <stdint.h>
#include <algorithm>
#include <map>
#include <tuple>
#include <vector>

// ch: the chain's split-cluster indices (n), link (n_link, normally n - 1), coarse[split cluster] -> cluster, qStart / qEnd of the clusters.
// Output: the rewritten chain and links; returns the new chain length, *n_link_out the new link count; -1 where the reference would read
// outside an array (a link index past the end).
extern "C" int oracle_switchindex(int n, const uint32_t* ch_in, int n_link, const uint8_t* link_in, const int32_t* coarse, const uint32_t* cl_qs,
                                  const uint32_t* cl_qe, uint32_t* ch_out, uint8_t* link_out, int* n_link_out) {
  std::vector<unsigned int> ch(ch_in, ch_in + n);
  std::vector<bool> link(link_in, link_in + n_link);
  for (int c = 0; c < n; c++) ch[c] = (unsigned int)coarse[ch[c]];                                      // :42-48
  if (link.size() > 0) {                                                                                // :52-69
    std::vector<bool> rm(link.size(), 0);
    for (int c = 1; c < (int)ch.size(); c++) if (ch[c] == ch[c - 1]) { if (c - 1 >= (int)rm.size()) return -1; rm[c - 1] = 1; }
    int sm = 0;
    for (int c = 0; c < (int)link.size(); c++) if (rm[c] == 0) { link[sm] = link[c]; sm++; }
    link.resize(sm);
  }
  ch.resize(std::distance(ch.begin(), std::unique(ch.begin(), ch.end())));                              // :73-80
  {                                                                                                     // :84-143
    std::map<int, int> appeartimes, start_pos, end_pos;
    for (int c = 0; c < (int)ch.size(); c++) {
      int ats = ch[c];
      if (appeartimes.count(ats) > 0) { appeartimes[ats] += 1; end_pos[ats] = c + 1; }
      else { appeartimes[ats] = 1; start_pos[ats] = c; end_pos[ats] = c + 1; }
    }
    if (start_pos.size() != 0) {
      std::vector<std::tuple<int, int>> start_end;
      for (auto ait = start_pos.begin(); ait != start_pos.end(); ++ait)
        if (end_pos[ait->first] > ait->second + 1) start_end.push_back(std::make_tuple(ait->second, end_pos[ait->first]));
      std::sort(start_end.begin(), start_end.end());
      std::vector<unsigned int> newch; std::vector<bool> newlink;
      int ste = 0, nc = 0;
      while (ste < (int)start_end.size()) {
        while (nc <= std::get<0>(start_end[ste])) {
          newch.push_back(ch[nc]);
          if (newch.size() > 1) { if (nc - 1 < 0 || nc - 1 >= (int)link.size()) return -1; newlink.push_back(link[nc - 1]); }
          nc++;
        }
        nc = std::get<1>(start_end[ste]);
        ste++;
      }
      while (nc < (int)ch.size()) {
        newch.push_back(ch[nc]);
        if (newch.size() > 1) { if (nc - 1 < 0 || nc - 1 >= (int)link.size()) return -1; newlink.push_back(link[nc - 1]); }
        nc++;
      }
      ch = newch; link = newlink;
    }
  }
  {                                                                                                     // :147-166
    std::vector<bool> cremove(ch.size(), 0);
    for (int c = 1; c < (int)ch.size(); c++) {
      int cr = ch[c], cp = ch[c - 1];
      if (cremove[c - 1] == 0 && cl_qs[cr] >= cl_qs[cp] && cl_qe[cr] <= cl_qe[cp]) cremove[c] = 1;
    }
    int sc = 0;
    for (int c = 0; c < (int)ch.size(); c++)
      if (cremove[c] == 0) {
        ch[sc] = ch[c];
        if (sc >= 1) { if (c - 1 >= (int)link.size() || sc - 1 >= (int)link.size()) return -1; link[sc - 1] = link[c - 1]; }
        sc++;
      }
    ch.resize(sc);
    if (sc - 1 < 0) return -1;                                           // link.resize(-1)
    link.resize(sc - 1);
  }
  for (size_t c = 0; c < ch.size(); c++) ch_out[c] = ch[c];
  for (size_t c = 0; c < link.size(); c++) link_out[c] = link[c];
  *n_link_out = (int)link.size();
  return (int)ch.size();
}
