// lra_amd/csrc/rank.hip -- SURVEY §8a rows a16 / a17, the per-read bookkeeping between CalculateStatistics and the text records.
// Host code only (no device work: a read has one to three alignments), kept in the library next to the emitters so that a caller
// holding device results reproduces MapRead's last steps without the reference's Alignment / SegAlignmentGroup classes:
//   SegAlignmentGroup::SetFromSegAlignment   Alignment.h:944-983      AlignmentsOrder::Update / operator() / Sort   Alignment.h:1021-1061
//   SimpleMapQV                              Mapping_ultility.h:497-595
//   OUTPUT / output_unaligned                Mapping_ultility.h:445-493
#include "common.h"
#include <algorithm>
#include <cmath>
#include <string>
#include <vector>
#include <string.h>

namespace {
const unsigned READ_REVERSE = 0x10, READ_SECONDARY = 0x100, READ_SUPPLEMENTARY = 0x800;   // Alignment.h:17-19
// (int) of a float as the reference binary performs it (x86-64 cvttss2si): NaN and values outside int's range give INT_MIN.  SimpleMapQV
// reaches this with y = NumOfAnchors0 / 0 when the second group is empty (a primary chain that produced no SegAlignment, Map_lowacc.h:574).
inline int f2i(float v) { return (v != v || v >= 2147483648.0f || v < -2147483648.0f) ? (int)0x80000000 : (int)v; }
}

// SetFromSegAlignment :944-983 for group g over recs[seg_off[g] .. seg_off[g+1]); `g` must come zero-initialised the way the
// SegAlignmentGroup constructor leaves it (:925-941) -- lra_group_alignments does that.
extern "C" int lra_group_alignments(lra_aln_record* recs, const int32_t* seg_off, int n_groups, lra_aln_group* groups) {
  if (!recs || !seg_off || !groups || n_groups < 0) return LRA_ERR_INVALID;
  for (int gi = 0; gi < n_groups; gi++) {
    lra_aln_group& g = groups[gi];
    memset(&g, 0, sizeof g);
    g.first = seg_off[gi]; g.count = seg_off[gi + 1] - seg_off[gi];
    if (g.count == 0) continue;
    lra_aln_record* S = recs + g.first;
    g.is_secondary = S[0].is_secondary; g.NumOfAnchors0 = S[0].NumOfAnchors0;
    for (int s = 0; s < g.count; s++) {
      g.NumOfAnchors1 += S[s].NumOfAnchors1;
      g.q_start = std::min(g.q_start, S[s].q_start); g.q_end = std::max(g.q_end, S[s].q_end);      // (starts stay 0: the constructor's zeros)
      g.t_start = std::min(g.t_start, S[s].t_start); g.t_end = std::max(g.t_end, S[s].t_end);
      g.nm += S[s].nm; g.nmm += S[s].nmm; g.ndel += S[s].ndel; g.nins += S[s].nins;
      g.value += S[s].value;
    }
    int pry = 0;
    for (int s = 0; s < g.count; s++) if (S[s].supplementary == 0) pry++;
    if (pry == 0) S[0].supplementary = 0;
    for (int s = 0; s < g.count; s++) {
      if (s >= 1) S[s].is_secondary = g.is_secondary;
      if (S[s].strand == 1) S[s].flag |= READ_REVERSE;
      if (S[s].supplementary == 1) S[s].flag |= READ_SUPPLEMENTARY;
    }
  }
  return LRA_OK;
}

// AlignmentsOrder::Update :1021-1046: index[old_end .. n_groups) = the new groups ordered by (value, NumOfAnchors0) descending with
// std::sort; the first of them primary, the rest secondary; every secondary group's records get the SECONDARY flag and typeofaln 2
// (unless 3).  index[0 .. old_end) is left as it is (the incremental use of Map_highacc.h:737).
extern "C" int lra_order_alignments(lra_aln_group* groups, int n_groups, lra_aln_record* recs, int32_t* index, int old_end) {
  if (!groups || !recs || !index || n_groups < 0 || old_end < 0 || old_end > n_groups) return LRA_ERR_INVALID;
  if (old_end == n_groups) return LRA_OK;                               // (the reference indexes index[Oldend] unconditionally)
  for (int i = old_end; i < n_groups; i++) index[i] = i;
  std::sort(index + old_end, index + n_groups, [&](int i, int j) {
    if (groups[i].value != groups[j].value) return groups[i].value > groups[j].value;
    return groups[i].NumOfAnchors0 > groups[j].NumOfAnchors0;
  });
  groups[index[old_end]].is_secondary = 0;
  for (int i = old_end + 1; i < n_groups; i++) groups[index[i]].is_secondary = 1;
  for (int i = 0; i < n_groups; i++)
    if (groups[i].is_secondary == 1)
      for (int z = 0; z < groups[i].count; z++) {
        lra_aln_record& r = recs[groups[i].first + z];
        r.flag |= READ_SECONDARY;
        if (r.typeofaln != 3) r.typeofaln = 2;
      }
  return LRA_OK;
}

// SimpleMapQV Mapping_ultility.h:497-595.  read_type: 0 raw / clr as Options::clr, 1 ont, other values = any other type.
extern "C" int lra_simple_mapqv(const lra_aln_group* groups, const int32_t* index, int n_groups, lra_aln_record* recs, int bypass_clustering,
                                int is_clr, int is_ont, int globalK) {
  if (!groups || !index || !recs || n_groups < 0) return LRA_ERR_INVALID;
  float q_coef;
  if (bypass_clustering && is_clr) q_coef = 4.0f;
  else if (bypass_clustering && is_ont) q_coef = 30.0f;
  else q_coef = 1.0f;
  const int len = n_groups;
  auto pen = [&](const lra_aln_record& a) {
    float p;
    if (!bypass_clustering) { p = (a.NumOfAnchors0 > 20 ? 1.0f : 0.05f) * a.NumOfAnchors0; p = (a.NumOfAnchors0 >= 5 ? 1.0f : 0.1f) * p; }
    else { p = (a.NumOfAnchors0 > 10 ? 1.0f : 0.05f) * a.NumOfAnchors0; p = (a.NumOfAnchors0 >= 5 ? 1.0f : 0.02f) * p; }
    return p;
  };
  auto ident = [&](const lra_aln_record& a) {
    float identity;
    if (a.nmm + a.ndel + a.nins == 0) identity = 1.0f;
    else identity = ((float)a.nm) / (a.nmm + a.ndel + a.nins);
    return identity < 1 ? identity : 1;
  };
  for (int r = 0; r < len; r++) {
    const lra_aln_group& G = groups[index[r]];
    if (r == 0 && len == 1) {
      for (int s = G.count - 1; s >= 0; s--) {
        lra_aln_record& a = recs[G.first + s];
        const float pen_cm_1 = pen(a), identity = ident(a);
        const float l = a.value > 3 ? logf(a.value / globalK) : 0;
        long mapq;
        if (!bypass_clustering) mapq = f2i(pen_cm_1 * q_coef * l * identity);
        else mapq = f2i(pen_cm_1 * q_coef * identity);
        mapq = mapq > 0 ? mapq : 0;
        a.mapqv = (unsigned char)(mapq < 60 ? mapq : 60);
      }
    } else if (r == 0 && len > 1) {
      const lra_aln_group& N = groups[index[r + 1]];
      const float x = N.value / G.value;
      float y = 1.0f;
      for (int s = G.count - 1; s >= 0; s--) {
        lra_aln_record& a = recs[G.first + s];
        if (bypass_clustering) y = ((float)G.NumOfAnchors0) / ((float)N.NumOfAnchors0);
        const float pen_cm_1 = pen(a);
        float identity;                                                  // (:556-564: here the clamp comes after l, same values)
        if (a.nmm + a.ndel + a.nins == 0) identity = 1.0f;
        else identity = ((float)a.nm) / (a.nmm + a.ndel + a.nins);
        const float l = a.value > 3 ? logf(a.value / globalK) : 0;
        identity = identity < 1 ? identity : 1;
        long mapq;
        if (x >= 0.990f) mapq = f2i(pen_cm_1 * (1.0f - x) * y * identity);
        else if (!bypass_clustering) mapq = f2i(pen_cm_1 * q_coef * (1.0f - x) * l * y * identity);
        else mapq = f2i(pen_cm_1 * q_coef * (1.0f - x) * y * identity);
        mapq -= (int)(4.343f * logf(len) + .499f);
        mapq = mapq > 0 ? mapq : 0;
        a.mapqv = (unsigned char)(mapq < 60 ? mapq : 60);
        if (r == 0 && len == 2 && a.mapqv == 0) a.mapqv = 1;
      }
    } else {
      for (int s = G.count - 1; s >= 0; s--) recs[G.first + s].mapqv = 0;
    }
  }
  return LRA_OK;
}

// OUTPUT Mapping_ultility.h:453-493 (+ output_unaligned :445-451): the first min(n_groups, PrintNumAln) alignments in order, each one's
// segments last to first with order = size-1-s; format 's' SAM, 'b' BED, 'p' PAF, 'P' PAF with CIGAR ("pc").  unaligned_rec is used when
// the read has no alignment and read_unaligned is set (SimplePrintSAM, format 's' only).  ('a', PrintPairwise, is not built.)
// (every record's text is written once, straight into `text`; the sizing-call convention of the C entry point is kept by the wrapper below)
int lra_output_read_str(const lra_aln_group* groups, const int32_t* index, int n_groups, lra_aln_record* recs, int print_num_aln, char format, int hard_clip,
                        const char* passthrough, int read_unaligned, const lra_aln_record* unaligned_rec, std::string& text) {
  if (n_groups < 0 || (n_groups > 0 && (!groups || !index || !recs))) return LRA_ERR_INVALID;
  std::vector<char> buf;
  auto call = [&](auto&& fn) {                                          // (the pairwise format only: its C formatter sizes, then fills)
    uint64_t n = 0;
    fn((char*)nullptr, (uint64_t)0, &n);
    buf.resize((size_t)n + 1);
    const int rc = fn(buf.data(), n, &n);
    text.append(buf.data(), (size_t)n);
    return rc;
  };
  if (n_groups > 0 && groups[index[0]].count > 0) {
    const int na = std::min(n_groups, print_num_aln);
    for (int a = 0; a < na; a++) {
      const lra_aln_group& G = groups[index[a]];
      lra_aln_record* S = recs + G.first;
      for (int s = G.count - 1; s >= 0; s--) {
        S[s].order = G.count - 1 - s;
        int rc = LRA_OK;
        if (format == 'b') rc = lra_format_bed_str(&S[s], text);
        else if (format == 's') rc = lra_format_sam_str(S, G.count, s, hard_clip, passthrough, text);
        else if (format == 'p' || format == 'P') rc = lra_format_paf_str(&S[s], format == 'P', text);
        else if (format == 'a') {                                         // PrintPairwise (Alignment.h:564-589) on CreateAlignmentStrings' strings
          const lra_aln_record& R = S[s];
          if (!R.blocks || !R.strand_read || !R.chrom_text) return LRA_ERR_INVALID;
          uint64_t n = 0; uint32_t refLen = 0;
          lra_alignment_strings(R.strand_read, R.chrom_text, R.blocks, R.n_blocks, nullptr, nullptr, nullptr, 0, &n, &refLen);
          std::vector<char> qs((size_t)n + 1), as((size_t)n + 1), ts((size_t)n + 1);
          rc = lra_alignment_strings(R.strand_read, R.chrom_text, R.blocks, R.n_blocks, qs.data(), as.data(), ts.data(), n, &n, &refLen);
          if (!rc) rc = call([&](char* o, uint64_t c, uint64_t* l) {
            return lra_format_pairwise(R.read_name, R.chrom, R.n_blocks, R.n_blocks ? R.blocks[0] : 0, R.n_blocks ? R.blocks[1] : 0, refLen, qs.data(), as.data(),
                                       ts.data(), n, o, c, l);
          });
        }
        else return LRA_ERR_INVALID;
        if (rc) return rc;
      }
    }
  } else if (read_unaligned == 1) {
    if (format == 's' && unaligned_rec) {
      const int rc = lra_format_sam_simple_str(unaligned_rec, hard_clip, passthrough, text);
      if (rc) return rc;
    }
  }
  return LRA_OK;
}
extern "C" int lra_output_read(const lra_aln_group* groups, const int32_t* index, int n_groups, lra_aln_record* recs, int print_num_aln, char format,
                               int hard_clip, const char* passthrough, int read_unaligned, const lra_aln_record* unaligned_rec, char* out, uint64_t cap,
                               uint64_t* len) {
  std::string text;
  const int rc = lra_output_read_str(groups, index, n_groups, recs, print_num_aln, format, hard_clip, passthrough, read_unaligned, unaligned_rec, text);
  if (rc) return rc;
  if (len) *len = text.size();
  if (!out || cap < text.size()) return text.empty() ? LRA_OK : LRA_ERR_INVALID;
  memcpy(out, text.data(), text.size());
  return LRA_OK;
}
