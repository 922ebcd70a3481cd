// lra_amd/csrc/emit.hip -- SURVEY §8a row a17: the text records.  Host code only (iostream formatting with the same libstdc++ the
// reference uses, so `float` fields print identically); kept in the library so that a caller holding device results gets the
// reference's exact SAM / PAF / BED lines without linking the reference's Alignment class.
//   Alignment::PrintBed :591-598, PrintPAF :600-656, PrintSAM :658-808, SimplePrintSAM :811-905   (Alignment.h)
#include "common.h"
#include <algorithm>
#include <sstream>
#include <string>
#include <string.h>

namespace {

int deliver(const std::string& s, char* out, uint64_t cap, uint64_t* len) {
  if (len) *len = s.size();
  if (!out || cap < s.size()) return LRA_ERR_INVALID;
  memcpy(out, s.data(), s.size());
  return LRA_OK;
}

const char* tp_of(int typeofaln) { return typeofaln == 0 ? "P" : typeofaln == 1 ? "S" : "I"; }

void clipped_cigar(std::ostream& o, const lra_aln_record& r, char clipOp) {
  if (r.pre_clip > 0) o << r.pre_clip << clipOp;
  o << (r.cigar ? r.cigar : "");
  if (r.suf_clip > 0) o << r.suf_clip << clipOp;
}

void unaligned_record(std::ostream& o, const lra_aln_record& r) {      // :663-681, :815-833
  o << "4\t*\t0\t0\t*\t*\t0\t0\t";
  o.write(r.read, r.read_len);
  o << "\t";
  if (r.qual == nullptr) o << "*";
  else o << std::string(r.qual, (size_t)r.read_len);
}

}  // namespace

// (the text appended to s_: the record writers build a read's records in one string, once)
int lra_format_bed_str(const lra_aln_record* r, std::string& s_) {
  if (!r) return LRA_ERR_INVALID;
  std::ostringstream o;
  o << r->chrom << "\t" << r->t_start << "\t" << r->t_end << "\t" << (int)(unsigned char)r->mapqv << "\t" << r->read_name << "\t" << r->read_len << "\t"
    << r->q_start << "\t" << r->q_end << "\t" << r->nm << "\t" << r->nmm << "\t" << r->nins << "\t" << r->ndel << "\t" << r->value << "\t" << r->flag << "\t"
    << r->NumOfAnchors1 << "\t" << r->NumOfAnchors1 / (float)r->read_len << std::endl;
  s_ += o.str();
  return LRA_OK;
}
extern "C" int lra_format_bed(const lra_aln_record* r, char* out, uint64_t cap, uint64_t* len) {
  std::string s_;
  const int rc = lra_format_bed_str(r, s_);
  return rc ? rc : deliver(s_, out, cap, len);
}

// (the text appended to s_: the record writers build a read's records in one string, once)
int lra_format_paf_str(const lra_aln_record* r, int print_cigar, std::string& s_) {
  if (!r) return LRA_ERR_INVALID;
  std::ostringstream o;
  const char strandChar = r->strand == 1 ? '-' : '+';
  o << r->read_name << "\t" << r->read_len << "\t";
  if (r->strand == 0) o << r->q_start << "\t" << r->q_end << "\t";
  else o << (uint32_t)((uint32_t)r->read_len - r->q_end) << "\t" << (uint32_t)((uint32_t)r->read_len - r->q_start) << "\t";
  o << strandChar << "\t" << r->chrom << "\t" << r->genome_len << "\t" << r->t_start << "\t" << r->t_end << "\t" << r->nm << "\t"
    << r->nm + r->nmm + r->ndel + r->nins << "\t" << (int)(unsigned char)r->mapqv;
  o << "\tOR:i:" << r->order << "\tNM:i:" << r->nmm + r->ndel + r->nins << "\tNX:i:" << r->nmm << "\tND:i:" << r->ndel << "\tTD:i:" << r->tdel
    << "\tNI:i:" << r->nins << "\tTI:i:" << r->tins << "\tSD:i:" << r->nSmallDel << "\tME:i:" << r->nMedDel << "\tLD:i:" << r->nLargeDel
    << "\tSI:i:" << r->nSmallIns << "\tMI:i:" << r->nMedIns << "\tLI:i:" << r->nLargeIns << "\tN0:i:" << r->NumOfAnchors0 << "\tNV:f:" << r->value
    << "\tAS:i:" << (int)r->value << "\tTP:A:" << tp_of(r->typeofaln);
  if (r->NumOfAnchors1 > 0) o << "\tNA:i:" << r->NumOfAnchors1;
  if (r->runtime > 0) o << "\tRT:i:" << r->runtime;
  if (print_cigar) { o << "\tCG:z:"; clipped_cigar(o, *r, 'S'); }
  o << std::endl;
  s_ += o.str();
  return LRA_OK;
}
extern "C" int lra_format_paf(const lra_aln_record* r, int print_cigar, char* out, uint64_t cap, uint64_t* len) {
  std::string s_;
  const int rc = lra_format_paf_str(r, print_cigar, s_);
  return rc ? rc : deliver(s_, out, cap, len);
}

// (the text appended to s_: the record writers build a read's records in one string, once)
int lra_format_sam_str(const lra_aln_record* g, int n_group, int as, int hard_clip, const char* passthrough, std::string& s_) {
  if (!g || n_group < 1 || as < 0 || as >= n_group) return LRA_ERR_INVALID;
  const lra_aln_record& r = g[as];
  // The long fields -- CIGAR, read, qualities: 40 KB of a 30 kb read's 43 KB record -- are appended to s_ as they are; the short ones go through an ostream as the
  // reference's do (the float fields print through the same libstdc++), in pieces flushed between the long fields.
  std::ostringstream o;
  auto flush = [&]() { s_ += o.str(); o.str(std::string()); };
  auto cigar = [&](const lra_aln_record& x, char clipOp) {               // clipped_cigar, appended
    if (x.pre_clip > 0) { s_ += std::to_string(x.pre_clip); s_ += clipOp; }
    if (x.cigar) s_ += x.cigar;
    if (x.suf_clip > 0) { s_ += std::to_string(x.suf_clip); s_ += clipOp; }
  };
  o << r.read_name << "\t";
  if (r.n_blocks == 0) unaligned_record(o, r);
  else {
    o << (unsigned int)r.flag << "\t" << r.chrom << "\t" << (uint32_t)(r.t_start + 1) << "\t" << (unsigned int)(unsigned char)r.mapqv << "\t";
    flush();
    cigar(r, (r.supplementary && hard_clip) ? 'H' : 'S');
    o << "\t*\t0\t" << (uint32_t)(r.t_end - r.t_start) << "\t";
    flush();
    if (!r.supplementary) s_.append(r.read, (size_t)r.read_len);
    else if (hard_clip) s_.append(r.read + r.q_start, (size_t)(r.q_end - r.q_start));
    else s_.append(r.read, (size_t)r.read_len);
    s_ += "\t";
    if (r.qual == nullptr || r.qual[0] == '*') s_ += "*";
    else if (r.supplementary && hard_clip) s_ += std::string(std::string(r.qual), r.first_block_qpos, r.last_block_qend - r.first_block_qpos);
    else s_.append(r.qual, (size_t)r.read_len);
    o << "\tNM:i:" << r.nmm + r.ndel + r.nins << "\tMM:i:" << r.nmm + r.ndel + r.nins << "\tNX:i:" << r.nmm << "\tND:i:" << r.ndel << "\tTD:i:" << r.tdel
      << "\tNI:i:" << r.nins << "\tTI:i:" << r.tins << "\tNV:f:" << r.value << "\tAS:i:" << (int)r.value << "\tAO:i:" << r.order
      << "\tN0:i:" << r.NumOfAnchors0 << "\tRT:i:" << r.runtime << "\tTP:A:" << tp_of(r.typeofaln)
      << "\tSD:i:" << r.nSmallDel << "\tME:i:" << r.nMedDel << "\tLD:i:" << r.nLargeDel << "\tSI:i:" << r.nSmallIns << "\tMI:i:" << r.nMedIns
      << "\tLI:i:" << r.nLargeIns;
    if (r.md) o << "\tMD:Z:" << r.md;                                    // opts.printMD (:763-767); the string comes from lra_md_string
    if (n_group > 1) o << "\tSA:Z:";
    for (int ag = n_group - 1; ag >= 0; ag--) {
      if (ag == as) continue;
      o << g[ag].chrom << "," << (uint32_t)(g[ag].t_start + 1) << "," << (g[ag].strand == 0 ? "+" : "-") << ",";
      flush();
      cigar(g[ag], 'S');
      o << "," << (unsigned int)(unsigned char)g[ag].mapqv << "," << (int)g[ag].nm << ";";
    }
  }
  if (passthrough) o << "\t" << passthrough;
  o << std::endl;
  flush();
  return LRA_OK;
}
extern "C" int lra_format_sam(const lra_aln_record* g, int n_group, int as, int hard_clip, const char* passthrough, char* out, uint64_t cap,
                              uint64_t* len) {
  std::string s_;
  const int rc = lra_format_sam_str(g, n_group, as, hard_clip, passthrough, s_);
  return rc ? rc : deliver(s_, out, cap, len);
}

// (the text appended to s_: the record writers build a read's records in one string, once)
int lra_format_sam_simple_str(const lra_aln_record* rp, int hard_clip, const char* passthrough, std::string& s_) {
  if (!rp) return LRA_ERR_INVALID;
  const lra_aln_record& r = *rp;
  std::ostringstream o;
  o << r.read_name << "\t";
  if (r.n_blocks == 0) unaligned_record(o, r);
  else {
    o << (unsigned int)r.flag << "\t" << r.chrom << "\t" << (uint32_t)(r.t_start + 1) << "\t" << (unsigned int)(unsigned char)r.mapqv << "\t";
    clipped_cigar(o, r, hard_clip ? 'H' : 'S');
    o << "\t*\t0\t" << (uint32_t)(r.t_end - r.t_start) << "\t";
    std::string qualStr;
    if (hard_clip) {
      o << std::string(r.read + r.first_block_qpos, r.last_block_qend - r.first_block_qpos);
      if (r.qual != nullptr && strncmp(r.qual, "*", 1) != 0) qualStr = std::string(r.qual + r.first_block_qpos, r.last_block_qend - r.first_block_qpos);
      else qualStr = "*";
    } else {
      o << std::string(r.read, (size_t)r.read_len);
      if (r.qual == nullptr || r.qual[0] == '*') qualStr = "*";       // the reference dereferences qual here; NULL never reaches it
      else qualStr.assign(r.qual, (size_t)r.read_len);
    }
    o << "\t";
    if (r.qual == nullptr) o << "*"; else o << qualStr;
    o << "\tRT:i:" << r.runtime << "\tNM:i:" << r.nmm + r.ndel + r.nins << "\tNX:i:" << r.nmm << "\tND:i:" << r.ndel << "\tTD:i:" << r.tdel
      << "\tNI:i:" << r.nins << "\tTI:i:" << r.tins << "\tN0:i:" << r.NumOfAnchors0 << "\tNV:f:" << r.value << "\tAS:i:" << (int)r.value
      << "\tAO:i:" << r.order;
  }
  if (passthrough) o << "\t" << passthrough;
  o << std::endl;
  s_ += o.str();
  return LRA_OK;
}
extern "C" int lra_format_sam_simple(const lra_aln_record* rp, int hard_clip, const char* passthrough, char* out, uint64_t cap, uint64_t* len) {
  std::string s_;
  const int rc = lra_format_sam_simple_str(rp, hard_clip, passthrough, s_);
  return rc ? rc : deliver(s_, out, cap, len);
}

// The SAM header lra writes before the first record: "@PG" (lra.cpp:665-671) and GenomeHeader::WriteSAMHeader (Genome.h:85-89).
// command_line = "lra align" followed by the argv words as lra.cpp:667-670 joins them; version = lraVersion (lra.cpp:36).
extern "C" int lra_format_sam_header(const char* version, const char* command_line, const char* const* chrom_names, const uint64_t* chrom_pos, int n_chrom,
                                     char* out, uint64_t cap, uint64_t* len) {
  if (!version || !command_line || n_chrom < 0 || (n_chrom > 0 && (!chrom_names || !chrom_pos))) return LRA_ERR_INVALID;
  std::ostringstream o;
  o << "@PG\tID:lra\tPN:lra\tVN:" << version << "\tCL:" << command_line << std::endl;
  for (int i = 0; i < n_chrom; i++) o << "@SQ\tSN:" << chrom_names[i] << "\tLN:" << chrom_pos[i + 1] - chrom_pos[i] << std::endl;
  return deliver(o.str(), out, cap, len);
}

// ---- alignment strings, MD, pairwise view (Alignment.h:204-245, :247-331, :564-589): host code over one alignment's blocks -------
namespace {
int seq_map(unsigned char c) {                                           // seqMap (SeqUtils.h:7-40): non-ACGT -> 0
  switch (c) {
    case 1: case 5: case 'C': case 'c': return 1;
    case 2: case 6: case 'G': case 'g': return 2;
    case 3: case 7: case 'T': case 't': return 3;
    default: return 0;
  }
}
void alignment_strings(const char* query, const char* text, const int32_t* B, int nb, std::string& qs, std::string& as, std::string& ts, uint32_t& refLen) {
  qs.clear(); as.clear(); ts.clear(); refLen = 0;
  if (nb == 0) return;
  uint32_t q = (uint32_t)B[0], t = (uint32_t)B[1];
  auto pair = [&]() { qs.push_back(query[q]); ts.push_back(text[t]); as.push_back(seq_map((unsigned char)query[q]) != seq_map((unsigned char)text[t]) ? '*' : '|'); q++; t++; };
  for (int b = 0; b < nb; b++) {
    for (int bl = 0; bl < B[3 * b + 2]; bl++) pair();
    if (b == nb - 1) continue;
    int queryGapLen = B[3 * (b + 1)] - B[3 * b] - B[3 * b + 2], textGapLen = B[3 * (b + 1) + 1] - B[3 * b + 1] - B[3 * b + 2];
    if (queryGapLen > 0 || textGapLen > 0) {
      int commonGapLen = queryGapLen;
      if (queryGapLen > textGapLen) commonGapLen = textGapLen;
      textGapLen -= commonGapLen; queryGapLen -= commonGapLen;
      for (int g = 0; g < queryGapLen; g++, q++) { ts.push_back('-'); as.push_back(' '); qs.push_back(query[q]); }
      for (int g = 0; g < textGapLen; g++, t++) { ts.push_back(text[t]); as.push_back(' '); qs.push_back('-'); }
      for (int g = 0; g < commonGapLen; g++) pair();
    }
  }
  refLen = t - 0;                                                        // refLen = t - refStart with refStart = 0 (:330)
}
}  // namespace

// CreateAlignmentStrings (:247-331): blocks = (qPos, tPos, length) triples; the three strings have the same length *len (each buffer
// needs cap >= *len; call once with NULL buffers for the length).  *ref_len = Alignment::refLen as the reference leaves it.
extern "C" int lra_alignment_strings(const char* query, const char* text, const int32_t* blocks, int n_blocks, char* q_out, char* a_out, char* t_out,
                                     uint64_t cap, uint64_t* len, uint32_t* ref_len) {
  if (!query || !text || n_blocks < 0 || (n_blocks > 0 && !blocks)) return LRA_ERR_INVALID;
  std::string qs, as, ts; uint32_t rl = 0;
  alignment_strings(query, text, blocks, n_blocks, qs, as, ts, rl);
  if (len) *len = qs.size();
  if (ref_len) *ref_len = rl;
  if (!q_out || !a_out || !t_out || cap < qs.size()) return qs.empty() ? LRA_OK : LRA_ERR_INVALID;
  memcpy(q_out, qs.data(), qs.size()); memcpy(a_out, as.data(), as.size()); memcpy(t_out, ts.data(), ts.size());
  return LRA_OK;
}

// AlignmentStringsToMD (:204-245), literally (including what it does at the end of the strings: std::string's terminator is read)
extern "C" int lra_md_string(const char* query_str, const char* ref_str, uint64_t n, char* out, uint64_t cap, uint64_t* len) {
  if (!query_str || !ref_str) return LRA_ERR_INVALID;
  std::string query(query_str, (size_t)n), text(ref_str, (size_t)n);
  for (size_t i = 0; i < query.size(); i++) query[i] = (char)toupper(query[i]);
  for (size_t i = 0; i < text.size(); i++) text[i] = (char)toupper(text[i]);
  std::ostringstream md;
  const std::string& cq = query; const std::string& ct = text;           // const operator[]: [size()] is the terminator
  int s = 0;
  while (s < (int)text.size()) {
    int i = s, match = 0;
    while ((i < (int)text.size() && ct[i] == cq[i]) || ct[i] == '-') { if (ct[i] == cq[i]) match++; i++; }
    md << match;
    s = i;
    if (ct[i] != cq[i] && ct[i] != '-' && cq[i] != '-') { i++; md << text.substr(s, 1); }
    else if (ct[i] != '-' && cq[i] == '-') {
      while (i < (int)text.size() && ct[i] != '-' && cq[i] == '-') i++;
      md << "^" << text.substr(s, i - s);
    }
    while (i < (int)text.size() && ct[i] == '-' && cq[i] == '=') i++;
    s = i;
  }
  return deliver(md.str(), out, cap, len);
}

// PrintPairwise (:564-589): the 50-column view; first_q / first_t = blocks[0].qPos / tPos (GetQStart / GetTStart), ref_len = refLen
extern "C" int lra_format_pairwise(const char* read_name, const char* chrom, int n_blocks, int first_q, int first_t, uint32_t ref_len, const char* query_str,
                                   const char* align_str, const char* ref_str, uint64_t n, char* out, uint64_t cap, uint64_t* len) {
  if (!read_name || !chrom || !query_str || !align_str || !ref_str) return LRA_ERR_INVALID;
  const std::string queryString(query_str, (size_t)n), alignString(align_str, (size_t)n), refString(ref_str, (size_t)n);
  std::ostringstream o;
  int i = 0, q = 0, t = 0;
  o << read_name << std::endl;
  if (n_blocks > 0) o << "Interval:\t" << chrom << ":" << first_t << "-" << (uint32_t)first_t + ref_len << std::endl;
  const int gq = n_blocks > 0 ? first_q : 0, gt = n_blocks > 0 ? first_t : 0;
  while (i < (int)queryString.size()) {
    const int end = std::min((int)queryString.size(), i + 50);
    const std::string qsub = queryString.substr(i, end - i);
    o.width(10);
    o << q + gq << " q: " << qsub << std::endl;
    q += (int)(qsub.size() - std::count(qsub.begin(), qsub.end(), '-'));
    o << "              " << alignString.substr(i, end - i) << std::endl;
    const std::string tsub = refString.substr(i, end - i);
    o.width(10);
    o << t + gt << " t: " << tsub << std::endl;
    t += (int)(tsub.size() - std::count(tsub.begin(), tsub.end(), '-'));
    o << std::endl;
    i = end;
  }
  return deliver(o.str(), out, cap, len);
}
