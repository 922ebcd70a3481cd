// lra_amd/csrc/input.hip -- SURVEY §8(f) row 2, the input side: FASTA / FASTQ reads into batches in the layout lra_map_reads_*_batch take, and the host-buffer
// form of the boundary.  Host code only (it lives in the library so that a C++ host binds one .so).
//
// Replaces   Input::Initialize (Input.h:87-168: a file is FASTA if it starts with '>', FASTQ if it starts with '@' and its third line with '+'),
//            Input::GetNext for those two types (Input.h:182-283: the name is the first whitespace-delimited token behind the header's first character;
//            sequence characters are upper-cased and blanks dropped; FASTA sequence lines run to the next '>' at a line start; a FASTQ record is four
//            lines, and one with an empty line among them ends the file; files are read one after the other) and
//            Input::BufferedRead (Input.h:405-421: reads are added while the batch holds fewer than maxBufferSize bases).
// BAM / SAM / CRAM input (Input.h:284-390) goes through htslib in the reference and is not built.
#include "common.h"
#include "map_state.h"
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

struct lra_reads {
  std::vector<std::string> files;
  size_t cur = 0;
  std::ifstream strm;
  int type = -1;                                   // 0 FASTA, 1 FASTQ
  bool open_ok = false;
  std::string error;                               // a record the reference would abort on: lra_reads_next_batch returns LRA_ERR_INVALID from then on
  // the current batch
  std::string seq, names, quals;
  std::vector<uint64_t> off, name_off, qual_off;
  std::vector<const char*> name_ptr, seq_ptr, qual_ptr;
  std::vector<int32_t> len;
};

namespace {

bool is_fasta(std::istream& s) { return !(s.eof() || !s.good()) && s.peek() == '>'; }
bool is_fastq(std::istream& s) {                    // Input.h:66-85: '@', and '+' opens the third line; the two lines are put back
  if (s.eof() || !s.good() || s.peek() != '@') return false;
  const std::streampos at = s.tellg();
  std::string l0, l1;
  std::getline(s, l0); std::getline(s, l1);
  const bool res = s.peek() == '+';
  s.clear(); s.seekg(at);
  return res;
}
bool open_file(lra_reads* r) {                      // Input.h:87-168 without the htslib branch
  r->strm.close(); r->strm.clear();
  r->strm.open(r->files[r->cur].c_str());
  if (is_fasta(r->strm)) { r->type = 0; return true; }
  if (is_fastq(r->strm)) { r->type = 1; return true; }
  r->type = -1;
  return false;
}
std::string first_token_behind_first_char(const std::string& header) {   // `nameStrm >> c >> read.name`
  std::stringstream ss(header);
  char c; std::string name;
  ss >> c >> name;
  return name;
}
void squeeze_upper(std::string& s) { size_t j = 0; for (size_t i = 0; i < s.size(); i++) if (s[i] != ' ') s[j++] = (char)toupper((unsigned char)s[i]); s.resize(j); }
void squeeze(std::string& s) { size_t j = 0; for (size_t i = 0; i < s.size(); i++) if (s[i] != ' ') s[j++] = s[i]; s.resize(j); }

// Input::GetNext for FASTA / FASTQ (Input.h:182-283)
bool get_next(lra_reads* r, std::string& name, std::string& seq, std::string& qual) {
  name.clear(); seq.clear(); qual.clear();
  if (!r->open_ok) return false;
  if (r->type == 0 && r->strm.eof()) {                                     // any more FASTA files?
    r->strm.close();
    ++r->cur;
    if (r->cur >= r->files.size() || !open_file(r)) { r->open_ok = false; return false; }
  }
  if (r->strm.eof()) return false;
  if (r->type == 0) {
    std::string header;
    std::getline(r->strm, header);
    name = first_token_behind_first_char(header);
    int c = r->strm.peek();
    while (c != EOF && c != '>') {
      std::string line;
      std::getline(r->strm, line);
      squeeze_upper(line);
      seq += line;
      c = r->strm.peek();
    }
    if (c == EOF) r->strm.get();
    return true;
  }
  std::string header, sep;
  std::getline(r->strm, header); std::getline(r->strm, seq); std::getline(r->strm, sep); std::getline(r->strm, qual);
  if (header.empty() || seq.empty() || sep.empty() || qual.empty()) {      // this file is over: the next one
    r->strm.close();
    ++r->cur;
    if (r->cur >= r->files.size() || !open_file(r)) { r->open_ok = false; return false; }
    if (r->type == 1) { std::getline(r->strm, header); std::getline(r->strm, seq); std::getline(r->strm, sep); std::getline(r->strm, qual); }
  }
  if (header.empty() || seq.empty() || sep.empty() || qual.empty()) return false;
  name = first_token_behind_first_char(header);
  squeeze_upper(seq);
  squeeze(qual);
  // the reference asserts qual.size() == seq.size() (Input.h: the FASTQ branch of GetNext) and, without asserts, hands the formatter a quality string of another
  // length than the read (it would read past it): the input ends here WITH an error -- never as a normal end of file, which would drop the rest silently
  if (qual.size() != seq.size()) {
    r->open_ok = false;
    r->error = "FASTQ record '" + name + "' of " + r->files[r->cur] + ": quality string of " + std::to_string(qual.size()) + " characters for a read of " +
               std::to_string(seq.size()) + " bases";
    return false;
  }
  return true;
}

}  // namespace

extern "C" int lra_reads_open(const char* const* files, int n_files, lra_reads** out) {
  if (!files || n_files < 1 || !out) return LRA_ERR_INVALID;
  lra_reads* r = new lra_reads();
  for (int i = 0; i < n_files; i++) r->files.push_back(files[i] ? files[i] : "");
  r->open_ok = open_file(r);
  if (!r->open_ok) { delete r; *out = nullptr; return LRA_ERR_INVALID; }   // the reference prints "Cannot determine format of input reads." and exits
  *out = r;
  return LRA_OK;
}

extern "C" void lra_reads_close(lra_reads* r) { delete r; }

extern "C" int lra_reads_next_batch(lra_reads* r, uint64_t max_bases, lra_read_batch* b) {
  if (!r || !b) return LRA_ERR_INVALID;
  memset(b, 0, sizeof *b);
  r->seq.clear(); r->names.clear(); r->quals.clear(); r->off.assign(1, 0); r->name_off.assign(1, 0); r->qual_off.assign(1, 0); r->len.clear();
  std::string name, seq, qual;
  uint64_t total = 0;
  std::vector<uint8_t> hasq;
  while (total < max_bases && get_next(r, name, seq, qual)) {             // BufferedRead :412
    r->seq += seq; r->off.push_back(r->seq.size());
    r->names += name; r->names.push_back('\0'); r->name_off.push_back(r->names.size());
    hasq.push_back(!qual.empty());
    r->quals += qual; r->quals.push_back('\0'); r->qual_off.push_back(r->quals.size());
    r->len.push_back((int32_t)seq.size());
    total += seq.size();
  }
  const size_t n = r->len.size();
  r->seq.append(64, '\0');                                                // the padding the device kernels read past the last read
  r->name_ptr.resize(n); r->seq_ptr.resize(n); r->qual_ptr.resize(n);
  for (size_t i = 0; i < n; i++) {
    r->name_ptr[i] = r->names.data() + r->name_off[i]; r->seq_ptr[i] = r->seq.data() + r->off[i];
    r->qual_ptr[i] = hasq[i] ? r->quals.data() + r->qual_off[i] : nullptr;
  }
  b->n_reads = (int32_t)n; b->total_bases = total; b->seq = r->seq.data(); b->off = r->off.data(); b->read_len = r->len.data();
  b->names = r->name_ptr.data(); b->reads = r->seq_ptr.data(); b->quals = r->qual_ptr.data();
  return r->error.empty() ? LRA_OK : LRA_ERR_INVALID;                      // the batch still holds the reads in front of the bad record
}

extern "C" const char* lra_reads_last_error(const lra_reads* r) { return r ? r->error.c_str() : ""; }

// The boundary with host buffers: the reads of a batch (upper-case bases back to back, n_reads + 1 offsets; what lra_reads_next_batch returns) are copied to
// the device and mapped by the driver opts->bypassClustering selects (MapRead.h:228-240).  The device copies live in context buffers.
extern "C" int lra_map_reads_host(lra_ctx* ctx, int n_reads, const char* h_seq, const uint64_t* h_off, const lra_map_opts* opts, lra_map_result* out) {
  if (!ctx || !opts || !out || n_reads < 0 || (n_reads && (!h_seq || !h_off))) return LRA_ERR_INVALID;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const uint64_t tot = n_reads ? h_off[n_reads] : 0;
  char* d_seq = (char*)lra_ensure(ctx, 179, tot + 128);
  uint64_t* d_off = (uint64_t*)lra_ensure(ctx, 180, ((size_t)n_reads + 2) * 8);
  if (!d_seq || !d_off) return LRA_ERR_NOMEM;
  if (tot) LRA_HIP_CHECK(ctx, hipMemcpyAsync(d_seq, h_seq, tot, hipMemcpyHostToDevice, ctx->stream));
  LRA_HIP_CHECK(ctx, hipMemsetAsync(d_seq + tot, 0, 64, ctx->stream));
  if (n_reads) LRA_HIP_CHECK(ctx, hipMemcpyAsync(d_off, h_off, ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return opts->bypassClustering ? lra_map_reads_lowacc_batch(ctx, n_reads, d_seq, d_off, tot, opts, out)
                                : lra_map_reads_highacc_batch(ctx, n_reads, d_seq, d_off, tot, opts, out);
}
