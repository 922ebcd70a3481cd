// lra_amd/csrc/map_state.h -- what the drivers of the path (mapread.hip: MapRead_lowacc, mapread_highacc.hip: MapRead_highacc) keep per context.
#pragma once
#include "common.h"
#include <string>
#include <vector>

struct lra_map_sig {                             // what a text of lra_map_records was made from
  const void* blocks = nullptr; const void* runs = nullptr; int32_t n_reads = 0; uint64_t n_aln = 0; int32_t fmt = 0, pna = 0, hard = 0; const char* pass = nullptr;
  bool operator==(const lra_map_sig& o) const {
    return blocks == o.blocks && runs == o.runs && n_reads == o.n_reads && n_aln == o.n_aln && fmt == o.fmt && pna == o.pna && hard == o.hard && pass == o.pass;
  }
};
struct lra_map_state {
  std::vector<uint64_t> chrom_pos;                 // Genome::header.pos, n_chrom + 1 entries
  uint64_t* d_chrom_pos = nullptr;
  void* gli_buf = nullptr; lra_local_index_result gli{};   // the genome's LocalIndex (the .gli payload), built on the device
  uint64_t* d_gso = nullptr; uint64_t n_gwin = 0;  // its seqOffsets
  int gli_window = 0;
  bool borrowed = false;                           // reference data shared from another context (lra_ctx_share_reference): not freed here
  std::vector<float> lut;                          // LogLookUpTable.h:9-15
  std::string last_text; std::vector<uint64_t> last_off; lra_map_sig last_sig;   // lra_map_records: sizing call -> filling call
};

// RefineBreakpoint over the consecutive SegAlignments of every job (Map_lowacc.h:586-596, Map_highacc.h:723-727); mapread.hip
int lra_refine_breakpoints(lra_ctx* ctx, uint64_t nJ, uint64_t nA, const uint64_t* d_job_aln_off, const int32_t* d_strand, const uint64_t* q_off, const int32_t* q_len,
                           const uint64_t* t_off, const int64_t* t_len, const char* strands, const char* genome, lra_refine_result* fres);
