// lra_amd/csrc/map_state.h -- what the drivers of the path (mapread.hip: MapRead_lowacc, mapread_highacc.hip: MapRead_highacc) keep per context.
#pragma once
#include <stdlib.h>
#include "common.h"
#include <string>
#include <vector>

// Host memory for a batch's record text and its largest snapshot arrays: uninitialised (a std::string / std::vector would zero 1.4 GB from one thread first) and
// handed back to a small process-wide pool instead of the allocator -- a fresh 1.4 GB mapping costs its 350 k page faults on every batch, a reused one none.
void* lra_host_pool_get(size_t bytes, size_t* cap);
void lra_host_pool_put(void* p, size_t cap);
struct lra_text_buf {
  char* p = nullptr; size_t n = 0, cap = 0;
  lra_text_buf() = default;
  lra_text_buf(const lra_text_buf&) = delete;
  lra_text_buf& operator=(const lra_text_buf&) = delete;
  ~lra_text_buf() { clear(); }
  void alloc(size_t bytes) { clear(); if (bytes) { p = (char*)lra_host_pool_get(bytes, &cap); n = p ? bytes : 0; } }
  void clear() { if (p) lra_host_pool_put(p, cap); p = nullptr; n = 0; cap = 0; }
  void swap(lra_text_buf& o) { char* tp = p; p = o.p; o.p = tp; size_t t2 = n; n = o.n; o.n = t2; t2 = cap; cap = o.cap; o.cap = t2; }
  const char* data() const { return p; }
  char* data() { return p; }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
};
template <typename T>
struct lra_pod_buf {                                                     // the same for an array of T
  lra_text_buf b;
  bool alloc(size_t count) { b.alloc(count * sizeof(T)); return count == 0 || b.data() != nullptr; }
  size_t size() const { return b.size() / sizeof(T); }
  bool empty() const { return b.empty(); }
  T* data() { return (T*)b.data(); }
  const T* data() const { return (const T*)b.data(); }
  const T& operator[](size_t i) const { return ((const T*)b.data())[i]; }
};

struct lra_map_sig {                             // what a text of lra_map_records was made from
  const void* blocks = nullptr; const void* runs = nullptr; int32_t n_reads = 0; uint64_t n_aln = 0; int32_t fmt = 0, pna = 0, hard = 0; const char* pass = nullptr;
  int32_t flagged_unaligned = 0; const void* status = nullptr;   // (the text of a flagged read depends on both)
  bool operator==(const lra_map_sig& o) const {
    return blocks == o.blocks && runs == o.runs && n_reads == o.n_reads && n_aln == o.n_aln && fmt == o.fmt && pna == o.pna && hard == o.hard && pass == o.pass &&
           flagged_unaligned == o.flagged_unaligned && status == o.status;
  }
};
struct lra_map_state {
  std::vector<uint64_t> chrom_pos;                 // Genome::header.pos, n_chrom + 1 entries
  uint64_t* d_chrom_pos = nullptr;
  void* gli_buf = nullptr; lra_local_index_result gli{};   // the genome's LocalIndex (the .gli payload), built on the device
  uint64_t* d_gso = nullptr; uint64_t n_gwin = 0;  // its seqOffsets
  int gli_window = 0, gli_k = 0, gli_w = 0;      // glIndex.localIndexWindow / k / w: what the genome's local index was built (or written by `lra index`) with
  bool borrowed = false;                           // reference data shared from another context (lra_ctx_share_reference): not freed here
  std::shared_ptr<lra_gen_cell> cell = std::make_shared<lra_gen_cell>();   // gen bumped by every loader of this context (chromosome table, local index); dead once it is destroyed
  std::shared_ptr<lra_gen_cell> owner_cell; uint64_t owner_generation = 0;  // a borrower: whose data, at which generation (see seed_state.h)
  std::vector<float> lut;                          // LogLookUpTable.h:9-15
  lra_text_buf last_text; std::vector<uint64_t> last_off; lra_map_sig last_sig;   // lra_map_records: sizing call -> filling call
};

int lra_map_count_flagged(lra_ctx* ctx, lra_map_result* out);   // mapread.hip: counters.n_flagged_reads of a finished batch
int lra_map_check_shared(lra_ctx* ctx);   // mapread.hip: borrowed reference data still current?
// RefineBreakpoint over the consecutive SegAlignments of every job (Map_lowacc.h:586-596, Map_highacc.h:723-727); mapread.hip
int lra_refine_breakpoints(lra_ctx* ctx, uint64_t nJ, uint64_t nA, const uint64_t* d_job_aln_off, const int32_t* d_strand, const uint64_t* q_off, const int32_t* q_len,
                           const uint64_t* t_off, const int64_t* t_len, const char* strands, const char* genome, lra_refine_result* fres);
