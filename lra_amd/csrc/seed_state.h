// lra_amd/csrc/seed_state.h -- device-resident state shared by the seeding (seed.hip) and clustering (cluster.hip) stages.
#pragma once
#include "common.h"

struct lra_seed_state {
  bool borrowed = false;   // genome / index / directory belong to another context (lra_ctx_share_reference)
  // Ownership tracking for shared reference data: an owner bumps `generation` whenever one of its loaders replaces the genome / index / directory;
  // a borrower remembers whose data it holds and at which generation, and the batch entry points refuse to run on stale pointers (the owner
  // reloaded: share again).  A loader called on a borrower first drops the borrowed pointers (nothing of the owner's is freed) and makes the
  // context an owner of what it loads.
  // The generation lives in a refcounted cell (common.h) the borrowers hold too: the check works when the owner has been destroyed (cell->dead).
  std::shared_ptr<lra_gen_cell> cell = std::make_shared<lra_gen_cell>();
  std::shared_ptr<lra_gen_cell> owner_cell; uint64_t owner_generation = 0;
  unsigned char* genome = nullptr; uint64_t genome_len = 0;
  uint64_t* idx_key = nullptr; uint32_t* idx_pos = nullptr; uint64_t n_idx = 0;
  // batch buffers (grown on demand)
  uint32_t* counts32 = nullptr; uint64_t* counts64 = nullptr; uint64_t* mm_off = nullptr; uint64_t* match_off = nullptr;
  uint32_t* n_forward = nullptr; size_t cap_reads = 0;
  uint64_t* mm_key = nullptr; uint32_t* mm_pos = nullptr; uint32_t* lb = nullptr; uint32_t* ub = nullptr; size_t cap_mm = 0;
  uint64_t* tk_lb = nullptr; uint64_t* tk_lbm1 = nullptr; uint64_t* tk_ubm1 = nullptr;
  uint32_t* dir = nullptr; uint32_t nbuckets = 0; int dir_shift = 0;
  uint32_t* match_qi = nullptr; uint32_t* match_ti = nullptr; uint32_t* sep_qpos = nullptr; uint32_t* sep_tpos = nullptr; uint64_t* sep_qkey = nullptr; size_t cap_match = 0;
  int last_n_reads = 0; uint64_t last_n_matches = 0;   // shape of the current seed result (inputs of the clean stage)
  // lra_map_reads_lowacc_batch with opts.defer_seed_matches: a read with more tier-1 matches than defer_T leaves the batch behind CompareLists (its match list is
  // emptied, defer_flag[r] = 1); 0 = off.  Set by the driver around its lra_seed_batch call, never by the stage entry point itself.
  uint32_t defer_T = 0; uint8_t* defer_flag = nullptr; size_t cap_defer = 0;
  uint32_t* tmp_qi = nullptr; uint32_t* tmp_ti = nullptr; size_t cap_tmp = 0; uint64_t* cap_cnt = nullptr; uint64_t* cap_off = nullptr;
};


// current cluster result (cluster.hip), inputs of the linear-extension stage
struct lra_cluster_state {
  int n_reads = 0; uint64_t n_clusters = 0, n_matches = 0;
  const uint64_t* cluster_off = nullptr; const uint64_t* c_start = nullptr; const uint64_t* c_end = nullptr;
  const int* c_strand = nullptr; const int* c_chrom = nullptr;
  const uint32_t* cl_q = nullptr; const uint32_t* cl_t = nullptr;
  const uint64_t* chrom_pos = nullptr; int n_chrom = 0;
};
