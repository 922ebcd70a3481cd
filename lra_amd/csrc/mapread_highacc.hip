// lra_amd/csrc/mapread_highacc.hip -- the drop-in boundary of the high-accuracy path: MapRead_highacc for a batch of reads behind ONE call
// (gfx950 only).
//
// Replaces, for n_reads reads at a time, the body of
//     int MapRead_highacc(forMatches, revMatches, LookUpTable, Read&, Genome&, genomemm, glIndex, opts, output, svsigstrm, timing,
//                         indelRefineBuffers, strands, readRC, semaphore)                     (Map_highacc.h:37-798)
// and the part of MapRead in front of it (MapRead.h:169-239), which the reference enters for -CCS and -CONTIG (opts.bypassClustering == 0),
// between "the read's bases" and "its alignments with their statistics".  The per-read tail (SetFromSegAlignment, AlignmentsOrder::Update,
// SimpleMapQV, OUTPUT; Map_highacc.h:733-789) is lra_map_records, as on the low-accuracy path.
//
// Every stage is one of the library's batched entry points, called in the reference's order (the statement each call stands for is cited at
// the call).  Anchors, matches, extended clusters, chains and blocks stay in HBM from the first stage to the last.  What the reference does
// between the stages with std::vector bookkeeping -- a handful of integers per read: which clusters a chain names, renumbering them after the
// unused ones are dropped (:285-318), the (read, chain) -> job tables -- is done on the host from small downloads of the stage results
// (cluster boxes, chain index lists; a few hundred bytes per read), and goes back as index arrays.
//
// A -CCS read one of whose clusters has at most one anchor per 100 read bases ("sparse", :413-416) takes the REFINEclusters branch (:429-447) and goes on
// with K = glIndex.k.  Such reads are rare; they are collected by the first pass over the batch, run as a second, small batch through the same code
// with that branch switched on, and the two results are merged on the device into one (lra_map_reads_highacc_batch at the end of this file).
#include "common.h"
#include "seed_state.h"
#include "scan.h"
#include "map_state.h"
#include "map_merge.h"
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

namespace {
using lra_merge::PassView; using lra_merge::FROM_B; using lra_merge::al256; using lra_merge::k_gather_reads;

template <typename T>
int dl(lra_ctx* ctx, std::vector<T>& v, const T* d, size_t n) {
  v.resize(n);
  if (!n) return LRA_OK;
  if (!d) return LRA_ERR_INVALID;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(v.data(), d, n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return LRA_OK;
}
template <typename T>
T* up(lra_ctx* ctx, int slot, const std::vector<T>& v) {
  T* d = (T*)lra_ensure(ctx, slot, (v.size() + 4) * sizeof(T));
  if (!d) return nullptr;
  if (!v.empty() && hipMemcpyAsync(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return nullptr;
  // the host vector may die before the copy is issued from pageable memory: wait
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) return nullptr;
  return d;
}
template <typename T>
T* room(lra_ctx* ctx, int slot, size_t n) { return (T*)lra_ensure(ctx, slot, (n + 4) * sizeof(T)); }

inline dim3 grid(uint64_t n) { return dim3((unsigned)((n + 255) / 256)); }

__global__ void k_add_off2(int n, const uint64_t* __restrict__ off, uint64_t add, uint64_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= n) out[i] = off[i];
  if (i >= 1 && i <= n) out[n + i] = off[i] + add;
}
__global__ void k_iota(uint64_t n, uint64_t* o64, int32_t* o32) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= n) { if (o64) o64[i] = i; if (o32) o32[i] = (int32_t)i; }
}
// the clusters the chains use, renumbered (Map_highacc.h:285-318), with t relative to the chromosome (:449-460): one wave per new cluster
__global__ void __launch_bounds__(64) k_gather_clusters(uint64_t n, const uint32_t* __restrict__ src, const uint64_t* __restrict__ oldOff, const uint32_t* __restrict__ oq,
                                                        const uint32_t* __restrict__ ot, const uint32_t* __restrict__ obox, const int32_t* __restrict__ ostrand,
                                                        const int32_t* __restrict__ ochrom, const float* __restrict__ ofreq, const uint64_t* __restrict__ chromPos,
                                                        const uint64_t* __restrict__ newOff, uint32_t* mq, uint32_t* mt, uint32_t* box, int32_t* strand, int32_t* chrom, float* freq) {
  const uint64_t c = blockIdx.x;
  if (c >= n) return;
  const uint32_t s = src[c];
  const int ci = ochrom[s];
  const uint32_t off = (uint32_t)chromPos[ci];
  const uint64_t a = oldOff[s], m = oldOff[s + 1] - a, d = newOff[c];
  for (uint64_t i = threadIdx.x; i < m; i += 64) { mq[d + i] = oq[a + i]; mt[d + i] = ot[a + i] - off; }
  if (threadIdx.x == 0) {
    box[4 * c] = obox[4 * s]; box[4 * c + 1] = obox[4 * s + 1]; box[4 * c + 2] = obox[4 * s + 2] - off; box[4 * c + 3] = obox[4 * s + 3] - off;
    strand[c] = ostrand[s]; chrom[c] = ci; freq[c] = ofreq[s];
  }
}
// Cluster_SameDiag entry k of extended cluster i (Clustering.h:360-390): GetqStart, GettStart, length and the GetqEnd the chain filters see
__global__ void k_sd_entries(uint64_t nItems, const uint64_t* __restrict__ groupOff, const uint32_t* __restrict__ gStart, const uint32_t* __restrict__ gEnd,
                             const uint64_t* __restrict__ anchorOff, const uint32_t* __restrict__ Q, const uint32_t* __restrict__ T, const int32_t* __restrict__ Ln,
                             const int32_t* __restrict__ strand, const uint32_t* __restrict__ entryItem, uint64_t nEntries, uint32_t* eq, uint32_t* et, int32_t* el, uint32_t* eqe) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nEntries) return;
  const uint32_t i = entryItem[g];
  const uint64_t a0 = anchorOff[i];
  const uint64_t first = a0 + gStart[g], last = a0 + gEnd[g] - 1;
  const uint32_t qs = Q[first], ql = Q[last] + (uint32_t)Ln[last];
  const int len = ql >= qs ? (int)(ql - qs) : 0;
  eq[g] = qs; el[g] = len; et[g] = strand[i] == 0 ? T[first] : T[last]; eqe[g] = Q[last] + (uint32_t)len;
}
__global__ void k_entry_item(uint64_t nItems, const uint64_t* __restrict__ groupOff, uint32_t* entryItem) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nItems) return;
  for (uint64_t g = groupOff[i]; g < groupOff[i + 1]; g++) entryItem[g] = (uint32_t)i;
}
// the clusters of every piece, in SplitChain::sptc order, as the sparse DP wants them
__global__ void k_piece_clusters(uint64_t nPieces, const uint64_t* __restrict__ pieceOff, const uint32_t* __restrict__ pieceJob, const uint32_t* __restrict__ sptc,
                                 const uint64_t* __restrict__ chainOff, const uint64_t* __restrict__ groupOff, const int32_t* __restrict__ strand, uint64_t* cStart,
                                 uint32_t* cCount, int32_t* cStrand, uint32_t* slotItem) {
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nPieces) return;
  const uint64_t base = chainOff[pieceJob[p]];
  for (uint64_t x = pieceOff[p]; x < pieceOff[p + 1]; x++) {
    const uint64_t item = base + sptc[x];
    cStart[x] = groupOff[item]; cCount[x] = (uint32_t)(groupOff[item + 1] - groupOff[item]); cStrand[x] = strand[item]; slotItem[x] = (uint32_t)item;
  }
}
__global__ void k_final_len(uint64_t nPieces, int na2, const uint32_t* __restrict__ nChains, const uint32_t* __restrict__ chainLen, uint32_t* len) {
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < nPieces) len[p] = nChains[p] ? chainLen[p * na2] : 0;
}
// FinalChain of piece p, packed: the merged entries the second sparse DP chained (trace-back order)
__global__ void __launch_bounds__(64) k_pack_final(uint64_t nPieces, int na2, const uint64_t* __restrict__ fOff, const uint64_t* __restrict__ chainStart,
                                                   const uint32_t* __restrict__ cq, const uint32_t* __restrict__ ct, const int32_t* __restrict__ cl, const uint8_t* __restrict__ cs,
                                                   const uint32_t* __restrict__ ccl, const uint32_t* __restrict__ can, const uint64_t* __restrict__ pieceOff,
                                                   const uint32_t* __restrict__ slotItem, const uint64_t* __restrict__ groupOff, const uint32_t* __restrict__ eqe, uint32_t* fq,
                                                   uint32_t* ft, int32_t* fl, uint32_t* fqe, uint8_t* fs, int32_t* fItem, uint32_t* fEntry) {
  const uint64_t p = blockIdx.x;
  if (p >= nPieces) return;
  const uint64_t d = fOff[p], n = fOff[p + 1] - d, s = chainStart[p * na2];
  for (uint64_t i = threadIdx.x; i < n; i += 64) {
    const uint32_t item = slotItem[pieceOff[p] + ccl[s + i]], en = can[s + i];
    fq[d + i] = cq[s + i]; ft[d + i] = ct[s + i]; fl[d + i] = cl[s + i]; fs[d + i] = cs[s + i]; fItem[d + i] = (int32_t)item; fEntry[d + i] = en;
    fqe[d + i] = eqe[groupOff[item] + en];
  }
}
__global__ void k_compact_kept(uint64_t nPieces, const uint64_t* __restrict__ fOff, const uint8_t* __restrict__ keep, const uint64_t* __restrict__ kOff,
                               const int32_t* __restrict__ fItem, const uint32_t* __restrict__ fEntry, int32_t* kItem, uint32_t* kEntry) {
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nPieces) return;
  uint64_t o = kOff[p];
  for (uint64_t i = fOff[p]; i < fOff[p + 1]; i++) if (keep[i]) { kItem[o] = fItem[i]; kEntry[o] = fEntry[i]; o++; }
}
// UltimateChain (LocalRefineAlignment.h:575-576): the extended clusters' own anchors; per piece the strand / chromosome of its first anchor's
// cluster (:590-593) and the chain's value / NumOfAnchors0 (:627-629)
__global__ void k_ultimate(uint64_t nU, const uint32_t* __restrict__ anchor, const int32_t* __restrict__ cluster, const uint64_t* __restrict__ anchorOff,
                           const uint32_t* __restrict__ Q, const uint32_t* __restrict__ T, const int32_t* __restrict__ Ln, uint32_t* uq, uint32_t* ut, int32_t* ul) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nU) return;
  const uint64_t a = anchorOff[cluster[i]] + anchor[i];
  uq[i] = Q[a]; ut[i] = T[a]; ul[i] = Ln[a];
}
__global__ void k_piece_fields(uint64_t nPieces, const uint64_t* __restrict__ uOff, const int32_t* __restrict__ cluster, const int32_t* __restrict__ strand,
                               const int32_t* __restrict__ chrom, const uint32_t* __restrict__ pieceJob, const float* __restrict__ jobValue, const int32_t* __restrict__ jobN0,
                               int32_t* cStrand, int32_t* cChrom, float* cValue, int32_t* cN0, int32_t* cN1) {
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nPieces) return;
  const uint64_t a = uOff[p], n = uOff[p + 1] - a;
  const int it = n ? cluster[a] : -1;
  cStrand[p] = it >= 0 ? (strand[it] != 0) : 0; cChrom[p] = it >= 0 ? chrom[it] : 0;
  cValue[p] = jobValue[pieceJob[p]]; cN0[p] = jobN0[pieceJob[p]]; cN1[p] = (int32_t)n;
}
__global__ void k_aln_address2(uint64_t n_jobs, int num_aln, const uint64_t* __restrict__ job_aln_off, const int32_t* __restrict__ strand,
                               const int32_t* __restrict__ chrom, const uint64_t* __restrict__ read_off, uint64_t rc_base, const uint64_t* __restrict__ chrom_pos,
                               uint32_t* __restrict__ aln_read, uint64_t* __restrict__ q_off, int32_t* __restrict__ q_len, uint64_t* __restrict__ t_off,
                               int64_t* __restrict__ t_len) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_jobs) return;
  const uint32_t r = (uint32_t)(j / (uint64_t)num_aln);
  for (uint64_t a = job_aln_off[j]; a < job_aln_off[j + 1]; a++) {
    aln_read[a] = r;
    q_off[a] = read_off[r] + (strand[a] ? rc_base : 0);
    q_len[a] = (int32_t)(read_off[r + 1] - read_off[r]);
    const int c = chrom[a];
    t_off[a] = chrom_pos[c];
    t_len[a] = (int64_t)(chrom_pos[c + 1] - chrom_pos[c]);
  }
}
// the reference calls CalculateStatistics twice (Map_highacc.h:721, :731); tdel, tins and the six size-class counters are never reset between
// the calls (Alignment.h:440-512; only the constructor zeroes them, :85-86), so the second call's values sit on top of the first call's
__global__ void k_add_counts(uint64_t nA, int32_t* counts, const int32_t* __restrict__ first) {
  const uint64_t a = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= nA) return;
  for (int k = 4; k < 12; k++) counts[18 * a + k] += first[18 * a + k];
}


}  // namespace

extern "C" void lra_map_opts_preset_ccs(lra_map_opts* o) {
  if (!o) return;
  memset(o, 0, sizeof *o);
  // -CCS (lra.cpp:306-340) over the defaults of Options.h:127-230.  globalK: the preset says 25, but `lra align` then reads the index file, and ReadIndex
  // overwrites opts.globalK with the K the index was built with (MMIndex.h:409, lra.cpp:623) -- 17 for `lra index -CCS` (lra.cpp:890-896); globalW stays 20.
  o->globalK = 17; o->globalW = 20; o->globalMaxFreq = 150;
  o->localK = 7; o->localW = 5; o->localMaxFreq = 15; o->localIndexWindow = 256;
  o->refineBand = 7; o->localMatch = 4; o->localMismatch = -3; o->localIndel = -4; o->localBand = 15;
  o->refineSpaceDist = 30000; o->anchorstoosparse = 0.005f; o->splitdist = 50000; o->window = 100;
  o->second_anchorbonus = 2.0f; o->bypassClustering = 0; o->skipBandedRefine = 0; o->refineBreakpoint = 0;
  o->clean.globalK = 17; o->clean.cleanMaxDiag = 150; o->clean.minDiagCluster = 10; o->clean.bypassClustering = 0; o->clean.cleanClustersize = 100;
  o->clean.SecondCleanMinDiagCluster = 30; o->clean.SecondCleanMaxDiag = 100; o->clean.punish_anchorfreq = 10; o->clean.anchorPerlength = 10;
  o->sdp.rate = 10.0f; o->sdp.NumAln = 2; o->sdp.alnthres = 0.7f; o->sdp.gapopen = 4.0f; o->sdp.gapextend = 15.0f; o->sdp.gaproot = 1.5f;
  o->sdp.gapCeiling1 = 2000; o->sdp.gapCeiling2 = 3000; o->sdp.mode = 0; o->sdp.globalK = 17;
  o->readType = LRA_READ_CCS; o->hardClip = 1; o->PrintNumAln = 1; o->printFormat = 's';
  o->fine.globalK = 17; o->fine.RoughClustermaxGap = 500; o->fine.maxDiag = 500; o->fine.maxGap = 400; o->fine.minClusterSize = 10; o->fine.minUniqueStretchNum = 1;
  o->fine.minUniqueStretchDist = 50;
  o->merge_dist = 100;
}

extern "C" void lra_map_opts_preset_contig(lra_map_opts* o) {
  if (!o) return;
  lra_map_opts_preset_ccs(o);
  // -CONTIG (lra.cpp:268-305): what differs from -CCS on this path
  o->globalK = 19; o->globalW = 10; o->globalMaxFreq = 30; o->refineBand = 50; o->refineSpaceDist = 50000;
  o->clean.globalK = 19; o->clean.minDiagCluster = 30;
  o->sdp.rate = 1.0f; o->sdp.gapextend = 20.0f; o->sdp.gapCeiling1 = 3000; o->sdp.gapCeiling2 = 5000; o->sdp.globalK = 19;
  o->fine.globalK = 19; o->fine.maxDiag = 100; o->fine.maxGap = 500;
  o->readType = LRA_READ_CONTIG;
}

// One pass over a batch.  sparse_pass = false: every read whose chains' clusters are dense enough (Map_highacc.h:413-416, sparse == 0) is taken through the path with
// K = opts.globalK; the others are listed in *sparse_reads and left without chains.  sparse_pass = true (the batch holds such reads only): the clusters go through
// REFINEclusters (:429-447) and the path continues with K = glIndex.k, W = glIndex.w.  h_stat / h_reached: the host copies of d_read_status / d_job_reached.
static int highacc_core(lra_ctx* ctx, int n_reads, const char* d_seq, const uint64_t* d_read_off, uint64_t total_bases, const lra_map_opts* o, lra_map_result* out,
                        bool sparse_pass, std::vector<int>* sparse_reads, std::vector<uint32_t>& hstat, std::vector<uint8_t>& h_reached) {
  memset(out, 0, sizeof *out);
  lra_map_state* m = ctx->map;
  out->n_reads = n_reads;
  hipStream_t st = ctx->stream;
  const int R = n_reads, K = o->globalK, W = o->globalW;
  const int Kt = sparse_pass ? o->localK : K, Wt = sparse_pass ? o->localW : W;   // K, W after :466-468
  const uint64_t* CH = m->chrom_pos.data();
  const int nChr = (int)m->chrom_pos.size() - 1;
  const char* genome = (const char*)ctx->seed->genome;
  const uint64_t tot = total_bases;
  int rc;
  hstat.assign((size_t)R, 0);                                             // every stage's LRA_ST_* bits per read
  std::vector<uint64_t> h_read_off;
  if ((rc = dl(ctx, h_read_off, d_read_off, (size_t)R + 1))) return rc;
  // ---- a1-a4 (MapRead.h:169-203), a5 (Map_highacc.h:41-42)
  lra_seed_result sres;
  {
    const bool ahead = ctx->ahead.valid && ctx->ahead.n_reads == R && ctx->ahead.d_seq == d_seq && ctx->ahead.d_read_off == d_read_off &&
                       ctx->ahead.k == K && ctx->ahead.w == W && ctx->ahead.max_freq == o->globalMaxFreq;   // (lra_seed_prefetch + lra_ctx_adopt_seed)
    ctx->ahead.valid = false;
    if (ahead) sres = ctx->ahead.res;
    else if ((rc = lra_seed_batch(ctx, R, d_seq, d_read_off, K, W, o->globalMaxFreq, &sres))) return rc;
  }
  lra_cluster_result cres;
  if ((rc = lra_clean_matches_batch(ctx, &o->clean, CH, nChr, &cres))) return rc;
  lra_fine_result fc;
  if ((rc = lra_fine_clusters_batch(ctx, &cres, &o->fine, CH, nChr, &fc))) return rc;
  const uint64_t nC = fc.n_clusters;
  std::vector<uint64_t> cl_off, cl_moff; std::vector<uint32_t> cl_box, st32; std::vector<int32_t> cl_strand, cl_chrom; std::vector<float> cl_freq;
  if ((rc = dl(ctx, cl_off, fc.d_cluster_off, (size_t)R + 1)) || (rc = dl(ctx, cl_moff, fc.d_match_off, nC + 1)) || (rc = dl(ctx, cl_box, fc.d_box, 4 * nC)) ||
      (rc = dl(ctx, cl_strand, fc.d_strand, nC)) || (rc = dl(ctx, cl_chrom, fc.d_chrom, nC)) || (rc = dl(ctx, cl_freq, fc.d_anchorfreq, nC)) ||
      (rc = dl(ctx, st32, fc.d_status, (size_t)R))) return rc;
  if (nC == 0) { cl_moff.assign(1, 0); }
  for (int r = 0; r < R; r++) hstat[r] |= st32[r];
  // the reads forward, then reverse complemented (strands[2], MapRead.h:166-168)
  char* both = (char*)lra_ensure(ctx, 57, 2 * tot + 64);
  if (!both) return LRA_ERR_NOMEM;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(both, d_seq, tot, hipMemcpyDeviceToDevice, st));
  LRA_HIP_CHECK(ctx, hipMemsetAsync(both + 2 * tot, 0, 64, st));
  if ((rc = lra_create_rc_batch(ctx, R, d_seq, d_read_off, both + tot))) return rc;
  const int na = std::max(1, o->sdp.NumAln);
  const uint64_t S = (uint64_t)R * na;
  // the chains of all reads, in (read, h) order
  struct Chain { int read, h; std::vector<uint32_t> ch; std::vector<uint8_t> link; float value; int n0; };
  std::vector<Chain> chains;
  if (nC) {
    // ---- a6: SplitClusters + DecideSplitClustersValue (:153-155)
    std::vector<uint32_t> hq[4];
    for (int k = 0; k < 4; k++) { hq[k].resize(nC); for (uint64_t c = 0; c < nC; c++) hq[k][c] = cl_box[4 * c + k]; }
    uint32_t* d_b[4];
    for (int k = 0; k < 4; k++) if (!(d_b[k] = up(ctx, 100 + k, hq[k]))) return LRA_ERR_NOMEM;
    lra_split_clusters_result sc;
    if ((rc = lra_split_clusters_batch(ctx, R, fc.d_cluster_off, d_b[0], d_b[1], d_b[2], d_b[3], fc.d_strand, fc.d_anchorfreq, fc.d_match_off, fc.d_q,
                                       o->readType == LRA_READ_CONTIG, K, &sc))) return rc;
    std::vector<uint64_t> sp_off;
    if ((rc = dl(ctx, sp_off, sc.d_split_off, (size_t)R + 1))) return rc;
    // ---- a8 SDP#C (:224-229): rate halved where splitclusters.size() / clusters.size() > 20
    std::vector<float> hrate((size_t)R);
    for (int r = 0; r < R; r++) {
      const uint64_t ncl = cl_off[r + 1] - cl_off[r], nsp = sp_off[r + 1] - sp_off[r];
      float rate = o->sdp.rate;
      if (ncl && nsp / ncl > 20) rate = (float)(rate / 2.0);
      hrate[r] = rate;
    }
    float* d_rate = up(ctx, 104, hrate);
    if (!d_rate) return LRA_ERR_NOMEM;
    lra_chain_result c1;
    if ((rc = lra_sparse_dp_boxes_batch(ctx, R, sc.d_split_off, sc.d_qs, sc.d_qe, sc.d_ts, sc.d_te, sc.d_strand, sc.d_val, sc.d_num_anchors, d_read_off, d_rate, &o->sdp,
                                        &c1))) return rc;
    const int na1 = c1.num_aln;
    std::vector<uint32_t> n_ch, ch_len, ch_cl; std::vector<uint64_t> ch_start; std::vector<float> ch_val; std::vector<int32_t> ch_na; std::vector<uint8_t> ch_link;
    if ((rc = dl(ctx, n_ch, c1.d_n_chains, (size_t)R)) || (rc = dl(ctx, ch_start, c1.d_chain_start, (size_t)R * na1)) || (rc = dl(ctx, ch_len, c1.d_chain_len, (size_t)R * na1)) ||
        (rc = dl(ctx, ch_val, c1.d_chain_value, (size_t)R * na1)) || (rc = dl(ctx, ch_na, c1.d_chain_num_anchors, (size_t)R * na1)) ||
        (rc = dl(ctx, ch_cl, c1.d_chain_cluster, c1.n_frags)) || (rc = dl(ctx, ch_link, c1.d_chain_link, c1.n_frags)) || (rc = dl(ctx, st32, c1.d_status, (size_t)R))) return rc;
    for (int r = 0; r < R; r++) hstat[r] |= st32[r];
    // ---- switchindex (:274)
    std::vector<uint64_t> sw_off(1, 0), sw_sb, sw_cb; std::vector<uint32_t> sw_ch, sw_nl; std::vector<uint8_t> sw_lk;
    std::vector<int> sw_read, sw_h;
    for (int r = 0; r < R; r++) {
      if (hstat[r]) continue;
      for (uint32_t h = 0; h < n_ch[r] && (int)h < na; h++) {
        const size_t slot = (size_t)r * na1 + h;
        const uint64_t a = ch_start[slot]; const uint32_t n = ch_len[slot];
        for (uint32_t i = 0; i < n; i++) { sw_ch.push_back(ch_cl[a + i]); sw_lk.push_back(i + 1 < n ? ch_link[a + i] : 0); }
        sw_off.push_back(sw_ch.size()); sw_nl.push_back(n ? n - 1 : 0); sw_sb.push_back(sp_off[r]); sw_cb.push_back(cl_off[r]); sw_read.push_back(r); sw_h.push_back((int)h);
      }
    }
    const uint64_t nSw = sw_read.size();
    if (nSw) {
      uint64_t* d_o = up(ctx, 105, sw_off); uint32_t* d_c = up(ctx, 106, sw_ch); uint8_t* d_l = up(ctx, 107, sw_lk); uint32_t* d_n = up(ctx, 108, sw_nl);
      uint64_t* d_sb = up(ctx, 109, sw_sb); uint64_t* d_cb = up(ctx, 110, sw_cb);
      if (!d_o || !d_c || !d_l || !d_n || !d_sb || !d_cb) return LRA_ERR_NOMEM;
      lra_switchindex_result sw;
      if ((rc = lra_switchindex_batch(ctx, nSw, d_o, d_c, d_l, d_n, d_sb, d_cb, sc.d_coarse, d_b[0], d_b[1], sw_ch.size(), &sw))) return rc;
      std::vector<uint32_t> o_ch, o_n, o_nl, o_st; std::vector<uint8_t> o_lk;
      if ((rc = dl(ctx, o_ch, sw.d_ch, sw_ch.size())) || (rc = dl(ctx, o_lk, sw.d_link, sw_ch.size())) || (rc = dl(ctx, o_n, sw.d_n, nSw)) || (rc = dl(ctx, o_nl, sw.d_n_link, nSw)) ||
          (rc = dl(ctx, o_st, sw.d_status, nSw))) return rc;
      for (uint64_t c = 0; c < nSw; c++) {
        const int r = sw_read[c];
        hstat[r] |= o_st[c];
        Chain x; x.read = r; x.h = sw_h[c];
        const size_t slot = (size_t)r * na1 + x.h;
        x.value = ch_val[slot]; x.n0 = ch_na[slot];
        x.ch.assign(o_ch.begin() + sw_off[c], o_ch.begin() + sw_off[c] + o_n[c]);
        x.link.assign(o_lk.begin() + sw_off[c], o_lk.begin() + sw_off[c] + o_nl[c]);
        chains.push_back(std::move(x));
      }
    }
  }
  // ---- clusters no chain names are dropped, the rest renumbered (:285-318); sparse (:413-416)
  std::vector<uint32_t> src;                                              // new cluster -> fine cluster
  std::vector<uint64_t> new_moff(1, 0);
  {
    std::vector<Chain> kept;
    size_t i = 0;
    std::vector<int> newid;
    while (i < chains.size()) {
      const int r = chains[i].read;
      size_t j = i;
      while (j < chains.size() && chains[j].read == r) j++;
      const uint64_t c0 = cl_off[r], ncl = cl_off[r + 1] - c0;
      newid.assign(ncl, -1);
      bool ok = hstat[r] == 0;
      for (size_t x = i; x < j && ok; x++) for (uint32_t c : chains[x].ch) { if (c >= ncl) { hstat[r] |= LRA_ST_OOB_SLOT; ok = false; break; } newid[c] = 0; }
      if (ok) {
        const uint32_t L = (uint32_t)(h_read_off[r + 1] - h_read_off[r]);
        for (uint64_t c = 0; c < ncl && ok; c++)
          if (newid[c] == 0) {
            const uint64_t g = c0 + c;
            const float dens = (float)(cl_moff[g + 1] - cl_moff[g]) / (cl_box[4 * g + 1] - cl_box[4 * g]);
            if (dens <= 0.01f && L <= 50000 && !sparse_pass) { sparse_reads->push_back(r); ok = false; break; }
          }
      }
      if (ok) {
        for (uint64_t c = 0; c < ncl; c++)
          if (newid[c] == 0) { newid[c] = (int)src.size(); src.push_back((uint32_t)(c0 + c)); new_moff.push_back(new_moff.back() + (cl_moff[c0 + c + 1] - cl_moff[c0 + c])); }
        for (size_t x = i; x < j; x++) { for (uint32_t& c : chains[x].ch) c = (uint32_t)newid[c]; kept.push_back(std::move(chains[x])); }
      }
      i = j;
    }
    chains.swap(kept);
  }
  const uint64_t nCh = chains.size(), nNew = src.size(), nM = new_moff.back();
  // job = (read, h) slot; chain c sits at slot chains[c].read * na + chains[c].h
  uint8_t* job_reached = (uint8_t*)lra_ensure(ctx, 82, S + 64);
  uint32_t* read_status = (uint32_t*)lra_ensure(ctx, 81, ((size_t)R + 1) * 4);
  if (!job_reached || !read_status) return LRA_ERR_NOMEM;
  h_reached.assign(S, 0);
  lra_alignments_result ares; memset(&ares, 0, sizeof ares);
  lra_refine_result fres; memset(&fres, 0, sizeof fres);
  lra_stats_result tres; memset(&tres, 0, sizeof tres);
  uint32_t* aln_read = nullptr;
  lra_map_counters& cnt = out->counters;
  cnt.n_minimizers = sres.n_minimizers; cnt.n_matches = sres.n_matches; cnt.n_clusters = nC;
  uint64_t nA = 0;
  if (nCh) {
    // ---- the refined clusters: RefinedClusters[s] = &clusters[s] with t relative to the chromosome (:449-460)
    uint32_t* d_src = up(ctx, 111, src); uint64_t* d_nmoff = up(ctx, 112, new_moff);
    uint32_t* mq = room<uint32_t>(ctx, 113, nM); uint32_t* mt = room<uint32_t>(ctx, 114, nM); uint32_t* box = room<uint32_t>(ctx, 115, 4 * nNew);
    int32_t* strand = room<int32_t>(ctx, 116, nNew); int32_t* chrom = room<int32_t>(ctx, 117, nNew); float* freq = room<float>(ctx, 118, nNew);
    if (!d_src || !d_nmoff || !mq || !mt || !box || !strand || !chrom || !freq) return LRA_ERR_NOMEM;
    uint64_t nMr = nM;
    if (!sparse_pass)
      hipLaunchKernelGGL(k_gather_clusters, dim3((unsigned)nNew), dim3(64), 0, st, nNew, d_src, fc.d_match_off, fc.d_q, fc.d_t, fc.d_box, fc.d_strand, fc.d_chrom, fc.d_anchorfreq,
                         m->d_chrom_pos, d_nmoff, mq, mt, box, strand, chrom, freq);
    else {
      // ---- REFINEclusters (:429-447): the read's two local indexes (:398-402), every cluster re-seeded window by window; anchorfreq inherited (:444)
      if (!m->gli_buf) return lra_set_err(ctx, LRA_ERR_INVALID, "a read takes the REFINEclusters branch: the genome's local index is needed (lra_ctx_build_local_index)");
      if (m->gli_window != o->localIndexWindow || m->gli_k != o->localK || m->gli_w != o->localW)
        return lra_set_err(ctx, LRA_ERR_INVALID, "the genome's local index has k = %d, w = %d, windows of %d bases; the options say %d, %d, %d (lra_map_opts_apply_local_index: glIndex.Read overrides them)",
                           m->gli_k, m->gli_w, m->gli_window, o->localK, o->localW, o->localIndexWindow);
      uint64_t* off2 = (uint64_t*)lra_ensure(ctx, 58, (2 * (size_t)R + 2) * 8);
      uint8_t* active = (uint8_t*)lra_ensure(ctx, 65, 2 * (size_t)R + 64);
      if (!off2 || !active) return LRA_ERR_NOMEM;
      hipLaunchKernelGGL(k_add_off2, dim3((R + 256) / 256), dim3(256), 0, st, R, d_read_off, tot, off2);
      LRA_HIP_CHECK(ctx, hipMemsetAsync(active, 1, 2 * (size_t)R, st));
      lra_local_index_result rli;
      if ((rc = lra_local_index_masked_batch(ctx, 2 * R, both, off2, active, o->localK, o->localW, o->localIndexWindow, o->localMaxFreq, &rli))) return rc;
      std::vector<uint64_t> rc_off((size_t)R + 1, 0), c_start(nNew); std::vector<uint32_t> c_cnt(nNew), bq[4]; std::vector<int32_t> c_str(nNew); std::vector<float> c_fr(nNew);
      for (int k = 0; k < 4; k++) bq[k].resize(nNew);
      {
        size_t c = 0;
        for (int r = 0; r < R; r++) { while (c < nNew && src[c] < cl_off[r + 1]) c++; rc_off[r + 1] = c; }
      }
      for (uint64_t c = 0; c < nNew; c++) {
        const uint32_t g = src[c];
        c_start[c] = cl_moff[g]; c_cnt[c] = (uint32_t)(cl_moff[g + 1] - cl_moff[g]); c_str[c] = cl_strand[g]; c_fr[c] = cl_freq[g];
        for (int k = 0; k < 4; k++) bq[k][c] = cl_box[4 * g + k];
      }
      uint64_t* d_rco2 = up(ctx, 166, rc_off); uint64_t* d_cs = up(ctx, 167, c_start); uint32_t* d_cc = up(ctx, 168, c_cnt);
      uint32_t* d_bq[4];
      for (int k = 0; k < 4; k++) if (!(d_bq[k] = up(ctx, 100 + k, bq[k]))) return LRA_ERR_NOMEM;
      if (!d_rco2 || !d_cs || !d_cc) return LRA_ERR_NOMEM;
      LRA_HIP_CHECK(ctx, hipMemcpyAsync(strand, c_str.data(), nNew * 4, hipMemcpyHostToDevice, st));
      LRA_HIP_CHECK(ctx, hipMemcpyAsync(freq, c_fr.data(), nNew * 4, hipMemcpyHostToDevice, st));
      LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
      lra_rsc_opts ro; ro.window = o->window; ro.smallK = o->localK; ro.K = K; ro.limitrefine = 1; ro.max_freq = o->localMaxFreq; ro.local_window = o->localIndexWindow;
      lra_refined_clusters_result rr;
      if ((rc = lra_refine_clusters_batch(ctx, R, d_rco2, d_cs, d_cc, strand, d_bq[0], d_bq[1], d_bq[2], d_bq[3], fc.d_q, fc.d_t, fc.n_matches, d_read_off, CH, nChr, &rli, m->n_gwin,
                                          m->d_gso, m->gli.d_tuple_bnd, m->gli.d_tuples, &ro, &rr))) return rc;
      std::vector<uint64_t> rmoff; std::vector<uint32_t> rst;
      if ((rc = dl(ctx, rmoff, rr.d_match_off, nNew + 1)) || (rc = dl(ctx, rst, rr.d_status, nNew))) return rc;
      nMr = rr.n_matches;
      mq = room<uint32_t>(ctx, 113, nMr); mt = room<uint32_t>(ctx, 114, nMr);
      if (!mq || !mt) return LRA_ERR_NOMEM;
      LRA_HIP_CHECK(ctx, hipMemcpyAsync(mq, rr.d_match_q, nMr * 4, hipMemcpyDeviceToDevice, st));
      LRA_HIP_CHECK(ctx, hipMemcpyAsync(mt, rr.d_match_t, nMr * 4, hipMemcpyDeviceToDevice, st));
      LRA_HIP_CHECK(ctx, hipMemcpyAsync(box, rr.d_box, nNew * 16, hipMemcpyDeviceToDevice, st));
      LRA_HIP_CHECK(ctx, hipMemcpyAsync(chrom, rr.d_chrom, nNew * 4, hipMemcpyDeviceToDevice, st));
      LRA_HIP_CHECK(ctx, hipMemcpyAsync(d_nmoff, rmoff.data(), (nNew + 1) * 8, hipMemcpyHostToDevice, st));
      LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
      // clusters REFINEclusters left without matches leave the chains (:475-487; `link` keeps its length, as in the reference)
      for (Chain& x : chains) {
        size_t cp = 0;
        for (size_t k = 0; k < x.ch.size(); k++) {
          const uint32_t c = x.ch[k];
          if (rst[c] & ~(uint32_t)LRA_ST_REJECTED) hstat[x.read] |= rst[c] & ~(uint32_t)LRA_ST_REJECTED;
          if (rmoff[c + 1] > rmoff[c]) x.ch[cp++] = c;
        }
        x.ch.resize(cp);
      }
    }
    // chains as CSR over the new cluster numbers
    std::vector<uint64_t> rco((size_t)R + 1, 0), coff(1, 0), lkoff(1, 0); std::vector<uint32_t> chv, it_cl, it_rd; std::vector<int32_t> it_pv, it_nx; std::vector<uint8_t> lkv;
    std::vector<float> jval(nCh); std::vector<int32_t> jn0(nCh);
    for (uint64_t c = 0; c < nCh; c++) {
      const Chain& x = chains[c];
      rco[x.read + 1]++;
      for (size_t k = 0; k < x.ch.size(); k++) {
        chv.push_back(x.ch[k]); it_cl.push_back(x.ch[k]); it_rd.push_back((uint32_t)x.read);
        it_pv.push_back(k > 0 ? (int32_t)x.ch[k - 1] : -1); it_nx.push_back(k + 1 < x.ch.size() ? (int32_t)x.ch[k + 1] : -1);
      }
      coff.push_back(chv.size());
      lkv.insert(lkv.end(), x.link.begin(), x.link.end()); lkv.push_back(0); lkoff.push_back(lkv.size());
      jval[c] = x.value; jn0[c] = x.n0;
      if (!x.ch.empty()) h_reached[(size_t)x.read * na + x.h] = 1;        // `alignments.resize(alignments.size() + 1)` (:698-699)
    }
    for (int r = 0; r < R; r++) rco[r + 1] += rco[r];
    const uint64_t nItems = chv.size();
    uint64_t* d_rco = up(ctx, 119, rco); uint64_t* d_coff = up(ctx, 120, coff); uint32_t* d_chv = up(ctx, 121, chv);
    if (!d_rco || !d_coff || !d_chv) return LRA_ERR_NOMEM;
    // ---- a11 caller: RefineBtwnClusters_chain over every chain (:515-520)
    lra_btwn_clusters_result bres;
    if ((rc = lra_refine_btwn_clusters_batch(ctx, R, d_rco, nCh, d_coff, d_chv, nNew, d_nmoff, nMr, mq, mt, box, strand, chrom, freq, d_read_off, both, tot, genome, CH, nChr, Kt, Wt,
                                             o->readType, o->anchorstoosparse, o->localMatch, o->localMismatch, o->localIndel, o->localMaxFreq, &bres))) return rc;
    cnt.n_btwn_problems = bres.n_problems; cnt.n_btwn_rounds = bres.n_rounds; cnt.n_refined_after_btwn = bres.n_matches;
    // ---- a7 cluster version: LinearExtend_chain (:573-582), then MergeMatchesSameDiag (:642)
    uint32_t* d_icl = up(ctx, 122, it_cl); int32_t* d_ipv = up(ctx, 123, it_pv); int32_t* d_inx = up(ctx, 124, it_nx); uint32_t* d_ird = up(ctx, 125, it_rd);
    if (!d_icl || !d_ipv || !d_inx || !d_ird) return LRA_ERR_NOMEM;
    lra_ext_clusters_result er;
    if ((rc = lra_linear_extend_clusters_batch(ctx, nItems, d_icl, d_ipv, d_inx, d_ird, nNew, bres.d_match_off, bres.n_matches, (uint32_t*)bres.d_q, (uint32_t*)bres.d_t, box,
                                               strand, chrom, freq, d_seq, d_read_off, genome, CH, nChr, 1, Kt, 1, &er))) return rc;
    lra_same_diag_result sd;
    if ((rc = lra_merge_same_diag_batch(ctx, nItems, er.d_anchor_off, er.d_q, er.d_t, er.d_len, er.d_overlap, er.d_strand, o->merge_dist, &sd))) return rc;
    if ((rc = dl(ctx, st32, sd.d_status, nItems))) return rc;
    for (uint64_t i = 0; i < nItems; i++) hstat[it_rd[i]] |= st32[i];
    cnt.n_merged_clusters = nItems; cnt.n_sdp2_anchors = sd.n_groups;
    const uint64_t nE = sd.n_groups;
    uint32_t* entryItem = room<uint32_t>(ctx, 126, nE); uint32_t* eq = room<uint32_t>(ctx, 127, nE); uint32_t* et = room<uint32_t>(ctx, 128, nE);
    int32_t* el = room<int32_t>(ctx, 129, nE); uint32_t* eqe = room<uint32_t>(ctx, 130, nE);
    if (!entryItem || !eq || !et || !el || !eqe) return LRA_ERR_NOMEM;
    hipLaunchKernelGGL(k_entry_item, grid(nItems), dim3(256), 0, st, nItems, sd.d_group_off, entryItem);
    if (nE) hipLaunchKernelGGL(k_sd_entries, grid(nE), dim3(256), 0, st, nItems, sd.d_group_off, sd.d_start, sd.d_end, er.d_anchor_off, er.d_q, er.d_t, er.d_len, er.d_strand,
                               (const uint32_t*)entryItem, nE, eq, et, el, eqe);
    // ---- a9 high-accuracy SPLITChain + LSC (:705-707)
    uint64_t* d_lkoff = up(ctx, 131, lkoff); uint8_t* d_lkv = up(ctx, 132, lkv);
    if (!d_lkoff || !d_lkv) return LRA_ERR_NOMEM;
    lra_hsplit_result hs;
    if ((rc = lra_split_chains_highacc_batch(ctx, nCh, d_coff, nItems, er.d_strand, er.d_chrom, er.d_box, d_lkoff, d_lkv, o->splitdist, &hs))) return rc;
    const uint64_t P = hs.n_pieces;
    // ---- LocalRefineAlignment, first part (LocalRefineAlignment.h:556-577): sparse DP over every piece's merged entries, the chain filters,
    // SwitchToOriginalAnchors
    uint64_t* cStart = room<uint64_t>(ctx, 133, nItems); uint32_t* cCount = room<uint32_t>(ctx, 134, nItems); int32_t* cStrand = room<int32_t>(ctx, 135, nItems);
    uint32_t* slotItem = room<uint32_t>(ctx, 136, nItems);
    const uint64_t nIo = std::max<uint64_t>(std::max<uint64_t>(P, nItems), 1);
    uint64_t* iota64 = room<uint64_t>(ctx, 137, nIo + 1); int32_t* iota32 = room<int32_t>(ctx, 138, nIo + 1);
    if (!cStart || !cCount || !cStrand || !slotItem || !iota64 || !iota32) return LRA_ERR_NOMEM;
    hipLaunchKernelGGL(k_iota, grid(nIo + 1), dim3(256), 0, st, nIo, iota64, iota32);
    if (P) hipLaunchKernelGGL(k_piece_clusters, grid(P), dim3(256), 0, st, P, hs.d_piece_off, hs.d_piece_job, hs.d_sptc, (const uint64_t*)d_coff, sd.d_group_off, er.d_strand,
                              cStart, cCount, cStrand, slotItem);
    lra_sdp_opts s2 = o->sdp; s2.mode = LRA_SDP_SINGLE_CLUSTER; s2.rate = o->second_anchorbonus;
    lra_chain_result c2;
    if ((rc = lra_sparse_dp_batch(ctx, (int)P, hs.d_piece_off, cStart, cCount, cStrand, eq, et, el, iota64, nullptr, &s2, &c2))) return rc;
    cnt.n_sdp2_entries = c2.n_subproblem_entries;
    std::vector<uint32_t> p_job;
    if ((rc = dl(ctx, p_job, hs.d_piece_job, P)) || (rc = dl(ctx, st32, c2.d_status, P))) return rc;
    for (uint64_t p = 0; p < P; p++) hstat[chains[p_job[p]].read] |= st32[p];
    const int na2 = c2.num_aln;
    uint32_t* fLen = room<uint32_t>(ctx, 139, P); uint64_t* fOff = room<uint64_t>(ctx, 140, P + 1);
    if (!fLen || !fOff) return LRA_ERR_NOMEM;
    if (P) hipLaunchKernelGGL(k_final_len, grid(P), dim3(256), 0, st, P, na2, c2.d_n_chains, c2.d_chain_len, fLen);
    if ((rc = lra_exclusive_scan<uint32_t>(ctx, (long)P, fLen, fOff))) return rc;
    uint64_t nF = 0;
    LRA_HIP_CHECK(ctx, hipMemcpyAsync(&nF, fOff + P, 8, hipMemcpyDeviceToHost, st));
    LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
    uint32_t* fq = room<uint32_t>(ctx, 141, nF); uint32_t* ft = room<uint32_t>(ctx, 142, nF); int32_t* fl = room<int32_t>(ctx, 143, nF); uint32_t* fqe = room<uint32_t>(ctx, 144, nF);
    uint8_t* fs = room<uint8_t>(ctx, 145, nF); int32_t* fItem = room<int32_t>(ctx, 146, nF); uint32_t* fEntry = room<uint32_t>(ctx, 147, nF);
    if (!fq || !ft || !fl || !fqe || !fs || !fItem || !fEntry) return LRA_ERR_NOMEM;
    if (P) hipLaunchKernelGGL(k_pack_final, dim3((unsigned)P), dim3(64), 0, st, P, na2, (const uint64_t*)fOff, c2.d_chain_start, c2.d_chain_q, c2.d_chain_t, c2.d_chain_alen,
                              c2.d_chain_strand, c2.d_chain_cluster, c2.d_chain_anchor, hs.d_piece_off, (const uint32_t*)slotItem, sd.d_group_off, (const uint32_t*)eqe, fq, ft, fl,
                              fqe, fs, fItem, fEntry);
    const int32_t ops[3] = {1, 3, 4};                                      // RemoveSmallPairedIndels, RemovePairedIndels(refineEnd = false), RemoveSpuriousAnchors (:567-571)
    lra_filter_result flt;
    if ((rc = lra_filter_chains_ex_batch(ctx, P, fOff, nF, fq, ft, fl, fqe, fs, nullptr, ops, 3, &flt))) return rc;
    uint64_t* kOff = room<uint64_t>(ctx, 148, P + 1);
    if (!kOff) return LRA_ERR_NOMEM;
    uint64_t nK = 0;
    if (P) {
      if ((rc = lra_exclusive_scan<uint32_t>(ctx, (long)P, flt.d_n_kept, kOff))) return rc;
      LRA_HIP_CHECK(ctx, hipMemcpyAsync(&nK, kOff + P, 8, hipMemcpyDeviceToHost, st));
      LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
    } else LRA_HIP_CHECK(ctx, hipMemsetAsync(kOff, 0, 8, st));
    int32_t* kItem = room<int32_t>(ctx, 149, nK); uint32_t* kEntry = room<uint32_t>(ctx, 150, nK);
    if (!kItem || !kEntry) return LRA_ERR_NOMEM;
    if (P) hipLaunchKernelGGL(k_compact_kept, grid(P), dim3(256), 0, st, P, (const uint64_t*)fOff, flt.d_keep, (const uint64_t*)kOff, (const int32_t*)fItem,
                              (const uint32_t*)fEntry, kItem, kEntry);
    lra_original_anchors_result oa; memset(&oa, 0, sizeof oa);
    if ((rc = lra_switch_to_original_anchors_batch(ctx, P, kOff, nK, kItem, kEntry, &sd, iota32, &oa))) return rc;
    const uint64_t nU = oa.n_anchors;
    uint32_t* uq = room<uint32_t>(ctx, 151, nU); uint32_t* ut = room<uint32_t>(ctx, 152, nU); int32_t* ul = room<int32_t>(ctx, 153, nU);
    int32_t* pStrand = room<int32_t>(ctx, 154, P); int32_t* pChrom = room<int32_t>(ctx, 155, P); float* pValue = room<float>(ctx, 156, P);
    int32_t* pN0 = room<int32_t>(ctx, 157, P); int32_t* pN1 = room<int32_t>(ctx, 158, P);
    float* d_jval = up(ctx, 159, jval); int32_t* d_jn0 = up(ctx, 160, jn0);
    if (!uq || !ut || !ul || !pStrand || !pChrom || !pValue || !pN0 || !pN1 || !d_jval || !d_jn0) return LRA_ERR_NOMEM;
    if (nU) hipLaunchKernelGGL(k_ultimate, grid(nU), dim3(256), 0, st, nU, oa.d_anchor, oa.d_cluster, er.d_anchor_off, er.d_q, er.d_t, er.d_len, uq, ut, ul);
    const uint64_t* uOff = P ? oa.d_chain_off : (const uint64_t*)kOff;
    if (P) hipLaunchKernelGGL(k_piece_fields, grid(P), dim3(256), 0, st, P, uOff, oa.d_cluster, er.d_strand, er.d_chrom, hs.d_piece_job, (const float*)d_jval,
                              (const int32_t*)d_jn0, pStrand, pChrom, pValue, pN0, pN1);
    // ---- jobs = (read, h) slots; the chains of a job = the pieces of its SPLITChain
    std::vector<uint64_t> jpo; std::vector<uint32_t> jlsc;
    if ((rc = dl(ctx, jpo, hs.d_job_piece_off, nCh + 1)) || (rc = dl(ctx, jlsc, hs.d_job_lsc, nCh))) return rc;
    std::vector<uint64_t> job_co(S + 1, 0); std::vector<uint32_t> job_rd(S), job_lsc(S, 0); std::vector<int32_t> job_h(S);
    {
      uint64_t c = 0;
      for (uint64_t s = 0; s < S; s++) {
        job_rd[s] = (uint32_t)(s / na); job_h[s] = (int32_t)(s % na);
        job_co[s] = c < nCh ? jpo[c] : P;
        if (c < nCh && (uint64_t)chains[c].read * na + chains[c].h == s) { job_lsc[s] = jlsc[c]; c++; }
      }
      job_co[S] = P;
      // a slot without a chain: empty range at the next chain's first piece
      for (uint64_t s = S; s-- > 0;) if (job_co[s] > job_co[s + 1]) job_co[s] = job_co[s + 1];
    }
    uint64_t* d_jco = up(ctx, 161, job_co); uint32_t* d_jrd = up(ctx, 162, job_rd); int32_t* d_jh = up(ctx, 163, job_h); uint32_t* d_jlsc = up(ctx, 164, job_lsc);
    if (!d_jco || !d_jrd || !d_jh || !d_jlsc) return LRA_ERR_NOMEM;
    // ---- a13, the walk (LocalRefineAlignment.h:577-766) with tinyOpts (:404-409, :466-467)
    lra_lra_opts lo; lo.localW = o->localW; lo.globalW = o->localW; lo.localMaxFreq = o->localMaxFreq; lo.match = o->localMatch; lo.mismatch = o->localMismatch;
    lo.indel = o->localIndel; lo.localBand = o->localBand; lo.refineBySDP = 1; lo.isOnt = (o->readType == LRA_READ_ONT || o->readType == LRA_READ_CLR) ? 1 : 0;
    lo.gapopen = o->sdp.gapopen; lo.gapextend = o->sdp.gapextend; lo.gaproot = o->sdp.gaproot; lo.gapCeiling1 = o->sdp.gapCeiling1; lo.gapCeiling2 = o->sdp.gapCeiling2;
    if ((rc = lra_local_refine_highacc_batch(ctx, S, d_jco, d_jrd, d_jh, d_jlsc, P, uOff, pStrand, pChrom, pValue, pN0, pN1, nU, uq, ut, ul, d_read_off, both, tot, genome, CH,
                                             nChr, &lo, &ares))) return rc;
    nA = ares.n_alignments;
    cnt.n_a13_blocks = ares.n_blocks; cnt.n_large_spaces = ares.n_big;
    if ((rc = dl(ctx, st32, ares.d_status, (size_t)S))) return rc;
    for (uint64_t s = 0; s < S; s++) hstat[s / na] |= st32[s];
    // ---- a14 (endAlign = true), a16, a15, a16 again on every SegAlignment (Map_highacc.h:717-732)
    aln_read = (uint32_t*)lra_ensure(ctx, 59, (nA + 1) * 4);
    uint64_t* q_off = (uint64_t*)lra_ensure(ctx, 60, (nA + 1) * 8);
    int32_t* q_len = (int32_t*)lra_ensure(ctx, 61, (nA + 1) * 4);
    uint64_t* t_off = (uint64_t*)lra_ensure(ctx, 62, (nA + 1) * 8);
    int64_t* t_len = (int64_t*)lra_ensure(ctx, 63, (nA + 1) * 8);
    if (!aln_read || !q_off || !q_len || !t_off || !t_len) return LRA_ERR_NOMEM;
    hipLaunchKernelGGL(k_aln_address2, grid(S), dim3(256), 0, st, S, na, ares.d_job_aln_off, ares.d_strand, ares.d_chrom, d_read_off, tot, (const uint64_t*)m->d_chrom_pos,
                       aln_read, q_off, q_len, t_off, t_len);
    if (nA) {
      if (o->skipBandedRefine) {
        fres.n_aln = (int)nA; fres.n_blocks = ares.n_blocks; fres.d_block_off = ares.d_block_off; fres.d_blocks = ares.d_blocks; fres.d_status = nullptr;
      } else if ((rc = lra_indel_refine_batch(ctx, (int)nA, ares.d_blocks, ares.d_block_off, ares.n_blocks, both, q_off, q_len, genome, t_off, t_len, o->refineBand,
                                              o->localMatch, o->localMismatch, o->localIndel, 1, &fres))) return rc;
      if (fres.d_status) {
        int32_t* keep = (int32_t*)lra_ensure(ctx, 64, (nA + 1) * 4);
        if (!keep) return LRA_ERR_NOMEM;
        LRA_HIP_CHECK(ctx, hipMemcpyAsync(keep, fres.d_status, nA * 4, hipMemcpyDeviceToDevice, st));
        fres.d_status = keep;
        std::vector<int32_t> fst; std::vector<uint32_t> ar;
        if ((rc = dl(ctx, fst, (const int32_t*)keep, nA)) || (rc = dl(ctx, ar, (const uint32_t*)aln_read, nA))) return rc;
        for (uint64_t a = 0; a < nA; a++) hstat[ar[a]] |= (uint32_t)fst[a];
      }
      if ((rc = lra_calculate_statistics_batch(ctx, (int)nA, fres.d_blocks, fres.d_block_off, both, q_off, q_len, genome, t_off, m->lut.data(), (int)m->lut.size(), &tres)))
        return rc;
      int32_t* first_counts = (int32_t*)lra_ensure(ctx, 165, (nA + 1) * 18 * 4);
      if (!first_counts) return LRA_ERR_NOMEM;
      LRA_HIP_CHECK(ctx, hipMemcpyAsync(first_counts, tres.d_counts, nA * 18 * 4, hipMemcpyDeviceToDevice, st));
      if (!o->refineBreakpoint &&                                         // sic: `if (opts.refineBreakpoint == false)` (:723)
          (rc = lra_refine_breakpoints(ctx, S, nA, ares.d_job_aln_off, ares.d_strand, q_off, q_len, t_off, t_len, both, genome, &fres))) return rc;
      if ((rc = lra_calculate_statistics_batch(ctx, (int)nA, fres.d_blocks, fres.d_block_off, both, q_off, q_len, genome, t_off, m->lut.data(), (int)m->lut.size(), &tres)))
        return rc;
      hipLaunchKernelGGL(k_add_counts, grid(nA), dim3(256), 0, st, nA, (int32_t*)tres.d_counts, (const int32_t*)first_counts);
    }
  }
  for (int r = 0; r < R; r++) if (hstat[r]) for (int h = 0; h < na; h++) h_reached[(size_t)r * na + h] = 0;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(job_reached, h_reached.data(), S, hipMemcpyHostToDevice, st));
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(read_status, hstat.data(), (size_t)R * 4, hipMemcpyHostToDevice, st));
  if (!nCh) {                                                             // no read reached a chain: every job is empty
    uint64_t* z = (uint64_t*)lra_ensure(ctx, 161, (S + 2) * 8);
    if (!z) return LRA_ERR_NOMEM;
    LRA_HIP_CHECK(ctx, hipMemsetAsync(z, 0, (S + 2) * 8, st));
    ares.d_job_aln_off = z; ares.d_status = (const uint32_t*)z; ares.n_jobs = S;   // (zeros serve as both)
  }
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  LRA_HIP_CHECK(ctx, hipGetLastError());
  out->num_aln = na; out->n_jobs = S; out->n_alignments = nA; out->n_blocks = fres.n_blocks; out->n_runs = tres.n_runs;
  out->d_job_aln_off = ares.d_job_aln_off; out->d_job_status = ares.d_status; out->d_job_reached = job_reached; out->d_read_status = read_status;
  out->d_aln_read = aln_read; out->d_strand = ares.d_strand; out->d_supp = ares.d_supp; out->d_secondary = ares.d_secondary; out->d_n0 = ares.d_n0; out->d_n1 = ares.d_n1;
  out->d_chrom = ares.d_chrom; out->d_first_sdp_value = ares.d_value;
  out->d_block_off = fres.d_block_off; out->d_blocks = fres.d_blocks; out->d_refine_status = fres.d_status;
  out->d_counts = tres.d_counts; out->d_value = tres.d_value; out->d_run_off = tres.d_run_off; out->d_runs = tres.d_runs;
  out->d_strands = both; out->rc_base = tot;
  cnt.n_segments = fres.n_segments; cnt.n_rows = fres.n_rows; cnt.n_cells = fres.n_cells; cnt.n_aog = fres.n_aog;
  return LRA_OK;
}

static int highacc_batch_impl(lra_ctx* ctx, int n_reads, const char* d_seq, const uint64_t* d_read_off, uint64_t total_bases, const lra_map_opts* o, lra_map_result* out);
extern "C" int lra_map_reads_highacc_batch(lra_ctx* ctx, int n_reads, const char* d_seq, const uint64_t* d_read_off, uint64_t total_bases, const lra_map_opts* o,
                                           lra_map_result* out) {
  int rc = highacc_batch_impl(ctx, n_reads, d_seq, d_read_off, total_bases, o, out);
  if (rc == LRA_OK && out && n_reads > 0) rc = lra_map_count_flagged(ctx, out);
  return rc;
}
static int highacc_batch_impl(lra_ctx* ctx, int n_reads, const char* d_seq, const uint64_t* d_read_off, uint64_t total_bases, const lra_map_opts* o, lra_map_result* out) {
  if (!ctx || !o || !out || n_reads < 0) return LRA_ERR_INVALID;
  memset(out, 0, sizeof *out);
  lra_map_state* m = ctx->map;
  if (!m || m->chrom_pos.size() < 2 || !ctx->seed || !ctx->seed->genome || !ctx->seed->idx_key)
    return lra_set_err(ctx, LRA_ERR_INVALID, "reference not loaded (genome, global index, chromosome table)");
  if (o->bypassClustering) return lra_set_err(ctx, LRA_ERR_INVALID, "lra_map_reads_highacc_batch is the path of opts.bypassClustering == 0 (-CCS, -CONTIG)");
  if (m->gli_buf && (m->gli_window != o->localIndexWindow || m->gli_k != o->localK || m->gli_w != o->localW))    // (whether or not a read of this batch takes the branch that reads glIndex)
    return lra_set_err(ctx, LRA_ERR_INVALID, "the genome's local index has k = %d, w = %d, windows of %d bases; the options say %d, %d, %d (lra_map_opts_apply_local_index: glIndex.Read overrides them)",
                       m->gli_k, m->gli_w, m->gli_window, o->localK, o->localW, o->localIndexWindow);
  { int rcs = lra_map_check_shared(ctx); if (rcs) return rcs; }
  out->n_reads = n_reads;
  m->last_text.clear(); m->last_sig = lra_map_sig{};
  if (n_reads == 0) return LRA_OK;
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int R = n_reads;
  std::vector<int> sparse;
  std::vector<uint32_t> hsA, hsB; std::vector<uint8_t> hrA, hrB;
  int rc = highacc_core(ctx, R, d_seq, d_read_off, total_bases, o, out, false, &sparse, hsA, hrA);
  if (rc || sparse.empty()) return rc;
  // ---- the reads that take the REFINEclusters branch: a second, small batch
  const int na = out->num_aln;
  const uint64_t S = (uint64_t)R * na, nA1 = out->n_alignments, nB1 = out->n_blocks, nR1 = out->n_runs;
  const lra_map_counters cntA = out->counters;
  // keep pass A's per-alignment arrays (the second pass reuses the buffers they live in)
  char* hold = (char*)lra_ensure(ctx, 170, al256((S + 2) * 8) + 9 * al256((nA1 + 2) * 4) + 2 * al256((nA1 + 2) * 8) + al256((nA1 + 1) * 72) + al256((nB1 + 1) * 12) + al256((nR1 + 1) * 4) + 4096);
  if (!hold) return LRA_ERR_NOMEM;
  char* hp = hold;
  auto keep = [&](const void* srcp, size_t bytes) -> const void* {
    char* d = hp; hp += al256(bytes + 8);
    if (srcp && bytes) (void)hipMemcpyAsync(d, srcp, bytes, hipMemcpyDeviceToDevice, st);
    return srcp ? d : nullptr;
  };
  PassView A;
  A.jo = (const uint64_t*)keep(out->d_job_aln_off, (S + 1) * 8);
  A.strand = (const int32_t*)keep(out->d_strand, nA1 * 4); A.supp = (const int32_t*)keep(out->d_supp, nA1 * 4); A.sec = (const int32_t*)keep(out->d_secondary, nA1 * 4);
  A.n0 = (const int32_t*)keep(out->d_n0, nA1 * 4); A.n1 = (const int32_t*)keep(out->d_n1, nA1 * 4); A.chrom = (const int32_t*)keep(out->d_chrom, nA1 * 4);
  A.fval = (const float*)keep(out->d_first_sdp_value, nA1 * 4); A.rstat = (const int32_t*)keep(out->d_refine_status, nA1 * 4); A.value = (const float*)keep(out->d_value, nA1 * 4);
  A.boff = (const uint64_t*)keep(out->d_block_off, (nA1 + 1) * 8); A.roff = (const uint64_t*)keep(out->d_run_off, (nA1 + 1) * 8);
  A.counts = (const int32_t*)keep(out->d_counts, nA1 * 72); A.blocks = (const int32_t*)keep(out->d_blocks, nB1 * 12); A.runs = (const uint32_t*)keep(out->d_runs, nR1 * 4);
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  std::vector<uint64_t> h_off;
  if ((rc = dl(ctx, h_off, d_read_off, (size_t)R + 1))) return rc;
  const int R2 = (int)sparse.size();
  std::vector<uint32_t> pick(sparse.begin(), sparse.end()); std::vector<uint64_t> off2((size_t)R2 + 1, 0);
  for (int i = 0; i < R2; i++) off2[i + 1] = off2[i] + (h_off[sparse[i] + 1] - h_off[sparse[i]]);
  const uint64_t tot2 = off2[R2];
  uint32_t* d_pick = up(ctx, 173, pick); uint64_t* d_off2 = up(ctx, 174, off2);
  char* d_seq2 = (char*)lra_ensure(ctx, 172, tot2 + 128);
  if (!d_pick || !d_off2 || !d_seq2) return LRA_ERR_NOMEM;
  LRA_HIP_CHECK(ctx, hipMemsetAsync(d_seq2 + tot2, 0, 64, st));
  hipLaunchKernelGGL(k_gather_reads, dim3(R2), dim3(64), 0, st, R2, (const uint32_t*)d_pick, d_read_off, d_seq, (const uint64_t*)d_off2, d_seq2);
  lra_map_result o2;
  if ((rc = highacc_core(ctx, R2, d_seq2, d_off2, tot2, o, &o2, true, nullptr, hsB, hrB))) return rc;
  if (o2.num_aln != na) return lra_set_err(ctx, LRA_ERR_INVALID, "passes disagree on NumAln");
  // ---- merge: slot (r, h) from pass B for the reads of the second batch, from pass A otherwise
  std::vector<uint64_t> srcSlot(S);
  std::vector<int> inB((size_t)R, -1);
  for (int i = 0; i < R2; i++) inB[sparse[i]] = i;
  for (int r = 0; r < R; r++)
    for (int h = 0; h < na; h++) {
      const size_t s = (size_t)r * na + h;
      if (inB[r] >= 0) { srcSlot[s] = ((uint64_t)inB[r] * na + h) | FROM_B; hrA[s] = hrB[(size_t)inB[r] * na + h] ? (uint8_t)(hrB[(size_t)inB[r] * na + h] | 2) : 0; }   // bit 1: K = glIndex.k for SimpleMapQV
      else srcSlot[s] = s;
    }
  for (int i = 0; i < R2; i++) hsA[sparse[i]] = hsB[i];
  PassView B;
  B.jo = o2.d_job_aln_off; B.strand = o2.d_strand; B.supp = o2.d_supp; B.sec = o2.d_secondary; B.n0 = o2.d_n0; B.n1 = o2.d_n1; B.chrom = o2.d_chrom; B.fval = o2.d_first_sdp_value;
  B.boff = o2.d_block_off; B.blocks = o2.d_blocks; B.rstat = o2.d_refine_status; B.counts = o2.d_counts; B.value = o2.d_value; B.roff = o2.d_run_off; B.runs = o2.d_runs;
  if (nA1 == 0) A.jo = nullptr;
  if (o2.n_alignments == 0) B.jo = nullptr;
  const uint64_t nA = nA1 + o2.n_alignments, nBk = nB1 + o2.n_blocks, nRn = nR1 + o2.n_runs;
  A.jstat = nullptr; B.jstat = nullptr;                                    // every stage's status bits are in the read's status word already
  uint64_t* d_src = up(ctx, 175, srcSlot);
  if (!d_src) return LRA_ERR_NOMEM;
  lra_map_result mo; memset(&mo, 0, sizeof mo);
  if ((rc = lra_merge::merge_passes(ctx, 171, S, na, d_src, A, B, nA, nBk, nRn, 0, nullptr, &mo))) return rc;
  // the batch's strands buffer again (the second pass overwrote it with its own reads)
  char* both = (char*)lra_ensure(ctx, 57, 2 * total_bases + 64);
  uint8_t* job_reached = (uint8_t*)lra_ensure(ctx, 82, S + 64);
  uint32_t* read_status = (uint32_t*)lra_ensure(ctx, 81, ((size_t)R + 1) * 4);
  if (!both || !job_reached || !read_status) return LRA_ERR_NOMEM;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(both, d_seq, total_bases, hipMemcpyDeviceToDevice, st));
  LRA_HIP_CHECK(ctx, hipMemsetAsync(both + 2 * total_bases, 0, 64, st));
  if ((rc = lra_create_rc_batch(ctx, R, d_seq, d_read_off, both + total_bases))) return rc;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(job_reached, hrA.data(), S, hipMemcpyHostToDevice, st));
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(read_status, hsA.data(), (size_t)R * 4, hipMemcpyHostToDevice, st));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(st));
  LRA_HIP_CHECK(ctx, hipGetLastError());
  *out = mo;
  out->n_reads = R; out->num_aln = na; out->d_job_reached = job_reached; out->d_read_status = read_status; out->d_strands = both; out->rc_base = total_bases;
  out->counters = cntA;
  out->counters.n_clusters += o2.counters.n_clusters; out->counters.n_cells += o2.counters.n_cells; out->counters.n_rows += o2.counters.n_rows;
  out->counters.n_segments += o2.counters.n_segments; out->counters.n_a13_blocks += o2.counters.n_a13_blocks;
  return LRA_OK;
}
