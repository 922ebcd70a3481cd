// lra_amd/csrc/ctx.hip -- context, error reporting, scratch arenas.
#include "common.h"
#include <thread>
#include <algorithm>
#include <stdarg.h>

int lra_set_err(lra_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  // (a failed hipMalloc leaves hipErrorOutOfMemory as the runtime's LAST ERROR; a caller that shares the process with other HIP users -- torch checks hipGetLastError
  // after its own calls -- would be told of it long after this call has returned its own code: the library reports its errors through its return values only)
  if (code == LRA_ERR_NOMEM) (void)hipGetLastError();
  return code;
}

void* lra_scratch(lra_ctx* ctx, int slot, size_t bytes) {
  if (bytes <= ctx->scratch_bytes[slot] && ctx->scratch[slot]) return ctx->scratch[slot];
  if (ctx->scratch[slot]) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(ctx->scratch[slot]);
    ctx->scratch[slot] = nullptr;
    ctx->scratch_bytes[slot] = 0;
  }
  size_t want = bytes + (bytes > (size_t(256) << 20) ? bytes / 16 : bytes / 4) + 4096;   // growth slack: 25 %, 6 % for large buffers
  void* p = nullptr;
  if (hipMalloc(&p, want) != hipSuccess) {
    lra_set_err(ctx, LRA_ERR_NOMEM, "hipMalloc(%zu) failed", want);
    return nullptr;
  }
  ctx->scratch[slot] = p;
  ctx->scratch_bytes[slot] = want;
  return p;
}

int lra_host_threads() {
  static const int cached = []() {
    long hw = (long)std::thread::hardware_concurrency();
    if (hw < 1) hw = 1;
    long quota = -1, period = 100000;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                 // cgroup v2: "<quota|max> <period>"
      char q[64] = {0};
      if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atol(q);
      fclose(f);
    } else {                                                               // cgroup v1
      if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%ld", &quota) != 1) quota = -1; fclose(g); }
      if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%ld", &period) != 1) period = 100000; fclose(g); }
    }
    if (quota > 0 && period > 0) hw = std::min(hw, std::max(1L, quota / period - 4));
    return (int)hw;
  }();
  return cached;
}
extern "C" int lra_host_thread_budget(void) { return lra_host_threads(); }

void* lra_pinned(lra_ctx* ctx, size_t bytes) {
  if (ctx->pin_buf && ctx->pin_bytes >= bytes) return ctx->pin_buf;
  if (ctx->pin_buf) { (void)hipStreamSynchronize(ctx->stream); (void)hipHostFree(ctx->pin_buf); ctx->pin_buf = nullptr; ctx->pin_bytes = 0; }
  const size_t want = bytes + bytes / 4 + 4096;
  void* p = nullptr;
  if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { lra_set_err(ctx, LRA_ERR_NOMEM, "hipHostMalloc(%zu) failed", want); return nullptr; }
  ctx->pin_buf = p; ctx->pin_bytes = want;
  return p;
}

void* lra_ensure(lra_ctx* ctx, int idx, size_t bytes) {
  if (ctx->gbuf[idx] && ctx->gbytes[idx] >= bytes) return ctx->gbuf[idx];
  if (ctx->gbuf[idx]) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(ctx->gbuf[idx]); ctx->gbuf[idx] = nullptr; ctx->gbytes[idx] = 0; }
  size_t want = bytes + (bytes > (size_t(4) << 30) ? bytes / 32 : bytes > (size_t(256) << 20) ? bytes / 16 : bytes / 4) + 4096;   // growth slack: 25 %, 6 % from 256 MB, 3 % for multi-GB buffers
  void* p = nullptr;
  if (hipMalloc(&p, want) != hipSuccess) { lra_set_err(ctx, LRA_ERR_NOMEM, "hipMalloc(%zu) failed", want); return nullptr; }
  ctx->gbuf[idx] = p; ctx->gbytes[idx] = want;
  return p;
}

extern "C" int lra_abi_version(void) { return LRA_ABI_VERSION; }

extern "C" int lra_ctx_create(int device_id, lra_ctx** out) {
  if (!out) return LRA_ERR_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev)
    return LRA_ERR_HIP;
  if (hipSetDevice(device_id) != hipSuccess) return LRA_ERR_HIP;
  lra_ctx* c = new lra_ctx();
  c->device = device_id;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) c->num_cu = prop.multiProcessorCount;
  *out = c;
  return LRA_OK;
}

bool lra_handover_idle(lra_ctx* ctx);   // mapread.hip
void lra_seed_release_batch(lra_ctx* ctx);   // seed.hip
extern "C" int lra_ctx_release_buffers(lra_ctx* ctx, uint64_t* bytes) {
  if (bytes) *bytes = 0;
  if (!ctx) return LRA_ERR_INVALID;
  if (!lra_handover_idle(ctx)) return lra_set_err(ctx, LRA_ERR_INVALID, "lra_ctx_release_buffers: a batch is between the halves of a two-stage batch (or its result has not been released)");
  LRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  uint64_t freed = 0;
  for (lra_ctx* c = ctx; c; c = c->child) {                                // the companion contexts hang off the child chain (second pass / back half, the handover sets)
    if (c->stream || c == ctx) (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < lra_ctx::N_SIDE; i++) if (c->side[i]) (void)hipStreamSynchronize(c->side[i]);
    for (int i = 0; i < 4; i++) if (c->scratch[i]) { (void)hipFree(c->scratch[i]); freed += c->scratch_bytes[i]; c->scratch[i] = nullptr; c->scratch_bytes[i] = 0; }
    for (int i = 0; i < 192; i++) if (c->gbuf[i]) { (void)hipFree(c->gbuf[i]); freed += c->gbytes[i]; c->gbuf[i] = nullptr; c->gbytes[i] = 0; }
    if (c->aux) { (void)hipFree(c->aux); freed += c->aux_bytes; c->aux = nullptr; c->aux_bytes = 0; }
    if (c->out_buf) { (void)hipFree(c->out_buf); freed += c->out_bytes; c->out_buf = nullptr; c->out_bytes = 0; }
    c->ahead.valid = false;                                                // (a seed result adopted ahead of its batch pointed into the side context's arrays: stays valid there, but the pairing is over)
    size_t f0 = 0, f1 = 0, tot = 0;
    (void)hipMemGetInfo(&f0, &tot);
    lra_seed_release_batch(c);                                             // (its arrays carry no sizes: what the device says it got back)
    (void)hipMemGetInfo(&f1, &tot);
    if (f1 > f0) freed += f1 - f0;
  }
  if (bytes) *bytes = freed;
  return LRA_OK;
}

extern "C" void lra_ctx_destroy(lra_ctx* ctx) {
  if (!ctx) return;
  if (getenv("LRA_MEM_REPORT")) {                                          // analysis: what the context holds when it goes (the growable buffers by slot, the scratch slots)
    size_t tot = 0;
    for (int i = 0; i < 192; i++) if (ctx->gbuf[i]) { tot += ctx->gbytes[i]; if (ctx->gbytes[i] >= (size_t(256) << 20)) fprintf(stderr, "[mem] ctx %p buffer %3d: %8.2f GB\n", (void*)ctx, i, ctx->gbytes[i] / 1e9); }
    for (int i = 0; i < 4; i++) if (ctx->scratch[i]) { tot += ctx->scratch_bytes[i]; fprintf(stderr, "[mem] ctx %p scratch %d: %8.2f GB\n", (void*)ctx, i, ctx->scratch_bytes[i] / 1e9); }
    fprintf(stderr, "[mem] ctx %p buffers + scratch: %.2f GB (aux %.2f GB, out %.2f GB)\n", (void*)ctx, tot / 1e9, ctx->aux_bytes / 1e9, ctx->out_bytes / 1e9);
  }
  lra_handover_free(ctx);
  if (ctx->child) { lra_ctx_destroy(ctx->child); ctx->child = nullptr; }   // borrows this context's reference: first
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (int i = 0; i < 4; i++)
    if (ctx->scratch[i]) (void)hipFree(ctx->scratch[i]);
  lra_seed_free(ctx);
  lra_cluster_free(ctx);
  lra_map_free(ctx);
  if (ctx->aux) (void)hipFree(ctx->aux);
  if (ctx->out_buf) (void)hipFree(ctx->out_buf);
  if (ctx->pin_buf) (void)hipHostFree(ctx->pin_buf);
  if (ctx->scan_tmp) (void)hipFree(ctx->scan_tmp);
  for (int i = 0; i < 192; i++) if (ctx->gbuf[i]) (void)hipFree(ctx->gbuf[i]);
  for (auto& r : ctx->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  for (auto e : ctx->free_events) (void)hipEventDestroy(e);
  for (int i = 0; i < lra_ctx::N_SIDE; i++) {
    if (ctx->side[i]) { (void)hipStreamSynchronize(ctx->side[i]); (void)hipStreamDestroy(ctx->side[i]); }
    if (ctx->ev_join[i]) (void)hipEventDestroy(ctx->ev_join[i]);
  }
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_mid) (void)hipEventDestroy(ctx->ev_mid);
  if (ctx->owns_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

hipStream_t lra_side_fork(lra_ctx* ctx, int i) {
  if (i < 0 || i >= lra_ctx::N_SIDE) return ctx->stream;
  if (!ctx->side[i]) {
    if (((ctx->low_priority || ctx->prio != 0) ? hipStreamCreateWithPriority(&ctx->side[i], hipStreamNonBlocking, ctx->prio) : hipStreamCreateWithFlags(&ctx->side[i], hipStreamNonBlocking)) != hipSuccess) {
      ctx->side[i] = nullptr; return ctx->stream;
    }
    // without its two events the side stream could not be ordered against the main one: fall back to the main stream, as when the stream itself cannot be made
    if (!ctx->ev_fork && hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess) ctx->ev_fork = nullptr;
    if (ctx->ev_fork && !ctx->ev_join[i] && hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming) != hipSuccess) ctx->ev_join[i] = nullptr;
    if (!ctx->ev_fork || !ctx->ev_join[i]) { (void)hipStreamDestroy(ctx->side[i]); ctx->side[i] = nullptr; return ctx->stream; }
  }
  (void)hipEventRecord(ctx->ev_fork, ctx->stream);
  (void)hipStreamWaitEvent(ctx->side[i], ctx->ev_fork, 0);
  return ctx->side[i];
}
void lra_side_join(lra_ctx* ctx, int i) {
  if (i < 0 || i >= lra_ctx::N_SIDE || !ctx->side[i]) return;
  (void)hipEventRecord(ctx->ev_join[i], ctx->side[i]);
  (void)hipStreamWaitEvent(ctx->stream, ctx->ev_join[i], 0);
}

// The context's side streams run at the priority of the stream it is bound to: a host that maps several sub-batches at once gives each context a stream of
// its own priority (the high one runs as if alone, the others fill what it leaves idle), and a stage's forked kernels must not jump that order.
extern "C" int lra_ctx_set_stream(lra_ctx* ctx, void* stream) {
  if (!ctx) return LRA_ERR_INVALID;
  ctx->stream = (hipStream_t)stream;
  int prio = 0;
  if (ctx->low_priority || !stream || hipStreamGetPriority(ctx->stream, &prio) != hipSuccess) return LRA_OK;   // (a second-pass context keeps the priority it was made with)
  if (prio != ctx->prio) {
    for (int i = 0; i < lra_ctx::N_SIDE; i++)
      if (ctx->side[i]) { (void)hipStreamSynchronize(ctx->side[i]); (void)hipStreamDestroy(ctx->side[i]); ctx->side[i] = nullptr; }
    ctx->prio = prio;
  }
  return LRA_OK;
}

extern "C" const char* lra_ctx_last_error(lra_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

// ---- per-kernel timing ------------------------------------------------------------------
static hipEvent_t get_event(lra_ctx* ctx) {
  if (!ctx->free_events.empty()) { hipEvent_t e = ctx->free_events.back(); ctx->free_events.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

void lra_time_begin(lra_ctx* ctx, const char* name, hipStream_t stream) {
  if (!ctx->timing) return;
  lra_time_rec r{name, get_event(ctx), get_event(ctx), stream ? stream : ctx->stream};
  (void)hipEventRecord(r.a, r.stream);
  ctx->recs.push_back(r);
}

// closes the last record opened on that stream
void lra_time_end(lra_ctx* ctx, hipStream_t stream) {
  if (!ctx->timing) return;
  const hipStream_t s = stream ? stream : ctx->stream;
  for (size_t i = ctx->recs.size(); i-- > 0;)
    if (ctx->recs[i].stream == s) { (void)hipEventRecord(ctx->recs[i].b, s); return; }
}

extern "C" int lra_ctx_timing_enable(lra_ctx* ctx, int on) {
  if (!ctx) return LRA_ERR_INVALID;
  ctx->timing = on != 0;
  for (lra_ctx* c = ctx->child; c; c = c->child) c->timing = ctx->timing;   // (the companion, and the handover contexts of two-stage batches behind it)
  return LRA_OK;
}

extern "C" int lra_ctx_timing_reset(lra_ctx* ctx) {
  if (!ctx) return LRA_ERR_INVALID;
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& r : ctx->recs) { ctx->free_events.push_back(r.a); ctx->free_events.push_back(r.b); }
  ctx->recs.clear();
  if (ctx->child) return lra_ctx_timing_reset(ctx->child);
  return LRA_OK;
}

extern "C" int lra_ctx_timing_get(lra_ctx* ctx, const char* name, double* total_ms, int* launches) {
  if (!ctx || !name) return LRA_ERR_INVALID;
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  double tot = 0; int n = 0;
  for (lra_ctx* c = ctx; c; c = c->child) {                               // a batch's second pass counts with the batch
    if (c != ctx) LRA_HIP_CHECK(ctx, hipStreamSynchronize(c->stream));
    for (auto& r : c->recs)
      if (strcmp(r.name, name) == 0) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { tot += ms; n++; }
      }
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = n;
  return n ? LRA_OK : lra_set_err(ctx, LRA_ERR_INVALID, "no timing records for kernel '%s'", name);
}

extern "C" int lra_copy_to_host(lra_ctx* ctx, void* h_dst, const void* d_src, uint64_t bytes) {
  if (!ctx) return LRA_ERR_INVALID;
  if (bytes == 0) return LRA_OK;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  LRA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return LRA_OK;
}

extern "C" int lra_copy_device(lra_ctx* ctx, void* d_dst, const void* d_src, uint64_t bytes) {
  if (!ctx) return LRA_ERR_INVALID;
  if (bytes == 0) return LRA_OK;
  LRA_HIP_CHECK(ctx, hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return LRA_OK;
}
