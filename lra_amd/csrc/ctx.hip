// lra_amd/csrc/ctx.hip -- context, error reporting, scratch arenas.
#include "common.h"
#include <stdarg.h>

int lra_set_err(lra_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
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
  size_t want = bytes + bytes / 4 + 4096;
  void* p = nullptr;
  if (hipMalloc(&p, want) != hipSuccess) {
    lra_set_err(ctx, LRA_ERR_NOMEM, "hipMalloc(%zu) failed", want);
    return nullptr;
  }
  ctx->scratch[slot] = p;
  ctx->scratch_bytes[slot] = want;
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

extern "C" void lra_ctx_destroy(lra_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (int i = 0; i < 4; i++)
    if (ctx->scratch[i]) (void)hipFree(ctx->scratch[i]);
  delete ctx;
}

extern "C" int lra_ctx_set_stream(lra_ctx* ctx, void* stream) {
  if (!ctx) return LRA_ERR_INVALID;
  ctx->stream = (hipStream_t)stream;
  return LRA_OK;
}

extern "C" const char* lra_ctx_last_error(lra_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }
