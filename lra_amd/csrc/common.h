// lra_amd/csrc/common.h -- shared host-side plumbing of liblra_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include "../../include/lra_hip.h"

struct lra_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  // growable device scratch arenas (never shrunk; freed in lra_ctx_destroy)
  void* scratch[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t scratch_bytes[4] = {0, 0, 0, 0};
  int num_cu = 256;
};

int lra_set_err(lra_ctx* ctx, int code, const char* fmt, ...);
// returns a device buffer of >= bytes (slot 0..3), growing it if needed (synchronises the
// stream before freeing the old one)
void* lra_scratch(lra_ctx* ctx, int slot, size_t bytes);

#define LRA_HIP_CHECK(ctx, call)                                                         \
  do {                                                                                   \
    hipError_t e__ = (call);                                                             \
    if (e__ != hipSuccess)                                                               \
      return lra_set_err(ctx, LRA_ERR_HIP, "%s failed: %s (%s:%d)", #call,               \
                         hipGetErrorString(e__), __FILE__, __LINE__);                    \
  } while (0)
