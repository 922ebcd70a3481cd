"""Device context: owns an lra_ctx bound to one GPU and to torch's current stream."""
import ctypes as C

import torch

from ._lib import LraError, load_library


class Context:
    """One per process / GPU.  torch is used only for device memory and streams."""

    def __init__(self, device=0):
        if not torch.cuda.is_available():
            raise LraError("no GPU visible: lra_amd runs only on an MI355X (there is no CPU fallback)")
        self.lib = load_library()
        self.device = torch.device("cuda", device)
        h = C.c_void_p()
        rc = self.lib.lra_ctx_create(device, C.byref(h))
        if rc != 0:
            raise LraError("lra_ctx_create failed (%d)" % rc)
        self.h = h
        self.bind_stream()

    @classmethod
    def borrowed(cls, handle, device):
        """A view of a context the library owns (the companion context of two-stage batches): never destroyed from here."""
        c = cls.__new__(cls)
        c.lib = load_library(); c.device = device; c.h = C.c_void_p(handle); c._borrowed = True
        return c

    def bind_stream(self, stream=None):
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        self.check(self.lib.lra_ctx_set_stream(self.h, C.c_void_p(s.cuda_stream)))

    def check(self, rc):
        if rc != 0:
            raise LraError("liblra_hip error %d: %s" % (rc, self.lib.lra_ctx_last_error(self.h).decode()))

    def to_host(self, dev_ptr, count, dtype):
        """Copy `count` items of numpy `dtype` from a raw device pointer into a new numpy array."""
        import numpy as np
        out = np.empty(int(count), dtype=dtype)
        if count:
            self.check(self.lib.lra_copy_to_host(self.h, C.c_void_p(out.ctypes.data), C.c_void_p(dev_ptr), C.c_uint64(out.nbytes)))
        return out

    def to_tensor(self, dev_ptr, count, dtype):
        """Copy `count` items from a raw device pointer into a new torch tensor on this GPU (async on the stream)."""
        out = torch.empty(int(count), dtype=dtype, device=self.device)
        if count:
            self.check(self.lib.lra_copy_device(self.h, C.c_void_p(out.data_ptr()), C.c_void_p(dev_ptr), C.c_uint64(out.numel() * out.element_size())))
        return out

    def timing(self, on=True):
        self.check(self.lib.lra_ctx_timing_enable(self.h, 1 if on else 0))

    def timing_reset(self):
        self.check(self.lib.lra_ctx_timing_reset(self.h))

    def timing_get(self, name):
        """(total_ms, launches) of the named kernel since the last reset; (0.0, 0) if it never ran."""
        ms, n = C.c_double(0), C.c_int(0)
        rc = self.lib.lra_ctx_timing_get(self.h, name.encode(), C.byref(ms), C.byref(n))
        return (ms.value, n.value) if rc == 0 else (0.0, 0)

    def release_buffers(self):
        """lra_ctx_release_buffers: the context's (and its companions') growable work buffers back to the device; the reference stays loaded.  Every result of an
        earlier call is void afterwards.  -> bytes freed"""
        n = C.c_uint64(0)
        self.check(self.lib.lra_ctx_release_buffers(self.h, C.byref(n)))
        return int(n.value)

    def close(self):
        if self.h:
            if not getattr(self, "_borrowed", False):
                self.lib.lra_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ptr(t):
    """Device pointer of a torch tensor as c_void_p (None -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    if isinstance(t, int):
        return C.c_void_p(t)            # already a raw device address (context-owned result array)
    return C.c_void_p(t.data_ptr())
