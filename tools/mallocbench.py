import ctypes, time
hip = ctypes.CDLL("libamdhip64.so")
hip.hipSetDevice(0)
p = ctypes.c_void_p()
for gb in (8, 24, 48):
    n = ctypes.c_size_t(gb << 30)
    for rep in range(3):
        t0 = time.perf_counter(); rc = hip.hipMalloc(ctypes.byref(p), n); t1 = time.perf_counter()
        hip.hipMemsetAsync(p, 0, ctypes.c_size_t(1 << 20), None); hip.hipDeviceSynchronize()
        t2 = time.perf_counter(); hip.hipFree(p); t3 = time.perf_counter()
        print(gb, "GB malloc %.2f ms free %.2f ms rc %d" % ((t1 - t0) * 1e3, (t3 - t2) * 1e3, rc))
