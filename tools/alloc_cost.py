"""How long the device allocator takes for the large per-call tables (MultPoly's line tables: 38 GB for one round of
65536 coefficients at a 1024-bit key): hipMalloc / first touch / hipFree, three times each size."""
import ctypes as C
import time

hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipDeviceSynchronize.argtypes = []
print("GB,malloc_ms,memset_ms,free_ms")
for gb in (1, 8, 38, 38, 38, 69):
    p = C.c_void_p()
    t0 = time.perf_counter()
    rc = hip.hipMalloc(C.byref(p), gb << 30)
    hip.hipDeviceSynchronize()
    t1 = time.perf_counter()
    assert rc == 0, rc
    hip.hipMemset(p, 0, gb << 30)
    hip.hipDeviceSynchronize()
    t2 = time.perf_counter()
    hip.hipFree(p)
    hip.hipDeviceSynchronize()
    t3 = time.perf_counter()
    print("%d,%.1f,%.1f,%.1f" % (gb, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
