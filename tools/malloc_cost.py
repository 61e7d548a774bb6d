"""hipMalloc / hipFree cost against the size (the workspace of a decode is allocated once per context and grown on demand:
what a bigger plan costs a short run).  Usage: python tools/malloc_cost.py"""
import ctypes as C
import time

hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipDeviceSynchronize.argtypes = []
p = C.c_void_p()
hip.hipMalloc(C.byref(p), 1 << 20)
hip.hipFree(p)
for gb in (1, 4, 16, 64, 128, 230):
    n = gb << 30
    t0 = time.perf_counter()
    rc = hip.hipMalloc(C.byref(p), n)
    t1 = time.perf_counter()
    hip.hipMemset(p, 0, min(n, 1 << 30))
    hip.hipDeviceSynchronize()
    t2 = time.perf_counter()
    hip.hipFree(p)
    t3 = time.perf_counter()
    print(f"{gb:4d} GB: rc {rc} hipMalloc {t1 - t0:.3f} s, first 1 GB memset {t2 - t1:.3f} s, hipFree {t3 - t2:.3f} s", flush=True)
