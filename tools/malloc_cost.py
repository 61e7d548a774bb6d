"""hipMalloc / first-touch / hipFree cost against the size (the workspace of a decode is allocated once per context and
grown on demand: what a bigger plan costs a short run).  Every size twice in a row (is a freed block cheaper to get
again?), with a memset of the WHOLE buffer behind the first allocation (first touch).  Usage: python tools/malloc_cost.py"""
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
for gb in (8, 16, 24, 32, 48, 64, 96, 128, 192, 230):
    n = gb << 30
    line = f"{gb:4d} GB:"
    for rep in range(2):
        t0 = time.perf_counter()
        rc = hip.hipMalloc(C.byref(p), n)
        t1 = time.perf_counter()
        hip.hipMemset(p, 0, n)
        hip.hipDeviceSynchronize()
        t2 = time.perf_counter()
        hip.hipMemset(p, 0, n)
        hip.hipDeviceSynchronize()
        t3 = time.perf_counter()
        hip.hipFree(p)
        t4 = time.perf_counter()
        line += f"  [rc {rc} malloc {t1 - t0:.3f} s, first memset {t2 - t1:.3f} s, second {t3 - t2:.3f} s, free {t4 - t3:.3f} s]"
    print(line, flush=True)
