"""Diagnostic: K1's std::sort replay alone (aasm_debug_sort_replay) on contigs of n records with duplicate keys; wall time
of the call minus an empty call's, and with the -DAASM_KPROF build (AASM_LIB_OVERRIDE) the kernel's sections on stderr."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, alignasm_amd as A
from alignasm_amd import api
for nc, n in [(1, 1333), (1, 1866), (1, 26666), (5000, 1333), (1024, 1333), (1, 133333)]:
    rng = np.random.default_rng(n)
    qs = rng.integers(0, max(2, (3 * n) // 4), nc * n).astype(np.int64); qe = qs + 1
    off = (np.arange(nc + 1) * n).astype(np.int64)
    perm = np.zeros(nc * n, np.int32)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        rc = api.LIB.aasm_debug_sort_replay(off.ctypes.data_as(C.c_void_p), C.c_int64(nc), qs.ctypes.data_as(C.c_void_p), qe.ctypes.data_as(C.c_void_p),
                                            perm.ctypes.data_as(C.c_void_p), C.c_int(0), C.c_int(0))
        best = min(best, time.perf_counter() - t)
    assert rc == 0
    print("contigs %d x %d records: call %.3f ms (uploads and the download included)" % (nc, n, best * 1e3), flush=True)
