"""Port / reference time ratio: the REAL solve_ctg_read() prefix (K1 ... K8, paf_data.cpp:223-738, with its
four N x N tables and MAX_PATH_COUNT = 10000; oracle/_ref/libaasm_ref_prefix.so, glibc flavour = the shipped allocator)
beside the port (oracle/alignasm_oracle.cpp) stopped at the same line, same contigs, same threads.  Run in the build
container (the library needs /root/reference to be built): python tools/ref_prefix_time.py [threads ...]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import aasm_testlib as T   # noqa: E402

CONFIGS = [("C1 10x100 seed 1", 10, 100, 1, False), ("C2 50x1000 seed 11", 50, 1000, 11, False), ("C3 slice 40x1000 seed 21", 40, 1000, 21, False),
           ("C5 slice 8x1000 dense seed 31", 8, 1000, 31, True)]


def main():
    threads = [int(x) for x in sys.argv[1:]] or [1, 8]
    R, O = T.ref_prefix(mono=False), T.oracle()
    assert R is not None, "make -C oracle (needs /root/reference)"
    O.oracle_time_prefix.restype = C.c_double
    rows = []
    for name, nc, nr, seed, dense in CONFIGS:
        hb = T.synth(nc, nr, seed, dense=dense)
        for th in threads:
            best = lambda f: min(f() for _ in range(3 if nc * nr <= 1000 else 1))
            tr = best(lambda: R.refp_time_batch(C.byref(hb.view), C.c_int64(0), C.c_int64(nc), th, 0))
            tp = best(lambda: O.oracle_time_prefix(C.byref(hb.view), C.c_int64(0), C.c_int64(nc), th, 0, C.c_int64(10000)))
            row = dict(config=name, threads=th, reference_prefix_s=round(tr, 4), port_prefix_s=round(tp, 4), reference_over_port=round(tr / tp, 2),
                       reference_ms_per_contig=round(1e3 * tr / nc, 3), port_ms_per_contig=round(1e3 * tp / nc, 3))
            rows.append(row)
            print(json.dumps(row), flush=True)
    return rows


if __name__ == "__main__":
    main()
