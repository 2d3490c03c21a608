"""Diagnostic: the output writer alone (rows formatted + written) at several host-thread counts; the outputs come from the CPU oracle.
   python tools/writer_probe.py <contigs> <threads,threads,...> [dir]"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import aasm_testlib as T
import alignasm_amd as A
from alignasm_amd import api
nc = int(sys.argv[1]); thr_list = [int(x) for x in sys.argv[2].split(",")]
d = sys.argv[3] if len(sys.argv) > 3 else "/tmp/aasm_e2e/wrb"
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), flush=True)
paf = A.Paf.synth(nc, 1000, 21)
hb = paf.batch()
t = time.time()
o = T.Opts(4, 0, 0, 0, 0); out = T.BatchOut()
assert T.oracle().oracle_solve_batch(C.byref(hb.view), C.byref(o), 16, C.byref(out)) == 0
print("oracle %.2fs" % (time.time() - t), flush=True)
os.makedirs(d, exist_ok=True)
for thr in thr_list:
    api.LIB.aasm_set_host_threads(C.c_int(thr))
    best = 1e9
    for rep in range(3):
        for f in ("m.paf", "a.paf", "l.paf"):
            if os.path.exists(d + "/" + f): os.remove(d + "/" + f)      # (freeing a 1 GB file's page cache inside rename() is not the writer's time)
        t = time.time()
        paf.write_outputs(out, d + "/m.paf", d + "/a.paf", d + "/l.paf")
        best = min(best, time.time() - t)
    sz = sum(os.path.getsize(d + "/" + f) for f in ("m.paf", "a.paf", "l.paf"))
    print("threads %3d: write_outputs %.3f s, %.1f MB -> %.0f MB/s (%.0f MB/s per thread)" % (thr, best, sz / 1e6, sz / 1e6 / best, sz / 1e6 / best / thr), flush=True)
for f in ("m.paf", "a.paf", "l.paf"): os.remove(d + "/" + f)
