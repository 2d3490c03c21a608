import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import alignasm_amd as A
paf = A.Paf.synth(5000, 1000, 21, no_cs=True)
db = A.DeviceBatch(paf)
for rep in range(3):
    t0 = time.perf_counter(); res = db.solve(max_paths=4); t1 = time.perf_counter(); raw = res.fetch_raw(); t2 = time.perf_counter(); A.api.free_out(raw); out = res.fetch(); t3 = time.perf_counter(); res.close()
    print("solve %.1f ms  aasm_result_fetch %.1f ms  fetch + numpy copies %.1f ms  main elems %d alt elems %d all elems %d" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, len(out["main"]), len(out["alt"]), len(out["all"])))
