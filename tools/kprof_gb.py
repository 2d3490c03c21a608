"""Diagnostic: per-phase cycles of aasm_k46_graph's workgroups, thread 0 (needs the -DAASM_KPROF build)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["AASM_LIB_OVERRIDE"] = os.environ.get("AASM_KPROF_LIB") or os.path.join(ROOT, "alignasm_amd", "libalignasm_amd_kprof.so")
sys.path.insert(0, ROOT)
import numpy as np, alignasm_amd as A
nc, nr, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
paf = A.Paf.synth(nc, nr, seed, no_cs=True)
db = A.DeviceBatch(paf)
for _ in range(2):
    res = db.solve(max_paths=4, timing=True, keep_debug=True)
p = res.debug("prof_gb", np.int64)[: nc * 8].reshape(nc, 8)
print("mean cycles per workgroup: rows, scan, note, place, headers:", np.round(p.mean(0)).astype(int).tolist()[:5], "sum", int(p.sum(1).mean()))
print({k: round(v, 3) for k, v in res.stats()["phase_ms"].items() if v > 0})
