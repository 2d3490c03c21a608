"""Diagnostic: per-section cycle sums of aasm_k7_heap / aasm_k9_select (needs the -DAASM_KPROF build)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["AASM_LIB_OVERRIDE"] = os.environ.get("AASM_KPROF_LIB") or os.path.join(ROOT, "alignasm_amd", "libalignasm_amd_kprof.so")
sys.path.insert(0, ROOT)
import numpy as np, alignasm_amd as A
nc, nr, K, dense, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
dup = int(sys.argv[6]) if len(sys.argv) > 6 else 0
paf = A.Paf.synth(nc, nr, seed, dense=bool(dense), dup_every=dup, no_cs=True)
db = A.DeviceBatch(paf)
for _ in range(2):
    res = db.solve(max_paths=K, timing=True, keep_debug=True)
st = res.stats()
for name in ("prof_heap", "prof_sel"):
    p = res.debug(name, np.int64)[: nc * 8].reshape(nc, 8)
    print(name, "per-contig sum mean/max:", int(p.sum(1).mean()), int(p.sum(1).max()))
    print(name, "mean cycles/contig per section:", np.round(p.mean(0)).astype(int).tolist(), "sum", int(p.sum(1).mean()))
ps = res.debug("prof_sel", np.int64)[: nc * 8].reshape(nc, 8)
print("select: elapsed shader cycles / 100MHz ticks -> GHz:", float((ps[:,6] / np.maximum(ps[:,7],1)).mean()) * 0.1, " wave elapsed ms (mean/max):", ps[:,7].mean() / 1e5, ps[:,7].max() / 1e5)
print(json.dumps({k: round(v, 3) for k, v in st["phase_ms"].items() if v > 0}))
print("V/contig", st["n_vertices"] / nc, "E/contig", st["n_edges"] / nc, "H/contig", st["n_heap_nodes"] / nc)
