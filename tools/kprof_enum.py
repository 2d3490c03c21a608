"""Diagnostic: per-contig cycle sums of aasm_k8_enum's sections (needs the -DAASM_KPROF build): 0 pops + pushes, 1 flush of the
insertion buffer (sort + run merges), 2 refill of the front, 3 successor fetch, 4 far-tier scans."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["AASM_LIB_OVERRIDE"] = os.environ.get("AASM_KPROF_LIB") or os.path.join(ROOT, "alignasm_amd", "libalignasm_amd_kprof.so")
sys.path.insert(0, ROOT)
import numpy as np, alignasm_amd as A
nc, nr, K, dense, seed = (int(x) for x in sys.argv[1:6])
paf = A.Paf.synth(nc, nr, seed, dense=bool(dense), no_cs=True)
db = A.DeviceBatch(paf)
for _ in range(2):
    res = db.solve(max_paths=K, timing=True, keep_debug=True)
st = res.stats()
p = res.debug("prof_heap", np.int64)[: nc * 8].reshape(nc, 8)[:, :5]
tot = p.sum(1)
print("enum_ms", round(st["phase_ms"]["enum"], 3), "mean cycles per section", np.round(p.mean(0)).astype(int).tolist(), "sum mean/max", int(tot.mean()), int(tot.max()))
print("percentiles of the per-contig sum (50, 90, 99, 100):", [int(np.percentile(tot, q)) for q in (50, 90, 99, 100)])
for c in np.argsort(-tot)[:5]:
    print("contig", int(c), "sections", p[c].tolist())
