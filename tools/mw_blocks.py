"""Diagnostic (-DAASM_KPROF build): kb_heap_mw's per-block phase times on a dense batch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["AASM_LIB_OVERRIDE"] = os.environ.get("AASM_KPROF_LIB") or os.path.join(ROOT, "alignasm_amd", "libalignasm_amd_kprof.so")
sys.path.insert(0, ROOT)
import numpy as np, alignasm_amd as A
nc, nr, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
paf = A.Paf.synth(nc, nr, 12345, dense=True, no_cs=True)
db = A.DeviceBatch(paf)
for _ in range(2):
    res = db.solve(max_paths=K, timing=True, keep_debug=True)
st = res.stats()
p = res.debug("prof_heap", np.int64)[: nc * 8].reshape(nc, 8).astype(float)
ms = p[:, :3] / 1e5
print("heap phase ms", round(st["phase_ms"]["heap"], 2))
for i, name in enumerate(("phase0", "phase1", "phase2")):
    print(name, "ms per block: mean %.2f  p50 %.2f  p90 %.2f  max %.2f" % (ms[:, i].mean(), np.median(ms[:, i]), np.percentile(ms[:, i], 90), ms[:, i].max()))
print("H per block: mean %.0f max %.0f; corr(H, phase1) %.3f" % (p[:, 3].mean(), p[:, 3].max(), np.corrcoef(p[:, 3], ms[:, 1])[0, 1]))
print("phase1 ns per node: mean %.1f, heaviest block %.1f" % ((ms[:, 1] * 1e6 / p[:, 3]).mean(), ms[:, 1].max() * 1e6 / p[np.argmax(ms[:, 1]), 3]))
