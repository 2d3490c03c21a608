"""Diagnostic: aasm_k0_cs_ranges alone (records with cs tags uploaded, ranges derived on the device): kernel time from the phase timers."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import alignasm_amd as A
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
src = A.Paf.synth(nc, 1000, 21)
os.makedirs("/tmp/aasm_probe", exist_ok=True); src.save("/tmp/aasm_probe/k0.paf"); src.close()
hb = A.Paf.read("/tmp/aasm_probe/k0.paf", device_ranges=True)
db = A.DeviceBatch(hb)
best = {}
for r in range(4):
    res = db.solve(max_paths=4, timing=True); st = res.stats(); res.close()
    for k, v in st["phase_ms"].items():
        if v > 0: best[k] = min(best.get(k, 1e9), v)
print(json.dumps({"contigs": nc, "total_ms": round(st["total_ms"], 3), "phases": {k: round(v, 3) for k, v in best.items()}}))
