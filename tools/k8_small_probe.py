"""Diagnostic: K8 at the shipped K with the 64-entry front against the 40-entry one (more waves per CU) on the whole C3 batch."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import alignasm_amd as A
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
paf = A.Paf.synth(nc, 1000, 21, no_cs=True)
db = A.DeviceBatch(paf)
for small in (False, True, False, True):
    best = 1e9
    for _ in range(3):
        res = db.solve(max_paths=10000, timing=True, enum_small=small); st = res.stats(); res.close()
        best = min(best, st["phase_ms"]["enum"])
    print(json.dumps({"contigs": nc, "front40": small, "enum_ms": round(best, 3)}), flush=True)
