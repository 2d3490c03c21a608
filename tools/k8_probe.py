"""Diagnostic: K8 at the shipped K on n contigs of the C3 batch (phase times from HIP events)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import alignasm_amd as A
nc, K = int(sys.argv[1]), int(sys.argv[2])
heap = len(sys.argv) > 3 and sys.argv[3] == "heap"
paf = A.Paf.synth(nc, 1000, 21, no_cs=True)
db = A.DeviceBatch(paf)
for _ in range(3):
    res = db.solve(max_paths=K, timing=True, enum_heap=heap)
    st = res.stats()
    res.close()
print(json.dumps({"contigs": nc, "K": K, "form": "heap" if heap else "runs", "enum_ms": round(st["phase_ms"]["enum"], 3), "total_ms": round(st["total_ms"], 3),
                  "paths_found": st["n_paths_found"], "pq_pushes": st.get("pq_pushes")}))
