"""f4: what ONE giant contig costs, phase by phase (one wave walks each chain; the rest of the chip idles)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import alignasm_amd as A
rows = []
for n, dense, K in ((1000, 0, 4), (20000, 0, 4), (50000, 0, 4), (100000, 0, 4), (5000, 1, 4), (20000, 1, 4)):
    paf = A.Paf.synth(1, n, 77, dense=bool(dense), no_cs=True)
    db = A.DeviceBatch(paf)
    for _ in range(2):
        res = db.solve(max_paths=K, timing=True)
        st = res.stats()
        res.close()
    ph = {k: round(v, 3) for k, v in st["phase_ms"].items() if v >= 0.05}
    rows.append({"records": n, "dense": dense, "K": K, "V": st["n_vertices"], "E": st["n_edges"], "H": st["n_heap_nodes"], "total_ms": round(st["total_ms"], 2),
                 "us_per_record": round(st["total_ms"] * 1e3 / n, 2), "device_MB": st["device_bytes"] >> 20, "phase_ms": ph})
    print(json.dumps(rows[-1]), flush=True)
    db.close(); paf.close()
