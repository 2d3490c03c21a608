"""Diagnostic: K9 work counters (window-DP edges / vertices, path edges, conversions) and time, plain C3 against C3 with duplicated records."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import alignasm_amd as A
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
for dup in (0, 3):
    paf = A.Paf.synth(nc, 1000, 21, dup_every=dup, no_cs=True)
    db = A.DeviceBatch(paf)
    for _ in range(2):
        res = db.solve(max_paths=4, timing=True); st = res.stats(); res.close()
    print(json.dumps({"dup": dup, "select_ms": round(st["phase_ms"]["select"], 3), "final_ms": round(st["phase_ms"]["final"], 3), "converted": st["n_paths_converted"],
                      "ispr_edges": st["ispr_edges"], "ispr_vertices": st["ispr_vertices"], "path_edges": st["path_edges"], "out_elems": st["out_elems"],
                      "V": st["n_vertices"], "E": st["n_edges"], "pairs": st["n_pairs"]}))
    db.close(); paf.close()
