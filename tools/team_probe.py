"""K7's team form (several workgroups per contig of the wide-tree class): a giant dense contig and a few big ones, by team size;
every team size's outputs are compared byte for byte with the one-workgroup form's."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import alignasm_amd as A

def outs(res):
    o = res.fetch()
    return {k: v.tobytes() for k, v in o.items() if isinstance(v, np.ndarray)}

teams = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4,8,16,32").split(",")]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
shapes = [tuple(int(y) for y in x.split("x")) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [(1, 5000, 1), (1, 20000, 1), (6, 3000, 1), (1, 20000, 0)]
for nc, n, dense in shapes:
    paf = A.Paf.synth(nc, n, 77, dense=bool(dense), no_cs=True)
    db = A.DeviceBatch(paf)
    base = None
    for team in teams:
        kw = dict(max_paths=K, timing=True, heap_team=team)
        if not dense: kw["heap_waves"] = "all"
        for _ in range(3):
            res = db.solve(**kw)
            st = res.stats()
            if _ < 2: res.close()
        got = outs(res)
        res.close()
        if base is None: base = got
        same = all(base[k] == got[k] for k in base)
        print(json.dumps({"contigs": nc, "records": n, "dense": dense, "team": team, "heap_ms": round(st["phase_ms"]["heap"], 2), "total_ms": round(st["total_ms"], 2),
                          "H": st["n_heap_nodes"], "same_as_team1": same}), flush=True)
        if not same: sys.exit(1)
    db.close(); paf.close()
