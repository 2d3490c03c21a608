"""K7 on a dense batch: the several-waves class launched largest node bound first (default) against input order; outputs are
compared byte for byte."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import alignasm_amd as A

def outs(res):
    return {k: v.tobytes() for k, v in res.fetch().items() if isinstance(v, np.ndarray)}

K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
shapes = [tuple(int(y) for y in x.split("x")) for x in (sys.argv[2] if len(sys.argv) > 2 else "1250x1000,700x1000,400x1500,100x3000,5000x250").split(",")]
for nc, n in shapes:
    paf = A.Paf.synth(nc, n, 31, dense=True, no_cs=True)
    db = A.DeviceBatch(paf)
    base = None
    for input_order in (True, False):
        for it in range(3):
            res = db.solve(max_paths=K, timing=True, heap_input_order=input_order)
            st = res.stats()
            if it < 2: res.close()
        got = outs(res)
        res.close()
        if base is None: base = got
        same = all(base[k] == got[k] for k in base)
        print(json.dumps({"contigs": nc, "records": n, "order": "input" if input_order else "largest first", "heap_ms": round(st["phase_ms"]["heap"], 2), "total_ms": round(st["total_ms"], 2), "same_outputs": same}), flush=True)
        if not same: sys.exit(1)
    db.close(); paf.close()
