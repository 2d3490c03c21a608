"""Diagnostic (host emulation, exact counts): what the K9 upgrade loop does per conversion on a workload's graphs."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, aasm_testlib as T
NAMES = ["conversions", "path_edges", "edges_copied_in_runs", "steps", "steps_pair_head", "steps_known", "ispr_calls", "ispr_w2_fast", "ispr_lds_dp",
         "ispr_stream", "ispr_generic", "ispr_empty", "start_ne_u", "steps_to_dest", "cw_fills", "lds_dp_window_sum", "settled_d2", "settled_d3", "W<=4", "W<=8", "W<=16", "W<=32", "W>32", "-", "E<=16", "E<=32", "E<=64", "E<=128", "E>128", "-", "W16E64", "W32E64"]
for name, kw, nc, K in (("c3", {}, 40, 4), ("c3_dup3", {"dup_every": 3}, 40, 4), ("c5", {"dense": True}, 6, 16), ("c3_heavy", {"heavy_tail": True}, 60, 4)):
    hb = T.synth(nc, 1000, 21 if "dense" not in kw else 31, **kw)
    a = np.zeros(64, np.int64)
    T.emul().emul_k9_stats(a.ctypes.data_as(C.c_void_p), 1)
    T.emul_solve(hb, K)
    T.emul().emul_k9_stats(a.ctypes.data_as(C.c_void_p), 1)
    d = dict(zip(NAMES, (int(x) for x in a)))
    cv = max(1, d["conversions"])
    print(name, json.dumps({k: round(v / cv, 1) for k, v in d.items()}))
