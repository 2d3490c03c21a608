"""Scratch robustness run: many back-to-back solves of batches of different shapes on one context (the scans, the
merged zero fills and the two streams are reused across solves); every result is compared with the first one of its shape."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import alignasm_amd as A

shapes = [(5000, 1000, False, 4), (37, 300, True, 16), (1, 5000, False, 4), (800, 50, False, 10000), (2600, 40, False, 4), (3, 1, False, 4)]
dbs, first = [], []
for nc, nr, dense, K in shapes:
    paf = A.Paf.synth(nc, nr, 5 + nc, dense=dense, no_cs=True); hb = paf.batch(); paf.close()
    dbs.append(A.DeviceBatch(hb)); first.append(None)
t0 = time.time(); n = 0
for rep in range(60):
    for i, (nc, nr, dense, K) in enumerate(shapes):
        if i == 0 and rep % 10:                      # the big one every tenth round
            continue
        res = dbs[i].solve(max_paths=K)
        out = res.fetch(); res.close()
        key = (out["main_off"].tobytes(), out["main"].tobytes(), out["alt"].tobytes(), out["all"].tobytes(), out["status"].tobytes())
        if first[i] is None:
            first[i] = key
        assert key == first[i], (rep, i)
        n += 1
print("stress ok:", n, "solves in %.1f s" % (time.time() - t0))
