"""K8 stress: the sorted-front / sorted-runs queue (64- and 40-entry fronts) against the d-ary heap form on many contigs of mixed
shape and many K: kfound, every popped distance, every pop's insertion index, every numbered push."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import aasm_testlib as T
api = T.api()
n_checked = 0
for seed, nc, nr, dense, dup, heavy in ((1, 60, 300, False, 0, True), (2, 30, 500, False, 3, False), (3, 24, 220, True, 0, False), (4, 40, 120, True, 2, True),
                                       (5, 80, 90, False, 0, True), (6, 12, 900, False, 5, False), (7, 10, 400, True, 4, False)):
    hb = T.synth(nc, nr, seed, dense=dense, dup_every=dup, heavy_tail=heavy)
    db = api.DeviceBatch(hb)
    for K in (22, 65, 130, 700, 2500, 10000):
        got = {}
        for form in ("heap", "runs", "runs40"):
            res = db.solve(max_paths=K, keep_debug=True, enum_heap=(form == "heap"), enum_small=(form == "runs40"))
            kf = res.debug("kfound", np.int32)[:nc].copy()
            kd = res.debug("kd", T.DIST_DT)[:nc * K].copy()
            kl = res.debug("klast", np.int32)[:nc * K].copy()
            cand = res.debug("kcand", np.int32)[:nc * (3 * K + 1) * 8].reshape(nc * (3 * K + 1), 8)[:, :2].copy()
            got[form] = (kf, kd, kl, cand)
            res.close()
        for other in ("runs", "runs40"):
            a, b = got["heap"], got[other]
            assert np.array_equal(a[0], b[0]), (seed, K, other)
            for c in range(nc):
                n = int(a[0][c]); S = 3 * K + 1
                assert np.array_equal(a[1][c * K:c * K + n], b[1][c * K:c * K + n]), (seed, K, other, c, "kd")
                assert np.array_equal(a[2][c * K:c * K + n], b[2][c * K:c * K + n]), (seed, K, other, c, "klast")
                pushed = int(a[2][c * K:c * K + n].max()) + 1
                assert np.array_equal(a[3][c * S:c * S + pushed], b[3][c * S:c * S + pushed]), (seed, K, other, c, "cand")
                n_checked += 1
    db.close()
    print("seed", seed, "ok", flush=True)
print("k8 stress ok:", n_checked, "contig x K x form comparisons")
