import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import aasm_testlib as T
import test_fuzz as F
api = T.api()
n = 0
for seed in range(2000, 2200):
    for style in (0, 1, 2):
        for nmax in (20, 70):
            hb = F.make_batch(seed, 6, nmax, 400, style)
            for K, nsl in ((10000, False), (3, True)):
                want = T.oracle_solve(hb, K, nsl)
                got = api.solve_batch(hb, max_paths=K, non_skip_linkable=nsl)
                d = T.diff_outputs(want, got)
                assert d == [], (seed, style, nmax, K, nsl, d)
                n += 1
print("long fuzz ok:", n, "batches")
