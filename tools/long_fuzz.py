"""Extra adversarial batches (tests/test_fuzz.py's generator), HIP against the oracle.  Arguments: first seed (2000), number of seeds (200)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import aasm_testlib as T
import test_fuzz as F
api = T.api()
n = 0
s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for seed in range(s0, s0 + (int(sys.argv[2]) if len(sys.argv) > 2 else 200)):
    for style in (0, 1, 2):
        for nmax in (20, 70):
            hb = F.make_batch(seed, 6, nmax, 400, style)
            for K, nsl in ((10000, False), (3, True)):
                want = T.oracle_solve(hb, K, nsl)
                got = api.solve_batch(hb, max_paths=K, non_skip_linkable=nsl, chain=os.environ.get("AASM_FUZZ_CHAIN", "auto"))
                d = T.diff_outputs(want, got)
                assert d == [], (seed, style, nmax, K, nsl, d)
                n += 1
    if (seed - s0) % 50 == 49: print("long fuzz:", n, "batches ok", flush=True)
print("long fuzz ok:", n, "batches")
