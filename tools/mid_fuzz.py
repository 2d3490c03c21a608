"""Mid-size batches (150 ... 600 contigs x 40 ... 300 records: every per-contig and per-vertex launch has more than 64 blocks, so the
XCD-aware block mapping and the largest-first order of the several-waves heaps are in play), HIP against the oracle."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import aasm_testlib as T
api = T.api()
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
n = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    nc = rnd.randint(150, 600); nr = rnd.choice([40, 80, 150, 300]); seed = rnd.randint(1, 10 ** 6)
    dense = rnd.random() < 0.4; dup = rnd.choice([0, 0, 3, 7]); shuf = rnd.random() < 0.3; heavy = rnd.random() < 0.3
    K = rnd.choice([1, 4, 16, 300]); nsl = rnd.random() < 0.25
    hb = T.synth(nc, nr, seed, dense=dense, dup_every=dup, shuffle=shuf, heavy_tail=heavy)
    want = T.oracle_solve(hb, K, nsl)
    got = api.solve_batch(hb, max_paths=K, non_skip_linkable=nsl, chain=os.environ.get("AASM_FUZZ_CHAIN", "auto"))   # (the chain class: auto / all / half / none)
    d = T.diff_outputs(want, got)
    assert d == [], (nc, nr, seed, dense, dup, shuf, heavy, K, nsl, d)
    n += 1
    if n % 10 == 0: print("mid fuzz:", n, "batches ok", flush=True)
print("mid fuzz ok:", n, "batches")
