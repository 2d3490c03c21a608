"""Extra parity run for the dense-graph kernels (lanes-over-edges pre-passes, several-waves heaps, streamed window DPs): random
dense batches of 2 ... 6 contigs x 60 ... 900 records, with and without duplicated records / shuffling / NON_SKIP_LINKABLE,
K in {1, 4, 16, 10000}; HIP outputs (and stats) against the oracle.  Every fourth batch puts every contig in the several-waves class,
every other one of those in input order."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import aasm_testlib as T
api = T.api()
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 20261004)   # second argument: another seed
n = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 120):
    nc = rnd.randint(2, 6); nr = rnd.choice([60, 130, 260, 400, 650, 900]); seed = rnd.randint(1, 10 ** 6)
    dup = rnd.choice([0, 0, 3, 7]); shuf = rnd.random() < 0.3; K = rnd.choice([1, 4, 16, 10000]); nsl = rnd.random() < 0.25
    hb = T.synth(nc, nr, seed, dense=True, dup_every=dup, shuffle=shuf)
    want = T.oracle_solve(hb, K, nsl)
    kw = {}
    if it % 4 == 3: kw = {"heap_waves": "all", "heap_input_order": it % 8 == 7}
    got = api.solve_batch(hb, max_paths=K, non_skip_linkable=nsl, **kw)
    d = T.diff_outputs(want, got)
    assert d == [], (nc, nr, seed, dup, shuf, K, nsl, d)
    n += 1
    if n % 20 == 0: print("dense fuzz:", n, "batches ok", flush=True)
print("dense fuzz ok:", n, "batches")
