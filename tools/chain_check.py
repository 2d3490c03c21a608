"""The chain class (aasm_k67_chain) against the oracle and against the three-launch form: outputs, statistics, and the heap arena
+ roots word for word (chain = "all" / "none"), on a spread of batch shapes."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import alignasm_amd as A
import aasm_testlib as T
bad = 0
cases = [(6, 120, 42, dict(dup_every=7, shuffle=True), 16), (9, 60, 17, dict(heavy_tail=True, dup_every=5), 16), (3, 400, 5, {}, 4), (40, 300, 8, dict(heavy_tail=True), 4),
         (200, 200, 3, dict(dup_every=3, shuffle=True), 64), (1, 3000, 77, {}, 4), (64, 1000, 21, {}, 4), (12, 150, 31, dict(dense=True), 16), (300, 100, 9, dict(heavy_tail=True, dup_every=4), 10000)]
for nc, nr, seed, kw, K in cases:
    hb = T.synth(nc, nr, seed, **kw)
    want = T.oracle_solve(hb, K)
    db = A.DeviceBatch(hb)
    arena = {}
    for chain in ("none", "all", "auto"):
        t = time.time()
        res = db.solve(max_paths=K, keep_debug=True, chain=chain, timing=True)
        got = res.fetch(); got["stats"] = res.stats()
        d = T.diff_outputs(want, got)
        hoff, hcnt = res.debug("hoff", np.int64), res.debug("h_cnt", np.int32)
        hn = res.debug("hnodes", np.int32)
        nodes = np.concatenate([hn[12 * int(hoff[c]): 12 * (int(hoff[c]) + int(hcnt[c]))] for c in range(nc)]) if nc else hn[:0]
        arena[chain] = (nodes, res.debug("h_root", np.int32).copy(), res.debug("tnx16", np.int32).copy(), res.debug("st_n", np.int32).copy())
        extra = (res.debug("sp_d", np.int64).reshape(-1, 4).copy(), res.debug("sp_best", np.int32).copy(), res.debug("voff", np.int64)[:nc + 1].copy(), res.debug("rev_order", np.int32).copy())
        nch = int(res.debug("counters", np.int64)[17])
        print(nc, nr, kw, K, chain, "chain contigs", nch, "diff", d, "ms %.3f" % got["stats"]["total_ms"], {k: round(v, 3) for k, v in got["stats"]["phase_ms"].items() if k in ("sptree", "heap", "heap_prep", "chain")}, flush=True)
        bad += len(d)
        res.close()
    H = want["stats"]["n_heap_nodes"]
    for i, nm in enumerate(("hnodes", "h_root", "tnx16", "st_n")):
        a, b = arena["none"][i], arena["all"][i]
        VT = int(extra[2][nc])
        if nm != "hnodes": a, b = a[:VT * (16 if nm == "tnx16" else 1)], b[:VT * (16 if nm == "tnx16" else 1)]   # (allocations are padded to 256 bytes)
        if not np.array_equal(a, b):
            print("   ARENA DIFF", nm, int((a != b).sum()), "words"); bad += 1
            idx = np.nonzero(a != b)[0][:6]
            div = 16 if nm == "tnx16" else 1
            for i in idx:
                v = int(i) // div
                cc = int(np.searchsorted(extra[2], v, side="right") - 1)
                print("      at", int(i), "vertex", v, "contig", cc, "local", v - int(extra[2][cc]), "of", int(extra[2][cc + 1] - extra[2][cc]), "none", int(a[i]), "all", int(b[i]), "d", extra[0][v].tolist(), "best", int(extra[1][v]))
    db.close()
print("chain_check:", "OK" if bad == 0 else "FAILED %d" % bad)
sys.exit(1 if bad else 0)
