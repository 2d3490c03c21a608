"""GPU tier, K8 (k_shortest_walks.hpp:217-249): the sorted-front / sorted-runs queue (aasm_enum.h) against the oracle AND
against the d-ary heap form of the same kernel - every popped distance, the insertion index of every pop, and the
(heap node, predecessor) record of every push, at K large enough that the insertion buffer is flushed, runs are merged
down several levels, cut at K - found, and the front is refilled from the run heads many times."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

#        contigs, records, seed, K, dense, dup_every
CASES = [(3, 1000, 21, 10000, False, 0),     # the bench's k10000 shape
         (2, 400, 31, 10000, True, 0),       # dense: hundreds of distinct sums, deep heaps
         (3, 1000, 5, 2500, False, 3),       # ties from duplicated records
         (4, 600, 9, 70, False, 0),          # K just past one front (64)
         (4, 600, 9, 64, False, 0),
         (4, 600, 9, 200, True, 0),
         (2, 1000, 3, 1000, False, 0),
         (6, 30, 4, 10000, False, 0),        # tiny graphs: the queue runs empty long before K
         (1, 3000, 8, 30000, False, 0)]      # more levels (lmax = 9)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "c%dx%d_s%d_k%d_%s%s" % (c[0], c[1], c[2], c[3], "D" if c[4] else "S", "_dup%d" % c[5] if c[5] else ""))
def test_enumeration_queue_forms_agree_with_each_other_and_the_oracle(T, case):
    nc, nr, seed, K, dense, dup = case
    api = T.api()
    hb = T.synth(nc, nr, seed, dense=dense, dup_every=dup)
    db = api.DeviceBatch(hb)
    got = {}
    for form in ("runs", "runs40", "heap"):                          # 64-entry front, the 40-entry front of big batches, the d-ary heap
        res = db.solve(max_paths=K, keep_debug=True, enum_heap=(form == "heap"), enum_small=(form == "runs40"))
        kf = res.debug("kfound", np.int32)[:nc].copy()
        kd = res.debug("kd", T.DIST_DT)[:nc * K].copy()
        klast = res.debug("klast", np.int32)[:nc * K].copy()
        S = 3 * K + 1
        cand = res.debug("kcand", np.int32)[:nc * S * 8].reshape(nc * S, 8)    # {heap node, predecessor, qry (2 words)}, {anom, qnz, qtot, -}
        knodes, kprev = cand[:, 0].copy(), cand[:, 1].copy()
        got[form] = (kf, kd, klast, knodes, kprev, res.fetch())
        if form == "runs":
            assert T.diff_intermediates(hb, res.debug, K) == []      # kfound + every popped distance vs the oracle
        res.close()
    db.close()
    for other in ("heap", "runs40"):
        a, b = got["runs"], got[other]
        assert np.array_equal(a[0], b[0]), other
        for c in range(nc):
            n = int(a[0][c])
            assert n >= 1
            for f in ("qry", "ref", "anom", "qnz", "qtot"):
                assert np.array_equal(a[1][f][c * K:c * K + n], b[1][f][c * K:c * K + n]), (other, c, f)
            assert np.array_equal(a[2][c * K:c * K + n], b[2][c * K:c * K + n]), (other, c)
            S = 3 * K + 1
            pushed = int(a[2][c * K:c * K + n].max()) + 1            # every push up to the last popped one was numbered alike
            assert np.array_equal(a[3][c * S:c * S + pushed], b[3][c * S:c * S + pushed]), (other, c)
            assert np.array_equal(a[4][c * S:c * S + pushed], b[4][c * S:c * S + pushed]), (other, c)
        assert T.diff_outputs(a[5], b[5]) == [], other
    assert T.diff_outputs(T.oracle_solve(hb, K), got["runs"][5]) == []


def test_queue_forms_agree_on_mixed_batches_and_many_k(T):
    """Heavy-tailed, dense and duplicate-heavy contigs at K around the front's capacity (22, 65, 130) and above: the two front
    capacities of the sorted-runs queue against the d-ary heap form (tools/k8_stress.py runs the long version)."""
    api = T.api()
    n_checked = 0
    for seed, nc, nr, dense, dup, heavy in ((1, 40, 300, False, 0, True), (3, 16, 220, True, 0, False), (4, 30, 120, True, 2, True)):
        hb = T.synth(nc, nr, seed, dense=dense, dup_every=dup, heavy_tail=heavy)
        db = api.DeviceBatch(hb)
        for K in (22, 65, 130, 2500):
            got = {}
            for form in ("heap", "runs", "runs40"):
                res = db.solve(max_paths=K, keep_debug=True, enum_heap=(form == "heap"), enum_small=(form == "runs40"))
                got[form] = (res.debug("kfound", np.int32)[:nc].copy(), res.debug("kd", T.DIST_DT)[:nc * K].copy(), res.debug("klast", np.int32)[:nc * K].copy())
                res.close()
            for other in ("runs", "runs40"):
                a, b = got["heap"], got[other]
                assert np.array_equal(a[0], b[0]), (seed, K, other)
                for c in range(nc):
                    n = int(a[0][c])
                    assert np.array_equal(a[1][c * K:c * K + n], b[1][c * K:c * K + n]) and np.array_equal(a[2][c * K:c * K + n], b[2][c * K:c * K + n]), (seed, K, other, c)
                    n_checked += 1
        db.close()
    assert n_checked == 2 * 4 * (40 + 16 + 30)
