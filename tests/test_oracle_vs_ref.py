"""Pins the oracle's generic algorithms against the REAL reference headers
(oracle/_ref/*.so, built by oracle/Makefile from /root/reference/src where it exists):
PafDistance predicates, Dial BFS, Kahn orders, DAG shortest-path tree, persistent leftist
heaps, k-walk enumeration and path recovery.  Skipped when the _ref libraries are absent."""
import ctypes as C
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.ref


def _need(T, mono=True):
    lib = T.ref(mono)
    if lib is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this box and no prebuilt .so)")
    return lib


def test_pafdistance_truth_tables(T):
    R, O = _need(T), T.oracle()
    vals = [-2, -1, 0, 1, 3]
    rng = np.random.default_rng(1)
    cands = [np.array(v, np.int64) for v in itertools.product(vals, [-1, 0, 2], [-1, 0, 1], [-2, -1, 0, 1, 2], [-1, 0, 1, 2])]
    cands += [np.array([-1, -1, -1, -1, 0], np.int64), np.array([-1, -1, -1, 1, -1], np.int64), np.array([-1, -1, -1, -2, 2], np.int64)]
    idx = rng.integers(0, len(cands), size=(6000, 2))
    P = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
    for i, j in idx:
        a, b = cands[i], cands[j]
        for mode in (0, 1):
            assert R.ref_dist_lt(P(a), P(b), mode) == O.oracle_dist_lt(P(a), P(b), mode), (a, b, mode)
        assert R.ref_dist_eq(P(a), P(b)) == O.oracle_dist_eq(P(a), P(b)), (a, b)


GRAPH_CASES = [(4, 100, 1, False, 0), (2, 400, 7, False, 0), (2, 250, 31, True, 0), (3, 200, 5, False, 3), (2, 120, 9, True, 2)]


@pytest.mark.parametrize("case", GRAPH_CASES, ids=str)
def test_generic_algorithms_match_reference_exactly(T, case):
    """Monotonic-allocator flavour: pointer order == allocation order -> exact equality of
    distances, every recovered path, orders, tree, anomaly distances and heap shape."""
    R, O = _need(T, True), T.oracle()
    nc, nr, seed, dense, dup = case
    hb = T.synth(nc, nr, seed, dense=dense, dup_every=dup)
    K = 3000
    for c in range(nc):
        n, rp, col, w = T.contig_graph(hb, c, K)
        a = T.generic_run(O, "oracle_", n, rp, col, w, n - 2, n - 1, K)
        r = T.generic_run(R, "ref_", n, rp, col, w, n - 2, n - 1, K)
        for key in ("nd", "dist", "anom", "rev", "fwd", "best", "d", "hroot", "hcount"):
            assert np.array_equal(a[key], r[key]), (c, key)
        assert all(np.array_equal(x, y) for x, y in zip(a["paths"], r["paths"])), c


def test_glibc_flavour_differs_only_in_tie_order(T):
    """Shipped behaviour (glibc malloc): the PQ breaks distance ties on raw pointers
    (k_shortest_walks.hpp:231).  Everything except the ORDER inside a class of
    `<`-equivalent distances must still agree."""
    from fractions import Fraction
    R, O = _need(T, False), T.oracle()
    hb = T.synth(4, 100, 1)
    K = 5000

    def classes(d):
        d = d.reshape(-1, 5)
        return [(int(x[0] + x[1]), int(x[2]), Fraction(int(x[3]), int(x[4]) if x[4] else 1)) for x in d]
    for c in range(4):
        n, rp, col, w = T.contig_graph(hb, c, K)
        a = T.generic_run(O, "oracle_", n, rp, col, w, n - 2, n - 1, K, with_paths=False)
        r = T.generic_run(R, "ref_", n, rp, col, w, n - 2, n - 1, K, with_paths=False)
        for key in ("nd", "anom", "rev", "fwd", "best", "d", "hroot", "hcount"):
            assert np.array_equal(a[key], r[key]), (c, key)
        assert classes(a["dist"]) == classes(r["dist"])


def test_random_dags_match_reference(T):
    R, O = _need(T, True), T.oracle()
    rng = np.random.default_rng(123)
    for trial in range(40):
        n = int(rng.integers(4, 40))
        rows = [[] for _ in range(n)]
        for u in range(n - 1):
            for v in range(u + 1, n):
                if rng.random() < 0.25:
                    rows[u].append((v, [int(rng.integers(0, 6)), int(rng.integers(0, 6)), int(rng.integers(0, 3)), int(rng.integers(0, 2)), 1]))
            if not rows[u]:
                v = int(rng.integers(u + 1, n))
                rows[u].append((v, [1, 1, 0, 1, 1]))
            rng.shuffle(rows[u])              # adjacency order is significant
        rp = np.zeros(n + 1, np.int64)
        col, w = [], []
        for u in range(n):
            for v, ww in rows[u]:
                col.append(v); w.extend(ww)
            rp[u + 1] = len(col)
        col, w = np.array(col, np.int64), np.array(w, np.int64)
        a = T.generic_run(O, "oracle_", n, rp, col, w, 0, n - 1, 500)
        r = T.generic_run(R, "ref_", n, rp, col, w, 0, n - 1, 500)
        for key in ("nd", "dist", "anom", "rev", "fwd", "best", "d", "hroot", "hcount"):
            assert np.array_equal(a[key], r[key]), (trial, key)
        assert all(np.array_equal(x, y) for x, y in zip(a["paths"], r["paths"])), trial
