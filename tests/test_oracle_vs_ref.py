"""Pins the oracle's generic algorithms against the REAL reference headers
(oracle/_ref/*.so, built by oracle/Makefile from /root/reference/src where it exists):
PafDistance predicates, Dial BFS, Kahn orders, DAG shortest-path tree, persistent leftist
heaps, k-walk enumeration and path recovery, and the header-only K1/K2 pieces of paf_data.hpp
(std::sort over real PafReadData objects with the real operator<, qry_contains,
qry_partial_overlap, the PafOutputData constructor).  Skipped when the _ref libraries are absent."""
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


# ---- K1 / K2: header-only code of paf_data.hpp:69-86,95-104 ---------------------------------
def _sort_cases():
    """(qs, qe) arrays with heavy duplicate keys around libstdc++'s thresholds (hazard B1)."""
    rng = np.random.default_rng(17)
    out = []
    for n in (1, 2, 3, 15, 16, 17, 18, 31, 32, 33, 64, 100, 257, 1000, 1025, 2600, 5000):
        for distinct in (1, 2, 5, max(2, n // 3), 10 * n):
            qs = rng.integers(0, distinct, n).astype(np.int64)
            qe = qs + rng.integers(0, 3, n).astype(np.int64)
            out.append((qs, qe))
    base = np.arange(300, dtype=np.int64)
    for qs in (base, base[::-1].copy(), base % 7, base // 5, np.where(base % 2 == 0, base, 300 - base)):
        qs = np.ascontiguousarray(qs, dtype=np.int64)
        out.append((qs, qs + 1))
    return out


def _ref_perm(R, qs, qe):
    perm = np.zeros(len(qs), np.int32)
    R.ref_sort_perm.restype = C.c_int64
    R.ref_sort_perm(qs.ctypes.data_as(C.c_void_p), qe.ctypes.data_as(C.c_void_p), C.c_int64(len(qs)), perm.ctypes.data_as(C.c_void_p))
    return perm


def test_std_sort_over_real_pafreaddata(T):
    """paf_data.cpp:232,241: std::sort of a copy of std::vector<PafReadData> with the real
    operator< -- against the oracle's std::sort over its own Rec and against the replay the
    product's kb_sort_fix runs (same kernel body, host build)."""
    from test_sort_replay import _kernel, _replay, _std
    for mono in (True, False):
        R = _need(T, mono)
        for qs, qe in _sort_cases():
            want = _ref_perm(R, qs, qe)
            assert np.array_equal(_std(T, qs, qe), want), (len(qs), mono)
            assert np.array_equal(_replay(T, qs, qe), want), (len(qs), mono)
            if mono and len(qs) > 1:
                assert np.array_equal(_kernel(T, qs, qe), want), len(qs)     # kb_sort_fix (LDS work-list form up to 3072 records, sequential beyond)


def test_read_predicates_truth_table(T):
    R, O = _need(T), T.oracle()
    vals = range(0, 7)
    n = 0
    for a_qs in vals:
        for a_qe in range(a_qs, 7):
            for b_qs in vals:
                for b_qe in range(b_qs, 7):
                    args = [C.c_int64(x) for x in (a_qs, a_qe, b_qs, b_qe)]
                    assert R.ref_read_lt(*args) == O.oracle_read_lt(*args)
                    assert R.ref_qry_contains(*args) == O.oracle_qry_contains(*args)
                    assert R.ref_qry_partial_overlap(*args) == O.oracle_qry_partial_overlap(*args)
                    # what K2 (kb_ov_merge) evaluates on SORTED neighbours: for a <= b in sort order with
                    # b.qs <= a.qe, partial overlap  <=>  a.qs < b.qs and a.qe < b.qe
                    if (a_qs, a_qe) <= (b_qs, b_qe) and b_qs <= a_qe:
                        assert bool(R.ref_qry_partial_overlap(*args)) == (a_qs < b_qs and a_qe < b_qe)
                    n += 1
    assert n == 28 * 28


def test_output_constructor(T):
    R, O = _need(T), T.oracle()
    rng = np.random.default_rng(4)
    for _ in range(200):
        a = rng.integers(-5, 1 << 40, 5).astype(np.int64)
        a[0] = int(rng.integers(0, 1 << 20))
        r, o = np.zeros(12, np.int64), np.zeros(6, np.int64)
        R.ref_output_from_read(a.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p))
        O.oracle_output_from_read(a.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p))
        assert np.array_equal(r[:6], o) and np.array_equal(r[:6], [a[0], a[1], a[2], a[3], a[4], 0])
        assert np.array_equal(r[6:], [-1, 0, 0, 0, 0, 0])          # default constructor (:95-97)
