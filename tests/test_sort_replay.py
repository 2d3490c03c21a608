"""kb_sort_fix replays libstdc++'s (unstable) std::sort so that records with duplicate
(qry_str, qry_end) keys land in the order the reference's std::sort gives them
(SURVEY.md hazard B1).  Checked here against std::sort itself (via the oracle library)."""
import ctypes as C

import numpy as np
import pytest


def _replay(T, qs, qe, depth=-1):
    idx = np.arange(len(qs), dtype=np.int32)
    T.emul().emul_std_sort_replay(idx.ctypes.data_as(C.c_void_p), C.c_int64(len(qs)), qs.ctypes.data_as(C.c_void_p),
                                  qe.ctypes.data_as(C.c_void_p), C.c_int(depth))
    return idx


def _kernel(T, qs, qe):
    """kb_sort_fix as the pipeline runs it for one contig (host build of the kernel body)."""
    perm = np.full(len(qs), -1, np.int32)
    T.emul().emul_sort_fix_kernel(perm.ctypes.data_as(C.c_void_p), C.c_int64(len(qs)), qs.ctypes.data_as(C.c_void_p), qe.ctypes.data_as(C.c_void_p))
    return perm


def _kernel_depth(T, qs, qe, depth):
    """kb_sort_fix with `depth` partition levels before the heap sort fallback."""
    perm = np.full(len(qs), -1, np.int32)
    T.emul().emul_sort_fix_kernel2(perm.ctypes.data_as(C.c_void_p), C.c_int64(len(qs)), qs.ctypes.data_as(C.c_void_p), qe.ctypes.data_as(C.c_void_p),
                                   C.c_int(depth + 1))
    return perm


def _std(T, qs, qe, heap_only=0):
    perm = np.zeros(len(qs), np.int32)
    T.oracle().oracle_std_sort_perm(qs.ctypes.data_as(C.c_void_p), qe.ctypes.data_as(C.c_void_p), C.c_int64(len(qs)),
                                    perm.ctypes.data_as(C.c_void_p), C.c_int(heap_only))
    return perm


@pytest.mark.parametrize("n", [1, 2, 16, 17, 18, 33, 100, 257, 1000, 5000])
def test_replay_equals_std_sort_with_heavy_duplicates(T, n):
    rng = np.random.default_rng(n)
    for distinct in (2, 5, max(2, n // 3), 10 * n):
        qs = rng.integers(0, distinct, n).astype(np.int64)
        qe = qs + rng.integers(0, 3, n).astype(np.int64)
        assert np.array_equal(_replay(T, qs, qe), _std(T, qs, qe)), (n, distinct)
        if n > 1:
            assert np.array_equal(_kernel(T, qs, qe), _std(T, qs, qe)), (n, distinct)


def test_replay_orders_patterns(T):
    for n in (40, 300, 2048):
        base = np.arange(n, dtype=np.int64)
        for qs in (base, base[::-1].copy(), (base % 7), (base // 5), np.where(base % 2 == 0, base, n - base)):
            qs = np.ascontiguousarray(qs, dtype=np.int64)
            qe = qs + 1
            assert np.array_equal(_replay(T, qs, qe), _std(T, qs, qe))
            assert np.array_equal(_kernel(T, qs, qe), _std(T, qs, qe))


def test_heapsort_fallback_equals_partial_sort(T):
    """Introsort's depth-limit fallback (std::__partial_sort(first, last, last))."""
    rng = np.random.default_rng(5)
    for n in (17, 64, 333, 2000):
        qs = rng.integers(0, n // 4 + 2, n).astype(np.int64)
        qe = qs + rng.integers(0, 2, n).astype(np.int64)
        assert np.array_equal(_replay(T, qs, qe, depth=0), _std(T, qs, qe, heap_only=1))


@pytest.mark.parametrize("n", [17, 95, 96, 200, 1536, 1537, 4000, 20000])
def test_kernel_depth_limit_equals_sequential_replay(T, n):
    """The kernel's three range forms (wave in global scratch, wave in LDS, lane per range) at every depth limit,
    against the sequential replay (itself pinned to std::sort and to std::partial_sort above)."""
    rng = np.random.default_rng(n)
    for distinct in (3, max(2, n // 3), 4 * n):
        qs = rng.integers(0, distinct, n).astype(np.int64)
        qe = qs + rng.integers(0, 3, n).astype(np.int64)
        for depth in (0, 1, 2, 3, 5, 8):
            assert np.array_equal(_kernel_depth(T, qs, qe, depth), _replay(T, qs, qe, depth=depth)), (n, distinct, depth)
        assert np.array_equal(_kernel(T, qs, qe), _std(T, qs, qe)), (n, distinct)


def _gpu_replay(T, rec_off, qs, qe, depth_test=0):
    api = T.api()
    perm = np.full(len(qs), -1, np.int32)
    rc = api.LIB.aasm_debug_sort_replay(rec_off.ctypes.data_as(C.c_void_p), C.c_int64(len(rec_off) - 1), qs.ctypes.data_as(C.c_void_p),
                                        qe.ctypes.data_as(C.c_void_p), perm.ctypes.data_as(C.c_void_p), C.c_int(depth_test), C.c_int(0))
    assert rc == 0, api.LIB.aasm_last_error()
    return perm


@pytest.mark.gpu
def test_gpu_replay_equals_std_sort(T):
    """The 64-lane kernel on the card against std::sort itself: every range form (wave in global scratch, wave in LDS,
    lane per range), key sets from two distinct values to all distinct, ordered / reversed / organ-pipe inputs."""
    rng = np.random.default_rng(2026)
    sizes = [17, 18, 63, 64, 65, 95, 96, 97, 128, 200, 333, 1000, 1535, 1536, 1537, 1600, 3000, 5000, 20000, 70000]
    qs_l, qe_l, off = [], [], [0]
    for n in sizes:
        base = np.arange(n, dtype=np.int64)
        shapes = [rng.integers(0, d, n) for d in (2, 5, max(2, n // 3), 10 * n)]
        shapes += [base, base[::-1], base % 7, base // 5, np.where(base % 2 == 0, base, n - base), np.zeros(n, np.int64)]
        for q in shapes:
            q = np.ascontiguousarray(q, dtype=np.int64)
            qs_l.append(q); qe_l.append(q + rng.integers(0, 3, n)); off.append(off[-1] + n)
    qs = np.concatenate(qs_l); qe = np.ascontiguousarray(np.concatenate(qe_l), dtype=np.int64); off = np.array(off, np.int64)
    got = _gpu_replay(T, off, qs, qe)
    for c in range(len(off) - 1):
        a, b = off[c], off[c + 1]
        want = _std(T, np.ascontiguousarray(qs[a:b]), np.ascontiguousarray(qe[a:b]))
        assert np.array_equal(got[a:b], want), (c, b - a)


@pytest.mark.gpu
def test_gpu_replay_depth_limit(T):
    """Heap sort fallbacks of all three range forms on the card (depth limits 0 .. 8) against the sequential replay."""
    rng = np.random.default_rng(7)
    for n in (17, 96, 200, 1536, 1537, 4000, 20000):
        for distinct in (3, max(2, n // 3), 4 * n):
            qs = rng.integers(0, distinct, n).astype(np.int64)
            qe = qs + rng.integers(0, 3, n).astype(np.int64)
            off = np.array([0, n], np.int64)
            for depth in (0, 1, 2, 3, 5, 8):
                assert np.array_equal(_gpu_replay(T, off, qs, qe, depth + 1), _replay(T, qs, qe, depth=depth)), (n, distinct, depth)
