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
