"""Dial's bucketed BFS, k_weighted_bfs() (k_weighted_bfs.hpp:16-37) - north_star's "LDS-staged ... buckets and wavefront
ballot / prefix-sum for frontier compaction".  The solver runs it with lim = 2 on the anomaly weights and keeps one scalar
(paf_data.cpp:704-715), which the pipeline folds into its forward sweep (checked there as anom_dis[dest]); the product also carries
the algorithm itself, aasm_sssp_dial: buckets as LIFO stacks staged in LDS, a popped row relaxed by the lanes, pushes compacted per
bucket by ballot + prefix count.  dist is order-independent, pre is not (the first vertex in the reference's pop order that reaches
the final distance), so equality of pre means the kernel keeps the reference's LIFO order.
CPU tier: the oracle's restatement against the REAL header (oracle/_ref) and against recorded vectors (tests/golden/ref_dial.npz,
written by this file's main()); GPU tier: aasm_sssp_dial against both, on digraphs with cycles, parallel edges, zero-cost cycles,
unreachable parts, rows longer than a wave and stacks deeper than the LDS window."""
import ctypes as C
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "ref_dial.npz")


def random_digraph(rng, n, p, lim, parallel=0.2, fan=0):
    rows = [[] for _ in range(n)]
    for u in range(n):
        for v in range(n):
            if u != v and rng.random() < p:
                rows[u].append((v, int(rng.integers(0, lim + 1))))
        for _ in range(fan if u < 2 else 0):                          # a few rows longer than a wave
            rows[u].append((int(rng.integers(0, n)), int(rng.integers(0, lim + 1))))
        rng.shuffle(rows[u])                                          # list order is significant
        if rows[u] and rng.random() < parallel:
            rows[u].append((rows[u][0][0], int(rng.integers(0, lim + 1))))   # a parallel edge, maybe with another cost
    rp = np.zeros(n + 1, np.int64)
    col, cost = [], []
    for u in range(n):
        for v, c in rows[u]:
            col.append(v); cost.append(c)
        rp[u + 1] = len(col)
    return rp, np.array(col, np.int64), np.array(cost, np.int64)


def cases():
    rng = np.random.default_rng(7)
    out = []
    for n, p, lim, fan in ((2, 1.0, 2, 0), (9, 0.3, 2, 0), (30, 0.15, 2, 0), (60, 0.08, 2, 0), (120, 0.04, 2, 0), (50, 0.1, 0, 0), (50, 0.1, 1, 0), (80, 0.06, 5, 0),
                           (90, 0.05, 7, 0), (300, 0.01, 2, 150), (1500, 0.002, 2, 900), (2500, 0.0, 2, 2400)):
        for _ in range(2):
            rp, col, cost = random_digraph(rng, n, p, lim, fan=fan)
            out.append((n, rp, col, cost, int(rng.integers(0, min(n, 2) if fan else n)), lim))
    return out


def run(lib, prefix, n, rp, col, cost, src, lim):
    dist, pre = np.zeros(n, np.int64), np.zeros(n, np.int64)
    P = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
    fn = getattr(lib, prefix + "dial_bfs")
    fn.restype = C.c_int64
    assert fn(C.c_int64(n), P(rp), P(col), P(cost), C.c_int64(src), C.c_int64(lim), P(dist), P(pre)) == n
    return dist, pre


@pytest.mark.ref
def test_oracle_dial_equals_the_real_header(T):
    R = T.ref(False)
    if R is None:
        pytest.skip("oracle/_ref not built")
    for n, rp, col, cost, src, lim in cases():
        dr, pr = run(R, "ref_", n, rp, col, cost, src, lim)
        do, po = run(T.oracle(), "oracle_", n, rp, col, cost, src, lim)
        assert np.array_equal(dr, do) and np.array_equal(pr, po)


def test_oracle_dial_matches_recorded_reference_vectors(T):
    z = np.load(GOLD)
    cs = cases()
    assert int(z["n_cases"]) == len(cs)
    reached = 0
    for i, (n, rp, col, cost, src, lim) in enumerate(cs):
        assert np.array_equal(z[f"c{i}_rp"], rp) and np.array_equal(z[f"c{i}_col"], col)      # the generator still makes the recorded graphs
        do, po = run(T.oracle(), "oracle_", n, rp, col, cost, src, lim)
        assert np.array_equal(do, z[f"c{i}_dist"]) and np.array_equal(po, z[f"c{i}_pre"]), i
        reached += int((do >= 0).sum())
    assert reached > 3000


@pytest.mark.gpu
def test_hip_dial_matches_reference_vectors_and_oracle(T):
    api = T.api()
    z = np.load(GOLD)
    cs = cases()
    # one batch with every graph (a wave each), and every graph alone
    voff, rps, cols, costs, srcs = [0], [np.zeros(1, np.int64)], [], [], []
    for n, rp, col, cost, src, lim in cs:
        if lim != 2:
            continue
        rps.append(rp[1:] + rps[-1][-1]); cols.append(col); costs.append(cost); srcs.append(src); voff.append(voff[-1] + n)
    dist, pre = api.sssp_dial(np.array(voff), np.concatenate(rps), np.concatenate(cols), np.concatenate(costs), np.array(srcs), lim=2)
    k = 0
    for i, (n, rp, col, cost, src, lim) in enumerate(cs):
        d1, p1 = api.sssp_dial(np.array([0, n]), rp, col, cost, np.array([src]), lim=lim)
        assert np.array_equal(d1, z[f"c{i}_dist"]) and np.array_equal(p1, z[f"c{i}_pre"]), ("alone", i)
        do, po = run(T.oracle(), "oracle_", n, rp, col, cost, src, lim)
        assert np.array_equal(d1, do) and np.array_equal(p1, po)
        if lim == 2:
            a, b = voff[k], voff[k + 1]
            assert np.array_equal(dist[a:b], z[f"c{i}_dist"]) and np.array_equal(pre[a:b], z[f"c{i}_pre"]), ("batch", i)
            k += 1


@pytest.mark.gpu
def test_hip_dial_on_a_contig_graph_gives_the_pipeline_scalar(T):
    """The solver's own use (paf_data.cpp:704-715): anomaly weights of a contig's graph, lim = 2 - dist[dest] is the anom_dis[dest]
    the pipeline's forward sweep computes, and the oracle's."""
    api = T.api()
    hb = T.synth(3, 300, 5, dup_every=4)
    for c in range(3):
        o = T.oracle_debug(hb, c, 4, False)
        rp, col, anom = o["csr_rowptr"], o["csr_col"], o["csr_w_anom"]
        n = len(rp) - 1
        dist, pre = api.sssp_dial(np.array([0, n]), rp, col, anom, np.array([n - 2]), lim=2)
        assert dist[n - 1] == o["anom_dis_dest"][0]
        do, po = run(T.oracle(), "oracle_", n, rp, col, anom, n - 2, 2)
        assert np.array_equal(dist, do) and np.array_equal(pre, po)


@pytest.mark.gpu
def test_hip_dial_rejects_costs_beyond_lim(T):
    api = T.api()
    with pytest.raises(api.AlignasmError):
        api.sssp_dial(np.array([0, 2]), np.array([0, 1, 1]), np.array([1]), np.array([3]), np.array([0]), lim=2)


def main():
    """python tests/test_dial.py: records tests/golden/ref_dial.npz from the REAL header (oracle/_ref/libaasm_ref_algos.so)."""
    import sys
    sys.path[:0] = [os.path.dirname(HERE), HERE]
    import aasm_testlib as T
    R = T.ref(False)
    assert R is not None, "make -C oracle (needs /root/reference)"
    out = {}
    cs = cases()
    for i, (n, rp, col, cost, src, lim) in enumerate(cs):
        d, p = run(R, "ref_", n, rp, col, cost, src, lim)
        out[f"c{i}_rp"] = rp; out[f"c{i}_col"] = col.astype(np.int32); out[f"c{i}_dist"] = d.astype(np.int32); out[f"c{i}_pre"] = p.astype(np.int32)
    out["n_cases"] = np.array(len(cs))
    out["source"] = np.array(["reference k_weighted_bfs.hpp:16-37 via oracle/_ref/libaasm_ref_algos.so (ref_dial_bfs)"])
    np.savez_compressed(GOLD, **out)
    print("wrote", GOLD, os.path.getsize(GOLD), "bytes,", len(cs), "cases")


if __name__ == "__main__":
    main()
