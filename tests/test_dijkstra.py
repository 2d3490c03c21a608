"""Row *J: the solver's generic SSSP, dijkstra() (k_shortest_walks.hpp:69-87).  The reference's CLI never reaches it
(is_dag = true, paf_data.cpp:728) but the solver class offers it for graphs with cycles and north_star names it.
CPU tier: the oracle's restatement against the REAL header (oracle/_ref) and against recorded vectors
(ref_algos.npz); GPU tier: the product's aasm_sssp_dijkstra against both, on digraphs WITH cycles and on contig
DAGs, where its distances must also be key-equivalent to the DAG relaxation the CLI path uses (K6)."""
import ctypes as C

import numpy as np
import pytest


def random_digraph(rng, n, p, cyclic=True):
    rows = [[] for _ in range(n)]
    for u in range(n):
        for v in range(n):
            if u != v and (cyclic or v > u) and rng.random() < p:
                rows[u].append((v, [int(rng.integers(0, 9)), int(rng.integers(0, 9)), int(rng.integers(0, 3)), int(rng.integers(0, 2)), 1]))
        rng.shuffle(rows[u])                    # list order is significant
        if rows[u] and rng.random() < 0.2:
            rows[u].append(rows[u][0])         # a parallel edge
    rp = np.zeros(n + 1, np.int64)
    col, w = [], []
    for u in range(n):
        for v, ww in rows[u]:
            col.append(v); w.extend(ww)
        rp[u + 1] = len(col)
    return rp, np.array(col, np.int64), np.array(w, np.int64)


def run(lib, prefix, n, rp, col, w, src):
    d, prv = np.zeros(5 * n, np.int64), np.zeros(n, np.int64)
    P = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
    fn = getattr(lib, prefix + "generic_dijkstra")
    fn.restype = C.c_int64
    assert fn(C.c_int64(n), P(rp), P(col), P(w), C.c_int64(src), P(d), P(prv)) == n
    return d.reshape(n, 5), prv


def cases():
    rng = np.random.default_rng(42)
    out = []
    for n, p, cyc in ((2, 1.0, True), (8, 0.3, True), (25, 0.15, True), (40, 0.1, False), (60, 0.08, True), (120, 0.04, True)):
        for _ in range(3):
            rp, col, w = random_digraph(rng, n, p, cyc)
            out.append((n, rp, col, w, int(rng.integers(0, n))))
    return out


@pytest.mark.ref
def test_oracle_dijkstra_equals_the_real_header(T):
    R = T.ref(True)
    if R is None:
        pytest.skip("oracle/_ref not built")
    for n, rp, col, w, src in cases():
        dr, pr = run(R, "ref_", n, rp, col, w, src)
        do, po = run(T.oracle(), "oracle_", n, rp, col, w, src)
        assert np.array_equal(dr, do) and np.array_equal(pr, po)


def test_oracle_dijkstra_matches_recorded_reference_vectors(T):
    import os
    z = np.load(os.path.join(T.GOLDEN, "ref_algos.npz"))
    for i in range(int(z["n_dj"][0])):
        n, src = (int(x) for x in z[f"dj{i}_meta"])
        do, po = run(T.oracle(), "oracle_", n, z[f"dj{i}_rowptr"].copy(), z[f"dj{i}_col"].copy(), z[f"dj{i}_w"].copy(), src)
        assert np.array_equal(do, z[f"dj{i}_d"]) and np.array_equal(po, z[f"dj{i}_prv"])


def _keys(d):
    """The CALC_SUM order's view of a distance: (score sum, anom, ratio); unreachable = None."""
    from fractions import Fraction
    return [None if (x[0] == -1 and x[1] == -1 and x[2] == -1) else (int(x[0] + x[1]), int(x[2]), Fraction(int(x[3]), int(x[4]) if x[4] else 1)) for x in d]


@pytest.mark.gpu
def test_device_dijkstra_equals_reference_on_cyclic_graphs(T):
    import os
    api = T.api()
    z = np.load(os.path.join(T.GOLDEN, "ref_algos.npz"))
    gs = [(int(z[f"dj{i}_meta"][0]), z[f"dj{i}_rowptr"], z[f"dj{i}_col"], z[f"dj{i}_w"], int(z[f"dj{i}_meta"][1]), z[f"dj{i}_d"], z[f"dj{i}_prv"]) for i in range(int(z["n_dj"][0]))]
    voff = np.concatenate([[0], np.cumsum([g[0] for g in gs])]).astype(np.int64)
    rowptr = np.concatenate([[0]] + [g[1][1:] + sum(len(h[2]) for h in gs[:i]) for i, g in enumerate(gs)]).astype(np.int64)
    col = np.concatenate([g[2] for g in gs]).astype(np.int32)
    w = np.concatenate([g[3] for g in gs])
    d, prev = api.sssp_dijkstra(voff, rowptr, col, w, [g[4] for g in gs])
    for i, g in enumerate(gs):                     # the whole batch in one launch, one wave per graph
        assert np.array_equal(d[voff[i]:voff[i + 1]], g[5]), i
        assert np.array_equal(prev[voff[i]:voff[i + 1]], g[6]), i


@pytest.mark.gpu
def test_device_dijkstra_on_contig_dags_is_key_equivalent_to_the_dag_relaxation(T):
    """On the DAGs the CLI path solves, dijkstra (from dest over the reversed graph, as the solver would call it
    with is_dag = false) gives distances that are equivalent, in the order the solver compares with, to the DAG
    relaxation's (K6 / k_shortest_walks.hpp:160-175) -- and equal to the oracle's dijkstra bit for bit."""
    api = T.api()
    hb = T.synth(3, 150, 9, dup_every=4)
    for c in range(3):
        n, rp, col, w = T.contig_graph(hb, c, 16)
        # reversed graph, in-lists in ascending (source, position) as k_shortest_walks.hpp:180-183 builds it
        rrows = [[] for _ in range(n)]
        for u in range(n):
            for e in range(rp[u], rp[u + 1]):
                rrows[col[e]].append((u, w[5 * e:5 * e + 5]))
        rrp = np.zeros(n + 1, np.int64); rcol = []; rw = []
        for v in range(n):
            for u, ww in rrows[v]:
                rcol.append(u); rw.extend(int(x) for x in ww)
            rrp[v + 1] = len(rcol)
        rcol, rw = np.array(rcol, np.int64), np.array(rw, np.int64)
        d, prev = api.sssp_dijkstra([0, n], rrp, rcol, rw, [n - 1])
        do, po = run(T.oracle(), "oracle_", n, rrp, rcol, rw, n - 1)
        assert np.array_equal(d, do) and np.array_equal(prev, po)
        o = T.oracle_debug(hb, c, 16)
        dag = np.stack([o["sp_d_qry"], o["sp_d_ref"], o["sp_d_anom"], o["sp_d_qnz"], o["sp_d_qtot"]], 1)
        assert _keys(d) == _keys(dag)


@pytest.mark.gpu
def test_device_dijkstra_rejects_bad_input(T):
    api = T.api()
    with pytest.raises(api.AlignasmError) as e:
        api.sssp_dijkstra([0, 2], [0, 1, 1], [5], [1, 1, 0, 0, 1], [0])
    assert e.value.code == -1
    with pytest.raises(api.AlignasmError) as e:
        api.sssp_dijkstra([0, 2], [0, 1, 1], [1], [-5, 1, 0, 0, 1], [0])
    assert e.value.code == -5
