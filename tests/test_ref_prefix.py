"""K1 ... K8 of solve_ctg_read() pinned to the reference's OWN statements.

oracle/_ref/libaasm_ref_prefix[_mono].so is /root/reference/src/paf_data.cpp:1-738 compiled from where it lies
(piped to g++ up to the first line that names an ankerl type; oracle/Makefile), i.e. the real solve_ctg_read()
from `std::sort` through `k_walk_solver.k_shortest_walks(src, dest, MAX_PATH_COUNT)`, closed by our epilogue
(oracle/ref_prefix_epilogue.inc) that copies its locals out.  Against it, array by array:

  `ref` tier (live, where the library exists):  oracle/alignasm_oracle.cpp  AND  the product's kernel bodies in the
      1-lane host emulation - the sorted order, parts, every (i, j) cut of the N x N tables, vertex ids, adjacency
      order, all five weight fields, anom_dis[dest], d / best, both Kahn orders, every heap node and root, all
      10 000 distances - on the fuzz generator's adversarial shapes, C1, C2-sized contigs, dense / duplicated /
      shuffled / heavy-tailed files, NON_SKIP_LINKABLE on and off, with the bump-allocator flavour (hazard B3);
      the glibc flavour (what the shipped binary does) must agree on everything but the order inside distance ties;
  CPU + GPU tiers (anywhere): the same comparison against VECTORS recorded from that library
      (tests/golden/ref_prefix.npz, written by tests/golden/make_ref_prefix.py) - tests/test_ref_prefix_golden.py.
"""
import numpy as np
import pytest

from test_fuzz import make_batch

pytestmark = pytest.mark.ref


def _need(T, mono=True):
    lib = T.ref_prefix(mono)
    if lib is None:
        pytest.skip("oracle/_ref/libaasm_ref_prefix*.so not built (no /root/reference on this box and no prebuilt .so)")
    return lib


def _diff_oracle(T, hb, contigs, nsl, n_paths=0):
    """oracle_debug vs the reference prefix on every recorded array; returns (mismatches, #distances, #pair vertices)."""
    bad, nd, npairs = [], 0, 0
    for c in contigs:
        if hb.arrays["ctg_rec_off"][c + 1] - hb.arrays["ctg_rec_off"][c] <= 1:
            continue
        r = T.ref_prefix_debug(hb, c, nsl=nsl, n_paths=n_paths)
        o = T.oracle_debug(hb, c, 10000, nsl)
        assert "vtx_index_mismatch" not in r
        assert r["n_index_entries"][0] == len(r["vtx_i"])              # the N x N index table holds exactly the listed vertices
        for n in T.PREFIX_NAMES:
            if not np.array_equal(o[n], r[n]):
                bad.append((c, n))
        nd += len(r["kd_qry"]); npairs += len(r["pair_pe_q"])
    return bad, nd, npairs


@pytest.mark.parametrize("style", [0, 1, 2])
def test_oracle_matches_reference_prefix_on_adversarial_batches(T, style):
    """The fuzz generator's shapes (containment, equal starts, duplicates, few-base records, unmatched bases between
    ranges): K2's merge body, K3 / K4 and the rest, both NON_SKIP_LINKABLE settings."""
    _need(T)
    nd = npairs = 0
    for seed in range(40):
        hb = make_batch(seed, 6, 30, 400, style)
        for nsl in (False, True):
            bad, a, b = _diff_oracle(T, hb, range(6), nsl)
            assert bad == [], (seed, style, nsl, bad[:4])
            nd += a; npairs += b
    assert nd > 20000 and (npairs > 500 or style == 1)


CASES = [
    # (contigs, recs, seed, dense, dup_every, shuffle, heavy_tail, nsl)
    (10, 100, 1, False, 0, False, False, False),       # C1 (BASELINE configs[0]) in full
    (3, 1000, 11, False, 0, False, False, False),      # C2-sized contigs (seed of configs[1])
    (2, 1000, 21, False, 0, False, False, False),      # C3-sized contigs (seed of configs[2])
    (2, 600, 31, True, 0, False, False, False),        # C5's dense graphs
    (2, 400, 31, True, 0, False, False, True),
    (4, 300, 5, False, 3, False, False, False),        # duplicated records: ties everywhere
    (6, 150, 9, False, 3, True, False, False),         # shuffled input + duplicate keys: the unstable std::sort itself
    (6, 200, 7, False, 0, False, False, True),         # NON_SKIP_LINKABLE
    (30, 40, 10, True, 0, True, True, False),          # ragged sizes
    (8, 40, 13, True, 1, True, False, False),          # every record duplicated
    (5, 2, 3, False, 0, False, False, False),
    (1, 2600, 17, False, 3, True, False, False),       # longer than one sort chunk of the product
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "c%dx%d_s%d_%s%s%s%s%s" % (
    c[0], c[1], c[2], "D" if c[3] else "S", f"_dup{c[4]}" if c[4] else "", "_shuf" if c[5] else "", "_ht" if c[6] else "", "_nsl" if c[7] else ""))
def test_oracle_matches_reference_prefix_on_config_shapes(T, case):
    _need(T)
    nc, nr, seed, dense, dup, shuf, heavy, nsl = case
    hb = T.synth(nc, nr, seed, dense=dense, dup_every=dup, shuffle=shuf, heavy_tail=heavy)
    bad, nd, _ = _diff_oracle(T, hb, range(nc), nsl)
    assert bad == [], bad[:6]
    assert nd > 0


@pytest.mark.parametrize("case", [CASES[0], CASES[3], CASES[5], CASES[6], CASES[7], CASES[9]], ids=lambda c: "emul_c%dx%d_s%d" % (c[0], c[1], c[2]))
def test_kernel_bodies_match_reference_prefix(T, case):
    """The PRODUCT's kernel bodies (1-lane host emulation; the GPU tier runs the same comparison on the card against the
    recorded vectors) against the reference prefix directly, no oracle in between."""
    _need(T)
    nc, nr, seed, dense, dup, shuf, heavy, nsl = case
    hb = T.synth(nc, nr, seed, dense=dense, dup_every=dup, shuffle=shuf, heavy_tail=heavy)
    T.emul_solve(hb, 10000, nsl)
    assert T.diff_intermediates(hb, T.emul_debug, 10000, nsl, expect=lambda c: T.ref_prefix_debug(hb, c, nsl=nsl)) == []
    T.emul_solve(hb, 16, nsl)                                      # a smaller K sees a prefix of the reference's 10 000
    assert T.diff_intermediates(hb, T.emul_debug, 16, nsl, expect=lambda c: T.ref_prefix_debug(hb, c, nsl=nsl)) == []


def test_recovered_walks_are_paths_of_the_reference_graph(T):
    """kth_shortest_walk_recover on the reference's own graph: each walk runs src -> dest over edges of the graph, and
    its weights add up to the k-th distance."""
    _need(T)
    hb = T.synth(3, 200, 5, dup_every=3)
    for c in range(3):
        r = T.ref_prefix_debug(hb, c, n_paths=50)
        rp, col, wq, wr = r["csr_rowptr"], r["csr_col"], r["csr_w_qry"], r["csr_w_ref"]
        src, dest = r["src_dest"]
        for k in range(len(r["path_off"]) - 1):
            a, b = r["path_off"][k], r["path_off"][k + 1]
            u, v = r["path_u"][a:b], r["path_v"][a:b]
            assert u[0] == src and v[-1] == dest and np.array_equal(u[1:], v[:-1])
            for x, y, q, w in zip(u, v, r["path_w_qry"][a:b], r["path_w_ref"][a:b]):
                row = slice(rp[x], rp[x + 1])
                hit = np.nonzero(col[row] == y)[0]
                assert len(hit) == 1 and wq[row][hit[0]] == q and wr[row][hit[0]] == w
            assert r["path_w_qry"][a:b].sum() == r["kd_qry"][k] and r["path_w_ref"][a:b].sum() == r["kd_ref"][k]


def test_glibc_flavour_differs_only_inside_distance_ties(T):
    """The shipped allocator: the k-walk queue breaks ties on raw node addresses (k_shortest_walks.hpp:231, hazard B3).
    Everything up to the heaps is identical; the distance sequence is identical as a sequence of (sum, anom) keys."""
    _need(T, True); _need(T, False)
    hb = T.synth(4, 300, 5, dup_every=3)
    inv = 0
    for c in range(4):
        m = T.ref_prefix_debug(hb, c, mono=True)
        g = T.ref_prefix_debug(hb, c, mono=False)
        for n in T.PREFIX_NAMES:
            if n.startswith("kd_"):
                continue
            assert np.array_equal(m[n], g[n]), (c, n)
        assert np.array_equal(m["kd_qry"] + m["kd_ref"], g["kd_qry"] + g["kd_ref"]) and np.array_equal(m["kd_anom"], g["kd_anom"])
        inv += int((np.diff(g["heap_addr"]) < 0).sum())
        assert (np.diff(m["heap_addr"]) > 0).all()                 # bump allocator: address order == allocation order
    assert inv >= 0
