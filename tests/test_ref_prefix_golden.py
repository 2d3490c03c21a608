"""K1 ... K8 against vectors recorded from the REAL reference (no /root/reference needed at run time).

tests/golden/ref_prefix.npz holds input batches and, per contig, the locals of the reference's own solve_ctg_read()
at paf_data.cpp:738 - sorted order, part ids, every (i, j) cut of the N x N tables, vertex ids, adjacency order with
all weight fields, anom_dis[dest], d / best, both Kahn orders, every sidetrack-heap node and root, the k-walk
distances - recorded by tests/golden/make_ref_prefix.py from oracle/_ref/libaasm_ref_prefix_mono.so (the reference's
lines compiled in place; oracle/Makefile).  Checked against them here:
  CPU tier: the oracle (so "HIP == oracle" elsewhere means "== reference" for these stages) and the product's
            kernel bodies in the 1-lane emulation;
  GPU tier: the HIP kernels' intermediates on the card, at K = 10 000 and at small K, with either K7 kernel."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def V(T):
    return T.RefPrefixVectors()


def test_fixture_covers_the_stages(T, V):
    assert len(V.tags) >= 15
    n_pairs = n_nodes = n_dist = n_nsl = 0
    for tag in V.tags:
        hb, nsl, full = V.batch(tag)
        n_nsl += nsl
        for c in range(hb.n_contigs):
            r = V.contig(tag, c)
            if r:
                assert set(T.PREFIX_NAMES) <= set(r)
                n_pairs += len(r["pair_pe_q"]); n_nodes += int(r["heap_nodes"][0]); n_dist += len(r["kd_qry"])
    assert n_pairs > 2000 and n_nodes > 100000 and n_dist > 100000 and n_nsl >= 4


def test_oracle_matches_recorded_reference_prefix(T, V):
    for tag in V.tags:
        hb, nsl, full = V.batch(tag)
        for c in range(hb.n_contigs):
            r = V.contig(tag, c)
            if not r:
                continue
            o = T.oracle_debug(hb, c, 10000, nsl)
            assert len(o["kd_qry"]) == r["kfound"][0], (tag, c)
            for n in T.PREFIX_NAMES:
                want = r[n]
                assert np.array_equal(o[n][:len(want)] if n.startswith("kd_") else o[n], want), (tag, c, n)


def test_kernel_bodies_match_recorded_reference_prefix(T, V):
    for tag in V.tags:
        hb, nsl, full = V.batch(tag)
        for K in ((10000, 4) if full else (64,)):
            T.emul_solve(hb, K, nsl)
            assert T.diff_intermediates(hb, T.emul_debug, K, nsl, expect=lambda c: V.contig(tag, c)) == [], (tag, K)


@pytest.mark.gpu
@pytest.mark.parametrize("heap_waves,chain", [("auto", "auto"), ("all", "auto"), ("none", "auto"), ("auto", "none"), ("auto", "half")])
def test_hip_intermediates_match_recorded_reference_prefix(T, V, heap_waves, chain):
    """The HIP path's intermediates on the card against what the reference's own statements computed - with either K7 kernel
    forced, and with the sparse contigs in the chain class (aasm_k67_chain; the default for batches this small), in the three
    launches, or split between the two."""
    api = T.api()
    for tag in V.tags:
        hb, nsl, full = V.batch(tag)
        db = api.DeviceBatch(hb)
        for K in ((10000, 4) if full else (64, 1)):
            res = db.solve(max_paths=K, non_skip_linkable=nsl, keep_debug=True, heap_waves=heap_waves, chain=chain)
            bad = T.diff_intermediates(hb, res.debug, K, nsl, expect=lambda c: V.contig(tag, c))
            res.close()
            assert bad == [], (tag, K, bad[:6])
        db.close()
