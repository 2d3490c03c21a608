"""CPU tier: the product's kernel bodies (1-lane host emulation, tests/host_emul) against
the oracle -- final outputs bit-exact and every intermediate (sorted order, parts, pair
cuts, CSR rows + weights, shortest-path tree, Kahn orders, heaps, k distances)."""
import numpy as np
import pytest

CASES = [
    # (contigs, recs, seed, K, dense, dup_every, shuffle, heavy_tail, nsl)
    (10, 100, 1, 10000, False, 0, False, False, False),      # BASELINE config C1 shape
    (3, 700, 11, 4, False, 0, False, False, False),
    (3, 300, 31, 16, True, 0, False, False, False),
    (2, 300, 31, 10000, True, 0, False, False, False),
    (4, 300, 5, 10000, False, 3, False, False, False),       # co-optimal ties (.all paths)
    (6, 200, 7, 10000, False, 0, False, False, True),        # NON_SKIP_LINKABLE
    (3, 250, 8, 10000, True, 0, False, False, True),
    (6, 150, 9, 10000, False, 3, True, False, False),        # shuffled input + duplicate keys (std::sort replay)
    (40, 50, 10, 1, False, 0, False, True, False),           # ragged sizes incl. tiny contigs
    (30, 40, 10, 10000, True, 0, True, True, False),
    (5, 1, 3, 10000, False, 0, False, False, False),         # single-record contigs
    (5, 2, 3, 10000, False, 0, False, False, False),
    (8, 40, 13, 10000, True, 1, True, False, False),         # every record duplicated
    (2, 2600, 17, 4, False, 3, True, False, False),          # contigs longer than one sort chunk (3 x 1024), shuffled + duplicates
    (3, 1024, 19, 1, False, 0, True, False, False),          # exactly one full chunk
    (2, 1025, 23, 1, False, 5, True, False, False),          # one record into the second chunk
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "c%dx%d_s%d_k%d_%s%s%s%s%s" % (
    c[0], c[1], c[2], c[3], "D" if c[4] else "S", f"_dup{c[5]}" if c[5] else "", "_shuf" if c[6] else "", "_ht" if c[7] else "", "_nsl" if c[8] else ""))
def test_outputs_and_intermediates(T, case):
    nc, nr, seed, K, dense, dup, shuf, heavy, nsl = case
    hb = T.synth(nc, nr, seed, dense=dense, dup_every=dup, shuffle=shuf, heavy_tail=heavy)
    want = T.oracle_solve(hb, K, nsl)
    got = T.emul_solve(hb, K, nsl)
    assert T.diff_outputs(want, got) == []
    assert T.diff_intermediates(hb, T.emul_debug, K, nsl) == []
    got = T.emul_solve(hb, K, nsl, heap_waves="all", heap_input_order=True)   # (default: largest node bound first, a block per contig of the class)
    assert T.diff_outputs(want, got) == []
    assert want["stats"]["n_internal_errors"] == 0


@pytest.mark.parametrize("case", [CASES[0], CASES[4], CASES[7], CASES[12]], ids=lambda c: "seq_c%dx%d_s%d" % (c[0], c[1], c[2]))
def test_sequential_select_fallback(T, case):
    """The one-wave-per-contig selection kernel (fallback for huge conversion counts)."""
    nc, nr, seed, K, dense, dup, shuf, heavy, nsl = case
    hb = T.synth(nc, nr, seed, dense=dense, dup_every=dup, shuffle=shuf, heavy_tail=heavy)
    want = T.oracle_solve(hb, K, nsl)
    got = T.emul_solve(hb, K, nsl, sequential_select=True)
    assert T.diff_outputs(want, got) == []


@pytest.mark.parametrize("case", [CASES[0], CASES[2], CASES[4], CASES[9], CASES[10], CASES[12]], ids=lambda c: "mw_c%dx%d_s%d" % (c[0], c[1], c[2]))
def test_multiwave_heap_kernel_is_bit_identical(T, case):
    """kb_heap_mw (several waves per contig: BFS numbering, per-vertex regions of a provisional arena, ticket
    queue, compaction into BFS order) forced on every contig: final outputs AND the heap arena (indices
    included), roots and k distances equal the oracle's, as with the one-wave kernel.  (One lane here: the
    logic; the GPU tier runs it with four real waves.)"""
    nc, nr, seed, K, dense, dup, shuf, heavy, nsl = case
    hb = T.synth(nc, nr, seed, dense=dense, dup_every=dup, shuffle=shuf, heavy_tail=heavy)
    want = T.oracle_solve(hb, K, nsl)
    got = T.emul_solve(hb, K, nsl, heap_waves="all")
    assert T.diff_outputs(want, got) == []
    assert T.diff_intermediates(hb, T.emul_debug, K, nsl) == []


@pytest.mark.parametrize("chain", ["none", "half", "all"])
@pytest.mark.parametrize("case", [CASES[0], CASES[2], CASES[4], CASES[7], CASES[9]], ids=lambda c: "chain_c%dx%d_s%d" % (c[0], c[1], c[2]))
def test_chain_class_forms(T, case, chain):
    """The chain class (kb_chain: a contig's sweep, pre-pass and heaps in one workgroup; the default for the sparse contigs of
    batches this small) against the three-launch form (`none`) and a split of the batch between the two (`half`): outputs and every
    intermediate.  One lane and the three roles one after the other here - the host logic, the work lists and the count-downs;
    the GPU tier runs the three waves beside each other."""
    nc, nr, seed, K, dense, dup, shuf, heavy, nsl = case
    hb = T.synth(nc, nr, seed, dense=dense, dup_every=dup, shuffle=shuf, heavy_tail=heavy)
    want = T.oracle_solve(hb, K, nsl)
    got = T.emul_solve(hb, K, nsl, chain=chain, chain_own_queue=True)   # the heap wave with its own BFS queue (the form for classes of more than 896 contigs)
    assert T.diff_outputs(want, got) == []
    assert T.diff_intermediates(hb, T.emul_debug, K, nsl) == []
    got = T.emul_solve(hb, K, nsl, chain=chain)                          # the order from a wave of its own (default)
    assert T.diff_outputs(want, got) == []
    assert T.diff_intermediates(hb, T.emul_debug, K, nsl) == []
    got = T.emul_solve(hb, K, nsl, chain=chain, test_small_root_ring=True)   # ... with a 4-entry ring of roots: parents beyond it (h_root), as on frontiers wider than 512
    assert T.diff_outputs(want, got) == []
    assert T.diff_intermediates(hb, T.emul_debug, K, nsl) == []
    n_class = int(T.emul_debug("counters", np.int64)[17])
    sizes = np.diff(hb.arrays["ctg_rec_off"])
    if chain == "none" or dense:
        assert n_class == 0
    elif chain == "all":
        assert n_class == int((sizes > 1).sum())
    else:
        assert n_class == int(((sizes >= -(-int(sizes.sum()) // len(sizes))) & (sizes > 1)).sum())


@pytest.mark.parametrize("case", [CASES[0], CASES[1], CASES[4], CASES[5], CASES[7], CASES[8], CASES[14], (2, 2000, 5, 4, False, 0, False, False, False), (1, 2450, 7, 1, False, 0, True, False, True), (12, 700, 3, 4, False, 0, False, True, False)], ids=lambda c: "graph_c%dx%d_s%d%s" % (c[0], c[1], c[2], "_nsl" if c[8] else ""))
def test_graph_build_forms(T, case):
    """Rows, reversed CSR and the sweeps' vertex headers of a contig by one workgroup (kb_graph_build: the default where the batch is
    sparse and every contig has at most 1 792 vertices and 4 096 edges, or - the form with more LDS - 3 584 and 8 192; the last case: a heavy-tailed batch with
    contigs of both forms and one beyond them, which the separate launches build beside them) against the separate launches (row_fill, scan, rev_fill,
    rev_place, rev_hdr): the outputs, and every array the later kernels read, byte for byte."""
    nc, nr, seed, K, dense, dup, shuf, heavy, nsl = case
    hb = T.synth(nc, nr, seed, dense=dense, dup_every=dup, shuffle=shuf, heavy_tail=heavy)
    want = T.oracle_solve(hb, K, nsl)
    names = [("csr_col", np.int32), ("csr_w_qry", np.int64), ("csr_w_ref", np.int32), ("csr_w_flags", np.uint8), ("rptr", np.int64), ("r_pk", np.int32),
             ("rvh", np.int32), ("fvh", np.int32), ("cnt_tmp2", np.int32)]
    got = {}
    for form in (False, True):
        out = T.emul_solve(hb, K, nsl, graph_launches=form)
        assert T.diff_outputs(want, out) == []
        assert T.diff_intermediates(hb, T.emul_debug, K, nsl) == []
        got[form] = {n: T.emul_debug(n, dt).copy() for n, dt in names}
        n_s, n_l, n_rest = (int(x) for x in T.emul_debug("counters", np.int64)[18:21])   # contigs by form: one workgroup (small / large), separate launches
        assert (n_s + n_l == 0) if form else (n_s + n_l > 0)
        assert (T.emul().emul_debug_fetch(b"indeg", None, 0) >= 0) == (n_rest > 0)   # (global in-degree counters only where the separate launches have contigs)
        if heavy and nr >= 700 and not form:
            assert n_s > 0 and n_l > 0 and n_rest > 0                # the mixed batch: all three at once
    for n, _ in names:
        assert np.array_equal(got[False][n], got[True][n]), n
    # ... and the 16-hop jump records of K9's recovery, which the small contigs get from a workgroup with the contig's tree in LDS
    # (kb_tnx16_wg) - with the chain class off, or batches this small never reach it
    got7 = {}
    for form in (False, True):
        out = T.emul_solve(hb, K, nsl, graph_launches=form, chain="none")
        assert T.diff_outputs(want, out) == []
        got7[form] = T.emul_debug("tnx16", np.int32).copy()
    assert np.array_equal(got7[False], got7[True])


def test_all_pool_overflow_reruns_the_pick(T):
    """Every record duplicated: hundreds of co-optimal walks per contig, the `.all` lists hold 130x the records - far beyond the pool the
    first pick is given (R + 1 024 elements).  The pick is re-run with the exact size (one read-back brings the demand and the output
    totals; the totals are read again after the re-run)."""
    hb = T.synth(3, 30, 5, dup_every=1)
    want = T.oracle_solve(hb, 10000)
    assert len(want["all"]) > len(hb.arrays["qry_str"]) + 1024
    assert T.diff_outputs(want, T.emul_solve(hb, 10000)) == []
