"""GPU tier: the HIP path through the C-ABI against the oracle on the same seeded inputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [
    (10, 100, 1, 10000, False, 0, False, False, False),      # BASELINE config C1: 10 x 100, K as shipped
    (50, 1000, 11, 1, False, 0, False, False, False),        # BASELINE config C2 at full size: 50 x 1000 (single chromosome), K=1
    (6, 1000, 21, 4, False, 0, False, False, False),         # C3 shape
    (4, 600, 31, 16, True, 0, False, False, False),          # C5 shape (dense, K=16)
    (2, 300, 31, 10000, True, 0, False, False, False),
    (4, 300, 5, 10000, False, 3, False, False, False),
    (6, 200, 7, 10000, False, 0, False, False, True),
    (3, 250, 8, 10000, True, 0, False, False, True),
    (6, 150, 9, 10000, False, 3, True, False, False),
    (200, 50, 10, 4, False, 0, False, True, False),
    (30, 40, 10, 10000, True, 0, True, True, False),
    (5, 1, 3, 10000, False, 0, False, False, False),
    (5, 2, 3, 10000, False, 0, False, False, False),
    (8, 40, 13, 10000, True, 1, True, False, False),
    (2, 20000, 77, 16, False, 0, False, False, False),      # giant contigs (row f4): one wave still walks each
    (2, 6000, 77, 16, True, 0, False, False, False),
    (2, 2600, 17, 4, False, 3, True, False, False),          # K1: contigs of 3 sort chunks, shuffled + duplicate keys
    (2, 1025, 23, 1, False, 5, True, False, False),
    (1, 9000, 41, 1, False, 4, True, False, False),          # 9 chunks: cross-chunk ranking + std::sort replay
]


def _id(c):
    return "c%dx%d_s%d_k%d_%s%s%s%s%s" % (c[0], c[1], c[2], c[3], "D" if c[4] else "S", f"_dup{c[5]}" if c[5] else "",
                                          "_shuf" if c[6] else "", "_ht" if c[7] else "", "_nsl" if c[8] else "")


@pytest.mark.parametrize("case", CASES, ids=_id)
def test_hip_outputs_match_oracle(T, case):
    nc, nr, seed, K, dense, dup, shuf, heavy, nsl = case
    api = T.api()
    hb = T.synth(nc, nr, seed, dense=dense, dup_every=dup, shuffle=shuf, heavy_tail=heavy)
    want = T.oracle_solve(hb, K, nsl)
    got = api.solve_batch(hb, max_paths=K, non_skip_linkable=nsl)
    assert T.diff_outputs(want, got) == []


@pytest.mark.parametrize("case", [CASES[0], CASES[3], CASES[5], CASES[8], CASES[13]], ids=_id)
def test_hip_intermediates_match_oracle(T, case):
    nc, nr, seed, K, dense, dup, shuf, heavy, nsl = case
    api = T.api()
    hb = T.synth(nc, nr, seed, dense=dense, dup_every=dup, shuffle=shuf, heavy_tail=heavy)
    db = api.DeviceBatch(hb)
    res = db.solve(max_paths=K, non_skip_linkable=nsl, keep_debug=True)
    assert T.diff_intermediates(hb, res.debug, K, nsl) == []
    res.close(); db.close()


# ---- rows + reversed CSR + sweep headers of a contig by one workgroup (aasm_k46_graph): the default for sparse batches whose contigs all have
# at most 1 792 vertices and 4 096 edges (a second form: 3 584 / 8 192) - most sparse cases above; here against the separate launches (row_fill, scan, rev_fill, rev_place, rev_hdr)
@pytest.mark.parametrize("case", [CASES[0], CASES[1], CASES[2], CASES[5], CASES[6], CASES[8], CASES[9], CASES[12], (5, 1250, 3, 4, False, 0, False, False, False), (3, 1200, 8, 4, False, 7, True, False, True),
                                  (3, 2000, 5, 4, False, 0, False, False, False), (2, 2450, 7, 1, False, 0, True, False, True),
                                  (12, 700, 3, 4, False, 0, False, True, False), (300, 100, 9, 4, False, 0, False, True, False)], ids=_id)   # (3 584 vertices / 8 192 edges: the form with more LDS; heavy-tailed: contigs of both forms and of the separate launches in one batch)
def test_graph_build_forms(T, case):
    """Outputs and intermediates against the oracle in both forms, and every array the later kernels read byte for byte between them
    (in-lists in list order, the two header records, in-list starts, the forward headers, the pending counts)."""
    nc, nr, seed, K, dense, dup, shuf, heavy, nsl = case
    api = T.api()
    hb = T.synth(nc, nr, seed, dense=dense, dup_every=dup, shuffle=shuf, heavy_tail=heavy)
    want = T.oracle_solve(hb, K, nsl)
    names = [("csr_col", np.int32), ("csr_w_qry", np.int64), ("csr_w_ref", np.int32), ("csr_w_flags", np.uint8), ("rptr", np.int64), ("r_pk", np.int32),
             ("rvh", np.int32), ("fvh", np.int32), ("cnt_tmp2", np.int32)]
    db = api.DeviceBatch(hb)
    kept = {}
    for form in (False, True):
        res = db.solve(max_paths=K, non_skip_linkable=nsl, keep_debug=True, graph_launches=form)
        assert T.diff_outputs(want, res.fetch()) == []
        assert T.diff_intermediates(hb, res.debug, K, nsl) == []
        kept[form] = {n: res.debug(n, dt).copy() for n, dt in names}
        st = res.stats()
        kept[form]["sizes"] = (st["n_edges"], st["n_vertices"])
        n_s, n_l, n_rest = (int(x) for x in res.debug("counters", np.int64)[18:21])   # contigs by form: one workgroup (small / large), separate launches
        assert (n_s + n_l == 0) if form else (dense or n_s + n_l > 0)
        if (nc, nr) == (12, 700) and not form:
            assert n_s > 0 and n_l > 0 and n_rest > 0
        res.close()
    db.close()
    ET, VT = kept[True]["sizes"]
    for n, _ in names:
        a, b = kept[False][n], kept[True][n]
        m = {"rptr": VT + 1, "r_pk": 4 * ET, "rvh": 12 * VT, "fvh": 8 * VT, "cnt_tmp2": VT}.get(n, ET)   # (allocations are padded: the cells the kernels own)
        assert np.array_equal(a[:m], b[:m]), n
    # ... and the 16-hop jump records of K9's recovery, which the small contigs get from a workgroup with the contig's tree in LDS
    # (aasm_k9_tnx16_wg) - with the chain class off, or batches this small never reach it
    db = api.DeviceBatch(hb)
    kept7 = {}
    for form in (False, True):
        res = db.solve(max_paths=K, non_skip_linkable=nsl, keep_debug=True, graph_launches=form, chain="none")
        assert T.diff_outputs(want, res.fetch()) == []
        kept7[form] = res.debug("tnx16", np.int32).copy()
        res.close()
    db.close()
    assert np.array_equal(kept7[False][:16 * VT], kept7[True][:16 * VT])


# ---- the chain class (aasm_k67_chain: sweep, pre-pass and heaps of a contig beside each other).  By default it takes every sparse
# contig of a batch of <= 1 536 contigs - i.e. every sparse case above - and the long tail of bigger ones; here the other forms
@pytest.mark.parametrize("chain", ["none", "half", "all"])
@pytest.mark.parametrize("case", [CASES[0], CASES[2], CASES[5], CASES[6], CASES[8], CASES[9], CASES[11], CASES[14], CASES[16]], ids=_id)
def test_chain_class_forms_match_oracle(T, case, chain):
    """none: the three launches (K6 sweep, K7 pre-pass, K7 heaps) for every contig; half: the contigs of at least the mean size in
    the class, the others in the three launches, both at the same time; all: the class for every sparse contig."""
    nc, nr, seed, K, dense, dup, shuf, heavy, nsl = case
    api = T.api()
    hb = T.synth(nc, nr, seed, dense=dense, dup_every=dup, shuffle=shuf, heavy_tail=heavy)
    want = T.oracle_solve(hb, K, nsl)
    got = api.solve_batch(hb, max_paths=K, non_skip_linkable=nsl, chain=chain)
    assert T.diff_outputs(want, got) == []
    got = api.solve_batch(hb, max_paths=K, non_skip_linkable=nsl, chain=chain, test_small_root_ring=True)   # (a 4-entry ring of roots in the class's heap wave: parents beyond it come from h_root, as on frontiers wider than 512)
    assert T.diff_outputs(want, got) == []


@pytest.mark.parametrize("chain", ["none", "half", "all"])
@pytest.mark.parametrize("case", [CASES[0], CASES[5], CASES[8], CASES[9]], ids=_id)
def test_chain_class_forms_leave_the_same_intermediates(T, case, chain):
    """... and every intermediate (both sweeps' orders, d / best, every heap node and root in the reference's allocation order,
    the K distances) equals the oracle's whichever form built it."""
    nc, nr, seed, K, dense, dup, shuf, heavy, nsl = case
    api = T.api()
    hb = T.synth(nc, nr, seed, dense=dense, dup_every=dup, shuffle=shuf, heavy_tail=heavy)
    db = api.DeviceBatch(hb)
    for own_queue in (True, False):                                 # the heap wave with its own BFS queue / the order from a wave of its own (default up to 896 contigs)
        res = db.solve(max_paths=K, non_skip_linkable=nsl, keep_debug=True, chain=chain, chain_own_queue=own_queue)
        assert T.diff_intermediates(hb, res.debug, K, nsl) == []
        res.close()
    db.close()


def test_chain_class_beyond_one_residency_round_and_as_the_long_tail(T):
    """More contigs of the class than the chip holds at once (a second round of workgroups starts as the first ones end), and
    the default rule on a big batch: only the contigs of >= 4x the mean and >= 2 048 records form the class."""
    api = T.api()
    hb = T.synth(3000, 30, 5, dup_every=6)
    want = T.oracle_solve(hb, 4)
    assert T.diff_outputs(want, api.solve_batch(hb, max_paths=4, chain="all")) == []
    hb = T.synth(1700, 300, 9, heavy_tail=True)                      # 510 k records in more contigs than the every-contig rule takes (1 536), the longest ~2 400: the default rule picks a handful
    sizes = np.diff(hb.arrays["ctg_rec_off"])
    assert 0 < (sizes >= max(2048, 4 * 300)).sum() < 40
    want = T.oracle_solve(hb, 4)
    db = api.DeviceBatch(hb)
    res = db.solve(max_paths=4, keep_debug=True)
    n_class = int(res.debug("counters", np.int64)[17])
    got = res.fetch(); got["stats"] = res.stats()
    res.close(); db.close()
    assert n_class == int((sizes >= max(2048, 4 * 300)).sum())
    assert T.diff_outputs(want, got) == []


@pytest.mark.parametrize("own_queue", [False, True], ids=["order_wave", "own_queue"])
@pytest.mark.parametrize("how", [1, 2], ids=["header_never_comes", "prep_wave_never_reports"])
def test_chain_class_waits_end_in_an_error_not_a_hang(T, how, own_queue):
    """Every wait of aasm_k67_chain has an exit every wave reaches.  Test hooks (opts.reserved[2] bits 3 / 4): the pre-pass wave of
    contig 0 never publishes the root's header - the heap wave learns from `prep_done` that it will not come; or it does not even
    report that it is done - the heap wave's own patience (1 s under the hook, 30 s otherwise) ends the wait.  Either way contig 0
    ends with AASM_E_INTERNAL, every other contig with the oracle's result, the launch drains, and the next solve is clean."""
    import time
    api = T.api()
    hb = T.synth(12, 90, 7, dup_every=5)
    want = T.oracle_solve(hb, 16)
    t0 = time.time()
    got = api.solve_batch(hb, max_paths=16, chain="all", test_chain_lost=how, chain_own_queue=own_queue)   # (the order wave waits for the header and gives up the same way; the heap wave learns it from `ord_done`)
    assert time.time() - t0 < 20
    assert got["status"][0] == -6 and (got["status"][1:] == 0).all()
    mo, ao = want["main_off"], want["alt_off"]
    gmo, gao = got["main_off"], got["alt_off"]
    assert np.array_equal(want["main"][mo[1]:], got["main"][gmo[1]:]) and np.array_equal(want["alt"][ao[1]:], got["alt"][gao[1]:])
    assert T.diff_outputs(want, api.solve_batch(hb, max_paths=16, chain="all")) == []


def test_all_pool_overflow_reruns_the_pick(T):
    """Every record duplicated: hundreds of co-optimal walks per contig, `.all` lists 130x the records - beyond the pool of the first
    pick (R + 1 024 elements): the pick is re-run with the exact size, and the output totals are read again after it."""
    api = T.api()
    hb = T.synth(3, 30, 5, dup_every=1)
    want = T.oracle_solve(hb, 10000)
    assert len(want["all"]) > len(hb.arrays["qry_str"]) + 1024
    assert T.diff_outputs(want, api.solve_batch(hb, max_paths=10000)) == []


def test_repeat_solve_is_deterministic(T):
    api = T.api()
    hb = T.synth(50, 200, 77, dup_every=5, shuffle=True)
    db = api.DeviceBatch(hb)
    outs = []
    for _ in range(3):
        r = db.solve(max_paths=64)
        outs.append(r.fetch()); r.close()
    db.close()
    assert T.diff_outputs(outs[0], outs[1]) == [] and T.diff_outputs(outs[0], outs[2]) == []


def test_out_of_memory_ranges_are_split_and_concatenated(T):
    """A contig range that does not fit is halved recursively (contigs are independent); the
    concatenated result must equal the unsplit one.  opts.reserved[1] simulates the
    out-of-memory hipMalloc."""
    api = T.api()
    hb = T.synth(23, 70, 41, heavy_tail=True, dup_every=6)
    whole = api.solve_batch(hb, max_paths=32)
    n0 = api.debug_counter("range_splits")
    split = api.solve_batch(hb, max_paths=32, test_max_contigs=4)
    assert api.debug_counter("range_splits") - n0 >= 5
    assert T.diff_outputs(whole, split, stats=False) == []
    for k in ("n_vertices", "n_edges", "n_heap_nodes", "n_paths_found", "n_pairs"):
        assert whole["stats"][k] == split["stats"][k], k


def test_hip_failure_is_reported_not_retried_as_out_of_memory(T):
    """A HIP error that is not an allocation failure (here: an injected invalid launch) must come back
    as AASM_E_HIP with the runtime's own message, and must not trigger the out-of-memory range split."""
    api = T.api()
    hb = T.synth(16, 60, 3)
    n0 = api.debug_counter("range_splits")
    with pytest.raises(api.AlignasmError) as e:
        api.solve_batch(hb, max_paths=8, test_inject_launch_failure=True)
    assert e.value.code == -3 and "kernel launch" in str(e.value) and "memory" not in str(e.value).lower(), str(e.value)
    assert api.debug_counter("range_splits") == n0
    want = T.oracle_solve(hb, 8)                                   # and the device context is still usable afterwards
    assert T.diff_outputs(want, api.solve_batch(hb, max_paths=8)) == []


def test_stalled_scan_look_back_ends_in_an_error_not_a_hang(T):
    """aasm_scan_chain's look-back waits for predecessors that took their ticket earlier; a ticket counter left behind by an
    aborted launch (test hook: opts.reserved[2] bit 2 dirties it ahead of the first scan) means a predecessor that never
    publishes.  The lanes give up after SCAN_STALL_S seconds, raise a host-visible flag, the launch drains, the solve
    returns AASM_E_HIP before anything is sized by the scan - and the context is clean for the next solve."""
    import time
    api = T.api()
    hb = T.synth(60, 100, 3)                                       # 6 000 records: the first scan has two tiles
    t0 = time.time()
    with pytest.raises(api.AlignasmError) as e:
        api.solve_batch(hb, max_paths=8, test_dirty_scan=True)
    assert e.value.code == -3 and "look-back stalled" in str(e.value), str(e.value)
    assert time.time() - t0 < 60
    want = T.oracle_solve(hb, 8)
    assert T.diff_outputs(want, api.solve_batch(hb, max_paths=8)) == []


def test_cold_arena_takes_few_device_allocations(T):
    """The workspace arena grows by doubling: a cold solve of a batch that needs GBs takes a handful of
    hipMalloc calls, a warm one none."""
    api = T.api()
    hb = T.synth(300, 400, 13)
    db = api.DeviceBatch(hb)
    db.solve(max_paths=4).close()
    n1 = api.debug_counter("device_mallocs")
    assert n1 <= 12
    db.solve(max_paths=4).close()
    assert api.debug_counter("device_mallocs") == n1
    db.close()


@pytest.mark.parametrize("case", [CASES[0], CASES[5], CASES[8], CASES[13]], ids=_id)
def test_sequential_select_fallback_on_gpu(T, case):
    nc, nr, seed, K, dense, dup, shuf, heavy, nsl = case
    api = T.api()
    hb = T.synth(nc, nr, seed, dense=dense, dup_every=dup, shuffle=shuf, heavy_tail=heavy)
    want = T.oracle_solve(hb, K, nsl)
    got = api.solve_batch(hb, max_paths=K, non_skip_linkable=nsl, sequential_select=True)
    assert T.diff_outputs(want, got) == []


# ---- K0: match ranges parsed from the cs tags on the device (one thread per record)
def test_device_cs_ranges(T):
    from test_cs_device import TAGS, _consumed, _row
    api = T.api()
    rows = []
    for cs in TAGS:
        q, r = _consumed(cs)
        for fwd in (True, False):
            rows.append(_row(cs, fwd, qs=1000 * (len(rows) + 1), ql=q, rl=r))
    text = b"".join(rows) + api.Paf.synth(8, 150, 3, dup_every=5).to_text()
    io = T.io_oracle()                                           # oracle-side get_overlap_range (paf_data.cpp:90-123), not the product's host codec
    host = {k: np.asarray(v, np.int64) for k, v in io.to_arrays(io.read_paf(text)).items()}
    dev = api.Paf.parse(text, device_ranges=True)
    db = api.DeviceBatch(dev)
    res = db.solve(max_paths=4, keep_debug=True)
    n = int(host["rec_rng_off"][-1])
    for name, key in (("rql_w", "rng_qry_l"), ("rqr_w", "rng_qry_r"), ("rrl_w", "rng_ref_l")):
        assert np.array_equal(T.k0_ranges(res.debug)[name][:n], host[key]), name
    res.close(); db.close()


def test_device_cs_solve_equals_host_range_solve_and_oracle(T):
    api = T.api()
    text = api.Paf.synth(40, 200, 9, dup_every=4, shuffle=True).to_text()
    host_paf, dev_paf = api.Paf.parse(text), api.Paf.parse(text, device_ranges=True)
    want = T.oracle_solve(host_paf.batch(), 16)
    a = api.solve_batch(host_paf, max_paths=16)
    b = api.solve_batch(dev_paf, max_paths=16)
    assert T.diff_outputs(want, a) == [] and T.diff_outputs(want, b) == []
    c = api.solve_batch(dev_paf, max_paths=16, test_max_contigs=7)   # ranges split on the host: cs text re-based per range
    assert T.diff_outputs(b, c, stats=False) == []


def test_device_cs_reports_malformed_tags(T):
    from test_cs_device import BAD, _consumed, _row
    api, io = T.api(), T.io_oracle()
    good = b":10*ac:5+gg:3-t:2"
    q, r = _consumed(good)
    for k, bad in enumerate(BAD):
        rows = [_row(good, True, qs=1000 * (i + 1), ql=q, rl=r) for i in range(70)]
        rows[41] = _row(bad, k % 2 == 0, qs=42000, ql=q, rl=r)
        with pytest.raises(io.CsError) as want:                   # what the reference's codec throws for this file
            io.read_paf(b"".join(rows))
        with pytest.raises(api.AlignasmError) as e:
            api.solve_batch(api.Paf.parse(b"".join(rows), device_ranges=True), max_paths=4)
        assert e.value.code == -7 and "(record 41)" in str(e.value) and str(want.value) in str(e.value), (k, str(e.value), str(want.value))


# ---- row T1: the device's PafDistance predicates against the REAL header's results (ref_algos.npz)
def test_device_pafdistance_predicates_match_reference_truth_tables(T):
    import ctypes as C
    import os
    api = T.api()
    z = np.load(os.path.join(T.GOLDEN, "ref_algos.npz"))
    a, b, want = np.ascontiguousarray(z["t1_a"]), np.ascontiguousarray(z["t1_b"]), z["t1_res"]
    got = np.zeros(len(a), np.uint8)
    rc = api.LIB.aasm_debug_predicates(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), C.c_int64(len(a)), got.ctypes.data_as(C.c_void_p), 0)
    assert rc == 0, api.LIB.aasm_last_error()
    assert np.array_equal(got & 7, want)                      # dist_lt<CALC_SUM>, dist_lt<QRY_SCORE>, dist_eq: paf_data.hpp:142-168
    assert np.array_equal((got >> 3) & 1, want & 1)           # K7's node-key test is operator< in CALC_SUM mode, max() included
    real = ((a >= 0).all(1) & (b >= 0).all(1))                # K8's queue order is defined on distances of real walks (components >= 0)
    assert real.sum() > 2000
    assert np.array_equal((got[real] >> 4) & 1, want[real] & 1)


def test_input_guards_report_overflow(T):
    """Coordinates outside [0, 2^40) / oversized contigs are refused with AASM_E_OVERFLOW and a message
    instead of being squeezed into the device's int32 fields; on a batch that is already resident the
    device flags the contig (ctg_status -5) and solves the others."""
    api = T.api()
    hb = T.synth(6, 50, 3)
    bad = {k: v.copy() for k, v in hb.arrays.items()}
    r = int(bad["ctg_rec_off"][2]) + 5
    bad["qry_total"][r] = 1 << 41
    from alignasm_amd._abi import HostBatch
    hb2 = HostBatch(bad)
    with pytest.raises(api.AlignasmError) as e:
        api.solve_batch(hb2, max_paths=4)
    assert e.value.code == -5 and "record %d" % r in str(e.value) and "2^40" in str(e.value)
    db = api.DeviceBatch(hb2)                                  # resident batch: no host-side scan
    res = db.solve(max_paths=4)
    out = res.fetch(); res.close(); db.close()
    assert list(out["status"]) == [0, 0, -5, 0, 0, 0] and out["stats"]["n_internal_errors"] == 1
    want = T.oracle_solve(hb, 4)
    mo = out["main_off"]
    assert mo[3] == mo[2]                                     # nothing emitted for the refused contig
    for c in (0, 1, 3, 4, 5):
        wm = want["main"][want["main_off"][c]:want["main_off"][c + 1]]
        assert np.array_equal(wm, out["main"][mo[c]:mo[c + 1]]), c


# ---- K7 with several waves per contig (kb_heap_mw), forced on every contig
@pytest.mark.parametrize("case", [CASES[0], CASES[3], CASES[4], CASES[5], CASES[10], CASES[13], CASES[15]], ids=_id)
def test_multiwave_heaps_match_oracle(T, case):
    nc, nr, seed, K, dense, dup, shuf, heavy, nsl = case
    api = T.api()
    hb = T.synth(nc, nr, seed, dense=dense, dup_every=dup, shuffle=shuf, heavy_tail=heavy)
    want = T.oracle_solve(hb, K, nsl)
    got = api.solve_batch(hb, max_paths=K, non_skip_linkable=nsl, heap_waves="all")
    assert T.diff_outputs(want, got) == []


@pytest.mark.parametrize("case", [CASES[3], CASES[5], CASES[13]], ids=_id)
def test_multiwave_heap_arena_is_bit_identical(T, case):
    """The compaction restores the reference's allocation order: arena indices, child pointers, roots and the k
    distances equal the oracle's exactly (and so the one-wave kernel's)."""
    nc, nr, seed, K, dense, dup, shuf, heavy, nsl = case
    api = T.api()
    hb = T.synth(nc, nr, seed, dense=dense, dup_every=dup, shuffle=shuf, heavy_tail=heavy)
    db = api.DeviceBatch(hb)
    res = db.solve(max_paths=K, non_skip_linkable=nsl, keep_debug=True, heap_waves="all")
    assert T.diff_intermediates(hb, res.debug, K, nsl) == []
    res.close(); db.close()


@pytest.mark.parametrize("nc", [500, 900], ids=["8waves", "4waves"])
def test_multiwave_heaps_with_fewer_waves_per_contig(T, nc):
    """The launch gives a contig 16, 8 or 4 waves depending on how many contigs share the chip: the many-contig
    variants (the small cases above all get 16)."""
    api = T.api()
    hb = T.synth(nc, 40, 77, dense=True, dup_every=4)
    want = T.oracle_solve(hb, 8)
    got = api.solve_batch(hb, max_paths=8, heap_waves="all")
    assert T.diff_outputs(want, got) == []


def test_multiwave_heaps_launch_order_does_not_matter(T):
    """By default the contigs of the several-waves class are launched largest node bound first, a block per contig of the
    class (kb_mw_rank); input order with a block per contig of the batch is the other form.  A mixed dense / sparse batch of
    contigs of very different sizes: both equal the oracle."""
    api = T.api()
    hb = T.synth(40, 300, 4242, dense=True, heavy_tail=True, dup_every=5)
    want = T.oracle_solve(hb, 16)
    for input_order in (False, True):
        got = api.solve_batch(hb, max_paths=16, heap_input_order=input_order)
        assert T.diff_outputs(want, got) == [], input_order
        got = api.solve_batch(hb, max_paths=16, heap_waves="all", heap_input_order=input_order)
        assert T.diff_outputs(want, got) == [], input_order


# ---- K7, one wave per contig (kb_heap) forced on every contig, the dense ones included (by default those go to
# kb_heap_mw): both kernels must leave the same arena
@pytest.mark.parametrize("case", [CASES[2], CASES[5], CASES[9], CASES[13]], ids=_id)
def test_one_wave_heap_kernel_arena_is_bit_identical(T, case):
    nc, nr, seed, K, dense, dup, shuf, heavy, nsl = case
    api = T.api()
    hb = T.synth(nc, nr, seed, dense=dense, dup_every=dup, shuffle=shuf, heavy_tail=heavy)
    db = api.DeviceBatch(hb)
    res = db.solve(max_paths=K, non_skip_linkable=nsl, keep_debug=True, heap_waves="none")
    assert T.diff_intermediates(hb, res.debug, K, nsl) == []
    res.close(); db.close()


def test_reserved_workspace_serves_the_next_solve(T):
    """aasm_reserve_workspace (what the CLI runs beside its reader): after it, a solve that fits the reservation
    makes no device allocation of its own."""
    api = T.api()
    assert api.reserve_workspace(0, 3 << 30) == 0
    hb = T.synth(300, 400, 19)
    db = api.DeviceBatch(hb)
    n0 = api.debug_counter("device_mallocs")
    want = T.oracle_solve(hb, 4)
    res = db.solve(max_paths=4)
    assert api.debug_counter("device_mallocs") == n0
    assert T.diff_outputs(want, res.fetch()) == []
    res.close(); db.close()
    assert api.reserve_workspace(99, 1 << 20) != 0          # no such device: an error code, not a crash


#          contigs, records, seed, K, dense, dup_every, shuffle
WIDE = [(4, 600, 31, 16, True, 0, False),      # C5 shape: most window DPs too dense for the 63-position LDS copy (streamed rows)
        (2, 1500, 7, 16, True, 0, False),      # wider windows: 64 ... 127 positions on the LDS overlay, beyond that global state
        (3, 300, 8, 10000, True, 0, False)]


@pytest.mark.parametrize("case", WIDE, ids=lambda c: "c%dx%d_s%d_k%d_%s%s" % (c[0], c[1], c[2], c[3], "D" if c[4] else "S", "_dup%d" % c[5] if c[5] else ""))
def test_wide_and_dense_window_dps_match_oracle(T, case):
    """K9's three window-DP forms beside the LDS copy (rows streamed over LDS state for <= 63 and for <= 127 positions, global
    state beyond) on dense graphs, where they carry most of the work: outputs and intermediates against the oracle."""
    nc, nr, seed, K, dense, dup, shuf = case
    api = T.api()
    hb = T.synth(nc, nr, seed, dense=dense, dup_every=dup, shuffle=shuf)
    want = T.oracle_solve(hb, K)
    got = api.solve_batch(hb, max_paths=K)
    assert T.diff_outputs(want, got) == []
