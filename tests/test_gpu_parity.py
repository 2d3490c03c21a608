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


def test_out_of_memory_ranges_are_split_and_concatenated(T, monkeypatch):
    """A contig range that does not fit is halved recursively (contigs are independent); the
    concatenated result must equal the unsplit one.  AASM_TEST_MAX_CONTIGS simulates the
    hipMalloc failure."""
    api = T.api()
    hb = T.synth(23, 70, 41, heavy_tail=True, dup_every=6)
    whole = api.solve_batch(hb, max_paths=32)
    monkeypatch.setenv("AASM_TEST_MAX_CONTIGS", "4")
    split = api.solve_batch(hb, max_paths=32)
    monkeypatch.delenv("AASM_TEST_MAX_CONTIGS")
    assert T.diff_outputs(whole, split, stats=False) == []
    for k in ("n_vertices", "n_edges", "n_heap_nodes", "n_paths_found", "n_pairs"):
        assert whole["stats"][k] == split["stats"][k], k


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
        assert np.array_equal(res.debug(name, np.int64)[:n], host[key]), name
    res.close(); db.close()


def test_device_cs_solve_equals_host_range_solve_and_oracle(T, monkeypatch):
    api = T.api()
    text = api.Paf.synth(40, 200, 9, dup_every=4, shuffle=True).to_text()
    host_paf, dev_paf = api.Paf.parse(text), api.Paf.parse(text, device_ranges=True)
    want = T.oracle_solve(host_paf.batch(), 16)
    a = api.solve_batch(host_paf, max_paths=16)
    b = api.solve_batch(dev_paf, max_paths=16)
    assert T.diff_outputs(want, a) == [] and T.diff_outputs(want, b) == []
    monkeypatch.setenv("AASM_TEST_MAX_CONTIGS", "7")               # ranges split on the host: cs text re-based per range
    c = api.solve_batch(dev_paf, max_paths=16)
    monkeypatch.delenv("AASM_TEST_MAX_CONTIGS")
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
