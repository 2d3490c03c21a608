"""K9 cannot be pinned to the reference in this image (paf_data.cpp:739+ needs ankerl/unordered_dense.h), so it gets a THIRD
reading: tests/k9_checker.py (plain Python, own shape, no oracle include) re-derives every upgraded path from the walk the heaps
gave (pinned) and the graph (pinned), checks the reference's own Debug asserts on it (:913-918), and - for every window DP of at
most 12 vertices - enumerates ALL a -> b paths and asserts that none is smaller under the QRY_SCORE_MODE order than the one taken.
CPU tier: the product's kernel bodies in the 1-lane emulation; GPU tier: the HIP path (tests below marked gpu), incl. the chain
asserts on every conversion of the C3 batch and the C5 share.  A third reading is not a pin: DESIGN.md section 2 says so."""
import numpy as np
import pytest

import k9_checker as K

CASES = [
    # contigs, records, seed, K, dense, dup_every, shuffle, heavy_tail
    (6, 120, 42, 16, False, 7, True, False),
    (4, 250, 3, 10000, False, 0, False, False),
    (5, 90, 8, 64, False, 3, True, True),
    (2, 160, 31, 16, True, 0, False, False),
    (3, 60, 31, 10000, True, 4, True, False),
]


def _check_batch(T, hb, fetch, nc, brute_max=12):
    a = K.collect(fetch, hb.arrays["ctg_rec_off"])
    stats, bad, nconv = {}, [], 0
    for c in range(nc):
        if int(a["ctgV"][c]) == 0:
            continue
        g = K.graph_of(a, c)
        for j, (pa, pb) in enumerate(K.conversions_of(a, c)):
            nconv += 1
            bad += ["contig %d conversion %d: %s" % (c, j, x) for x in K.check_conversion(g, pa, pb, brute_max=brute_max, stats=stats)]
    return nconv, stats, bad


@pytest.mark.parametrize("case", CASES, ids=lambda c: "c%dx%d_s%d_k%d_%s" % (c[0], c[1], c[2], c[3], "D" if c[4] else "S"))
def test_third_reading_agrees_with_the_emulated_kernel_bodies(T, case):
    nc, nr, seed, Kp, dense, dup, shuf, heavy = case
    hb = T.synth(nc, nr, seed, dense=dense, dup_every=dup, shuffle=shuf, heavy_tail=heavy)
    T.emul_solve(hb, Kp)
    nconv, stats, bad = _check_batch(T, hb, T.emul_debug, nc)
    assert bad == [], bad[:5]
    assert nconv >= nc and stats.get("dp", 0) > 0 and stats.get("brute", 0) > 0, (nconv, stats)
    if dense:
        assert stats["brute_paths"] > stats["brute"], stats           # ... and some windows really have alternatives to rule out


def test_the_checker_notices_a_worse_path(T):
    """The checker itself: replace one window's path by a valid but longer detour (or cut the chain) and it must say so."""
    hb = T.synth(2, 160, 31, dense=True)
    T.emul_solve(hb, 16)
    a = K.collect(T.emul_debug, hb.arrays["ctg_rec_off"])
    g = K.graph_of(a, 0)
    pa, pb = K.conversions_of(a, 0)[0]
    assert K.check_conversion(g, pa, pb) == []
    assert K.check_conversion(g, pa, pb[:-1]) != []                   # does not reach dest
    assert K.check_conversion(g, pa, pb[:3] + pb[4:]) != []           # not chained
    found = False
    for t in range(1, len(pb) - 2):                                   # a shortcut u -> v for two edges u -> x -> v of the path: a valid walk that skips a record
        (u, x), (x2, v) = pb[t], pb[t + 1]
        if g.has_edge(u, v):
            worse = pb[:t] + [(u, v)] + pb[t + 2:]
            assert K.check_conversion(g, pa, worse) != []
            found = True
            break
    assert found


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES + [(40, 300, 77, 4, False, 0, False, True), (3, 700, 5, 16, True, 0, False, False)],
                         ids=lambda c: "c%dx%d_s%d_k%d_%s" % (c[0], c[1], c[2], c[3], "D" if c[4] else "S"))
def test_third_reading_agrees_with_the_hip_path(T, case):
    nc, nr, seed, Kp, dense, dup, shuf, heavy = case
    api = T.api()
    hb = T.synth(nc, nr, seed, dense=dense, dup_every=dup, shuffle=shuf, heavy_tail=heavy)
    db = api.DeviceBatch(hb)
    res = db.solve(max_paths=Kp, keep_debug=True)
    nconv, stats, bad = _check_batch(T, hb, res.debug, nc)
    res.close(); db.close()
    assert bad == [], bad[:5]
    assert nconv >= 1 and stats.get("brute", 0) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["c3", "c5_share"])
def test_chain_asserts_hold_on_every_conversion_at_full_size(T, shape):
    """paf_data.cpp:913-918 on every converted path of the C3 batch (5 000 x 1 000, K = 4) and of the C5 per-GPU share
    (1 250 dense x 1 000, K = 16): src first, dest last, consecutive edges chained, every edge an edge of the graph."""
    api = T.api()
    if shape == "c3":
        paf, Kp = api.Paf.synth(5000, 1000, 21, no_cs=True), 4
    else:
        paf, Kp = api.Paf.synth(1250, 1000, 31, dense=True, no_cs=True), 16
    db = api.DeviceBatch(paf)
    res = db.solve(max_paths=Kp, keep_debug=True)
    rec_off = paf.batch().arrays["ctg_rec_off"]
    a = K.collect(res.debug, rec_off)
    n, nedges, bad = K.chain_invariants_batch(a["voff"], a["ctgV"], a["rowptr"], a["col"], a["cv_ctg"], a["cv_roff"], a["cv_la"], a["cv_path"], rec_off)
    st = res.stats()
    res.close(); db.close(); paf.close()
    assert bad == [], bad[:5]
    assert n == st["n_paths_converted"] and nedges > 100 * n
