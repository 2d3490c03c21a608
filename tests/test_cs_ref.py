"""cs codec pinned to the REFERENCE'S OWN CODE (get_overlap_range / get_edited_paf_data, paf_data.cpp:15-220).

  ref tier  (needs oracle/_ref/libaasm_ref_cs.so, built from /root/reference where it exists): the product's host codec
            (aasm_cs_match_ranges / aasm_cs_edit) AND the I/O oracle (oracle/paf_io_oracle.py) against the real functions,
            live, on > 5 000 tags x strands and > 15 000 clips, every exception text included.
  CPU tier  the same two against tests/golden/ref_cs.json.gz (vectors recorded from the real functions by
            tests/golden/make_ref_cs.py), and K0's kernel body in the 1-lane host emulation against those vectors.
  GPU tier  K0 (aasm_k0_cs_ranges) on the device against those vectors: ranges of every accepted row, and the reference's
            exception text for rejected rows.
"""
import ctypes as C
import gzip
import json
import os
import random

import numpy as np
import pytest

import cs_cases as G
from alignasm_amd._abi import BatchOut, Opts


@pytest.fixture(scope="module")
def golden(T):
    with gzip.open(os.path.join(T.GOLDEN, "ref_cs.json.gz"), "rb") as f:
        g = json.loads(f.read())
    assert len(g["cases"]) > 1300
    return g


def _same_ranges(want, got):
    if want[0] == "err":
        return got[0] == "err" and got[2] == want[2]
    return got == want


@pytest.mark.ref
def test_product_codec_and_io_oracle_equal_the_real_reference_codec(T):
    if T.ref_cs() is None:
        pytest.skip("oracle/_ref/libaasm_ref_cs.so not built (no /root/reference here)")
    rng = random.Random(3)
    n_rows = n_rej = n_clip = n_clip_err = 0
    texts = set()
    for row in G.rows(1, 2400, 600):
        want = T.ref_cs_ranges(row)
        n_rows += 1
        for who, got in (("product", T.product_cs_ranges(row)), ("io_oracle", T.io_cs_ranges(row))):
            assert _same_ranges(want, got), (who, row, want[:3] if want[0] == "err" else "ok", got[:3] if got[0] == "err" else got[1][:3])
        if want[0] == "err":
            assert want[1] == -1                                   # std::invalid_argument, never anything else
            n_rej += 1
            texts.add(want[2])
            continue
        for q, r_, step in ((x, y, 1 if row["fwd"] else -1) for x, y in [(want[1], None)]):
            for ql, qr, rl, rr in q:                               # the product drops ref_r: it must be derivable
                assert rr == rl + (qr - ql) * step
        for clip in G.clips(rng, row, [(a, b, c) for a, b, c, _ in want[1]], 4):
            we = T.ref_cs_edit(row, clip)
            n_clip += 1
            for who, ge in (("product", T.product_cs_edit(row, clip)), ("io_oracle", T.io_cs_edit(row, clip))):
                if we[0] == "err":
                    assert ge[0] == "err" and ge[2] == we[2], (who, row, clip, we, ge)
                else:
                    assert ge == we, (who, row, clip, we, ge)
            if we[0] == "err":
                assert we[1] == -3                                 # std::logic_error
                n_clip_err += 1
                texts.add(we[2])
    assert n_rows > 5000 and n_rej > 1500 and n_clip > 12000 and n_clip_err > 800
    assert len(texts) == 8                                         # every throw site of paf_data.cpp:29-220 was reached


def test_product_codec_and_io_oracle_equal_the_recorded_reference_vectors(T, golden):
    n_clip = 0
    for case in golden["cases"]:
        want = ("err", case["err"][0], case["err"][1]) if "err" in case else ("ok", [tuple(x) for x in case["ranges"]])
        for who, got in (("product", T.product_cs_ranges(case)), ("io_oracle", T.io_cs_ranges(case))):
            assert _same_ranges(want, got), (who, case["cs"][:80], case["fwd"])
        for c in case.get("clips", ()):
            n_clip += 1
            for who, ge in (("product", T.product_cs_edit(case, c["clip"], golden["mat_num_in"], golden["aln_len_in"])),
                            ("io_oracle", T.io_cs_edit(case, c["clip"], golden["mat_num_in"], golden["aln_len_in"]))):
                if "err" in c:
                    assert ge[0] == "err" and ge[2] == c["err"][1], (who, case["cs"][:80], c)
                else:
                    assert ge == ("ok", c["cs"], c["mat"], c["aln"], c["cut"]), (who, case["cs"][:80], c, ge)
    assert n_clip > 2000


# ---- K0: the same vectors as PAF text through the solver's own cs parser --------------------------------------------------
def _paf_line(case, name):
    """PAF text of a recorded row (inverse of the reader's coordinate handling, alignasm.cpp:137-147)."""
    lo, hi = min(case["rs"], case["re"]), max(case["rs"], case["re"])
    f = [name, "20000000000", str(case["qs"]), str(case["qe"] + 1), "+" if case["fwd"] else "-", "chr1", "300000000", str(lo), str(hi + 1),
         "7", "9", "60", "tp:A:P", case["cs"]]
    return ("\t".join(f) + "\n").encode()


def _file_level(case):
    cs = case["cs"]
    return cs.startswith("cs:Z:") and not any(ch in cs for ch in "\t\n\r ")


def _accepted_text(golden, per_contig=16):
    rows = [c for c in golden["cases"] if "err" not in c and _file_level(c)]
    text = b"".join(_paf_line(c, "ctg%d" % (i // per_contig)) for i, c in enumerate(rows))
    return rows, text


def _check_ranges(rows, off, ql, qr, rl):
    assert len(off) == len(rows) + 1
    for i, c in enumerate(rows):
        a, b = int(off[i]), int(off[i + 1])
        want = c["ranges"]
        assert b - a == len(want), (i, c["cs"][:80])
        assert [list(map(int, t)) for t in zip(ql[a:b], qr[a:b], rl[a:b])] == [w[:3] for w in want], (i, c["cs"][:80], c["fwd"])


def _rejected_files(golden, n, seed=5):
    """Small files with ONE rejected recorded row among accepted ones -> (text, index of the bad record, reference text)."""
    good = [c for c in golden["cases"] if "err" not in c and _file_level(c) and len(c["cs"]) < 200][:40]
    bad = [c for c in golden["cases"] if "err" in c and _file_level(c)]
    by_text = {}
    for c in bad:
        by_text.setdefault(c["err"][1], []).append(c)
    assert len(by_text) == 5                           # every get_overlap_range throw site but the missing-"cs:Z:" one (a reader error at file level)
    rng = random.Random(seed)
    picks = [rng.choice(v) for v in by_text.values() for _ in range(max(1, n // len(by_text)))]
    out = []
    for k, c in enumerate(picks):
        rows = [good[(k + j) % len(good)] for j in range(9)]
        at = 2 + k % 6
        rows[at] = c
        out.append((b"".join(_paf_line(r, "ctg%d" % (j // 5)) for j, r in enumerate(rows)), at, c["err"][1]))
    return out


def test_k0_emulation_equals_the_recorded_reference_vectors(T, golden):
    api = T.api()
    rows, text = _accepted_text(golden)
    dev = api.Paf.parse(text, device_ranges=True)
    view = dev.view()
    out = BatchOut()
    assert T.emul().emul_solve_batch(C.byref(view), C.byref(Opts(1, 0, 0, 0, 1)), C.byref(out)) == 0
    T.emul().emul_free_out(C.byref(out))
    off = dev.batch().arrays["rec_rng_off"]
    _check_ranges(rows, off, *(T.k0_ranges(T.emul_debug)[k] for k in ("rql_w", "rqr_w", "rrl_w")))
    for text, at, msg in _rejected_files(golden, 10):
        with pytest.raises(api.AlignasmError) as e:
            api.Paf.parse(text)                                     # the host reader: same text as the reference
        assert msg in str(e.value)
        dev = api.Paf.parse(text, device_ranges=True)
        out = BatchOut()
        assert T.emul().emul_solve_batch(C.byref(dev.view()), C.byref(Opts(1, 0, 0, 0, 1)), C.byref(out)) == -7
        assert T.emul().emul_last_bad_record() == at


@pytest.mark.gpu
def test_k0_device_equals_the_recorded_reference_vectors(T, golden):
    api = T.api()
    rows, text = _accepted_text(golden)
    assert len(rows) > 700
    dev = api.Paf.parse(text, device_ranges=True)
    db = api.DeviceBatch(dev)
    res = db.solve(max_paths=1, keep_debug=True)
    off = dev.batch().arrays["rec_rng_off"]
    _check_ranges(rows, off, *(T.k0_ranges(res.debug)[k] for k in ("rql_w", "rqr_w", "rrl_w")))
    res.close(); db.close()


@pytest.mark.gpu
def test_k0_device_rejects_with_the_reference_exception_text(T, golden):
    api = T.api()
    files = _rejected_files(golden, 40)
    assert len(files) >= 40
    for text, at, msg in files:
        with pytest.raises(api.AlignasmError) as e:
            api.solve_batch(api.Paf.parse(text, device_ranges=True), max_paths=1)
        assert e.value.code == -7 and "(record %d)" % at in str(e.value) and msg in str(e.value), (at, msg, str(e.value))
