"""Product host codec (alignasm_amd/csrc/aasm_paf.cpp: reader, cs scanner, --alt merge, writers)
against the independent Python restatement of the reference's I/O side (oracle/paf_io_oracle.py:
alignasm.cpp:76-332,398-490, paf_data.cpp:19-220)."""
import ctypes as C
import random

import numpy as np
import pytest

BATCH_KEYS = ("ctg_rec_off", "qry_str", "qry_end", "ref_str", "ref_end", "qry_total", "ref_chr", "aln_fwd", "map_qul",
              "rec_rng_off", "rng_qry_l", "rng_qry_r", "rng_ref_l")


def _same_batch(a, b):
    return [k for k in BATCH_KEYS if not np.array_equal(np.asarray(a[k], np.int64), np.asarray(b[k], np.int64))]


@pytest.mark.parametrize("case", [(6, 80, 3, 4, True), (3, 300, 9, 0, False), (40, 12, 5, 3, True), (1, 1, 2, 0, False)], ids=str)
def test_reader_matches_io_oracle(T, case):
    nc, nr, seed, dup, shuf = case
    api, io = T.api(), T.io_oracle()
    text = api.Paf.synth(nc, nr, seed, dup_every=dup, shuffle=shuf).to_text()
    want = io.to_arrays(io.read_paf(text))
    for dev in (False, True):
        paf = api.Paf.parse(text, device_ranges=dev)
        got = paf.batch().arrays if not dev else None
        if not dev:
            assert _same_batch(want, got) == []
        else:                                       # range COUNTS only (the GPU derives the ranges: test_gpu_parity)
            v = paf.view()
            from alignasm_amd._abi import _np_from
            assert np.array_equal(_np_from(v.rec_rng_off, int(v.n_records) + 1, np.int64), want["rec_rng_off"])
        paf.close()


def _random_cs(rng, n_ops):
    ops, q, r = [], 0, 0
    for i in range(n_ops):
        if i % 2 == 0:
            n = rng.randint(1, 40)
            ops.append(":%d" % n); q += n; r += n
        else:
            k = rng.choice("*+-")
            if k == "*":
                ops.append("*" + rng.choice("acgt") + rng.choice("acgt")); q += 1; r += 1
            else:
                n = rng.randint(1, 5)
                ops.append(k + "".join(rng.choice("acgtn") for _ in range(n)))
                if k == "+":
                    q += n
                else:
                    r += n
    return "cs:Z:" + "".join(ops), q, r


def _row(io, cs, fwd, qs, ql, rs, rl):
    r = io.PafRow()
    r.aln_fwd, r.qry_str, r.qry_end = fwd, qs, qs + ql - 1
    r.ref_str, r.ref_end = (rs, rs + rl - 1) if fwd else (rs + rl - 1, rs)
    r.cs_string, r.mat_num, r.aln_len = cs, 7, 9
    return r


DAMAGED = ["cs:Z::0", "cs:Z::-5", "cs:Z::", "cs:Z::12x", "cs:Z:*a", "cs:Z:*a1", "cs:Z:+", "cs:Z:-:4", "cs:Z:=ACGT", "cs:Z::5~gt12ag",
           "cs:Y::5", "cs:Z", "cs:Z::99999999999999999999", "cs:Z::00000000000000000000005", "cs:Z::9223372036854775807", "", "cs:Z:", "cs:Z::3+ac*t"]


def test_match_ranges_and_errors_match_io_oracle(T):
    api, io = T.api(), T.io_oracle()
    rng = random.Random(11)
    cap = 256
    ql, qr, rl = (np.zeros(cap, np.int64) for _ in range(3))
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    tags = [_random_cs(rng, rng.randint(1, 60)) for _ in range(300)]
    tags += [(t, 5, 5) for t in DAMAGED]
    n_bad = 0
    for cs, q, r in tags:
        for fwd in (True, False):
            for dq, dr in ((0, 0), (1, 0), (0, -1)):          # exact, or coordinates the tag does not consume
                row = _row(io, cs, fwd, 1000, max(q + dq, 1), 50000, max(r + dr, 1))
                b = cs.encode()
                n = api.LIB.aasm_cs_match_ranges(b, C.c_int64(len(b)), 1 if fwd else 0, C.c_int64(row.qry_str), C.c_int64(row.qry_end),
                                                 C.c_int64(row.ref_str), C.c_int64(row.ref_end), P(ql), P(qr), P(rl), C.c_int64(cap))
                try:
                    io.get_overlap_range(row, cs)
                except io.CsError as e:
                    assert n == -7 and api.LIB.aasm_last_error().decode() == str(e), (cs, fwd, dq, dr)
                    n_bad += 1
                    continue
                assert n == len(row.qry_rng), (cs, fwd)
                assert [(int(a), int(b_)) for a, b_ in zip(ql[:n], qr[:n])] == row.qry_rng and [int(x) for x in rl[:n]] == [x[0] for x in row.ref_rng], (cs, fwd)
    assert n_bad > 100


def test_cs_edit_matches_io_oracle_on_random_clips(T):
    """get_edited_paf_data (paf_data.cpp:125-220): a few thousand clips, both strands, incl. clips
    inside insertions (logic_error), clips whose reference span is inconsistent, and uncut records."""
    api, io = T.api(), T.io_oracle()
    rng = random.Random(5)
    out = C.create_string_buffer(1 << 14)
    n_ok = n_err = 0
    for _ in range(700):
        cs, q, r = _random_cs(rng, rng.randint(1, 40))
        for fwd in (True, False):
            row = _row(io, cs, fwd, 2000, q, 70000, r)
            io.get_overlap_range(row, cs)
            for _clip in range(4):
                a = rng.randint(row.qry_str, row.qry_end)
                b_ = rng.randint(a, row.qry_end)
                if rng.random() < 0.2:
                    a = row.qry_str
                if rng.random() < 0.2:
                    b_ = row.qry_end
                # reference coordinates of the clip: exact when both ends sit in a match run, otherwise whatever (-> error path)
                def ref_of(x):
                    for (l, rr), (fl, _) in zip(row.qry_rng, row.ref_rng):
                        if l <= x <= rr:
                            return fl + (x - l) * (1 if fwd else -1)
                    return row.ref_str
                ers, ere = ref_of(a), ref_of(b_)
                m, al, cut = C.c_int32(row.mat_num), C.c_int32(row.aln_len), C.c_int32(-1)
                b = cs.encode()
                n = api.LIB.aasm_cs_edit(b, C.c_int64(len(b)), 1 if fwd else 0, C.c_int64(row.qry_str), C.c_int64(row.qry_end), C.c_int64(a), C.c_int64(b_),
                                         C.c_int64(ers), C.c_int64(ere), out, C.c_int64(len(out)), C.byref(m), C.byref(al), C.byref(cut))
                try:
                    want = io.get_edited_paf_data(a, b_, ers, ere, row)
                except (io.CsLogicError, io.CsError) as e:
                    assert n == -7 and api.LIB.aasm_last_error().decode() == str(e), (cs, fwd, a, b_)
                    n_err += 1
                    continue
                assert n >= 0, (cs, fwd, a, b_, api.LIB.aasm_last_error())
                assert (out.value.decode(), m.value, al.value, bool(cut.value)) == want, (cs, fwd, a, b_)
                n_ok += 1
    assert n_ok > 1500 and n_err > 300


def _alt_text(T, ctg_names, seed):
    from test_alt_merge import _alt_text as f
    return f(T, ctg_names, seed)


def _names(text):
    names = []
    for line in text.decode().splitlines():
        if not names or names[-1] != line.split("\t")[0]:
            names.append(line.split("\t")[0])
    return names


@pytest.mark.parametrize("baseline", [0.5, 0.05, 0.99])
def test_alt_merge_matches_io_oracle(T, baseline):
    api, io = T.api(), T.io_oracle()
    text = api.Paf.synth(3, 12, 7).to_text()
    alt = _alt_text(T, _names(text), 99)
    st = io.read_paf(text)
    io.merge_alt(st, alt, baseline)
    paf = api.Paf.parse(text)
    paf.merge_alt(alt, baseline)
    assert _same_batch(io.to_arrays(st), paf.batch().arrays) == []
    assert sum(len(c) for c in st.paf_data) > 36


@pytest.mark.parametrize("case", [(40, 150, 21, 6, 64, False, False), (12, 200, 31, 0, 16, True, False), (30, 60, 8, 2, 10000, False, True)], ids=str)
def test_writers_match_io_oracle(T, tmp_path, case):
    """process_output / process_max_output on a few thousand output rows (clipped and unclipped,
    both strands, .all ties): the product's writer against the oracle-side renderer, same solution."""
    nc, nr, seed, dup, K, dense, with_alt = case
    api, io = T.api(), T.io_oracle()
    text = api.Paf.synth(nc, nr, seed, dup_every=dup, dense=dense).to_text()
    alt = _alt_text(T, _names(text), 5) if with_alt else None
    st = io.read_paf(text)
    paf = api.Paf.parse(text)
    if alt:
        io.merge_alt(st, alt, 0.5)
        paf.merge_alt(alt, 0.5)
    hb = T.io_oracle_batch(st)
    sol = T.oracle_solve(hb, K)
    want = io.render_outputs(st, sol)
    from alignasm_amd._abi import BatchOut, Opts
    out = BatchOut()
    assert T.oracle().oracle_solve_batch(C.byref(hb.view), C.byref(Opts(K, 0, 0, 0, 0)), 2, C.byref(out)) == 0
    paths = [str(tmp_path / n) for n in ("x.aln.paf", "x.aln.alt.paf", "x.aln.all.paf")]
    paf.write_outputs(out, *paths)
    T.oracle().oracle_free_out(C.byref(out))
    got = [open(p, "rb").read() for p in paths]
    assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2]
    rows = want[0].count(b"\n") + want[1].count(b"\n") + want[2].count(b"\n")
    assert rows > 300
    if with_alt:
        assert b"xi:Z:A_" in want[0] + want[1] + want[2]
    cut = sum(1 for ln in want[0].decode().splitlines() if ln.split("\t")[14] not in text.decode())
    assert cut > 0


def test_writer_in_several_rounds_matches_io_oracle(T, tmp_path):
    """The writer formats and writes in rounds of ~4 MB per host thread (formatting of round r + 1 beside the write of
    round r): tens of MB of output over 3 threads = several rounds and shares, byte-identical to the oracle-side renderer."""
    api, io = T.api(), T.io_oracle()
    text = api.Paf.synth(260, 150, 23, dup_every=5).to_text()
    st = io.read_paf(text)
    paf = api.Paf.parse(text)
    hb = T.io_oracle_batch(st)
    K = 6
    sol = T.oracle_solve(hb, K)
    want = io.render_outputs(st, sol)
    from alignasm_amd._abi import BatchOut, Opts
    out = BatchOut()
    assert T.oracle().oracle_solve_batch(C.byref(hb.view), C.byref(Opts(K, 0, 0, 0, 0)), 2, C.byref(out)) == 0
    paths = [str(tmp_path / n) for n in ("y.aln.paf", "y.aln.alt.paf", "y.aln.all.paf")]
    prev = api.set_host_threads(3)
    try:
        paf.write_outputs(out, *paths)
    finally:
        api.set_host_threads(prev)
    T.oracle().oracle_free_out(C.byref(out))
    got = [open(p, "rb").read() for p in paths]
    assert len(want[0]) > 3 * (4 << 20)                           # more than one round of three 4 MB shares
    assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2]


def test_a_row_that_cannot_be_written_leaves_no_file_behind(T, tmp_path):
    """get_edited_paf_data throws for a clip that does not fit its cs tag (paf_data.cpp:211-218); the reference dies with the
    exception.  Here the write fails with the same text - also when the bad row sits in a LATER round of the writer than the
    ones already on disk - and neither a truncated output nor the temporary file stays."""
    api = T.api()
    text = api.Paf.synth(260, 150, 23, dup_every=5).to_text()
    paf = api.Paf.parse(text)
    from alignasm_amd._abi import BatchOut, Opts, OutElem
    out = BatchOut()
    assert T.oracle().oracle_solve_batch(C.byref(paf.view()), C.byref(Opts(6, 0, 0, 0, 0)), 2, C.byref(out)) == 0
    n_main = int(C.cast(out.main_off, C.POINTER(C.c_int64))[out.n_contigs])
    elems = C.cast(out.main_elems, C.POINTER(OutElem))
    k = n_main - 7                                                  # far behind the first rounds (3 threads x ~4 MB each)
    elems[k].edited_qry_str += 1                                    # the clip no longer matches its reference span
    paths = [str(tmp_path / n) for n in ("z.aln.paf", "z.aln.alt.paf", "z.aln.all.paf")]
    prev = api.set_host_threads(3)
    try:
        with pytest.raises(api.AlignasmError) as e:
            paf.write_outputs(out, *paths)
    finally:
        api.set_host_threads(prev)
    T.oracle().oracle_free_out(C.byref(out))
    assert "Edited cs tag does not match edited PAF coordinates" in str(e.value) or "clipped inside a cs insertion" in str(e.value)
    assert [p.name for p in tmp_path.iterdir()] == []


def test_writer_renders_unusual_run_lengths_like_the_reference(T, tmp_path):
    """A kept ':' run is copied as it stands only when it is written the way std::to_string writes it; ":007" (which
    std::from_chars accepts) has to come out as ":7" in a re-cut row and unchanged in a row that is not cut - rows of both
    kinds, run lengths padded at the front of the tag, in the middle and at its end."""
    import re
    api, io = T.api(), T.io_oracle()
    text = api.Paf.synth(30, 120, 17, dup_every=4).to_text().decode()
    rows = []
    for i, ln in enumerate(text.splitlines()):
        f = ln.split("\t")
        runs = list(re.finditer(r":(\d+)", f[-1]))
        if runs and i % 3 != 2:
            m = runs[(0, len(runs) // 2, len(runs) - 1)[i % 3]] if i % 2 else runs[0]
            f[-1] = f[-1][:m.start()] + ":00" + m.group(1) + f[-1][m.end():]
        rows.append("\t".join(f))
    text = ("\n".join(rows) + "\n").encode()
    st = io.read_paf(text)
    paf = api.Paf.parse(text)
    hb = T.io_oracle_batch(st)
    K = 8
    want = io.render_outputs(st, T.oracle_solve(hb, K))
    from alignasm_amd._abi import BatchOut, Opts
    out = BatchOut()
    assert T.oracle().oracle_solve_batch(C.byref(hb.view), C.byref(Opts(K, 0, 0, 0, 0)), 2, C.byref(out)) == 0
    paths = [str(tmp_path / n) for n in ("w.aln.paf", "w.aln.alt.paf", "w.aln.all.paf")]
    paf.write_outputs(out, *paths)
    T.oracle().oracle_free_out(C.byref(out))
    got = [open(p, "rb").read() for p in paths]
    assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2]
    tags = [ln.split(b"\t")[14] for ln in want[0].splitlines()]
    assert any(b":00" in t for t in tags) and any(b":00" not in t for t in tags)       # uncut rows keep the padding, re-cut rows lose it
