"""Host codec: PAF reader, cs:Z match ranges, cs re-cut and the 15-column writers."""
import os

import numpy as np
import pytest


def test_text_roundtrip_reproduces_the_batch(T):
    api = T.api()
    paf = api.Paf.synth(6, 80, 3, dup_every=4, shuffle=True)
    direct = paf.batch()
    text = paf.to_text()
    again = api.Paf.parse(text)
    parsed = again.batch()
    def relabel(a):                    # the reader numbers reference names by first appearance (alignasm.cpp:119-123)
        seen = {}
        return np.array([seen.setdefault(int(x), len(seen)) for x in a], np.int32)
    for k in direct.arrays:            # ranges from the generator == ranges parsed from the cs strings
        if k == "ref_chr":
            assert np.array_equal(relabel(direct.arrays[k]), parsed.arrays[k])
        else:
            assert np.array_equal(direct.arrays[k], parsed.arrays[k]), k
    assert again.to_text() == text


def test_missing_cs_tag_and_bad_rows_are_reported(T):
    api = T.api()
    row = b"ctg1\t1000\t0\t100\t+\tchr1\t5000\t10\t110\t100\t100\t60\ttp:A:P\n"
    with pytest.raises(api.AlignasmError) as e:
        api.Paf.parse(row)
    assert "Missing cs:Z tag in PAF record for query 'ctg1'" in str(e.value)
    with pytest.raises(api.AlignasmError) as e:
        api.Paf.parse(row[:-1] + b"\tcs:Z::99\n")           # consumes 99 != 100 bases
    assert "cs tag consumption does not match PAF coordinates" in str(e.value)
    with pytest.raises(api.AlignasmError):
        api.Paf.parse(row[:-1] + b"\tcs:Z::50?x:50\n")


def test_reverse_strand_ranges(T):
    import ctypes as C
    api = T.api()
    cs = b"cs:Z::10*ac:5+gg:3-t:2"      # ref-forward order; query len 10+1+5+2+3+2=23, ref len 10+1+5+3+1+2=22
    cap = 16
    ql, qr, rl = (np.zeros(cap, np.int64) for _ in range(3))
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    n = api.LIB.aasm_cs_match_ranges(cs, C.c_int64(len(cs)), 1, C.c_int64(100), C.c_int64(122), C.c_int64(1000), C.c_int64(1021), P(ql), P(qr), P(rl), C.c_int64(cap))
    assert n == 4
    assert list(ql[:4]) == [100, 111, 118, 121] and list(qr[:4]) == [109, 115, 120, 122] and list(rl[:4]) == [1000, 1011, 1016, 1020]
    # '-' strand: ref_str > ref_end, ops walked backwards, ref descending
    n = api.LIB.aasm_cs_match_ranges(cs, C.c_int64(len(cs)), 0, C.c_int64(100), C.c_int64(122), C.c_int64(1021), C.c_int64(1000), P(ql), P(qr), P(rl), C.c_int64(cap))
    assert n == 4
    assert list(ql[:4]) == [100, 102, 107, 113] and list(qr[:4]) == [101, 104, 111, 122] and list(rl[:4]) == [1021, 1018, 1015, 1009]


def _solve_and_write(T, paf, tmp_path, K=64):
    import ctypes as C
    from alignasm_amd._abi import BatchOut, Opts
    api = T.api()
    view = paf.view()
    out = BatchOut()
    assert T.oracle().oracle_solve_batch(C.byref(view), C.byref(Opts(K, 0, 0, 0, 0)), 2, C.byref(out)) == 0
    paths = [str(tmp_path / n) for n in ("x.aln.paf", "x.aln.alt.paf", "x.aln.all.paf")]
    paf.write_outputs(out, *paths)
    T.oracle().oracle_free_out(C.byref(out))
    return [open(p, "rb").read() for p in paths]


def test_writer_invariants(T, tmp_path):
    """Every emitted line is a valid 15-column PAF row; uncut rows equal the input row; the
    re-cut cs tag consumes exactly the edited query/reference span (the writer verifies the
    reference's own logic_error checks, paf_data.cpp:211-218, and would raise)."""
    api = T.api()
    paf = api.Paf.synth(5, 150, 21, dup_every=6)
    text = paf.to_text().decode().splitlines()
    main, alt, allp = _solve_and_write(T, paf, tmp_path)
    assert main and main.endswith(b"\n")
    cut = 0
    for line in main.decode().splitlines() + alt.decode().splitlines() + allp.decode().splitlines():
        f = line.split("\t")
        assert len(f) == 15 and f[12] in ("tp:A:P", "tp:A:S") and f[13].startswith("xi:Z:P_") and f[14].startswith("cs:Z:")
        row = int(f[13][7:])
        src = text[row].split("\t")
        assert f[0].split(".")[0] == src[0] and f[1] == src[1] and f[4] == src[4] and f[5] == src[5] and f[6] == src[6] and f[11] == src[11]
        qs, qe, rs, re_ = int(f[2]), int(f[3]), int(f[7]), int(f[8])
        assert int(src[2]) <= qs < qe <= int(src[3]) and int(src[7]) <= rs < re_ <= int(src[8])
        if (qs, qe) == (int(src[2]), int(src[3])):
            assert f[7:11] == src[7:11] and f[14] == src[13]          # not cut: identical (alignasm.cpp:419-425)
        else:
            cut += 1
            # recount the edited cs
            import re
            q = r = m = 0
            for op in re.findall(r":[0-9]+|\*[a-z][a-z]|[+-][a-z]+", f[14][5:]):
                if op[0] == ":": n = int(op[1:]); q += n; r += n; m += n
                elif op[0] == "*": q += 1; r += 1
                elif op[0] == "+": q += len(op) - 1
                else: r += len(op) - 1
            assert q == qe - qs and r == re_ - rs and m == int(f[9])
    assert cut > 0


def test_cli_rejects_wrong_extension(T, tmp_path):
    import subprocess
    exe = os.path.join(T.ROOT, "alignasm_amd", "alignasm")
    if not os.path.exists(exe):
        pytest.skip("CLI not built")
    p = tmp_path / "x.txt"
    p.write_text("")
    r = subprocess.run([exe, str(p)], capture_output=True, text=True)
    assert r.returncode == 1 and "Wrong PAF file" in r.stderr
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1


def _arrays(paf):
    hb = paf.batch()
    return {k: v.copy() for k, v in hb.arrays.items()}


def test_reader_result_does_not_depend_on_thread_count(T):
    """The row-parallel reader (chunks cut at line starts, contigs and reference names that
    continue across a cut) gives the same batch for every thread count, and the same text back."""
    api = T.api()
    text = api.Paf.synth(40, 120, 5, dup_every=5, shuffle=True).to_text()
    assert len(text) > (1 << 20)                       # several chunks
    old = api.set_host_threads(1)
    try:
        ref = api.Paf.parse(text)
        want = _arrays(ref)
        for n in (2, 3, 7, 16, 0):
            api.set_host_threads(n)
            got = api.Paf.parse(text)
            assert got.n_contigs == ref.n_contigs == 40
            for k in want:
                assert np.array_equal(want[k], _arrays(got)[k]), (n, k)
            assert got.to_text() == text
    finally:
        api.set_host_threads(old)


def test_fast_cs_scanner_equals_the_tokenizer_path(T):
    """Ranges written by the reader's fused scanner == get_overlap_range through the cs
    tokenizer (aasm_cs_match_ranges), on both strands, record by record."""
    import ctypes as C
    api = T.api()
    paf = api.Paf.synth(4, 150, 9)
    text = paf.to_text()
    a = _arrays(api.Paf.parse(text))
    rows = text.decode().splitlines()
    P = lambda x: x.ctypes.data_as(C.c_void_p)
    n_rev = 0
    for r, row in enumerate(rows):
        cs = row.split("\t")[-1].encode()
        lo, hi = int(a["rec_rng_off"][r]), int(a["rec_rng_off"][r + 1])
        cap = hi - lo + 4
        ql, qr, rl = (np.zeros(cap, np.int64) for _ in range(3))
        n = api.LIB.aasm_cs_match_ranges(cs, C.c_int64(len(cs)), int(a["aln_fwd"][r]), C.c_int64(int(a["qry_str"][r])), C.c_int64(int(a["qry_end"][r])),
                                         C.c_int64(int(a["ref_str"][r])), C.c_int64(int(a["ref_end"][r])), P(ql), P(qr), P(rl), C.c_int64(cap))
        assert n == hi - lo
        assert np.array_equal(ql[:n], a["rng_qry_l"][lo:hi]) and np.array_equal(qr[:n], a["rng_qry_r"][lo:hi]) and np.array_equal(rl[:n], a["rng_ref_l"][lo:hi])
        n_rev += int(a["aln_fwd"][r] == 0)
    assert n_rev >= 10


def test_reader_line_endings_blank_lines_and_errors_in_big_files(T):
    api = T.api()
    text = api.Paf.synth(30, 100, 13).to_text()
    want = _arrays(api.Paf.parse(text))
    lines = text.split(b"\n")[:-1]
    crlf = b"\r\n".join(lines) + b"\r\n"
    gaps = b"\n\n".join(lines)                           # blank lines are skipped, no final newline
    for variant in (crlf, gaps):
        got = _arrays(api.Paf.parse(variant))
        for k in want:
            assert np.array_equal(want[k], got[k]), k
    # an error deep inside a multi-chunk file is reported with its row number, as by the serial reader
    bad_row = 2345
    broken = list(lines)
    broken[bad_row] = broken[bad_row].replace(b"cs:Z::", b"cs:Z::9999999", 1)
    with pytest.raises(api.AlignasmError) as e:
        api.Paf.parse(b"\n".join(broken) + b"\n")
    assert "cs tag consumption does not match PAF coordinates (row %d)" % bad_row in str(e.value)
    broken = list(lines)
    broken[bad_row] = b"\t".join(broken[bad_row].split(b"\t")[:12])
    with pytest.raises(api.AlignasmError) as e:
        api.Paf.parse(b"\n".join(broken) + b"\n")
    assert "Missing cs:Z tag" in str(e.value)


def test_writers_do_not_depend_on_thread_count(T, tmp_path):
    api = T.api()
    paf = api.Paf.synth(24, 150, 21, dup_every=6)
    old = api.set_host_threads(1)
    try:
        want = _solve_and_write(T, paf, tmp_path)
        for n in (3, 8):
            api.set_host_threads(n)
            assert _solve_and_write(T, paf, tmp_path) == want
    finally:
        api.set_host_threads(old)
    assert all(len(x) > 0 for x in (want[0], want[2]))
