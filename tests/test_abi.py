"""C-ABI surface: the library loads, exports every symbol the header declares, and fails
loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "alignasm_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(aasm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(T):
    api = T.api()
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(api.LIB, n), f"{n} declared in include/alignasm_amd.h but not exported"
    assert set(api.EXPORTED) <= set(names)
    assert api.LIB.aasm_abi_version() == 3


def test_struct_layouts_match_header():
    from alignasm_amd import _abi
    assert C.sizeof(_abi.OutElem) == 40 and _abi.OUT_ELEM_DTYPE.itemsize == 40
    assert C.sizeof(_abi.BatchIn) == 3 * 8 + 13 * 8 + 2 * 8          # ABI 2: + cs_text, rec_cs_off
    assert C.sizeof(_abi.Opts) == 32
    assert C.sizeof(_abi.Stats) == 16 * 8 + 16 * 4 + 4 + 12
    assert C.sizeof(_abi.SynthCfg) == 40


def test_no_cpu_fallback_without_gpu(T):
    api = T.api()
    if api.device_count() > 0:
        pytest.skip("a GPU is present; the no-device error path is exercised on CPU-only boxes")
    hb = T.synth(2, 20, 1)
    with pytest.raises(api.AlignasmError) as ei:
        api.solve_batch(hb, max_paths=4)
    assert ei.value.code == -2          # AASM_E_NODEVICE


def test_product_never_references_the_oracle():
    # the product tree must not mention the checker libraries
    for dirpath, _, files in os.walk(os.path.join(ROOT, "alignasm_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hpp", ".cpp", ".hip")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in txt and "libaasm_emul" not in txt and "oracle_solve" not in txt, (dirpath, f)


def test_range_solve_and_writer_session_reject_misuse(T, tmp_path):
    """Argument checks of the entry points a streaming caller uses (no device needed to fail them): contig ranges outside the
    batch, writer ranges out of order or beyond the file; a closed-without-commit session leaves nothing on disk."""
    import ctypes as C
    from alignasm_amd._abi import BatchOut, Opts
    api = T.api()
    paf = api.Paf.synth(5, 20, 3)
    view = paf.view()
    out = BatchOut()
    for c0, c1 in ((-1, 2), (3, 3), (4, 2), (0, 6)):
        assert api.LIB.aasm_solve_batch_range(C.byref(view), C.c_int64(c0), C.c_int64(c1), C.byref(Opts(4, 0, 0, 0, 0)), C.byref(out)) == -1   # AASM_E_INVAL
    if api.device_count() == 0:                                     # a valid range without a device: the loud failure, no fallback
        assert api.LIB.aasm_solve_batch_range(C.byref(view), C.c_int64(0), C.c_int64(5), C.byref(Opts(4, 0, 0, 0, 0)), C.byref(out)) == -2
    sol = BatchOut()
    assert T.oracle().oracle_solve_batch(C.byref(view), C.byref(Opts(4, 0, 0, 0, 0)), 1, C.byref(sol)) == 0
    w = C.c_void_p()
    names = [str(tmp_path / n).encode() for n in ("m.paf", "a.paf", "l.paf")]
    assert api.LIB.aasm_writer_open(names[0], names[1], names[2], C.byref(w)) == 0
    assert api.LIB.aasm_writer_append(w, paf._h, C.byref(sol), C.c_int64(1)) == -1        # must start at contig 0
    assert api.LIB.aasm_writer_append(w, paf._h, C.byref(sol), C.c_int64(0)) == 0
    assert api.LIB.aasm_writer_append(w, paf._h, C.byref(sol), C.c_int64(0)) == -1        # that range is already written
    assert api.LIB.aasm_writer_append(w, paf._h, C.byref(sol), C.c_int64(5)) == -1        # beyond the file
    assert api.LIB.aasm_writer_close(w, 0) == 0                                           # abandon: nothing stays
    assert list(tmp_path.iterdir()) == []
    w = C.c_void_p()
    assert api.LIB.aasm_writer_open(names[0], names[1], names[2], C.byref(w)) == 0
    assert api.LIB.aasm_writer_append(w, paf._h, C.byref(sol), C.c_int64(0)) == 0
    assert api.LIB.aasm_writer_close(w, 1) == 0
    one_piece = [str(tmp_path / n) for n in ("m1.paf", "a1.paf", "l1.paf")]
    paf.write_outputs(sol, *one_piece)
    for a, b in zip(names, one_piece):
        assert open(a, "rb").read() == open(b, "rb").read()
    assert sorted(p.name for p in tmp_path.iterdir()) == ["a.paf", "a1.paf", "l.paf", "l1.paf", "m.paf", "m1.paf"]
    T.oracle().oracle_free_out(C.byref(sol))
