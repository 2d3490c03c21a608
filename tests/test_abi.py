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
