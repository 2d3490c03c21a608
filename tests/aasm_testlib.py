"""Shared helpers for the test tiers.

Loads the CHECKERS (never used by the product):
  oracle/liboracle.so                     CPU restatement of the reference
  oracle/_ref/libaasm_ref_algos*.so       the real reference's algorithm headers (optional;
                                          built only where /root/reference exists, the
                                          prebuilt files travel to the GPU box)
  tests/host_emul/libaasm_emul.so         1-lane host build of the product's kernel bodies
and the PRODUCT through alignasm_amd.api (libalignasm_amd.so, C-ABI).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from alignasm_amd._abi import BatchOut, HostBatch, Opts, unpack_out

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libaasm_ref_algos.so")
REF_MONO_SO = os.path.join(ROOT, "oracle", "_ref", "libaasm_ref_algos_mono.so")
REF_CS_SO = os.path.join(ROOT, "oracle", "_ref", "libaasm_ref_cs.so")
REF_PREFIX_SO = os.path.join(ROOT, "oracle", "_ref", "libaasm_ref_prefix.so")
REF_PREFIX_MONO_SO = os.path.join(ROOT, "oracle", "_ref", "libaasm_ref_prefix_mono.so")
EMUL_SO = os.path.join(ROOT, "tests", "host_emul", "libaasm_emul.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")

_i64p = C.POINTER(C.c_int64)


def _P(a):
    return a.ctypes.data_as(_i64p)


def _ensure(path, make_dir):
    if not os.path.exists(path):
        subprocess.run(["make", "-s", "-C", make_dir], check=True)
    return path


_cache = {}


def oracle():
    if "o" not in _cache:
        lib = C.CDLL(_ensure(ORACLE_SO, os.path.join(ROOT, "oracle")))
        for fn in ("oracle_debug_size", "oracle_debug_copy", "oracle_generic_kwalks", "oracle_generic_path", "oracle_generic_fetch"):
            getattr(lib, fn).restype = C.c_int64
        _cache["o"] = lib
    return _cache["o"]


def emul():
    if "e" not in _cache:
        lib = C.CDLL(_ensure(EMUL_SO, os.path.join(ROOT, "tests", "host_emul")))
        lib.emul_debug_fetch.restype = C.c_int64
        lib.emul_last_bad_record.restype = C.c_int64
        _cache["e"] = lib
    return _cache["e"]


def ref(mono=True):
    key = "rm" if mono else "r"
    path = REF_MONO_SO if mono else REF_SO
    if key not in _cache:
        if not os.path.exists(path):
            return None
        lib = C.CDLL(path)
        for fn in ("ref_generic_kwalks", "ref_generic_path", "ref_generic_fetch", "ref_generic_heap", "ref_generic_arena_inversions"):
            getattr(lib, fn).restype = C.c_int64
        _cache[key] = lib
    return _cache[key]


def ref_cs():
    """oracle/_ref/libaasm_ref_cs.so: the REAL get_overlap_range / get_edited_paf_data (paf_data.cpp:15-220), or None."""
    if "rcs" not in _cache:
        if not os.path.exists(REF_CS_SO):
            return None
        lib = C.CDLL(REF_CS_SO)
        lib.ref_cs_overlap_range.restype = C.c_int64
        lib.ref_cs_edit.restype = C.c_int64
        _cache["rcs"] = lib
    return _cache["rcs"]


def ref_prefix(mono=True):
    """oracle/_ref/libaasm_ref_prefix[_mono].so: the REAL solve_ctg_read() from the sort through the k-walk distances
    (paf_data.cpp:223-738 = K1 ... K8) with its locals copied out, or None where it was not built."""
    key = "rpm" if mono else "rp"
    path = REF_PREFIX_MONO_SO if mono else REF_PREFIX_SO
    if key not in _cache:
        if not os.path.exists(path):
            return None
        lib = C.CDLL(path)
        lib.refp_debug_size.restype = C.c_int64
        lib.refp_debug_copy.restype = C.c_int64
        lib.refp_time_batch.restype = C.c_double
        _cache[key] = lib
    return _cache[key]


# the arrays the reference prefix and the oracle both record (same names, same meaning)
PREFIX_NAMES = ("perm", "part_idx", "vtx_i", "vtx_j", "pair_pe_q", "pair_pe_r", "pair_st_q", "pair_st_r", "csr_rowptr", "csr_col",
                "csr_w_qry", "csr_w_ref", "csr_w_anom", "csr_w_qnz", "csr_w_qtot", "anom_dis_dest", "sp_d_qry", "sp_d_ref",
                "sp_d_anom", "sp_d_qnz", "sp_d_qtot", "sp_best", "rev_order", "fwd_order", "kd_qry", "kd_ref", "kd_anom",
                "kd_qnz", "kd_qtot", "heap_nodes", "heap_key_qry", "heap_left", "heap_right", "heap_u", "heap_v", "heap_rank", "heap_root")
# what only the reference prefix records
PREFIX_EXTRA = ("parts", "pair_ovidx_i", "pair_ovidx_j", "n_index_entries", "csr_w_calcsum", "src_dest", "anom_dis", "heap_key_ref",
                "heap_key_anom", "heap_key_qnz", "heap_key_qtot", "heap_addr", "k_last", "k_prev", "k_node", "path_off", "path_u",
                "path_v", "path_w_qry", "path_w_ref", "ctg_sorted_index", "vtx_index_mismatch")


def ref_prefix_debug(hb: HostBatch, contig, nsl=False, n_paths=0, mono=True, names=None):
    """Locals of the REAL solve_ctg_read() at paf_data.cpp:738 for ONE contig (K is the reference's own 10 000)."""
    lib = ref_prefix(mono)
    rc = lib.refp_debug_solve(C.byref(hb.view), C.c_int64(contig), 1 if nsl else 0, C.c_int64(n_paths))
    assert rc == 0, rc
    out = {}
    for n in (names or PREFIX_NAMES + PREFIX_EXTRA):
        sz = lib.refp_debug_size(n.encode())
        if sz < 0:
            continue
        a = np.zeros(sz, np.int64)
        lib.refp_debug_copy(n.encode(), _P(a), C.c_int64(sz))
        out[n] = a
    return out


class RefPrefixVectors:
    """tests/golden/ref_prefix.npz: input batches + what the REAL solve_ctg_read() prefix computed from them
    (recorded by tests/golden/make_ref_prefix.py from oracle/_ref/libaasm_ref_prefix_mono.so)."""

    def __init__(self, path=None):
        self.z = np.load(path or os.path.join(GOLDEN, "ref_prefix.npz"))
        self.tags = [str(t) for t in self.z["tags"]]

    def batch(self, tag):
        """-> (HostBatch, nsl, all 10 000 distances recorded?)"""
        A = {}
        pre = f"{tag}/in/"
        for k in self.z.files:
            if k.startswith(pre):
                name = k[len(pre):]
                a = self.z[k]
                if name.endswith("~d"):
                    name, a = name[:-2], np.cumsum(a.astype(np.int64))
                A[name] = a
        A["rng_qry_r"] = A["rng_qry_l"].astype(np.int64) + A.pop("rng_qry_r~len")
        return HostBatch(A), bool(self.z[f"{tag}/nsl"][0]), bool(self.z[f"{tag}/full"][0])

    def contig(self, tag, c):
        """The PREFIX_NAMES arrays of contig c (int64), + "kfound" = how many distances the reference found."""
        out = {}
        pre = f"{tag}/c{c}/"
        for k in self.z.files:
            if k.startswith(pre):
                name = k[len(pre):]
                a = self.z[k].astype(np.int64)
                if name.endswith("~b"):
                    name, a = name[:-2], np.where(a > 0, np.arange(len(a)) - a, -1)
                out[name] = a
        return out


_cs_buf = {}


def _cs_bufs(cap):
    if _cs_buf.get("cap", 0) < cap:
        _cs_buf.update(cap=cap, a=[np.zeros(cap, np.int64) for _ in range(4)], err=C.create_string_buffer(256), out=C.create_string_buffer(1 << 16))
    return _cs_buf


def ref_cs_ranges(row):
    """Reference get_overlap_range on a tests/cs_cases.py row -> ("ok", [(ql, qr, rl, rr), ...]) or ("err", code, text)."""
    cs = row["cs"].encode()
    b = _cs_bufs(max(64, len(cs)))
    n = ref_cs().ref_cs_overlap_range(cs, C.c_int64(len(cs)), 1 if row["fwd"] else 0, C.c_int64(row["qs"]), C.c_int64(row["qe"]), C.c_int64(row["rs"]),
                                      C.c_int64(row["re"]), _P(b["a"][0]), _P(b["a"][1]), _P(b["a"][2]), _P(b["a"][3]), C.c_int64(b["cap"]), b["err"], C.c_int64(256))
    if n < 0:
        return ("err", int(n), b["err"].value.decode())
    return ("ok", [tuple(int(b["a"][k][i]) for k in range(4)) for i in range(n)])


def ref_cs_edit(row, clip, mat_num=7, aln_len=9):
    """Reference get_edited_paf_data -> ("ok", cs, mat_num, aln_len, is_cut) or ("err", code, text)."""
    cs = row["cs"].encode()
    b = _cs_bufs(max(64, len(cs)))
    m, al, cut = C.c_int32(mat_num), C.c_int32(aln_len), C.c_int32(-1)
    n = ref_cs().ref_cs_edit(cs, C.c_int64(len(cs)), 1 if row["fwd"] else 0, C.c_int64(row["qs"]), C.c_int64(row["qe"]), C.c_int64(row["rs"]), C.c_int64(row["re"]),
                             C.c_int64(clip[0]), C.c_int64(clip[1]), C.c_int64(clip[2]), C.c_int64(clip[3]),
                             b["out"], C.c_int64(len(b["out"])), C.byref(m), C.byref(al), C.byref(cut), b["err"], C.c_int64(256))
    if n < 0:
        return ("err", int(n), b["err"].value.decode())
    assert n < len(b["out"])
    return ("ok", b["out"].value.decode(), m.value, al.value, bool(cut.value))


def product_cs_ranges(row):
    """Product host codec (aasm_cs_match_ranges) in the same shape; the product stores no ref_r (derivable)."""
    a = api()
    cs = row["cs"].encode()
    b = _cs_bufs(max(64, len(cs)))
    P = lambda x: x.ctypes.data_as(C.c_void_p)
    n = a.LIB.aasm_cs_match_ranges(cs, C.c_int64(len(cs)), 1 if row["fwd"] else 0, C.c_int64(row["qs"]), C.c_int64(row["qe"]), C.c_int64(row["rs"]),
                                   C.c_int64(row["re"]), P(b["a"][0]), P(b["a"][1]), P(b["a"][2]), C.c_int64(b["cap"]))
    if n < 0:
        return ("err", int(n), a.LIB.aasm_last_error().decode())
    step = 1 if row["fwd"] else -1
    return ("ok", [(int(b["a"][0][i]), int(b["a"][1][i]), int(b["a"][2][i]), int(b["a"][2][i] + (b["a"][1][i] - b["a"][0][i]) * step)) for i in range(n)])


def product_cs_edit(row, clip, mat_num=7, aln_len=9):
    a = api()
    cs = row["cs"].encode()
    b = _cs_bufs(max(64, len(cs)))
    m, al, cut = C.c_int32(mat_num), C.c_int32(aln_len), C.c_int32(-1)
    n = a.LIB.aasm_cs_edit(cs, C.c_int64(len(cs)), 1 if row["fwd"] else 0, C.c_int64(row["qs"]), C.c_int64(row["qe"]), C.c_int64(clip[0]), C.c_int64(clip[1]),
                           C.c_int64(clip[2]), C.c_int64(clip[3]), b["out"], C.c_int64(len(b["out"])), C.byref(m), C.byref(al), C.byref(cut))
    if n < 0:
        return ("err", int(n), a.LIB.aasm_last_error().decode())
    return ("ok", b["out"].value.decode(), m.value, al.value, bool(cut.value))


def io_cs_ranges(row):
    """oracle/paf_io_oracle.py get_overlap_range in the same shape."""
    io = io_oracle()
    r = io.PafRow()
    r.aln_fwd, r.qry_str, r.qry_end, r.ref_str, r.ref_end, r.cs_string = row["fwd"], row["qs"], row["qe"], row["rs"], row["re"], row["cs"]
    try:
        io.get_overlap_range(r, row["cs"])
    except io.CsError as e:
        return ("err", -1, str(e))
    return ("ok", [(q[0], q[1], f[0], f[1]) for q, f in zip(r.qry_rng, r.ref_rng)])


def io_cs_edit(row, clip, mat_num=7, aln_len=9):
    io = io_oracle()
    r = io.PafRow()
    r.aln_fwd, r.qry_str, r.qry_end, r.ref_str, r.ref_end, r.cs_string = row["fwd"], row["qs"], row["qe"], row["rs"], row["re"], row["cs"]
    r.mat_num, r.aln_len = mat_num, aln_len
    try:
        return ("ok",) + tuple(io.get_edited_paf_data(clip[0], clip[1], clip[2], clip[3], r))
    except io.CsLogicError as e:
        return ("err", -3, str(e))
    except io.CsError as e:
        return ("err", -1, str(e))


def api():
    import alignasm_amd.api as a
    return a


def io_oracle():
    """oracle/paf_io_oracle.py: plain-Python restatement of the reference's reader, cs codec,
    --alt merge and writers (test infrastructure; independent of the product's aasm_paf.cpp)."""
    if "io" not in _cache:
        import importlib.util
        spec = importlib.util.spec_from_file_location("paf_io_oracle", os.path.join(ROOT, "oracle", "paf_io_oracle.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _cache["io"] = mod
    return _cache["io"]


def io_oracle_batch(st) -> HostBatch:
    """The records an oracle-side reader state holds, as a HostBatch for oracle_solve / the product."""
    return HostBatch(io_oracle().to_arrays(st))


def io_oracle_files(text, alt_text=None, alt_baseline=0.5, K=10000, nsl=False, solver=None):
    """The whole reference pipeline on the oracle side: reader -> [--alt merge] -> solve -> writers.
    Returns (main, alt, all) bytes.  solver(hb, K, nsl) defaults to the CPU oracle."""
    io = io_oracle()
    st = io.read_paf(text)
    if alt_text:
        io.merge_alt(st, alt_text, alt_baseline)
    sol = (solver or oracle_solve)(io_oracle_batch(st), K, nsl)
    return io.render_outputs(st, sol)


def synth(n_contigs, recs, seed, dense=False, dup_every=0, shuffle=False, heavy_tail=False) -> HostBatch:
    paf = api().Paf.synth(n_contigs, recs, seed, dense=dense, heavy_tail=heavy_tail, dup_every=dup_every, shuffle=shuffle, no_cs=True)
    hb = paf.batch()
    paf.close()
    return hb


def oracle_solve(hb: HostBatch, K=10000, nsl=False, threads=4):
    o = Opts(int(K), 1 if nsl else 0, 0, 0, 0)
    out = BatchOut()
    rc = oracle().oracle_solve_batch(C.byref(hb.view), C.byref(o), int(threads), C.byref(out))
    assert rc == 0, rc
    try:
        return unpack_out(out)
    finally:
        oracle().oracle_free_out(C.byref(out))


def emul_solve(hb: HostBatch, K=10000, nsl=False, sequential_select=False, heap_waves="auto", heap_input_order=False, chain="auto", graph_launches=False, chain_own_queue=False, test_small_root_ring=False):
    o = Opts(int(K), 1 if nsl else 0, 0, 0, 1)
    o.reserved[0] = (1 if sequential_select else 0) | {"auto": 0, "all": 2, "none": 4}[heap_waves] | ((1 << 8) if heap_input_order else 0) | {"auto": 0, "all": 64, "none": 128, "half": 192}[chain] | ((1 << 16) if graph_launches else 0)
    o.reserved[2] = (32 if chain_own_queue else 0) | (64 if test_small_root_ring else 0)
    out = BatchOut()
    rc = emul().emul_solve_batch(C.byref(hb.view), C.byref(o), C.byref(out))
    assert rc == 0, rc
    try:
        return unpack_out(out)
    finally:
        emul().emul_free_out(C.byref(out))


def k0_ranges(fetch):
    """K0's output (one 32-byte record {qry_l, qry_r, ref_l, -} per match range) as the three columns the caller's arrays hold:
    a dict with the old per-array names, for `fetch` = emul_debug or DeviceResult.debug."""
    rec = fetch("rng_rec", np.int64)
    rec = rec[:len(rec) // 4 * 4].reshape(-1, 4)
    return {"rql_w": rec[:, 0].copy(), "rqr_w": rec[:, 1].copy(), "rrl_w": rec[:, 2].copy()}


def emul_debug(name, dtype):
    n = emul().emul_debug_fetch(name.encode(), None, C.c_int64(0))
    assert n >= 0, name
    buf = np.zeros(n // np.dtype(dtype).itemsize, dtype)
    emul().emul_debug_fetch(name.encode(), buf.ctypes.data_as(C.c_void_p), C.c_int64(buf.nbytes))
    return buf


OUT_KEYS = ("main_off", "alt_off", "all_path_off", "all_elem_off", "main", "alt", "all", "status")
STAT_KEYS = ("n_vertices", "n_pairs", "n_edges", "n_heap_nodes", "n_paths_found", "n_paths_converted",
             "n_unconnectable", "n_internal_errors", "n_single")


def diff_outputs(want, got, stats=True):
    bad = [k for k in OUT_KEYS if not np.array_equal(want[k], got[k])]
    if stats:
        bad += [f"stats.{k}" for k in STAT_KEYS if want["stats"][k] != got["stats"][k]]
    return bad


def oracle_debug(hb: HostBatch, contig, K=10000, nsl=False):
    """Intermediates of ONE contig from the oracle, as a dict of int64 arrays."""
    o = Opts(int(K), 1 if nsl else 0, 0, 0, 0)
    lib = oracle()
    assert lib.oracle_debug_solve(C.byref(hb.view), C.byref(o), C.c_int64(contig)) == 0
    names = ["perm", "part_idx", "vtx_i", "vtx_j", "pair_pe_q", "pair_pe_r", "pair_st_q", "pair_st_r", "csr_rowptr", "csr_col",
             "csr_w_qry", "csr_w_ref", "csr_w_anom", "csr_w_qnz", "csr_w_qtot", "anom_dis_dest", "sp_d_qry", "sp_d_ref",
             "sp_d_anom", "sp_d_qnz", "sp_d_qtot", "sp_best", "rev_order", "fwd_order", "kd_qry", "kd_ref", "kd_anom",
             "kd_qnz", "kd_qtot", "heap_nodes", "heap_key_qry", "heap_left", "heap_right", "heap_u", "heap_v", "heap_rank", "heap_root"]
    out = {}
    for n in names:
        sz = lib.oracle_debug_size(n.encode())
        if sz < 0:
            continue
        a = np.zeros(sz, np.int64)
        lib.oracle_debug_copy(n.encode(), _P(a), C.c_int64(sz))
        out[n] = a
    return out


DIST_DT = np.dtype([("qry", np.int64), ("ref", np.int64), ("anom", np.int32), ("qnz", np.int32), ("qtot", np.int32), ("pad", np.int32)])
HNODE_DT = np.dtype([("kq", np.int64), ("kr", np.int64), ("ka", np.int32), ("kn", np.int32), ("kt", np.int32), ("rank", np.int32),
                     ("left", np.int32), ("right", np.int32), ("u", np.int32), ("v", np.int32)])


def diff_intermediates(hb, fetch, K=10000, nsl=False, contigs=None, expect=None):
    """Compare per-contig intermediates of a product/emulation run with the oracle - or, with
    `expect(c)` (a dict of the PREFIX_NAMES arrays: vectors recorded from the REAL reference's
    solve_ctg_read() prefix, or that library live), with the reference itself.  The reference always
    runs K = 10 000; a run at a smaller K is compared with the first K distances.

    `fetch(name, dtype)` returns the batch-level workspace array `name`.
    Returns a list of (contig, what) mismatches.
    """
    rec_off = hb.arrays["ctg_rec_off"]
    C_ = len(rec_off) - 1
    ctgV = fetch("ctgV", np.int32)[:C_]
    voff = fetch("voff", np.int64)[:C_ + 1]      # debug arrays carry allocation padding
    perm = fetch("perm", np.int32)
    s_pid = fetch("s_pid", np.int32)
    bad = []
    has_graph = int(voff[-1]) > 0
    if has_graph:
        v_i, v_j, v_slot = fetch("v_i", np.int32), fetch("v_j", np.int32), fetch("v_slot", np.int64)
        rowptr, col = fetch("csr_rowptr", np.int64), fetch("csr_col", np.int32)
        wq, wr, fl = fetch("csr_w_qry", np.int64), fetch("csr_w_ref", np.int32), fetch("csr_w_flags", np.uint8)
        ov = {k: fetch(k, np.int64) for k in ("ov_peq", "ov_per", "ov_stq", "ov_str")}
        sp_d, sp_best = fetch("sp_d", DIST_DT), fetch("sp_best", np.int32)
        rev_order, fwd_order = fetch("rev_order", np.int32), fetch("fwd_order", np.int32)
        anom_dest, kfound, h_cnt = fetch("anom_dest", np.int32), fetch("kfound", np.int32), fetch("h_cnt", np.int32)
        kd = fetch("kd", DIST_DT)
        hoff, hnodes, h_root = fetch("hoff", np.int64), fetch("hnodes", HNODE_DT), fetch("h_root", np.int32)
    for c in (contigs if contigs is not None else range(C_)):
        b, N = int(rec_off[c]), int(rec_off[c + 1] - rec_off[c])
        if N <= 1:
            continue
        o = expect(c) if expect is not None else oracle_debug(hb, c, K, nsl)
        if expect is not None and len(o["kd_qry"]) > K:
            o = dict(o)
            for k in ("kd_qry", "kd_ref", "kd_anom", "kd_qnz", "kd_qtot"):
                o[k] = o[k][:K]

        def chk(what, a, bexp):
            if not np.array_equal(np.asarray(a, np.int64), np.asarray(bexp, np.int64)):
                bad.append((c, what))
        chk("perm", perm[b:b + N], o["perm"])
        chk("part_idx", s_pid[b:b + N], o["part_idx"])
        V = int(ctgV[c])
        if V != len(o["csr_rowptr"]) - 1:
            bad.append((c, "V"))
            continue
        vb = int(voff[c])
        chk("vtx_i", v_i[vb:vb + V - 2], o["vtx_i"])
        chk("vtx_j", v_j[vb:vb + V - 2], o["vtx_j"])
        sl = v_slot[vb + N:vb + V - 2]
        chk("pair_pe_q", ov["ov_peq"][sl], o["pair_pe_q"]); chk("pair_pe_r", ov["ov_per"][sl], o["pair_pe_r"])
        chk("pair_st_q", ov["ov_stq"][sl], o["pair_st_q"]); chk("pair_st_r", ov["ov_str"][sl], o["pair_st_r"])
        e0, e1 = int(rowptr[vb]), int(rowptr[vb + V])
        chk("rowptr", rowptr[vb:vb + V + 1] - e0, o["csr_rowptr"])
        chk("col", col[e0:e1], o["csr_col"])
        chk("w_qry", wq[e0:e1], o["csr_w_qry"]); chk("w_ref", wr[e0:e1], o["csr_w_ref"])
        chk("w_anom", fl[e0:e1] & 3, o["csr_w_anom"]); chk("w_qnz", (fl[e0:e1] >> 2) & 1, o["csr_w_qnz"]); chk("w_qtot", (fl[e0:e1] >> 3) & 1, o["csr_w_qtot"])
        chk("anom_dest", [anom_dest[c]], o["anom_dis_dest"])
        d = sp_d[vb:vb + V]
        for f, k in (("qry", "sp_d_qry"), ("ref", "sp_d_ref"), ("anom", "sp_d_anom"), ("qnz", "sp_d_qnz"), ("qtot", "sp_d_qtot")):
            chk(k, d[f], o[k])
        chk("sp_best", sp_best[vb:vb + V], o["sp_best"])
        chk("rev_order", rev_order[vb:vb + V], o["rev_order"])
        chk("fwd_order", fwd_order[vb:vb + V], o["fwd_order"])
        nf = int(kfound[c])
        ncmp = len(o["kd_qry"])                                        # a recorded list may be cut short ("kfound" = its full length)
        chk("kfound", [nf], [min(K, int(o["kfound"][0]))] if "kfound" in o else [ncmp])
        kk = kd[c * K:c * K + min(nf, ncmp)]
        for f, k in (("qry", "kd_qry"), ("ref", "kd_ref"), ("anom", "kd_anom"), ("qnz", "kd_qnz"), ("qtot", "kd_qtot")):
            chk(k, kk[f], o[k][:len(kk)])
        chk("heap_count", [h_cnt[c]], o["heap_nodes"])
        hn = hnodes[int(hoff[c]):int(hoff[c]) + int(h_cnt[c])]
        chk("heap_key_qry", hn["kq"], o["heap_key_qry"]); chk("heap_left", hn["left"], o["heap_left"]); chk("heap_right", hn["right"], o["heap_right"])
        chk("heap_u", hn["u"], o["heap_u"]); chk("heap_v", hn["v"], o["heap_v"]); chk("heap_rank", hn["rank"] & 0xff, o["heap_rank"])   # upper bytes cache the child ranks
        chk("heap_root", h_root[vb:vb + V], o["heap_root"])
    return bad


# ---- generic-graph harness (oracle vs real reference headers) -----------------------
def generic_run(lib, prefix, n, rowptr, col, w, src, sink, K, with_paths=True):
    d = np.zeros(max(K, 1) * 5, np.int64)
    nd = getattr(lib, prefix + "generic_kwalks")(C.c_int64(n), _P(rowptr), _P(col), _P(w), C.c_int64(src), C.c_int64(sink),
                                                 C.c_int64(K), _P(d), C.c_int64(K))
    out = {"nd": int(nd), "dist": d[:nd * 5].copy()}
    for what, name, sz in ((0, "anom", n), (1, "rev", n), (2, "fwd", n), (3, "best", n), (4, "d", 5 * n), (5, "hroot", n), (6, "hcount", 1)):
        a = np.zeros(sz, np.int64)
        getattr(lib, prefix + "generic_fetch")(what, _P(a), C.c_int64(sz))
        out[name] = a
    if with_paths:
        buf = np.zeros(2 * (n + 8), np.int64)
        paths = []
        for k in range(nd):
            m = getattr(lib, prefix + "generic_path")(C.c_int64(src), C.c_int64(sink), C.c_int64(k), _P(buf), C.c_int64(n + 8))
            paths.append(buf[:2 * m].copy())
        out["paths"] = paths
    return out


def contig_graph(hb, contig, K=10000, nsl=False):
    o = oracle_debug(hb, contig, K, nsl)
    rp, col = o["csr_rowptr"], o["csr_col"]
    w = np.stack([o["csr_w_qry"], o["csr_w_ref"], o["csr_w_anom"], o["csr_w_qnz"], o["csr_w_qtot"]], 1).reshape(-1).copy()
    n = len(rp) - 1
    return n, rp, col, w
