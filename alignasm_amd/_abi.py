"""ctypes mirror of include/alignasm_amd.h (ABI version 3).

Plumbing only: no computation lives in Python.  The structs are shared by the product
binding (alignasm_amd.api) and by the test-side loaders of the oracle libraries.
"""
import ctypes as C

import numpy as np

AASM_N_PHASES = 16
PHASE_NAMES = ["sort", "pairs", "edges", "revcsr", "sptree", "fwd", "heap", "enum", "select", "gather", "heap_prep", "topo", "misc", "cs", "final", "chain"]

AASM_OK = 0
AASM_E_INVAL, AASM_E_NODEVICE, AASM_E_HIP, AASM_E_NOMEM = -1, -2, -3, -4
AASM_E_OVERFLOW, AASM_E_INTERNAL, AASM_E_PARSE, AASM_E_IO = -5, -6, -7, -8

_IN_ARRAYS = [
    ("ctg_rec_off", np.int64), ("qry_str", np.int64), ("qry_end", np.int64), ("ref_str", np.int64),
    ("ref_end", np.int64), ("qry_total", np.int64), ("ref_chr", np.int32), ("aln_fwd", np.uint8),
    ("map_qul", np.uint8), ("rec_rng_off", np.int64), ("rng_qry_l", np.int64), ("rng_qry_r", np.int64),
    ("rng_ref_l", np.int64),
]


class BatchIn(C.Structure):
    _fields_ = [("n_contigs", C.c_int64), ("n_records", C.c_int64), ("n_ranges", C.c_int64)] + [
        (name, C.c_void_p) for name, _ in _IN_ARRAYS
    ] + [("cs_text", C.c_void_p), ("rec_cs_off", C.c_void_p)]


class Opts(C.Structure):
    _fields_ = [
        ("max_paths", C.c_int32), ("non_skip_linkable", C.c_int32), ("device", C.c_int32),
        ("collect_timing", C.c_int32), ("keep_debug", C.c_int32), ("reserved", C.c_int32 * 3),
    ]


class OutElem(C.Structure):
    _fields_ = [
        ("edited_qry_str", C.c_int64), ("edited_qry_end", C.c_int64), ("edited_ref_str", C.c_int64),
        ("edited_ref_end", C.c_int64), ("ctg_index", C.c_int32), ("is_alt_path", C.c_int32),
    ]


OUT_ELEM_DTYPE = np.dtype(
    [("qs", np.int64), ("qe", np.int64), ("rs", np.int64), ("re", np.int64), ("ctg_index", np.int32), ("is_alt", np.int32)]
)

_STAT_I64 = [
    "n_vertices", "n_pairs", "n_edges", "n_heap_nodes", "n_paths_found", "n_paths_converted",
    "n_unconnectable", "n_internal_errors", "n_single", "range_steps", "device_bytes",
    "ispr_edges", "ispr_vertices", "path_edges", "out_elems", "pq_pushes",
]


class Stats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in _STAT_I64] + [
        ("phase_ms", C.c_float * AASM_N_PHASES), ("total_ms", C.c_float), ("reserved_f", C.c_float * 3),
    ]

    def as_dict(self):
        d = {n: int(getattr(self, n)) for n in _STAT_I64}
        d["phase_ms"] = {PHASE_NAMES[i]: float(self.phase_ms[i]) for i in range(len(PHASE_NAMES))}
        d["total_ms"] = float(self.total_ms)
        return d


class BatchOut(C.Structure):
    _fields_ = [
        ("n_contigs", C.c_int64), ("main_off", C.c_void_p), ("alt_off", C.c_void_p), ("all_path_off", C.c_void_p),
        ("all_elem_off", C.c_void_p), ("main_elems", C.c_void_p), ("alt_elems", C.c_void_p), ("all_elems", C.c_void_p),
        ("n_all_paths", C.c_int64), ("ctg_status", C.c_void_p), ("stats", Stats),
    ]


class SynthCfg(C.Structure):
    _fields_ = [
        ("n_contigs", C.c_int64), ("recs_per_contig", C.c_int64), ("seed", C.c_uint64), ("dense", C.c_int32),
        ("heavy_tail", C.c_int32), ("dup_every", C.c_int32), ("reserved", C.c_int32),
    ]


def _np_from(ptr, n, dtype):
    if n <= 0 or not ptr:
        return np.zeros(0, dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


def unpack_out(out: BatchOut):
    """BatchOut -> dict of numpy arrays (copies; the C side can be freed afterwards)."""
    c = int(out.n_contigs)
    main_off = _np_from(out.main_off, c + 1, np.int64)
    alt_off = _np_from(out.alt_off, c + 1, np.int64)
    all_path_off = _np_from(out.all_path_off, c + 1, np.int64)
    npaths = int(out.n_all_paths)
    all_elem_off = _np_from(out.all_elem_off, npaths + 1, np.int64)
    return {
        "n_contigs": c,
        "main_off": main_off,
        "alt_off": alt_off,
        "all_path_off": all_path_off,
        "all_elem_off": all_elem_off,
        "main": _np_from(out.main_elems, int(main_off[-1]) if c else 0, OUT_ELEM_DTYPE),
        "alt": _np_from(out.alt_elems, int(alt_off[-1]) if c else 0, OUT_ELEM_DTYPE),
        "all": _np_from(out.all_elems, int(all_elem_off[-1]) if npaths else 0, OUT_ELEM_DTYPE),
        "status": _np_from(out.ctg_status, c, np.int32),
        "stats": out.stats.as_dict(),
    }


class HostBatch:
    """A batch held in numpy arrays + the BatchIn view pointing at them."""

    def __init__(self, arrays: dict):
        self.arrays = {}
        for name, dt in _IN_ARRAYS:
            self.arrays[name] = np.ascontiguousarray(arrays[name], dtype=dt)
        self.view = BatchIn()
        self.view.n_contigs = len(self.arrays["ctg_rec_off"]) - 1
        self.view.n_records = len(self.arrays["qry_str"])
        self.view.n_ranges = len(self.arrays["rng_qry_l"])
        for name, _ in _IN_ARRAYS:
            setattr(self.view, name, self.arrays[name].ctypes.data)

    @property
    def n_contigs(self):
        return int(self.view.n_contigs)

    @staticmethod
    def from_view(view: BatchIn):
        """Copy a borrowed BatchIn (e.g. from aasm_paf_batch) into numpy arrays."""
        c, r, g = int(view.n_contigs), int(view.n_records), int(view.n_ranges)
        sizes = {
            "ctg_rec_off": c + 1, "qry_str": r, "qry_end": r, "ref_str": r, "ref_end": r, "qry_total": r,
            "ref_chr": r, "aln_fwd": r, "map_qul": r, "rec_rng_off": r + 1, "rng_qry_l": g, "rng_qry_r": g, "rng_ref_l": g,
        }
        return HostBatch({name: _np_from(getattr(view, name), sizes[name], dt) for name, dt in _IN_ARRAYS})

    @staticmethod
    def from_view_range(view: BatchIn, c0, c1):
        """Copy only contigs [c0, c1) of a borrowed BatchIn (offsets rebased to 0)."""
        off = _np_from(view.ctg_rec_off, int(view.n_contigs) + 1, np.int64)
        r0, r1 = int(off[c0]), int(off[c1])
        isz = {n: np.dtype(dt).itemsize for n, dt in _IN_ARRAYS}
        ro = _np_from(view.rec_rng_off + r0 * 8, r1 - r0 + 1, np.int64)
        g0, g1 = int(ro[0]), int(ro[-1])
        out = {"ctg_rec_off": off[c0:c1 + 1] - r0, "rec_rng_off": ro - g0}
        for name, dt in _IN_ARRAYS:
            if name in out:
                continue
            base = getattr(view, name)
            if name.startswith("rng_"):
                out[name] = _np_from(base + g0 * isz[name], g1 - g0, dt)
            else:
                out[name] = _np_from(base + r0 * isz[name], r1 - r0, dt)
        return HostBatch(out)

    def subset(self, contigs):
        """New HostBatch holding only the given contigs (used for sharding and tests)."""
        a = self.arrays
        off = a["ctg_rec_off"]
        rec_idx = np.concatenate([np.arange(off[c], off[c + 1]) for c in contigs]) if len(contigs) else np.zeros(0, np.int64)
        new_off = np.zeros(len(contigs) + 1, np.int64)
        new_off[1:] = np.cumsum([off[c + 1] - off[c] for c in contigs])
        ro = a["rec_rng_off"]
        lens = ro[rec_idx + 1] - ro[rec_idx] if len(rec_idx) else np.zeros(0, np.int64)
        new_ro = np.zeros(len(rec_idx) + 1, np.int64)
        new_ro[1:] = np.cumsum(lens)
        if len(rec_idx):
            rng_idx = np.concatenate([np.arange(ro[r], ro[r + 1]) for r in rec_idx])
        else:
            rng_idx = np.zeros(0, np.int64)
        out = {"ctg_rec_off": new_off, "rec_rng_off": new_ro}
        for name in ("qry_str", "qry_end", "ref_str", "ref_end", "qry_total", "ref_chr", "aln_fwd", "map_qul"):
            out[name] = a[name][rec_idx]
        for name in ("rng_qry_l", "rng_qry_r", "rng_ref_l"):
            out[name] = a[name][rng_idx]
        return HostBatch(out)
