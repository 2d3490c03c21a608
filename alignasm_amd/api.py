"""ctypes binding of libalignasm_amd.so (the C-ABI of include/alignasm_amd.h).

Host-side mirror of the reference's operator boundary for the per-contig path inference
(`solve_ctg_read`, /root/reference/src/paf_data.hpp:193): `solve_batch()` takes the records
of many contigs and returns, per contig, the three lists the reference writes to
`.aln.paf`, `.aln.alt.paf` and `.aln.all.paf`.

There is no CPU fallback: if the shared library is missing the import raises, and every
solve entry point raises `AlignasmError(AASM_E_NODEVICE)` when no MI355X/HIP device is
usable.  The CPU oracle under oracle/ is test infrastructure and is never loaded here.
"""
import ctypes as C
import os

import numpy as np

from ._abi import (AASM_OK, BatchIn, BatchOut, HostBatch, Opts, Stats, SynthCfg, unpack_out)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AASM_LIB_OVERRIDE") or os.path.join(_HERE, "libalignasm_amd.so")   # override: diagnostic builds (tools/)


class AlignasmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"alignasm_amd error {code}: {msg}")
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C alignasm_amd/csrc).  alignasm_amd has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    lib.aasm_last_error.restype = C.c_char_p
    lib.aasm_debug_fetch.restype = C.c_int64
    lib.aasm_debug_counter.restype = C.c_int64
    lib.aasm_paf_n_contigs.restype = C.c_int64
    lib.aasm_cs_match_ranges.restype = C.c_int64
    lib.aasm_cs_edit.restype = C.c_int64
    return lib


LIB = _load()

# every symbol include/alignasm_amd.h declares (checked by tests/test_abi.py)
EXPORTED = [
    "aasm_abi_version", "aasm_device_count", "aasm_init", "aasm_last_error", "aasm_solve_batch", "aasm_solve_batch_multi", "aasm_solve_device",
    "aasm_result_stats", "aasm_result_fetch", "aasm_result_free", "aasm_free_out", "aasm_upload_batch", "aasm_upload_free",
    "aasm_contig_costs", "aasm_partition_contigs", "aasm_partition_costs", "aasm_solve_batch_range", "aasm_writer_open", "aasm_writer_append", "aasm_writer_close", "aasm_reserve_workspace", "aasm_sssp_dijkstra", "aasm_sssp_dial", "aasm_debug_fetch", "aasm_debug_counter", "aasm_debug_predicates", "aasm_debug_sort_replay", "aasm_paf_read", "aasm_paf_read_opts", "aasm_paf_parse_mem", "aasm_paf_parse_mem_opts", "aasm_paf_merge_alt", "aasm_paf_merge_alt_mem", "aasm_paf_free", "aasm_paf_batch", "aasm_paf_n_contigs",
    "aasm_paf_write_outputs", "aasm_set_host_threads", "aasm_cs_match_ranges", "aasm_cs_edit", "aasm_synth_paf", "aasm_synth_paf_range", "aasm_paf_to_text", "aasm_paf_save",
]


def _check(rc):
    if rc != AASM_OK:
        raise AlignasmError(rc, (LIB.aasm_last_error() or b"").decode(errors="replace"))


def set_host_threads(n):
    """Threads of the PAF reader / writers (0 = all); returns the previous setting."""
    return int(LIB.aasm_set_host_threads(int(n)))


def device_count():
    return int(LIB.aasm_device_count())


def sssp_dijkstra(g_voff, rowptr, col, w5, src, device=0):
    """dijkstra() of the reference's solver (k_shortest_walks.hpp:69-87) on the GPU over a batch of graphs.
    Returns (d: [V, 5] int64, prev: [V] int32)."""
    g_voff = np.ascontiguousarray(g_voff, np.int64); rowptr = np.ascontiguousarray(rowptr, np.int64)
    col = np.ascontiguousarray(col, np.int32); w5 = np.ascontiguousarray(w5, np.int64).reshape(-1); src = np.ascontiguousarray(src, np.int32)
    VT = int(g_voff[-1])
    d = np.zeros((VT, 5), np.int64)
    prev = np.zeros(VT, np.int32)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    _check(LIB.aasm_sssp_dijkstra(C.c_int64(len(g_voff) - 1), P(g_voff), P(rowptr), P(col), P(w5), P(src), P(d), P(prev), int(device)))
    return d, prev


def sssp_dial(g_voff, rowptr, col, cost, src, lim=2, device=0):
    """k_weighted_bfs() of the reference (Dial's bucketed BFS, k_weighted_bfs.hpp:16-37) on the GPU over a batch of digraphs.
    Returns (dist: [V] int64, -1 = unreachable; pre: [V] int64 local ids, -1 = none)."""
    g_voff = np.ascontiguousarray(g_voff, np.int64); rowptr = np.ascontiguousarray(rowptr, np.int64)
    col = np.ascontiguousarray(col, np.int32); cost = np.ascontiguousarray(cost, np.int32); src = np.ascontiguousarray(src, np.int32)
    VT = int(g_voff[-1])
    dist, pre = np.zeros(VT, np.int64), np.zeros(VT, np.int64)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    _check(LIB.aasm_sssp_dial(C.c_int64(len(g_voff) - 1), P(g_voff), P(rowptr), P(col), P(cost), P(src), int(lim), P(dist), P(pre), int(device)))
    return dist, pre


def debug_counter(name):
    """Process-wide diagnostic counter: "range_splits", "device_mallocs", "stream_syncs"."""
    return int(LIB.aasm_debug_counter(name.encode()))


def reserve_workspace(device=0, nbytes=0):
    """Create the device context and grow its workspace arena to `nbytes` ahead of the first solve (a fresh process
    otherwise pays HIP's start-up and the arena's hipMalloc calls inside it).  Returns the C-ABI code (0 = ok;
    AASM_E_NOMEM only means the solve will allocate for itself)."""
    return int(LIB.aasm_reserve_workspace(C.c_int(int(device)), C.c_int64(int(nbytes))))


def make_opts(max_paths=10000, non_skip_linkable=False, device=0, timing=False, keep_debug=False, sequential_select=False,
              test_max_contigs=0, test_inject_launch_failure=False, heap_waves="auto", enum_heap=False, wrap_devices=False, enum_small=False, sort_depth_test=0, heap_input_order=False, heap_block_waves=0, grid_order=False, test_dirty_scan=False, chain="auto", test_chain_lost=0, graph_launches=False, chain_own_queue=False, test_small_root_ring=False):
    o = Opts(int(max_paths), 1 if non_skip_linkable else 0, int(device), 1 if timing else 0, 1 if keep_debug else 0)
    # bit 0: force the one-wave-per-contig selection kernel; bits 1 / 2: K7 with several waves per contig for every contig / for none
    # bit 3: K8 with the d-ary heap queue instead of the sorted-front / sorted-runs queue (cross-check of the two)
    # bits 6 / 7: the chain class (aasm_k67_chain: a contig's sweep, pre-pass and heaps beside each other) for every sparse contig / for none
    o.reserved[0] = (1 if sequential_select else 0) | {"auto": 0, "all": 2, "none": 4}[heap_waves] | (8 if enum_heap else 0) | (16 if enum_small else 0) | (32 if grid_order else 0) | {"auto": 0, "all": 64, "none": 128, "half": 192}[chain] | ((1 << 8) if heap_input_order else (int(heap_block_waves) << 8)) | ((1 << 16) if graph_launches else 0)   # bit 16: rows, reversed CSR and sweep headers by the separate launches even where one workgroup per contig would do (aasm_k46_graph); bits 8-15: 1 = K7's several-waves class in input order instead of largest first; 4 / 8 / 16 = that many waves per contig of it
    o.reserved[1] = int(test_max_contigs)              # test hook: longer contig ranges "do not fit" (range split)
    o.reserved[2] = (1 if test_inject_launch_failure else 0) | (2 if wrap_devices else 0) | (4 if test_dirty_scan else 0) | (8 if test_chain_lost == 1 else 16 if test_chain_lost == 2 else 0) | (32 if chain_own_queue else 0) | (64 if test_small_root_ring else 0) | ((int(sort_depth_test) & 0xff) << 8)   # bit 1: shards wrap around the devices that exist
    return o


class Paf:
    """Parsed (or synthesised) PAF file held by the library."""

    def __init__(self, handle):
        self._h = handle

    @staticmethod
    def read(path, device_ranges=False):
        """device_ranges: leave the cs -> match-range conversion to the GPU (AASM_READ_DEVICE_RANGES)."""
        h = C.c_void_p()
        _check(LIB.aasm_paf_read_opts(os.fsencode(path), 1 if device_ranges else 0, C.byref(h)))
        return Paf(h)

    @staticmethod
    def parse(text: bytes, device_ranges=False):
        h = C.c_void_p()
        _check(LIB.aasm_paf_parse_mem_opts(text, C.c_int64(len(text)), 1 if device_ranges else 0, C.byref(h)))
        return Paf(h)

    @staticmethod
    def synth(n_contigs, recs_per_contig, seed, dense=False, heavy_tail=False, dup_every=0, shuffle=False, no_cs=False,
              first=0, count=None, records_only=False):
        """The synthetic file (n_contigs, ...), or its contigs [first, first + count); records_only: no match ranges / cs
        tags (enough for the contig cost model that cuts the file into shards)."""
        cfg = SynthCfg(n_contigs, recs_per_contig, seed, 1 if dense else 0, 1 if heavy_tail else 0, dup_every,
                       (1 if shuffle else 0) | (2 if no_cs else 0) | (4 if records_only else 0))
        h = C.c_void_p()
        if first == 0 and count is None:
            _check(LIB.aasm_synth_paf(C.byref(cfg), C.byref(h)))
        else:
            _check(LIB.aasm_synth_paf_range(C.byref(cfg), C.c_int64(int(first)), C.c_int64(int(n_contigs - first if count is None else count)), C.byref(h)))
        return Paf(h)

    def merge_alt(self, text: bytes, alt_baseline=0.5):
        """--alt: merge a second PAF (sub-contig re-alignments), alignasm.cpp:186-332."""
        _check(LIB.aasm_paf_merge_alt_mem(self._h, text, C.c_int64(len(text)), C.c_double(alt_baseline)))

    @property
    def n_contigs(self):
        return int(LIB.aasm_paf_n_contigs(self._h))

    def view(self) -> BatchIn:
        v = BatchIn()
        _check(LIB.aasm_paf_batch(self._h, C.byref(v)))
        return v

    def batch(self) -> HostBatch:
        return HostBatch.from_view(self.view())

    def to_text(self) -> bytes:
        p = C.c_void_p()
        n = C.c_int64()
        _check(LIB.aasm_paf_to_text(self._h, C.byref(p), C.byref(n)))
        try:
            return C.string_at(p, n.value)
        finally:
            C.CDLL(None).free(p)

    def save(self, path):
        """Write the batch as PAF text (rows with cs tags) to `path`."""
        _check(LIB.aasm_paf_save(self._h, os.fsencode(path)))

    def write_outputs(self, out: BatchOut, main_path, alt_path, all_path):
        _check(LIB.aasm_paf_write_outputs(self._h, C.byref(out), os.fsencode(main_path), os.fsencode(alt_path), os.fsencode(all_path)))

    def close(self):
        if self._h:
            LIB.aasm_paf_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def solve_batch_raw(view: BatchIn, opts: Opts) -> BatchOut:
    out = BatchOut()
    _check(LIB.aasm_solve_batch(C.byref(view), C.byref(opts), C.byref(out)))
    return out


def free_out(out: BatchOut):
    LIB.aasm_free_out(C.byref(out))


def solve_batch(batch, max_paths=10000, non_skip_linkable=False, device=0, timing=False, n_devices=1, sequential_select=False,
                test_max_contigs=0, test_inject_launch_failure=False, heap_waves="auto", enum_heap=False, wrap_devices=False, heap_input_order=False, test_dirty_scan=False, chain="auto", test_chain_lost=0, graph_launches=False, chain_own_queue=False, test_small_root_ring=False):
    """solve_ctg_read over a batch (HostBatch or Paf).  Returns a dict of numpy arrays."""
    view = batch.view if isinstance(batch, HostBatch) else batch.view()
    opts = make_opts(max_paths, non_skip_linkable, device, timing, False, sequential_select, test_max_contigs, test_inject_launch_failure, heap_waves, enum_heap, wrap_devices, heap_input_order=heap_input_order, test_dirty_scan=test_dirty_scan, chain=chain, test_chain_lost=test_chain_lost, graph_launches=graph_launches, chain_own_queue=chain_own_queue, test_small_root_ring=test_small_root_ring)
    if n_devices > 1:
        out = BatchOut()
        _check(LIB.aasm_solve_batch_multi(C.byref(view), C.byref(opts), int(n_devices), C.byref(out)))
    else:
        out = solve_batch_raw(view, opts)
    try:
        return unpack_out(out)
    finally:
        free_out(out)


class DeviceBatch:
    """A batch uploaded once to HBM; solve() can then be timed without PCIe traffic."""

    def __init__(self, batch, device=0):
        view = batch.view if isinstance(batch, HostBatch) else batch.view()
        self._keep = batch
        self.device = device
        self._up = C.c_void_p()
        self.dev_view = BatchIn()
        _check(LIB.aasm_upload_batch(C.byref(view), int(device), C.byref(self._up), C.byref(self.dev_view)))
        self.n_contigs = int(view.n_contigs)
        self.n_records = int(view.n_records)

    def solve(self, max_paths=10000, non_skip_linkable=False, timing=False, keep_debug=False, stream=None, heap_waves="auto", enum_heap=False, enum_small=False, sort_depth_test=0, heap_input_order=False, heap_block_waves=0, grid_order=False, chain="auto", graph_launches=False, chain_own_queue=False):
        res = C.c_void_p()
        opts = make_opts(max_paths, non_skip_linkable, self.device, timing, keep_debug, heap_waves=heap_waves, enum_heap=enum_heap, enum_small=enum_small, sort_depth_test=sort_depth_test, heap_input_order=heap_input_order, heap_block_waves=heap_block_waves, grid_order=grid_order, chain=chain, graph_launches=graph_launches, chain_own_queue=chain_own_queue)
        _check(LIB.aasm_solve_device(C.byref(self.dev_view), C.byref(opts), C.c_void_p(stream or 0), C.byref(res)))
        return DeviceResult(res)

    def close(self):
        if self._up:
            LIB.aasm_upload_free(self._up)
            self._up = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceResult:
    def __init__(self, handle):
        self._h = handle

    def stats(self):
        st = Stats()
        _check(LIB.aasm_result_stats(self._h, C.byref(st)))
        return st.as_dict()

    def fetch(self):
        out = self.fetch_raw()
        try:
            return unpack_out(out)
        finally:
            free_out(out)

    def fetch_raw(self) -> BatchOut:
        """D2H + ragged pack into the C structure (what a C caller gets); release it with free_out()."""
        out = BatchOut()
        _check(LIB.aasm_result_fetch(self._h, C.byref(out)))
        return out

    def debug(self, name, dtype):
        n = LIB.aasm_debug_fetch(self._h, name.encode(), None, C.c_int64(0))
        if n < 0:
            raise KeyError(name)
        buf = np.zeros(n // np.dtype(dtype).itemsize, dtype)
        LIB.aasm_debug_fetch(self._h, name.encode(), buf.ctypes.data_as(C.c_void_p), C.c_int64(buf.nbytes))
        return buf

    def close(self):
        if self._h:
            LIB.aasm_result_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
