"""Static per-contig partition used when one job spans several GPUs.

Contigs are independent (the reference runs one TBB task per contig over ONE input,
/root/reference/src/alignasm.cpp:346-361), so sharding needs no collective on the data
path: every rank solves its own contiguous block of contigs and the outputs are
concatenated in contig order.  The cost model and the cut are the C library's
(aasm_contig_costs / aasm_partition_contigs, csrc/aasm_shard.cpp), the same ones
aasm_solve_batch_multi uses for the one-process / many-devices case.
"""
import ctypes as C

import numpy as np

from ._abi import HostBatch
from .api import LIB, _check


def _view(batch):
    return batch.view if isinstance(batch, HostBatch) else batch.view()


def contig_costs(batch):
    """Estimated GPU cost per contig: records + a graph-density term from the part sizes."""
    view = _view(batch)
    cost = np.zeros(int(view.n_contigs), np.float64)
    _check(LIB.aasm_contig_costs(C.byref(view), cost.ctypes.data_as(C.c_void_p)))
    return cost


def partition_contigs(batch, n_shards):
    """Cut points [c_0=0, c_1, ..., c_n=C] of the contiguous, cost-balanced partition."""
    view = _view(batch)
    n_shards = max(1, min(int(n_shards), int(view.n_contigs)))
    cuts = np.zeros(n_shards + 1, np.int64)
    _check(LIB.aasm_partition_contigs(C.byref(view), n_shards, cuts.ctypes.data_as(C.c_void_p)))
    return [int(x) for x in cuts]


def partition_costs(cost, n_shards):
    """The same cut for caller-supplied per-contig costs."""
    cost = np.ascontiguousarray(cost, np.float64)
    n_shards = max(1, min(int(n_shards), len(cost)))
    cuts = np.zeros(n_shards + 1, np.int64)
    _check(LIB.aasm_partition_costs(cost.ctypes.data_as(C.c_void_p), C.c_int64(len(cost)), n_shards, cuts.ctypes.data_as(C.c_void_p)))
    return [int(x) for x in cuts]


def concat_outputs(parts):
    """Concatenate per-shard result dicts (alignasm_amd._abi.unpack_out) in contig order."""
    out = {"n_contigs": sum(p["n_contigs"] for p in parts)}
    for off_key, elem_key in (("main_off", "main"), ("alt_off", "alt")):
        offs, base = [np.zeros(1, np.int64)], 0
        for p in parts:
            offs.append(p[off_key][1:] + base)
            base += int(p[off_key][-1])
        out[off_key] = np.concatenate(offs)
        out[elem_key] = np.concatenate([p[elem_key] for p in parts])
    poffs, eoffs, pbase, ebase = [np.zeros(1, np.int64)], [np.zeros(1, np.int64)], 0, 0
    for p in parts:
        poffs.append(p["all_path_off"][1:] + pbase)
        eoffs.append(p["all_elem_off"][1:] + ebase)
        pbase += int(p["all_path_off"][-1])
        ebase += int(p["all_elem_off"][-1])
    out["all_path_off"] = np.concatenate(poffs)
    out["all_elem_off"] = np.concatenate(eoffs)
    out["all"] = np.concatenate([p["all"] for p in parts])
    out["status"] = np.concatenate([p["status"] for p in parts])
    return out
