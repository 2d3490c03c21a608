"""Static per-contig partition used when one job spans several GPUs.

Contigs are independent (the reference runs one TBB task per contig,
/root/reference/src/alignasm.cpp:351-359), so sharding needs no collective on the data
path: every rank solves its own contiguous block of contigs and the outputs are
concatenated in contig order.  The same cost model is used by the C library
(aasm_solve_batch_multi) for the one-process / many-devices case.
"""
import numpy as np


def contig_costs(ctg_rec_off):
    n = np.diff(np.asarray(ctg_rec_off, dtype=np.int64)).astype(np.float64)
    return n + 16.0          # per-contig chains dominate: cost ~ records + fixed term


def partition_contigs(ctg_rec_off, n_shards):
    """Cut points [c_0=0, c_1, ..., c_n=C] of a contiguous, cost-balanced partition."""
    cost = contig_costs(ctg_rec_off)
    C = len(cost)
    n_shards = max(1, min(int(n_shards), C))
    pre = np.concatenate([[0.0], np.cumsum(cost)])
    cuts = [0]
    for d in range(1, n_shards):
        c = int(np.searchsorted(pre, pre[-1] * d / n_shards, side="left"))
        c = max(c, cuts[-1] + 1)
        c = min(c, C - (n_shards - d))
        cuts.append(c)
    cuts.append(C)
    return cuts


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
