"""Regenerates tests/golden/ref_prefix.npz: vectors recorded from the REAL solve_ctg_read() prefix.

Needs oracle/_ref/libaasm_ref_prefix_mono.so (oracle/Makefile: /root/reference/src/paf_data.cpp:1-738 piped to g++
from where it lies + our epilogue; bump-allocator flavour, so that node address order == allocation order).
The file holds DATA only: the input batches (record coordinates + match ranges, from the seeded generators
tests/test_fuzz.py::make_batch and the product's synthetic-PAF generator) and, per contig, what the reference's
own statements computed from them - sorted order, part ids, every (i, j) cut, vertex ids, adjacency lists with
all weight fields, anom_dis[dest], d / best, both Kahn orders, every heap node and root, the k-walk distances
(all 10 000 for the batches marked full, the first 64 otherwise) - the arrays of aasm_testlib.PREFIX_NAMES.

    python tests/golden/make_ref_prefix.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import aasm_testlib as T   # noqa: E402
from test_fuzz import make_batch   # noqa: E402

KD = ("kd_qry", "kd_ref", "kd_anom", "kd_qnz", "kd_qtot")
KD_SHORT = 64

# (tag, maker, nsl, keep all 10 000 distances)
BATCHES = [
    ("fuzz0", lambda: make_batch(3, 6, 30, 400, 0), False, True),
    ("fuzz1", lambda: make_batch(4, 6, 30, 400, 1), False, True),
    ("fuzz2", lambda: make_batch(5, 6, 30, 400, 2), False, True),
    ("fuzz0n", lambda: make_batch(6, 6, 30, 400, 0), True, False),
    ("fuzz1n", lambda: make_batch(7, 6, 30, 400, 1), True, False),
    ("fuzz2n", lambda: make_batch(8, 6, 30, 400, 2), True, False),
    ("fuzz0b", lambda: make_batch(9, 8, 60, 400, 0), False, False),
    ("fuzz2b", lambda: make_batch(10, 8, 60, 400, 2), False, False),
    ("c1", lambda: T.synth(10, 100, 1), False, False),                         # BASELINE configs[0] in full
    ("c2one", lambda: T.synth(1, 1000, 11), False, True),                      # one contig of configs[1]'s size
    ("dense", lambda: T.synth(1, 300, 31, dense=True), False, True),
    ("densen", lambda: T.synth(2, 200, 31, dense=True), True, False),
    ("dup3", lambda: T.synth(2, 200, 5, dup_every=3), False, True),
    ("dupshuf", lambda: T.synth(3, 150, 9, dup_every=3, shuffle=True), False, False),
    ("nsl", lambda: T.synth(3, 200, 7), True, False),
    ("alldup", lambda: T.synth(4, 40, 13, dense=True, dup_every=1, shuffle=True), False, False),
    ("ragged", lambda: T.synth(8, 40, 10, dense=True, shuffle=True, heavy_tail=True), False, False),
]


def main():
    assert T.ref_prefix(True) is not None, "build oracle/_ref first (make -C oracle)"
    out = {}
    tags = []
    for tag, mk, nsl, full in BATCHES:
        hb = mk()
        tags.append(tag)
        out[f"{tag}/nsl"] = np.array([1 if nsl else 0], np.int8)
        out[f"{tag}/full"] = np.array([1 if full else 0], np.int8)
        for k, a in hb.arrays.items():
            if k == "rng_qry_r":
                out[f"{tag}/in/{k}~len"] = (a - hb.arrays["rng_qry_l"]).astype(np.int32)           # storage only: r - l
                continue
            if k.startswith("rng_"):
                out[f"{tag}/in/{k}~d"] = np.diff(a.astype(np.int64), prepend=0).astype(np.int32)   # storage only: first differences
                continue
            small = a
            if a.dtype == np.int64 and a.size and np.abs(a).max() < 2 ** 31:
                small = a.astype(np.int32)                                     # storage only; the loader widens again
            out[f"{tag}/in/{k}"] = small
        off = hb.arrays["ctg_rec_off"]
        for c in range(len(off) - 1):
            if off[c + 1] - off[c] <= 1:
                continue
            r = T.ref_prefix_debug(hb, c, nsl=nsl, names=T.PREFIX_NAMES)
            o = T.oracle_debug(hb, c, 10000, nsl)
            for n in T.PREFIX_NAMES:
                assert np.array_equal(r[n], o[n]), (tag, c, n)                 # (not needed for the record; a generator-side sanity check)
                a = r[n]
                if n in KD and not full:
                    a = a[:KD_SHORT]
                if n == "heap_right":                           # storage only: distance back to the child (0 = none)
                    out[f"{tag}/c{c}/{n}~b"] = np.where(a >= 0, np.arange(len(a)) - a, 0).astype(np.int32)
                    continue
                if a.size and np.abs(a).max() < 2 ** 31:
                    a = a.astype(np.int32)
                out[f"{tag}/c{c}/{n}"] = a
            out[f"{tag}/c{c}/kfound"] = np.array([len(r["kd_qry"])], np.int32)
    out["tags"] = np.array(tags)
    out["source"] = np.array(["reference paf_data.cpp:223-738 via oracle/_ref/libaasm_ref_prefix_mono.so (MAX_PATH_COUNT = 10000)"])
    path = os.path.join(HERE, "ref_prefix.npz")
    np.savez_compressed(path, **out)
    print("wrote %s: %d batches, %d arrays, %d bytes" % (path, len(tags), len(out), os.path.getsize(path)))


if __name__ == "__main__":
    main()
