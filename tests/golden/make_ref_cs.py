"""Regenerates tests/golden/ref_cs.json.gz: vectors recorded from the REAL reference cs codec.

Needs oracle/_ref/libaasm_ref_cs.so (oracle/Makefile builds it from /root/reference/src/paf_data.cpp:15-220
where that tree exists).  The file holds DATA only: rows (cs tag, strand, closed coordinates) from the seeded
generator tests/cs_cases.py, and for each what get_overlap_range returned (ranges, or the exception text) and
what get_edited_paf_data returned for a few clips (edited cs / mat_num / aln_len / is_cut, or the exception text).

    python tests/golden/make_ref_cs.py
"""
import gzip
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import aasm_testlib as T   # noqa: E402
import cs_cases as G       # noqa: E402

SEED, N_VALID, N_DAMAGED, N_CLIPS = 7, 520, 160, 3


def main():
    assert T.ref_cs() is not None, "build oracle/_ref first (make -C oracle)"
    rng = random.Random(SEED + 1)
    cases = []
    for row in G.rows(SEED, N_VALID, N_DAMAGED):
        r = T.ref_cs_ranges(row)
        case = dict(row)
        if r[0] == "err":
            case["err"] = [r[1], r[2]]
        else:
            case["ranges"] = [list(x) for x in r[1]]
            case["clips"] = []
            for clip in G.clips(rng, row, [(a, b, c) for a, b, c, _ in r[1]], N_CLIPS):
                e = T.ref_cs_edit(row, clip, 7, 9)
                case["clips"].append({"clip": list(clip), "err": [e[1], e[2]]} if e[0] == "err" else
                                     {"clip": list(clip), "cs": e[1], "mat": e[2], "aln": e[3], "cut": e[4]})
        cases.append(case)
    blob = json.dumps({"source": "reference paf_data.cpp:15-220 via oracle/_ref/libaasm_ref_cs.so", "generator": "tests/cs_cases.py",
                       "seed": SEED, "mat_num_in": 7, "aln_len_in": 9, "cases": cases}, separators=(",", ":")).encode()
    path = os.path.join(HERE, "ref_cs.json.gz")
    with open(path, "wb") as f:
        with gzip.GzipFile(fileobj=f, mode="wb", mtime=0) as g:
            g.write(blob)
    n_err = sum(1 for c in cases if "err" in c)
    print("wrote %s: %d rows (%d rejected), %d clips, %d bytes" % (path, len(cases), n_err, sum(len(c.get("clips", ())) for c in cases), os.path.getsize(path)))


if __name__ == "__main__":
    main()
