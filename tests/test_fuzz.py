"""Differential fuzz: batches that do NOT come from the synthetic-PAF generator -- random
intervals with containment, equal starts, duplicates, both strands, several reference names,
records of a few bases, match ranges with unmatched bases in between -- solved by the kernel
bodies (host emulation; HIP in the gpu tier) and by the oracle.  The two formulate the graph
construction differently (closed-form rows over a sparse slot table vs the literal loops), so
agreement on adversarial inputs is not self-fulfilling."""
import numpy as np
import pytest

from alignasm_amd._abi import HostBatch


def make_batch(seed, n_contigs, n_max, L, style):
    rng = np.random.default_rng(seed)
    A = {k: [] for k in ("qry_str", "qry_end", "ref_str", "ref_end", "qry_total", "ref_chr", "aln_fwd", "map_qul", "rng_qry_l", "rng_qry_r", "rng_ref_l")}
    coff, roff = [0], [0]
    for _ in range(n_contigs):
        n = int(rng.integers(1, n_max + 1))
        qt = L + int(rng.integers(0, 50))
        for i in range(n):
            if style == 0:                               # anything goes
                qs = int(rng.integers(0, L - 2)); qe = min(L - 1, qs + int(rng.integers(1, max(2, L // 3))))
            elif style == 1:                             # few distinct boundaries: duplicates, containment, equal starts
                qs = int(rng.integers(0, 6)) * (L // 8); qe = min(L - 1, qs + int(rng.integers(1, 4)) * (L // 8))
            else:                                        # a chain with overlaps
                qs = min(L - 3, i * (L // (n + 1)) + int(rng.integers(0, 5))); qe = min(L - 1, qs + int(rng.integers(L // (n + 1), 2 * L // (n + 1) + 2)))
            if qe <= qs:
                qe = qs + 1
            fwd = int(rng.integers(0, 2)); rlo = int(rng.integers(0, 100000)); span = qe - qs
            rs, re = (rlo, rlo + span) if fwd else (rlo + span, rlo)
            for k, v in (("qry_str", qs), ("qry_end", qe), ("ref_str", rs), ("ref_end", re), ("qry_total", qt),
                         ("ref_chr", int(rng.integers(0, 3))), ("aln_fwd", fwd), ("map_qul", int(rng.choice([0, 0, 10, 60])))):
                A[k].append(v)
            k = int(rng.integers(1, 4))
            cuts = sorted(set(int(x) for x in rng.integers(qs, qe + 1, size=k - 1))) if k > 1 else []
            pts = [qs] + [x for x in cuts if qs < x <= qe] + [qe + 1]
            step = 1 if fwd else -1
            for a, b in zip(pts[:-1], pts[1:]):
                l, r = a, (b - 1 if b == qe + 1 else b - 2)   # base b-1 stays unmatched (a substitution) between two ranges
                if r >= l:
                    A["rng_qry_l"].append(l); A["rng_qry_r"].append(r); A["rng_ref_l"].append(rs + (l - qs) * step)
            roff.append(len(A["rng_qry_l"]))
        coff.append(len(A["qry_str"]))
    A["ctg_rec_off"], A["rec_rng_off"] = coff, roff
    return HostBatch(A)


def _run(T, solve, seeds, n_max):
    seen = dict(paths=0, alt=0, allp=0)
    for seed in seeds:
        for style in (0, 1, 2):
            hb = make_batch(seed, 6, n_max, 400, style)
            for K, nsl in ((10000, False), (3, True)):
                want = T.oracle_solve(hb, K, nsl)
                got = solve(hb, K, nsl)
                assert T.diff_outputs(want, got) == [], (seed, style, K, nsl)
                seen["paths"] += want["stats"]["n_paths_found"]; seen["alt"] += len(want["alt"]); seen["allp"] += len(want["all"])
    assert seen["paths"] > 1000 and seen["alt"] > 0 and seen["allp"] > 0      # the inputs do reach K8 / K9's alt and tie logic


def test_fuzz_emulation_vs_oracle(T):
    _run(T, lambda hb, K, nsl: T.emul_solve(hb, K, nsl), range(24), 25)


def test_fuzz_emulation_vs_oracle_longer_contigs(T):
    _run(T, lambda hb, K, nsl: T.emul_solve(hb, K, nsl), range(500, 506), 60)


@pytest.mark.gpu
def test_fuzz_hip_vs_oracle(T):
    api = T.api()
    _run(T, lambda hb, K, nsl: api.solve_batch(hb, max_paths=K, non_skip_linkable=nsl), range(100, 130), 30)


@pytest.mark.gpu
def test_fuzz_hip_vs_oracle_many_contigs(T):
    """Thousands of small adversarial contigs per batch: above 2 560 contigs the sweeps run two contigs per wave
    (32 lanes each), which the six-contig batches above never reach."""
    api = T.api()
    for seed in (7, 8):
        for style in (0, 1, 2):
            hb = make_batch(seed, 2700, 14, 300, style)
            for K, nsl in ((10000, False), (3, True)):
                want = T.oracle_solve(hb, K, nsl)
                got = api.solve_batch(hb, max_paths=K, non_skip_linkable=nsl)
                assert T.diff_outputs(want, got) == [], (seed, style, K, nsl)
