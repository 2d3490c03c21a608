"""Seeded case generator for the cs-codec parity tests (tests/test_cs_ref.py, tests/golden/make_ref_cs.py).

A case = one PafReadData-level row (cs tag, strand, closed query / reference intervals as the reader
stores them: ref_str > ref_end on the '-' strand, alignasm.cpp:137-147) + a few clips of it.
Own code; holds no reference text.
"""
import random

LETTERS = "acgtn"

# tags the reference codec must reject or treat specially (paf_data.cpp:30-69, from_chars semantics)
DAMAGED = [
    "cs:Z::0", "cs:Z::-5", "cs:Z::", "cs:Z::12x", "cs:Z:*a", "cs:Z:*a1", "cs:Z:+", "cs:Z:-:4", "cs:Z:=ACGT", "cs:Z::5~gt12ag",
    "cs:Y::5", "cs:Z", "cs:Z::99999999999999999999", "cs:Z::9223372036854775808", "", "cs:Z:", "cs:Z::3+ac*t", "cs:Z::+5", "cs:Z:: 5",
    "cs:Z::5 ", "cs:Z:*ac*", "cs:Z:-", "cs:Z::5-", "cs:Z:5:5", "cs:z::5", "cs:Z::5\t", "cs:Z:*1a", "cs:Z:+a+", "cs:Z::1:", "cs:Z:::1",
]
# tags the reference accepts although they look odd: leading zeros (from_chars), upper case (isalpha), 19 digits
ODD_VALID = [
    ("cs:Z::007", 7, 7), ("cs:Z::00000000000000000000005*ac", 6, 6), ("cs:Z::3*AC:2", 6, 6), ("cs:Z::3+ACGT:2", 9, 5), ("cs:Z::3-NNn:2", 5, 8),
    ("cs:Z:*ag", 1, 1), ("cs:Z:+a", 1, 0), ("cs:Z:-a", 0, 1), ("cs:Z::1", 1, 1), ("cs:Z:*ac*gt*ca", 3, 3), ("cs:Z:+ac-gt+a:1", 4, 3),
    ("cs:Z::0000123456", 123456, 123456), ("cs:Z::1-a:1-c:1-g:1", 4, 7), ("cs:Z:*zz:1+xyz", 5, 2),
]


def random_tag(rng, n_ops, alternating=None):
    """(tag, query bases, reference bases).  alternating: match runs between edits (a real aligner's shape), else any order
    (adjacent ':' runs, edits back to back, edits first / last)."""
    if alternating is None:
        alternating = rng.random() < 0.5
    ops, q, r = [], 0, 0
    for i in range(n_ops):
        kind = (":" if i % 2 == 0 else rng.choice("*+-")) if alternating else rng.choice("::*+-")
        if kind == ":":
            n = rng.choice((1, 1, 2, 9, 10, 11, 99, 100, rng.randint(1, 40), rng.randint(1, 400), rng.randint(1000, 123456)))
            ops.append((":%d" if rng.random() < 0.97 else ":%03d") % n)
            q += n; r += n
        elif kind == "*":
            a, b = rng.choice(LETTERS), rng.choice(LETTERS)
            ops.append("*" + (a + b if rng.random() < 0.95 else (a + b).upper()))
            q += 1; r += 1
        else:
            n = rng.choice((1, 1, 2, 3, 5, rng.randint(1, 12), 64, 70, 129))
            s = "".join(rng.choice(LETTERS) for _ in range(n))
            ops.append(kind + (s if rng.random() < 0.95 else s.upper()))
            if kind == "+":
                q += n
            else:
                r += n
    return "cs:Z:" + "".join(ops), q, r


def mutate(rng, tag):
    cs = list(tag)
    lo = 0 if rng.random() < 0.1 else min(5, len(cs))
    how = rng.randrange(5)
    pos = rng.randrange(lo, len(cs)) if len(cs) > lo else 0
    if how == 0 and cs:
        cs[pos] = rng.choice(":*+-acgt0123456789?Z =~")
    elif how == 1 and cs:
        del cs[pos]
    elif how == 2:
        cs.insert(pos, rng.choice(":*+-a5"))
    elif how == 3:
        cs = cs[:pos]
    else:
        cs.insert(pos, rng.choice("0123456789") * rng.choice((1, 18, 19, 25)))
    return "".join(cs)


def consumed(tag):
    """Bases a WELL-FORMED prefix of the tag consumes (for damaged tags: any plausible coordinates)."""
    import re
    q = r = 0
    for op in re.findall(r":[0-9]{1,12}|\*[A-Za-z][A-Za-z]|[+-][A-Za-z]+", tag[5:]):
        if op[0] == ":":
            q += int(op[1:]); r += int(op[1:])
        elif op[0] == "*":
            q += 1; r += 1
        elif op[0] == "+":
            q += len(op) - 1
        else:
            r += len(op) - 1
    return max(q, 1), max(r, 1)


def rows(seed, n_valid, n_damaged):
    """Yields dicts {cs, fwd, qs, qe, rs, re}: every tag on both strands; one in eight with coordinates
    the tag does not consume (-> 'cs tag consumption does not match PAF coordinates')."""
    rng = random.Random(seed)
    tags = [random_tag(rng, rng.choice((1, 1, 2, 3, 5, rng.randint(1, 30), rng.randint(1, 90)))) for _ in range(n_valid)]
    tags += ODD_VALID
    bad = [(t,) + consumed(t) for t in DAMAGED]
    while len(bad) < n_damaged:
        t = mutate(rng, rng.choice(tags)[0])
        bad.append((t,) + consumed(t))
    for k, (cs, q, r) in enumerate(tags + bad):
        for fwd in (True, False):
            dq, dr = rng.choice(((0, 0),) * 7 + ((1, 0), (0, -1), (-1, 1)))
            ql, rl = max(q + dq, 1), max(r + dr, 1)
            qs = rng.choice((0, 1, 1000, 123456789))
            rs = rng.choice((0, 5, 50000, 249000000))
            row = {"cs": cs, "fwd": fwd, "qs": qs, "qe": qs + ql - 1}
            row["rs"], row["re"] = (rs, rs + rl - 1) if fwd else (rs + rl - 1, rs)
            yield row


def clips(rng, row, ranges, n):
    """n clips (e_qs, e_qe, e_rs, e_re) of an accepted row.  Reference coordinates are exact when the clip end sits in a
    match run (what K2 produces), otherwise arbitrary (-> the logic_error paths)."""
    qs, qe, fwd = row["qs"], row["qe"], row["fwd"]
    step = 1 if fwd else -1

    def ref_of(x):
        for ql, qr, rl in ranges:
            if ql <= x <= qr:
                return rl + (x - ql) * step
        return row["rs"]

    out = [(qs, qe, row["rs"], row["re"])]                          # uncut
    while len(out) < n:
        if ranges and rng.random() < 0.7:                           # both ends on matched bases
            i = rng.randrange(len(ranges)); j = rng.randrange(i, len(ranges))
            a = rng.randint(ranges[i][0], ranges[i][1])
            b = rng.randint(max(a, ranges[j][0]), ranges[j][1])
        else:
            a = rng.randint(qs, qe); b = rng.randint(a, qe)
        if rng.random() < 0.15:
            a = qs
        if rng.random() < 0.15:
            b = qe
        ers, ere = ref_of(a), ref_of(b)
        if rng.random() < 0.05:
            ere += step                                             # inconsistent reference span
        out.append((a, b, ers, ere))
    return out
