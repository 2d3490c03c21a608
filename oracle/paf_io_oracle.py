"""oracle/paf_io_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

File-level CPU restatement of the reference's I/O side of the hot path, in plain Python
(small inputs only), independent of alignasm_amd/csrc/aasm_paf.cpp: nothing here calls,
imports or shares code with the product.  Used only by tests/, tests/golden/make_golden.py
and (never) by the product.  All citations are into /root/reference/src.

  parse_short_cs, get_overlap_range ........ paf_data.cpp:29-72, 90-123
  get_edited_paf_data ...................... paf_data.cpp:125-220
  PAF reader ............................... alignasm.cpp:76-183
  --alt second-PAF merge ................... alignasm.cpp:186-332
  process_output / process_max_output ...... alignasm.cpp:398-490

PINNING STATUS: the cs codec part (parse_short_cs, get_overlap_range, get_edited_paf_data) is PINNED to the
reference's own code since round 3: oracle/_ref/libaasm_ref_cs.so is built from the head of paf_data.cpp and
tests/test_cs_ref.py diffs this file against it live (6 000 tags, 17 000 clips, every exception text) and against the
recorded vectors tests/golden/ref_cs.json.gz.  The rest - reader framing, --alt merge, writers - stays "parity
unpinned": alignasm.cpp cannot be compiled in this image (argparse, csv-parser, indicators and oneTBB are absent, no
stand-ins are written) and the reference ships no fixtures, so that part is a second, independent reading of the source
text.  Its value: the product's reader and writers (C++, single-pass scanners) are diffed against a formulation that keeps
the reference's own structure (operation list -> filter -> reverse -> render), so a misreading would have to be made
twice, in two different shapes, to go unnoticed.

csv-parser behaviour that is assumed, not restated (parity unpinned): rows are
split on TAB, one row per '\\n'-terminated line, empty lines skipped, fields containing TAB,
quote or newline (never present in PAF) would be quoted by make_tsv_writer.
"""
import re

TYPE_MAIN, TYPE_ALT = 0, 1                      # paf_data.hpp:32-33
CS_TAG_START = 5                                # paf_data.hpp:25


class CsError(ValueError):
    """std::invalid_argument thrown by the cs codec (paf_data.cpp:31,46,52,63,66,121)."""


class CsLogicError(RuntimeError):
    """std::logic_error thrown by get_edited_paf_data (paf_data.cpp:160,217)."""


class PafRow:
    """PafReadData (paf_data.hpp:51-67)."""
    __slots__ = ("paf_index", "ctg_index", "cs_string", "mat_num", "aln_len", "ref_rng", "qry_rng", "ref_total", "qry_total",
                 "qry_str", "qry_end", "ref_str", "ref_end", "ref_chr", "map_qul", "aln_fwd", "cord")


_ALPHA = re.compile(r"[A-Za-z]*")               # std::isalpha in the "C" locale (paf_data.cpp:25-27)
_DIGITS = re.compile(r"[0-9]+")


def parse_short_cs(cs):
    """paf_data.cpp:29-72 -> list of (type, length, text)."""
    if not cs.startswith("cs:Z:"):                                              # :30-32
        raise CsError("PAF record does not contain a short-form cs:Z tag")
    ops, pos, n = [], CS_TAG_START, len(cs)
    while pos < n:                                                              # :36
        start = pos
        t = cs[pos]
        pos += 1
        if t == ":":                                                            # :41-49  std::from_chars<int64_t>
            m = _DIGITS.match(cs, pos)                                          # (no sign, no leading blanks accepted)
            if m is None or int(m.group()) <= 0 or int(m.group()) >= 1 << 63:
                raise CsError("Invalid :length operation in cs tag")
            length, pos = int(m.group()), m.end()
        elif t == "*":                                                          # :50-56
            if pos + 2 > n or not cs[pos].isascii() or not cs[pos].isalpha() or not cs[pos + 1].isascii() or not cs[pos + 1].isalpha():
                raise CsError("Invalid substitution operation in cs tag")
            pos += 2
            length = 1
        elif t in "+-":                                                         # :57-64
            m = _ALPHA.match(cs, pos)
            length = m.end() - pos
            pos = m.end()
            if length == 0:
                raise CsError("Empty indel operation in cs tag")
        else:                                                                   # :65-67
            raise CsError("Unsupported operation in short-form cs tag")
        ops.append((t, length, cs[start:pos]))
    return ops


def _query_oriented(ops, aln_fwd):              # paf_data.cpp:75-86
    return ops if aln_fwd else ops[::-1]


def get_overlap_range(r, cs):
    """paf_data.cpp:90-123: fills r.qry_rng / r.ref_rng (lists of closed [l, r] pairs)."""
    ops = parse_short_cs(cs)
    step = 1 if r.aln_fwd else -1                                               # :92
    ref_i, qry_i = r.ref_str, r.qry_str
    r.ref_rng, r.qry_rng = [], []
    for t, length, _ in _query_oriented(ops, r.aln_fwd):
        if t == ":":                                                            # :101-107
            r.ref_rng.append((ref_i, ref_i + (length - 1) * step))
            r.qry_rng.append((qry_i, qry_i + length - 1))
            ref_i += length * step
            qry_i += length
        elif t == "+":
            qry_i += length
        elif t == "-":
            ref_i += length * step
        else:                                                                   # '*', :112-116
            ref_i += step
            qry_i += 1
    if qry_i != r.qry_end + 1 or ref_i != r.ref_end + step:                     # :119-122
        raise CsError("cs tag consumption does not match PAF coordinates")


def get_edited_paf_data(e_qs, e_qe, e_rs, e_re, r):
    """paf_data.cpp:125-220 -> (cs string, mat_num, aln_len, is_cut)."""
    is_cut = e_qs != r.qry_str or e_qe != r.qry_end                             # :131-132
    if not is_cut:
        return r.cs_string, r.mat_num, r.aln_len, False                         # :133-136
    kept, q = [], r.qry_str
    for t, length, text in _query_oriented(parse_short_cs(r.cs_string), r.aln_fwd):
        if t == ":":                                                            # :144-152
            lo, hi = max(q, e_qs), min(q + length - 1, e_qe)
            if lo <= hi:
                kept.append((":", hi - lo + 1, None))
            q += length
        elif t == "+":                                                          # :153-164
            end = q + length - 1
            if q <= e_qe and e_qs <= end:
                if q < e_qs or e_qe < end:
                    raise CsLogicError("Alignment was clipped inside a cs insertion")
                kept.append((t, length, text))
            q += length
        elif t == "*":                                                          # :165-170
            if e_qs <= q <= e_qe:
                kept.append((t, length, text))
            q += 1
        else:                                                                   # '-', :171-177
            if e_qs < q <= e_qe:
                kept.append((t, length, text))
    if not r.aln_fwd:                                                           # :180-182
        kept.reverse()
    out, mat, aln, qb, rb = ["cs:Z:"], 0, 0, 0, 0                               # :184-207
    for t, length, text in kept:
        if t == ":":
            out.append(":%d" % length)
            mat += length; aln += length; qb += length; rb += length
        else:
            out.append(text)
            if t == "+":
                qb += length; aln += length
            elif t == "-":
                rb += length; aln += length
            else:
                qb += 1; rb += 1; aln += 1
    if qb != e_qe - e_qs + 1 or rb != abs(e_re - e_rs) + 1:                     # :209-218
        raise CsLogicError("Edited cs tag does not match edited PAF coordinates")
    return "".join(out), mat, aln, True


class PafFile:
    """State of main() between the reader and the writers (alignasm.cpp:86-99)."""

    def __init__(self):
        self.chr_map, self.chr_rev, self.paf_map = {}, {}, {}
        self.paf_data, self.ctg_names = [], []


def _find_cs_tag(f):                            # alignasm.cpp:100-108
    for x in f[12:]:
        if x.startswith("cs:Z:"):
            return x
    return ""


def _rows(text):
    if isinstance(text, bytes):
        text = text.decode()
    return [ln.split("\t") for ln in text.split("\n") if ln != ""]


def _fill_common(r, f, st):                     # the field conversions shared by :135-176 and :271-303
    r.ref_total = int(f[6])
    r.ref_str, r.ref_end = int(f[7]), int(f[8]) - 1                             # 0-based closed (:147-151)
    if f[5] not in st.chr_map:                                                  # :119-123 / :262-266
        st.chr_map[f[5]] = len(st.chr_map)
        st.chr_rev[st.chr_map[f[5]]] = f[5]
    r.ref_chr = st.chr_map[f[5]]
    r.aln_fwd = f[4][0] == "+"
    if not r.aln_fwd:                                                           # :155-159
        r.ref_str, r.ref_end = r.ref_end, r.ref_str
    r.map_qul = int(f[11])
    r.mat_num, r.aln_len = int(f[9]), int(f[10])


def read_paf(text):
    """alignasm.cpp:110-183.  Raises SystemExit(1)-style ValueError on a missing cs tag (:165-168)."""
    st = PafFile()
    ctg, cur, paf_index = "", [], 0
    for row_global, f in enumerate(_rows(text)):
        qry_chr = f[0]
        if ctg == "":
            ctg = qry_chr
        if f[5] not in st.chr_map:
            st.chr_map[f[5]] = len(st.chr_map)
            st.chr_rev[st.chr_map[f[5]]] = f[5]
        if ctg != qry_chr:                                                      # :125-133
            st.paf_data.append(cur); st.ctg_names.append(ctg)
            ctg, cur = qry_chr, []
            paf_index += 1
        r = PafRow()
        st.paf_map[qry_chr] = paf_index                                         # :136
        r.paf_index, r.ctg_index = paf_index, len(cur)
        r.qry_total = int(f[1])
        r.qry_str, r.qry_end = int(f[2]), int(f[3]) - 1                         # :141-145
        _fill_common(r, f, st)
        cs = _find_cs_tag(f)
        if cs == "":
            raise ValueError("Missing cs:Z tag in PAF record for query '%s'" % qry_chr)
        r.cs_string = cs
        r.cord = (TYPE_MAIN, row_global)                                        # :172
        get_overlap_range(r, cs)                                                # :174
        cur.append(r)
    st.ctg_names.append(ctg); st.paf_data.append(cur)                           # :180-181
    return st


def _parse_piece_name(s):                       # parseString, alignasm.cpp:209-233
    pos = s.find(":")
    if pos < 0:
        raise ValueError("Invalid input string format")
    end = s.find("-", pos + 1)
    if end < 0:
        end = len(s)
    m = re.match(r"-?[0-9]+", s[pos + 1:end])                                   # std::from_chars<int64_t> on [start, end)
    if m is None:
        raise ValueError("Error parsing number")
    return s[:pos], int(m.group()) - 1


def merge_alt(st, alt_text, alt_baseline):
    """alignasm.cpp:203-332 (after the extension / empty-file checks of :186-201)."""
    grp = None                                   # (tar_qry_offset, tar_real_qry_chr)
    tar_flag, tar_ratio, best = False, 0.0, None

    def flush():                                 # :244-252
        if grp is None or tar_flag:
            return
        if best is None:
            raise NotImplementedError("group without any aln_ratio > 0: the reference appends a default PafReadData")
        tgt = st.paf_data[st.paf_map.setdefault(grp[1], 0)]
        best.ctg_index = len(tgt)
        tgt.append(best)
    for row_global, f in enumerate(_rows(alt_text)):
        if f[5] not in st.chr_map:
            st.chr_map[f[5]] = len(st.chr_map)
            st.chr_rev[st.chr_map[f[5]]] = f[5]
        real, offset = _parse_piece_name(f[0])                                  # :268
        last = st.paf_data[st.paf_map.setdefault(real, 0)][-1]                  # :269 (operator[] inserts 0 for unknown names)
        r = PafRow()
        r.paf_index, r.ctg_index = last.paf_index, 0
        r.qry_total = last.qry_total                                            # :274
        r.qry_str, r.qry_end = int(f[2]) + offset, int(f[3]) + offset - 1       # :275-277
        _fill_common(r, f, st)
        cs = _find_cs_tag(f)
        if cs == "":
            raise ValueError("Missing cs:Z tag in alternative PAF record for query '%s'" % f[0])
        r.cs_string = cs
        r.cord = (TYPE_ALT, row_global)                                         # :302
        get_overlap_range(r, cs)
        if grp is None or grp != (offset, real):                                # :305-314
            flush()
            grp, tar_flag, tar_ratio, best = (offset, real), False, 0.0, None
        ratio = float(f[10]) / float(f[1])                                      # :316
        if ratio > tar_ratio:                                                   # :318-321 (a COPY: ctg_index is set on the copy pushed later)
            tar_ratio = ratio
            best = PafRow()
            for k in PafRow.__slots__:
                setattr(best, k, getattr(r, k))
        if ratio > alt_baseline:                                                # :323-327
            tgt = st.paf_data[st.paf_map.setdefault(real, 0)]
            r.ctg_index = len(tgt)
            tgt.append(r)
            tar_flag = True
    flush()                                                                     # :331


def to_arrays(st):
    """The solver-relevant fields of every record as the flat batch of include/alignasm_amd.h
    (plain Python lists / dict; the caller wraps them into numpy for oracle_solve_batch)."""
    a = {k: [] for k in ("qry_str", "qry_end", "ref_str", "ref_end", "qry_total", "ref_chr", "aln_fwd", "map_qul", "rng_qry_l", "rng_qry_r", "rng_ref_l")}
    ctg_off, rng_off = [0], [0]
    for recs in st.paf_data:
        for r in recs:
            a["qry_str"].append(r.qry_str); a["qry_end"].append(r.qry_end); a["ref_str"].append(r.ref_str); a["ref_end"].append(r.ref_end)
            a["qry_total"].append(r.qry_total); a["ref_chr"].append(r.ref_chr); a["aln_fwd"].append(1 if r.aln_fwd else 0); a["map_qul"].append(r.map_qul)
            for (ql, qr), (rl, _) in zip(r.qry_rng, r.ref_rng):
                a["rng_qry_l"].append(ql); a["rng_qry_r"].append(qr); a["rng_ref_l"].append(rl)
            rng_off.append(len(a["rng_qry_l"]))
        ctg_off.append(len(a["qry_str"]))
    a["ctg_rec_off"], a["rec_rng_off"] = ctg_off, rng_off
    return a


def _line(st, i, name, e):                      # the 15 fields of alignasm.cpp:426-440 / 467-481
    r = st.paf_data[i][int(e["ctg_index"])]
    qs, qe, rs, re_ = int(e["qs"]), int(e["qe"]), int(e["rs"]), int(e["re"])
    cs, mat, aln, _ = get_edited_paf_data(qs, qe, rs, re_, r)
    xi = "xi:Z:" + ("P_" if r.cord[0] == TYPE_MAIN else "A_") + str(r.cord[1])   # :398-405
    return "\t".join([name, str(r.qry_total), str(qs), str(qe + 1), "+" if r.aln_fwd else "-", st.chr_rev[r.ref_chr], str(r.ref_total),
                      str(rs if r.aln_fwd else re_), str((re_ if r.aln_fwd else rs) + 1), str(mat), str(aln), str(r.map_qul),
                      "tp:A:S" if int(e["is_alt"]) else "tp:A:P", xi, cs]) + "\n"


def render_outputs(st, sol):
    """process_output x2 + process_max_output (alignasm.cpp:407-490) -> (main, alt, all) bytes.
    `sol`: dict with main_off/main, alt_off/alt, all_path_off/all_elem_off/all (structured arrays
    or lists of dicts with keys qs, qe, rs, re, ctg_index, is_alt)."""
    main, alt, allp = [], [], []
    for i, name in enumerate(st.ctg_names):
        for e in sol["main"][int(sol["main_off"][i]):int(sol["main_off"][i + 1])]:
            main.append(_line(st, i, name, e))
        for e in sol["alt"][int(sol["alt_off"][i]):int(sol["alt_off"][i + 1])]:
            alt.append(_line(st, i, name, e))
        cnt = 0
        for p in range(int(sol["all_path_off"][i]), int(sol["all_path_off"][i + 1])):
            cnt += 1                                                            # :456
            for e in sol["all"][int(sol["all_elem_off"][p]):int(sol["all_elem_off"][p + 1])]:
                allp.append(_line(st, i, name + "." + str(cnt), e))             # :467
    return "".join(main).encode(), "".join(alt).encode(), "".join(allp).encode()
