// aasm_paf.cpp -- host codec: PAF reader, short-form cs:Z codec, 15-column writers.
//
// Own implementation of the behaviour of the reference's host I/O around the hot path
// (SURVEY.md 8(f) row f1), needed so `alignasm <input.paf>` and the three output files
// stay drop-in:
//   reader ................ src/alignasm.cpp:76-183
//   cs tokenizer .......... src/paf_data.cpp:29-72
//   get_overlap_range ..... src/paf_data.cpp:90-123
//   get_edited_paf_data ... src/paf_data.cpp:125-220
//   writers ............... src/alignasm.cpp:398-490 (field list: SURVEY.md Appendix D)
// The reference reads/writes through csv-parser 2.2.1; its quoting of fields that hold a
// tab/quote/newline is not reproduced (parity-unpinned, irrelevant for PAF).
#include "aasm_paf.hpp"

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string_view>
#include <atomic>
#include <functional>
#include <condition_variable>
#include <mutex>
#include <memory>
#include <chrono>
#include <thread>
#include <unordered_map>
#include <climits>
#include <sys/stat.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

namespace aasm {

static thread_local std::string g_last_error;
void set_last_error(const std::string &msg) { g_last_error = msg; }
const char *last_error_cstr() { return g_last_error.c_str(); }

// ---- short-form cs:Z codec ----------------------------------------------------------------
// One scanner for everything the reference does with a cs tag (tokenizer paf_data.cpp:29-72,
// get_overlap_range :90-123, get_edited_paf_data :125-220).  cs_scan walks the text ONCE, first
// operation to last, and hands every operation {type, payload length, text span} to a visitor;
// no operation list is built.  A '-' strand record is walked in the same text order with the
// query cursor running DOWN from qry_end + 1 (the reference walks its operation list backwards
// with the cursor running up): an operation then covers [cursor - len, cursor - 1], match ranges
// come out last to first, and the operations a re-cut keeps come out already in output order
// (the reference reverses its kept list, :180-182).
// Error texts are the reference's (drop-in); the order of detection is too: a tokenizer error
// anywhere in the tag wins over what the walk finds (the reference tokenizes the whole tag first).
enum CsErr { CS_OK = 0, CS_E_TAG, CS_E_LENGTH, CS_E_SUBST, CS_E_INDEL, CS_E_OP, CS_E_CONSUME, CS_E_INS_CLIP, CS_E_EDIT };
static const char *cs_err_text(CsErr e) {
    switch (e) {
        case CS_E_TAG: return "PAF record does not contain a short-form cs:Z tag";          // :31
        case CS_E_LENGTH: return "Invalid :length operation in cs tag";                      // :46
        case CS_E_SUBST: return "Invalid substitution operation in cs tag";                  // :52
        case CS_E_INDEL: return "Empty indel operation in cs tag";                           // :63
        case CS_E_OP: return "Unsupported operation in short-form cs tag";                   // :66
        case CS_E_CONSUME: return "cs tag consumption does not match PAF coordinates";       // :121
        case CS_E_INS_CLIP: return "Alignment was clipped inside a cs insertion";            // :160
        case CS_E_EDIT: return "Edited cs tag does not match edited PAF coordinates";        // :217
        default: return "";
    }
}
static inline bool alpha_ascii(char c) { return (unsigned)(((unsigned char)c | 32) - 'a') < 26u; }   // isalpha, "C" locale

// visit(type, n, text, text_len): n = run length of ':' / payload letters of '+' '-' / 1 for '*'
template <class V> static CsErr cs_scan(const char *cs, int64_t len, V &&visit) {
    if (len < 5 || std::memcmp(cs, "cs:Z:", 5) != 0) return CS_E_TAG;
    const char *p = cs + 5, *const e = cs + len;
    while (p < e) {
        const char *const op = p;
        const char t = *p++;
        int64_t n;
        if (t == ':') {                                       // what std::from_chars<int64_t> accepts: [-]digits, in range
            const bool neg = p < e && *p == '-';
            if (neg) p++;
            const char *const d0 = p;
            uint64_t v = 0;
            bool ovf = false;
            for (; p < e && (unsigned)(*p - '0') <= 9u; p++) {
                const uint64_t d = (uint64_t)(*p - '0');
                if (v > ((uint64_t)INT64_MAX - d) / 10) ovf = true; else v = v * 10 + d;
            }
            if (p == d0 || neg || ovf || v == 0) return CS_E_LENGTH;
            n = (int64_t)v;
        } else if (t == '*') {
            if (p + 2 > e || !alpha_ascii(p[0]) || !alpha_ascii(p[1])) return CS_E_SUBST;
            p += 2;
            n = 1;
        } else if (t == '+' || t == '-') {
            const char *const s0 = p;
            while (p < e && alpha_ascii(*p)) p++;
            n = p - s0;
            if (n == 0) return CS_E_INDEL;
        } else return CS_E_OP;
        visit(t, n, op, (int64_t)(p - op));
    }
    return CS_OK;
}

// Query / reference cursor of one walk.  fwd: q = next query base, r = next reference base (both
// ascending).  '-' strand, text order: q = exclusive upper end of the query bases not yet covered
// (descending), r = lowest reference base not yet consumed (ascending: the text of a '-' strand
// record runs along the reference).  at() is what the reference's qry_index holds when it meets
// the same operation.
struct CsCursor {
    bool fwd; int64_t q, r;
    CsCursor(bool f, int64_t qs, int64_t qe, int64_t rs, int64_t re) : fwd(f), q(f ? qs : qe + 1), r(f ? rs : re) {}
    int64_t lo(int64_t n) const { return fwd ? q : q - n; }                 // first query base an n-base operation covers
    void take_q(int64_t n) { q += fwd ? n : -n; }
    bool done(int64_t qs, int64_t qe, int64_t rs, int64_t re) const {       // :119-122
        return fwd ? (q == qe + 1 && r == re + 1) : (q == qs && r == rs + 1);
    }
};

// why a tag was rejected (the fast paths below only say "no"): the reference's message
static CsErr cs_diagnose(const char *cs, int64_t len, bool fwd, int64_t qs, int64_t qe, int64_t rs, int64_t re) {
    CsCursor c(fwd, qs, qe, rs, re);
    const CsErr e = cs_scan(cs, len, [&](char t, int64_t n, const char *, int64_t) {
        if (t == ':' || t == '*') { c.take_q(n); c.r += n; } else if (t == '+') c.take_q(n); else c.r += n;
    });
    if (e != CS_OK) return e;
    return c.done(qs, qe, rs, re) ? CS_OK : CS_E_CONSUME;
}
std::string cs_error_message(const char *cs, int64_t cs_len, bool aln_fwd, int64_t qry_str, int64_t qry_end, int64_t ref_str, int64_t ref_end) {
    return cs_err_text(cs_diagnose(cs, cs_len, aln_fwd, qry_str, qry_end, ref_str, ref_end));
}

static int64_t fast_ranges(const char *p, const char *e, bool fwd, int64_t qs, int64_t qe, int64_t rs, int64_t re, int64_t *ql, int64_t *qr, int64_t *rl);

// get_overlap_range into growing vectors (serial reader, --alt rows, aasm_cs_match_ranges); returns the count or -1 + message
template <class VEC>
static int64_t match_ranges(const char *cs, int64_t cs_len, bool aln_fwd, int64_t qry_str, int64_t qry_end,
                            int64_t ref_str, int64_t ref_end, VEC *ql, VEC *qr, VEC *rl, std::string &err) {
    int64_t colons = 0, n = -1;
    for (int64_t i = 5; i < cs_len; i++) colons += cs[i] == ':';             // upper bound of the range count
    if (cs_len >= 5 && std::memcmp(cs, "cs:Z:", 5) == 0) {
        const size_t base = ql ? ql->size() : 0;
        std::vector<int64_t> tmp;
        int64_t *a, *b, *c;
        if (ql) { ql->resize(base + colons); qr->resize(base + colons); rl->resize(base + colons); a = ql->data() + base; b = qr->data() + base; c = rl->data() + base; }
        else { tmp.resize(3 * (size_t)colons + 3); a = tmp.data(); b = a + colons + 1; c = b + colons + 1; }
        n = fast_ranges(cs + 5, cs + cs_len, aln_fwd, qry_str, qry_end, ref_str, ref_end, a, b, c);
        if (ql) { const size_t keep = base + (size_t)(n > 0 ? n : 0); ql->resize(keep); qr->resize(keep); rl->resize(keep); }
    }
    if (n < 0) { err = cs_err_text(cs_diagnose(cs, cs_len, aln_fwd, qry_str, qry_end, ref_str, ref_end)); return -1; }
    return n;
}

struct Edit { std::string cs; int32_t mat_num, aln_len; bool is_cut; };

// get_edited_paf_data (paf_data.cpp:125-220) in one pass over the tag: every operation is clipped
// against the edited query interval [eq_s, eq_e] as it is met and what survives is rendered at once.
static bool edit_cs(const char *cs, int64_t cs_len, bool aln_fwd, int64_t qry_str, int64_t qry_end,
                    int32_t mat_num, int32_t aln_len, int64_t eq_s, int64_t eq_e, int64_t er_s, int64_t er_e,
                    Edit &out, std::string &err) {
    if (eq_s == qry_str && eq_e == qry_end) {                                  // not cut: the record's own tag (:131-136)
        out.cs.assign(cs, (size_t)cs_len);
        out.mat_num = mat_num; out.aln_len = aln_len; out.is_cut = false;
        return true;
    }
    out.cs.assign("cs:Z:"); out.mat_num = 0; out.aln_len = 0; out.is_cut = true;
    CsCursor c(aln_fwd, qry_str, qry_end, 0, 0);
    int64_t q_bases = 0, r_bases = 0;
    bool ins_clipped = false;
    char num[24];
    const CsErr te = cs_scan(cs, cs_len, [&](char t, int64_t n, const char *text, int64_t text_len) {
        if (t == '-') {                                                        // sits between query bases at() - 1 and at(): kept when both stay (:171-177)
            if (eq_s < c.q && c.q <= eq_e) { out.cs.append(text, (size_t)text_len); r_bases += n; out.aln_len += (int32_t)n; }
            return;
        }
        const int64_t lo = c.lo(n), hi = lo + n - 1;                           // the query bases this operation covers
        const int64_t a = std::max(lo, eq_s), b = std::min(hi, eq_e);
        c.take_q(n);
        if (a > b) return;                                                     // entirely clipped away
        if (t == ':') {                                                        // :144-152: the surviving part of the run
            const int64_t keep = b - a + 1;
            char *e = num + sizeof num, *p = e;
            for (uint64_t u = (uint64_t)keep; ; u /= 10) { *--p = (char)('0' + u % 10); if (u < 10) break; }
            *--p = ':';
            out.cs.append(p, (size_t)(e - p));
            out.mat_num += (int32_t)keep; out.aln_len += (int32_t)keep; q_bases += keep; r_bases += keep;
        } else if (t == '+') {                                                 // :153-164: all of it or an error
            if (a != lo || b != hi) { ins_clipped = true; return; }
            out.cs.append(text, (size_t)text_len); q_bases += n; out.aln_len += (int32_t)n;
        } else {                                                               // '*', :165-170
            out.cs.append(text, (size_t)text_len); q_bases += 1; r_bases += 1; out.aln_len += 1;
        }
    });
    CsErr e = te;
    if (e == CS_OK && ins_clipped) e = CS_E_INS_CLIP;
    if (e == CS_OK && (q_bases != eq_e - eq_s + 1 || r_bases != std::llabs(er_e - er_s) + 1)) e = CS_E_EDIT;   // :209-218
    if (e != CS_OK) { err = cs_err_text(e); return false; }
    return true;
}

// The writers' form of the same walk.  What a re-cut keeps is one stretch of the tag: query coverage is monotone along the
// text, only the first and the last kept ':' run can lose bases (they are rendered anew), '*' '+' '-' operations stay or go
// whole - so the row's tag is  [:head]  +  the kept operations as they stand (ONE copy)  +  [:tail], and the walk itself
// only adds up numbers.  `irregular` (a kept run not written the way std::to_string writes it, e.g. ":007") sends the
// row through edit_cs, which renders every operation.
struct CutPlan { int64_t head_keep, tail_keep; const char *v0, *v1; int32_t mat_num, aln_len; bool irregular; };
static CsErr plan_cut(const char *cs, int64_t cs_len, bool aln_fwd, int64_t qry_str, int64_t qry_end, int64_t eq_s, int64_t eq_e,
                      int64_t er_s, int64_t er_e, CutPlan &pl) {
    pl.head_keep = pl.tail_keep = 0; pl.v0 = pl.v1 = nullptr; pl.mat_num = 0; pl.aln_len = 0; pl.irregular = false;
    CsCursor c(aln_fwd, qry_str, qry_end, 0, 0);
    int64_t q_bases = 0, r_bases = 0;
    bool ins_clipped = false, any = false;
    auto whole = [&](const char *text, int64_t text_len) {                     // an operation kept as it stands
        if (pl.tail_keep) pl.irregular = true;                                 // (cannot happen: nothing is kept behind a shortened run)
        if (!pl.v0) { pl.v0 = text; pl.v1 = text + text_len; }
        else if (text == pl.v1) pl.v1 = text + text_len;
        else pl.irregular = true;                                              // (cannot happen: the kept operations are neighbours)
        any = true;
    };
    const CsErr te = cs_scan(cs, cs_len, [&](char t, int64_t n, const char *text, int64_t text_len) {
        if (t == '-') {
            if (eq_s < c.q && c.q <= eq_e) { whole(text, text_len); r_bases += n; pl.aln_len += (int32_t)n; }
            return;
        }
        const int64_t lo = c.lo(n), hi = lo + n - 1;
        const int64_t a = std::max(lo, eq_s), b = std::min(hi, eq_e);
        c.take_q(n);
        if (a > b) return;
        if (t == ':') {
            const int64_t keep = b - a + 1;
            pl.mat_num += (int32_t)keep; pl.aln_len += (int32_t)keep; q_bases += keep; r_bases += keep;
            if (keep == n) {
                if (text[1] == '0') pl.irregular = true;                       // ":007" comes out as ":7"
                whole(text, text_len);
            } else if (!any) { pl.head_keep = keep; any = true; }
            else if (!pl.tail_keep) pl.tail_keep = keep;
            else pl.irregular = true;
        } else if (t == '+') {
            if (a != lo || b != hi) { ins_clipped = true; return; }
            whole(text, text_len); q_bases += n; pl.aln_len += (int32_t)n;
        } else { whole(text, text_len); q_bases += 1; r_bases += 1; pl.aln_len += 1; }
    });
    if (te != CS_OK) return te;
    if (ins_clipped) return CS_E_INS_CLIP;
    if (q_bases != eq_e - eq_s + 1 || r_bases != std::llabs(er_e - er_s) + 1) return CS_E_EDIT;
    return CS_OK;
}

// ---- reader (alignasm.cpp:76-183) ---------------------------------------------------
static bool parse_i64(std::string_view f, int64_t &v) {
    if (f.empty()) return false;
    char *endp = nullptr;
    std::string tmp(f);
    v = std::strtoll(tmp.c_str(), &endp, 10);
    return endp && *endp == '\0';
}

static int parse_text(const char *text, int64_t len, aasm_paf &paf) {
    std::unordered_map<std::string, int32_t> chr_map;
    std::string ctg_chr;
    std::vector<std::string_view> f;
    int32_t row_global_index = 0;
    paf.ctg_rec_off.assign(1, 0);
    paf.cs_off.assign(1, 0);
    paf.rec_rng_off.assign(1, 0);
    int64_t p = 0;
    while (p < len) {
        const char *nl = (const char *)std::memchr(text + p, '\n', len - p);
        int64_t e = nl ? (nl - text) : len;
        int64_t le = e;
        if (le > p && text[le - 1] == '\r') le--;
        if (le > p) {
            f.clear();
            int64_t s = p;
            for (int64_t i = p; i <= le; i++)
                if (i == le || text[i] == '\t') { f.emplace_back(text + s, i - s); s = i + 1; }
            if (f.size() < 12) { paf.error = "PAF row " + std::to_string(row_global_index) + " has fewer than 12 columns"; return AASM_E_PARSE; }
            std::string qry_chr(f[0]), ref_chr(f[5]);
            if (ctg_chr.empty()) ctg_chr = qry_chr;                    // :115-117
            auto it = chr_map.find(ref_chr);
            int32_t chr_id;
            if (it == chr_map.end()) { chr_id = (int32_t)paf.chr_name.size(); chr_map.emplace(ref_chr, chr_id); paf.chr_name.push_back(ref_chr); }
            else chr_id = it->second;
            if (ctg_chr != qry_chr) {                                  // :125-133
                paf.ctg_name.push_back(ctg_chr);
                paf.ctg_rec_off.push_back(paf.n_records());
                ctg_chr = qry_chr;
            }
            int64_t qtot, qs, qe, rtot, rs, re, mq, mat, aln;
            if (!parse_i64(f[1], qtot) || !parse_i64(f[2], qs) || !parse_i64(f[3], qe) || !parse_i64(f[6], rtot) ||
                !parse_i64(f[7], rs) || !parse_i64(f[8], re) || !parse_i64(f[9], mat) || !parse_i64(f[10], aln) ||
                !parse_i64(f[11], mq)) {
                paf.error = "PAF row " + std::to_string(row_global_index) + ": non-numeric field";
                return AASM_E_PARSE;
            }
            qe--; re--;                                                // closed intervals, :141-151
            bool fwd = !f[4].empty() && f[4][0] == '+';
            if (!fwd) std::swap(rs, re);                               // :155-159
            std::string_view cs;
            for (size_t i = 12; i < f.size(); i++)                     // find_cs_tag, :100-108
                if (f[i].size() >= 5 && f[i].substr(0, 5) == "cs:Z:") { cs = f[i]; break; }
            if (cs.empty()) { paf.error = "Missing cs:Z tag in PAF record for query '" + qry_chr + "'"; return AASM_E_PARSE; }
            std::string err;
            int64_t nr = match_ranges(cs.data(), (int64_t)cs.size(), fwd, qs, qe, rs, re, &paf.rng_qry_l, &paf.rng_qry_r, &paf.rng_ref_l, err);
            if (nr < 0) { paf.error = err + " (row " + std::to_string(row_global_index) + ")"; return AASM_E_PARSE; }
            paf.rec_rng_off.push_back((int64_t)paf.rng_qry_l.size());
            paf.qry_str.push_back(qs); paf.qry_end.push_back(qe); paf.ref_str.push_back(rs); paf.ref_end.push_back(re);
            paf.qry_total.push_back(qtot); paf.ref_total.push_back(rtot);
            paf.ref_chr.push_back(chr_id); paf.mat_num.push_back((int32_t)mat); paf.aln_len.push_back((int32_t)aln);
            paf.row_index.push_back(row_global_index); paf.cord_type.push_back(0);
            paf.aln_fwd.push_back(fwd ? 1 : 0); paf.map_qul.push_back((uint8_t)mq);
            paf.cs_pool.insert(paf.cs_pool.end(), cs.data(), cs.data() + cs.size());
            paf.cs_off.push_back((int64_t)paf.cs_pool.size());
            row_global_index++;
        }
        p = e + 1;
    }
    if (paf.n_records() == 0) { paf.error = "empty PAF"; return AASM_E_PARSE; }
    paf.ctg_name.push_back(ctg_chr);                                   // :180-181
    paf.ctg_rec_off.push_back(paf.n_records());
    return AASM_OK;
}

// ---- parallel reader -------------------------------------------------------------------
// Same result as parse_text(), built for whole-genome files (GBs of cs text): the file is cut
// at line boundaries into one chunk per host thread.  Pass 1 indexes every row (line start,
// cs field) and counts its match ranges (= ':' operations), which fixes every offset in the
// final arrays; pass 2 parses rows straight into those arrays with a fused cs scanner that
// allocates nothing.  Anything the fast path does not recognise as a well-formed row sends the
// whole file through parse_text(), which owns the error messages and the odd cases.
static std::atomic<int> g_host_threads{0};
// CPUs this process may really use: the machine's, cut to the affinity mask and to the cgroup's CPU quota (a container with
// 16 CPUs' worth of quota on a 256-thread host: 64 busy threads there spend most of every period throttled)
static int usable_cpus() {
    int n = (int)std::thread::hardware_concurrency();
    if (n <= 0) n = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int a = CPU_COUNT(&set); if (a > 0 && a < n) n = a; }
    auto quota = [&](const char *path, const char *path_period) {
        FILE *f = std::fopen(path, "r");
        if (!f) return;
        char a[64] = {0};
        long long q = -1, per = 100000;
        if (path_period) {                                             // cgroup v1: two files
            if (std::fscanf(f, "%lld", &q) != 1) q = -1;
            FILE *g = std::fopen(path_period, "r");
            if (g) { if (std::fscanf(g, "%lld", &per) != 1) per = 100000; std::fclose(g); }
        } else if (std::fscanf(f, "%63s %lld", a, &per) == 2 && std::strcmp(a, "max") != 0) q = std::atoll(a);   // cgroup v2: "<quota|max> <period>"
        std::fclose(f);
        if (q > 0 && per > 0) { const int c = (int)((q + per - 1) / per); if (c > 0 && c < n) n = c; }
    };
    quota("/sys/fs/cgroup/cpu.max", nullptr);
    quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");
    return n;
}
int host_threads() {
    int n = g_host_threads.load();
    if (n <= 0) { static const int cpus = usable_cpus(); n = cpus; if (n > 64) n = 64; }
    return n;
}

static inline bool fast_i64(const char *s, const char *e, int64_t &v) {       // [-]digits, <= 18 of them
    bool neg = false;
    if (s < e && *s == '-') { neg = true; s++; }
    const int64_t n = e - s;
    if (n <= 0 || n > 18) return false;
    int64_t x = 0;
    for (; s < e; s++) { const unsigned d = (unsigned)(*s - '0'); if (d > 9) return false; x = x * 10 + d; }
    v = neg ? -x : x;
    return true;
}
static inline bool field_i64(const char *s, const char *e, int64_t &v) {
    return fast_i64(s, e, v) || parse_i64(std::string_view(s, (size_t)(e - s)), v);
}

// get_overlap_range (paf_data.cpp:90-123) fused with the tokenizer (:29-72): walks the cs text
// once and writes the ranges in query order.  For a '-' strand row the reference visits the
// operations last to first; walking them first to last from the query END gives the same
// ranges in reverse, so they are written and then flipped in place.  Returns the count, or -1.
static int64_t fast_ranges(const char *p, const char *e, bool fwd, int64_t qs, int64_t qe, int64_t rs, int64_t re,
                           int64_t *ql, int64_t *qr, int64_t *rl) {
    int64_t n = 0;
    int64_t q = fwd ? qs : qe + 1, r = fwd ? rs : re;     // fwd: next query / ref base; rev: query end (exclusive) / lowest ref base not yet consumed
    while (p < e) {
        const char t = *p++;
        if (t == ':') {
            int64_t len = 0;
            const char *d0 = p;
            while (p < e && (unsigned)(*p - '0') <= 9u) {                // int64 range, as std::from_chars (:43-47)
                const int64_t d = *p - '0';
                if (len > (INT64_MAX - d) / 10) return -1;
                len = len * 10 + d; p++;
            }
            if (p == d0 || len <= 0) return -1;
            if (fwd) { ql[n] = q; qr[n] = q + len - 1; rl[n] = r; q += len; r += len; }
            else { ql[n] = q - len; qr[n] = q - 1; rl[n] = r + len - 1; q -= len; r += len; }
            n++;
        } else if (t == '*') {
            if (p + 2 > e || !alpha_ascii(p[0]) || !alpha_ascii(p[1])) return -1;
            p += 2;
            if (fwd) { q++; r++; } else { q--; r++; }
        } else if (t == '+' || t == '-') {
            const char *s0 = p;
            while (p < e && alpha_ascii(*p)) p++;
            const int64_t len = p - s0;
            if (len == 0) return -1;
            if (t == '+') q += fwd ? len : -len; else r += len;
        } else return -1;
    }
    if (fwd ? (q != qe + 1 || r != re + 1) : (q != qs || r != rs + 1)) return -1;   // :119-122
    if (!fwd)
        for (int64_t a = 0, b = n - 1; a < b; a++, b--) { std::swap(ql[a], ql[b]); std::swap(qr[a], qr[b]); std::swap(rl[a], rl[b]); }
    return n;
}

struct RowIdx { int64_t line, cs; int32_t line_len, cs_len, n_colon; };
struct ReadChunk {
    int64_t b = 0, e = 0;                      // byte range (whole lines)
    std::vector<RowIdx> rows;
    int64_t n_ranges = 0, cs_bytes = 0;
    bool bad = false;
    // pass 2
    std::vector<std::string_view> chr_names;   // first-appearance order inside the chunk
    std::vector<std::pair<std::string_view, int64_t>> runs;   // (query name, first global row) of each run of equal names
};

static void read_pass1(const char *text, ReadChunk &ck) {
    int64_t p = ck.b;
    while (p < ck.e) {
        const char *nl = (const char *)std::memchr(text + p, '\n', (size_t)(ck.e - p));
        const int64_t e = nl ? (nl - text) : ck.e;
        int64_t le = e;
        if (le > p && text[le - 1] == '\r') le--;
        if (le > p) {
            // the 12 mandatory columns, then the first tag that starts with cs:Z: (find_cs_tag, alignasm.cpp:100-108)
            const char *f = text + p, *end = text + le;
            int nf = 0;
            while (nf < 12) { const char *t = (const char *)std::memchr(f, '\t', (size_t)(end - f)); if (!t) break; f = t + 1; nf++; }
            const char *cs = nullptr, *cs_end = nullptr;
            if (nf == 12) {
                while (f < end) {
                    const char *t = (const char *)std::memchr(f, '\t', (size_t)(end - f));
                    const char *fe = t ? t : end;
                    if (fe - f >= 5 && std::memcmp(f, "cs:Z:", 5) == 0) { cs = f; cs_end = fe; break; }
                    if (!t) break;
                    f = t + 1;
                }
            }
            if (!cs || le - p > INT32_MAX) { ck.bad = true; return; }
            int64_t colons = 0;
            for (const char *c = cs + 5; c < cs_end; c++) colons += (*c == ':');
            if (colons > INT32_MAX) { ck.bad = true; return; }
            ck.rows.push_back(RowIdx{p, (int64_t)(cs - text), (int32_t)(le - p), (int32_t)(cs_end - cs), (int32_t)colons});
            ck.n_ranges += colons;
            ck.cs_bytes += cs_end - cs;
        }
        p = e + 1;
    }
}

static void read_pass2(const char *text, ReadChunk &ck, aasm_paf &paf, int64_t row0, int64_t rng0, int64_t cs0) {
    std::unordered_map<std::string_view, int32_t> chr_map;
    std::string_view last_chr, cur_ctg;
    int32_t last_chr_id = -1;
    bool have_ctg = false;
    int64_t ro = rng0, co = cs0;
    int64_t *QL = paf.rng_qry_l.data(), *QR = paf.rng_qry_r.data(), *RL = paf.rng_ref_l.data();
    for (size_t i = 0; i < ck.rows.size(); i++) {
        const RowIdx &ri = ck.rows[i];
        const int64_t g = row0 + (int64_t)i;
        const char *f[13];
        const char *c = text + ri.line, *end = c + ri.line_len;
        f[0] = c;
        for (int k = 1; k <= 12; k++) { c = (const char *)std::memchr(c, '\t', (size_t)(end - c)) + 1; f[k] = c; }   // pass 1 saw 12 tabs
        int64_t qtot, qs, qe, rtot, rs, re, mq, mat, aln;
        if (!field_i64(f[1], f[2] - 1, qtot) || !field_i64(f[2], f[3] - 1, qs) || !field_i64(f[3], f[4] - 1, qe) ||
            !field_i64(f[6], f[7] - 1, rtot) || !field_i64(f[7], f[8] - 1, rs) || !field_i64(f[8], f[9] - 1, re) ||
            !field_i64(f[9], f[10] - 1, mat) || !field_i64(f[10], f[11] - 1, aln) || !field_i64(f[11], f[12] - 1, mq)) { ck.bad = true; return; }
        qe--; re--;                                                    // closed intervals, :141-151
        const bool fwd = f[5] - 1 > f[4] && f[4][0] == '+';
        if (!fwd) std::swap(rs, re);                                   // :155-159
        const std::string_view qname(f[0], (size_t)(f[1] - 1 - f[0])), rname(f[5], (size_t)(f[6] - 1 - f[5]));
        if (!have_ctg || qname != cur_ctg) { ck.runs.emplace_back(qname, g); cur_ctg = qname; have_ctg = true; }   // :125-133
        int32_t chr_id;
        if (last_chr_id >= 0 && rname == last_chr) chr_id = last_chr_id;
        else {
            auto it = chr_map.find(rname);
            if (it == chr_map.end()) { chr_id = (int32_t)ck.chr_names.size(); chr_map.emplace(rname, chr_id); ck.chr_names.push_back(rname); }
            else chr_id = it->second;
            last_chr = rname; last_chr_id = chr_id;
        }
        const char *cs = text + ri.cs;
        if (paf.device_ranges) ro += ri.n_colon;                       // the GPU parses (and validates) the tag
        else {
            const int64_t nr = fast_ranges(cs + 5, cs + ri.cs_len, fwd, qs, qe, rs, re, QL + ro, QR + ro, RL + ro);
            if (nr < 0) { ck.bad = true; return; }
            ro += nr;
        }
        std::memcpy(paf.cs_pool.data() + co, cs, (size_t)ri.cs_len);
        co += ri.cs_len;
        paf.qry_str[g] = qs; paf.qry_end[g] = qe; paf.ref_str[g] = rs; paf.ref_end[g] = re;
        paf.qry_total[g] = qtot; paf.ref_total[g] = rtot;
        paf.ref_chr[g] = chr_id;                                       // chunk-local id, remapped after the join
        paf.mat_num[g] = (int32_t)mat; paf.aln_len[g] = (int32_t)aln;
        paf.row_index[g] = (int32_t)g; paf.cord_type[g] = 0;
        paf.aln_fwd[g] = fwd ? 1 : 0; paf.map_qul[g] = (uint8_t)mq;
        paf.rec_rng_off[g + 1] = ro; paf.cs_off[g + 1] = co;
    }
    if (ro != rng0 + ck.n_ranges) ck.bad = true;                       // a ':' that was not an operation
}

template <class F> static void run_threads(int n, F fn) {
    std::vector<std::thread> th;
    for (int t = 1; t < n; t++) th.emplace_back(fn, t);
    fn(0);
    for (auto &x : th) x.join();
}

static int parse_text_mt(const char *text, int64_t len, aasm_paf &paf, int flags) {
    paf.device_ranges = (flags & AASM_READ_DEVICE_RANGES) != 0;
    int T = host_threads();
    if (len < (1 << 16)) T = 1;
    std::vector<ReadChunk> ck((size_t)T);
    for (int t = 0; t < T; t++) {                                      // cut at line starts
        int64_t b = len * t / T;
        if (t > 0) { const char *nl = (const char *)std::memchr(text + b - 1, '\n', (size_t)(len - b + 1)); b = nl ? (nl - text) + 1 : len; }
        ck[t].b = b;
        if (t > 0) ck[t - 1].e = b;
    }
    ck[T - 1].e = len;
    for (int t = 0; t + 1 < T; t++) if (ck[t].e < ck[t].b) ck[t].e = ck[t].b;
    const auto tr0 = std::chrono::steady_clock::now();
    run_threads(T, [&](int t) { read_pass1(text, ck[t]); });
    const auto tr1 = std::chrono::steady_clock::now();
    std::vector<int64_t> row0(T + 1, 0), rng0(T + 1, 0), cs0(T + 1, 0);
    bool bad = false;
    for (int t = 0; t < T; t++) {
        bad |= ck[t].bad;
        row0[t + 1] = row0[t] + (int64_t)ck[t].rows.size(); rng0[t + 1] = rng0[t] + ck[t].n_ranges; cs0[t + 1] = cs0[t] + ck[t].cs_bytes;
    }
    const int64_t R = row0[T];
    if (bad || R == 0 || R > INT32_MAX) return parse_text(text, len, paf);   // errors / empty input: the serial reader reports
    {   // sixteen arrays to size (the per-record ones are zero-filled by resize: 0.4 GB at whole-genome scale): one thread each
        const std::vector<std::function<void()>> jobs = {
            [&] { paf.qry_str.resize(R); }, [&] { paf.qry_end.resize(R); }, [&] { paf.ref_str.resize(R); }, [&] { paf.ref_end.resize(R); },
            [&] { paf.qry_total.resize(R); }, [&] { paf.ref_total.resize(R); }, [&] { paf.ref_chr.resize(R); }, [&] { paf.mat_num.resize(R); },
            [&] { paf.aln_len.resize(R); }, [&] { paf.row_index.resize(R); }, [&] { paf.aln_fwd.resize(R); paf.map_qul.resize(R); paf.cord_type.resize(R); },
            [&] { paf.cs_off.resize(R + 1); }, [&] { paf.rec_rng_off.resize(R + 1); },
            [&] { if (!paf.device_ranges) { paf.rng_qry_l.resize(rng0[T]); paf.rng_qry_r.resize(rng0[T]); paf.rng_ref_l.resize(rng0[T]); } },
            [&] { paf.cs_pool.resize(cs0[T]); }};
        const int J = (int)jobs.size(), TJ = std::min(T, J);
        run_threads(TJ, [&](int t) { for (int j = t; j < J; j += TJ) jobs[j](); });
    }
    paf.cs_off[0] = 0; paf.rec_rng_off[0] = 0;
    const auto tr2 = std::chrono::steady_clock::now();
    run_threads(T, [&](int t) { read_pass2(text, ck[t], paf, row0[t], rng0[t], cs0[t]); });
    const auto tr3 = std::chrono::steady_clock::now();
    if (std::getenv("AASM_IO_TIMING"))
        std::fprintf(stderr, "aasm io: reader %d threads: index %.3f s, allocate %.3f s, parse + copy %.3f s (%.1f MB)\n", T, std::chrono::duration<double>(tr1 - tr0).count(),
                     std::chrono::duration<double>(tr2 - tr1).count(), std::chrono::duration<double>(tr3 - tr2).count(), len / 1e6);
    for (int t = 0; t < T; t++) bad |= ck[t].bad;
    if (bad) { paf = aasm_paf(); return parse_text(text, len, paf); }            // (host ranges: parse_text builds them)
    // reference names numbered by first appearance in the file (chr_map, :119-123)
    std::unordered_map<std::string_view, int32_t> chr_map;
    std::vector<std::vector<int32_t>> remap((size_t)T);
    for (int t = 0; t < T; t++)
        for (std::string_view nm : ck[t].chr_names) {
            auto it = chr_map.find(nm);
            if (it == chr_map.end()) { it = chr_map.emplace(nm, (int32_t)paf.chr_name.size()).first; paf.chr_name.emplace_back(nm); }
            remap[t].push_back(it->second);
        }
    run_threads(T, [&](int t) { for (int64_t g = row0[t]; g < row0[t + 1]; g++) paf.ref_chr[g] = remap[t][paf.ref_chr[g]]; });
    // contigs = runs of consecutive rows with one query name; a run may continue across a chunk cut
    paf.ctg_rec_off.clear();
    std::string_view prev;
    bool have = false;
    for (int t = 0; t < T; t++)
        for (auto &run : ck[t].runs) {
            if (have && run.first == prev) continue;
            paf.ctg_name.emplace_back(run.first); paf.ctg_rec_off.push_back(run.second);
            prev = run.first; have = true;
        }
    paf.ctg_rec_off.push_back(R);
    return AASM_OK;
}

// ---- --alt merge (alignasm.cpp:186-332) ------------------------------------------------
// Rows of the second PAF are re-alignments of sub-contig pieces named "<contig>:<start>-<end>"; a row moves back into
// contig coordinates by start - 1.  What the reference's streaming loop amounts to, and how it is done here:
//   1. every row becomes an AltRec (coordinates shifted, match ranges built) - scan_alt_rows, stops at the first bad row;
//   2. a PIECE is a maximal run of consecutive rows with one (contig name, shift); pieces are found on the row list;
//   3. of a piece, the rows whose aln_len / piece_length exceeds the baseline join their contig, in row order; a piece
//      without such a row contributes its first row of maximal positive ratio instead (and a piece whose ratios are all
//      <= 0 a zeroed record, as the reference's value-initialised PafReadData, :243,:314);
//   4. a joining row carries the qry_total of its contig's last record at the time it was read (:269-274): the contig's
//      own, unless a zeroed record is the last one at that moment.  The reference appends a piece's stand-in record only
//      when the NEXT piece's first row has been read (:305-306 after :269), so that row never sees it.
struct AltRec {
    int64_t qs = 0, qe = 0, rs = 0, re = 0, qtot = 0, rtot = 0;
    int32_t chr = 0, mat = 0, aln = 0, row = 0;
    uint8_t fwd = 0, mq = 0, type = 0;             // type: TYPE_ALT for a real row, TYPE_MAIN (0) for the zeroed record
    std::string cs;
    std::vector<int64_t> ql, qr, rl;
    int32_t ctg = 0;                               // contig the piece name resolves to (unknown names: 0, operator[] at :269)
    int64_t shift = 0;                             // start - 1 of the piece
    std::string_view piece;                        // contig part of the piece name (points into the alt text)
    double ratio = 0;                              // aln_len / piece length (:316)
};

// "<name>:<start>-<end>" -> (name, start - 1); the reference's parseString (:209-233)
static bool split_piece_name(std::string_view q, std::string_view &name, int64_t &shift, std::string &err) {
    const size_t colon = q.find(':');
    if (colon == std::string_view::npos) { err = "Invalid input string format"; return false; }
    name = q.substr(0, colon);
    size_t dash = q.find('-', colon + 1);
    if (dash == std::string_view::npos) dash = q.size();
    int64_t start;
    if (!parse_i64(q.substr(colon + 1, dash - colon - 1), start)) { err = "Error parsing number"; return false; }
    shift = start - 1;
    return true;
}

static int scan_alt_rows(const char *text, int64_t len, aasm_paf &paf, std::unordered_map<std::string, int32_t> &chr_map,
                         const std::unordered_map<std::string_view, int32_t> &ctg_of, std::vector<AltRec> &rows) {
    std::vector<std::string_view> f;
    for (int64_t p = 0; p < len;) {
        const char *nl = (const char *)std::memchr(text + p, '\n', len - p);
        const int64_t e = nl ? (nl - text) : len;
        int64_t le = e;
        if (le > p && text[le - 1] == '\r') le--;
        const int64_t b = p;
        p = e + 1;
        if (le == b) continue;
        f.clear();
        for (int64_t i = b, st = b; i <= le; i++)
            if (i == le || text[i] == '\t') { f.emplace_back(text + st, i - st); st = i + 1; }
        const int32_t row = (int32_t)rows.size();
        if (f.size() < 12) { paf.error = "alt PAF row " + std::to_string(row) + " has fewer than 12 columns"; return AASM_E_PARSE; }
        AltRec r;
        r.row = row; r.type = 1;
        std::string ref_name(f[5]);
        auto ci = chr_map.find(ref_name);
        if (ci == chr_map.end()) { ci = chr_map.emplace(ref_name, (int32_t)paf.chr_name.size()).first; paf.chr_name.push_back(ref_name); }
        r.chr = ci->second;
        if (!split_piece_name(f[0], r.piece, r.shift, paf.error)) return AASM_E_PARSE;
        auto ct = ctg_of.find(r.piece);
        r.ctg = ct == ctg_of.end() ? 0 : ct->second;
        int64_t piece_len, qs, qe, re, mq, mat, aln;
        if (!parse_i64(f[1], piece_len) || !parse_i64(f[2], qs) || !parse_i64(f[3], qe) || !parse_i64(f[6], r.rtot) ||
            !parse_i64(f[7], r.rs) || !parse_i64(f[8], re) || !parse_i64(f[9], mat) || !parse_i64(f[10], aln) || !parse_i64(f[11], mq)) {
            paf.error = "alt PAF row " + std::to_string(row) + ": non-numeric field";
            return AASM_E_PARSE;
        }
        r.qs = qs + r.shift; r.qe = qe + r.shift - 1;                               // closed interval in contig coordinates (:275-277)
        r.re = re - 1;
        r.fwd = (!f[4].empty() && f[4][0] == '+') ? 1 : 0;
        if (!r.fwd) std::swap(r.rs, r.re);
        r.mq = (uint8_t)mq; r.mat = (int32_t)mat; r.aln = (int32_t)aln;
        r.ratio = (double)aln / (double)piece_len;
        std::string_view cs;
        for (size_t i = 12; i < f.size() && cs.empty(); i++)
            if (f[i].substr(0, 5) == "cs:Z:") cs = f[i];
        if (cs.empty()) { paf.error = "Missing cs:Z tag in alternative PAF record for query '" + std::string(f[0]) + "'"; return AASM_E_PARSE; }
        r.cs.assign(cs.data(), cs.size());
        std::string err;
        if (match_ranges(cs.data(), (int64_t)cs.size(), r.fwd != 0, r.qs, r.qe, r.rs, r.re, &r.ql, &r.qr, &r.rl, err) < 0) {
            paf.error = err + " (alt row " + std::to_string(row) + ")";
            return AASM_E_PARSE;
        }
        rows.push_back(std::move(r));
    }
    return AASM_OK;
}

static int merge_alt_text(const char *text, int64_t len, double baseline, aasm_paf &paf) {
    const int64_t C = paf.n_contigs();
    std::unordered_map<std::string, int32_t> chr_map;
    for (size_t i = 0; i < paf.chr_name.size(); i++) chr_map.emplace(paf.chr_name[i], (int32_t)i);
    std::unordered_map<std::string_view, int32_t> ctg_of;
    for (int64_t c = 0; c < C; c++) ctg_of[paf.ctg_name[c]] = (int32_t)c;           // a repeated name resolves to its last contig (:136)
    std::vector<AltRec> rows;
    if (int rc = scan_alt_rows(text, len, paf, chr_map, ctg_of, rows)) return rc;

    std::vector<std::vector<AltRec>> added(C);
    std::vector<int64_t> qtot_now(C);                                               // qry_total of each contig's last record so far
    for (int64_t c = 0; c < C; c++) qtot_now[c] = paf.qry_total[paf.ctg_rec_off[c + 1] - 1];
    // qtot_now follows the reference's `paf_data[...].back().qry_total_length` (:269-274) through the same events in the same
    // order: a row reads it when the row is read; a row above the baseline goes in at once (and becomes the last record); a piece's
    // stand-in - its best row, or the zeroed record - goes in only when the NEXT piece's first row has been read (:305-306).
    int32_t pend_c = -1;                                                            // contig and qry_total of the stand-in that has not gone in yet
    int64_t pend_qtot = 0;
    for (size_t g0 = 0, g1; g0 < rows.size(); g0 = g1) {
        for (g1 = g0 + 1; g1 < rows.size() && rows[g1].shift == rows[g0].shift && rows[g1].piece == rows[g0].piece;) g1++;
        const int32_t c = rows[g0].ctg;
        size_t over = 0, top = g1;                                                  // rows above the baseline; first row of the largest positive ratio
        for (size_t i = g0; i < g1; i++) {
            rows[i].qtot = qtot_now[c];
            if (i == g0 && pend_c >= 0) { qtot_now[pend_c] = pend_qtot; pend_c = -1; }
            if (rows[i].ratio > baseline) { over++; qtot_now[c] = rows[i].qtot; }
            if (rows[i].ratio > (top == g1 ? 0.0 : rows[top].ratio)) top = i;
        }
        if (over) {
            for (size_t i = g0; i < g1; i++)
                if (rows[i].ratio > baseline) added[c].push_back(std::move(rows[i]));
        } else if (top != g1) {
            pend_c = c; pend_qtot = rows[top].qtot;
            added[c].push_back(std::move(rows[top]));
        } else {
            added[c].emplace_back();                                                // every ratio <= 0: the zeroed record
            pend_c = c; pend_qtot = 0;
        }
    }
    // rebuild the flat arrays: every contig = its main records followed by the appended ones
    aasm_paf n;
    n.ctg_name = paf.ctg_name; n.chr_name = paf.chr_name; n.has_cs = paf.has_cs; n.device_ranges = paf.device_ranges;
    n.ctg_rec_off.assign(1, 0); n.cs_off.assign(1, 0); n.rec_rng_off.assign(1, 0);
    for (int64_t c = 0; c < C; c++) {
        for (int64_t r = paf.ctg_rec_off[c]; r < paf.ctg_rec_off[c + 1]; r++) {
            n.qry_str.push_back(paf.qry_str[r]); n.qry_end.push_back(paf.qry_end[r]); n.ref_str.push_back(paf.ref_str[r]); n.ref_end.push_back(paf.ref_end[r]);
            n.qry_total.push_back(paf.qry_total[r]); n.ref_total.push_back(paf.ref_total[r]); n.ref_chr.push_back(paf.ref_chr[r]);
            n.mat_num.push_back(paf.mat_num[r]); n.aln_len.push_back(paf.aln_len[r]); n.row_index.push_back(paf.row_index[r]);
            n.cord_type.push_back(paf.cord_type[r]); n.aln_fwd.push_back(paf.aln_fwd[r]); n.map_qul.push_back(paf.map_qul[r]);
            n.cs_pool.insert(n.cs_pool.end(), paf.cs_pool.data() + paf.cs_off[r], paf.cs_pool.data() + paf.cs_off[r + 1]);
            n.cs_off.push_back((int64_t)n.cs_pool.size());
            if (!paf.device_ranges)
                for (int64_t t = paf.rec_rng_off[r]; t < paf.rec_rng_off[r + 1]; t++) { n.rng_qry_l.push_back(paf.rng_qry_l[t]); n.rng_qry_r.push_back(paf.rng_qry_r[t]); n.rng_ref_l.push_back(paf.rng_ref_l[t]); }
            n.rec_rng_off.push_back(n.rec_rng_off.back() + (paf.rec_rng_off[r + 1] - paf.rec_rng_off[r]));
        }
        for (const AltRec &a : added[c]) {
            n.qry_str.push_back(a.qs); n.qry_end.push_back(a.qe); n.ref_str.push_back(a.rs); n.ref_end.push_back(a.re);
            n.qry_total.push_back(a.qtot); n.ref_total.push_back(a.rtot); n.ref_chr.push_back(a.chr);
            n.mat_num.push_back(a.mat); n.aln_len.push_back(a.aln); n.row_index.push_back(a.row);
            n.cord_type.push_back(a.type); n.aln_fwd.push_back(a.fwd); n.map_qul.push_back(a.mq);   // TYPE_ALT (:302)
            n.cs_pool.insert(n.cs_pool.end(), a.cs.begin(), a.cs.end()); n.cs_off.push_back((int64_t)n.cs_pool.size());
            if (!paf.device_ranges) {
                n.rng_qry_l.insert(n.rng_qry_l.end(), a.ql.begin(), a.ql.end()); n.rng_qry_r.insert(n.rng_qry_r.end(), a.qr.begin(), a.qr.end());
                n.rng_ref_l.insert(n.rng_ref_l.end(), a.rl.begin(), a.rl.end());
            }
            n.rec_rng_off.push_back(n.rec_rng_off.back() + (int64_t)a.ql.size());
        }
        n.ctg_rec_off.push_back((int64_t)n.qry_str.size());
    }
    paf = std::move(n);
    return AASM_OK;
}

// ---- writers (alignasm.cpp:398-490) ---------------------------------------------------
static inline void put_i64(std::string &s, int64_t v) {
    char buf[24];
    char *e = buf + sizeof buf, *p = e;
    uint64_t u = v < 0 ? 0 - (uint64_t)v : (uint64_t)v;
    do { *--p = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) *--p = '-';
    s.append(p, (size_t)(e - p));
}

static int emit_line(const aasm_paf &paf, int64_t contig, const std::string &name, const aasm_out_elem &o,
                     std::string &buf, std::string &err) {
    const int64_t r = paf.ctg_rec_off[contig] + o.ctg_index;
    if (o.ctg_index < 0 || r >= paf.ctg_rec_off[contig + 1]) { err = "output element refers to a record outside its contig"; return AASM_E_INVAL; }
    const bool fwd = paf.aln_fwd[r] != 0;
    Edit ed;
    CutPlan pl;
    const char *cs = paf.cs_pool.data() + paf.cs_off[r];
    const int64_t cs_len = paf.cs_off[r + 1] - paf.cs_off[r];
    const bool uncut = o.edited_qry_str == paf.qry_str[r] && o.edited_qry_end == paf.qry_end[r];   // not cut: the record's own cs / mat_num / aln_len (paf_data.cpp:131-136)
    bool planned = false;
    if (uncut) { ed.mat_num = paf.mat_num[r]; ed.aln_len = paf.aln_len[r]; ed.is_cut = false; }
    else {
        const CsErr pe = plan_cut(cs, cs_len, fwd, paf.qry_str[r], paf.qry_end[r], o.edited_qry_str, o.edited_qry_end, o.edited_ref_str, o.edited_ref_end, pl);
        if (pe != CS_OK) { err = cs_err_text(pe); return AASM_E_PARSE; }
        if (!pl.irregular) { planned = true; ed.mat_num = pl.mat_num; ed.aln_len = pl.aln_len; }
        else if (!edit_cs(cs, cs_len, fwd, paf.qry_str[r], paf.qry_end[r], paf.mat_num[r], paf.aln_len[r], o.edited_qry_str,
                          o.edited_qry_end, o.edited_ref_str, o.edited_ref_end, ed, err))
            return AASM_E_PARSE;
    }
    buf += name; buf += '\t';
    put_i64(buf, paf.qry_total[r]); buf += '\t';
    put_i64(buf, o.edited_qry_str); buf += '\t';
    put_i64(buf, o.edited_qry_end + 1); buf += '\t';
    buf += fwd ? '+' : '-'; buf += '\t';
    buf += paf.chr_name[paf.ref_chr[r]]; buf += '\t';
    put_i64(buf, paf.ref_total[r]); buf += '\t';
    put_i64(buf, fwd ? o.edited_ref_str : o.edited_ref_end); buf += '\t';
    put_i64(buf, (fwd ? o.edited_ref_end : o.edited_ref_str) + 1); buf += '\t';
    put_i64(buf, ed.mat_num); buf += '\t';
    put_i64(buf, ed.aln_len); buf += '\t';
    put_i64(buf, paf.map_qul[r]); buf += '\t';
    buf += o.is_alt_path ? "tp:A:S" : "tp:A:P"; buf += '\t';
    buf += "xi:Z:"; buf += paf.cord_type[r] == 0 ? "P_" : "A_"; put_i64(buf, paf.row_index[r]); buf += '\t';
    if (uncut) buf.append(cs, (size_t)cs_len);
    else if (planned) {
        buf += "cs:Z:";
        if (pl.head_keep) { buf += ':'; put_i64(buf, pl.head_keep); }
        if (pl.v0) buf.append(pl.v0, (size_t)(pl.v1 - pl.v0));
        if (pl.tail_keep) { buf += ':'; put_i64(buf, pl.tail_keep); }
    } else buf += ed.cs;
    buf += '\n';
    return AASM_OK;
}


// buffers -> one file, in order; every thread writes its own buffer at its own offset
static int write_buffers(const char *path, const std::vector<std::string> &bufs) {
    // the rounds go to a temporary name beside the target, which takes its place only when every row is on disk: a row that
    // cannot be formatted (a cs tag clipped inside an insertion: get_edited_paf_data throws) or a failing write leaves no
    // truncated .paf behind
    const std::string tmp_path = std::string(path) + ".tmp." + std::to_string((long long)::getpid());
    const int fd = ::open(tmp_path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) { set_last_error(std::string("cannot open ") + path + " for writing"); return AASM_E_IO; }
    const int T = (int)bufs.size();
    std::vector<int64_t> off((size_t)T + 1, 0);
    for (int t = 0; t < T; t++) off[t + 1] = off[t] + (int64_t)bufs[t].size();
    std::atomic<int> rc{AASM_OK};
    run_threads(T, [&](int t) {
        const char *p = bufs[t].data();
        int64_t left = (int64_t)bufs[t].size(), o = off[t];
        while (left > 0) {
            const ssize_t w = ::pwrite(fd, p, (size_t)std::min<int64_t>(left, 1 << 30), (off_t)o);
            if (w <= 0) { rc = AASM_E_IO; return; }
            p += w; o += w; left -= w;
        }
    });
    if (::close(fd) != 0) rc = AASM_E_IO;
    if (rc == AASM_OK && ::rename(tmp_path.c_str(), path) != 0) rc = AASM_E_IO;
    if (rc != AASM_OK) { ::unlink(tmp_path.c_str()); set_last_error(std::string("write to ") + path + " failed"); }
    return rc;
}

// One output file, rows in contig order (as process_output writes them), formatted AND written by T threads.  The contigs
// are cut into chunks of about a megabyte of output; thread t takes chunks t, t + T, ...: it formats a chunk into its own
// buffer (kept from call to call), learns the chunk's file offset from the chunk before it (the offset of chunk k + 1 is
// known once chunk k is FORMATTED, not written) and writes the buffer itself, at once - the copy into the page cache reads
// what its own core has just produced.  Measured on the 2-socket box (tools/write_probe2.cpp, producers of 1 GB/s each into
// one 3 GB file): every producer writing its own chunk 14.7 GB/s at 16 threads, one writer thread taking the chunks in
// order 7.9 (it reads lines that are dirty in other cores' caches), and the earlier form here (rounds of 4 MB per thread,
// formatted by one set of threads and written by a second set) 3-4 GB/s.
template <class EMIT>   // EMIT(contig, buf, err) -> rc : appends every line of one contig (contigs 0 .. C-1 of this call)
static int append_rounds(int fd, int64_t &file_off, const char *path, int64_t C, const std::vector<int64_t> &weight_prefix, std::vector<std::string> *bufs,
                         int max_threads, EMIT emit) {
    const int64_t W = weight_prefix[C];
    int T = host_threads();
    if (max_threads > 0 && T > max_threads) T = max_threads;
    if (W < (1 << 20)) T = 1;
    const auto t0 = std::chrono::steady_clock::now();
    static const bool nowrite = std::getenv("AASM_IO_NOWRITE") != nullptr;   // diagnostic: format only
    const int64_t target = (int64_t)1 << 20;
    std::vector<int64_t> ccut(1, 0);                                  // chunk k = contigs [ccut[k], ccut[k + 1])
    while (ccut.back() < C) {
        int64_t c = std::lower_bound(weight_prefix.begin(), weight_prefix.end(), weight_prefix[ccut.back()] + target) - weight_prefix.begin();
        if (c <= ccut.back()) c = ccut.back() + 1;
        if (c > C) c = C;
        ccut.push_back(c);
    }
    const int64_t NCH = (int64_t)ccut.size() - 1;
    if (T > NCH) T = NCH > 0 ? (int)NCH : 1;
    if (bufs[0].size() < (size_t)T) bufs[0].resize((size_t)T);
    std::mutex mu;
    std::condition_variable cv;
    std::vector<int64_t> start((size_t)NCH + 1, -1);                  // (under mu) file offset of chunk k, once chunk k - 1 is formatted
    start[0] = file_off;
    int64_t stop_at = NCH;                                            // (under mu) chunks from here on are not wanted any more: a row failed to format, a write failed
    int wrc = AASM_OK;
    std::vector<std::string> errs((size_t)T);
    std::vector<int> rcs((size_t)T, AASM_OK);
    std::vector<int64_t> fail_k((size_t)T, -1);
    run_threads(T, [&](int t) {
        std::string &b = bufs[0][t];
        for (int64_t k = t; k < NCH; k += T) {
            { std::lock_guard<std::mutex> lk(mu); if (k >= stop_at) break; }
            b.clear();
            const size_t want = (size_t)(weight_prefix[ccut[k + 1]] - weight_prefix[ccut[k]]) + 4096;   // the weights are byte estimates: no regrowth copies
            if (b.capacity() < want) b.reserve(want);
            int rc = AASM_OK;
            for (int64_t c = ccut[k]; c < ccut[k + 1] && rc == AASM_OK; c++) rc = emit(c, b, errs[t]);
            int64_t o;
            {
                std::unique_lock<std::mutex> lk(mu);
                if (rc != AASM_OK) { rcs[t] = rc; fail_k[t] = k; if (k < stop_at) stop_at = k; cv.notify_all(); break; }
                cv.wait(lk, [&] { return start[k] >= 0 || k > stop_at; });   // the chunk before this one: formatted yet?  (never, if an earlier chunk failed)
                if (start[k] < 0) break;
                o = start[k];
                start[k + 1] = o + (int64_t)b.size();
                cv.notify_all();
            }
            const char *q = b.data();
            int64_t left = (int64_t)b.size();
            while (left > 0 && !nowrite) {
                const ssize_t wr = ::pwrite(fd, q, (size_t)std::min<int64_t>(left, 1 << 30), (off_t)o);
                if (wr <= 0) { std::lock_guard<std::mutex> lk(mu); wrc = AASM_E_IO; if (k < stop_at) stop_at = k; cv.notify_all(); break; }
                q += wr; o += wr; left -= wr;
            }
        }
    });
    int rc = AASM_OK;
    int64_t first = -1;
    for (int t = 0; t < T; t++)
        if (rcs[t] != AASM_OK && (first < 0 || fail_k[t] < first)) { first = fail_k[t]; rc = rcs[t]; set_last_error(errs[t]); }   // the first failing contig in file order
    if (rc == AASM_OK && wrc != AASM_OK) { rc = wrc; set_last_error(std::string("write to ") + path + " failed"); }
    const int64_t end = rc == AASM_OK ? start[NCH] : file_off;
    if (std::getenv("AASM_IO_TIMING"))
        std::fprintf(stderr, "aasm io: %s format + write %.3f s (%.1f MB, %d threads, %lld chunks)\n", path,
                     std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), (end - file_off) / 1e6, T, (long long)NCH);
    if (rc == AASM_OK) file_off = end;
    return rc;
}

// the batch as PAF text (12 columns + tp + cs), rows shared out over the host threads
static void format_rows_mt(const aasm_paf &paf, std::vector<std::string> &bufs) {
    const int64_t R = paf.n_records();
    int T = host_threads();
    if (R < 4096) T = 1;
    bufs.assign((size_t)T, std::string());
    std::vector<int32_t> ctg_of((size_t)R);
    for (int64_t c = 0; c < paf.n_contigs(); c++) for (int64_t r = paf.ctg_rec_off[c]; r < paf.ctg_rec_off[c + 1]; r++) ctg_of[r] = (int32_t)c;
    run_threads(T, [&](int t) {
        std::string &buf = bufs[t];
        const int64_t r0 = R * t / T, r1 = R * (t + 1) / T;
        if (r1 > r0) buf.reserve((size_t)((paf.cs_off[r1] - paf.cs_off[r0]) + (r1 - r0) * 96));
        for (int64_t r = r0; r < r1; r++) {
            const bool fwd = paf.aln_fwd[r] != 0;
            int64_t rs = paf.ref_str[r], re = paf.ref_end[r];
            if (!fwd) std::swap(rs, re);
            buf += paf.ctg_name[ctg_of[r]]; buf += '\t';
            put_i64(buf, paf.qry_total[r]); buf += '\t';
            put_i64(buf, paf.qry_str[r]); buf += '\t';
            put_i64(buf, paf.qry_end[r] + 1); buf += '\t';
            buf += fwd ? '+' : '-'; buf += '\t';
            buf += paf.chr_name[paf.ref_chr[r]]; buf += '\t';
            put_i64(buf, paf.ref_total[r]); buf += '\t';
            put_i64(buf, rs); buf += '\t';
            put_i64(buf, re + 1); buf += '\t';
            put_i64(buf, paf.mat_num[r]); buf += '\t';
            put_i64(buf, paf.aln_len[r]); buf += '\t';
            put_i64(buf, paf.map_qul[r]); buf += "\ttp:A:P\t";
            buf.append(paf.cs_pool.data() + paf.cs_off[r], (size_t)(paf.cs_off[r + 1] - paf.cs_off[r]));
            buf += '\n';
        }
    });
}

}  // namespace aasm

using namespace aasm;

extern "C" {

const char *aasm_last_error(void) { return last_error_cstr(); }

int aasm_paf_parse_mem(const char *text, int64_t len, aasm_paf **out) { return aasm_paf_parse_mem_opts(text, len, 0, out); }
int aasm_paf_read(const char *path, aasm_paf **out) { return aasm_paf_read_opts(path, 0, out); }

int aasm_paf_parse_mem_opts(const char *text, int64_t len, int flags, aasm_paf **out) {
    if (!text || !out) return AASM_E_INVAL;
    aasm_paf *paf = new aasm_paf();
    int rc = parse_text_mt(text, len, *paf, flags);
    if (rc != AASM_OK) { set_last_error(paf->error); delete paf; *out = nullptr; return rc; }
    *out = paf;
    return AASM_OK;
}

int aasm_paf_read_opts(const char *path, int flags, aasm_paf **out) {
    if (!path || !out) return AASM_E_INVAL;
    // the reader threads page the file in themselves: map it instead of copying it
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) { set_last_error(std::string("cannot open ") + path); return AASM_E_IO; }
    struct stat st;
    if (::fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {             // pipes etc.: read into memory
        std::string data;
        char buf[1 << 16];
        ssize_t n;
        while ((n = ::read(fd, buf, sizeof buf)) > 0) data.append(buf, (size_t)n);
        ::close(fd);
        return aasm_paf_parse_mem_opts(data.data(), (int64_t)data.size(), flags, out);
    }
    if (st.st_size == 0) { ::close(fd); return aasm_paf_parse_mem_opts("", 0, flags, out); }
    void *m = ::mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) { set_last_error(std::string("cannot map ") + path); return AASM_E_IO; }
    ::madvise(m, (size_t)st.st_size, MADV_WILLNEED);
    const int rc = aasm_paf_parse_mem_opts((const char *)m, (int64_t)st.st_size, flags, out);
    ::munmap(m, (size_t)st.st_size);
    return rc;
}

int aasm_set_host_threads(int n) { const int old = g_host_threads.exchange(n < 0 ? 0 : n); return old; }

// --alt: merge a second PAF of sub-contig re-alignments (alignasm.cpp:186-332)
int aasm_paf_merge_alt_mem(aasm_paf *paf, const char *text, int64_t len, double alt_baseline) {
    if (!paf || !text) return AASM_E_INVAL;
    if (len == 0) return AASM_OK;                                   // empty file == no --alt (:196-200)
    int rc = merge_alt_text(text, len, alt_baseline, *paf);
    if (rc != AASM_OK) set_last_error(paf->error);
    return rc;
}
int aasm_paf_merge_alt(aasm_paf *paf, const char *path, double alt_baseline) {
    if (!paf || !path) return AASM_E_INVAL;
    FILE *fp = std::fopen(path, "rb");
    if (!fp) { set_last_error(std::string("cannot open ") + path); return AASM_E_IO; }
    std::string data;
    char buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, fp)) > 0) data.append(buf, n);
    std::fclose(fp);
    return aasm_paf_merge_alt_mem(paf, data.data(), (int64_t)data.size(), alt_baseline);
}

void aasm_paf_free(aasm_paf *paf) { delete paf; }
int64_t aasm_paf_n_contigs(const aasm_paf *paf) { return paf ? paf->n_contigs() : 0; }

int aasm_paf_batch(const aasm_paf *paf, aasm_batch_in *v) {
    if (!paf || !v) return AASM_E_INVAL;
    std::memset(v, 0, sizeof(*v));
    v->n_contigs = paf->n_contigs();
    v->n_records = paf->n_records();
    v->n_ranges = paf->rec_rng_off.empty() ? 0 : paf->rec_rng_off.back();
    v->ctg_rec_off = paf->ctg_rec_off.data();
    v->qry_str = paf->qry_str.data(); v->qry_end = paf->qry_end.data();
    v->ref_str = paf->ref_str.data(); v->ref_end = paf->ref_end.data();
    v->qry_total = paf->qry_total.data();
    v->ref_chr = paf->ref_chr.data(); v->aln_fwd = paf->aln_fwd.data(); v->map_qul = paf->map_qul.data();
    v->rec_rng_off = paf->rec_rng_off.data();
    if (!paf->device_ranges) { v->rng_qry_l = paf->rng_qry_l.data(); v->rng_qry_r = paf->rng_qry_r.data(); v->rng_ref_l = paf->rng_ref_l.data(); }
    if (paf->has_cs) { v->cs_text = paf->cs_pool.data(); v->rec_cs_off = paf->cs_off.data(); }
    return AASM_OK;
}

// ---- output session: the three files are opened once (under temporary names), receive the rows of consecutive contig
// ranges - so that a caller can solve one range while the rows of the range before are written - and take their final
// names only when everything is on disk (commit); a row that cannot be formatted (a cs tag clipped inside an insertion:
// get_edited_paf_data throws) or a failing write leaves no truncated .paf behind.
struct aasm_writer {
    std::string path[3], tmp[3];
    std::vector<std::string> bufs[3][2];  // per file: the two sets of per-thread row buffers of append_rounds (fresh buffers per call cost a page fault per 4 KB written)
    int fd[3] = {-1, -1, -1};
    int64_t off[3] = {0, 0, 0};
    int64_t next_contig = 0;
    bool failed = false;
};

int aasm_writer_open(const char *main_path, const char *alt_path, const char *all_path, aasm_writer **w_out) {
    if (!w_out) return AASM_E_INVAL;
    aasm_writer *w = new aasm_writer();
    const char *p[3] = {main_path, alt_path, all_path};
    for (int i = 0; i < 3; i++) {
        if (!p[i]) continue;
        w->path[i] = p[i];
        w->tmp[i] = w->path[i] + ".tmp." + std::to_string((long long)::getpid());
        w->fd[i] = ::open(w->tmp[i].c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (w->fd[i] < 0) {
            set_last_error(std::string("cannot open ") + p[i] + " for writing");
            w->path[i].clear();
            aasm_writer_close(w, 0);
            return AASM_E_IO;
        }
    }
    *w_out = w;
    return AASM_OK;
}

int aasm_writer_close(aasm_writer *w, int commit) {
    if (!w) return AASM_E_INVAL;
    int rc = AASM_OK;
    for (int i = 0; i < 3; i++) {
        if (w->fd[i] >= 0 && ::close(w->fd[i]) != 0) rc = AASM_E_IO;
        w->fd[i] = -1;
    }
    const bool keep = commit && !w->failed && rc == AASM_OK;
    for (int i = 0; i < 3; i++) {
        if (w->path[i].empty()) continue;
        if (keep) { if (::rename(w->tmp[i].c_str(), w->path[i].c_str()) != 0) { rc = AASM_E_IO; set_last_error("cannot rename " + w->tmp[i]); } }
        else ::unlink(w->tmp[i].c_str());
    }
    if (commit && w->failed && rc == AASM_OK) rc = AASM_E_IO;
    delete w;
    return rc;
}

// rows of contigs [contig0, contig0 + out->n_contigs) of `paf`; ranges must arrive in order, without gaps
int aasm_writer_append(aasm_writer *w, const aasm_paf *paf, const aasm_batch_out *out, int64_t contig0) {
    if (!w || !paf || !out || contig0 != w->next_contig || contig0 + out->n_contigs > paf->n_contigs()) return AASM_E_INVAL;
    if (!paf->has_cs) { set_last_error("PAF was generated without cs strings"); return AASM_E_INVAL; }
    if (w->failed) return AASM_E_IO;
    const int64_t C = out->n_contigs;
    // upper estimate of a contig's output bytes: every element costs its record's cs tag + the 14 other columns
    auto bytes_prefix = [&](const int64_t *off, const aasm_out_elem *el, int64_t extra_name) {
        std::vector<int64_t> wp((size_t)C + 1, 0);
        run_threads(host_threads(), [&](int t) {
            const int T = host_threads();
            for (int64_t c = C * t / T; c < C * (t + 1) / T; c++) {
                int64_t b = 0;
                const int64_t r0 = paf->ctg_rec_off[contig0 + c], nrec = paf->ctg_rec_off[contig0 + c + 1] - r0;
                for (int64_t k = off[c]; k < off[c + 1]; k++) {
                    const int64_t ci = el[k].ctg_index;
                    if (ci >= 0 && ci < nrec) b += paf->cs_off[r0 + ci + 1] - paf->cs_off[r0 + ci];
                    b += 150 + (int64_t)paf->ctg_name[contig0 + c].size() + extra_name;
                }
                wp[c + 1] = b;
            }
        });
        for (int64_t c = 0; c < C; c++) wp[c + 1] += wp[c];
        return wp;
    };
    int rcs[3] = {AASM_OK, AASM_OK, AASM_OK};
    std::string errs[3];
    auto do_main = [&]() {                                              // process_output, :407-443
        if (w->fd[0] < 0) return;
        const std::vector<int64_t> wp = bytes_prefix(out->main_off, out->main_elems, 0);
        rcs[0] = append_rounds(w->fd[0], w->off[0], w->path[0].c_str(), C, wp, w->bufs[0], 0, [&](int64_t c, std::string &buf, std::string &err) {
            for (int64_t k = out->main_off[c]; k < out->main_off[c + 1]; k++) {
                const int r = emit_line(*paf, contig0 + c, paf->ctg_name[contig0 + c], out->main_elems[k], buf, err);
                if (r != AASM_OK) return r;
            }
            return (int)AASM_OK;
        });
        if (rcs[0] != AASM_OK) errs[0] = aasm_last_error();
    };
    const int side_threads = std::max(1, host_threads() / 4);          // the two small files take a quarter of the threads again, beside the big one
    auto do_rest = [&]() {                                              // the two small files, beside the big one (three inodes: three write locks)
        if (w->fd[1] >= 0) {
            const std::vector<int64_t> wp = bytes_prefix(out->alt_off, out->alt_elems, 0);
            rcs[1] = append_rounds(w->fd[1], w->off[1], w->path[1].c_str(), C, wp, w->bufs[1], side_threads, [&](int64_t c, std::string &buf, std::string &err) {
                for (int64_t k = out->alt_off[c]; k < out->alt_off[c + 1]; k++) {
                    const int r = emit_line(*paf, contig0 + c, paf->ctg_name[contig0 + c], out->alt_elems[k], buf, err);
                    if (r != AASM_OK) return r;
                }
                return (int)AASM_OK;
            });
            if (rcs[1] != AASM_OK) errs[1] = aasm_last_error();
        }
        if (w->fd[2] >= 0 && rcs[1] == AASM_OK) {                        // process_max_output, :445-485
            std::vector<int64_t> eoff((size_t)C + 1);
            for (int64_t c = 0; c <= C; c++) eoff[c] = out->all_elem_off[out->all_path_off[c]];
            const std::vector<int64_t> wp = bytes_prefix(eoff.data(), out->all_elems, 12);
            rcs[2] = append_rounds(w->fd[2], w->off[2], w->path[2].c_str(), C, wp, w->bufs[2], side_threads, [&](int64_t c, std::string &buf, std::string &err) {
                int32_t cnt = 0;
                for (int64_t pth = out->all_path_off[c]; pth < out->all_path_off[c + 1]; pth++) {
                    ++cnt;
                    const std::string name = paf->ctg_name[contig0 + c] + "." + std::to_string(cnt);
                    for (int64_t k = out->all_elem_off[pth]; k < out->all_elem_off[pth + 1]; k++) {
                        const int r = emit_line(*paf, contig0 + c, name, out->all_elems[k], buf, err);
                        if (r != AASM_OK) return r;
                    }
                }
                return (int)AASM_OK;
            });
            if (rcs[2] != AASM_OK) errs[2] = aasm_last_error();
        }
    };
    {
        std::thread side(do_rest);
        do_main();
        side.join();
    }
    w->next_contig = contig0 + C;
    for (int i = 0; i < 3; i++)
        if (rcs[i] != AASM_OK) { w->failed = true; set_last_error(errs[i]); return rcs[i]; }   // (the main file's error first: file order of the reference's writers)
    return AASM_OK;
}

int aasm_paf_write_outputs(const aasm_paf *paf, const aasm_batch_out *out, const char *main_path, const char *alt_path,
                           const char *all_path) {
    if (!paf || !out || out->n_contigs != paf->n_contigs()) return AASM_E_INVAL;
    if (!paf->has_cs) { set_last_error("PAF was generated without cs strings"); return AASM_E_INVAL; }
    aasm_writer *w = nullptr;
    int rc = aasm_writer_open(main_path, alt_path, all_path, &w);
    if (rc != AASM_OK) return rc;
    rc = aasm_writer_append(w, paf, out, 0);
    const std::string msg = rc != AASM_OK ? aasm_last_error() : "";
    const int rc2 = aasm_writer_close(w, rc == AASM_OK ? 1 : 0);
    if (rc != AASM_OK) { set_last_error(msg); return rc; }
    return rc2;
}

int64_t aasm_cs_match_ranges(const char *cs, int64_t cs_len, int aln_fwd, int64_t qry_str, int64_t qry_end, int64_t ref_str,
                             int64_t ref_end, int64_t *qry_l, int64_t *qry_r, int64_t *ref_l, int64_t cap) {
    std::vector<int64_t> a, b, c;
    std::string err;
    int64_t n = match_ranges(cs, cs_len, aln_fwd != 0, qry_str, qry_end, ref_str, ref_end, &a, &b, &c, err);
    if (n < 0) { set_last_error(err); return AASM_E_PARSE; }
    for (int64_t i = 0; i < n && i < cap; i++) {
        if (qry_l) qry_l[i] = a[i];
        if (qry_r) qry_r[i] = b[i];
        if (ref_l) ref_l[i] = c[i];
    }
    return n;
}

int64_t aasm_cs_edit(const char *cs, int64_t cs_len, int aln_fwd, int64_t qry_str, int64_t qry_end, int64_t e_qry_str,
                     int64_t e_qry_end, int64_t e_ref_str, int64_t e_ref_end, char *out_cs, int64_t cap, int32_t *mat_num,
                     int32_t *aln_len, int32_t *is_cut) {
    Edit ed;
    std::string err;
    // mat_num / aln_len of the uncut record are passed IN through the out pointers
    int32_t m0 = mat_num ? *mat_num : 0, a0 = aln_len ? *aln_len : 0;
    if (!edit_cs(cs, cs_len, aln_fwd != 0, qry_str, qry_end, m0, a0, e_qry_str, e_qry_end, e_ref_str, e_ref_end, ed, err)) {
        set_last_error(err);
        return AASM_E_PARSE;
    }
    if (mat_num) *mat_num = ed.mat_num;
    if (aln_len) *aln_len = ed.aln_len;
    if (is_cut) *is_cut = ed.is_cut ? 1 : 0;
    int64_t n = (int64_t)ed.cs.size();
    if (out_cs && cap > 0) { int64_t m = std::min(n, cap - 1); std::memcpy(out_cs, ed.cs.data(), m); out_cs[m] = 0; }
    return n;
}

int aasm_paf_to_text(const aasm_paf *paf, char **text, int64_t *len) {
    if (!paf || !text || !len) return AASM_E_INVAL;
    if (!paf->has_cs) { set_last_error("PAF was generated without cs strings"); return AASM_E_INVAL; }
    std::vector<std::string> bufs;
    format_rows_mt(*paf, bufs);
    size_t total = 0;
    for (auto &b : bufs) total += b.size();
    *len = (int64_t)total;
    *text = (char *)std::malloc(total + 1);
    if (!*text) return AASM_E_NOMEM;
    size_t o = 0;
    for (auto &b : bufs) { std::memcpy(*text + o, b.data(), b.size()); o += b.size(); }
    (*text)[total] = 0;
    return AASM_OK;
}

int aasm_paf_save(const aasm_paf *paf, const char *path) {
    if (!paf || !path) return AASM_E_INVAL;
    if (!paf->has_cs) { set_last_error("PAF was generated without cs strings"); return AASM_E_INVAL; }
    std::vector<std::string> bufs;
    format_rows_mt(*paf, bufs);
    return write_buffers(path, bufs);
}

}  // extern "C"
