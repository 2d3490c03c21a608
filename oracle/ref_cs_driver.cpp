// oracle/ref_cs_driver.cpp -- TEST INFRASTRUCTURE ONLY (tests/, tests/golden/make_ref_cs.py).
//
// C entry points over the REAL reference cs codec: get_overlap_range() and
// get_edited_paf_data() (/root/reference/src/paf_data.cpp:90-220, with parse_short_cs :29-72).
// Those functions sit in the first 220 lines of paf_data.cpp, ahead of solve_ctg_read(), and use
// only <charconv>/<string_view>/<vector> + paf_data.hpp; the one include of that file this image
// lacks (<ankerl/unordered_dense.h>, line 13) is needed from line 739 on only.  oracle/Makefile
// therefore pipes the head of paf_data.cpp -- every line before `void solve_ctg_read(`, minus that
// one #include -- from where it lies straight into g++'s stdin and links the object with this
// driver.  Nothing is substituted, no stand-in header is written, no line of reference source is
// stored in the repo or under oracle/_ref/ (the object file is deleted after the link).
//
// This file only declares what paf_data.hpp declares and turns exceptions into codes + text.
#include "paf_data.hpp"

#include <cstring>
#include <stdexcept>
#include <string>

bool NON_SKIP_LINKABLE = false;   // defined in alignasm.cpp:26 in the reference; unused by the cs codec

namespace {
int fail(char* err, int64_t err_cap, int code, const char* what) {
    if (err && err_cap > 0) {
        std::strncpy(err, what, size_t(err_cap) - 1);
        err[err_cap - 1] = 0;
    }
    return code;
}
PafReadData make_row(const char* cs, int64_t cs_len, int aln_fwd, int64_t qry_str, int64_t qry_end, int64_t ref_str, int64_t ref_end) {
    PafReadData r{};
    r.cs_string.assign(cs, size_t(cs_len));
    r.aln_fwd = aln_fwd != 0;
    r.qry_str = qry_str; r.qry_end = qry_end;
    r.ref_str = ref_str; r.ref_end = ref_end;
    return r;
}
}  // namespace

extern "C" {

// get_overlap_range(): returns the number of ranges (>= 0; the first min(n, cap) are written),
// -1 = std::invalid_argument, -2 = other exception; the exception text goes to err.
int64_t ref_cs_overlap_range(const char* cs, int64_t cs_len, int aln_fwd, int64_t qry_str, int64_t qry_end, int64_t ref_str, int64_t ref_end,
                             int64_t* qry_l, int64_t* qry_r, int64_t* ref_l, int64_t* ref_r, int64_t cap, char* err, int64_t err_cap) {
    try {
        PafReadData r = make_row(cs, cs_len, aln_fwd, qry_str, qry_end, ref_str, ref_end);
        get_overlap_range(r, std::string_view(r.cs_string));
        const int64_t n = int64_t(r.qry_overlap_range.size());
        if (r.ref_overlap_range.size() != r.qry_overlap_range.size()) return fail(err, err_cap, -2, "range lists differ in length");
        for (int64_t i = 0; i < n && i < cap; ++i) {
            qry_l[i] = r.qry_overlap_range[size_t(i)].first;  qry_r[i] = r.qry_overlap_range[size_t(i)].second;
            ref_l[i] = r.ref_overlap_range[size_t(i)].first;  ref_r[i] = r.ref_overlap_range[size_t(i)].second;
        }
        return n;
    } catch (const std::invalid_argument& e) {
        return fail(err, err_cap, -1, e.what());
    } catch (const std::exception& e) {
        return fail(err, err_cap, -2, e.what());
    }
}

// get_edited_paf_data(): returns the length of the edited cs string (written to out, NUL-terminated
// when it fits), -1 = std::invalid_argument, -3 = std::logic_error (clip inside an insertion, base
// counts that do not match the clipped coordinates), -2 = other.  mat_num / aln_len are in-out:
// the record's columns 10 / 11 on entry, PafEditData's on return.
int64_t ref_cs_edit(const char* cs, int64_t cs_len, int aln_fwd, int64_t qry_str, int64_t qry_end, int64_t ref_str, int64_t ref_end,
                    int64_t e_qry_str, int64_t e_qry_end, int64_t e_ref_str, int64_t e_ref_end,
                    char* out, int64_t out_cap, int32_t* mat_num, int32_t* aln_len, int32_t* is_cut, char* err, int64_t err_cap) {
    try {
        PafReadData r = make_row(cs, cs_len, aln_fwd, qry_str, qry_end, ref_str, ref_end);
        r.mat_num = *mat_num; r.aln_len = *aln_len;
        PafOutputData o;
        o.edited_qry_str = e_qry_str; o.edited_qry_end = e_qry_end;
        o.edited_ref_str = e_ref_str; o.edited_ref_end = e_ref_end;
        PafEditData ed = get_edited_paf_data(o, r);
        *mat_num = ed.mat_num; *aln_len = ed.aln_len; *is_cut = ed.is_cut ? 1 : 0;
        const int64_t n = int64_t(ed.edit_cs_string.size());
        if (out && out_cap > 0) {
            const int64_t m = n < out_cap - 1 ? n : out_cap - 1;
            std::memcpy(out, ed.edit_cs_string.data(), size_t(m));
            out[m] = 0;
        }
        return n;
    } catch (const std::invalid_argument& e) {
        return fail(err, err_cap, -1, e.what());
    } catch (const std::logic_error& e) {
        return fail(err, err_cap, -3, e.what());
    } catch (const std::exception& e) {
        return fail(err, err_cap, -2, e.what());
    }
}

}  // extern "C"
