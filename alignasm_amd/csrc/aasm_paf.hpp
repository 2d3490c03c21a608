// aasm_paf.hpp -- host-side PAF container shared by the codec, the generator and the CLI.
// Mirrors what the reference keeps per record in PafReadData (src/paf_data.hpp:51-67)
// plus the per-file tables of src/alignasm.cpp:87-98 (contig names, reference names).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/alignasm_amd.h"

namespace aasm {
// resize() without the zero fill: the big arrays (match ranges, cs text: GBs at whole-genome
// scale) are sized once and then written, and first touched, by all reader threads at once
template <class T> struct default_init_alloc : std::allocator<T> {
    template <class U> struct rebind { using other = default_init_alloc<U>; };
    template <class U, class... A> void construct(U *p, A &&...a) {
        if constexpr (sizeof...(A) == 0) ::new ((void *)p) U; else ::new ((void *)p) U(std::forward<A>(a)...);
    }
};
using big_i64 = std::vector<int64_t, default_init_alloc<int64_t>>;
using big_char = std::vector<char, default_init_alloc<char>>;
}  // namespace aasm

struct aasm_paf {
    // per contig (consecutive rows with the same query name, alignasm.cpp:125-133)
    std::vector<std::string> ctg_name;
    std::vector<int64_t> ctg_rec_off;          // [C+1]
    // per record, input order
    std::vector<int64_t> qry_str, qry_end, ref_str, ref_end, qry_total, ref_total;
    std::vector<int32_t> ref_chr, mat_num, aln_len, row_index;
    std::vector<uint8_t> aln_fwd, map_qul, cord_type;   // cord_type: TYPE_MAIN=0 / TYPE_ALT=1
    std::vector<int64_t> cs_off;               // [R+1] into cs_pool (each entry "cs:Z:...")
    aasm::big_char cs_pool;
    bool has_cs = true;                        // generator may skip cs strings (bench)
    bool device_ranges = false;                // AASM_READ_DEVICE_RANGES: rng_* stay empty, rec_rng_off holds the counts
    // reference names (chr_map / chr_rev_map, alignasm.cpp:90-93)
    std::vector<std::string> chr_name;
    // match ranges (get_overlap_range, paf_data.cpp:90-123)
    std::vector<int64_t> rec_rng_off;          // [R+1]
    aasm::big_i64 rng_qry_l, rng_qry_r, rng_ref_l;
    std::string error;

    int64_t n_contigs() const { return (int64_t)ctg_name.size(); }
    int64_t n_records() const { return (int64_t)qry_str.size(); }
};

namespace aasm {
// cs codec (own implementation of the behaviour of paf_data.cpp:19-220)
void set_last_error(const std::string &msg);
// message of the tokenizer / get_overlap_range for one record (used when the device reports a bad cs tag)
std::string cs_error_message(const char *cs, int64_t cs_len, bool aln_fwd, int64_t qry_str, int64_t qry_end, int64_t ref_str, int64_t ref_end);
int host_threads();                          // aasm_set_host_threads (0 = all hardware threads), resolved
// aasm_shard.cpp: per-contig cost estimate and the contiguous cost-balanced partition built on it
void contig_costs(const aasm_batch_in *in, double *cost);
void partition_by_cost(const double *cost, int64_t C, int n_shards, int64_t *cuts);
}  // namespace aasm
