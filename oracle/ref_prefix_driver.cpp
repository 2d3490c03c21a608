// oracle/ref_prefix_driver.cpp -- TEST INFRASTRUCTURE ONLY (tests/, tests/golden/make_ref_prefix.py,
// tools/ref_prefix_time.py).  Own driver around the REAL reference's solve_ctg_read() prefix.
//
// What is reference code in libaasm_ref_prefix.so: /root/reference/src/paf_data.cpp:1-738 (minus the
// `#include <ankerl/...>` line), i.e. solve_ctg_read() from the sort through
// `k_walk_solver.k_shortest_walks(src, dest, MAX_PATH_COUNT)`:
//     K1 sort + parts (:241-261), K2 N x N tables + cut merge (:263-378), K3 linkable / get_score
//     (:422-521), K4 make_Graph (:531-696), K5 anomaly BFS (:704-715), K6-K8 (:728-730 -> the headers),
// compiled from where it lies (piped to g++'s stdin, see oracle/Makefile), closed by
// oracle/ref_prefix_epilogue.inc (own code that copies the function's locals into refp::Dump).
// K9 (:739-921, :1489-1649) needs ankerl::unordered_dense and is NOT in this library.
//
// This file: builds std::vector<PafReadData> from an aasm_batch_in the way alignasm.cpp:135-176 fills the
// fields solve_ctg_read reads, calls the function, hands the dump out by name (same names as
// oracle_debug_* in alignasm_oracle.cpp, so the two can be diffed array by array).
#include "ref_prefix_pre.h"
#include "../include/alignasm_amd.h"

#include <chrono>
#include <cstring>
#include <thread>

bool NON_SKIP_LINKABLE = false;                    // alignasm.cpp:26 in the reference
thread_local refp::Dump *refp::g_dump = nullptr;

#ifdef REF_MONOTONIC_NEW
// Flavour "mono" (as in ref_harness.cpp): a bump allocator that never reuses memory, so that the
// reference's pointer tie-break in the k-walk queue (k_shortest_walks.hpp:231) equals allocation order.
#include <sys/mman.h>
#include <new>
namespace {
char *g_arena = nullptr; size_t g_arena_cap = 0, g_arena_top = 0;
void arena_reset() {
    if (!g_arena) {
        g_arena_cap = (size_t)48 << 30;
        g_arena = (char *)mmap(nullptr, g_arena_cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (g_arena == MAP_FAILED) { g_arena = nullptr; g_arena_cap = 0; }
    }
    g_arena_top = 0;
}
}
void *operator new(size_t n) {
    if (!g_arena) arena_reset();
    size_t a = (g_arena_top + 15) & ~(size_t)15;
    if (!g_arena || a + n > g_arena_cap) throw std::bad_alloc();
    g_arena_top = a + n;
    return g_arena + a;
}
void *operator new[](size_t n) { return operator new(n); }
void operator delete(void *) noexcept {}
void operator delete[](void *) noexcept {}
void operator delete(void *, size_t) noexcept {}
void operator delete[](void *, size_t) noexcept {}
#endif

namespace {

refp::Dump g_store;   // one capture at a time (tests are single-threaded)

void load_contig(const aasm_batch_in *in, int64_t c, std::vector<PafReadData> &recs) {
    const int64_t b = in->ctg_rec_off[c], e = in->ctg_rec_off[c + 1];
    recs.clear();
    recs.reserve(size_t(e - b));
    for (int64_t r = b; r < e; r++) {
        PafReadData x{};
        x.paf_index = int32_t(r);
        x.ctg_index = int32_t(r - b);                 // alignasm.cpp:334-338
        x.ctg_sorted_index = -1;
        x.qry_str = in->qry_str[r]; x.qry_end = in->qry_end[r];
        x.ref_str = in->ref_str[r]; x.ref_end = in->ref_end[r];      // already swapped for '-' rows (alignasm.cpp:156-159)
        x.qry_total_length = in->qry_total[r];
        x.ref_total_length = 0;
        x.ref_chr = in->ref_chr[r];
        x.map_qul = in->map_qul[r];
        x.aln_fwd = in->aln_fwd[r] != 0;
        const int64_t step = x.aln_fwd ? 1 : -1;
        for (int64_t k = in->rec_rng_off[r]; k < in->rec_rng_off[r + 1]; k++) {   // what get_overlap_range() leaves (paf_data.cpp:101-107)
            const int64_t l = in->rng_qry_l[k], rr = in->rng_qry_r[k], f = in->rng_ref_l[k];
            x.qry_overlap_range.emplace_back(l, rr);
            x.ref_overlap_range.emplace_back(f, f + (rr - l) * step);
        }
        recs.push_back(std::move(x));
    }
}

void run_prefix(std::vector<PafReadData> &recs) {
    std::vector<PafOutputData> out, alt;
    std::vector<std::vector<PafOutputData>> mx;
    solve_ctg_read(recs, out, alt, mx);
}

} // namespace

extern "C" {

// Run the reference's solve_ctg_read() prefix on ONE contig and capture its locals.  n_paths: how many of the
// k walks to recover.  Returns 0, or -1 on bad arguments, -2 if the reference threw.
int refp_debug_solve(const aasm_batch_in *in, int64_t contig, int nsl, int64_t n_paths) {
    if (!in || contig < 0 || contig >= in->n_contigs) return -1;
    std::map<std::string, std::vector<int64_t>>().swap(g_store.arr);
#ifdef REF_MONOTONIC_NEW
    arena_reset();
#endif
    g_store.max_paths = n_paths;
    NON_SKIP_LINKABLE = nsl != 0;
    int rc = 0;
    {
        std::vector<PafReadData> recs;
        load_contig(in, contig, recs);
        if (recs.empty()) return 0;
        refp::g_dump = &g_store;
        try { run_prefix(recs); } catch (...) { rc = -2; }
        refp::g_dump = nullptr;
        auto &si = g_store.arr["ctg_sorted_index"];                    // paf_data.cpp:244 writes it into the caller's rows
        for (const auto &r : recs) si.push_back(r.ctg_sorted_index);
    }
    NON_SKIP_LINKABLE = false;
    return rc;
}
int64_t refp_debug_size(const char *name) {
    auto it = g_store.arr.find(name);
    return it == g_store.arr.end() ? -1 : int64_t(it->second.size());
}
int64_t refp_debug_copy(const char *name, int64_t *dst, int64_t cap) {
    auto it = g_store.arr.find(name);
    if (it == g_store.arr.end()) return -1;
    const int64_t n = std::min<int64_t>(cap, int64_t(it->second.size()));
    std::memcpy(dst, it->second.data(), size_t(n) * 8);
    return n;
}

// Time the prefix (K1-K8 at the reference's own MAX_PATH_COUNT = 10000, N x N tables included) over contigs
// [c0, c1) on n_threads host threads, one contig per task (alignasm.cpp:351-359).  Seconds of wall time.
double refp_time_batch(const aasm_batch_in *in, int64_t c0, int64_t c1, int n_threads, int nsl) {
    if (!in || c0 < 0 || c1 > in->n_contigs || c0 > c1) return -1.0;
    if (n_threads < 1) n_threads = 1;
    NON_SKIP_LINKABLE = nsl != 0;
    const auto t0 = std::chrono::steady_clock::now();
    auto work = [&](int t) {
        std::vector<PafReadData> recs;
        for (int64_t c = c0 + t; c < c1; c += n_threads) {
            load_contig(in, c, recs);
            if (!recs.empty()) run_prefix(recs);
        }
    };
    if (n_threads == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; t++) th.emplace_back(work, t);
        for (auto &t : th) t.join();
    }
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    NON_SKIP_LINKABLE = false;
    return s;
}

} // extern "C"
