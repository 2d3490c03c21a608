#include <initializer_list>
// tests/host_emul/emul.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Compiles the product's kernel bodies (alignasm_amd/csrc/aasm_kernels.h) and launch
// sequence (aasm_pipeline.h) for the HOST with one lane per wave, so the indexing and
// arithmetic logic of every kernel can be diffed against the oracle in the CPU-only
// test tier (`pytest -m "not gpu"`).  The product library never links this file and has
// no CPU fallback: without a HIP device aasm_solve_batch() fails with AASM_E_NODEVICE.
#define AASM_HOST_EMUL 1
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../alignasm_amd/csrc/aasm_pipeline.h"

using namespace aasm;

namespace {
struct EmuBackend {
    static constexpr bool host_emulation = true;
    std::vector<void *> blocks;
    std::map<std::string, std::pair<void *, size_t>> named;
    bool fail = false;
    ~EmuBackend() { for (void *p : blocks) free(p); }
    void *alloc(const char *name, size_t bytes) {
        void *p = malloc(bytes + 64);
        if (!p) { fail = true; return nullptr; }
        memset(p, 0xA5, bytes + 64);               // poison: catches reads of never-written cells
        blocks.push_back(p);
        named[name] = {p, bytes};
        return p;
    }
    bool failed() const { return fail; }
    bool oom() const { return fail; }
    void zero(void *p, size_t n) { memset(p, 0, n); }
    void zero_alloc(void *p, size_t n) { memset(p, 0, n); }
    void fill_ff(void *p, size_t n) { memset(p, 0xFF, n); }
    void fill_byte(void *p, int v, size_t n) { memset(p, v, n); }
    void launch(int kn, int64_t nblocks, int nthreads, const WS &w) {
        if (nthreads > 1024) return;                // (the injected bad launch of the GPU tests: nothing to emulate)
        for (int64_t b = 0; b < nblocks; b++) {
            // one logical thread per block slot: bodies index with bid*nthreads+tid
            for (int t = 0; t < (kn == KN_SORT ? 1 : nthreads_emul(kn, nthreads)); t++) {
                alignas(16) static char lds[65536];
                static_assert(65536 >= AASM_SORTFIX_LDS_BYTES && 65536 >= AASM_SORT_LDS_BYTES && 65536 >= AASM_ENUM_LDS_BYTES && 65536 >= AASM_GB_LDS_BYTES_T(GB_MAXV_L, GB_MAXE_L), "emulation LDS");
                static_assert(AASM_SEL_LDS_BYTES <= AASM_SORT_LDS_BYTES && AASM_HEAP_LDS_BYTES <= AASM_SORT_LDS_BYTES && AASM_LDS_BYTES <= AASM_SORT_LDS_BYTES, "emulation LDS");
                KCtx k{t, kn == KN_SORT ? 1 : nthreads_emul(kn, nthreads), b, nblocks, 0, lds};
                run_kernel_body(kn, k, w);
            }
        }
    }
    // wave-per-X kernels run with ONE lane (AASM_WAVE == 1); thread-per-X kernels keep their block size
    static int nthreads_emul(int kn, int nthreads) {
        switch (kn) {
            case KN_SORT_FIX: case KN_GATHER_PARTS: case KN_ROW_FILL: case KN_GRAPH: case KN_GRAPH_L: case KN_TNX16_WG: case KN_SORT_ROWS_REV: case KN_REV_FILL_W: case KN_REV_FILL_ORD: case KN_REV_FILL_ORD_S: case KN_SIDETRACK_W:
            case KN_REV_SWEEP: case KN_FWD_SWEEP: case KN_REV_SWEEP_G: case KN_FWD_SWEEP_G: case KN_HEAP: case KN_HEAP_MW: case KN_HEAP_MW8: case KN_HEAP_MW16: case KN_ENUM: case KN_SELECT: case KN_GATHER_OUT: case KN_TOPO_FILL: case KN_SEL_RECOVER: case KN_SEL_CONVERT: case KN_SEL_FINAL: case KN_SEL_PLAN: case KN_SEL_PLANFILL:
                return 1;
            case KN_CHAIN3: return CHAIN_WAVES - 1;
            case KN_CHAIN: return CHAIN_WAVES;          // one "thread" per wave, in wave order: the sweep runs to its end, then the pre-pass, then the heaps
            default: return nthreads;
        }
    }
    void scan_i32(const int32_t *in, int64_t n, int64_t *out) { int64_t s = 0; for (int64_t i = 0; i < n; i++) { out[i] = s; s += in[i]; } out[n] = s; }
    void scan_i32_pair(const int32_t *a, int64_t *oa, const int32_t *b, int64_t *ob, int64_t n) { scan_i32(a, n, oa); scan_i32(b, n, ob); }
    void scan_u8(const uint8_t *in, int64_t n, int64_t *out) { int64_t s = 0; for (int64_t i = 0; i < n; i++) { out[i] = s; s += in[i]; } out[n] = s; }
    int64_t read_i64(const int64_t *p) { return *p; }
    void read_i64s(std::initializer_list<const int64_t *> ps, int64_t *out) { int i = 0; for (auto p : ps) out[i++] = *p; }
    void d2h(void *dst, const void *src, size_t n) { memcpy(dst, src, n); }
    void d2h_big(void *dst, const void *src, size_t n) { memcpy(dst, src, n); }
    void fork() {}
    void join() {}
    void use_side(bool) {}
    void fork_again() {}
    void fork2() {}
    void join2() {}
    void use_side2(bool) {}
    void phase_begin(int) {}
    void phase_end(int) {}
};
EmuBackend *g_be = nullptr;
WS g_ws;
}  // namespace

extern "C" {
// K9 step statistics of the last solves (tools/k9_steps.py); reset = 1 clears them
void emul_k9_stats(int64_t *dst, int reset) { for (int i = 0; i < 64; i++) { dst[i] = aasm::g_k9_stat[i]; if (reset) aasm::g_k9_stat[i] = 0; } }
static int64_t g_bad_record = -1;
int64_t emul_last_bad_record() { return g_bad_record; }   // record whose cs tag K0 rejected (AASM_E_PARSE)
int emul_solve_batch(const aasm_batch_in *in, const aasm_opts *opts, aasm_batch_out *out) {
    delete g_be;
    g_be = new EmuBackend();
    aasm_opts o{};
    if (opts) o = *opts;
    PipelineSizes sz;
    int rc = run_pipeline(*g_be, *in, o, g_ws, sz);
    g_bad_record = sz.bad_record;
    if (rc != AASM_OK) return rc;
    return fetch_results(*g_be, g_ws, sz, out);
}
void emul_free_out(aasm_batch_out *out) {
    if (!out) return;
    free(out->main_off); free(out->alt_off); free(out->all_path_off); free(out->all_elem_off);
    free(out->main_elems); free(out->alt_elems); free(out->all_elems); free(out->ctg_status);
    memset(out, 0, sizeof(*out));
}
int64_t emul_debug_fetch(const char *name, void *dst, int64_t cap) {
    if (!g_be) return -1;
    auto it = g_be->named.find(name);
    if (it == g_be->named.end()) return -1;
    if (dst) memcpy(dst, it->second.first, std::min<size_t>((size_t)cap, it->second.second));
    return (int64_t)it->second.second;
}
// kb_sort_fix itself (one contig, LDS form when n <= SF_MAX: work list of sub-ranges, leaf insertion sorts)
void emul_sort_fix_kernel2(int32_t *perm, int64_t n, const int64_t *qs, const int64_t *qe, int depth_test) {
    WS w;
    memset(&w, 0, sizeof(w));
    int64_t rec_off[2] = {0, n};
    int32_t dup = 1;
    std::vector<int64_t> t_qs(n + 1), t_qe(n + 1);
    std::vector<int32_t> t_ix(n + 1);
    w.C = 1; w.R0 = 0; w.rec_off = rec_off; w.dupflag = &dup; w.in_qs = qs; w.in_qe = qe; w.perm = perm;
    w.s_qs = t_qs.data(); w.s_qe = t_qe.data(); w.s_orig = t_ix.data();
    w.sort_depth_test = depth_test;                                  // 0: the real depth limit; d + 1: d partition levels
    alignas(16) static char lds[AASM_SORTFIX_LDS_BYTES];
    KCtx k{0, 1, 0, 1, 0, lds};
    kb_sort_fix(k, w);
}
void emul_sort_fix_kernel(int32_t *perm, int64_t n, const int64_t *qs, const int64_t *qe) { emul_sort_fix_kernel2(perm, n, qs, qe, 0); }
// libstdc++ std::sort replay used by kb_sort_fix, exposed for a direct test
void emul_std_sort_replay(int32_t *idx, int64_t n, const int64_t *qs, const int64_t *qe, int depth_override) {
    SortGlob acc{idx, qs, qe};
    int64_t st_first[72], st_last[72];
    int32_t st_depth[72];
    ss_std_sort(acc, n, depth_override, st_first, st_last, st_depth);
}
}
