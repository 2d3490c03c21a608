// aasm_gpu.hip -- gfx950 backend of the pipeline + the C-ABI entry points that touch the GPU.
//
// * one named __global__ per pipeline kernel (bodies: aasm_kernels.h), so rocprofv3
//   --kernel-trace shows `aasm_k6_rev_sweep` etc.;
// * exclusive scans (count -> offsets): ONE launch each, single pass with decoupled look-back (aasm_scan_chain);
// * a per-device arena: device memory is carved by bump allocation out of a few large
//   hipMalloc blocks that persist across solves (no hipMalloc in the steady state);
// * everything is enqueued on ONE HIP stream per device context; HIP events bracket each
//   phase on that stream when opts.collect_timing is set.
// There is NO CPU fallback here: without a usable HIP device every solve entry point
// returns AASM_E_NODEVICE.
#include <hip/hip_runtime.h>
#include <initializer_list>

#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>

#include "aasm_pipeline.h"
#include "aasm_paf.hpp"

namespace aasm {

// ------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------
// XCD-aware block -> work mapping (w.xcd_map): workgroups are dealt round-robin over the 8 XCDs, each with an L2 of its own, so
// block b takes work item (b % 8) * (G / 8) + b / 8 - every XCD walks ONE contiguous eighth of the work, and neighbouring items
// (vertices of one contig, conversions of one contig) meet in one L2 instead of eight
__device__ inline int64_t xcd_bid(int64_t b, int64_t g, int on) {
    if (!on || g < 64) return b;
    const int64_t g8 = g & ~(int64_t)7;
    return b < g8 ? (b & 7) * (g8 >> 3) + (b >> 3) : b;
}
#define AASM_DEF_KERNEL(name, KN, TPB)                                                        \
    __global__ void __launch_bounds__(TPB) name(WS w) {                                       \
        KCtx k{(int)threadIdx.x, (int)blockDim.x, xcd_bid((int64_t)blockIdx.x, (int64_t)gridDim.x, w.xcd_map), (int64_t)gridDim.x,    \
               (int)(threadIdx.x & 63), nullptr};                                             \
        run_kernel_body(KN, k, w);                                                            \
    }
// kernels whose wave keeps a working set in LDS (BYTES per 64-thread block); WAVES = waves per
// SIMD the register budget is sized for (5: <= 96 VGPRs, 6: <= 80): residency per CU is
// min(4 * WAVES, 160 KB / BYTES) blocks, and these kernels are latency-bound, so it is throughput
#define AASM_DEF_KERNEL_LDS(name, KN, TPB, BYTES, WAVES)                                      \
    __global__ void __launch_bounds__(TPB, WAVES) name(WS w) {                                    \
        __shared__ __attribute__((aligned(16))) char smem[BYTES];                             \
        KCtx k{(int)threadIdx.x, (int)blockDim.x, xcd_bid((int64_t)blockIdx.x, (int64_t)gridDim.x, w.xcd_map), (int64_t)gridDim.x,    \
               (int)(threadIdx.x & 63), smem};                                                \
        run_kernel_body(KN, k, w);                                                            \
    }
AASM_DEF_KERNEL(aasm_k0_cs_ranges, KN_CS_RANGES, 256)
AASM_DEF_KERNEL_LDS(aasm_k1_sort, KN_SORT, 256, AASM_SORT_LDS_BYTES, 2)
AASM_DEF_KERNEL(aasm_k1_sort_rank, KN_SORT_RANK, 256)
AASM_DEF_KERNEL_LDS(aasm_k1_sort_fix, KN_SORT_FIX, 64, AASM_SORTFIX_LDS_BYTES, 1)
AASM_DEF_KERNEL(aasm_k1_gather_parts, KN_GATHER_PARTS, 64)
AASM_DEF_KERNEL(aasm_k2_ov_count, KN_OV_COUNT, 256)
AASM_DEF_KERNEL(aasm_k2_ov_merge, KN_OV_MERGE, 256)
AASM_DEF_KERNEL(aasm_k2_vcount, KN_VCOUNT, 256)
AASM_DEF_KERNEL(aasm_k2_vfill_rec, KN_VFILL_REC, 256)
AASM_DEF_KERNEL(aasm_k2_vfill_slot, KN_VFILL_SLOT, 256)
AASM_DEF_KERNEL(aasm_k4_nsl, KN_NSL, 256)
AASM_DEF_KERNEL(aasm_k4_row_count, KN_ROW_COUNT, 256)
AASM_DEF_KERNEL(aasm_k4_row_fill, KN_ROW_FILL, 64)
AASM_DEF_KERNEL_LDS(aasm_k46_graph, KN_GRAPH, GB_TPB, AASM_GB_LDS_BYTES, 5)   // rows + reversed CSR + sweep headers of one contig: 31.8 KB of LDS, 5 workgroups (20 waves) per CU
AASM_DEF_KERNEL_LDS(aasm_k46_graph_l, KN_GRAPH_L, GB_TPB, AASM_GB_LDS_BYTES_T(GB_MAXV_L, GB_MAXE_L), 2)   // contigs of up to 3 584 vertices / 8 192 edges: 62 KB, two workgroups per CU
AASM_DEF_KERNEL(aasm_k6_rev_fill, KN_REV_FILL, 256)
AASM_DEF_KERNEL_LDS(aasm_k6_rev_fill_w, KN_REV_FILL_W, 64, AASM_REVF_LDS_BYTES, 8)
AASM_DEF_KERNEL_LDS(aasm_k6_rev_fill_ord, KN_REV_FILL_ORD, 64, AASM_REVO_LDS_BYTES, 2)   // 25 KB of LDS per block: 6 blocks per CU, i.e. at most 2 waves per SIMD
AASM_DEF_KERNEL_LDS(aasm_k6_rev_fill_ord_s, KN_REV_FILL_ORD_S, 64, AASM_REVO_LDS_BYTES_V(REV_ORD_MIDV), 6)   // contigs of <= 3 072 vertices: 6.7 KB
AASM_DEF_KERNEL_LDS(aasm_k6_rev_place, KN_SORT_ROWS_REV, 64, AASM_REVP_LDS_BYTES, 4)
AASM_DEF_KERNEL(aasm_k6_rev_hdr, KN_REV_HDR, 256)
AASM_DEF_KERNEL_LDS(aasm_k6_rev_sweep, KN_REV_SWEEP, 64, AASM_REV_LDS_BYTES, 8)
AASM_DEF_KERNEL_LDS(aasm_k5_fwd_sweep, KN_FWD_SWEEP, 64, AASM_FWD_LDS_BYTES, 8)
AASM_DEF_KERNEL_LDS(aasm_k6_rev_sweep_g, KN_REV_SWEEP_G, 64, (AASM_WAVE / AASM_SWEEP_G) * AASM_REV_LDS_BYTES, 4)
AASM_DEF_KERNEL_LDS(aasm_k5_fwd_sweep_g, KN_FWD_SWEEP_G, 64, (AASM_WAVE / AASM_SWEEP_G) * AASM_FWD_LDS_BYTES, 4)
AASM_DEF_KERNEL(aasm_k7_children, KN_CHILDREN, 256)
AASM_DEF_KERNEL(aasm_k7_heap_cap, KN_HEAP_CAP, 256)
AASM_DEF_KERNEL(aasm_k7_sidetrack, KN_SIDETRACK, 256)
AASM_DEF_KERNEL_LDS(aasm_k7_sidetrack_w, KN_SIDETRACK_W, 64, AASM_SIDE_LDS_BYTES, 8)
AASM_DEF_KERNEL(aasm_k7_heap_hdr, KN_HEAP_HDR, 256)
AASM_DEF_KERNEL(aasm_k7_prep, KN_K7_PREP, 256)
AASM_DEF_KERNEL(aasm_k9_tnx, KN_TNX, 256)
AASM_DEF_KERNEL(aasm_k9_tnx16, KN_TNX16, 256)
AASM_DEF_KERNEL_LDS(aasm_k9_tnx16_wg, KN_TNX16_WG, TNX_TPB, AASM_TNXWG_LDS_BYTES, 4)   // the 16-hop jump records of a small contig from its tree in LDS
AASM_DEF_KERNEL_LDS(aasm_k7_heap, KN_HEAP, 64, AASM_HEAP_LDS_BYTES, 5)
AASM_DEF_KERNEL_LDS(aasm_k67_chain, KN_CHAIN, 64 * CHAIN_WAVES, AASM_CHAIN_LDS_BYTES, 5)   // sweep + pre-pass + BFS order + heaps of one contig, a wave each (96 VGPRs, 19 spilled: worth it for the fifth wave slot per SIMD)
AASM_DEF_KERNEL_LDS(aasm_k67_chain3, KN_CHAIN3, 64 * (CHAIN_WAVES - 1), AASM_CHAIN_LDS_BYTES, 5)   // ... without the order wave (the heap wave keeps its own queue): classes of more than 896 contigs
AASM_DEF_KERNEL_LDS(aasm_k7_heap_mw, KN_HEAP_MW, 256, AASM_MW_LDS_BYTES(4), 4)
AASM_DEF_KERNEL_LDS(aasm_k7_heap_mw8, KN_HEAP_MW8, 512, AASM_MW_LDS_BYTES(8), 4)
AASM_DEF_KERNEL_LDS(aasm_k7_heap_mw16, KN_HEAP_MW16, 1024, AASM_MW_LDS_BYTES(16), 4)
AASM_DEF_KERNEL(aasm_k7_mw_rank, KN_MW_RANK, 256)
AASM_DEF_KERNEL_LDS(aasm_k8_enum, KN_ENUM, 64, AASM_ENUM2_LDS_BYTES, 4)
AASM_DEF_KERNEL_LDS(aasm_k8_enum_s, KN_ENUM_S, 64, AASM_ENUM2_LDS_BYTES_F(EQ_FSMALL), 5)
AASM_DEF_KERNEL_LDS(aasm_k8_enum_heap, KN_ENUM_HEAP, 64, AASM_ENUM_LDS_BYTES, 2)
AASM_DEF_KERNEL_LDS(aasm_k9_select, KN_SELECT, 64, AASM_SEL_LDS_BYTES, 5)
AASM_DEF_KERNEL(aasm_k9_sel_plan, KN_SEL_PLAN, 64)
AASM_DEF_KERNEL(aasm_k9_sel_planfill, KN_SEL_PLANFILL, 64)
AASM_DEF_KERNEL_LDS(aasm_k9_sel_recover, KN_SEL_RECOVER, 64, AASM_SELREC_LDS_BYTES, 8)
AASM_DEF_KERNEL(aasm_k9_sel_classify, KN_SEL_CLASSIFY, 256)
AASM_DEF_KERNEL_LDS(aasm_k9_sel_convert, KN_SEL_CONVERT, 64, AASM_SEL_LDS_BYTES, 5)
AASM_DEF_KERNEL(aasm_k9_sel_final, KN_SEL_FINAL, 64)
AASM_DEF_KERNEL(aasm_k9_topo_count, KN_TOPO_COUNT, 256)
AASM_DEF_KERNEL(aasm_k9_topo_fill, KN_TOPO_FILL, 64)
AASM_DEF_KERNEL(aasm_k9_gather_out, KN_GATHER_OUT, 64)

// ---- T1 truth tables on the device (test entry aasm_debug_predicates) ------------------
// One thread per pair (a, b) of 5-int64 PafDistance tuples {qry, ref, anom, qul_nonzero, qul_total}.
// out bit 0: dist_lt<CALC_SUM>(a, b)   bit 1: dist_lt<QRY_SCORE>(a, b)   bit 2: dist_eq(a, b)
//     bit 3: nodeq_key_lt (heap node holding key a, against key b; K7's descent test)
//     bit 4: pq_full_less (a, b as priority-queue candidates with equal node / insertion index; K8)
__global__ void __launch_bounds__(256) aasm_t1_predicates(const int64_t *a, const int64_t *b, int64_t n, uint8_t *out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Dist x, y;
    x.qry = a[5 * i]; x.ref = a[5 * i + 1]; x.anom = (int32_t)a[5 * i + 2]; x.qnz = (int32_t)a[5 * i + 3]; x.qtot = (int32_t)a[5 * i + 4]; x.pad = 0;
    y.qry = b[5 * i]; y.ref = b[5 * i + 1]; y.anom = (int32_t)b[5 * i + 2]; y.qnz = (int32_t)b[5 * i + 3]; y.qtot = (int32_t)b[5 * i + 4]; y.pad = 0;
    NodeQ nd;
    nd.q0.x = (int32_t)(uint32_t)(uint64_t)x.qry; nd.q0.y = (int32_t)((uint64_t)x.qry >> 32);
    nd.q0.z = (int32_t)(uint32_t)(uint64_t)x.ref; nd.q0.w = (int32_t)((uint64_t)x.ref >> 32);
    nd.q1.x = x.anom; nd.q1.y = x.qnz; nd.q1.z = x.qtot; nd.q1.w = 1;
    nd.q2.x = nd.q2.y = -1; nd.q2.z = nd.q2.w = 0;
    uint8_t r = 0;
    r |= dist_lt<CALC_SUM_MODE>(x, y) ? 1 : 0;
    r |= dist_lt<QRY_SCORE_MODE>(x, y) ? 2 : 0;
    r |= dist_eq(x, y) ? 4 : 0;
    r |= nodeq_key_lt(nd, y, y.qry + y.ref) ? 8 : 0;
    r |= pq_full_less(x, 7, 3, y, 7, 3) ? 16 : 0;
    out[i] = r;
}

// ---- generic SSSP: the solver's dijkstra() (k_shortest_walks.hpp:69-87) ------------------------------
// The reference's CLI never reaches it (is_dag = true, paf_data.cpp:728: the shortest-path tree is the DAG
// relaxation, K6), but the solver class offers it for graphs with cycles and BASELINE.json's north_star names it.
// One wave per graph, wave-uniform control: a binary min-heap of (Distance, vertex) in global memory with the
// reference's order (std::greater on std::pair: PafDistance operator< in CALC_SUM mode, then the vertex), lazy
// deletion by `dv != d[v]` (operator==), strict `d[to] > dv + w` relaxation of the popped vertex's list in list
// order (sequential: a list may name a vertex twice).  Results equal the reference's d[] and prev[] exactly.
struct DjEnt { Dist d; int32_t v, p0, p1, p2; };
__device__ __forceinline__ bool dj_ent_less(const DjEnt &a, const DjEnt &b) {       // std::pair<Distance, int64_t> operator<
    if (dist_lt<CALC_SUM_MODE>(a.d, b.d)) return true;
    if (dist_lt<CALC_SUM_MODE>(b.d, a.d)) return false;
    return a.v < b.v;
}
__global__ void __launch_bounds__(64) aasm_sssp_dijkstra_kernel(int64_t n_graphs, const int64_t *voff, const int64_t *rowptr, const int32_t *col,
                                                                const int64_t *w5, const int32_t *src, Dist *d, int32_t *prv, DjEnt *heap, const int64_t *hoff) {
    const int64_t g = blockIdx.x;
    if (g >= n_graphs) return;
    const int lane = threadIdx.x & 63;
    const int64_t vb = voff[g], V = voff[g + 1] - vb;
    Dist *dg = d + vb;
    int32_t *pg = prv + vb;
    DjEnt *H = heap + hoff[g];
    for (int64_t v = lane; v < V; v += 64) { dg[v] = dist_max(); pg[v] = -1; }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    int64_t n = 0;
    const int64_t cap = hoff[g + 1] - hoff[g];
    bool over = false;
    auto push = [&](const Dist &dd, int32_t v) {
        DjEnt x; x.d = dd; x.v = v; x.p0 = x.p1 = x.p2 = 0;
        if (n >= cap) { over = true; return; }                       // more relaxations than edges: a cycle keeps improving the order (the reference would not return)
        int64_t i = n++;
        while (i > 0) {
            const int64_t p = (i - 1) >> 1;
            const DjEnt pe = H[p];
            if (!uni(dj_ent_less(x, pe))) break;
            if (lane == 0) H[i] = pe;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            i = p;
        }
        if (lane == 0) H[i] = x;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    };
    const int32_t s = src[g];
    if (lane == 0) dg[s] = dist_zero();                              // IDENTITY_DISTANCE (:74)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    push(dist_zero(), s);
    while (n > 0 && !over) {
        const DjEnt top = H[0];
        const DjEnt x = H[--n];
        if (n > 0) {                                                 // pop: the last entry sinks from the root
            int64_t i = 0;
            while (true) {
                int64_t c = 2 * i + 1;
                if (c >= n) break;
                DjEnt ce = H[c];
                if (c + 1 < n) { const DjEnt ce2 = H[c + 1]; if (uni(dj_ent_less(ce2, ce))) { ce = ce2; c++; } }
                if (!uni(dj_ent_less(ce, x))) break;
                if (lane == 0) H[i] = ce;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                i = c;
            }
            if (lane == 0) H[i] = x;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
        const int32_t v = uni(top.v);
        const Dist dv = uni(top.d);
        if (!uni(dist_eq(dv, dg[v]))) continue;                      // :79 (operator!=)
        for (int64_t e = rowptr[vb + v]; e < rowptr[vb + v + 1]; e++) {
            const int32_t to = uni(col[e]);
            Dist wd; wd.qry = w5[5 * e]; wd.ref = w5[5 * e + 1]; wd.anom = (int32_t)w5[5 * e + 2]; wd.qnz = (int32_t)w5[5 * e + 3]; wd.qtot = (int32_t)w5[5 * e + 4]; wd.pad = 0;
            const Dist cand = uni(dist_add(dv, wd));
            if (uni(dist_lt<CALC_SUM_MODE>(cand, dg[to]))) {         // d_[to] > dv + w (:81)
                if (lane == 0) { dg[to] = cand; pg[to] = v; }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                push(cand, to);
            }
        }
    }
    if (over && lane == 0) pg[s] = -2;                               // reported by the host entry
}

// ---- Dial's bucketed BFS (k_weighted_bfs.hpp:16-37), one wave per graph -------------------------------------------------
// The reference keeps lim + 1 circular buckets, each a LIFO stack, and walks d = 0, 1, ...: pop the top of bucket d mod (lim + 1),
// skip it when its distance is stale, relax its out-edges in list order; a successful relaxation (dist[nxt] == -1 or > d + cost)
// sets dist / pre and pushes nxt onto bucket (d + cost) mod (lim + 1).  dist is order-independent, pre is not: it names the FIRST
// vertex, in the reference's pop order, that reached the final distance - so the pops stay sequential and the relaxations of ONE
// popped row run on the lanes:
//  * buckets staged in LDS: every stack's top DIAL_WIN entries live in an LDS ring (ring slot = stack position mod DIAL_WIN);
//    a full ring spills its lower half to the stack's slice of global memory in one coalesced store, an empty one refills from it;
//  * a row is relaxed 64 edges at a time; the lanes that succeed are compacted PER BUCKET by ballot + prefix count, so a chunk's
//    pushes land on every stack in list order (what the LIFO pops then reverse, as in the reference);
//  * a chunk that names a head twice (parallel edges) is relaxed edge by edge - the second edge must see the first one's result.
// The solver drives it with lim = 2 on the anomaly weights and keeps one scalar (paf_data.cpp:704-715), which the pipeline folds
// into its forward sweep; this entry is the algorithm itself, for any digraph (cycles allowed) and weights 0 .. lim <= 7.
#define DIAL_WIN 256
#define DIAL_MAXB 8
struct DialLds { int32_t ring[DIAL_MAXB][DIAL_WIN]; };
__global__ void __launch_bounds__(64) aasm_sssp_dial_kernel(int64_t n_graphs, const int64_t *voff, const int64_t *rowptr, const int32_t *col, const int32_t *cost,
                                                            const int32_t *src, int32_t nb, int64_t *dist, int64_t *pre, int32_t *spill, const int64_t *soff) {
    __shared__ DialLds L;
    const int64_t g = blockIdx.x;
    if (g >= n_graphs) return;
    const int lane = threadIdx.x & 63;
    const int64_t vb = voff[g], V = voff[g + 1] - vb;
    int64_t *dg = dist + vb, *pg = pre + vb;
    const int64_t cap = (soff[g + 1] - soff[g]) / nb;                // per bucket
    int32_t *sp = spill + soff[g];
    for (int64_t v = lane; v < V; v += 64) { dg[v] = -1; pg[v] = -1; }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    int32_t base[DIAL_MAXB], cnt[DIAL_MAXB];                        // stack b = global [0, base) + ring [base, base + cnt)
#pragma unroll
    for (int b = 0; b < DIAL_MAXB; b++) { base[b] = 0; cnt[b] = 0; }
    bool over = false;
    // room for m more entries on stack b (m <= 64): spill the lower half of a ring that would overflow
    auto make_room = [&](int b, int32_t m) {
#pragma unroll
        for (int bb = 0; bb < DIAL_MAXB; bb++) if (bb == b && cnt[bb] + m > DIAL_WIN) {
            const int32_t n = DIAL_WIN / 2;
            if ((int64_t)base[bb] + n > cap) { over = true; return; }
            for (int32_t t = lane; t < n; t += 64) sp[(int64_t)bb * cap + base[bb] + t] = L.ring[bb][(base[bb] + t) & (DIAL_WIN - 1)];
            base[bb] += n; cnt[bb] -= n;
        }
    };
    auto push_lanes = [&](int b, bool mine, int32_t v) {             // the lanes with `mine` push v onto stack b, in lane order
        const uint64_t m = __ballot(mine);
        if (!m) return;
        make_room(b, __popcll(m));
        if (over) return;
#pragma unroll
        for (int bb = 0; bb < DIAL_MAXB; bb++) if (bb == b) {
            if (mine) L.ring[bb][(base[bb] + cnt[bb] + __popcll(m & ((1ull << lane) - 1ull))) & (DIAL_WIN - 1)] = v;
            cnt[bb] += __popcll(m);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    };
    const int32_t s0 = src[g];
    if (lane == 0) dg[s0] = 0;
    push_lanes(0, lane == 0, s0);
    int64_t maxd = 0;
    for (int64_t d = 0; d <= maxd && !over; d++) {
        const int b = (int)(d % nb);
        while (!over) {
            int32_t c_b = 0, b_b = 0;
#pragma unroll
            for (int bb = 0; bb < DIAL_MAXB; bb++) if (bb == b) { c_b = cnt[bb]; b_b = base[bb]; }
            if (c_b == 0) {
                if (b_b == 0) break;                                 // the bucket is empty
                const int32_t n = b_b < DIAL_WIN / 2 ? b_b : DIAL_WIN / 2;   // refill the ring from the stack's global part
                for (int32_t t = lane; t < n; t += 64) L.ring[b][(b_b - n + t) & (DIAL_WIN - 1)] = sp[(int64_t)b * cap + b_b - n + t];
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#pragma unroll
                for (int bb = 0; bb < DIAL_MAXB; bb++) if (bb == b) { base[bb] -= n; cnt[bb] += n; }
                continue;
            }
            const int32_t cur = __builtin_amdgcn_readfirstlane(L.ring[b][(b_b + c_b - 1) & (DIAL_WIN - 1)]);   // q.back(); q.pop_back()
#pragma unroll
            for (int bb = 0; bb < DIAL_MAXB; bb++) if (bb == b) cnt[bb]--;
            const int64_t dc = dg[cur];
            if (__builtin_amdgcn_readfirstlane((int32_t)(dc != d))) continue;   // stale (:24)
            const int64_t r0 = rowptr[vb + cur], r1 = rowptr[vb + cur + 1];
            for (int64_t e0 = r0; e0 < r1 && !over; e0 += 64) {
                const int64_t e = e0 + lane;
                const bool act = e < r1;
                const int32_t nxt = act ? col[e] : -1 - lane;
                const int32_t cs = act ? cost[e] : 0;
                // a head named twice in this chunk?  (lane i looks at the lanes below it)
                bool dup = false;
                const int32_t nlan = (int32_t)((r1 - e0 < 64) ? (r1 - e0) : 64);
                for (int32_t j = 0; j + 1 < nlan; j++) dup |= (lane > j) && (__builtin_amdgcn_readlane(nxt, j) == nxt);
                if (__ballot(dup)) {                                 // edge by edge, as the reference (:25-32)
                    for (int32_t j = 0; j < nlan && !over; j++) {
                        const int32_t nj = __builtin_amdgcn_readlane(nxt, j), cj = __builtin_amdgcn_readlane(cs, j);
                        const int64_t nd = d + cj, dn = dg[nj];
                        const bool ok = __builtin_amdgcn_readfirstlane((int32_t)(dn == -1 || dn > nd)) != 0;
                        if (!ok) continue;
                        if (lane == 0) { dg[nj] = nd; pg[nj] = cur; }
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                        push_lanes((int)(nd % nb), lane == 0, nj);
                        if (nd > maxd) maxd = nd;
                    }
                    continue;
                }
                const int64_t nd = d + cs;
                bool ok = false;
                if (act) { const int64_t dn = dg[nxt]; ok = dn == -1 || dn > nd; if (ok) { dg[nxt] = nd; pg[nxt] = cur; } }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                const int bk = (int)(nd % nb);
                for (int bb = 0; bb < nb && !over; bb++) push_lanes(bb, ok && bk == bb, nxt);
                int64_t mx = ok ? nd : 0;
                for (int o = 32; o >= 1; o >>= 1) { const int64_t y = __shfl_xor(mx, o, 64); mx = y > mx ? y : mx; }
                if (mx > maxd) maxd = mx;
            }
        }
    }
    if (over && lane == 0) pg[s0] = -2;                              // reported by the host entry
}

// ---- exclusive scan: T in -> int64 out[n+1] -----------------------------------------
// ONE launch per scan (the pipeline runs ~14 per batch, most over 7-15 M entries): tiles take their number from a
// ticket (so every tile's predecessors are running or done), publish their sum, and find their prefix by looking back
// over the published words, 64 tiles per look (single-pass scan with decoupled look-back).  A word is
// {state : 2, value : 62}; the sums here are counts (>= 0, far below 2^62).  The tile that finishes last clears the
// words and the counters, so the scratch buffer is ready for the next scan on the same stream.
#define SCAN_TPB 256
#define SCAN_IPT 16
#define SCAN_TILE (SCAN_TPB * SCAN_IPT)
#define SCAN_AGG 1ull
#define SCAN_PREFIX 2ull
#define SCAN_HDR 2                           // words ahead of the tile words: ticket, #tiles done
#define SCAN_STALL_S 10                      // look-back gives up after this many seconds without a published predecessor
#define SCAN_STALL_SLOT 63                   // word of DevCtx::pinned (host-pinned, device-visible) the stalled lane raises
__device__ __forceinline__ int64_t scan_wave_incl(int64_t x, int lane) {
    for (int d = 1; d < 64; d <<= 1) { const int64_t y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
    return x;
}
template <class T>
__global__ void __launch_bounds__(SCAN_TPB) aasm_scan_chain(const T *in, int64_t n, int64_t *out, unsigned long long *scr, int64_t nt, int64_t *stall_flag) {
    __shared__ int64_t sh_wave[SCAN_TPB / 64];
    __shared__ int64_t sh_prefix;
    __shared__ unsigned long long sh_tile;
    __shared__ int sh_last;
    unsigned long long *words = scr + SCAN_HDR;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) sh_tile = atomicAdd(&scr[0], 1ull);
    __syncthreads();
    const int64_t tile = (int64_t)sh_tile;
    const int64_t base = tile * SCAN_TILE + (int64_t)threadIdx.x * SCAN_IPT;
    int64_t v[SCAN_IPT], s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_IPT; i++) { v[i] = (base + i < n) ? (int64_t)in[base + i] : 0; s += v[i]; }
    const int64_t incl = scan_wave_incl(s, lane);
    if (lane == 63) sh_wave[wv] = incl;
    __syncthreads();
    int64_t wave_off = 0, total = 0;
#pragma unroll
    for (int i = 0; i < SCAN_TPB / 64; i++) { const int64_t t = sh_wave[i]; if (i < wv) wave_off += t; total += t; }
    // (a ticket beyond the last tile - the counter was left dirty by an aborted launch - has no elements and no slot among the
    // words the last tile clears: it publishes nothing and waits for nobody)
    if (wv == 0 && tile < nt) {                                      // the first wave publishes and looks back
        int64_t prefix = 0;
        if (tile > 0) {
            if (lane == 0) __hip_atomic_store(&words[tile], (SCAN_AGG << 62) | (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int64_t j = tile - 1;                                    // lane i looks at tile j - i
            for (;;) {
                const int64_t mine = j - lane;
                unsigned long long x = SCAN_PREFIX << 62;            // (tiles before the first one: an empty prefix)
                if (mine >= 0) {
                    // a predecessor publishes within microseconds (it took its ticket before this tile did); the guard is for a
                    // scratch buffer left dirty by an aborted launch: after SCAN_STALL_S seconds the lane gives up, raises the
                    // host-visible flag (the host turns it into AASM_E_HIP at its next wait) and the launch drains
                    int64_t polls = 0, t0 = 0;
                    do {
                        x = __hip_atomic_load(&words[mine], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((x >> 62) == 0 && (++polls & 4095) == 0) {
                            const int64_t now = wave_realtime();
                            if (t0 == 0) t0 = now;
                            else if (now - t0 > (int64_t)SCAN_STALL_S * 100000000) {
                                __hip_atomic_store(stall_flag, (int64_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                x = SCAN_PREFIX << 62;
                            }
                        }
                    } while ((x >> 62) == 0);
                }
                const uint64_t pm = __ballot((x >> 62) == SCAN_PREFIX);
                const int stop = pm ? __ffsll((long long)pm) - 1 : 64;   // nearest tile that knows its prefix
                int64_t val = (lane <= stop) ? (int64_t)(x & ((1ull << 62) - 1)) : 0;
                for (int d = 32; d >= 1; d >>= 1) val += __shfl_xor(val, d, 64);
                prefix += val;
                if (pm) break;
                j -= 64;
            }
        }
        if (lane == 0) {
            __hip_atomic_store(&words[tile], (SCAN_PREFIX << 62) | (unsigned long long)(prefix + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh_prefix = prefix;
            if (tile == nt - 1) out[n] = prefix + total;
        }
    }
    __syncthreads();
    int64_t run = sh_prefix + wave_off + incl - s;                   // exclusive prefix of this thread
#pragma unroll
    for (int i = 0; i < SCAN_IPT; i++) { if (base + i < n) out[base + i] = run; run += v[i]; }
    // ---- the last tile to get here resets the scratch words (every look-back is over by then)
    if (threadIdx.x == 0) sh_last = (atomicAdd(&scr[1], 1ull) == (unsigned long long)(nt - 1)) ? 1 : 0;
    __syncthreads();
    if (sh_last) {
        for (int64_t i = threadIdx.x; i < nt; i += SCAN_TPB) __hip_atomic_store(&words[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x == 0) { __hip_atomic_store(&scr[0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(&scr[1], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
}

// ---- fills: the memsets a solve needs between two launches (fresh arrays, counters, 0xFF / 0x7F patterns) as ONE launch ----
#define FILL_MAX 8
#define FILL_BLOCK_BYTES 65536
struct FillSegs { void *p[FILL_MAX]; uint64_t n[FILL_MAX]; uint32_t v[FILL_MAX]; uint32_t blk0[FILL_MAX + 1]; int cnt; };
__global__ void __launch_bounds__(256) aasm_multi_fill(FillSegs s) {
    const uint32_t b = blockIdx.x;
    int i = 0;
    while (i + 1 < s.cnt && b >= s.blk0[i + 1]) i++;
    const uint64_t off = (uint64_t)(b - s.blk0[i]) * FILL_BLOCK_BYTES;
    const uint64_t end = s.n[i] < off + FILL_BLOCK_BYTES ? s.n[i] : off + FILL_BLOCK_BYTES;
    char *p = (char *)s.p[i];
    const uint32_t v = s.v[i];
    if ((((uintptr_t)p) & 15) == 0) {
        uint4 q; q.x = q.y = q.z = q.w = v;
        uint64_t o = off + (uint64_t)threadIdx.x * 16;
        for (; o + 16 <= end; o += 256 * 16) *(uint4 *)(p + o) = q;
        const uint64_t tail = end & ~(uint64_t)15;                  // (off is a multiple of 16: the last partial quad of the segment)
        if (tail >= off && tail + threadIdx.x < end) p[tail + threadIdx.x] = (char)v;
    } else {
        for (uint64_t o = off + threadIdx.x; o < end; o += 256) p[o] = (char)v;
    }
}

// ---- short scans: ONE workgroup, no tickets, no look-back --------------------------------------------------------------
// Seven of a step's thirteen scans run over per-contig or per-conversion counts (5 000 - 11 000 elements on C3): two tiles of the
// single-pass scan above, which then spends 30 us on its ticket, its look-back across two workgroups and the reset of its words.
// One workgroup of 1 024 threads walks such an array 4 096 elements at a time instead; up to two arrays of the same length per
// launch (heap capacities + several-waves capacities, the conversions' two scratch sizes, main + alt output lengths).
#define SCAN_SMALL_MAX 131072
__global__ void __launch_bounds__(1024) aasm_scan_small(const int32_t *in_a, int64_t *out_a, const int32_t *in_b, int64_t *out_b, int64_t n) {
    __shared__ int64_t sh_w[2][16];
    __shared__ int64_t sh_carry[2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x < 2) sh_carry[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 4096) {
        const int64_t i0 = base + (int64_t)threadIdx.x * 4;
        int64_t va[4], vb[4], sa = 0, sb = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) { va[i] = (i0 + i < n) ? (int64_t)in_a[i0 + i] : 0; sa += va[i]; vb[i] = (in_b && i0 + i < n) ? (int64_t)in_b[i0 + i] : 0; sb += vb[i]; }
        const int64_t ia = scan_wave_incl(sa, lane), ib = scan_wave_incl(sb, lane);
        if (lane == 63) { sh_w[0][wv] = ia; sh_w[1][wv] = ib; }
        __syncthreads();
        int64_t oa = sh_carry[0], ob = sh_carry[1], ta = 0, tb = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) { const int64_t xa = sh_w[0][i], xb = sh_w[1][i]; if (i < wv) { oa += xa; ob += xb; } ta += xa; tb += xb; }
        int64_t ra = oa + ia - sa, rb = ob + ib - sb;
#pragma unroll
        for (int i = 0; i < 4; i++) { if (i0 + i < n) { out_a[i0 + i] = ra; if (in_b) out_b[i0 + i] = rb; } ra += va[i]; rb += vb[i]; }
        __syncthreads();
        if (threadIdx.x == 0) { sh_carry[0] += ta; sh_carry[1] += tb; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out_a[n] = sh_carry[0]; if (in_b) out_b[n] = sh_carry[1]; }
}

// ---- scalar read-back: up to twelve device words -> the host-mapped pinned words, ONE launch (it was a copyBuffer per word) ----
struct ScalarSrc { const int64_t *p[12]; };
__global__ void aasm_read_scalars(ScalarSrc src, int n, int64_t *dst) {
    const int i = (int)threadIdx.x;
    if (i < n) __hip_atomic_store(dst + i, *src.p[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------
// device context + backend
// ------------------------------------------------------------------------------------
struct ArenaBlock { char *p; size_t cap, used; };

struct DevCtx {
    int device = -1;
    bool ready = false;
    hipStream_t stream = nullptr, side = nullptr, side2 = nullptr;
    hipEvent_t ev_fork, ev_join, ev_fork2, ev_join2, ev_fork_b;
    int64_t *d_scratch2 = nullptr;      // scan tile sums of the side stream
    size_t d_scratch2_cap = 0;
    std::vector<ArenaBlock> blocks;
    int64_t *pinned = nullptr;          // host-pinned scalar read-back buffer (word SCAN_STALL_SLOT: raised by a scan whose look-back stalled)
    int64_t *pinned_dev = nullptr;      // the same buffer as the device addresses it
    char *stage[2] = {nullptr, nullptr};   // host-pinned staging chunks of the result fetch (created on first use)
    int64_t *d_scratch = nullptr;       // scan tile sums
    size_t d_scratch_cap = 0;
    uint64_t generation = 0;
    std::atomic<int> input_arrived{0};  // an upload for this context has begun: a warm-up that has not allocated yet stands back
    std::mutex mu;
    hipEvent_t ev_b[AASM_N_PHASES], ev_e[AASM_N_PHASES], ev_t0, ev_t1;
    int n_events_made = 0;              // ev_fork, ev_join, then timing_event(0 ..); ev_fork2 / ev_join2 are counted by n_events2
    int n_events2 = 0;
    hipEvent_t *timing_event(int i) { return i < AASM_N_PHASES ? &ev_b[i] : i < 2 * AASM_N_PHASES ? &ev_e[i - AASM_N_PHASES] : i == 2 * AASM_N_PHASES ? &ev_t0 : &ev_t1; }
    bool events = false;
    size_t peak_bytes = 0;
};
static DevCtx g_ctx[16];
static std::mutex g_init_mu;
static std::atomic<int64_t> g_n_range_splits{0}, g_n_device_mallocs{0}, g_n_stream_syncs{0};

static std::string hip_err(const char *what, hipError_t e) {
    return std::string(what) + ": " + hipGetErrorString(e);
}

static int ctx_init(int device) {
    if (device < 0 || device >= 16) { set_last_error("device ordinal out of range"); return AASM_E_INVAL; }
    std::lock_guard<std::mutex> lk(g_init_mu);
    DevCtx &cx = g_ctx[device];
    if (cx.ready) return AASM_OK;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { set_last_error("no HIP device available (this library has no CPU fallback)"); return AASM_E_NODEVICE; }
    if (device >= n) { set_last_error("device ordinal beyond hipGetDeviceCount"); return AASM_E_NODEVICE; }
    if ((e = hipSetDevice(device)) != hipSuccess) { set_last_error(hip_err("hipSetDevice", e)); return AASM_E_NODEVICE; }
    // everything or nothing: a failure leaves no half-made context behind (the next call starts over)
    auto fail = [&](const char *what, hipError_t err) {
        set_last_error(hip_err(what, err));
        if (cx.pinned) { hipHostFree(cx.pinned); cx.pinned = nullptr; }
        if (cx.n_events_made > 0) { hipEventDestroy(cx.ev_fork); }
        if (cx.n_events_made > 1) { hipEventDestroy(cx.ev_join); }
        for (int i = 2; i < cx.n_events_made; i++) hipEventDestroy(*cx.timing_event(i - 2));
        cx.n_events_made = 0;
        if (cx.n_events2 > 0) { hipEventDestroy(cx.ev_fork2); }
        if (cx.n_events2 > 1) { hipEventDestroy(cx.ev_join2); }
        if (cx.n_events2 > 2) { hipEventDestroy(cx.ev_fork_b); }
        cx.n_events2 = 0;
        if (cx.side2) { hipStreamDestroy(cx.side2); cx.side2 = nullptr; }
        if (cx.side) { hipStreamDestroy(cx.side); cx.side = nullptr; }
        if (cx.stream) { hipStreamDestroy(cx.stream); cx.stream = nullptr; }
        return AASM_E_NODEVICE;
    };
    if ((e = hipStreamCreateWithFlags(&cx.stream, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
    if ((e = hipStreamCreateWithFlags(&cx.side, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
    if ((e = hipStreamCreateWithFlags(&cx.side2, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
    if ((e = hipEventCreateWithFlags(&cx.ev_fork2, hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", e);
    cx.n_events2 = 1;
    if ((e = hipEventCreateWithFlags(&cx.ev_join2, hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", e);
    cx.n_events2 = 2;
    if ((e = hipEventCreateWithFlags(&cx.ev_fork_b, hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", e);
    cx.n_events2 = 3;
    if ((e = hipEventCreateWithFlags(&cx.ev_fork, hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", e);
    cx.n_events_made = 1;
    if ((e = hipEventCreateWithFlags(&cx.ev_join, hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", e);
    cx.n_events_made = 2;
    for (int i = 0; i < 2 * AASM_N_PHASES + 2; i++) {
        if ((e = hipEventCreate(cx.timing_event(i))) != hipSuccess) return fail("hipEventCreate", e);
        cx.n_events_made++;
    }
    if ((e = hipHostMalloc((void **)&cx.pinned, 64 * sizeof(int64_t), hipHostMallocMapped)) != hipSuccess) return fail("hipHostMalloc", e);
    std::memset(cx.pinned, 0, 64 * sizeof(int64_t));
    if ((e = hipHostGetDevicePointer((void **)&cx.pinned_dev, cx.pinned, 0)) != hipSuccess) return fail("hipHostGetDevicePointer", e);
    cx.events = true;
    cx.device = device;
    cx.ready = true;
    return AASM_OK;
}

struct GpuBackend {
    static constexpr bool host_emulation = false;
    DevCtx &cx;
    hipStream_t stream, main_stream;
    bool on_side = false, forked = false, forked2 = false;
    bool timing;
    bool fail = false, out_of_memory = false;
    bool phase_used[AASM_N_PHASES] = {false};
    size_t cur_block = 0, bytes = 0;
    std::map<std::string, std::pair<void *, size_t>> named;
    GpuBackend(DevCtx &c, hipStream_t s, bool t) : cx(c), stream(s), main_stream(s), timing(t) {
        for (auto &b : cx.blocks) b.used = 0;
        cx.generation++;
    }
    void hip_fail(const char *what, hipError_t e) { if (!fail) set_last_error(hip_err(what, e)); fail = true; }
    void *alloc(const char *name, size_t n) {
        n = (n + 255) & ~(size_t)255;
        while (cur_block < cx.blocks.size() && cx.blocks[cur_block].used + n > cx.blocks[cur_block].cap) cur_block++;
        if (cur_block >= cx.blocks.size()) {
            // a new block doubles the arena (first block: 1 GB), so a cold start takes a handful of hipMalloc
            // calls whatever the batch needs; when the doubled size does not fit, only what is asked for
            size_t have = 0;
            for (auto &b : cx.blocks) have += b.cap;
            size_t cap = have > ((size_t)1 << 30) ? have : ((size_t)1 << 30);
            if (cap < n) cap = n;
            char *p = nullptr;
            hipError_t e = hipMalloc((void **)&p, cap);
            g_n_device_mallocs++;
            if (e == hipErrorOutOfMemory && cap > n) { (void)hipGetLastError(); cap = n; e = hipMalloc((void **)&p, cap); g_n_device_mallocs++; }
            if (e != hipSuccess) {
                if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); out_of_memory = true; if (!fail) set_last_error("out of device memory (workspace arena)"); fail = true; }
                else hip_fail("hipMalloc", e);
                return nullptr;
            }
            cx.blocks.push_back(ArenaBlock{p, cap, 0});
            cur_block = cx.blocks.size() - 1;
        }
        ArenaBlock &b = cx.blocks[cur_block];
        void *p = b.p + b.used;
        b.used += n;
        bytes += n;
        if (bytes > cx.peak_bytes) cx.peak_bytes = bytes;
        named[name] = {p, n};
        return p;
    }
    bool failed() const { return fail; }
    bool test_dirty_scan = false;
    // a scan's look-back gave up (aasm_scan_chain): every later size is garbage - fail the solve before anything is sized by it
    bool scan_stalled() {
        if (!cx.pinned[SCAN_STALL_SLOT]) return false;
        if (!fail) set_last_error("scan look-back stalled (scratch words left dirty by an aborted launch?)");
        fail = true;
        return true;
    }
    bool oom() const { return out_of_memory; }
    // Fills are DEFERRED: a request joins a pending list (a zero fill of an array that sits right behind the last one in the arena
    // only extends it), and whatever touches the stream next issues the list first - as ONE launch (aasm_multi_fill), or a plain
    // memset when it is a single span.  A step of the pipeline had 19 fillBufferAligned dispatches.
    FillSegs fs_;
    bool fs_fresh_[FILL_MAX] = {false};                              // span i is made of WHOLE fresh allocations only (may be extended over the allocator's padding)
    int n_fs = 0;
    void flush_zero() {
        if (n_fs == 0) return;
        const int n = n_fs;
        n_fs = 0;
        if (fail) return;
        if (n == 1) {
            hipError_t e = hipMemsetAsync(fs_.p[0], (int)(fs_.v[0] & 0xff), (size_t)fs_.n[0], stream);
            if (e != hipSuccess) hip_fail("hipMemsetAsync", e);
            return;
        }
        uint32_t blocks = 0;
        for (int i = 0; i < n; i++) { fs_.blk0[i] = blocks; blocks += (uint32_t)((fs_.n[i] + FILL_BLOCK_BYTES - 1) / FILL_BLOCK_BYTES); }
        fs_.blk0[n] = blocks; fs_.cnt = n;
        hipLaunchKernelGGL(aasm_multi_fill, dim3(blocks), dim3(256), 0, stream, fs_);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) hip_fail("fill launch", e);
    }
    void add_fill(void *p, int v, size_t n, bool fresh) {
        if (!p || n == 0 || fail) return;
        char *c = (char *)p;
        const uint32_t vv = (uint32_t)(v & 0xff) * 0x01010101u;
        bool overlap = false;                                        // spans of one launch are written concurrently: an overlapping request waits for the list
        for (int i = 0; i < n_fs; i++) overlap |= c < (char *)fs_.p[i] + fs_.n[i] && (char *)fs_.p[i] < c + n;
        // A WHOLE fresh allocation right behind the last span - itself made of whole fresh allocations, so that the gap between the
        // two is nothing but the allocator's alignment padding, nobody's data - becomes one longer span.  (A partial fill as the last
        // span - zero(counters + CNT_POOL, 8) - must never be stretched over its neighbours' live bytes.)
        if (fresh && !overlap && n_fs > 0 && fs_fresh_[n_fs - 1] && fs_.v[n_fs - 1] == vv) {
            char *hi = (char *)fs_.p[n_fs - 1] + fs_.n[n_fs - 1];
            if (c >= hi && (size_t)(c - hi) <= 256) { fs_.n[n_fs - 1] = (uint64_t)(c + n - (char *)fs_.p[n_fs - 1]); return; }
        }
        if (overlap || n_fs == FILL_MAX || n >= ((size_t)4000 << 20) * 64) flush_zero();   // (a span's block count must fit 32 bits)
        fs_.p[n_fs] = p; fs_.n[n_fs] = n; fs_.v[n_fs] = vv; fs_fresh_[n_fs] = fresh; n_fs++;
    }
    void zero_alloc(void *p, size_t n) { add_fill(p, 0, n, true); }  // a WHOLE fresh allocation
    void zero(void *p, size_t n) { add_fill(p, 0, n, false); }
    void fill_byte(void *p, int v, size_t n) { add_fill(p, v, n, false); }
    void fill_ff(void *p, size_t n) { add_fill(p, 0xFF, n, false); }
    void launch(int kn, int64_t nblocks, int nthreads, const WS &w) {
        flush_zero();
        if (fail || nblocks <= 0) return;
        dim3 g((unsigned)nblocks), b((unsigned)nthreads);
        switch (kn) {
#define L(KN, name) case KN: hipLaunchKernelGGL(name, g, b, 0, stream, w); break;
            L(KN_CS_RANGES, aasm_k0_cs_ranges) L(KN_SORT, aasm_k1_sort) L(KN_SORT_RANK, aasm_k1_sort_rank) L(KN_SORT_FIX, aasm_k1_sort_fix) L(KN_GATHER_PARTS, aasm_k1_gather_parts)
            L(KN_OV_COUNT, aasm_k2_ov_count) L(KN_OV_MERGE, aasm_k2_ov_merge) L(KN_VCOUNT, aasm_k2_vcount)
            L(KN_VFILL_REC, aasm_k2_vfill_rec) L(KN_VFILL_SLOT, aasm_k2_vfill_slot) L(KN_NSL, aasm_k4_nsl)
            L(KN_ROW_COUNT, aasm_k4_row_count) L(KN_ROW_FILL, aasm_k4_row_fill) L(KN_GRAPH, aasm_k46_graph) L(KN_GRAPH_L, aasm_k46_graph_l) L(KN_REV_FILL, aasm_k6_rev_fill) L(KN_REV_FILL_W, aasm_k6_rev_fill_w) L(KN_REV_FILL_ORD, aasm_k6_rev_fill_ord) L(KN_REV_FILL_ORD_S, aasm_k6_rev_fill_ord_s)
            L(KN_SORT_ROWS_REV, aasm_k6_rev_place) L(KN_REV_HDR, aasm_k6_rev_hdr) L(KN_REV_SWEEP, aasm_k6_rev_sweep) L(KN_FWD_SWEEP, aasm_k5_fwd_sweep) L(KN_REV_SWEEP_G, aasm_k6_rev_sweep_g) L(KN_FWD_SWEEP_G, aasm_k5_fwd_sweep_g)
            L(KN_CHILDREN, aasm_k7_children)
            L(KN_HEAP_CAP, aasm_k7_heap_cap) L(KN_SIDETRACK, aasm_k7_sidetrack) L(KN_SIDETRACK_W, aasm_k7_sidetrack_w) L(KN_HEAP_HDR, aasm_k7_heap_hdr) L(KN_HEAP, aasm_k7_heap) L(KN_HEAP_MW, aasm_k7_heap_mw) L(KN_HEAP_MW8, aasm_k7_heap_mw8) L(KN_HEAP_MW16, aasm_k7_heap_mw16) L(KN_MW_RANK, aasm_k7_mw_rank) L(KN_ENUM, aasm_k8_enum) L(KN_ENUM_S, aasm_k8_enum_s) L(KN_ENUM_HEAP, aasm_k8_enum_heap) L(KN_SELECT, aasm_k9_select)
            L(KN_GATHER_OUT, aasm_k9_gather_out) L(KN_TOPO_COUNT, aasm_k9_topo_count) L(KN_TOPO_FILL, aasm_k9_topo_fill)
            L(KN_CHAIN, aasm_k67_chain) L(KN_CHAIN3, aasm_k67_chain3) L(KN_K7_PREP, aasm_k7_prep) L(KN_TNX, aasm_k9_tnx) L(KN_TNX16, aasm_k9_tnx16) L(KN_TNX16_WG, aasm_k9_tnx16_wg)
            L(KN_SEL_PLAN, aasm_k9_sel_plan) L(KN_SEL_PLANFILL, aasm_k9_sel_planfill) L(KN_SEL_RECOVER, aasm_k9_sel_recover) L(KN_SEL_CLASSIFY, aasm_k9_sel_classify) L(KN_SEL_CONVERT, aasm_k9_sel_convert) L(KN_SEL_FINAL, aasm_k9_sel_final)
#undef L
            default: break;
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) hip_fail("kernel launch", e);
    }
    template <class T> void scan_t(const T *in, int64_t n, int64_t *out) {
        flush_zero();
        if (fail) return;
        if (n <= 0) { zero(out, 8); return; }
        const int64_t nt = cdiv(n, SCAN_TILE);
        int64_t *&scr = on_side ? cx.d_scratch2 : cx.d_scratch;
        size_t &scr_cap = on_side ? cx.d_scratch2_cap : cx.d_scratch_cap;
        if ((size_t)nt + 8 > scr_cap) {
            hipDeviceSynchronize();                                  // rare growth: nothing may still use the old buffer
            if (scr) hipFree(scr);
            scr_cap = (size_t)nt * 2 + 1024;
            hipError_t e = hipMalloc((void **)&scr, scr_cap * 8);
            if (e == hipSuccess) e = hipMemsetAsync(scr, 0, scr_cap * 8, stream);   // (the scan kernel leaves the words zero again)
            if (e != hipSuccess) { scr = nullptr; scr_cap = 0; hip_fail("hipMalloc(scan)", e); return; }
        }
        if (test_dirty_scan && nt > 1) { (void)hipMemsetAsync(scr, 1, 1, stream); test_dirty_scan = false; }   // ticket counter = 1: tile 0 never runs
        hipLaunchKernelGGL(HIP_KERNEL_NAME(aasm_scan_chain<T>), dim3((unsigned)nt), dim3(SCAN_TPB), 0, stream, in, n, out, (unsigned long long *)scr, nt, cx.pinned_dev + SCAN_STALL_SLOT);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) hip_fail("scan launch", e);
    }
    void scan_small(const int32_t *a, int64_t *oa, const int32_t *b, int64_t *ob, int64_t n) {
        flush_zero();
        if (fail) return;
        hipLaunchKernelGGL(aasm_scan_small, dim3(1), dim3(1024), 0, stream, a, oa, b, ob, n);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) hip_fail("scan launch", e);
    }
    void scan_i32(const int32_t *in, int64_t n, int64_t *out) {
        if (n > 0 && n <= SCAN_SMALL_MAX && !test_dirty_scan) scan_small(in, out, nullptr, nullptr, n);
        else scan_t<int32_t>(in, n, out);
    }
    // two arrays of the same length: one launch when they are short
    void scan_i32_pair(const int32_t *a, int64_t *oa, const int32_t *b, int64_t *ob, int64_t n) {
        if (n > 0 && n <= SCAN_SMALL_MAX && !test_dirty_scan) scan_small(a, oa, b, ob, n);
        else { scan_t<int32_t>(a, n, oa); scan_t<int32_t>(b, n, ob); }
    }
    void scan_u8(const uint8_t *in, int64_t n, int64_t *out) { scan_t<uint8_t>(in, n, out); }
    int64_t read_i64(const int64_t *p) { int64_t v = 0; read_i64s({p}, &v); return v; }
    // several scalars, ONE launch and ONE wait: the kernel queues up behind the kernels that produce them and stores into the
    // host-mapped pinned words (kernel completion at the stream sync makes system-scope stores visible to the host)
    void read_i64s(std::initializer_list<const int64_t *> ps, int64_t *out) {
        int n = 0;
        for (auto p : ps) { (void)p; out[n++] = 0; }
        flush_zero();
        if (fail) return;
        ScalarSrc src;
        int i = 0;
        for (auto p : ps) { if (i < 12) src.p[i] = p; i++; }
        for (int j = i; j < 12; j++) src.p[j] = nullptr;
        hipLaunchKernelGGL(aasm_read_scalars, dim3(1), dim3(64), 0, stream, src, n, cx.pinned_dev);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        g_n_stream_syncs++;
        if (e != hipSuccess) { hip_fail("scalar read-back", e); return; }
        if (scan_stalled()) return;
        for (i = 0; i < n; i++) out[i] = cx.pinned[i];
    }
    void d2h(void *dst, const void *src, size_t n) {
        flush_zero();
        if (fail || n == 0) return;
        hipError_t e = hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) hip_fail("hipMemcpy D2H", e);
        scan_stalled();
    }
    void h2d(void *dst, const void *src, size_t n) {
        flush_zero();
        if (fail || n == 0) return;
        hipError_t e = hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) hip_fail("hipMemcpy H2D", e);
    }
    // second stream: fork = side waits for the main stream's work so far; join = main waits for side
    void fork() { flush_zero(); if (fail) return; hipEventRecord(cx.ev_fork, main_stream); hipStreamWaitEvent(cx.side, cx.ev_fork, 0); forked = true; }
    // a second hand-over main -> side later in the pipeline, through an event of its own: re-recording ev_fork while the side stream's first
    // wait on it has not executed yet moved THAT wait to the later record (measured: the forward sweep then ran beside K7 instead of
    // beside the reverse sweep, and K7 took 5.8 ms instead of 4.1)
    void fork_again() { flush_zero(); if (fail) return; hipEventRecord(cx.ev_fork_b, main_stream); hipStreamWaitEvent(cx.side, cx.ev_fork_b, 0); forked = true; }
    void use_side(bool on) { flush_zero(); on_side = on; stream = on ? cx.side : main_stream; }
    void join() { flush_zero(); if (fail || !forked) return; hipEventRecord(cx.ev_join, cx.side); hipStreamWaitEvent(main_stream, cx.ev_join, 0); forked = false; }
    // third stream (the chain class's workgroups): same protocol; it runs no scans, so it needs no scratch of its own
    void fork2() { flush_zero(); if (fail) return; hipEventRecord(cx.ev_fork2, main_stream); hipStreamWaitEvent(cx.side2, cx.ev_fork2, 0); forked2 = true; }
    void use_side2(bool on) { flush_zero(); stream = on ? cx.side2 : main_stream; }
    void join2() { flush_zero(); if (fail || !forked2) return; hipEventRecord(cx.ev_join2, cx.side2); hipStreamWaitEvent(main_stream, cx.ev_join2, 0); forked2 = false; }
    void phase_begin(int ph) { flush_zero(); if (timing && !fail) { hipError_t e = hipEventRecord(cx.ev_b[ph], stream); if (e != hipSuccess) hip_fail("hipEventRecord", e); phase_used[ph] = true; } }
    void phase_end(int ph) { flush_zero(); if (timing && !fail) { hipError_t e = hipEventRecord(cx.ev_e[ph], stream); if (e != hipSuccess) hip_fail("hipEventRecord", e); } }
};

}  // namespace aasm

using namespace aasm;
static inline bool host_coord_ok(int64_t x) { return x >= 0 && x < AASM_COORD_LIMIT; }

struct aasm_result {
    int device;
    uint64_t generation;
    WS w;
    PipelineSizes sz;
    aasm_stats stats;
    std::map<std::string, std::pair<void *, size_t>> named;
    hipStream_t stream;
};

// record (index in the batch handed to the failing solve) whose cs tag the device rejected
static thread_local int64_t g_bad_record = -1;

static int solve_on_device(DevCtx &cx, const aasm_batch_in &dev_in, const aasm_opts &opts, hipStream_t stream, aasm_result **res_out,
                           GpuBackend **be_out) {
    GpuBackend *be = *be_out;
    const bool timing = opts.collect_timing != 0;
    aasm_result *res = new aasm_result();
    res->device = cx.device; res->stream = stream;
    std::memset(&res->stats, 0, sizeof(res->stats));
    if (timing) hipEventRecord(cx.ev_t0, stream);
    be->test_dirty_scan = (opts.reserved[2] & 4) != 0;              // test hook: the next scan finds a ticket counter an aborted launch left behind
    int rc = run_pipeline(*be, dev_in, opts, res->w, res->sz);
    be->flush_zero();
    if (timing) hipEventRecord(cx.ev_t1, stream);
    be->join();
    be->join2();
    hipError_t e = hipStreamSynchronize(stream);
    if (e == hipSuccess) e = hipStreamSynchronize(cx.side);
    if (e == hipSuccess) e = hipStreamSynchronize(cx.side2);
    be->scan_stalled();
    if (rc == AASM_OK && be->failed()) rc = be->oom() ? AASM_E_NOMEM : AASM_E_HIP;
    if (rc == AASM_OK && e != hipSuccess) { set_last_error(hip_err("pipeline", e)); rc = AASM_E_HIP; }
    if (rc == AASM_E_PARSE) {
        g_bad_record = res->sz.bad_record;
        set_last_error("malformed cs:Z tag in record " + std::to_string(res->sz.bad_record) + " of the batch");
    }
    if (rc == AASM_E_HIP || rc == AASM_E_NOMEM || be->failed() || cx.pinned[SCAN_STALL_SLOT] != 0) {
        // a scan that did not run to its end (failed launch, aborted kernel) leaves tickets / tile words behind, and the single-pass
        // scan of the NEXT solve on this context relies on finding them zero: wipe both scratch buffers before anybody else comes
        // (whatever code the solve itself ends with: a raised stall flag left behind would fail every later solve on the context)
        (void)hipDeviceSynchronize();
        if (cx.d_scratch) (void)hipMemset(cx.d_scratch, 0, cx.d_scratch_cap * 8);
        if (cx.d_scratch2) (void)hipMemset(cx.d_scratch2, 0, cx.d_scratch2_cap * 8);
        cx.pinned[SCAN_STALL_SLOT] = 0;
        (void)hipGetLastError();
    }
    if (rc != AASM_OK) { delete res; return rc; }
    if (timing) {
        for (int i = 0; i < AASM_N_PHASES; i++)
            if (be->phase_used[i]) { float ms = 0; if (hipEventElapsedTime(&ms, cx.ev_b[i], cx.ev_e[i]) == hipSuccess) res->stats.phase_ms[i] = ms; }
        float ms = 0;
        if (hipEventElapsedTime(&ms, cx.ev_t0, cx.ev_t1) == hipSuccess) res->stats.total_ms = ms;
    }
    {   // counters are tiny: read them now so stats are available without a full fetch
        int64_t cnt[CNT_N];
        std::memset(cnt, 0, sizeof(cnt));
        be->d2h(cnt, res->w.counters, sizeof(cnt));
        aasm_stats &st = res->stats;
        st.n_vertices = res->sz.VT; st.n_edges = res->sz.ET;
        st.n_pairs = res->sz.S > 0 ? be->read_i64(res->w.ov_rank + res->sz.S) : 0;
        st.n_heap_nodes = cnt[CNT_HEAPNODES]; st.n_paths_found = cnt[CNT_PATHS]; st.n_paths_converted = cnt[CNT_CONVERTED];
        st.n_unconnectable = cnt[CNT_UNCONN]; st.range_steps = cnt[CNT_RANGE_STEPS];
        st.ispr_edges = cnt[CNT_ISPR_E]; st.ispr_vertices = cnt[CNT_ISPR_V]; st.path_edges = cnt[CNT_PATH_E];
        st.out_elems = cnt[CNT_OUT_E]; st.pq_pushes = cnt[CNT_PQ_PUSH];
        if (be->failed()) { delete res; return AASM_E_HIP; }
    }
    res->stats.device_bytes = (int64_t)be->bytes;
    res->generation = cx.generation;
    res->named = be->named;
    *res_out = res;
    return AASM_OK;
}

extern "C" {

int aasm_abi_version(void) { return AASM_ABI_VERSION; }

int aasm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int aasm_init(int device) { return ctx_init(device); }

int aasm_solve_device(const aasm_batch_in *dev_in, const aasm_opts *opts, void *stream, aasm_result **res) {
    if (!dev_in || !res) return AASM_E_INVAL;
    aasm_opts o;
    std::memset(&o, 0, sizeof(o));
    if (opts) o = *opts;
    int rc = ctx_init(o.device);
    if (rc != AASM_OK) return rc;
    DevCtx &cx = g_ctx[o.device];
    std::lock_guard<std::mutex> lk(cx.mu);
    hipSetDevice(o.device);
    hipStream_t s = stream ? (hipStream_t)stream : cx.stream;
    GpuBackend be(cx, s, o.collect_timing != 0);
    GpuBackend *bp = &be;
    return solve_on_device(cx, *dev_in, o, s, res, &bp);
}

int aasm_result_stats(const aasm_result *res, aasm_stats *stats) {
    if (!res || !stats) return AASM_E_INVAL;
    *stats = res->stats;
    return AASM_OK;
}

// minimal backend view for fetch (D2H only)
namespace {
#define AASM_STAGE_BYTES ((size_t)16 << 20)
struct FetchBackend {
    DevCtx &cx; hipStream_t stream; bool fail = false;
    void d2h(void *dst, const void *src, size_t n) {
        if (fail || n == 0) return;
        hipError_t e = hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) { set_last_error(hip_err("hipMemcpy D2H", e)); fail = true; }
    }
    // Large result arrays: a copy into pageable memory goes through the runtime's own staging at a few GB/s and
    // first-touches every destination page on one thread.  Here the DMA lands in two pinned 16 MB chunks in turn
    // while host threads move the chunk before it into the caller's array (page faults spread over the threads).
    void d2h_big(void *dst, const void *src, size_t n) {
        if (fail || n == 0) return;
        if (n < (AASM_STAGE_BYTES >> 2)) { d2h(dst, src, n); return; }
        for (int i = 0; i < 2; i++)
            if (!cx.stage[i] && hipHostMalloc((void **)&cx.stage[i], AASM_STAGE_BYTES) != hipSuccess) { (void)hipGetLastError(); cx.stage[i] = nullptr; d2h(dst, src, n); return; }
        const size_t nchunks = (n + AASM_STAGE_BYTES - 1) / AASM_STAGE_BYTES;
        const int T = std::max(1, std::min(host_threads(), 16));
        auto issue = [&](size_t c) {
            const size_t off = c * AASM_STAGE_BYTES, len = std::min(AASM_STAGE_BYTES, n - off);
            return hipMemcpyAsync(cx.stage[c & 1], (const char *)src + off, len, hipMemcpyDeviceToHost, stream);
        };
        hipError_t e = issue(0);
        for (size_t c = 0; c < nchunks && e == hipSuccess; c++) {
            e = hipStreamSynchronize(stream);                        // chunk c has landed
            if (e != hipSuccess) break;
            if (c + 1 < nchunks) e = issue(c + 1);                   // the next DMA runs beside the host copy of this one
            const size_t off = c * AASM_STAGE_BYTES, len = std::min(AASM_STAGE_BYTES, n - off);
            const char *from = cx.stage[c & 1];
            char *to = (char *)dst + off;
            std::vector<std::thread> th;
            for (int t = 1; t < T; t++) th.emplace_back([=] { const size_t a = len * t / T, b = len * (t + 1) / T; std::memcpy(to + a, from + a, b - a); });
            std::memcpy(to, from, len / T);
            for (auto &x : th) x.join();
        }
        if (e != hipSuccess) { set_last_error(hip_err("hipMemcpy D2H", e)); fail = true; }
    }
};
}

int aasm_result_fetch(aasm_result *res, aasm_batch_out *out) {
    if (!res || !out) return AASM_E_INVAL;
    DevCtx &cx = g_ctx[res->device];
    std::lock_guard<std::mutex> lk(cx.mu);
    if (res->generation != cx.generation) { set_last_error("result was invalidated by a later solve on the same device"); return AASM_E_INVAL; }
    hipSetDevice(res->device);
    FetchBackend fb{cx, res->stream};
    int rc = fetch_results(fb, res->w, res->sz, out);
    if (fb.fail) { aasm_free_out(out); return AASM_E_HIP; }
    if (rc != AASM_OK) return rc;
    // keep the device-side timers / sizes
    aasm_stats st = out->stats;
    std::memcpy(st.phase_ms, res->stats.phase_ms, sizeof(st.phase_ms));
    st.total_ms = res->stats.total_ms; st.device_bytes = res->stats.device_bytes;
    std::memcpy(st.reserved_f, res->stats.reserved_f, sizeof(st.reserved_f));
    out->stats = st;
    return AASM_OK;
}

void aasm_result_free(aasm_result *res) { delete res; }

void aasm_free_out(aasm_batch_out *out) {
    if (!out) return;
    free(out->main_off); free(out->alt_off); free(out->all_path_off); free(out->all_elem_off);
    free(out->main_elems); free(out->alt_elems); free(out->all_elems); free(out->ctg_status);
    std::memset(out, 0, sizeof(*out));
}

// Test entry (row T1): evaluates the device's PafDistance predicates on n pairs of host tuples.
int aasm_debug_predicates(const int64_t *a, const int64_t *b, int64_t n, uint8_t *out, int device) {
    if (!a || !b || !out || n <= 0) return AASM_E_INVAL;
    int rc = ctx_init(device);
    if (rc != AASM_OK) return rc;
    hipSetDevice(device);
    int64_t *da = nullptr, *db = nullptr;
    uint8_t *dout = nullptr;
    hipError_t e = hipMalloc((void **)&da, (size_t)n * 40);
    if (e == hipSuccess) e = hipMalloc((void **)&db, (size_t)n * 40);
    if (e == hipSuccess) e = hipMalloc((void **)&dout, (size_t)n);
    if (e == hipSuccess) e = hipMemcpy(da, a, (size_t)n * 40, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(db, b, (size_t)n * 40, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(aasm_t1_predicates, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, g_ctx[device].stream, da, db, n, dout);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(g_ctx[device].stream);
    if (e == hipSuccess) e = hipMemcpy(out, dout, (size_t)n, hipMemcpyDeviceToHost);
    hipFree(da); hipFree(db); hipFree(dout);
    if (e != hipSuccess) { set_last_error(hip_err("aasm_debug_predicates", e)); return AASM_E_HIP; }
    return AASM_OK;
}

// Test entry (hazard B1): K1's std::sort replay alone, on arbitrary keys.  rec_off[n_contigs + 1] from 0; perm_out[r] per
// record = the input index (relative to its contig) that std::sort leaves at sorted position r.  depth_test as aasm_opts.reserved[2] bits 8-15.
int aasm_debug_sort_replay(const int64_t *rec_off, int64_t n_contigs, const int64_t *qs, const int64_t *qe, int32_t *perm_out, int depth_test, int device) {
    if (!rec_off || !qs || !qe || !perm_out || n_contigs <= 0 || rec_off[0] != 0) return AASM_E_INVAL;
    for (int64_t c = 0; c < n_contigs; c++) if (rec_off[c + 1] < rec_off[c]) return AASM_E_INVAL;
    const int64_t R = rec_off[n_contigs];
    if (R <= 0 || R > INT32_MAX) return AASM_E_INVAL;
    int rc = ctx_init(device);
    if (rc != AASM_OK) return rc;
    hipSetDevice(device);
    std::vector<void *> dev;
    hipError_t e = hipSuccess;
    auto up = [&](const void *p, size_t bytes, int fill) -> void * {
        void *q = nullptr;
        if (e != hipSuccess) return nullptr;
        if ((e = hipMalloc(&q, bytes ? bytes : 8)) != hipSuccess) return nullptr;
        dev.push_back(q);
        if (p) e = hipMemcpy(q, p, bytes, hipMemcpyHostToDevice);
        else e = hipMemset(q, fill, bytes);
        return q;
    };
    WS w;
    std::memset(&w, 0, sizeof(w));
    w.C = n_contigs; w.R = R; w.R0 = 0; w.sort_depth_test = depth_test;
    w.rec_off = (const int64_t *)up(rec_off, (size_t)(n_contigs + 1) * 8, 0);
    w.in_qs = (const int64_t *)up(qs, (size_t)R * 8, 0); w.in_qe = (const int64_t *)up(qe, (size_t)R * 8, 0);
    w.s_qs = (int64_t *)up(nullptr, (size_t)R * 8, 0); w.s_qe = (int64_t *)up(nullptr, (size_t)R * 8, 0); w.s_orig = (int32_t *)up(nullptr, (size_t)R * 4, 0);
    w.perm = (int32_t *)up(nullptr, (size_t)R * 4, 0xff);
    std::vector<int32_t> ones((size_t)n_contigs, 1);
    w.dupflag = (int32_t *)up(ones.data(), (size_t)n_contigs * 4, 0);
#if defined(AASM_KPROF)
    w.prof_heap = (int64_t *)up(nullptr, (size_t)n_contigs * 64, 0);
#endif
    if (e == hipSuccess) {
        hipLaunchKernelGGL(aasm_k1_sort_fix, dim3((unsigned)n_contigs), dim3(64), 0, g_ctx[device].stream, w);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(g_ctx[device].stream);
    if (e == hipSuccess) e = hipMemcpy(perm_out, w.perm, (size_t)R * 4, hipMemcpyDeviceToHost);
#if defined(AASM_KPROF)
    if (e == hipSuccess) {                                           // diagnostic build: mean cycles per section over the contigs
        std::vector<int64_t> kp((size_t)n_contigs * 8);
        hipMemcpy(kp.data(), w.prof_heap, kp.size() * 8, hipMemcpyDeviceToHost);
        double m[8] = {0};
        for (int64_t c = 0; c < n_contigs; c++) for (int i = 0; i < 8; i++) m[i] += (double)kp[(size_t)c * 8 + i] / (double)n_contigs;
        fprintf(stderr, "sort_fix sections (mean cycles): load %.0f  A %.0f  B %.0f  C %.0f  global partitions %.0f  wave lifetime %.1f us\n", m[0], m[1], m[2], m[3], m[4], m[7] / 100.0);
    }
#endif
    for (void *q : dev) hipFree(q);
    if (e != hipSuccess) { set_last_error(hip_err("aasm_debug_sort_replay", e)); return AASM_E_HIP; }
    return AASM_OK;
}

// dijkstra (k_shortest_walks.hpp:69-87) over a batch of graphs; host pointers in and out
int aasm_sssp_dijkstra(int64_t n_graphs, const int64_t *g_voff, const int64_t *rowptr, const int32_t *col, const int64_t *w5,
                       const int32_t *src, int64_t *d5, int32_t *prev, int device) {
    if (n_graphs <= 0 || !g_voff || !rowptr || !col || !w5 || !src || !d5 || !prev) return AASM_E_INVAL;
    int rc = ctx_init(device);
    if (rc != AASM_OK) return rc;
    hipSetDevice(device);
    const int64_t VT = g_voff[n_graphs], ET = rowptr[VT];
    if (VT <= 0 || g_voff[0] != 0 || rowptr[0] != 0) { set_last_error("inconsistent graph offsets"); return AASM_E_INVAL; }
    std::vector<int64_t> hoff((size_t)n_graphs + 1, 0);
    for (int64_t g = 0; g < n_graphs; g++) {
        const int64_t v0 = g_voff[g], v1 = g_voff[g + 1];
        if (v1 <= v0 || src[g] < 0 || src[g] >= v1 - v0) { set_last_error("graph " + std::to_string(g) + ": empty, or source outside it"); return AASM_E_INVAL; }
        hoff[(size_t)g + 1] = hoff[(size_t)g] + (rowptr[v1] - rowptr[v0]) + 2;   // every successful relaxation pushes once: <= E + 1 entries
        for (int64_t e = rowptr[v0]; e < rowptr[v1]; e++)
            if (col[e] < 0 || col[e] >= v1 - v0) { set_last_error("graph " + std::to_string(g) + ": edge head outside the graph"); return AASM_E_INVAL; }
    }
    for (int64_t e = 0; e < ET; e++)
        if (!(host_coord_ok(w5[5 * e] + AASM_COORD_LIMIT / 2) && host_coord_ok(w5[5 * e + 1] + AASM_COORD_LIMIT / 2)) || w5[5 * e + 2] < 0 || w5[5 * e + 2] > 2 || w5[5 * e + 3] < 0 || w5[5 * e + 3] > 1 ||
            w5[5 * e + 4] < 0 || w5[5 * e + 4] > 1 || w5[5 * e] + w5[5 * e + 1] < 0) {
            set_last_error("edge " + std::to_string(e) + ": weight outside the supported range (score sum >= 0, |scores| < 2^39, anom 0..2, mapq counts 0..1)");
            return AASM_E_OVERFLOW;
        }
    std::vector<void *> dev;
    bool ok = true;
    hipError_t e = hipSuccess;
    auto up = [&](const void *p, size_t bytes) -> void * {
        void *q = nullptr;
        if (!ok) return nullptr;
        if ((e = hipMalloc(&q, bytes ? bytes : 8)) != hipSuccess) { ok = false; return nullptr; }
        dev.push_back(q);
        if (p && bytes && (e = hipMemcpy(q, p, bytes, hipMemcpyHostToDevice)) != hipSuccess) ok = false;
        return q;
    };
    const int64_t *d_voff = (const int64_t *)up(g_voff, (size_t)(n_graphs + 1) * 8), *d_rowptr = (const int64_t *)up(rowptr, (size_t)(VT + 1) * 8);
    const int32_t *d_col = (const int32_t *)up(col, (size_t)ET * 4), *d_src = (const int32_t *)up(src, (size_t)n_graphs * 4);
    const int64_t *d_w = (const int64_t *)up(w5, (size_t)ET * 40);
    Dist *d_d = (Dist *)up(nullptr, (size_t)VT * sizeof(Dist));
    int32_t *d_prv = (int32_t *)up(nullptr, (size_t)VT * 4);
    std::vector<Dist> hd((size_t)VT);
    // Heap capacity.  With a monotone order every successful relaxation pushes once (<= E + 1 entries), but CALC_SUM's third
    // key (the mapq ratio) is not monotone under addition and the reference re-expands a vertex whenever its distance
    // improves, so stale entries of one edge can pile up: a graph whose heap overflows is run again with 4x, 16x, 64x the room.
    int64_t overflowed = -1;
    for (int64_t mult = 1; ok && mult <= 64; mult *= 4) {
        std::vector<int64_t> ho((size_t)n_graphs + 1, 0);
        for (int64_t g = 0; g < n_graphs; g++) ho[(size_t)g + 1] = ho[(size_t)g] + mult * (hoff[(size_t)g + 1] - hoff[(size_t)g]);
        const size_t n_dev0 = dev.size();
        const int64_t *d_hoff = (const int64_t *)up(ho.data(), (size_t)(n_graphs + 1) * 8);
        DjEnt *d_heap = (DjEnt *)up(nullptr, (size_t)ho[(size_t)n_graphs] * sizeof(DjEnt));
        if (ok) {
            hipLaunchKernelGGL(aasm_sssp_dijkstra_kernel, dim3((unsigned)n_graphs), dim3(64), 0, g_ctx[device].stream, n_graphs, d_voff, d_rowptr, d_col, d_w, d_src, d_d, d_prv, d_heap, d_hoff);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(g_ctx[device].stream);
            if (e == hipSuccess) e = hipMemcpy(hd.data(), d_d, (size_t)VT * sizeof(Dist), hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(prev, d_prv, (size_t)VT * 4, hipMemcpyDeviceToHost);
            ok = e == hipSuccess;
        }
        while (dev.size() > n_dev0) { hipFree(dev.back()); dev.pop_back(); }
        overflowed = -1;
        if (ok) for (int64_t g = 0; g < n_graphs; g++) if (prev[g_voff[g] + src[g]] == -2) { overflowed = g; break; }
        if (overflowed < 0) break;
    }
    for (void *q : dev) hipFree(q);
    if (!ok) { set_last_error(hip_err("aasm_sssp_dijkstra", e)); return e == hipErrorOutOfMemory ? AASM_E_NOMEM : AASM_E_HIP; }
    if (overflowed >= 0) { set_last_error("graph " + std::to_string(overflowed) + ": dijkstra heap capacity exceeded at 64 x (E + 2) entries"); return AASM_E_OVERFLOW; }
    for (int64_t v = 0; v < VT; v++) { d5[5 * v] = hd[(size_t)v].qry; d5[5 * v + 1] = hd[(size_t)v].ref; d5[5 * v + 2] = hd[(size_t)v].anom; d5[5 * v + 3] = hd[(size_t)v].qnz; d5[5 * v + 4] = hd[(size_t)v].qtot; }
    return AASM_OK;
}

// Dial's bucketed BFS (k_weighted_bfs.hpp:16-37) over a batch of graphs; host pointers in and out
int aasm_sssp_dial(int64_t n_graphs, const int64_t *g_voff, const int64_t *rowptr, const int32_t *col, const int32_t *cost,
                   const int32_t *src, int lim, int64_t *dist, int64_t *pre, int device) {
    if (n_graphs <= 0 || !g_voff || !rowptr || !col || !cost || !src || !dist || !pre) return AASM_E_INVAL;
    if (lim < 0 || lim + 1 > DIAL_MAXB) { set_last_error("aasm_sssp_dial: lim outside 0 .. 7"); return AASM_E_INVAL; }
    int rc = ctx_init(device);
    if (rc != AASM_OK) return rc;
    hipSetDevice(device);
    const int nb = lim + 1;
    // the offsets first, before anything is read THROUGH them: graph starts strictly increasing from 0, row pointers non-decreasing from 0
    if (g_voff[0] != 0) { set_last_error("inconsistent graph offsets"); return AASM_E_INVAL; }
    for (int64_t g = 0; g < n_graphs; g++)
        if (g_voff[g + 1] <= g_voff[g]) { set_last_error("graph " + std::to_string(g) + ": empty, or graph offsets not increasing"); return AASM_E_INVAL; }
    const int64_t VT = g_voff[n_graphs];
    if (VT <= 0 || rowptr[0] != 0) { set_last_error("inconsistent graph offsets"); return AASM_E_INVAL; }
    for (int64_t v = 0; v < VT; v++)
        if (rowptr[v + 1] < rowptr[v]) { set_last_error("row pointers decrease at vertex " + std::to_string(v)); return AASM_E_INVAL; }
    const int64_t ET = rowptr[VT];
    std::vector<int64_t> soff((size_t)n_graphs + 1, 0);
    for (int64_t g = 0; g < n_graphs; g++) {
        const int64_t v0 = g_voff[g], v1 = g_voff[g + 1];
        if (v1 <= v0 || src[g] < 0 || src[g] >= v1 - v0) { set_last_error("graph " + std::to_string(g) + ": empty, or source outside it"); return AASM_E_INVAL; }
        for (int64_t e = rowptr[v0]; e < rowptr[v1]; e++) {
            if (col[e] < 0 || col[e] >= v1 - v0) { set_last_error("graph " + std::to_string(g) + ": edge head outside the graph"); return AASM_E_INVAL; }
            if (cost[e] < 0 || cost[e] > lim) { set_last_error("edge " + std::to_string(e) + ": cost outside 0 .. lim (the reference asserts it, k_weighted_bfs.hpp:27)"); return AASM_E_INVAL; }
        }
        // a vertex is pushed once per successful relaxation: its distance falls by at least one each time and by at most lim in all
        // after the first (a later pop has d' >= d), so <= lim + 1 pushes per vertex - and never more than one per edge, plus the source
        const int64_t E = rowptr[v1] - rowptr[v0], by_v = (v1 - v0) * (int64_t)nb;
        const int64_t per = ((E + 1 < by_v ? E + 1 : by_v) + DIAL_WIN + 63) / 64 * 64;
        soff[(size_t)g + 1] = soff[(size_t)g] + per * nb;
    }
    std::vector<void *> dev;
    bool ok = true;
    hipError_t e = hipSuccess;
    auto up = [&](const void *p, size_t bytes) -> void * {
        void *q = nullptr;
        if (!ok) return nullptr;
        if ((e = hipMalloc(&q, bytes ? bytes : 8)) != hipSuccess) { ok = false; return nullptr; }
        dev.push_back(q);
        if (p && bytes && (e = hipMemcpy(q, p, bytes, hipMemcpyHostToDevice)) != hipSuccess) ok = false;
        return q;
    };
    const int64_t *d_voff = (const int64_t *)up(g_voff, (size_t)(n_graphs + 1) * 8), *d_rowptr = (const int64_t *)up(rowptr, (size_t)(VT + 1) * 8);
    const int32_t *d_col = (const int32_t *)up(col, (size_t)ET * 4), *d_cost = (const int32_t *)up(cost, (size_t)ET * 4), *d_src = (const int32_t *)up(src, (size_t)n_graphs * 4);
    const int64_t *d_soff = (const int64_t *)up(soff.data(), (size_t)(n_graphs + 1) * 8);
    int64_t *d_dist = (int64_t *)up(nullptr, (size_t)VT * 8), *d_pre = (int64_t *)up(nullptr, (size_t)VT * 8);
    int32_t *d_spill = (int32_t *)up(nullptr, (size_t)soff[(size_t)n_graphs] * 4);
    if (ok) {
        hipLaunchKernelGGL(aasm_sssp_dial_kernel, dim3((unsigned)n_graphs), dim3(64), 0, g_ctx[device].stream, n_graphs, d_voff, d_rowptr, d_col, d_cost, d_src, (int32_t)nb, d_dist, d_pre, d_spill, d_soff);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(g_ctx[device].stream);
        if (e == hipSuccess) e = hipMemcpy(dist, d_dist, (size_t)VT * 8, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(pre, d_pre, (size_t)VT * 8, hipMemcpyDeviceToHost);
        ok = e == hipSuccess;
    }
    for (void *q : dev) hipFree(q);
    if (!ok) { set_last_error(hip_err("aasm_sssp_dial", e)); return e == hipErrorOutOfMemory ? AASM_E_NOMEM : AASM_E_HIP; }
    for (int64_t g = 0; g < n_graphs; g++)
        if (pre[g_voff[g] + src[g]] == -2) { set_last_error("graph " + std::to_string(g) + ": bucket capacity exceeded (must not happen)"); return AASM_E_INTERNAL; }
    return AASM_OK;
}

int64_t aasm_debug_counter(const char *name) {
    if (!name) return -1;
    const std::string n(name);
    if (n == "range_splits") return g_n_range_splits.load();
    if (n == "device_mallocs") return g_n_device_mallocs.load();
    if (n == "stream_syncs") return g_n_stream_syncs.load();
    return -1;
}

int64_t aasm_debug_fetch(aasm_result *res, const char *name, void *dst, int64_t dst_bytes) {
    if (!res || !name) return AASM_E_INVAL;
    DevCtx &cx = g_ctx[res->device];
    std::lock_guard<std::mutex> lk(cx.mu);
    if (res->generation != cx.generation) return AASM_E_INVAL;
    auto it = res->named.find(name);
    if (it == res->named.end()) return AASM_E_INVAL;
    if (dst) {
        hipSetDevice(res->device);
        size_t n = std::min<size_t>((size_t)dst_bytes, it->second.second);
        hipError_t e = hipMemcpy(dst, it->second.first, n, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { set_last_error(hip_err("debug fetch", e)); return AASM_E_HIP; }
    }
    return (int64_t)it->second.second;
}

// Upload a host batch once (bench / repeated solves): returns a device view.
// The upload lives in plain hipMalloc memory owned by the handle (not in the arena).
struct aasm_upload { int device; std::vector<void *> ptrs; aasm_batch_in view; };

static int upload_range(const aasm_batch_in *in, int64_t c0, int64_t c1, int device, aasm_upload **up_out, aasm_batch_in *dev_view) {
    if (!in || !up_out || !dev_view || c0 < 0 || c1 > in->n_contigs || c0 >= c1) return AASM_E_INVAL;
    int rc = ctx_init(device);
    if (rc != AASM_OK) return rc;
    hipSetDevice(device);
    g_ctx[device].input_arrived.store(1);
    aasm_upload *up = new aasm_upload();
    up->device = device;
    bool ok = true, oom = false;
    std::string why = "the batch carries neither match ranges nor cs tags";
    auto put = [&](const void *src, size_t bytes) -> void * {
        void *p = nullptr;
        if (!ok) return nullptr;
        hipError_t e = hipMalloc(&p, bytes ? bytes : 8);
        if (e != hipSuccess) { ok = false; oom = (e == hipErrorOutOfMemory); (void)hipGetLastError(); why = hip_err("hipMalloc(upload)", e); return nullptr; }
        up->ptrs.push_back(p);
        if (bytes && (e = hipMemcpy(p, src, bytes, hipMemcpyHostToDevice)) != hipSuccess) { ok = false; why = hip_err("hipMemcpy H2D", e); }
        return p;
    };
    // rebase the offsets of the contig range [c0, c1) to start at 0
    const int64_t C = c1 - c0, r0 = in->ctg_rec_off[c0], r1 = in->ctg_rec_off[c1], R = r1 - r0;
    const int64_t g0 = in->rec_rng_off[r0], g1 = in->rec_rng_off[r1], G = g1 - g0;
    std::vector<int64_t> coff(C + 1), roff(R + 1);
    for (int64_t c = 0; c <= C; c++) coff[c] = in->ctg_rec_off[c0 + c] - r0;
    for (int64_t r = 0; r <= R; r++) roff[r] = in->rec_rng_off[r0 + r] - g0;
    aasm_batch_in &v = up->view;
    v.n_contigs = C; v.n_records = R; v.n_ranges = G;
    v.ctg_rec_off = (const int64_t *)put(coff.data(), (C + 1) * 8);
    v.qry_str = (const int64_t *)put(in->qry_str + r0, R * 8);
    v.qry_end = (const int64_t *)put(in->qry_end + r0, R * 8);
    v.ref_str = (const int64_t *)put(in->ref_str + r0, R * 8);
    v.ref_end = (const int64_t *)put(in->ref_end + r0, R * 8);
    v.qry_total = (const int64_t *)put(in->qry_total + r0, R * 8);
    v.ref_chr = (const int32_t *)put(in->ref_chr + r0, R * 4);
    v.aln_fwd = (const uint8_t *)put(in->aln_fwd + r0, R);
    v.map_qul = (const uint8_t *)put(in->map_qul + r0, R);
    v.rec_rng_off = (const int64_t *)put(roff.data(), (R + 1) * 8);
    if (in->rng_qry_l) {
        v.rng_qry_l = (const int64_t *)put(in->rng_qry_l + g0, G * 8);
        v.rng_qry_r = (const int64_t *)put(in->rng_qry_r + g0, G * 8);
        v.rng_ref_l = (const int64_t *)put(in->rng_ref_l + g0, G * 8);
    } else if (in->cs_text && in->rec_cs_off) {                     // the device parses the cs tags (aasm_k0_cs_ranges)
        const int64_t t0 = in->rec_cs_off[r0], t1 = in->rec_cs_off[r1];
        std::vector<int64_t> toff(R + 1);
        for (int64_t r = 0; r <= R; r++) toff[r] = in->rec_cs_off[r0 + r] - t0;
        v.cs_text = (const char *)put(in->cs_text + t0, (size_t)(t1 - t0));
        v.rec_cs_off = (const int64_t *)put(toff.data(), (R + 1) * 8);
    } else ok = false;
    if (!ok) {
        for (void *p : up->ptrs) hipFree(p);
        delete up;
        set_last_error("device upload failed: " + why);
        return oom ? AASM_E_NOMEM : AASM_E_HIP;
    }
    *dev_view = up->view;
    *up_out = up;
    return AASM_OK;
}
int aasm_upload_batch(const aasm_batch_in *in, int device, aasm_upload **up_out, aasm_batch_in *dev_view) {
    if (!in) return AASM_E_INVAL;
    return upload_range(in, 0, in->n_contigs, device, up_out, dev_view);
}
void aasm_upload_free(aasm_upload *up) {
    if (!up) return;
    hipSetDevice(up->device);
    for (void *p : up->ptrs) hipFree(p);
    delete up;
}

// concatenate per-range results (contiguous contig ranges cut[d]..cut[d+1]) in contig order
static void concat_parts(std::vector<aasm_batch_out> &parts, const std::vector<int64_t> &cut, int64_t C, aasm_batch_out *out) {
    const int n_devices = (int)parts.size();
    std::memset(out, 0, sizeof(*out));
    out->n_contigs = C;
    int64_t nm = 0, na = 0, np = 0, ne = 0;
    for (auto &p : parts) { nm += p.main_off[p.n_contigs]; na += p.alt_off[p.n_contigs]; np += p.n_all_paths; ne += p.all_elem_off[p.n_all_paths]; }
    out->main_off = (int64_t *)calloc(C + 1, 8); out->alt_off = (int64_t *)calloc(C + 1, 8); out->all_path_off = (int64_t *)calloc(C + 1, 8);
    out->all_elem_off = (int64_t *)calloc(np + 1, 8); out->ctg_status = (int32_t *)calloc(C + 1, 4);
    out->main_elems = (aasm_out_elem *)calloc(nm + 1, sizeof(aasm_out_elem));
    out->alt_elems = (aasm_out_elem *)calloc(na + 1, sizeof(aasm_out_elem));
    out->all_elems = (aasm_out_elem *)calloc(ne + 1, sizeof(aasm_out_elem));
    out->n_all_paths = np;
    int64_t bm = 0, ba = 0, bp = 0, be_ = 0;
    for (int d = 0; d < n_devices; d++) {
        aasm_batch_out &p = parts[d];
        const int64_t pc = p.n_contigs, c0 = cut[d];
        for (int64_t c = 0; c < pc; c++) {
            out->main_off[c0 + c + 1] = bm + p.main_off[c + 1];
            out->alt_off[c0 + c + 1] = ba + p.alt_off[c + 1];
            out->all_path_off[c0 + c + 1] = bp + p.all_path_off[c + 1];
            out->ctg_status[c0 + c] = p.ctg_status[c];
        }
        for (int64_t q = 0; q < p.n_all_paths; q++) out->all_elem_off[bp + q + 1] = be_ + p.all_elem_off[q + 1];
        std::memcpy(out->main_elems + bm, p.main_elems, sizeof(aasm_out_elem) * (size_t)p.main_off[pc]);
        std::memcpy(out->alt_elems + ba, p.alt_elems, sizeof(aasm_out_elem) * (size_t)p.alt_off[pc]);
        std::memcpy(out->all_elems + be_, p.all_elems, sizeof(aasm_out_elem) * (size_t)p.all_elem_off[p.n_all_paths]);
        bm += p.main_off[pc]; ba += p.alt_off[pc]; bp += p.n_all_paths; be_ += p.all_elem_off[p.n_all_paths];
        aasm_stats &a = out->stats; const aasm_stats &b = p.stats;
        a.n_vertices += b.n_vertices; a.n_pairs += b.n_pairs; a.n_edges += b.n_edges; a.n_heap_nodes += b.n_heap_nodes;
        a.n_paths_found += b.n_paths_found; a.n_paths_converted += b.n_paths_converted; a.n_unconnectable += b.n_unconnectable;
        a.n_internal_errors += b.n_internal_errors; a.n_single += b.n_single; a.range_steps += b.range_steps;
        a.ispr_edges += b.ispr_edges; a.ispr_vertices += b.ispr_vertices; a.path_edges += b.path_edges; a.out_elems += b.out_elems; a.pq_pushes += b.pq_pushes;
        if (b.device_bytes > a.device_bytes) a.device_bytes = b.device_bytes;
        for (int i = 0; i < AASM_N_PHASES; i++) if (b.phase_ms[i] > a.phase_ms[i]) a.phase_ms[i] = b.phase_ms[i];
        if (b.total_ms > a.total_ms) a.total_ms = b.total_ms;
        for (int i = 0; i < 3; i++) if (b.reserved_f[i] > a.reserved_f[i]) a.reserved_f[i] = b.reserved_f[i];
        aasm_free_out(&p);
    }
}

static int validate_batch(const aasm_batch_in *in) {
    if (!in || in->n_contigs <= 0 || !in->ctg_rec_off || in->ctg_rec_off[0] != 0 || in->ctg_rec_off[in->n_contigs] != in->n_records) {
        set_last_error("inconsistent contig offsets");
        return AASM_E_INVAL;
    }
    for (int64_t c = 0; c < in->n_contigs; c++)
        if (in->ctg_rec_off[c + 1] <= in->ctg_rec_off[c]) { set_last_error("empty contig"); return AASM_E_INVAL; }
    if (!in->rec_rng_off || (!in->rng_qry_l && !(in->cs_text && in->rec_cs_off))) {
        set_last_error("the batch carries neither match ranges (rng_*) nor cs tags (cs_text / rec_cs_off)");
        return AASM_E_INVAL;
    }
    // ranges the device's narrowed fields are exact for (aasm_kernels.h AASM_COORD_LIMIT; the device repeats the
    // coordinate check per contig for batches that are handed over already resident)
    for (int64_t c = 0; c < in->n_contigs; c++)
        if (in->ctg_rec_off[c + 1] - in->ctg_rec_off[c] > (int64_t)INT32_MAX - 64) {
            set_last_error("contig " + std::to_string(c) + " has more than 2^31 records");
            return AASM_E_OVERFLOW;
        }
    for (int64_t r = 0; r < in->n_records; r++)
        if (!(host_coord_ok(in->qry_str[r]) && host_coord_ok(in->qry_end[r]) && host_coord_ok(in->ref_str[r]) && host_coord_ok(in->ref_end[r]) && host_coord_ok(in->qry_total[r]))) {
            set_last_error("record " + std::to_string(r) + ": coordinate outside [0, 2^40) (qry_str / qry_end / ref_str / ref_end / qry_total)");
            return AASM_E_OVERFLOW;
        }
    return AASM_OK;
}



// Make the device context and `bytes` of workspace arena ahead of the first solve (a fresh process pays the HIP start-up
// and ~25 ms per GB of hipMalloc otherwise inside its first solve): callers run it on a thread of its own while they still
// read their input.  Not finding the memory is not an error here - the solve allocates what it needs (and reports).
int aasm_reserve_workspace(int device, int64_t bytes) {
    int rc = ctx_init(device);
    if (rc != AASM_OK) return rc;
    DevCtx &cx = g_ctx[device];
    std::lock_guard<std::mutex> lk(cx.mu);
    // the input is on its way (or a solve has run): the free-memory figure below no longer leaves room for it, and a solve
    // that holds the lock first would make this call allocate memory nobody uses after it - the solve allocates for itself
    if (cx.input_arrived.load() != 0) return AASM_OK;
    hipSetDevice(device);
    size_t have = 0;
    for (auto &b : cx.blocks) have += b.cap;
    if (bytes <= 0 || have >= (size_t)bytes) return AASM_OK;
    size_t want = (size_t)bytes - have, free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && want > free_b / 2) want = free_b / 2;   // leave room for the input and the results
    want &= ~(size_t)255;
    if (want < ((size_t)64 << 20)) return AASM_OK;
    char *p = nullptr;
    hipError_t e = hipMalloc((void **)&p, want);
    g_n_device_mallocs++;
    if (e != hipSuccess) { (void)hipGetLastError(); return AASM_E_NOMEM; }
    cx.blocks.push_back(ArenaBlock{p, want, 0});
    return AASM_OK;
}

static void ctx_release_arena(int device) {
    DevCtx &cx = g_ctx[device];
    std::lock_guard<std::mutex> lk(cx.mu);
    hipSetDevice(device);
    hipStreamSynchronize(cx.stream);
    for (auto &b : cx.blocks) hipFree(b.p);
    cx.blocks.clear();
    cx.generation++;
}

static int solve_range_once(const aasm_batch_in *in, int64_t c0, int64_t c1, const aasm_opts &o, aasm_batch_out *out) {
    if (o.reserved[1] > 0 && c1 - c0 > o.reserved[1]) {               // test hook (opts.reserved[1]): pretend larger ranges do not fit
        set_last_error("range exceeds the test limit opts.reserved[1]");
        return AASM_E_NOMEM;
    }
    aasm_upload *up = nullptr;
    aasm_batch_in dv;
    auto t0 = std::chrono::steady_clock::now();
    int rc = upload_range(in, c0, c1, o.device, &up, &dv);
    if (rc != AASM_OK) return rc;
    auto t1 = std::chrono::steady_clock::now();
    aasm_result *res = nullptr;
    rc = aasm_solve_device(&dv, &o, nullptr, &res);
    if (rc == AASM_OK) {
        auto t2 = std::chrono::steady_clock::now();
        rc = aasm_result_fetch(res, out);
        auto t3 = std::chrono::steady_clock::now();
        if (rc == AASM_OK) {
            out->stats.reserved_f[0] = std::chrono::duration<float, std::milli>(t1 - t0).count();   // H2D upload
            out->stats.reserved_f[1] = std::chrono::duration<float, std::milli>(t3 - t2).count();   // D2H + pack
            out->stats.reserved_f[2] = std::chrono::duration<float, std::milli>(t2 - t1).count();   // solve wall
        }
    }
    if (rc == AASM_E_PARSE && g_bad_record >= 0 && in->cs_text && in->rec_cs_off) {   // say what the host codec says about that tag
        const int64_t r = in->ctg_rec_off[c0] + g_bad_record;
        const std::string why = cs_error_message(in->cs_text + in->rec_cs_off[r], in->rec_cs_off[r + 1] - in->rec_cs_off[r], in->aln_fwd[r] != 0,
                                                 in->qry_str[r], in->qry_end[r], in->ref_str[r], in->ref_end[r]);
        set_last_error((why.empty() ? std::string("malformed cs:Z tag") : why) + " (record " + std::to_string(r) + ")");
    }
    aasm_result_free(res);
    aasm_upload_free(up);
    return rc;
}

static void add_stats(aasm_stats &a, const aasm_stats &b, bool sum_time) {
    a.n_vertices += b.n_vertices; a.n_pairs += b.n_pairs; a.n_edges += b.n_edges; a.n_heap_nodes += b.n_heap_nodes;
    a.n_paths_found += b.n_paths_found; a.n_paths_converted += b.n_paths_converted; a.n_unconnectable += b.n_unconnectable;
    a.n_internal_errors += b.n_internal_errors; a.n_single += b.n_single; a.range_steps += b.range_steps;
    a.ispr_edges += b.ispr_edges; a.ispr_vertices += b.ispr_vertices; a.path_edges += b.path_edges; a.out_elems += b.out_elems; a.pq_pushes += b.pq_pushes;
    if (b.device_bytes > a.device_bytes) a.device_bytes = b.device_bytes;
    for (int i = 0; i < AASM_N_PHASES; i++) a.phase_ms[i] = sum_time ? a.phase_ms[i] + b.phase_ms[i] : (b.phase_ms[i] > a.phase_ms[i] ? b.phase_ms[i] : a.phase_ms[i]);
    a.total_ms = sum_time ? a.total_ms + b.total_ms : (b.total_ms > a.total_ms ? b.total_ms : a.total_ms);
    for (int i = 0; i < 3; i++) a.reserved_f[i] = sum_time ? a.reserved_f[i] + b.reserved_f[i] : (b.reserved_f[i] > a.reserved_f[i] ? b.reserved_f[i] : a.reserved_f[i]);
}

// A contig range that does not fit in device memory is split in halves (contigs are
// independent) after the arena of the failed attempt has been given back.
static int solve_range(const aasm_batch_in *in, int64_t c0, int64_t c1, const aasm_opts &o, aasm_batch_out *out) {
    int rc = solve_range_once(in, c0, c1, o, out);
    if (rc != AASM_E_NOMEM || c1 - c0 < 2) return rc;                    // only a true allocation failure is worth splitting for
    g_n_range_splits++;
    ctx_release_arena(o.device);
    const int64_t mid = c0 + (c1 - c0) / 2;
    std::vector<aasm_batch_out> parts(2);
    std::memset(&parts[0], 0, sizeof(aasm_batch_out)); std::memset(&parts[1], 0, sizeof(aasm_batch_out));
    rc = solve_range(in, c0, mid, o, &parts[0]);
    if (rc == AASM_OK) rc = solve_range(in, mid, c1, o, &parts[1]);
    if (rc != AASM_OK) { aasm_free_out(&parts[0]); aasm_free_out(&parts[1]); return rc; }
    std::vector<int64_t> cut{0, mid - c0, c1 - c0};
    aasm_stats st;
    std::memset(&st, 0, sizeof(st));
    add_stats(st, parts[0].stats, true); add_stats(st, parts[1].stats, true);
    concat_parts(parts, cut, c1 - c0, out);
    out->stats = st;
    return AASM_OK;
}

// contigs [c0, c1) of the batch: what a caller that streams a file runs per chunk (out covers c1 - c0 contigs)
int aasm_solve_batch_range(const aasm_batch_in *in, int64_t c0, int64_t c1, const aasm_opts *opts, aasm_batch_out *out) {
    if (!in || !out || c0 < 0 || c1 > in->n_contigs || c0 >= c1) return AASM_E_INVAL;
    aasm_opts o;
    std::memset(&o, 0, sizeof(o));
    if (opts) o = *opts;
    if (c0 == 0) { const int rc = validate_batch(in); if (rc != AASM_OK) return rc; }    // (the whole batch is checked once, with its first range)
    return solve_range(in, c0, c1, o, out);
}

int aasm_solve_batch(const aasm_batch_in *in, const aasm_opts *opts, aasm_batch_out *out) {
    if (!in || !out) return AASM_E_INVAL;
    aasm_opts o;
    std::memset(&o, 0, sizeof(o));
    if (opts) o = *opts;
    int rc = validate_batch(in);
    if (rc != AASM_OK) return rc;
    return solve_range(in, 0, in->n_contigs, o, out);
}

// Contig-sharded solve over n_devices GPUs of one node (devices opts.device .. +n-1):
// static contiguous partition balanced on a per-contig cost estimate (aasm_shard.cpp), one host thread and
// one stream per device, results concatenated in contig order.  No collective anywhere:
// contigs are independent (reference: one TBB task per contig, src/alignasm.cpp:351-359).
int aasm_solve_batch_multi(const aasm_batch_in *in, const aasm_opts *opts, int n_devices, aasm_batch_out *out) {
    if (!in || !out || n_devices < 1) return AASM_E_INVAL;
    aasm_opts o;
    std::memset(&o, 0, sizeof(o));
    if (opts) o = *opts;
    int rc = validate_batch(in);
    if (rc != AASM_OK) return rc;
    const int64_t C = in->n_contigs;
    if (n_devices > C) n_devices = (int)C;
    if (n_devices == 1) return solve_range(in, 0, C, o, out);
    // contiguous blocks balanced on the density-aware per-contig cost (aasm_shard.cpp)
    std::vector<double> cost((size_t)C);
    contig_costs(in, cost.data());
    std::vector<int64_t> cut(n_devices + 1, 0);
    partition_by_cost(cost.data(), C, n_devices, cut.data());
    // test hook (opts.reserved[2] bit 1, or AASM_TEST_WRAP_DEVICES=1 for the CLI): device ordinals wrap around the devices
    // that exist, so that the one-thread-per-shard path runs on a box with fewer GPUs than shards (the shards of one
    // device take turns on its context)
    const char *wrap_env = getenv("AASM_TEST_WRAP_DEVICES");
    const bool wrap = (o.reserved[2] & 2) != 0 || (wrap_env && wrap_env[0] == '1');
    const int ndev = std::max(1, aasm_device_count());
    std::vector<aasm_batch_out> parts(n_devices);
    std::vector<int> rcs(n_devices, AASM_OK);
    std::vector<std::string> errs(n_devices);
    {
        std::vector<std::thread> th;
        for (int d = 0; d < n_devices; d++)
            th.emplace_back([&, d] {
                aasm_opts od = o;
                od.device = wrap ? (o.device + d) % ndev : o.device + d;
                std::memset(&parts[d], 0, sizeof(parts[d]));
                static std::mutex turn[16];                              // (a context holds ONE result: a wrapped shard solves and fetches before the next one starts)
                std::unique_lock<std::mutex> lk(turn[od.device & 15], std::defer_lock);
                if (wrap) lk.lock();
                rcs[d] = solve_range(in, cut[d], cut[d + 1], od, &parts[d]);
                if (rcs[d] != AASM_OK) errs[d] = aasm_last_error();
            });
        for (auto &t : th) t.join();
    }
    for (int d = 0; d < n_devices; d++)
        if (rcs[d] != AASM_OK) {
            set_last_error("device " + std::to_string(o.device + d) + ": " + errs[d]);
            for (auto &p : parts) aasm_free_out(&p);
            return rcs[d];
        }
    concat_parts(parts, cut, C, out);
    return AASM_OK;
}

}  // extern "C"
