// aasm_dev.h -- execution context + wave64 primitives + the PafDistance arithmetic.
//
// Every kernel body in aasm_kernels.h is a function `kb_*(const KCtx&, const WS&)`.
//  * Product build (hipcc, gfx950): `__global__` wrappers in aasm_gpu.hip call the body
//    with the real thread/block ids; wave primitives are the CDNA4 64-lane ones
//    (__ballot -> 64-bit mask, __shfl over 64 lanes).
//  * tests/host_emul (g++, AASM_HOST_EMUL): the SAME bodies run with nthreads = 1 so the
//    arithmetic/indexing logic can be diffed against the oracle on a box without a GPU.
//    That build is test infrastructure only; the product library never contains it.
#pragma once
#include <stdint.h>

#if defined(AASM_HOST_EMUL)
#define AASM_DEV static inline
#define AASM_MEM inline
#define AASM_WAVE 1
#define AASM_UNROLL
#else
#include <hip/hip_runtime.h>
#define AASM_DEV static __device__ __forceinline__
#define AASM_MEM __device__ __forceinline__
#define AASM_WAVE 64
#define AASM_UNROLL _Pragma("unroll")
#endif

namespace aasm {

struct KCtx {
    int tid;          // thread in block
    int nthreads;     // block size
    int64_t bid;      // block id
    int64_t nblocks;
    int lane;         // tid % AASM_WAVE
    char *lds;        // per-block LDS scratch (AASM_LDS_BYTES), 16-byte aligned
};
#define AASM_LDS_BYTES 6144   // 6 KB per single-wave block -> 26 blocks per CU by LDS

// ------------------------------------------------------------------------------------
// wave primitives (wave64 on gfx950; trivial with one lane)
// ------------------------------------------------------------------------------------
#if defined(AASM_HOST_EMUL)
AASM_DEV uint64_t wave_ballot(bool p) { return p ? 1ull : 0ull; }
template <class T> AASM_DEV T wave_bcast(T x, int) { return x; }
template <class T> AASM_DEV T wave_shfl_up(T x, int, T fill) { (void)x; return fill; }
template <class T> AASM_DEV T wave_shfl_xor(T x, int) { return x; }
template <class T> AASM_DEV T wave_shfl_idx(T x, int) { return x; }
AASM_DEV void wave_lds_sync() {}
AASM_DEV void block_barrier() {}
AASM_DEV int32_t ld_shared_i32(const int32_t *p) { return *p; }
AASM_DEV void wave_sleep() {}
AASM_DEV int64_t wave_realtime() { return 0; }
template <class T> AASM_DEV void keep_load(T &) {}
AASM_DEV void store_drain() {}
AASM_DEV void wave_fence() {}
template <class T> AASM_DEV T atomic_add(T *p, T v) { T o = *p; *p = o + v; return o; }
AASM_DEV int32_t atomic_min_i32(int32_t *p, int32_t v) { int32_t o = *p; if (v < o) *p = v; return o; }
AASM_DEV void atomic_max_i64(int64_t *p, int64_t v) { if (v > *p) *p = v; }
AASM_DEV int popc64(uint64_t m) { return __builtin_popcountll(m); }
AASM_DEV int ffs64(uint64_t m) { return __builtin_ffsll((long long)m); }
#else
AASM_DEV uint64_t wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
AASM_DEV int wave_bcast(int x, int src) { return __shfl(x, src, 64); }
AASM_DEV int64_t wave_bcast(int64_t x, int src) {
    int lo = __shfl((int)(x & 0xffffffffll), src, 64), hi = __shfl((int)(x >> 32), src, 64);
    return ((int64_t)hi << 32) | (uint32_t)lo;
}
AASM_DEV int wave_shfl_up(int x, int d, int fill) {
    int y = __shfl_up(x, d, 64);
    return ((int)(threadIdx.x & 63) >= d) ? y : fill;
}
AASM_DEV int64_t wave_shfl_up(int64_t x, int d, int64_t fill) {
    int lo = __shfl_up((int)(x & 0xffffffffll), d, 64), hi = __shfl_up((int)(x >> 32), d, 64);
    int64_t y = ((int64_t)hi << 32) | (uint32_t)lo;
    return ((int)(threadIdx.x & 63) >= d) ? y : fill;
}
AASM_DEV int64_t wave_shfl_idx(int64_t x, int src) {                // src may differ from lane to lane
    const int lo = __shfl((int)(uint32_t)(uint64_t)x, src, 64), hi = __shfl((int)((uint64_t)x >> 32), src, 64);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
AASM_DEV int wave_shfl_idx(int x, int src) { return __shfl(x, src, 64); }
AASM_DEV int wave_shfl_xor(int x, int m) { return __shfl_xor(x, m, 64); }
AASM_DEV int64_t wave_shfl_xor(int64_t x, int m) {
    int lo = __shfl_xor((int)(x & 0xffffffffll), m, 64), hi = __shfl_xor((int)(x >> 32), m, 64);
    return ((int64_t)hi << 32) | (uint32_t)lo;
}
// LDS hand-off between the lanes of ONE wave (every kernel that uses it runs one wave per
// workgroup): the DS instructions of a wave execute in issue order, so only the compiler has to
// keep them in order.  __syncthreads() would also drain every outstanding global store
// (s_waitcnt vmcnt(0)), a full memory round trip per call.
AASM_DEV void wave_lds_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
// barrier + LDS/global visibility inside a multi-wave workgroup (K1 sort)
AASM_DEV void block_barrier() { __syncthreads(); }
// a word another wave of the SAME workgroup may have just stored (global memory): workgroup-scope load
AASM_DEV int32_t ld_shared_i32(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
AASM_DEV void wave_sleep() { __builtin_amdgcn_s_sleep(2); }
// a loaded value is wanted HERE, in a register: keeps the compiler from sinking the load into a later branch (a second round trip)
template <class T> AASM_DEV void keep_load(T &x) { asm volatile("" : "+v"(x)); }
AASM_DEV int64_t wave_realtime() { return (int64_t)__builtin_amdgcn_s_memrealtime(); }   // constant 100 MHz counter
// every store of this wave has reached the cache its workgroup shares (before it tells another wave about them)
AASM_DEV void store_drain() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
// order this wave's global-memory writes before its later reads (same CU, same L1)
AASM_DEV void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
template <class T> AASM_DEV T atomic_add(T *p, T v) { return atomicAdd(p, v); }
AASM_DEV unsigned long long atomic_add(int64_t *p, int64_t v) {
    return atomicAdd((unsigned long long *)p, (unsigned long long)v);
}
AASM_DEV int32_t atomic_min_i32(int32_t *p, int32_t v) { return atomicMin(p, v); }
AASM_DEV void atomic_max_i64(int64_t *p, int64_t v) { atomicMax((long long *)p, (long long)v); }   // (non-negative values)
AASM_DEV int popc64(uint64_t m) { return __popcll(m); }
AASM_DEV int ffs64(uint64_t m) { return __ffsll((long long)m); }
#endif

// Diagnostic build only (-DAASM_KPROF, tools/): cycle stamps per kernel section.  The
// product build compiles these to nothing.
#if defined(AASM_KPROF) && !defined(AASM_HOST_EMUL)
#define KPROF_DECL int64_t kp_t0 = 0, kp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define KPROF_START() do { kp_t0 = (int64_t)__builtin_amdgcn_s_memtime(); } while (0)
// light stamp: no drain of outstanding memory operations (a drain would itself expose every prefetch)
#define KPROF_STAMP(i) do { const int64_t kp_t1 = (int64_t)__builtin_amdgcn_s_memtime(); kp_acc[i] += kp_t1 - kp_t0; kp_t0 = kp_t1; } while (0)
#define KPROF_FLUSH(ptr, c, lane) do { if ((ptr) && (lane) == 0) for (int kp_i = 0; kp_i < 8; kp_i++) (ptr)[(c) * 8 + kp_i] = kp_acc[kp_i]; } while (0)
#else
#define KPROF_DECL
#define KPROF_START() do {} while (0)
#define KPROF_STAMP(i) do {} while (0)
#define KPROF_FLUSH(ptr, c, lane) do {} while (0)
#endif

// Wave-uniform values made explicitly scalar: v_readfirstlane moves them to SGPRs, so the
// arithmetic and the branches that follow run on the scalar unit instead of occupying all
// 64 VALU lanes with identical work.  Only valid where every lane holds the same value.
#if defined(AASM_HOST_EMUL)
AASM_DEV int32_t uni(int32_t x) { return x; }
#else
AASM_DEV int32_t uni(int32_t x) { return __builtin_amdgcn_readfirstlane(x); }
#endif
AASM_DEV int64_t uni(int64_t x) {
    return (int64_t)(((uint64_t)(uint32_t)uni((int32_t)((uint64_t)x >> 32)) << 32) | (uint32_t)uni((int32_t)(uint32_t)(uint64_t)x));
}
AASM_DEV bool uni(bool x) { return uni((int32_t)x) != 0; }

#define AASM_WAVE_MAX 64
// LaneArr<T>: an array of <= 64 elements held one per lane in registers (element j lives in
// lane j).  The 1-lane host emulation keeps a real array, so the same source runs in both.
//   FOR_LANE(j, n, lane) { ... a.at(j) ... }   every element j < n, all at once on the GPU
//   LA_GET(a, j, .field)                         read element j (wave-uniform j) into a scalar
//   LA_SET(a, j, lane, value)                    write element j (wave-uniform j)
#if defined(AASM_HOST_EMUL)
template <class T> struct LaneArr { T a[AASM_WAVE_MAX]; AASM_MEM T &at(int j) { return a[j]; } };
#define FOR_LANE(j, n, lane) for (int j = 0; j < (n); j++)
#define FOR_LANE_EQ(j, jj, lane) for (int j = (jj), _fe_once = 1; _fe_once; _fe_once = 0)
#define LA_GET(arr, j, field) ((arr).a[j] field)
#define LA_SET(arr, j, lane, value) ((arr).a[j] = (value))
#else
template <class T> struct LaneArr { T r; AASM_MEM T &at(int) { return r; } };
#define FOR_LANE(j, n, lane) for (int j = (lane), _fl_once = 1; _fl_once && j < (n); _fl_once = 0)
#define FOR_LANE_EQ(j, jj, lane) for (int j = (lane), _fe_once = 1; _fe_once && j == (jj); _fe_once = 0)   // element jj alone (lane jj)
#define LA_GET(arr, j, field) __builtin_amdgcn_readlane((arr).r field, (j))
#define LA_SET(arr, j, lane, value) do { if ((lane) == (j)) (arr).r = (value); } while (0)
#endif

// c[j] <- min of c[i] over j < i < n (INT32_MAX where there is none), n <= 64
// Device: DPP row shifts (a lane reads the lane d places above it inside its row of 16; out-of-row sources
// keep the identity) give the inclusive suffix-min of every row in four vector instructions; rows are joined
// through three broadcast reads when n reaches past the first row; one wavefront shift makes it exclusive.
#if !defined(AASM_HOST_EMUL)
template <int CTRL> AASM_DEV int32_t dpp_or(int32_t identity, int32_t x) { return __builtin_amdgcn_update_dpp(identity, x, CTRL, 0xf, 0xf, false); }
#endif
AASM_DEV void lane_excl_suffix_min(LaneArr<int32_t> &c, int n, int lane) {
#if defined(AASM_HOST_EMUL)
    int32_t run = INT32_MAX;
    for (int j = n - 1; j >= 0; j--) { const int32_t t = c.a[j]; c.a[j] = run; run = t < run ? t : run; }
    (void)lane;
#else
    int32_t x = lane < n ? c.r : INT32_MAX, y;
    y = dpp_or<0x101>(INT32_MAX, x); x = y < x ? y : x;               // row_shl:1
    y = dpp_or<0x102>(INT32_MAX, x); x = y < x ? y : x;               // row_shl:2
    y = dpp_or<0x104>(INT32_MAX, x); x = y < x ? y : x;               // row_shl:4
    y = dpp_or<0x108>(INT32_MAX, x); x = y < x ? y : x;               // row_shl:8
    if (n > 16) {                                                     // wave-uniform: spines longer than 15 nodes (heaps of > 32 k nodes)
        const int32_t s3 = __builtin_amdgcn_readlane(x, 48);
        int32_t s2 = __builtin_amdgcn_readlane(x, 32), s1 = __builtin_amdgcn_readlane(x, 16);
        s2 = s3 < s2 ? s3 : s2; s1 = s2 < s1 ? s2 : s1;
        const int32_t above = lane < 16 ? s1 : lane < 32 ? s2 : lane < 48 ? s3 : INT32_MAX;
        x = above < x ? above : x;
    }
    c.r = dpp_or<0x130>(INT32_MAX, x);                                // wave_shl:1: the value of lane + 1
#endif
}

// mask of the indices j in [0, n), n <= 64, for which pred(j) holds: lane j evaluates pred(j)
template <class P> AASM_DEV uint64_t wave_index_mask(int n, int lane, P pred) {
#if defined(AASM_HOST_EMUL)
    uint64_t m = 0;
    for (int j = 0; j < n; j++) if (pred(j)) m |= 1ull << j;
    (void)lane;
    return m;
#else
    return wave_ballot(lane < n && pred(lane));
#endif
}

AASM_DEV bool wave_any(bool p) { return wave_ballot(p) != 0; }
AASM_DEV int32_t hi32(uint64_t x) { return (int32_t)(x >> 32); }
AASM_DEV int32_t lo32(uint64_t x) { return (int32_t)(uint32_t)x; }
AASM_DEV uint64_t mk64(int32_t lo, int32_t hi) { return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo; }
AASM_DEV uint64_t lanemask_lt(int lane) { return lane >= 64 ? ~0ull : ((1ull << lane) - 1ull); }

// inclusive wave scans (Hillis-Steele over 64 lanes; identity with one lane)
AASM_DEV int wave_incl_add(int x) {
    for (int d = 1; d < AASM_WAVE; d <<= 1) x += wave_shfl_up(x, d, 0);
    return x;
}
AASM_DEV int64_t wave_incl_max(int64_t x, int64_t neutral) {
    for (int d = 1; d < AASM_WAVE; d <<= 1) { int64_t y = wave_shfl_up(x, d, neutral); x = y > x ? y : x; }
    return x;
}
AASM_DEV int64_t wave_incl_min(int64_t x, int64_t neutral) {
    for (int d = 1; d < AASM_WAVE; d <<= 1) { int64_t y = wave_shfl_up(x, d, neutral); x = y < x ? y : x; }
    return x;
}

AASM_DEV int64_t wave_incl_add(int64_t x) {
    for (int d = 1; d < AASM_WAVE; d <<= 1) x += wave_shfl_up(x, d, (int64_t)0);
    return x;
}
AASM_DEV int64_t wave_sum(int64_t x) {
    for (int d = 1; d < AASM_WAVE; d <<= 1) x += wave_shfl_up(x, d, (int64_t)0);
    return wave_bcast(x, AASM_WAVE - 1);
}

// ------------------------------------------------------------------------------------
// PafDistance (reference: src/paf_data.hpp:121-189) in 32 bytes.
// anom / qul_nonzero / qul_total are path-length-bounded counters -> int32; the two
// scores stay int64.  calc_sum_chk only feeds asserts in the reference and is dropped.
// ------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) Dist {
    int64_t qry, ref;
    int32_t anom, qnz, qtot, pad;
};
enum { CALC_SUM_MODE = 0, QRY_SCORE_MODE = 1 };

AASM_DEV Dist dist_zero() { Dist d; d.qry = 0; d.ref = 0; d.anom = 0; d.qnz = 0; d.qtot = 0; d.pad = 0; return d; }
AASM_DEV Dist dist_max() { Dist d; d.qry = -1; d.ref = -1; d.anom = -1; d.qnz = -1; d.qtot = 0; d.pad = 0; return d; }
// `x == PafDistance::max()` through operator== (paf_data.hpp:163-168): max has qtot 0 -> 1
AASM_DEV bool dist_is_max(const Dist &a) {
    int64_t tot = a.qtot ? a.qtot : 1;
    return a.qry == -1 && a.ref == -1 && a.anom == -1 && (int64_t)a.qnz == -tot;
}
AASM_DEV bool dist_eq(const Dist &a, const Dist &b) {               // paf_data.hpp:163-168
    int64_t tot = a.qtot ? a.qtot : 1, rtot = b.qtot ? b.qtot : 1;
    return a.qry == b.qry && a.ref == b.ref && a.anom == b.anom && (int64_t)a.qnz * rtot == (int64_t)b.qnz * tot;
}
template <int MODE> AASM_DEV bool dist_lt(const Dist &a, const Dist &b) {   // paf_data.hpp:142-159
    if (dist_is_max(a)) return false;
    if (dist_is_max(b)) return true;
    if (MODE == CALC_SUM_MODE) {
        int64_t sa = a.qry + a.ref, sb = b.qry + b.ref;
        if (sa != sb) return sa < sb;
    } else {
        if (a.qry != b.qry) return a.qry < b.qry;
        if (a.ref != b.ref) return a.ref < b.ref;
    }
    if (a.anom != b.anom) return a.anom < b.anom;
    const int32_t tot = a.qtot ? a.qtot : 1, rtot = b.qtot ? b.qtot : 1;
    return (int64_t)a.qnz * (int64_t)rtot > (int64_t)b.qnz * (int64_t)tot;      // 32 x 32 -> 64
}
AASM_DEV Dist dist_add(const Dist &a, const Dist &b) {               // paf_data.hpp:178-183
    Dist r; r.qry = a.qry + b.qry; r.ref = a.ref + b.ref; r.anom = a.anom + b.anom;
    r.qnz = a.qnz + b.qnz; r.qtot = a.qtot + b.qtot; r.pad = 0; return r;
}
AASM_DEV Dist dist_sub(const Dist &a, const Dist &b) {               // paf_data.hpp:184-188
    Dist r; r.qry = a.qry - b.qry; r.ref = a.ref - b.ref; r.anom = a.anom - b.anom;
    r.qnz = a.qnz - b.qnz; r.qtot = a.qtot - b.qtot; r.pad = 0; return r;
}

// constants, src/paf_data.hpp:21-29
#define AASM_REF_NEGATIVE_PENALTY 2
#define AASM_SV_BASELINE 1000000
#define AASM_SV_TRANS_PENALTY 2000
#define AASM_SV_INV_PENALTY 500
#define AASM_SV_FRONT_END_COEFFICIENT 2

// edge flag byte: bits 0-1 anom (0..2), bit 2 qul_nonzero, bit 3 qul_total
AASM_DEV uint8_t edge_flags(int anom, int qnz, int qtot) { return (uint8_t)(anom | (qnz << 2) | (qtot << 3)); }
AASM_DEV Dist edge_dist(int64_t wq, int32_t wr, uint8_t fl) {
    Dist d; d.qry = wq; d.ref = wr; d.anom = fl & 3; d.qnz = (fl >> 2) & 1; d.qtot = (fl >> 3) & 1; d.pad = 0; return d;
}

// persistent leftist-heap node (reference: src/leftist_heap.hpp:18-27), 48 bytes
struct __attribute__((aligned(16))) HNode {
    int64_t kq, kr;                 // key.qry_score, key.ref_score
    int32_t ka, kn, kt, rank;       // key.anom, key.qul_nonzero, key.qul_total;
                                    // rank = node_rank | rank(left) << 8 | rank(right) << 16 (child ranks cached)
    int32_t left, right, u, v;      // arena indices (-1 = nullptr), value = edge (u, v)
};
// a node as three 16-byte quads kept in registers: q0 = {kq, kr}, q1 = {ka, kn, kt, rank},
// q2 = {left, right, u, v}
struct __attribute__((aligned(16))) I4 { int32_t x, y, z, w; };
struct NodeQ { I4 q0, q1, q2; };
AASM_DEV NodeQ nodeq_load(const HNode *p) { const I4 *q = (const I4 *)p; NodeQ n; n.q0 = q[0]; n.q1 = q[1]; n.q2 = q[2]; return n; }
AASM_DEV void nodeq_store(HNode *p, const NodeQ &n) { I4 *q = (I4 *)p; q[0] = n.q0; q[1] = n.q1; q[2] = n.q2; }
AASM_DEV Dist nodeq_key(const NodeQ &n) {
    Dist d;
    d.qry = (int64_t)(((uint64_t)(uint32_t)n.q0.y << 32) | (uint32_t)n.q0.x);
    d.ref = (int64_t)(((uint64_t)(uint32_t)n.q0.w << 32) | (uint32_t)n.q0.z);
    d.anom = n.q1.x; d.qnz = n.q1.y; d.qtot = n.q1.z; d.pad = 0;
    return d;
}
AASM_DEV I4 uni(const I4 &q) { I4 r; r.x = uni(q.x); r.y = uni(q.y); r.z = uni(q.z); r.w = uni(q.w); return r; }
AASM_DEV NodeQ uni(const NodeQ &n) {
    NodeQ r;
    r.q0.x = uni(n.q0.x); r.q0.y = uni(n.q0.y); r.q0.z = uni(n.q0.z); r.q0.w = uni(n.q0.w);
    r.q1.x = uni(n.q1.x); r.q1.y = uni(n.q1.y); r.q1.z = uni(n.q1.z); r.q1.w = uni(n.q1.w);
    r.q2.x = uni(n.q2.x); r.q2.y = uni(n.q2.y); r.q2.z = uni(n.q2.z); r.q2.w = uni(n.q2.w);
    return r;
}
AASM_DEV Dist uni(const Dist &d) {
    Dist r; r.qry = uni(d.qry); r.ref = uni(d.ref); r.anom = uni(d.anom); r.qnz = uni(d.qnz); r.qtot = uni(d.qtot); r.pad = 0; return r;
}
// element j (wave-uniform j) of a lane array of distances
#if defined(AASM_HOST_EMUL)
AASM_DEV Dist la_get_dist(const LaneArr<Dist> &a, int j) { return a.a[j]; }
#else
AASM_DEV Dist la_get_dist(const LaneArr<Dist> &a, int j) {
    Dist d;
    const uint32_t q0 = (uint32_t)__builtin_amdgcn_readlane((int32_t)(uint32_t)(uint64_t)a.r.qry, j), q1 = (uint32_t)__builtin_amdgcn_readlane((int32_t)((uint64_t)a.r.qry >> 32), j);
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int32_t)(uint32_t)(uint64_t)a.r.ref, j), r1 = (uint32_t)__builtin_amdgcn_readlane((int32_t)((uint64_t)a.r.ref >> 32), j);
    d.qry = (int64_t)(((uint64_t)q1 << 32) | q0); d.ref = (int64_t)(((uint64_t)r1 << 32) | r0);
    d.anom = __builtin_amdgcn_readlane(a.r.anom, j); d.qnz = __builtin_amdgcn_readlane(a.r.qnz, j); d.qtot = __builtin_amdgcn_readlane(a.r.qtot, j);
    d.pad = __builtin_amdgcn_readlane(a.r.pad, j);
    return d;
}
#endif
AASM_DEV Dist hnode_key(const HNode &n) {
    Dist d; d.qry = n.kq; d.ref = n.kr; d.anom = n.ka; d.qnz = n.kn; d.qtot = n.kt; d.pad = 0; return d;
}

// one output element, same layout as aasm_out_elem (include/alignasm_amd.h)
struct OutElem {
    int64_t qs, qe, rs, re;
    int32_t ctg_index, is_alt;
};

}  // namespace aasm
