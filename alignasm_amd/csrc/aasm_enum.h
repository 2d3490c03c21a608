// aasm_enum.h -- K8, the k-walk enumeration (k_shortest_walks.hpp:217-249), device form.
//
// The reference pops K times from a binary heap of (Distance, heap node, insertion index) tuples and
// pushes up to three successors per pop.  The tuple order is total (kb_enum_heap's comment), so ANY
// exact min-queue pops the same sequence; what is fixed is the order of the pops and the insertion
// indices they hand out.  A d-ary heap makes every pop a chain of dependent levels (16 k cycles per
// pop at K = 10 000: 0.4 % of the HBM roof).  Measured on the batches of the bench: only 8-11 % of
// the pushes land among the 64 smallest entries, 40-60 % land behind rank 4 096, and two thirds are
// never popped.  So the queue here is a two-tier one whose slow tier is never walked entry by entry:
//
//  F   the |F| <= 64 smallest entries of the whole queue, SORTED, one per lane, in registers.
//      pop = the front lane (a head index moves; nothing shifts).  A push smaller than max(F) is
//      placed with one compare + ballot + one DPP wave shift of the 11 registers of an entry.
//  I   insertion buffer (LDS, 64 entries, unsorted): every other push, three lanes at a time.
//  R   sorted runs in global memory, one per level of a binary LSM tree (run of level l: <= 64 << l
//      entries).  A full I is sorted by a 21-stage bitonic network across the lanes and merged down
//      the levels: two sorted runs are merged 64 entries per step (the upper half of a 128-entry
//      bitonic merge is carried, the next 2 KB block comes from the run whose last loaded entry is
//      smaller), all loads and stores whole coalesced blocks.  An entry at position p of a sorted run
//      has p smaller entries in front of it, so a merged run is CUT at K - found entries: what lies
//      behind can never be popped.
//  refill (F ran empty): the 64 smallest of the run heads (one block per level, merged keeping the
//      lower half, each entry tagged with its level so the heads can be advanced).
//
// The memory latency of a pop (heap node of the popped entry -> root of the cross heap -> keys of the
// <= 3 successors: three dependent round trips) is paid once per REFILL for 64 entries at a time:
// every lane of F fetches the successor keys of its own entry and parks them in LDS (96 B per entry);
// an entry pushed straight into F is fetched the first time an unfetched entry reaches the front,
// together with every other unfetched one.  A pop then touches registers and LDS only.
#pragma once

namespace aasm {
// geometry of the run storage of one contig (host: pq stride; device: slot addresses)
#if defined(AASM_HOST_EMUL)
#define AASM_HD inline
#else
#define AASM_HD __host__ __device__ inline
#endif
AASM_HD int64_t enum_k64(int64_t K) { return (K + 63) / 64 * 64; }
AASM_HD int32_t enum_lmax(int64_t K) { int32_t l = 0; while (((int64_t)64 << l) < enum_k64(K)) l++; return l; }
AASM_HD int64_t enum_stride(int64_t K) { return 128 * (((int64_t)1 << enum_lmax(K)) - 1) + 3 * enum_k64(K); }
}  // namespace aasm

#if !defined(AASM_HOST_EMUL)

namespace aasm {

#define EQ_ILEN 64
#define EQ_IFLUSH 58                     // flush I when it holds more than this (a pop adds <= 3 entries)
#define EQ_MAXLEV 28
struct EnumLds {
    I4 succ[64][3][2];                   // per F slot: successor j = {key delta: sum (2 words), qry (2 words)}, {anom, qnz, qtot, heap node | -1}
    I4 ibuf[EQ_ILEN][2];                 // I: {sum (2), anom, qnz}, {qtot, node, cur, prev}
    int32_t run_slot[EQ_MAXLEV], run_head[EQ_MAXLEV], run_end[EQ_MAXLEV];
};
#define AASM_ENUM2_LDS_BYTES (64 * 96 + EQ_ILEN * 32 + 3 * EQ_MAXLEV * 4)
static_assert(sizeof(EnumLds) <= AASM_ENUM2_LDS_BYTES, "LDS budget");

struct QE { int64_t sum; int32_t anom, qnz, qtot, node, cur, prev, tag; };   // tag: refill only (source level)
AASM_DEV QE qe_inf() { QE e; e.sum = INT64_MAX; e.anom = e.qnz = e.qtot = e.node = e.cur = e.prev = e.tag = 0; return e; }
AASM_DEV bool qe_less_v(const QE &a, const QE &b) {                  // pq_full_less on entries, branch-free (one per lane)
    const int32_t ta = a.qtot ? a.qtot : 1, tb = b.qtot ? b.qtot : 1;
    const int64_t l = (int64_t)a.qnz * (int64_t)tb, r = (int64_t)b.qnz * (int64_t)ta;
    const int64_t ia = (int64_t)(((uint64_t)(uint32_t)a.node << 32) | (uint32_t)a.cur), ib = (int64_t)(((uint64_t)(uint32_t)b.node << 32) | (uint32_t)b.cur);
    return (a.sum < b.sum) | ((a.sum == b.sum) & ((a.anom < b.anom) | ((a.anom == b.anom) & ((l > r) | ((l == r) & (ia < ib))))));
}
AASM_DEV bool qe_less_s(const QE &a, const QE &b) {                  // the same on wave-uniform values
    if (a.sum != b.sum) return a.sum < b.sum;
    if (a.anom != b.anom) return a.anom < b.anom;
    const int32_t ta = a.qtot ? a.qtot : 1, tb = b.qtot ? b.qtot : 1;
    const int64_t l = (int64_t)a.qnz * (int64_t)tb, r = (int64_t)b.qnz * (int64_t)ta;
    if (l != r) return l > r;
    if (a.node != b.node) return a.node < b.node;
    return a.cur < b.cur;
}
AASM_DEV int32_t hi32(int64_t x) { return (int32_t)((uint64_t)x >> 32); }
AASM_DEV int32_t lo32(int64_t x) { return (int32_t)(uint32_t)(uint64_t)x; }
AASM_DEV int64_t mk64(int32_t lo, int32_t hi) { return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo); }

// value of lane (lane ^ M): DPP inside a row of 16, ds_swizzle inside 32 lanes, bpermute across the halves
template <int M> AASM_DEV int32_t lane_xor(int32_t x, int lane) {
    if (M == 1) return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xf, 0xf, true);         // quad_perm [1,0,3,2]
    else if (M == 2) return __builtin_amdgcn_mov_dpp(x, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
    else if (M == 3) return __builtin_amdgcn_mov_dpp(x, 0x1B, 0xf, 0xf, true);    // quad_perm [3,2,1,0]
    else if (M == 7) return __builtin_amdgcn_mov_dpp(x, 0x141, 0xf, 0xf, true);   // row_half_mirror
    else if (M == 15) return __builtin_amdgcn_mov_dpp(x, 0x140, 0xf, 0xf, true);  // row_mirror
    else if (M == 4 || M == 8 || M == 16 || M == 31) return __builtin_amdgcn_ds_swizzle(x, (M << 10) | 0x1F);
    else return __builtin_amdgcn_ds_bpermute((lane ^ M) << 2, x);
}
template <int M> AASM_DEV QE qe_xor(const QE &e, int lane) {
    QE p;
    p.sum = mk64(lane_xor<M>(lo32(e.sum), lane), lane_xor<M>(hi32(e.sum), lane));
    p.anom = lane_xor<M>(e.anom, lane); p.qnz = lane_xor<M>(e.qnz, lane); p.qtot = lane_xor<M>(e.qtot, lane);
    p.node = lane_xor<M>(e.node, lane); p.cur = lane_xor<M>(e.cur, lane); p.prev = lane_xor<M>(e.prev, lane); p.tag = lane_xor<M>(e.tag, lane);
    return p;
}
AASM_DEV QE qe_sel(bool take, const QE &p, const QE &e) {
    QE r;
    r.sum = take ? p.sum : e.sum; r.anom = take ? p.anom : e.anom; r.qnz = take ? p.qnz : e.qnz; r.qtot = take ? p.qtot : e.qtot;
    r.node = take ? p.node : e.node; r.cur = take ? p.cur : e.cur; r.prev = take ? p.prev : e.prev; r.tag = take ? p.tag : e.tag;
    return r;
}
// compare-exchange with lane ^ M: the lower lane of a pair keeps the smaller entry
template <int M> AASM_DEV void qe_cx(QE &e, int lane) {
    constexpr int HB = M >= 32 ? 32 : M >= 16 ? 16 : M >= 8 ? 8 : M >= 4 ? 4 : M >= 2 ? 2 : 1;
    const QE p = qe_xor<M>(e, lane);
    const bool lower = (lane & HB) == 0;
    const bool pl = qe_less_v(p, e), el = qe_less_v(e, p);
    e = qe_sel(lower ? pl : el, p, e);
}
AASM_DEV void qe_sort64(QE &e, int lane) {                          // bitonic sort, flip formulation: 21 stages
    qe_cx<1>(e, lane);
    qe_cx<3>(e, lane); qe_cx<1>(e, lane);
    qe_cx<7>(e, lane); qe_cx<2>(e, lane); qe_cx<1>(e, lane);
    qe_cx<15>(e, lane); qe_cx<4>(e, lane); qe_cx<2>(e, lane); qe_cx<1>(e, lane);
    qe_cx<31>(e, lane); qe_cx<8>(e, lane); qe_cx<4>(e, lane); qe_cx<2>(e, lane); qe_cx<1>(e, lane);
    qe_cx<63>(e, lane); qe_cx<16>(e, lane); qe_cx<8>(e, lane); qe_cx<4>(e, lane); qe_cx<2>(e, lane); qe_cx<1>(e, lane);
}
AASM_DEV void qe_clean64(QE &e, int lane) {                         // a bitonic sequence of 64 -> sorted
    qe_cx<32>(e, lane); qe_cx<16>(e, lane); qe_cx<8>(e, lane); qe_cx<4>(e, lane); qe_cx<2>(e, lane); qe_cx<1>(e, lane);
}
// x, y sorted ascending (one entry per lane) -> x = the 64 smallest of both, y = the 64 largest, both sorted
AASM_DEV void qe_merge(QE &x, QE &y, int lane, bool want_hi) {
    const QE yr = qe_xor<63>(y, lane);
    const bool sw = qe_less_v(yr, x);
    const QE lo = qe_sel(sw, yr, x), hi = qe_sel(sw, x, yr);
    x = lo; y = hi;
    qe_clean64(x, lane);
    if (want_hi) qe_clean64(y, lane);
}
AASM_DEV QE qe_uni(const QE &e, int j) {                            // the entry lane j holds (wave-uniform j)
    QE r;
    r.sum = mk64(__builtin_amdgcn_readlane(lo32(e.sum), j), __builtin_amdgcn_readlane(hi32(e.sum), j));
    r.anom = __builtin_amdgcn_readlane(e.anom, j); r.qnz = __builtin_amdgcn_readlane(e.qnz, j); r.qtot = __builtin_amdgcn_readlane(e.qtot, j);
    r.node = __builtin_amdgcn_readlane(e.node, j); r.cur = __builtin_amdgcn_readlane(e.cur, j); r.prev = __builtin_amdgcn_readlane(e.prev, j); r.tag = 0;
    return r;
}
// a 64-entry block of a run: entry (at + lane), or +inf behind the end
AASM_DEV QE qe_load(const PqK *g, int32_t at, int32_t end, int lane) {
    QE e = qe_inf();
    if (at + lane < end) {
        const I4 *p = (const I4 *)(g + at + lane);
        const I4 a = p[0], b = p[1];
        e.sum = mk64(a.x, a.y); e.anom = a.z; e.qnz = a.w; e.qtot = b.x; e.node = b.y; e.cur = b.z; e.prev = b.w;
    }
    return e;
}
AASM_DEV void qe_store(PqK *g, int32_t at, const QE &e) {
    I4 *p = (I4 *)(g + at);
    I4 a, b;
    a.x = lo32(e.sum); a.y = hi32(e.sum); a.z = e.anom; a.w = e.qnz; b.x = e.qtot; b.y = e.node; b.z = e.cur; b.w = e.prev;
    p[0] = a; p[1] = b;
}

struct EnumQ {
    EnumLds *L;
    PqK *g;                              // run storage of this contig
    int32_t lmax, k64;
    int32_t in;                          // entries in I
    int32_t nruns;                       // non-empty levels
};
AASM_DEV int32_t eq_cap(const EnumQ &q, int32_t l) { const int64_t c = (int64_t)64 << l; return l < q.lmax && c < q.k64 ? (int32_t)c : q.k64; }
AASM_DEV PqK *eq_slot(const EnumQ &q, int32_t l, int32_t s) {
    const int32_t lb = l < q.lmax ? l : q.lmax;
    return q.g + 128 * (((int64_t)1 << lb) - 1) + (int64_t)s * eq_cap(q, l);
}

// out[0 .. out_len) <- the out_len smallest of (carry, if has_carry) + A[0 .. a_len) + B[0 .. b_len); the inputs are sorted.
// With has_carry the A run is not read (a_len = 0): the carry is the sorted block in registers.
AASM_DEV void eq_merge(const PqK *A, int32_t a_len, const PqK *B, int32_t b_len, PqK *out, int32_t out_len, QE carry, bool has_carry, int lane) {
    int32_t ia = 0, ib = 0, o = 0;
    QE lastA = qe_inf(), lastB = qe_inf();
    lastB.sum = INT64_MIN;                                           // "nothing loaded yet": the first block comes from B
    if (!has_carry) { carry = qe_load(A, 0, a_len, lane); ia = 64; lastA = qe_uni(carry, 63); } else a_len = 0;
    while (o < out_len) {
        const bool availA = ia < a_len, availB = ib < b_len;
        if (!availA && !availB) break;
        // next block: from the run whose last loaded entry is smaller (all entries not loaded yet are then >= every
        // entry this step emits); an exhausted run leaves the choice to the other one
        const bool takeA = availA && (!availB || qe_less_s(lastA, lastB));
        QE blk;
        if (takeA) { blk = qe_load(A, ia, a_len, lane); ia += 64; lastA = qe_uni(blk, 63); }
        else { blk = qe_load(B, ib, b_len, lane); ib += 64; lastB = qe_uni(blk, 63); }
        qe_merge(carry, blk, lane, true);
        if (o + lane < out_len) qe_store(out, o + lane, carry);
        o += 64;
        carry = blk;
    }
    if (o + lane < out_len) qe_store(out, o + lane, carry);          // (out_len never exceeds the number of real entries)
    wave_fence();                                                    // later block loads of this wave see the run
}

// I -> a sorted block, merged down the levels of the LSM tree (one run per level at rest; level l < lmax has two
// slots of 64 << l entries, the top level three of k64: the output of a merge goes to a slot none of its inputs uses)
AASM_DEV void eq_flush(EnumQ &q, int32_t keep, int lane) {
    QE s = qe_inf();
    if (lane < q.in) {
        const I4 a = q.L->ibuf[lane][0], b = q.L->ibuf[lane][1];
        s.sum = mk64(a.x, a.y); s.anom = a.z; s.qnz = a.w; s.qtot = b.x; s.node = b.y; s.cur = b.z; s.prev = b.w;
    }
    const int32_t n = q.in;
    q.in = 0;
    qe_sort64(s, lane);
    if (keep < 1) keep = 1;
    const int32_t top = q.lmax;
    int32_t l = 0, len = n < keep ? n : keep, cslot = 0;
    bool in_regs = true;                                             // the run being carried down: in s, or in slot cslot of level l
    while (true) {
        const int32_t os = uni(q.L->run_slot[l]), oh = uni(q.L->run_head[l]), oe = uni(q.L->run_end[l]);
        if (oh >= oe) {                                              // level l is free: the carried run stays here
            if (in_regs) { if (lane < len) qe_store(eq_slot(q, l, 0), lane, s); wave_fence(); cslot = 0; }
            break;
        }
        int32_t tot = len + (oe - oh);
        if (tot > keep) tot = keep;
        const int32_t nl = l == top ? top : l + 1;
        int32_t ns;
        if (nl == l) ns = in_regs ? (os + 1) % 3 : 3 - os - cslot;   // inside the top level: the slot neither input uses
        else {
            const int32_t s2 = uni(q.L->run_slot[nl]);
            const bool occ2 = uni(q.L->run_head[nl]) < uni(q.L->run_end[nl]);
            ns = occ2 ? (nl == top ? (s2 + 1) % 3 : 1 - s2) : 0;
        }
        eq_merge(eq_slot(q, l, cslot), in_regs ? 0 : len, eq_slot(q, l, os) + oh, oe - oh, eq_slot(q, nl, ns), tot, s, in_regs, lane);
        if (lane == 0) { q.L->run_head[l] = 0; q.L->run_end[l] = 0; }
        wave_lds_sync();
        q.nruns--;
        in_regs = false; len = tot; cslot = ns;
        if (nl == l) break;                                          // the top level holds one run again
        l = nl;
    }
    if (lane == 0) { q.L->run_slot[l] = cslot; q.L->run_head[l] = 0; q.L->run_end[l] = len; }
    if (len > 0) q.nruns++;
    wave_lds_sync();
}

AASM_DEV void kb_enum_lsm(const KCtx &k, const WS &w) {             // one wave per contig
    const int64_t c = k.bid;
    const int64_t V = w.ctgV[c];
    const int lane = k.lane;
    if (lane == 0) w.kfound[c] = 0;
    if (V == 0 || w.status[c] != 0) return;
    const int64_t vb = w.voff[c];
    const int32_t K = w.K;
    Dist *kd = w.kd + c * (int64_t)K;
    int32_t *klast = w.klast + c * (int64_t)K, *knodes = w.knodes + c * (3 * (int64_t)K + 1), *kprev = w.kprev + c * (3 * (int64_t)K + 1);
    int64_t *kcq = w.kcq + c * (3 * (int64_t)K + 1);
    const HNode *nodes = w.hnodes + w.hoff[c];
    const int32_t *h = w.h_root + vb;
    const int32_t src = (int32_t)(V - 2);
    EnumQ q;
    q.L = (EnumLds *)k.lds; q.g = w.pq + c * w.pq_stride; q.lmax = enum_lmax(K); q.k64 = (int32_t)enum_k64(K); q.in = 0; q.nruns = 0;
    if (lane < EQ_MAXLEV) { q.L->run_slot[lane] = 0; q.L->run_head[lane] = 0; q.L->run_end[lane] = 0; }
    wave_lds_sync();

    const Dist dsrc = w.sp_d[vb + src];
    if (lane == 0) { kd[0] = dsrc; klast[0] = -1; }                  // :217-220
    int32_t found = 1, nn = 0;
    const int32_t hs = uni(h[src]);
    if (hs < 0) { if (lane == 0) { w.kfound[c] = 1; atomic_add(&w.counters[CNT_PATHS], (int64_t)1); } return; }   // :227-228

    // ---- F: sorted entries in lanes [hd, nf); f_slot = LDS successor slot | 0x100 while the successors are not fetched yet
    QE f = qe_inf();
    int64_t f_qry = 0;
    int32_t f_slot = 0;
    int32_t hd = 0, nf = 0;
    uint64_t freemask = ~0ull;
    QE maxF = qe_inf();

    auto f_insert = [&](const QE &x, int64_t xq) {                  // x (wave-uniform) into F; the caller has decided that it belongs there
        const int32_t below = popc64(wave_ballot(lane >= hd && lane < nf && qe_less_v(f, x)));
        int32_t pos = hd + below;
        if (nf == 64 && hd == 0) {                                   // full: the largest entry leaves for I
            if (lane == 63) {
                I4 a, b;
                a.x = lo32(f.sum); a.y = hi32(f.sum); a.z = f.anom; a.w = f.qnz; b.x = f.qtot; b.y = f.node; b.z = f.cur; b.w = f.prev;
                q.L->ibuf[q.in][0] = a; q.L->ibuf[q.in][1] = b;
            }
            freemask |= 1ull << (__builtin_amdgcn_readlane(f_slot, 63) & 63);
            q.in++; nf = 63;
        }
        const int32_t sl = ffs64(freemask) - 1;
        freemask &= ~(1ull << sl);
        bool mv; int32_t at;
        QE t; int64_t tq; int32_t ts;
#define EQ_SHIFT(CTRL) do { \
            t.sum = mk64(__builtin_amdgcn_update_dpp(0, lo32(f.sum), CTRL, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0, hi32(f.sum), CTRL, 0xf, 0xf, false)); \
            t.anom = __builtin_amdgcn_update_dpp(0, f.anom, CTRL, 0xf, 0xf, false); t.qnz = __builtin_amdgcn_update_dpp(0, f.qnz, CTRL, 0xf, 0xf, false); \
            t.qtot = __builtin_amdgcn_update_dpp(0, f.qtot, CTRL, 0xf, 0xf, false); t.node = __builtin_amdgcn_update_dpp(0, f.node, CTRL, 0xf, 0xf, false); \
            t.cur = __builtin_amdgcn_update_dpp(0, f.cur, CTRL, 0xf, 0xf, false); t.prev = __builtin_amdgcn_update_dpp(0, f.prev, CTRL, 0xf, 0xf, false); t.tag = 0; \
            tq = mk64(__builtin_amdgcn_update_dpp(0, lo32(f_qry), CTRL, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0, hi32(f_qry), CTRL, 0xf, 0xf, false)); \
            ts = __builtin_amdgcn_update_dpp(0, f_slot, CTRL, 0xf, 0xf, false); } while (0)
        if (nf < 64) {                                               // lanes [pos, nf) move up by one
            EQ_SHIFT(0x138);                                         // wave_shr:1: the value of lane - 1
            mv = lane > pos && lane <= nf; at = pos; nf++;
        } else {                                                     // room only below the head: lanes [hd, pos) move down by one
            EQ_SHIFT(0x130);                                         // wave_shl:1: the value of lane + 1
            mv = lane >= hd - 1 && lane < pos - 1; at = pos - 1; hd--;
        }
#undef EQ_SHIFT
        f = qe_sel(mv, t, f); f_qry = mv ? tq : f_qry; f_slot = mv ? ts : f_slot;
        if (lane == at) { f = x; f_qry = xq; f_slot = sl | 0x100; }
        maxF = qe_uni(f, nf - 1);
    };

    // the first entry (:239)
    {
        const HNode r0 = nodes[hs];
        const Dist d0 = dist_add(dsrc, hnode_key(r0));
        QE x; x.sum = uni(d0.qry + d0.ref); x.anom = uni(d0.anom); x.qnz = uni(d0.qnz); x.qtot = uni(d0.qtot); x.node = hs; x.cur = 0; x.prev = -1; x.tag = 0;
        if (lane == 0) { knodes[0] = hs; kprev[0] = -1; kcq[0] = d0.qry; }
        nn = 1;
        f_insert(x, uni(d0.qry));
    }

    while (found < K) {                                              // :240-248
        const bool need_refill = hd == nf;
        if (need_refill && q.in == 0 && q.nruns == 0) break;         // queue empty
        if (q.in > EQ_IFLUSH || (need_refill && q.in > 0)) { wave_lds_sync(); eq_flush(q, K - found, lane); }
        if (need_refill) {
            // ---- the 64 smallest entries of the run heads
            QE cnd = qe_inf();
            for (int32_t l = 0; l <= q.lmax; l++) {
                const int32_t rh = uni(q.L->run_head[l]), re = uni(q.L->run_end[l]);
                if (rh >= re) continue;
                QE blk = qe_load(eq_slot(q, l, uni(q.L->run_slot[l])), rh, re, lane);
                blk.tag = l;
                qe_merge(cnd, blk, lane, false);
            }
            const uint64_t fin = wave_ballot(cnd.sum != INT64_MAX);
            nf = popc64(fin); hd = 0;
            for (int32_t l = 0; l <= q.lmax; l++) {
                const int32_t took = popc64(wave_ballot(cnd.sum != INT64_MAX && cnd.tag == l));
                if (took) {
                    const int32_t rh = uni(q.L->run_head[l]) + took;
                    if (lane == 0) q.L->run_head[l] = rh;
                    if (rh >= uni(q.L->run_end[l])) q.nruns--;
                }
            }
            wave_lds_sync();
            f = cnd; f.tag = 0;
            f_slot = lane | 0x100;
            freemask = nf >= 64 ? 0ull : ~((1ull << nf) - 1ull);
            wave_fence();
            f_qry = lane < nf ? kcq[f.cur] : 0;
            maxF = qe_uni(f, nf - 1);
        }
        // ---- successors of every entry of F that has none yet: heap node -> cross root -> keys
        if (__builtin_amdgcn_readlane(f_slot, hd) & 0x100) {
            if (lane >= hd && lane < nf && (f_slot & 0x100)) {
                const NodeQ ch = nodeq_load(nodes + f.node);
                const int32_t cl = ch.q2.x, cr = ch.q2.y;
                const int32_t hv = h[ch.q2.w];
                const int32_t sid[3] = {hv, cl, cr};
                const int64_t cq = mk64(ch.q0.x, ch.q0.y), cs = cq + mk64(ch.q0.z, ch.q0.w);
                const int32_t sl = f_slot & 63;
                AASM_UNROLL
                for (int j = 0; j < 3; j++) {
                    I4 a, b;
                    a.x = a.y = a.z = a.w = 0; b.x = b.y = b.z = 0; b.w = -1;
                    if (sid[j] >= 0) {
                        const I4 *pn = (const I4 *)(nodes + sid[j]);
                        const I4 n0 = pn[0], n1 = pn[1];
                        int64_t dq = mk64(n0.x, n0.y), ds = dq + mk64(n0.z, n0.w);
                        int32_t da = n1.x, dn = n1.y, dt = n1.z;
                        if (j > 0) { dq -= cq; ds -= cs; da -= ch.q1.x; dn -= ch.q1.y; dt -= ch.q1.z; }   // same heap: add the difference (:246-247)
                        a.x = lo32(ds); a.y = hi32(ds); a.z = lo32(dq); a.w = hi32(dq); b.x = da; b.y = dn; b.z = dt; b.w = sid[j];
                    }
                    q.L->succ[sl][j][0] = a; q.L->succ[sl][j][1] = b;
                }
                f_slot = sl;
            }
            wave_lds_sync();
        }
        // ---- pop (:241-244)
        const QE top = qe_uni(f, hd);
        const int64_t tq = mk64(__builtin_amdgcn_readlane(lo32(f_qry), hd), __builtin_amdgcn_readlane(hi32(f_qry), hd));
        const int32_t tslot = __builtin_amdgcn_readlane(f_slot, hd) & 63;
        hd++;
        freemask |= 1ull << tslot;
        if (lane == 0) {
            Dist dtop; dtop.qry = tq; dtop.ref = top.sum - tq; dtop.anom = top.anom; dtop.qnz = top.qnz; dtop.qtot = top.qtot; dtop.pad = 0;
            kd[found] = dtop; klast[found] = top.cur;
        }
        found++;
        // ---- its successors, one per lane (:245-247): cross heap root, left, right
        QE x = qe_inf(); int64_t xq = 0;
        bool valid = false;
        if (lane < 3) {
            const I4 a = q.L->succ[tslot][lane][0], b = q.L->succ[tslot][lane][1];
            valid = b.w >= 0;
            x.sum = top.sum + mk64(a.x, a.y); xq = tq + mk64(a.z, a.w);
            x.anom = top.anom + b.x; x.qnz = top.qnz + b.y; x.qtot = top.qtot + b.z; x.node = b.w; x.prev = lane == 0 ? top.cur : top.prev;
        }
        const uint64_t vm = wave_ballot(valid);
        if (vm == 0) continue;
        x.cur = nn + popc64(vm & lanemask_lt(lane));
        if (valid) { knodes[x.cur] = x.node; kprev[x.cur] = x.prev; kcq[x.cur] = xq; }
        nn += popc64(vm);
        const bool room = q.in == 0 && q.nruns == 0 && (nf - hd) < 64;       // nothing behind F: F may grow at its end
        const uint64_t fm = wave_ballot(valid && hd < nf && qe_less_v(x, maxF));
        if (fm == 0 && !room) {                                      // the usual case: all of them go to I
            if (valid) {
                const int32_t at = q.in + popc64(vm & lanemask_lt(lane));
                I4 a, b;
                a.x = lo32(x.sum); a.y = hi32(x.sum); a.z = x.anom; a.w = x.qnz; b.x = x.qtot; b.y = x.node; b.z = x.cur; b.w = x.prev;
                q.L->ibuf[at][0] = a; q.L->ibuf[at][1] = b;
            }
            q.in += popc64(vm);
        } else {
            for (uint64_t m = vm; m; m &= m - 1) {
                const int j = ffs64(m) - 1;
                const QE xj = qe_uni(x, j);
                const int64_t xjq = mk64(__builtin_amdgcn_readlane(lo32(xq), j), __builtin_amdgcn_readlane(hi32(xq), j));
                const bool fits = hd < nf ? qe_less_s(xj, maxF) : false;
                if (fits || (q.in == 0 && q.nruns == 0 && (nf - hd) < 64)) f_insert(xj, xjq);
                else {
                    if (lane == 0) {
                        I4 a, b;
                        a.x = lo32(xj.sum); a.y = hi32(xj.sum); a.z = xj.anom; a.w = xj.qnz; b.x = xj.qtot; b.y = xj.node; b.z = xj.cur; b.w = xj.prev;
                        q.L->ibuf[q.in][0] = a; q.L->ibuf[q.in][1] = b;
                    }
                    q.in++;
                }
            }
        }
    }
    if (lane == 0) {
        w.kfound[c] = found;
        atomic_add(&w.counters[CNT_PATHS], (int64_t)found);
        atomic_add(&w.counters[CNT_PQ_PUSH], (int64_t)nn);
    }
}

}  // namespace aasm
#endif
