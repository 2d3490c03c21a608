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
//      placed with one compare + ballot + one DPP wave shift of the 7 registers of an entry.
//  I   insertion buffer (LDS, 64 entries, unsorted): every other push, three lanes at a time.
//  R   sorted runs in global memory, one per level of a binary LSM tree (run of level l: <= 64 << l
//      entries).  A full I is sorted by a 21-stage bitonic network across the lanes and merged down
//      the levels: two sorted runs are merged 64 entries per step (the upper half of a 128-entry
//      bitonic merge is carried, the next 2 KB block comes from the run whose last loaded entry is
//      smaller; the next block of either run is already in flight), all loads and stores whole
//      coalesced blocks.  An entry at position p of a sorted run has p smaller entries in front of
//      it, so a merged run is CUT at K - found entries: what lies behind can never be popped.  The
//      cut also yields a bound: once a run holds K - found entries, a push that is not smaller than
//      its last one can never be popped either and is dropped at once (it still gets its insertion
//      index and its candidate record, as every push of the reference does).
//  refill (F ran empty): the 64 smallest of the run heads (one block per level, merged keeping the
//      lower half, each entry tagged with its level so the heads can be advanced).
//
// FAR TIER (round 4).  Every pop pushes up to three successors, 30 000 for K = 10 000, and two thirds of them are never
// popped: sorting them into runs was 47 % of the kernel.  A pushed score sum is never below the popped one (sidetrack keys
// are >= 0, heap children >= their parent), so the queue is split by a threshold T on the score sum: entries with sum <= T
// live in F / I / R as above; entries with sum > T are APPENDED, unsorted, to a far buffer in global memory - one coalesced
// store, no sort, no merge.  When the near tier runs empty the far buffer is scanned once: T moves up to the lower quartile
// of a 64-entry sample of it, the entries at or below the new T go through I into the runs, the rest are compacted in
// place (entries that cannot be popped any more - not below the bound - are dropped on the way).  On the bench's graphs
// 11-13 k of the 30 k pushes ever reach the sorted tier, in ~8 scans of ~19 k entries altogether; 18 k stay where they
// were appended.  The pop sequence is unchanged: near entries are all smaller than far ones.
//
// Order key.  (Distance, node, index) compares the score sum, then anom, then the ratio
// qul_nonzero / qul_total (higher first, by cross-multiplication: paf_data.hpp:142-159), then node and
// index.  Cross-multiplying in every compare-exchange of the networks would cost four quarter-rate
// multiplies per stage, so a push computes ONE 64-bit key2 = anom << 42 | (2^42 - 1 - r) with
// r = trunc(RN_double(qnz * 2^41 / max(qtot, 1))): qtot <= path edges <= N + 1 < 2^20 (the host picks the
// heap form for a batch with a longer contig), so two different ratios differ by more than 2^-40, their
// correctly rounded quotients by more than 1, and equal ratios give the same quotient: r orders exactly
// as the cross-multiplication does.  An entry is then three unsigned 64-bit words (sum, key2, node:index),
// compared lexicographically.
//
// The memory latency of a pop (heap node of the popped entry -> root of the cross heap -> keys of the
// <= 3 successors: three dependent round trips) is paid once per REFILL for 64 entries at a time:
// every lane of F fetches the successor keys of its own entry and parks them in LDS (128 B per entry);
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
AASM_HD int64_t enum_far_off(int64_t K) { return 128 * (((int64_t)1 << enum_lmax(K)) - 1) + 3 * enum_k64(K); }   // the runs; behind them the far tier
AASM_HD int64_t enum_stride(int64_t K) { return enum_far_off(K) + 3 * enum_k64(K) + 64; }                            // (every push may land there: <= 3 K)
#define AASM_ENUM_MAX_N (((int64_t)1 << 20) - 2)     // longest contig (records) the key2 form is exact for
}  // namespace aasm

#if !defined(AASM_HOST_EMUL)

namespace aasm {

struct __attribute__((aligned(8))) I2 { int32_t x, y; };
#define EQ_ILEN 64
#define EQ_BATCH 16                      // entries popped per step at most (3 successors each: 48 lanes)
#define EQ_IFLUSH 40                     // flush I when it holds more than this (a step of P pops adds <= 3 P entries; P <= (64 - in) / 3)
#define EQ_MAXLEV 28
template <int FMAX> struct EnumLdsT {
    // per F slot, nine quads.  The entry itself: [0] {qry (2 words), anom, qnz}, [1].xy {qtot, prev}.  Its successors (cross heap
    // root, left, right), COMPLETE but for the insertion index: j at [2 + 2j] {sum (2), key2 (2)}, [3 + 2j] {qry (2), heap node | -1, anom},
    // and their {qnz, qtot} pairs at [1].zw, [8].xy, [8].zw
    I4 slot[FMAX][9];
    I2 ibuf[EQ_ILEN][3];                 // I: {sum}, {key2}, {index, node}
    int32_t run_slot[EQ_MAXLEV], run_head[EQ_MAXLEV], run_end[EQ_MAXLEV];
    int32_t freel[64];                   // stack of the F slots no entry uses
};
// F holds up to FMAX entries: 64 (11.1 KB of LDS per wave: 14 waves per CU), or 40 (7.7 KB: 20 waves per CU, i.e. the 5 per SIMD
// that keep 5 000 contigs resident at once - a batch of 4 000 took 28 ms against 22.6 ms for 3 500, the residency cliff of the 64 form)
#define AASM_ENUM2_LDS_BYTES_F(FMAX) ((FMAX) * 144 + EQ_ILEN * 24 + 3 * EQ_MAXLEV * 4 + 256)
#define AASM_ENUM2_LDS_BYTES AASM_ENUM2_LDS_BYTES_F(64)
#define EQ_FSMALL 40
typedef EnumLdsT<64> EnumLds;
static_assert(sizeof(EnumLdsT<64>) <= AASM_ENUM2_LDS_BYTES_F(64) && sizeof(EnumLdsT<EQ_FSMALL>) <= AASM_ENUM2_LDS_BYTES_F(EQ_FSMALL), "LDS budget");

struct QE { uint64_t sum, key2, nc; int32_t tag; };                  // nc = node << 32 | insertion index; tag: refill only (source level)
#define QE_INF_SUM 0x7fffffffffffffffull
AASM_DEV QE qe_inf() { QE e; e.sum = QE_INF_SUM; e.key2 = 0; e.nc = 0; e.tag = 0; return e; }
AASM_DEV bool qe_less(const QE &a, const QE &b) {                    // score sums are non-negative: unsigned order == signed order
    return (a.sum < b.sum) | ((a.sum == b.sum) & ((a.key2 < b.key2) | ((a.key2 == b.key2) & (a.nc < b.nc))));
}
AASM_DEV bool qe_less_u(const QE &a, const QE &b) {                  // the same on wave-uniform values
    if (a.sum != b.sum) return a.sum < b.sum;
    if (a.key2 != b.key2) return a.key2 < b.key2;
    return a.nc < b.nc;
}
AASM_DEV uint64_t qe_key2(int32_t anom, int32_t qnz, int32_t qtot) {
    const double r = (double)((uint64_t)(uint32_t)qnz << 41) / (double)(qtot ? qtot : 1);
    return ((uint64_t)(uint32_t)anom << 42) | ((((uint64_t)1 << 42) - 1) - (uint64_t)r);
}

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
template <int M> AASM_DEV uint64_t lane_xor64(uint64_t x, int lane) { return mk64(lane_xor<M>(lo32(x), lane), lane_xor<M>(hi32(x), lane)); }
template <int M> AASM_DEV QE qe_xor(const QE &e, int lane) {
    QE p;
    p.sum = lane_xor64<M>(e.sum, lane); p.key2 = lane_xor64<M>(e.key2, lane); p.nc = lane_xor64<M>(e.nc, lane); p.tag = lane_xor<M>(e.tag, lane);
    return p;
}
AASM_DEV QE qe_sel(bool take, const QE &p, const QE &e) {
    QE r;
    r.sum = take ? p.sum : e.sum; r.key2 = take ? p.key2 : e.key2; r.nc = take ? p.nc : e.nc; r.tag = take ? p.tag : e.tag;
    return r;
}
// compare-exchange with lane ^ M: the lower lane of a pair keeps the smaller entry (entries are distinct, except
// +inf padding, which may swap with itself)
template <int M> AASM_DEV void qe_cx(QE &e, int lane) {
    constexpr int HB = M >= 32 ? 32 : M >= 16 ? 16 : M >= 8 ? 8 : M >= 4 ? 4 : M >= 2 ? 2 : 1;
    const QE p = qe_xor<M>(e, lane);
    const bool upper = (lane & HB) != 0;
    e = qe_sel(qe_less(p, e) != upper, p, e);
}
AASM_DEV void qe_sort64(QE &e, int lane) {                          // bitonic sort, flip formulation: 21 stages
    qe_cx<1>(e, lane);
    qe_cx<3>(e, lane); qe_cx<1>(e, lane);
    qe_cx<7>(e, lane); qe_cx<2>(e, lane); qe_cx<1>(e, lane);
    qe_cx<15>(e, lane); qe_cx<4>(e, lane); qe_cx<2>(e, lane); qe_cx<1>(e, lane);
    qe_cx<31>(e, lane); qe_cx<8>(e, lane); qe_cx<4>(e, lane); qe_cx<2>(e, lane); qe_cx<1>(e, lane);
    qe_cx<63>(e, lane); qe_cx<16>(e, lane); qe_cx<8>(e, lane); qe_cx<4>(e, lane); qe_cx<2>(e, lane); qe_cx<1>(e, lane);
}
// x, y sorted ascending (one entry per lane) -> x = the 64 smallest of both, y = the 64 largest, both sorted: the flip
// stage leaves two bitonic sequences, six half-cleaners each sort them (interleaved: two independent chains)
AASM_DEV void qe_merge(QE &x, QE &y, int lane, bool want_hi) {
    const QE yr = qe_xor<63>(y, lane);
    const bool sw = qe_less(yr, x);
    const QE lo = qe_sel(sw, yr, x), hi = qe_sel(sw, x, yr);
    x = lo; y = hi;
    if (want_hi) {
        qe_cx<32>(x, lane); qe_cx<32>(y, lane); qe_cx<16>(x, lane); qe_cx<16>(y, lane); qe_cx<8>(x, lane); qe_cx<8>(y, lane);
        qe_cx<4>(x, lane); qe_cx<4>(y, lane); qe_cx<2>(x, lane); qe_cx<2>(y, lane); qe_cx<1>(x, lane); qe_cx<1>(y, lane);
    } else {
        qe_cx<32>(x, lane); qe_cx<16>(x, lane); qe_cx<8>(x, lane); qe_cx<4>(x, lane); qe_cx<2>(x, lane); qe_cx<1>(x, lane);
    }
}
AASM_DEV uint64_t uni_u64(uint64_t x, int j) { return mk64(__builtin_amdgcn_readlane(lo32(x), j), __builtin_amdgcn_readlane(hi32(x), j)); }
AASM_DEV QE qe_uni(const QE &e, int j) {                            // the entry lane j holds (wave-uniform j)
    QE r; r.sum = uni_u64(e.sum, j); r.key2 = uni_u64(e.key2, j); r.nc = uni_u64(e.nc, j); r.tag = 0; return r;
}
// a 64-entry block of a run: entry (at + lane), or +inf behind the end.  32 bytes per entry: {sum, key2}, {node:index, -}
AASM_DEV QE qe_load(const PqK *g, int32_t at, int32_t end, int lane) {
    QE e = qe_inf();
    if (at + lane < end) {
        const I4 *p = (const I4 *)(g + at + lane);
        const I4 a = p[0];
        const I2 b = *(const I2 *)(p + 1);
        e.sum = mk64(a.x, a.y); e.key2 = mk64(a.z, a.w); e.nc = mk64(b.x, b.y);
    }
    return e;
}
AASM_DEV void qe_store(PqK *g, int32_t at, const QE &e) {
    I4 *p = (I4 *)(g + at);
    I4 a; I2 b;
    a.x = lo32(e.sum); a.y = hi32(e.sum); a.z = lo32(e.key2); a.w = hi32(e.key2); b.x = lo32(e.nc); b.y = hi32(e.nc);
    p[0] = a; *(I2 *)(p + 1) = b;
}
AASM_DEV void qe_to_lds(I2 *dst, const QE &e) {
    I2 a, b, c;
    a.x = lo32(e.sum); a.y = hi32(e.sum); b.x = lo32(e.key2); b.y = hi32(e.key2); c.x = lo32(e.nc); c.y = hi32(e.nc);
    dst[0] = a; dst[1] = b; dst[2] = c;
}

template <int FMAX> struct EnumQT {
    EnumLdsT<FMAX> *L;
    PqK *g;                              // run storage of this contig
    int32_t lmax, k64;
    int32_t in;                          // entries in I
    int32_t nruns;                       // non-empty levels
    QE bound;                            // nothing >= bound can still be popped (+inf: no bound yet)
    uint64_t T;                          // far tier: entries with sum > T sit, unsorted, in far[0 .. far_n)
    PqK *far;
    int32_t far_n;
};
template <class Q> AASM_DEV int32_t eq_cap(const Q &q, int32_t l) { const int64_t c = (int64_t)64 << l; return l < q.lmax && c < q.k64 ? (int32_t)c : q.k64; }
template <class Q> AASM_DEV PqK *eq_slot(const Q &q, int32_t l, int32_t s) {
    const int32_t lb = l < q.lmax ? l : q.lmax;
    return q.g + 128 * (((int64_t)1 << lb) - 1) + (int64_t)s * eq_cap(q, l);
}

// out[0 .. out_len) <- the out_len smallest of (carry, if has_carry) + A[0 .. a_len) + B[0 .. b_len); the inputs are sorted.
// With has_carry the A run is not read (a_len = 0): the carry is the sorted block in registers.  Returns the last
// entry written when the output was cut short of the input (the queue's bound), +inf otherwise.
AASM_DEV QE eq_merge(const PqK *A, int32_t a_len, const PqK *B, int32_t b_len, PqK *out, int32_t out_len, QE carry, int32_t c_len, bool has_carry, int lane) {
    const bool cut = out_len < (has_carry ? c_len : a_len) + b_len;
    int32_t ia = 0, ib = 0, o = 0;
    QE lastA = qe_inf(), lastB = qe_inf();
    lastB.sum = 0; lastB.key2 = 0;                                   // "nothing loaded yet": the first block comes from B
    if (!has_carry) { carry = qe_load(A, 0, a_len, lane); ia = 64; lastA = qe_uni(carry, 63); } else a_len = 0;
    QE nA = qe_load(A, ia, a_len, lane), nB = qe_load(B, 0, b_len, lane);   // the next block of either run, in flight
    QE last = qe_inf();
    while (o < out_len) {
        const bool availA = ia < a_len, availB = ib < b_len;
        if (!availA && !availB) break;
        // next block: from the run whose last loaded entry is smaller (all entries not loaded yet are then >= every
        // entry this step emits); an exhausted run leaves the choice to the other one
        const bool takeA = availA && (!availB || qe_less(lastA, lastB));
        QE blk;
        if (takeA) { blk = nA; ia += 64; lastA = qe_uni(blk, 63); nA = qe_load(A, ia, a_len, lane); }
        else { blk = nB; ib += 64; lastB = qe_uni(blk, 63); nB = qe_load(B, ib, b_len, lane); }
        qe_merge(carry, blk, lane, true);
        if (o + lane < out_len) qe_store(out, o + lane, carry);
        if (cut && out_len - o <= 64) last = qe_uni(carry, out_len - o - 1);
        o += 64;
        carry = blk;
    }
    if (o + lane < out_len) qe_store(out, o + lane, carry);          // (out_len never exceeds the number of real entries)
    if (cut && o < out_len) last = qe_uni(carry, out_len - o - 1);
    wave_fence();                                                    // later block loads of this wave see the run
    return last;
}

// I -> a sorted block, merged down the levels of the LSM tree (one run per level at rest; level l < lmax has two
// slots of 64 << l entries, the top level three of k64: the output of a merge goes to a slot none of its inputs uses)
template <class Q> AASM_DEV void eq_flush(Q &q, int32_t keep, int lane) {
    QE s = qe_inf();
    if (lane < q.in) {
        const I2 a = q.L->ibuf[lane][0], b = q.L->ibuf[lane][1], c = q.L->ibuf[lane][2];
        s.sum = mk64(a.x, a.y); s.key2 = mk64(b.x, b.y); s.nc = mk64(c.x, c.y);
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
        const QE last = eq_merge(eq_slot(q, l, cslot), in_regs ? 0 : len, eq_slot(q, l, os) + oh, oe - oh, eq_slot(q, nl, ns), tot, s, len, in_regs, lane);
        if (last.sum != QE_INF_SUM && qe_less(last, q.bound)) q.bound = last;   // the run was cut at K - found entries: its last entry bounds the queue
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

// The near tier is empty: move T up and bring the far entries at or below it into the runs (through I).  T = the lower
// quartile of a 64-entry sample of the far buffer (any value of the sample makes progress: that entry itself moves).
template <class Q> AASM_DEV void eq_redistribute(Q &q, int32_t keep, int lane) {
    wave_fence();                                                    // the far entries were stored by lanes of this wave
    const int32_t n = q.far_n;
    const int32_t ns = n < 64 ? n : 64;
    QE sm = qe_inf();
    if (lane < ns) { const int32_t at = n < 64 ? lane : (int32_t)(((int64_t)lane * n) >> 6); sm = qe_load(q.far + at, 0, 1, 0); sm.key2 = 0; sm.nc = (uint64_t)lane; }
    qe_sort64(sm, lane);
    q.T = uni_u64(sm.sum, (ns + 3) / 4 - 1);
    int32_t wr = 0;
    for (int32_t base = 0; base < n; base += 64) {
        const QE e = qe_load(q.far, base, n, lane);
        const bool live = base + lane < n && qe_less(e, q.bound);    // (what is not below the bound can never be popped)
        const bool near = live && e.sum <= q.T;
        const uint64_t nm = wave_ballot(near), sm2 = wave_ballot(live && !near);
        const int32_t cnt = popc64(nm);
        if (cnt) {
            if (q.in + cnt > EQ_ILEN) { wave_lds_sync(); eq_flush(q, keep, lane); }
            if (near) qe_to_lds(q.L->ibuf[q.in + popc64(nm & lanemask_lt(lane))], e);
            q.in += cnt;
        }
        if (live && !near) qe_store(q.far, wr + popc64(sm2 & lanemask_lt(lane)), e);   // in place: wr + rank <= base + lane, and the block is in registers
        wr += popc64(sm2);
    }
    q.far_n = wr;
    wave_lds_sync();
    wave_fence();
}

template <int FMAX> AASM_DEV void kb_enum_lsm(const KCtx &k, const WS &w) {   // one wave per contig
    const int64_t c = k.bid;
    const int64_t V = w.ctgV[c];
    const int lane = k.lane;
    if (lane == 0) w.kfound[c] = 0;
    if (V == 0 || w.status[c] != 0) return;
    const int64_t vb = w.voff[c];
    const int32_t K = w.K;
    Dist *kd = w.kd + c * (int64_t)K;
    int32_t *klast = w.klast + c * (int64_t)K;
    I4 *kcand = w.kcand + 2 * c * (3 * (int64_t)K + 1);
    const HNode *nodes = heap_arena(w, c);
    const int32_t *h = w.h_root + vb;
    const int32_t src = (int32_t)(V - 2);
    EnumQT<FMAX> q;
    q.L = (EnumLdsT<FMAX> *)k.lds; q.g = w.pq + c * w.pq_stride; q.lmax = enum_lmax(K); q.k64 = (int32_t)enum_k64(K); q.in = 0; q.nruns = 0; q.bound = qe_inf();
    q.T = 0; q.far = q.g + enum_far_off(K); q.far_n = 0;
    if (lane < EQ_MAXLEV) { q.L->run_slot[lane] = 0; q.L->run_head[lane] = 0; q.L->run_end[lane] = 0; }
    wave_lds_sync();

    const Dist dsrc = w.sp_d[vb + src];
    if (lane == 0) { kd[0] = dsrc; klast[0] = -1; }                  // :217-220
    int32_t found = 1, nn = 0;
    const int32_t hs = uni(h[src]);
    if (hs < 0) { if (lane == 0) { w.kfound[c] = 1; atomic_add(&w.counters[CNT_PATHS], (int64_t)1); } return; }   // :227-228

    // ---- F: sorted entries in lanes [hd, nf); f_slot = LDS slot of the entry | 0x100 while its successors are not fetched yet
    QE f = qe_inf();
    int32_t f_slot = 0;
    int32_t hd = 0, nf = 0;
    int32_t ftop = FMAX;                                             // free slots: q.L->freel[0 .. ftop)
    if (lane < FMAX) q.L->freel[lane] = FMAX - 1 - lane;
    QE maxF = qe_inf();

    // x (wave-uniform; own = its {qry, anom, qnz}, {qtot, prev} words) into F; the caller has decided that it belongs there
    auto f_insert = [&](const QE &x, const I4 &own0, const I4 &own1) {
        const int32_t below = popc64(wave_ballot(lane >= hd && lane < nf && qe_less(f, x)));
        const int32_t pos = hd + below;
        if (nf - hd == FMAX) {                                       // full: the largest entry leaves for I
            if (lane == nf - 1) { qe_to_lds(q.L->ibuf[q.in], f); q.L->freel[ftop] = f_slot & 63; }
            ftop++; q.in++; nf--;
        }
        const int32_t sl = uni(q.L->freel[--ftop]);
        bool mv; int32_t at;
        QE t; int32_t ts;
#define EQ_SH(CTRL, v) __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false)
#define EQ_SHIFT(CTRL) do { \
            t.sum = mk64(EQ_SH(CTRL, lo32(f.sum)), EQ_SH(CTRL, hi32(f.sum))); t.key2 = mk64(EQ_SH(CTRL, lo32(f.key2)), EQ_SH(CTRL, hi32(f.key2))); \
            t.nc = mk64(EQ_SH(CTRL, lo32(f.nc)), EQ_SH(CTRL, hi32(f.nc))); t.tag = 0; ts = EQ_SH(CTRL, f_slot); } while (0)
        if (nf < 64) {                                               // lanes [pos, nf) move up by one
            EQ_SHIFT(0x138);                                         // wave_shr:1: the value of lane - 1
            mv = lane > pos && lane <= nf; at = pos; nf++;
        } else {                                                     // room only below the head: lanes [hd, pos) move down by one
            EQ_SHIFT(0x130);                                         // wave_shl:1: the value of lane + 1
            mv = lane >= hd - 1 && lane < pos - 1; at = pos - 1; hd--;
        }
#undef EQ_SHIFT
#undef EQ_SH
        f = qe_sel(mv, t, f); f_slot = mv ? ts : f_slot;
        if (lane == at) { f = x; f_slot = sl | 0x100; q.L->slot[sl][0] = own0; q.L->slot[sl][1] = own1; }
        maxF = qe_uni(f, nf - 1);
    };

    // the first entry (:239)
    {
        const HNode r0 = nodes[hs];
        const Dist d0 = dist_add(dsrc, hnode_key(r0));
        QE x; x.sum = (uint64_t)uni(d0.qry + d0.ref); x.key2 = uni_u64(qe_key2(d0.anom, d0.qnz, d0.qtot), 0); x.nc = mk64(0, hs); x.tag = 0;
        I4 o0, o1;
        o0.x = uni(lo32((uint64_t)d0.qry)); o0.y = uni(hi32((uint64_t)d0.qry)); o0.z = uni(d0.anom); o0.w = uni(d0.qnz); o1.x = uni(d0.qtot); o1.y = -1; o1.z = 0; o1.w = 0;
        if (lane == 0) { I4 cd; cd.x = hs; cd.y = -1; cd.z = o0.x; cd.w = o0.y; kcand[0] = cd; I4 ce; ce.x = o0.z; ce.y = o0.w; ce.z = o1.x; ce.w = 0; kcand[1] = ce; }
        nn = 1;
        q.T = x.sum;                                                 // the near tier starts with the first entry's score sum
        wave_lds_sync();
        f_insert(x, o0, o1);
    }

    const int32_t pi = (lane * 86) >> 8, pj = lane - 3 * pi;         // lane = 3 * (pop of the step) + (successor of that pop)
    KPROF_DECL;
    KPROF_START();
    while (found < K) {                                              // :240-248
        KPROF_STAMP(0);                                              // pops + pushes
        const bool need_refill = hd == nf;
        if (need_refill && q.in == 0 && q.nruns == 0) {
            if (q.far_n == 0) break;                                 // queue empty
            eq_redistribute(q, K - found, lane);                     // near tier empty: the next slice of the far one
            KPROF_STAMP(4);
            if (q.in == 0 && q.nruns == 0 && q.far_n == 0) break;    // (everything left was beyond the bound)
        }
        if (q.in > EQ_IFLUSH || (need_refill && q.in > 0)) { wave_lds_sync(); eq_flush(q, K - found, lane); KPROF_STAMP(1); }
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
            const uint64_t fin = wave_ballot(cnd.sum != QE_INF_SUM && lane < FMAX);
            nf = popc64(fin); hd = 0;
            if (FMAX < 64 && lane >= FMAX) cnd = qe_inf();          // (the entries behind the front's capacity stay in their runs)
            for (int32_t l = 0; l <= q.lmax; l++) {
                const int32_t took = popc64(wave_ballot(cnd.sum != QE_INF_SUM && cnd.tag == l));
                if (took) {
                    const int32_t rh = uni(q.L->run_head[l]) + took;
                    if (lane == 0) q.L->run_head[l] = rh;
                    if (rh >= uni(q.L->run_end[l])) q.nruns--;
                }
            }
            // a tighter bound: a run that still holds K - found - |F| entries behind the new front has, at that place, an entry with
            // K - found smaller ones in front of it (F's and its own): nothing from there on can be popped any more
            {
                int32_t need = K - found - nf;
                if (need < 1) need = 1;
                for (int32_t l = q.lmax; l >= 0 && l >= q.lmax - 1; l--) {
                    const int32_t rh = uni(q.L->run_head[l]), re = uni(q.L->run_end[l]);
                    if (re - rh < need) continue;
                    const QE cand = qe_uni(qe_load(eq_slot(q, l, uni(q.L->run_slot[l])), rh + need - 1 - lane, re, lane), 0);   // (lane 0 reads entry rh + need - 1)
                    if (qe_less_u(cand, q.bound)) q.bound = cand;
                    break;
                }
            }
            f = cnd; f.tag = 0;
            f_slot = lane | 0x200;                                   // 0x200: its own words are still in the candidate record
            ftop = FMAX - nf;
            if (lane < ftop) q.L->freel[lane] = FMAX - 1 - lane;     // slots nf .. FMAX - 1
            wave_lds_sync();
            maxF = qe_uni(f, nf - 1);
            wave_fence();                                            // candidate records written by lanes of this wave
            KPROF_STAMP(2);
        }
        // ---- how many entries this step pops: all of them must be there, fit K, and leave room in I for their successors
        int32_t P = nf - hd;
        if (P > EQ_BATCH) P = EQ_BATCH;
        if (P > K - found) P = K - found;
        if (3 * P > EQ_ILEN - q.in) P = (EQ_ILEN - q.in) / 3;
        // ---- successors of every entry of F that has none yet: heap node -> cross root -> keys; each lane builds the complete
        //      queue entries of its own entry's successors (their insertion index comes at the pop)
        if (wave_ballot(lane >= hd && lane < hd + P && (f_slot & 0x300)) != 0) {
            if (lane >= hd && lane < nf && (f_slot & 0x300)) {
                const int32_t sl = f_slot & 63;
                const int32_t fnode = hi32(f.nc), fcur = lo32(f.nc);
                I4 o0, o1;
                if (f_slot & 0x200) {
                    const I4 c0 = kcand[2 * (int64_t)fcur], c1 = kcand[2 * (int64_t)fcur + 1];
                    o0.x = c0.z; o0.y = c0.w; o0.z = c1.x; o0.w = c1.y; o1.x = c1.z; o1.y = c0.y;
                    q.L->slot[sl][0] = o0;
                } else { o0 = q.L->slot[sl][0]; o1 = q.L->slot[sl][1]; }
                const uint64_t oq = mk64(o0.x, o0.y);
                const NodeQ ch = nodeq_load(nodes + fnode);
                const int32_t cl = ch.q2.x, cr = ch.q2.y;
                const int32_t hv = h[ch.q2.w];
                const int32_t sid[3] = {hv, cl, cr};
                const uint64_t cq = mk64(ch.q0.x, ch.q0.y), cs = cq + mk64(ch.q0.z, ch.q0.w);
                int32_t pn[3], pt[3];
                AASM_UNROLL
                for (int j = 0; j < 3; j++) {
                    I4 a, b;
                    a.x = a.y = a.z = a.w = 0; b.x = b.y = b.w = 0; b.z = -1; pn[j] = 0; pt[j] = 0;
                    if (sid[j] >= 0) {
                        const I4 *np_ = (const I4 *)(nodes + sid[j]);
                        const I4 n0 = np_[0], n1 = np_[1];
                        uint64_t dq = mk64(n0.x, n0.y), ds = dq + mk64(n0.z, n0.w);
                        int32_t da = n1.x, dn = n1.y, dt = n1.z;
                        if (j > 0) { dq -= cq; ds -= cs; da -= ch.q1.x; dn -= ch.q1.y; dt -= ch.q1.z; }   // same heap: add the difference (:246-247)
                        const uint64_t ssum = f.sum + ds, sq = oq + dq;
                        const int32_t sa = o0.z + da, sn = o0.w + dn, st = o1.x + dt;
                        a.x = lo32(ssum); a.y = hi32(ssum); a.z = 0; a.w = 0;   // (key2 - a double division - is made at the pop, by the lane that pushes the successor: one division for
                                                                                 //  the whole step instead of three in a row per fetched entry, which is what a one-pop-per-step contig pays)
                        b.x = lo32(sq); b.y = hi32(sq); b.z = sid[j]; b.w = sa; pn[j] = sn; pt[j] = st;
                    }
                    q.L->slot[sl][2 + 2 * j] = a; q.L->slot[sl][3 + 2 * j] = b;
                }
                o1.z = pn[0]; o1.w = pt[0];
                I4 p8; p8.x = pn[1]; p8.y = pt[1]; p8.z = pn[2]; p8.w = pt[2];
                q.L->slot[sl][1] = o1; q.L->slot[sl][8] = p8;
                f_slot = sl;
            }
            wave_lds_sync();
#if defined(AASM_KPROF)
            __builtin_amdgcn_s_waitcnt(0);
#endif
            KPROF_STAMP(3);
        }
        // ---- the successors of the P front entries (:245-247), lane 3 i + j: successor j (cross heap root, left, right) of front entry i
        const int32_t srcl = (hd + pi) << 2;
        const int32_t pslot = __builtin_amdgcn_ds_bpermute(srcl, f_slot) & 63;
        const int32_t pcur = __builtin_amdgcn_ds_bpermute(srcl, lo32(f.nc));
        QE x = qe_inf();
        I4 x0, x1;
        x0.x = x0.y = x0.z = x0.w = 0; x1.x = x1.y = x1.z = x1.w = 0;
        bool valid = false;
        if (pi < P) {
            const I4 a = q.L->slot[pslot][2 + 2 * pj], b = q.L->slot[pslot][3 + 2 * pj], t1 = q.L->slot[pslot][1], t8 = q.L->slot[pslot][8];
            valid = b.z >= 0;
            x.sum = mk64(a.x, a.y); x.nc = (uint64_t)(uint32_t)b.z << 32;
            x0.x = b.x; x0.y = b.y; x0.z = b.w; x0.w = pj == 0 ? t1.z : pj == 1 ? t8.x : t8.z;
            x1.x = pj == 0 ? t1.w : pj == 1 ? t8.y : t8.w; x1.y = pj == 0 ? pcur : t1.y;
            x.key2 = qe_key2(x0.z, x0.w, x1.x);
        }
        // a successor smaller than a front entry behind its parent has to be popped before that entry: the step ends with its parent
        if (P > 1) {
            const QE elast = qe_uni(f, hd + P - 1);
            const uint64_t viol = wave_ballot(valid && pi < P - 1 && qe_less(x, elast));
            if (viol) { P = (((ffs64(viol) - 1) * 86) >> 8) + 1; valid = valid && pi < P; }
        }
        // ---- pop them (:241-244)
        if (lane >= hd && lane < hd + P) {
            const int32_t sl = f_slot & 63;
            const I4 t0 = q.L->slot[sl][0];
            const int32_t tt = q.L->slot[sl][1].x;
            const uint64_t tq = mk64(t0.x, t0.y);
            Dist dtop; dtop.qry = (int64_t)tq; dtop.ref = (int64_t)(f.sum - tq); dtop.anom = t0.z; dtop.qnz = t0.w; dtop.qtot = tt; dtop.pad = 0;
            const int32_t at = found + (lane - hd);
            kd[at] = dtop; klast[at] = lo32(f.nc);
            q.L->freel[ftop + (lane - hd)] = sl;
        }
        ftop += P; hd += P; found += P;
        const uint64_t vm = wave_ballot(valid);
        if (vm == 0) continue;
        const int32_t xcur = nn + popc64(vm & lanemask_lt(lane));
        x.nc |= (uint32_t)xcur;
        if (valid) {
            I4 cd; cd.x = hi32(x.nc); cd.y = x1.y; cd.z = x0.x; cd.w = x0.y;
            I4 ce; ce.x = x0.z; ce.y = x0.w; ce.z = x1.x; ce.w = 0;
            kcand[2 * (int64_t)xcur] = cd; kcand[2 * (int64_t)xcur + 1] = ce;
        }
        nn += popc64(vm);
        const bool alive = valid && qe_less(x, q.bound);             // (the others can never be popped: dropped here)
        const bool isfar = alive && x.sum > q.T;                     // beyond the threshold: appended, unsorted
        const uint64_t fm = wave_ballot(isfar);
        if (fm) {
            if (isfar) qe_store(q.far, q.far_n + popc64(fm & lanemask_lt(lane)), x);
            q.far_n += popc64(fm);
        }
        const bool live = alive && !isfar;
        const uint64_t lm = wave_ballot(live);
        if (lm == 0) continue;
        const bool behind = q.in != 0 || q.nruns != 0;               // something behind F: an entry >= max(F) goes to I (else F may grow at its end)
        const bool toF = live && (!behind || (hd < nf && qe_less(x, maxF)));
        const uint64_t im = wave_ballot(live && !toF);
        if (im) {                                                    // the usual case: to I, all at once
            if (live && !toF) qe_to_lds(q.L->ibuf[q.in + popc64(im & lanemask_lt(lane))], x);
            q.in += popc64(im);
        }
        for (uint64_t m = wave_ballot(toF); m; m &= m - 1) {        // into F, one at a time
            const int j = ffs64(m) - 1;
            const QE xj = qe_uni(x, j);
            const bool fits = hd < nf ? qe_less_u(xj, maxF) : false;
            if (fits || (q.in == 0 && q.nruns == 0 && (nf - hd) < FMAX)) {
                I4 o0, o1;
                o0.x = __builtin_amdgcn_readlane(x0.x, j); o0.y = __builtin_amdgcn_readlane(x0.y, j); o0.z = __builtin_amdgcn_readlane(x0.z, j); o0.w = __builtin_amdgcn_readlane(x0.w, j);
                o1.x = __builtin_amdgcn_readlane(x1.x, j); o1.y = __builtin_amdgcn_readlane(x1.y, j); o1.z = 0; o1.w = 0;
                wave_lds_sync();
                f_insert(xj, o0, o1);
            } else {
                if (lane == 0) qe_to_lds(q.L->ibuf[q.in], xj);
                q.in++;
            }
        }
    }
#if defined(AASM_KPROF)
    if (w.K >= 1000) KPROF_FLUSH(w.prof_heap, c, lane);              // diagnostic build: K8's sections replace K7's in the dump
#endif
    if (lane == 0) {
        w.kfound[c] = found;
        atomic_add(&w.counters[CNT_PATHS], (int64_t)found);
        atomic_add(&w.counters[CNT_PQ_PUSH], (int64_t)nn);
    }
}

}  // namespace aasm
#endif
