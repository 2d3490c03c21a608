// aasm_kernels.h -- kernel bodies of the per-contig path-inference pipeline (gfx950).
//
// Stage map (reference file:line -> kernel), see DESIGN.md for layouts and byte counts:
//   K1  sort + parts ............. paf_data.cpp:241-261 ........ kb_sort, kb_sort_fix, kb_gather_parts
//   K2  overlap pairs + cut ...... paf_data.cpp:294-378 ........ kb_ov_count, kb_ov_merge, kb_vcount, kb_vfill_*
//   K3/4 linkable/get_score/make_Graph paf_data.cpp:422-696 .... kb_nsl, kb_row_count, kb_row_fill
//   --  reversed CSR ............. k_shortest_walks.hpp:180-183 . kb_rev_fill, kb_rev_place, kb_rev_hdr
//   K6  Kahn(rev)+DAG-SP ......... k_shortest_walks.hpp:132-175 . kb_rev_sweep
//   K5/6 Kahn(fwd)+anomaly ....... paf_data.cpp:704-713,742-746 . kb_fwd_sweep
//   K7  sidetrack heaps .......... k_shortest_walks.hpp:191-215, leftist_heap.hpp:29-40 .. kb_children, kb_sidetrack, kb_heap_hdr, kb_heap, kb_heap_mw
//   K8  enumeration .............. k_shortest_walks.hpp:217-249 . kb_enum
//   K9  recover/upgrade/select ... k_shortest_walks.hpp:254-290, paf_data.cpp:750-921,1489-1649 .. kb_select
//
// Data-parallel stages run one thread per record / slot / vertex over the WHOLE batch
// (coalesced SoA reads, ballot/prefix compaction inside rows).  The order-sensitive
// stages run ONE WAVE PER CONTIG: contigs are independent, a 64-lane wave walks one
// contig's topological order, and the lanes fan out over the adjacency row of the
// vertex being processed (distinct targets per row -> conflict-free read-modify-write).
#pragma once
#include <cstddef>
#include "aasm_dev.h"

namespace aasm {

struct PqK;
struct WS {
    // ---- sizes / options
    int64_t C, R, R0, S, VT, ET;
    int32_t K, nsl;
    int32_t avg_sidetracks;              // mean #sidetracks per contig of the batch (K7's wave priority)
    int32_t xcd_map;                     // kernels take their work items XCD by XCD (aasm_gpu.hip)
    int32_t mw_compact;                  // 1: kb_heap_mw moves its nodes to the final arena (debug runs: arena indices as the reference allocates them); 0: they stay where they were built
    int32_t sort_depth_test;             // 0; tests: kb_sort_fix starts its introsort with (this - 1) partition levels instead of 2 lg N
    // ---- input batch (device), original record order; rec_off already points at the chunk
    const int64_t *rec_off, *in_qs, *in_qe, *in_rs, *in_re, *in_qt, *in_rng_off, *rql, *rqr, *rrl;
    const int32_t *in_chr;
    const uint8_t *in_fwd, *in_mq;
    // ---- K0 (optional): match ranges derived on the device from the cs tags
    const char *cs_text;
    const int64_t *cs_off;
    int64_t *rng_rec;                    // K0's output: one 32-byte record {qry_l, qry_r, ref_l, -} per match range (a lane writes whole 32-byte sectors)
    int64_t rng_stride;                  // words between consecutive ranges behind rql / rqr / rrl: 1 (the caller's three arrays) or 4 (K0's records)
    int32_t *cs_bad;                     // 0, or (smallest record, relative to R0, whose tag is malformed) - INT32_MAX
    // ---- K1: sorted records, indexed by (sorted position + rec_off[c] - R0)
    int32_t *perm, *dupflag, *np, *pstart;
    int64_t *s_qs, *s_qe, *s_rs, *s_re, *s_qt, *s_rb;
    int32_t *s_rn, *s_chr, *s_orig, *s_ctg, *s_pid;
    uint8_t *s_fl;                       // bit0 aln_fwd, bit1 map_qul > 0
    // ---- K2: overlap slots (i, j) for j in (i, jmax_i]
    int32_t *ov_cnt, *ov_rec, *ov_vid;
    int64_t *ov_off, *ov_rank, *ov_peq, *ov_per, *ov_stq, *ov_str;
    uint8_t *ov_ok;
    int32_t *dis_end, *next_cnt;         // NON_SKIP_LINKABLE truncation points
    // ---- vertices
    int32_t *ctgV, *v_i, *v_j, *v_ctg;
    int64_t *voff, *v_slot;
    // ---- forward CSR + reversed CSR
    int32_t *deg, *e_col, *e_wr, *indeg, *rcur, *r_e;
    I4 *tmp_pk;                          // the packed in-edge records as kb_rev_fill's atomics placed them (r_pk: in list order)
    int64_t *rowptr, *e_wq, *rptr;
    uint8_t *e_fl;
    // ---- sweeps
    Dist *sp_d;
    int32_t *sp_best, *rev_order, *cnt_tmp, *fwd_order, *fwd_pos, *an, *anom_dest, *cnt_tmp2;
    // ---- CSR copy in forward-topological order (positions contiguous; heads as positions)
    int32_t *tp_deg, *tp_vj;
    int64_t *tp_ptr;
    I4 *te_pk;                           // [ET] an edge of the topologically ordered copy: {position of the head, qry weight (2 words), ref weight | flags << 24} (pack_in_edge's layout)
    // ---- SP-tree children CSR
    int32_t *ccnt, *cval;                // SP-tree children: count per vertex, ids at cval[rptr[u] ..] (kb_children)
    // ---- heaps
    int64_t *hoff;
    int32_t *hcap_cnt;                   // per contig: capacity request (for the scan)
    HNode *hnodes;
    int32_t *h_root, *bq, *h_cnt;
    // ---- enumeration
    Dist *kd;
    int32_t *klast, *kfound;
    I4 *kcand;                           // K8: two quads per pushed candidate, by insertion index: {heap node, predecessor, qry_score (2 words)}, {anom, qul_nonzero, qul_total, -}
    PqK *pq;                             // K8: queue storage, pq_stride entries per contig (heap form: 3K + 1; run form: enum_stride(K), aasm_enum.h)
    int64_t pq_stride;
    // ---- selection / outputs
    int32_t *pathA, *pathB, *pathT, *pre2, *stamp;    // (u,v) pairs, 2*(N+2) ints per contig
    Dist *dist2;
    uint8_t *notalt;
    int32_t *mark_time;                               // per sorted record: first conversion ordinal that marked it
    // ---- parallel conversions (one wave per converted path)
    int32_t *plan_kk;                    // [C * 2 * SEL_PLAN_KEEP] {walk, kind} of a contig's first conversions, from kb_sel_plan to kb_sel_planfill
    int32_t *nconv, *cv_ctg, *cv_k, *cv_ord, *cv_kind, *cv_szr, *cv_szv, *cv_n, *cv_err, *cv_path, *cv_pre2, *cv_stamp, *cv_la;
    int64_t *conv_off, *cv_roff, *cv_voff, *cv_cov;
    OutElem *cv_out;
    Dist *cv_dist2;
    int64_t NCONV;
    OutElem *cur_out, *main_out, *alt_out, *pool;
    int32_t *main_len, *alt_len, *all_gen, *all_seq;
    int32_t *ar_ctg, *ar_gen, *ar_seq, *ar_len;      // .all path records
    int64_t *ar_off;
    int64_t pool_cap, ar_cap;
    int64_t *main_off, *alt_off;                      // compaction offsets [C+1]
    OutElem *main_c, *alt_c;                          // compacted outputs
    // ---- status / counters
    int32_t *status;
    int64_t *counters;
    int32_t *gb_flag; int32_t gb_off;  // per contig: 1 / 2 = its rows, reversed CSR and sweep headers come from one workgroup (kb_graph_build, the small / the large form); gb_off: nobody's (tests, probes)
    int64_t *prof_heap, *prof_sel, *prof_gb;    // [C * 8] cycle sums per section (diagnostic build only)
    // ---- K7 pre-pass: compacted sidetrack keys and the packed per-vertex header
    Dist *st_cost;                       // per vertex, at the front of its CSR row: its sidetrack keys in list order, edge head in .pad
    int32_t *st_n;                       // ... and how many
    I4 *vhdr, *vhdr2, *cinfo;            // kb_heap_hdr
    // ---- K7, several waves per contig (kb_heap_mw)
    int32_t mw_mode;                     // 0: by graph density, 1: every contig, 2: none (tests)
    int32_t *mw_flag, *mw_lg;            // per contig: class, per-insert node bound
    HNode *hprov;                        // provisional arena: one region per vertex, regions in BFS order
    int64_t *mw_off;                     // per contig: start of its part of the provisional arena
    int32_t *mw_cap;                     // per contig: provisional capacity (0 for the one-wave class)
    int32_t *mw_order, *mw_rs, *mw_fb;   // per BFS position: vertex, region start, final base
    int32_t *mw_rsv, *mw_used;           // per vertex: region start, nodes used
    int32_t *mw_list, *mw_sorted;        // the contigs of the several-waves class: in any order; by node bound, largest first
    int32_t *mw_key;                     // the node bounds, in mw_list's order
    int32_t mw_n, mw_base;               // contigs of the class; kb_heap_mw by list (mw_base >= 0): block b builds mw_sorted[mw_base + b]
    I4 *tnx;                             // next four vertices along best[] (kb_sidetrack) ...
    int32_t *tnx16;                      // ... and the next sixteen (kb_heap_hdr): path recovery reads one 64-byte record per sixteen tree edges
    I4 *rvh;                             // K6: per-vertex in-list header, 3 words (see kb_rev_hdr)
    I4 *r_pk;                            // K6: one packed record per in-edge, in in-list order
    I4 *fvh;                             // K5: per-vertex out-list header, 2 words (see kb_rev_hdr)
    // ---- the chain class (kb_chain): contigs whose K6 sweep, K7 pre-pass and K7 heaps run BESIDE each other in one workgroup
    int32_t chain_mode;                  // 0: by batch shape, 1: every sparse one-wave contig, 2: none (tests, probes)
    int32_t chain_all, chain_minN;       // mode 0: every contig of the batch (small batches) / contigs of at least this many records (the long tail)
    int32_t chain_test;                  // test hooks (opts.reserved[2] bits 3, 4): 1 = the prep wave of contig 0 never publishes dest's header; 2 = ... and never says it is done (the heap wave's patience is 1 s then)
    int32_t *chain_flag, *chain_list;    // per contig: in the class; the contigs of the class, in any order
    int32_t *pend;                       // per vertex: in-neighbours whose keys are not written yet (the prep wave counts them down)
    int32_t *cq;                         // per contig slice: vertices whose header can be built, in the order they became ready
    I4 *bfsq;                            // per contig slice: the SP tree in BFS order, a record {vertex, position of its parent, start of its keys, #keys} per position (kb_chain's order wave -> heap wave)
    int32_t chain_ord;                   // kb_chain: 1 = the BFS order comes from a wave of its own (default), 0 = the heap wave keeps its own queue (probes, tests)
    int32_t chain_rn;                    // test hook (opts.reserved[2] bit 6): kb_heap_ord's ring of roots has this many entries (default 0: all 512), so that parents beyond it - frontiers wider than the ring - come up on small inputs
};

enum { CNT_RANGE_STEPS = 0, CNT_UNCONN, CNT_POOL, CNT_AR, CNT_CONVERTED, CNT_HEAPNODES, CNT_PATHS, CNT_OVF, CNT_ISPR_E, CNT_ISPR_V, CNT_PATH_E, CNT_OUT_E, CNT_PQ_PUSH, CNT_MW, CNT_MAXN, CNT_LONGSORT, CNT_MAXV, CNT_CHAIN, CNT_GB_S, CNT_GB_L, CNT_GB_REST, CNT_N };

AASM_DEV void set_status(const WS &w, int64_t c, int code) { if (w.status[c] == 0) w.status[c] = code; }
// the chain class (kb_chain): contigs whose sweep, K7 pre-pass and heaps run in one workgroup; the one-stage kernels skip them
AASM_DEV bool in_chain_class(const WS &w, int64_t c) { return w.chain_flag != nullptr && w.chain_flag[c] != 0; }
AASM_DEV bool in_graph_class(const WS &w, int64_t c) { return w.gb_flag && w.gb_flag[c] != 0; }   // (built by kb_graph_build: the separate launches leave the contig alone)
// Input range the narrowed fields are exact for (aasm_dev.h: Dist counters and HNode key counters are
// int32, e_wr is int32): coordinates in [0, 2^40), so that a path sum over < 2^20 edges of
// 2 * coordinate stays far inside int64 and every per-edge reference weight fits int32 after the
// 1e6 cap (paf_data.cpp:487-490).  A contig outside it gets status AASM_E_OVERFLOW (-5).
#define AASM_COORD_LIMIT ((int64_t)1 << 40)
AASM_DEV bool coord_ok(int64_t x) { return x >= 0 && x < AASM_COORD_LIMIT; }

// ====================================================================================
// K0  match ranges from short-form cs tags (get_overlap_range, paf_data.cpp:90-123, over the
//     tokenizer parse_short_cs, :29-72).  One THREAD per record.
// ====================================================================================
// Every lane scans its own tag, eight bytes per load, through the tokenizer's little state
// machine (operation character -> close the previous operation, open the next; otherwise one
// more digit / letter of the payload), so all 64 lanes do useful work on every instruction - a
// wave-cooperative parse of ONE tag (ballots over 64-byte windows) was measured 6x slower: at
// ~8 bytes per operation most lanes idle.  The ranges of a record go to its own slice of the
// three range arrays (offsets from the reader's count of ':' operations).  A '-' strand record
// is walked in text order from the query END (the reference walks the operations last to
// first, :75-86) and its ranges are written back to front, so no second pass is needed.
// Bit-exact with the host codec (tests); a malformed tag (any of the tokenizer's errors or the
// consumption check, :119-122) puts the record's index into cs_bad and the solve fails with
// AASM_E_PARSE.
AASM_DEV bool cs_is_op(int c) { return c == ':' || c == '*' || c == '+' || c == '-'; }
struct CsScan {
    int64_t q, rr, val, n, cnt, o0;
    int32_t plen;
    int t;                  // open operation (its character), 0 before the first
    bool fwd, bad;
};
AASM_DEV void cs_close_op(CsScan &s, const WS &w) {                  // the open operation is complete
    if (s.t == ':') {                                                // :42-49, then :101-107
        if (s.plen < 1 || s.val <= 0 || s.n >= s.cnt) { s.bad = true; return; }   // (a length beyond int64 has set val = -1: from_chars' out_of_range)
        if (s.fwd) {
            const int64_t o = s.o0 + s.n;
            I4 lo, hi; const int64_t qr = s.q + s.val - 1;
            lo.x = lo32((uint64_t)s.q); lo.y = hi32((uint64_t)s.q); lo.z = lo32((uint64_t)qr); lo.w = hi32((uint64_t)qr);
            hi.x = lo32((uint64_t)s.rr); hi.y = hi32((uint64_t)s.rr); hi.z = 0; hi.w = 0;
            I4 *rec = (I4 *)(w.rng_rec + 4 * o); rec[0] = lo; rec[1] = hi;
            s.q += s.val;
        } else {
            const int64_t o = s.o0 + s.cnt - 1 - s.n;
            I4 lo, hi; const int64_t ql = s.q - s.val, qr = s.q - 1, rl = s.rr + s.val - 1;
            lo.x = lo32((uint64_t)ql); lo.y = hi32((uint64_t)ql); lo.z = lo32((uint64_t)qr); lo.w = hi32((uint64_t)qr);
            hi.x = lo32((uint64_t)rl); hi.y = hi32((uint64_t)rl); hi.z = 0; hi.w = 0;
            I4 *rec = (I4 *)(w.rng_rec + 4 * o); rec[0] = lo; rec[1] = hi;
            s.q -= s.val;
        }
        s.rr += s.val;
        s.n++;
    } else if (s.t == '*') {                                         // :50-56, :112-116
        if (s.plen != 2) { s.bad = true; return; }
        s.q += s.fwd ? 1 : -1; s.rr += 1;
    } else {                                                         // :57-64
        if (s.plen < 1) { s.bad = true; return; }
        if (s.t == '+') s.q += s.fwd ? s.plen : -s.plen; else s.rr += s.plen;
    }
}
AASM_DEV void kb_cs_ranges(const KCtx &k, const WS &w) {
    const int64_t br = k.bid * k.nthreads + k.tid;
    if (br >= w.R) return;
    const int64_t r = w.R0 + br;
    const int64_t p0 = w.cs_off[r], len = w.cs_off[r + 1] - p0;
    const uint8_t *cs = (const uint8_t *)w.cs_text + p0;
    const int64_t qs = w.in_qs[r], qe = w.in_qe[r], rs = w.in_rs[r], re = w.in_re[r];
    CsScan s;
    s.fwd = w.in_fwd[r] != 0;
    s.o0 = w.in_rng_off[r]; s.cnt = w.in_rng_off[r + 1] - s.o0;
    s.q = s.fwd ? qs : qe + 1; s.rr = s.fwd ? rs : re;              // fwd: next query / ref base; rev: query end (exclusive) / lowest ref base not yet consumed
    s.val = 0; s.n = 0; s.plen = 0; s.t = 0;
    s.bad = len < 5 || cs[0] != 'c' || cs[1] != 's' || cs[2] != ':' || cs[3] != 'Z' || cs[4] != ':';   // :30-33
    for (int64_t pos = 5; pos < len && !s.bad; ) {
        uint64_t wd = 0;                                             // the next <= 8 bytes, first byte lowest
        int nb = (int)((len - pos < 8) ? (len - pos) : 8);
        const int mis = (int)((uintptr_t)(cs + pos) & 7);
        if (mis == 0 && nb == 8) wd = *(const uint64_t *)(cs + pos);
        else { if (nb > 8 - mis) nb = 8 - mis; for (int t = 0; t < nb; t++) wd |= (uint64_t)cs[pos + t] << (8 * t); }   // up to the next aligned word / the end
        pos += nb;
        for (int t = 0; t < nb && !s.bad; t++) {
            const int c = (int)(wd & 0xff);
            wd >>= 8;
            if (cs_is_op(c)) {
                if (s.t) cs_close_op(s, w);
                s.t = c; s.plen = 0; s.val = 0;
            } else if (s.t == ':') {
                const unsigned dg = (unsigned)(c - '0');
                if (dg > 9u) s.bad = true;
                else {                                               // std::from_chars<int64_t>: any number of digits, leading zeros included, value <= INT64_MAX
                    if ((uint64_t)s.val < 100000000ull) s.val = (int64_t)((uint32_t)s.val * 10u + dg);   // (up to nine digits: 32-bit arithmetic, no overflow to look for)
                    else {
                        const uint64_t nv = (uint64_t)s.val * 10u + dg;
                        s.val = (s.val < 0 || s.val > INT64_MAX / 10 || nv > (uint64_t)INT64_MAX) ? -1 : (int64_t)nv;
                    }
                    s.plen = 1;
                }
            } else if (s.t && (unsigned)((c | 32) - 'a') < 26u) s.plen++;
            else s.bad = true;                                       // not a cs character, or payload before any operation (:66-68)
        }
    }
    if (!s.bad && s.t) cs_close_op(s, w);
    if (!s.bad) s.bad = (s.n != s.cnt) || (s.fwd ? (s.q != qe + 1 || s.rr != re + 1) : (s.q != qs || s.rr != rs + 1));   // :119-122
    if (s.bad) atomic_min_i32(w.cs_bad, (int32_t)(br - INT32_MAX));
}

// ====================================================================================
// K1  sort (paf_data.cpp:241; comparator paf_data.hpp:69-73)
// ====================================================================================
// One 256-thread block per contig.  The order wanted is the STABLE one - (qry_str, qry_end),
// ties by input position - which is the plain order of the unique key (qry_str, qry_end,
// index), so any comparison network gives it: chunks of 1024 records are sorted by a bitonic
// network in LDS (55 stages, 2 compare-exchanges per thread and stage); a contig of one chunk
// is done then, a longer one writes its sorted chunks to scratch and every record adds up its
// lower bounds in the other chunks (binary search) to get its rank.  std::sort is NOT stable,
// so when a contig holds duplicate (qry_str, qry_end) keys and N > 16 (libstdc++ switches from
// pure insertion sort to introsort there) kb_sort_fix replays libstdc++'s algorithm exactly
// (hazard B1).
#define SORT_CHUNK 1024
#define AASM_SORT_LDS_BYTES (SORT_CHUNK * 20)
struct SortLds { int64_t qs[SORT_CHUNK], qe[SORT_CHUNK]; int32_t idx[SORT_CHUNK]; };
static_assert(sizeof(SortLds) <= AASM_SORT_LDS_BYTES, "LDS budget");
AASM_DEV bool sortkey_lt(int64_t a0, int64_t a1, int32_t a2, int64_t b0, int64_t b1, int32_t b2) {
    if (a0 != b0) return a0 < b0;
    if (a1 != b1) return a1 < b1;
    return a2 < b2;
}
AASM_DEV void kb_sort(const KCtx &k, const WS &w) {
    const int64_t c = k.bid;
    const int64_t gb = w.rec_off[c], N = w.rec_off[c + 1] - gb, b = gb - w.R0;
    if (N <= 0) return;
    SortLds *L = (SortLds *)k.lds;
    const int64_t *qs = w.in_qs + gb, *qe = w.in_qe + gb;
    {   // A contig whose records already come in (qry_str, qry_end) order - a file written in query order - needs no network: the stable
        // order is the input order.  One pass over neighbouring pairs decides (and finds equal keys for kb_sort_fix on the way).
        int32_t *flag = (int32_t *)k.lds;                            // [0] out of order somewhere, [1] equal neighbours
        if (k.tid < 2) flag[k.tid] = 0;
        block_barrier();
        int unsorted = 0, dup = 0;
        for (int64_t r = k.tid; r + 1 < N; r += k.nthreads) {
            const int64_t a0 = qs[r], a1 = qe[r], b0 = qs[r + 1], b1 = qe[r + 1];
            unsorted |= (b0 < a0 || (b0 == a0 && b1 < a1)) ? 1 : 0;
            dup |= (b0 == a0 && b1 == a1) ? 1 : 0;
        }
        if (unsorted) flag[0] = 1;
        if (dup) flag[1] = 1;
        block_barrier();
        const bool in_order = flag[0] == 0, has_dup = flag[1] != 0;
        block_barrier();                                             // (the LDS words are the network's from here on)
        if (in_order) {
            for (int64_t r = k.tid; r < N; r += k.nthreads) w.perm[b + r] = (int32_t)r;
            if (k.tid == 0) w.dupflag[c] = 2 | ((has_dup && N > 16) ? 1 : 0);   // bit 1: the order is final (kb_sort_rank leaves the contig alone)
            return;
        }
    }
    const int64_t nch = (N + SORT_CHUNK - 1) / SORT_CHUNK;
    // scratch for the sorted chunks of a long contig: the sorted-record arrays K1's gather fills later
    int64_t *t_qs = w.s_qs + b, *t_qe = w.s_qe + b;
    int32_t *t_ix = w.s_orig + b;
    for (int64_t ch = 0; ch < nch; ch++) {
        const int64_t base = ch * SORT_CHUNK;
        const int32_t n = (int32_t)((N - base < SORT_CHUNK) ? (N - base) : SORT_CHUNK);
        int32_t P = 2;                                               // network size: next power of two >= n
        while (P < n) P <<= 1;
        for (int32_t t = k.tid; t < P; t += k.nthreads) {
            if (t < n) { L->qs[t] = qs[base + t]; L->qe[t] = qe[base + t]; L->idx[t] = (int32_t)(base + t); }
            else { L->qs[t] = INT64_MAX; L->qe[t] = INT64_MAX; L->idx[t] = INT32_MAX; }       // padding sorts last
        }
        block_barrier();
        for (int32_t kk = 2; kk <= P; kk <<= 1)
            for (int32_t j = kk >> 1; j > 0; j >>= 1) {
                for (int32_t t = k.tid; t < (P >> 1); t += k.nthreads) {
                    const int32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                    const bool up = (lo & kk) == 0;
                    const int64_t a0 = L->qs[lo], a1 = L->qe[lo], b0 = L->qs[hi], b1 = L->qe[hi];
                    const int32_t a2 = L->idx[lo], b2 = L->idx[hi];
                    if (sortkey_lt(b0, b1, b2, a0, a1, a2) == up) {
                        L->qs[lo] = b0; L->qe[lo] = b1; L->idx[lo] = b2;
                        L->qs[hi] = a0; L->qe[hi] = a1; L->idx[hi] = a2;
                    }
                }
                block_barrier();
            }
        if (nch == 1) { for (int32_t t = k.tid; t < n; t += k.nthreads) w.perm[b + t] = L->idx[t]; }
        else for (int32_t t = k.tid; t < n; t += k.nthreads) { t_qs[base + t] = L->qs[t]; t_qe[base + t] = L->qe[t]; t_ix[base + t] = L->idx[t]; }
        block_barrier();
    }
    if (nch > 1) {                                                   // the sorted chunks stay in scratch: kb_sort_rank merges them, a thread per record
        if (k.tid == 0) atomic_add(&w.counters[CNT_LONGSORT], (int64_t)1);
        return;
    }
    if (N > 16) {                                                    // duplicate keys are neighbours now
        int dup = 0;
        for (int64_t r = k.tid; r + 1 < N; r += k.nthreads) {
            const int32_t x = w.perm[b + r], y = w.perm[b + r + 1];
            dup |= (qs[x] == qs[y] && qe[x] == qe[y]) ? 1 : 0;
        }
        if (dup) w.dupflag[c] = 1;
    }
}
// contigs longer than one chunk: rank of a record = place in its own sorted chunk + its lower bounds in the other chunks of
// its contig (binary searches).  One THREAD PER RECORD of the batch (a block per contig spent 64 of the 71 ms sort of a
// 100 000-record contig here); a batch without such a contig leaves after one load.  A record with an equal (qry_str, qry_end)
// key anywhere in its contig shows up next to one of the lower bounds: dupflag as in kb_sort.
AASM_DEV void kb_sort_rank(const KCtx &k, const WS &w) {
    if (w.counters[CNT_LONGSORT] == 0) return;
    const int64_t g = k.bid * k.nthreads + k.tid;                    // record, relative to R0
    if (g >= w.R) return;
    int64_t lo_c = 0, hi_c = w.C;                                    // contig of the record: last c with rec_off[c] - R0 <= g
    while (hi_c - lo_c > 1) { const int64_t m = (lo_c + hi_c) >> 1; if (w.rec_off[m] - w.R0 <= g) lo_c = m; else hi_c = m; }
    const int64_t c = lo_c, gb = w.rec_off[c], N = w.rec_off[c + 1] - gb, b = gb - w.R0;
    if (N <= SORT_CHUNK || (w.dupflag[c] & 2)) return;
    const int64_t *t_qs = w.s_qs + b, *t_qe = w.s_qe + b;
    const int32_t *t_ix = w.s_orig + b;
    const int64_t i = g - b, nch = (N + SORT_CHUNK - 1) / SORT_CHUNK;
    const int64_t a0 = t_qs[i], a1 = t_qe[i];
    const int32_t a2 = t_ix[i];
    const int64_t own = i / SORT_CHUNK;
    int64_t rank = i - own * SORT_CHUNK;
    bool dup = (i > own * SORT_CHUNK && t_qs[i - 1] == a0 && t_qe[i - 1] == a1);
    for (int64_t ch = 0; ch < nch; ch++) {
        if (ch == own) continue;
        const int64_t base = ch * SORT_CHUNK, len = (N - base < SORT_CHUNK) ? (N - base) : SORT_CHUNK;
        int64_t lo = 0, hi = len;
        while (lo < hi) {
            const int64_t m = (lo + hi) >> 1;
            if (sortkey_lt(t_qs[base + m], t_qe[base + m], t_ix[base + m], a0, a1, a2)) lo = m + 1; else hi = m;
        }
        rank += lo;
        dup |= (lo < len && t_qs[base + lo] == a0 && t_qe[base + lo] == a1) || (lo > 0 && t_qs[base + lo - 1] == a0 && t_qe[base + lo - 1] == a1);
    }
    w.perm[b + rank] = a2;
    if (dup) w.dupflag[c] = 1;
}

// ---- libstdc++ (GCC 11) std::sort replayed (hazard B1) -------------------------------
// The algorithm (bits/stl_algo.h: __introsort_loop, __unguarded_partition_pivot, __partial_sort as the
// depth-limit fallback, __final_insertion_sort) written against an accessor, so that the same statements
// run on an index array in global memory (contigs too long for LDS) and on (key, index) tuples in LDS.
struct SortGlob {                        // elements are record indices; keys are looked up
    int32_t *a; const int64_t *qs, *qe;
    typedef int32_t E;
    AASM_MEM E get(int64_t i) const { return a[i]; }
    AASM_MEM void set(int64_t i, E v) const { a[i] = v; }
    AASM_MEM bool lt(E x, E y) const { if (qs[x] != qs[y]) return qs[x] < qs[y]; return qe[x] < qe[y]; }
    // first position >= lo whose element is not < pivot / last position <= hi whose element is not > pivot (a sentinel always exists)
    AASM_MEM int64_t scan_up(int64_t lo, E pivot) const { while (lt(a[lo], pivot)) ++lo; return lo; }
    AASM_MEM int64_t scan_down(int64_t hi, E pivot) const { while (lt(pivot, a[hi])) --hi; return hi; }
};
struct SortTuple { int64_t qs, qe; int32_t ix; };
AASM_DEV void keep_tuple(SortTuple &e) { keep_load(e.qs); keep_load(e.qe); keep_load(e.ix); }
struct SortGlobT {                       // (qry_str, qry_end, record index) tuples in global scratch, moved as a whole; `tab` = the contig's perm slice
    int64_t *qs, *qe; int32_t *ix; int32_t *tab;
    typedef SortTuple E;
    AASM_MEM E get(int64_t i) const { E e; e.qs = qs[i]; e.qe = qe[i]; e.ix = ix[i]; return e; }
    AASM_MEM E getk(int64_t i) const { E e; e.qs = qs[i]; e.qe = qe[i]; e.ix = 0; return e; }   // the keys alone
    AASM_MEM void set(int64_t i, const E &v) const { qs[i] = v.qs; qe[i] = v.qe; ix[i] = v.ix; }
    AASM_MEM bool lt(const E &x, const E &y) const { if (x.qs != y.qs) return x.qs < y.qs; return x.qe < y.qe; }
    AASM_MEM void tput(int64_t k, int64_t pos) const { tab[k] = (int32_t)pos; }
    AASM_MEM int64_t tget(int64_t k) const { return tab[k]; }
    AASM_MEM void sync() const { wave_fence(); }
    AASM_MEM int64_t scan_up(int64_t lo, const E &pivot) const { while (qs[lo] < pivot.qs || (qs[lo] == pivot.qs && qe[lo] < pivot.qe)) ++lo; return lo; }
    AASM_MEM int64_t scan_down(int64_t hi, const E &pivot) const { while (pivot.qs < qs[hi] || (pivot.qs == qs[hi] && pivot.qe < qe[hi])) --hi; return hi; }
};
#define SF_MAX 1536                      // records of a contig (or of a sub-range of a longer one) that fit the LDS form (38 KB: four per CU at a time)
#define SF_QN 256
#define SF_ST 72                         // depth of the two range stacks (introsort's depth limit is 2 lg N <= 62)
#ifndef SF_WMIN
#define SF_WMIN 96                       // ranges of this many elements and more are partitioned by the whole wave,
#endif
#ifndef SF_GW
#define SF_GW 8
#endif
#define SF_G (AASM_WAVE >= 64 ? SF_GW : 1) // shorter ones by this many lanes,
#define SF_NG (AASM_WAVE / SF_G)          // that many ranges at a time
#define SF_PAD 4                         // slack in front of and behind the elements (the partition scans read 4 at a time)
struct SfLds {
    int64_t qs[SF_MAX + 2 * SF_PAD], qe[SF_MAX + 2 * SF_PAD]; int32_t ix[SF_MAX + 2 * SF_PAD];
    int32_t qf[SF_QN], ql[SF_QN], qd[SF_QN];                       // work list of the lane-per-range phase
    int32_t af[SF_ST], al[SF_ST], ad[SF_ST];                       // stack of the wave-per-range phase
    int32_t gf[SF_ST], gl[SF_ST], gd[SF_ST];                       // stack of the global-memory form
    int16_t tab[SF_MAX + 2 * SF_PAD];                              // right stops of the partition in hand
    int32_t qn, ovf;
};
#define AASM_SORTFIX_LDS_BYTES ((SF_MAX + 2 * SF_PAD) * 22 + SF_QN * 12 + SF_ST * 24 + 16)
static_assert(sizeof(SfLds) <= AASM_SORTFIX_LDS_BYTES, "LDS budget");
struct SortLdsAcc {                      // elements are (qry_str, qry_end, record index) tuples, moved as a whole
    SfLds *L;
    typedef SortTuple E;
    AASM_MEM E get(int64_t i) const { E e; e.qs = L->qs[i]; e.qe = L->qe[i]; e.ix = L->ix[i]; return e; }   // (callers pass positions shifted by SF_PAD)
    AASM_MEM E getk(int64_t i) const { E e; e.qs = L->qs[i]; e.qe = L->qe[i]; e.ix = 0; return e; }   // the keys alone
    AASM_MEM void set(int64_t i, const E &v) const { L->qs[i] = v.qs; L->qe[i] = v.qe; L->ix[i] = v.ix; }
    AASM_MEM bool lt(const E &x, const E &y) const { if (x.qs != y.qs) return x.qs < y.qs; return x.qe < y.qe; }
    AASM_MEM void tput(int64_t k, int64_t pos) const { L->tab[k] = (int16_t)pos; }
    AASM_MEM int64_t tget(int64_t k) const { return L->tab[k]; }
    AASM_MEM void sync() const { wave_lds_sync(); }
    // the partition's two scans, four elements per LDS round trip (reading past the stop is harmless: the arrays
    // carry 4 elements of slack at both ends of the LDS block)
    AASM_MEM int64_t scan_up(int64_t lo, const E &pivot) const {
        while (true) {
            const int64_t q0 = L->qs[lo], q1 = L->qs[lo + 1], q2 = L->qs[lo + 2], q3 = L->qs[lo + 3];
            if (!(q0 < pivot.qs || (q0 == pivot.qs && L->qe[lo] < pivot.qe))) return lo;
            if (!(q1 < pivot.qs || (q1 == pivot.qs && L->qe[lo + 1] < pivot.qe))) return lo + 1;
            if (!(q2 < pivot.qs || (q2 == pivot.qs && L->qe[lo + 2] < pivot.qe))) return lo + 2;
            if (!(q3 < pivot.qs || (q3 == pivot.qs && L->qe[lo + 3] < pivot.qe))) return lo + 3;
            lo += 4;
        }
    }
    AASM_MEM int64_t scan_down(int64_t hi, const E &pivot) const {
        while (true) {
            const int64_t q0 = L->qs[hi], q1 = L->qs[hi - 1], q2 = L->qs[hi - 2], q3 = L->qs[hi - 3];
            if (!(pivot.qs < q0 || (pivot.qs == q0 && pivot.qe < L->qe[hi]))) return hi;
            if (!(pivot.qs < q1 || (pivot.qs == q1 && pivot.qe < L->qe[hi - 1]))) return hi - 1;
            if (!(pivot.qs < q2 || (pivot.qs == q2 && pivot.qe < L->qe[hi - 2]))) return hi - 2;
            if (!(pivot.qs < q3 || (pivot.qs == q3 && pivot.qe < L->qe[hi - 3]))) return hi - 3;
            hi -= 4;
        }
    }
};
template <class A> AASM_DEV void ss_swap(const A &a, int64_t i, int64_t j) { const typename A::E t = a.get(i); a.set(i, a.get(j)); a.set(j, t); }
template <class A> AASM_DEV void ss_push_heap(const A &a, int64_t base, int64_t hole, int64_t top, typename A::E val) {
    int64_t parent = (hole - 1) / 2;
    while (hole > top && a.lt(a.get(base + parent), val)) { a.set(base + hole, a.get(base + parent)); hole = parent; parent = (hole - 1) / 2; }
    a.set(base + hole, val);
}
template <class A> AASM_DEV void ss_adjust_heap(const A &a, int64_t base, int64_t hole, int64_t len, typename A::E val) {
    const int64_t top = hole;
    int64_t child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (a.lt(a.get(base + child), a.get(base + child - 1))) child--;
        a.set(base + hole, a.get(base + child));
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        a.set(base + hole, a.get(base + child - 1));
        hole = child - 1;
    }
    ss_push_heap(a, base, hole, top, val);
}
template <class A> AASM_DEV void ss_heapsort(const A &a, int64_t base, int64_t len) {   // __partial_sort(f, l, l)
    if (len >= 2) {
        int64_t parent = (len - 2) / 2;
        while (true) {
            ss_adjust_heap(a, base, parent, len, a.get(base + parent));
            if (parent == 0) break;
            parent--;
        }
    }
    int64_t last = len;
    while (last > 1) {
        --last;
        const typename A::E v = a.get(base + last);
        a.set(base + last, a.get(base));
        ss_adjust_heap(a, base, 0, last, v);
    }
}
template <class A> AASM_DEV void ss_unguarded_linear_insert(const A &a, int64_t last) {
    const typename A::E val = a.get(last);
    int64_t next = last - 1;
    while (a.lt(val, a.get(next))) { a.set(last, a.get(next)); last = next; --next; }
    a.set(last, val);
}
template <class A> AASM_DEV void ss_insertion_sort(const A &a, int64_t first, int64_t last) {
    if (first == last) return;
    for (int64_t i = first + 1; i != last; ++i) {
        if (a.lt(a.get(i), a.get(first))) {
            const typename A::E val = a.get(i);
            for (int64_t t = i; t > first; --t) a.set(t, a.get(t - 1));
            a.set(first, val);
        } else ss_unguarded_linear_insert(a, i);
    }
}
// __unguarded_partition_pivot(first, last): median of three to `first`, Hoare partition around it; returns the cut
template <class A> AASM_DEV int64_t ss_partition_pivot(const A &a, int64_t first, int64_t last) {
    const int64_t mid = first + (last - first) / 2;
    {   // __move_median_to_first(first, first+1, mid, last-1)
        const int64_t pa = first + 1, pb = mid, pc = last - 1;
        const typename A::E ea = a.get(pa), eb = a.get(pb), ec = a.get(pc);
        if (a.lt(ea, eb)) {
            if (a.lt(eb, ec)) ss_swap(a, first, pb);
            else if (a.lt(ea, ec)) ss_swap(a, first, pc);
            else ss_swap(a, first, pa);
        } else if (a.lt(ea, ec)) ss_swap(a, first, pa);
        else if (a.lt(eb, ec)) ss_swap(a, first, pc);
        else ss_swap(a, first, pb);
    }
    const typename A::E pivot = a.get(first);                        // (it stays at `first` during the partition)
    int64_t lo = first + 1, hi = last;
    while (true) {                                                   // __unguarded_partition(first+1, last, first)
        lo = a.scan_up(lo, pivot);
        --hi;
        hi = a.scan_down(hi, pivot);
        if (!(lo < hi)) break;
        ss_swap(a, lo, hi);
        ++lo;
    }
    return lo;
}
// (st_first / st_last / st_depth: 72 entries each, the caller's - the kernel lends LDS it no longer needs, so that this cold
// path costs the kernel no scratch memory)
template <class A> AASM_DEV void ss_std_sort(const A &a, int64_t n, int depth_override, int64_t *st_first, int64_t *st_last, int32_t *st_depth) {
    if (n <= 0) return;
    int lg = 0;
    for (int64_t t = n; t > 1; t >>= 1) lg++;
    int sp = 0;
    st_first[0] = 0; st_last[0] = n; st_depth[0] = depth_override >= 0 ? depth_override : 2 * lg;
    sp = 1;
    while (sp > 0) {
        --sp;
        int64_t first = st_first[sp], last = st_last[sp];
        int depth = st_depth[sp];
        while (last - first > 16) {
            if (depth == 0) { ss_heapsort(a, first, last - first); break; }
            --depth;
            const int64_t cut = ss_partition_pivot(a, first, last);
            if (sp < 72) { st_first[sp] = cut; st_last[sp] = last; st_depth[sp] = depth; sp++; }
            last = cut;
        }
    }
    // __final_insertion_sort
    if (n > 16) {
        ss_insertion_sort(a, 0, 16);
        for (int64_t i = 16; i != n; ++i) ss_unguarded_linear_insert(a, i);
    } else ss_insertion_sort(a, 0, n);
}

// __unguarded_partition_pivot(first, last) by the whole wave, with the result of the sequential statement.  In the
// array as it stands after the median step, call "left stops" A_0 < A_1 < ... the positions whose element is not < pivot
// and "right stops" B_0 > B_1 > ... those whose element is not > pivot.  The sequential loop swaps A_k with B_k for as
// long as A_k < B_k (a swap only touches positions outside the stretch the scans have yet to cross, so the k-th stops it
// finds are these), and returns min(A_m, B_(m-1)) for the first k = m that fails.  A_k < B_k holds for a prefix of k, the
// pairs are disjoint, and every B_k of the prefix lies right of every A_k of it - so: the right stops are listed from the
// right end in `tab` (ballot + prefix count per 64 positions, as far ahead as the left side needs them), the left side is
// walked 64 positions at a time, every lane ranks its left stop and swaps it with its partner if that lies to the right;
// the first lane that finds its partner on the wrong side gives the cut.  A position read late (after a swap put another
// element there) is a B_k of the prefix: right of every swap that can still happen, so its stale rank never matters.
// G = lanes that work on the range together: the whole wave for a long range, 8 for a short one (eight ranges at a time).
#define SS_U 4                           // chunks of G positions in flight
template <int G> AASM_DEV uint64_t group_ballot(bool p, int gbase) {
    const uint64_t m = wave_ballot(p);
    return G >= AASM_WAVE ? m : ((m >> gbase) & ((1ull << (G & 63)) - 1ull));
}
template <int G, class A> AASM_DEV int64_t ss_partition_group(const A &a, int64_t first, int64_t last, int lane) {
    typedef typename A::E E;
    const int gl = lane & (G - 1), gbase = lane - gl;
    E pivot;
    {   // __move_median_to_first(first, first+1, mid, last-1): every lane picks, the group's first lane swaps
        const int64_t pa = first + 1, pb = first + (last - first) / 2, pc = last - 1;
        E ea = a.get(pa), eb = a.get(pb), ec = a.get(pc), ef = a.get(first);
        keep_tuple(ea); keep_tuple(eb); keep_tuple(ec); keep_tuple(ef);
        int64_t pick;
        if (a.lt(ea, eb)) pick = a.lt(eb, ec) ? pb : (a.lt(ea, ec) ? pc : pa);
        else pick = a.lt(ea, ec) ? pa : (a.lt(eb, ec) ? pc : pb);
        pivot = pick == pa ? ea : (pick == pb ? eb : ec);            // (it stays at `first` during the partition)
        a.sync();
        if (gl == 0) { a.set(first, pivot); a.set(pick, ef); }
        a.sync();
    }
    const int64_t f = first + 1, l = last;
    int64_t nb = 0, top = l;                                         // right stops listed, positions >= top looked at
    int64_t na = 0, P = f, lim = l;                                  // swaps done, next left position, B_(na-1)
    int64_t cut = -1;
    // (statements below are selects rather than branches wherever the groups of a wave could part ways)
    while (cut < 0) {
        bool wrote = false;
        while (nb < na + SS_U * G && top > f) {                      // tab[first + k] = B_k
            E e[SS_U];
            AASM_UNROLL
            for (int u = 0; u < SS_U; u++) { const int64_t i = top - 1 - gl - u * G; e[u] = pivot; if (i >= f) e[u] = a.getk(i); }
            AASM_UNROLL
            for (int u = 0; u < SS_U; u++) keep_tuple(e[u]);             // (all loads of the round in flight together)
            AASM_UNROLL
            for (int u = 0; u < SS_U; u++) {
                if (G >= AASM_WAVE && top - u * G <= f) break;
                const int64_t i = top - 1 - gl - u * G;
                const bool rb = i >= f && !a.lt(pivot, e[u]);
                const uint64_t m = group_ballot<G>(rb, gbase);
                if (rb) a.tput(first + nb + popc64(m & lanemask_lt(gl)), i);
                nb += popc64(m);
            }
            top -= SS_U * G;
            wrote = true;
        }
        if (wrote) a.sync();
        E e[SS_U];
        AASM_UNROLL
        for (int u = 0; u < SS_U; u++) { const int64_t i = P + u * G + gl; e[u] = pivot; if (i < l) e[u] = a.get(i); }
        AASM_UNROLL
        for (int u = 0; u < SS_U; u++) keep_tuple(e[u]);
        AASM_UNROLL
        for (int u = 0; u < SS_U; u++) {
            if (G >= AASM_WAVE && cut >= 0) break;
            const int64_t i = P + u * G + gl;
            cut = (cut < 0 && P + u * G >= lim) ? lim : cut;
            const bool la = cut < 0 && i < l && !a.lt(e[u], pivot);
            const uint64_t m = group_ballot<G>(la, gbase);
            const int64_t kk = na + popc64(m & lanemask_lt(gl));
            int64_t j = -1;
            if (la && kk < nb) j = a.tget(first + kk);
            const bool sw = la && j > i;
            const uint64_t ms = group_ballot<G>(sw, gbase), mf = group_ballot<G>(la && !sw, gbase);
            if (sw) { E o = a.get(j); keep_tuple(o); a.set(i, o); a.set(j, e[u]); }
            const int top_lane = ms ? 63 - __builtin_clzll(ms) : 0;
            const int64_t jl = G >= AASM_WAVE ? wave_bcast(j, top_lane) : wave_shfl_idx(j, gbase + top_lane);
            lim = ms ? jl : lim;
            na += popc64(ms);
            const int64_t ifail = P + u * G + (ffs64(mf) - 1);
            cut = mf ? (ifail < lim ? ifail : lim) : cut;
        }
        P += SS_U * G;
    }
    a.sync();
    return cut;
}
template <class A> AASM_DEV int64_t ss_partition_wave(const A &a, int64_t first, int64_t last, int lane) { return ss_partition_group<AASM_WAVE>(a, first, last, lane); }

// The introsort of n tuples at L->..[SF_PAD, SF_PAD + n) with `depth` partition levels left, by one wave; the result goes
// to out[0..n) (record indices).  The partition phase of the introsort is a tree of independent sub-ranges - a range's
// partition only looks at its own elements:
//   A  ranges of SF_WMIN and more elements: one after the other, by the whole wave (ss_partition_wave);
//   B  shorter ranges: eight at a time from a work list in LDS, 8 lanes each (the same statement), halves of more than
//      16 elements go back on the list;
//   C  libstdc++'s final insertion sort: an insertion sort is a stable sort, the leaf ranges (<= 16 elements, or
//      heap-sorted at the depth limit) are ordered among themselves, and an element never moves past an equal one -
//      so an element's final place is its position minus the greater elements among the 15 in front of it plus the
//      smaller ones among the 15 behind it: counted per element, lane-parallel, no element moves.
// false = work list overflow (cannot happen: pending ranges are longer than 16 elements, so never more than n / 17).
#if defined(AASM_KPROF)
#define SF_KP_ARG , int64_t *kpa
#define SF_KP_PASS , kp_acc
#define SF_KP(i) do { const int64_t t1 = (int64_t)__builtin_amdgcn_s_memtime(); kpa[i] += t1 - kpt; kpt = t1; } while (0)
#else
#define SF_KP_ARG
#define SF_KP_PASS
#define SF_KP(i) do {} while (0)
#endif
AASM_DEV bool sf_sort_lds(SfLds *L, int32_t n, int32_t depth0, int32_t *out, int lane SF_KP_ARG) {
#if defined(AASM_KPROF)
    int64_t kpt = (int64_t)__builtin_amdgcn_s_memtime();
#endif
    SortLdsAcc acc{L};
    if (lane == 0) { L->af[0] = SF_PAD; L->al[0] = SF_PAD + n; L->ad[0] = depth0; L->qn = 0; L->ovf = 0; }
    wave_lds_sync();
    int32_t asp = n > 16 ? 1 : 0;
    while (asp > 0) {                                                // A
        --asp;
        const int32_t first = uni(L->af[asp]), last = uni(L->al[asp]), depth = uni(L->ad[asp]);
        wave_lds_sync();
        if (last - first < SF_WMIN || depth == 0) {                  // for the lanes (B)
            if (lane == 0) { const int32_t at = L->qn; if (at < SF_QN) { L->qf[at] = first; L->ql[at] = last; L->qd[at] = depth; L->qn = at + 1; } else L->ovf = 1; }
            wave_lds_sync();
            continue;
        }
        const int32_t cut = (int32_t)ss_partition_wave(acc, first, last, lane);
        const int32_t f2[2] = {cut, first}, l2[2] = {last, cut};     // (the left half is taken up first, as the sequential loop does)
        for (int t = 0; t < 2; t++)
            if (l2[t] - f2[t] > 16) {
                if (asp < SF_ST) { if (lane == 0) { L->af[asp] = f2[t]; L->al[asp] = l2[t]; L->ad[asp] = depth - 1; } asp++; }
                else if (lane == 0) L->ovf = 1;
            }
        wave_lds_sync();
    }
    SF_KP(1);
    while (!uni(L->ovf)) {                                           // B
        const int32_t qn = uni(L->qn);
        if (qn <= 0) break;
        const int32_t take = qn < SF_NG ? qn : SF_NG;               // the last min(qn, #groups) entries of the list, one per group of lanes
        const int grp = lane / SF_G, gl = lane & (SF_G - 1);
        int32_t first = 0, last = 0, depth = 0;
        const bool mine = grp < take;
        if (mine) { const int32_t e = qn - 1 - grp; first = L->qf[e]; last = L->ql[e]; depth = L->qd[e]; }
        wave_lds_sync();
        if (lane == 0) L->qn = qn - take;
        wave_lds_sync();
        if (mine) {
            if (depth == 0) { if (gl == 0) ss_heapsort(acc, first, last - first); }   // depth limit: __partial_sort(first, last, last); sorted for good
            else {
                const int32_t cut = (int32_t)ss_partition_group<SF_G>(acc, first, last, lane);
                if (gl == 0) {
                    const int32_t f2[2] = {first, cut}, l2[2] = {cut, last};
                    for (int t = 0; t < 2; t++)
                        if (l2[t] - f2[t] > 16) {
                            const int32_t at = (int32_t)atomic_add(&L->qn, (int32_t)1);
                            if (at < SF_QN) { L->qf[at] = f2[t]; L->ql[at] = l2[t]; L->qd[at] = depth - 1; }
                            else L->ovf = 1;
                        }
                }
            }
        }
        wave_lds_sync();
    }
    SF_KP(2);
    if (uni(L->ovf)) return false;
    for (int32_t i = lane; i < n; i += AASM_WAVE) {                  // C
        const int32_t p = SF_PAD + i;
        const int64_t k0 = L->qs[p], k1 = L->qe[p];
        int32_t rank = i;
        const int32_t dn = i < 15 ? i : 15, up = (n - 1 - i) < 15 ? (n - 1 - i) : 15;
        int64_t q0[15], q1[15];
        AASM_UNROLL
        for (int32_t d = 1; d <= 15; d++) { const int32_t j = d <= dn ? p - d : p; q0[d - 1] = L->qs[j]; q1[d - 1] = L->qe[j]; }   // (beyond the ends: the element itself, which counts for nothing)
        AASM_UNROLL
        for (int32_t d = 0; d < 15; d++) { keep_load(q0[d]); keep_load(q1[d]); }
        AASM_UNROLL
        for (int32_t d = 0; d < 15; d++) rank -= (q0[d] > k0 || (q0[d] == k0 && q1[d] > k1)) ? 1 : 0;
        AASM_UNROLL
        for (int32_t d = 1; d <= 15; d++) { const int32_t j = d <= up ? p + d : p; q0[d - 1] = L->qs[j]; q1[d - 1] = L->qe[j]; }
        AASM_UNROLL
        for (int32_t d = 0; d < 15; d++) { keep_load(q0[d]); keep_load(q1[d]); }
        AASM_UNROLL
        for (int32_t d = 0; d < 15; d++) rank += (q0[d] < k0 || (q0[d] == k0 && q1[d] < k1)) ? 1 : 0;
        out[rank] = L->ix[p];
    }
    SF_KP(3);
    return true;
}

// One wave per contig with duplicate keys.  N <= SF_MAX: the contig's (key, index) tuples go to LDS in input order and
// sf_sort_lds does the rest.  A longer contig keeps its tuples in global scratch (the sorted-record arrays K1's gather
// fills later - kb_sort_rank is done with them); the wave partitions there (ss_partition_wave, the right stops listed in
// the range's own slice of perm) until a range fits LDS, which then finishes it.
AASM_DEV void kb_sort_fix(const KCtx &k, const WS &w) {
    const int64_t c = k.bid;
    if (c >= w.C || !(w.dupflag[c] & 1)) return;
    const int64_t gb = w.rec_off[c], N = w.rec_off[c + 1] - gb, b = gb - w.R0;
    int32_t *perm = w.perm + b;
    SfLds *L = (SfLds *)k.lds;
    int lg = 0;
    for (int64_t t = N; t > 1; t >>= 1) lg++;
    const int32_t depth0 = w.sort_depth_test > 0 ? w.sort_depth_test - 1 : 2 * lg;
    bool ok = true;
    KPROF_DECL;
    KPROF_START();
#if defined(AASM_KPROF)
    const int64_t kp_real0 = wave_realtime();
#endif
    if (N <= SF_MAX) {
        for (int64_t i = k.lane; i < N; i += AASM_WAVE) { L->qs[SF_PAD + i] = w.in_qs[gb + i]; L->qe[SF_PAD + i] = w.in_qe[gb + i]; L->ix[SF_PAD + i] = (int32_t)i; }   // input order (:232)
        KPROF_STAMP(0);
        ok = sf_sort_lds(L, (int32_t)N, depth0, perm, k.lane SF_KP_PASS);
        KPROF_START();
    } else {
        SortGlobT g{w.s_qs + b, w.s_qe + b, w.s_orig + b, perm};
        for (int64_t i = k.lane; i < N; i += AASM_WAVE) { g.qs[i] = w.in_qs[gb + i]; g.qe[i] = w.in_qe[gb + i]; g.ix[i] = (int32_t)i; }
        if (k.lane == 0) { L->gf[0] = 0; L->gl[0] = (int32_t)N; L->gd[0] = depth0; }
        wave_fence();
        wave_lds_sync();
        int32_t gsp = 1;
        while (gsp > 0 && ok) {
            --gsp;
            const int32_t first = uni(L->gf[gsp]);
            int32_t last = uni(L->gl[gsp]), depth = uni(L->gd[gsp]);
            wave_lds_sync();
            bool done = false;
            while (last - first > SF_MAX) {
                if (depth == 0) {                                    // depth limit on a range too long for LDS: one lane heap-sorts it where it lies
                    if (k.lane == 0) ss_heapsort(g, first, last - first);
                    wave_fence();
                    for (int64_t i = first + k.lane; i < last; i += AASM_WAVE) perm[i] = g.ix[i];
                    done = true;
                    break;
                }
                --depth;
                const int32_t cut = (int32_t)ss_partition_wave(g, first, last, k.lane);
                KPROF_STAMP(4);
                if (gsp < SF_ST) { if (k.lane == 0) { L->gf[gsp] = cut; L->gl[gsp] = last; L->gd[gsp] = depth; } gsp++; }
                else ok = false;
                wave_lds_sync();
                last = cut;
            }
            if (done || !ok) continue;
            const int32_t n = last - first;
            for (int32_t i = k.lane; i < n; i += AASM_WAVE) { L->qs[SF_PAD + i] = g.qs[first + i]; L->qe[SF_PAD + i] = g.qe[first + i]; L->ix[SF_PAD + i] = g.ix[first + i]; }
            KPROF_STAMP(0);
            ok = sf_sort_lds(L, n, depth, perm + first, k.lane SF_KP_PASS);
            KPROF_START();
            wave_lds_sync();
        }
    }
    if (!ok) {                                                       // (cannot happen) sequential replay from the input
        wave_fence();
        if (k.lane == 0) {
            for (int64_t i = 0; i < N; i++) perm[i] = (int32_t)i;
            SortGlob acc{perm, w.in_qs + gb, w.in_qe + gb};
            ss_std_sort(acc, N, w.sort_depth_test > 0 ? w.sort_depth_test - 1 : -1, L->qs, L->qe, L->ix);   // (the LDS copy is free by now)
        }
    }
#if defined(AASM_KPROF)
    kp_acc[7] = wave_realtime() - kp_real0;                          // (100 MHz ticks, to calibrate the cycle counts)
#endif
    KPROF_FLUSH(w.prof_heap, c, k.lane);
}

// gather sorted SoA + parts (paf_data.cpp:248-261).  One wave per contig.
AASM_DEV void kb_gather_parts(const KCtx &k, const WS &w) {
    const int64_t c = k.bid;
    const int64_t gb = w.rec_off[c], N = w.rec_off[c + 1] - gb, b = gb - w.R0;
    if (N <= 0) { if (k.lane == 0) w.np[c] = 0; return; }
    const int64_t NEG = INT64_MIN;
    int64_t carry_max = -1;                       // part_end starts at -1 (:252)
    int32_t carry_parts = 0;
    int32_t *pst = w.pstart + b + c;              // N+1 slots per contig
    for (int64_t base = 0; base < N; base += AASM_WAVE) {
        const int64_t i = base + k.lane;
        const bool act = i < N;
        int64_t qs_i = 0, qe_i = NEG;
        if (act) {
            const int64_t o = gb + w.perm[b + i];
            qs_i = w.in_qs[o]; qe_i = w.in_qe[o];
            w.s_qs[b + i] = qs_i; w.s_qe[b + i] = qe_i;
            w.s_rs[b + i] = w.in_rs[o]; w.s_re[b + i] = w.in_re[o]; w.s_qt[b + i] = w.in_qt[o];
            w.s_chr[b + i] = w.in_chr[o];
            w.s_fl[b + i] = (uint8_t)((w.in_fwd[o] ? 1 : 0) | (w.in_mq[o] ? 2 : 0));
            w.s_orig[b + i] = (int32_t)(o - gb);
            w.s_ctg[b + i] = (int32_t)c;
            w.s_rb[b + i] = w.in_rng_off[o];
            w.s_rn[b + i] = (int32_t)(w.in_rng_off[o + 1] - w.in_rng_off[o]);
            if (!(coord_ok(qs_i) && coord_ok(qe_i) && coord_ok(w.in_rs[o]) && coord_ok(w.in_re[o]) && coord_ok(w.in_qt[o]))) w.status[c] = -5;   // AASM_E_OVERFLOW (benign race: same value)
        }
        const int64_t incl = wave_incl_max(qe_i, NEG);
        int64_t excl = wave_shfl_up(incl, 1, NEG);
        if (carry_max > excl) excl = carry_max;
        const bool flag = act && (excl < qs_i);
        const int inc = wave_incl_add(flag ? 1 : 0);
        const int32_t pid = carry_parts + inc - 1;
        if (act) {
            w.s_pid[b + i] = pid;
            if (flag) pst[pid] = (int32_t)i;
        }
        const int64_t tot_max = wave_bcast(incl, AASM_WAVE - 1);
        if (tot_max > carry_max) carry_max = tot_max;
        carry_parts += wave_bcast(inc, AASM_WAVE - 1);
    }
    if (k.lane == 0) { w.np[c] = carry_parts; pst[carry_parts] = (int32_t)N; }
}

// ====================================================================================
// K2  overlap slots (paf_data.cpp:294-378)
// ====================================================================================
// Slot table replaces the four N x N tables (paf_data.cpp:268-272,282): record i owns one
// slot per j in (i, jmax_i], jmax_i = last j with qry_str_j <= qry_end_i (the scan range
// of :297-299), so slot(i, j) = ov_off[i] + (j - i - 1) and every cell the reference ever
// writes has a slot; all other cells hold FAIL_EDIT / -1 there.
AASM_DEV void kb_ov_count(const KCtx &k, const WS &w) {             // thread per record
    const int64_t g = k.bid * k.nthreads + k.tid;
    if (g >= w.R) return;
    const int64_t c = w.s_ctg[g];
    const int64_t b = w.rec_off[c] - w.R0, N = w.rec_off[c + 1] - w.rec_off[c];
    const int64_t i = g - b, qe = w.s_qe[g];
    int64_t lo = i + 1, hi = N;                                      // first j with qs_j > qe
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (w.s_qs[b + mid] <= qe) lo = mid + 1; else hi = mid;
    }
    w.ov_cnt[g] = (N <= 1) ? 0 : (int32_t)(lo - 1 - i);
}

AASM_DEV void kb_ov_merge(const KCtx &k, const WS &w) {             // thread per slot
    const int64_t s = k.bid * k.nthreads + k.tid;
    int64_t steps = 0;
    bool unconn = false;
    if (s < w.S) {
        int64_t lo = 0, hi = w.R;                                    // last g with ov_off[g] <= s
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (w.ov_off[mid + 1] <= s) lo = mid + 1; else hi = mid;
        }
        const int64_t g = lo, gj = g + 1 + (s - w.ov_off[g]);
        w.ov_rec[s] = (int32_t)g;
        int64_t peq = -1, per = -1, stq = -1, str_ = -1;
        uint8_t ok = 0;
        // qry_partial_overlap (paf_data.hpp:78-86) on sorted records: qs_i <= qs_j, qs_j <= qe_i
        if (w.s_qs[g] < w.s_qs[gj] && w.s_qe[g] < w.s_qe[gj]) {
            const int64_t st = w.rng_stride;                                 // ranges as three arrays (st = 1) or as K0's 32-byte records (st = 4)
            const int64_t *il = w.rql + w.s_rb[g] * st, *ir = w.rqr + w.s_rb[g] * st, *irl = w.rrl + w.s_rb[g] * st;
            const int64_t *jl = w.rql + w.s_rb[gj] * st, *jr = w.rqr + w.s_rb[gj] * st, *jrl = w.rrl + w.s_rb[gj] * st;
            const int64_t ni = w.s_rn[g], nj = w.s_rn[gj];
            const int64_t step_i = (w.s_fl[g] & 1) ? 1 : -1, step_j = (w.s_fl[gj] & 1) ? 1 : -1;
            bool determined = false;
            int64_t min_gap = -1, mg_i = -1, mg_j = -1, p_i = 0, p_j = 0;
            if (ni > 0 && nj > 0) {
                // fast-forward: while r_i + 1 < l_j0 the loop of :308-359 only records the
                // (strictly shrinking) gap and advances p_i; jump to the first other range.
                const int64_t lj0 = jl[(0) * st];
                int64_t a = 0, z = ni;
                while (a < z) { const int64_t m = (a + z) >> 1; if (ir[(m) * st] + 1 < lj0) a = m + 1; else z = m; steps++; }
                if (a > 0) { min_gap = lj0 - (ir[(a - 1) * st] + 1); mg_i = a - 1; mg_j = 0; }
                p_i = a;
            }
            while (p_i < ni && p_j < nj) {
                steps++;
                const int64_t l_i = il[(p_i) * st], r_i = ir[(p_i) * st], l_j = jl[(p_j) * st], r_j = jr[(p_j) * st];
                if (l_i == l_j) {                                                 // :315-327
                    if (l_j == r_j) { p_j++; continue; }
                    peq = l_i; per = irl[(p_i) * st]; stq = l_j + 1; str_ = jrl[(p_j) * st] + step_j;
                    determined = true; break;
                }
                if (l_i < l_j) {                                                  // :328-346
                    if (l_j <= r_i + 1) {
                        peq = l_j - 1; per = irl[(p_i) * st] + ((l_j - 1) - l_i) * step_i; stq = l_j; str_ = jrl[(p_j) * st];
                        determined = true; break;
                    } else {
                        const int64_t gap = l_j - (r_i + 1);
                        if (min_gap == -1 || gap < min_gap) { min_gap = gap; mg_i = p_i; mg_j = p_j; }
                    }
                    p_i++;
                } else {                                                          // :347-358
                    if (l_i <= r_j - 1) {
                        peq = l_i; per = irl[(p_i) * st]; stq = l_i + 1; str_ = jrl[(p_j) * st] + (l_i + 1 - l_j) * step_j;
                        determined = true; break;
                    }
                    p_j++;
                }
            }
            if (determined || min_gap != -1) {                                    // :360-372
                if (!determined) {
                    const int64_t l_i = il[(mg_i) * st], r_i = ir[(mg_i) * st];
                    peq = r_i; per = irl[(mg_i) * st] + (r_i - l_i) * step_i; stq = jl[(mg_j) * st]; str_ = jrl[(mg_j) * st];
                }
                ok = 1;
            } else {
                unconn = true;                                                    // :373-375 (NDEBUG)
                peq = per = stq = str_ = -1;
            }
        }
        w.ov_peq[s] = peq; w.ov_per[s] = per; w.ov_stq[s] = stq; w.ov_str[s] = str_; w.ov_ok[s] = ok;
    }
    // statistics: one atomic per wave
    const int64_t tot = wave_sum(steps);
    if (k.lane == 0 && tot) atomic_add(&w.counters[CNT_RANGE_STEPS], tot);
    if (unconn) atomic_add(&w.counters[CNT_UNCONN], (int64_t)1);
}

#define REV_ORD_MAXV 12288               // most vertices of a contig that kb_rev_fill_ord can count in LDS
#define REV_ORD_MIDV 3072                // ... in the form that keeps 6 waves per SIMD resident (6.7 KB of LDS instead of 25 KB: 2 per SIMD)
AASM_DEV void kb_vcount(const KCtx &k, const WS &w) {               // thread per contig
    const int64_t c = k.bid * k.nthreads + k.tid;
    if (c >= w.C) return;
    const int64_t b = w.rec_off[c] - w.R0, N = w.rec_off[c + 1] - w.rec_off[c];
    if (N <= 1) { w.ctgV[c] = 0; return; }                          // N == 1 shortcut (:235-239)
    const int64_t P = w.ov_rank[w.ov_off[b + N]] - w.ov_rank[w.ov_off[b]];
    if (N + P + 2 > (int64_t)INT32_MAX - 64 || w.status[c] != 0) {  // vertex ids are int32; a contig flagged by the input guard is not solved
        if (w.status[c] == 0) w.status[c] = -5;                     // AASM_E_OVERFLOW
        w.ctgV[c] = 0;
        return;
    }
    w.ctgV[c] = (int32_t)(N + P + 2);                               // + src, dest (:699-700)
    atomic_max_i64(&w.counters[CNT_MAXV], N + P + 2);               // most vertices of a contig (kb_rev_fill_ord and kb_graph_build keep a counter per vertex of a contig in LDS)
    if (N > 4096) atomic_max_i64(&w.counters[CNT_MAXN], N);         // longest contig of the batch (K8 picks its queue form by it; short ones need not report)
}

AASM_DEV void kb_vfill_rec(const KCtx &k, const WS &w) {            // thread per record
    const int64_t g = k.bid * k.nthreads + k.tid;
    if (g >= w.R) return;
    const int64_t c = w.s_ctg[g];
    const int64_t V = w.ctgV[c];
    if (V == 0) return;
    const int64_t b = w.rec_off[c] - w.R0, i = g - b, vb = w.voff[c];
    w.v_i[vb + i] = (int32_t)i; w.v_j[vb + i] = (int32_t)i; w.v_slot[vb + i] = -1; w.v_ctg[vb + i] = (int32_t)c;
    if (i == 0) {
        w.v_i[vb + V - 2] = -1; w.v_j[vb + V - 2] = -1; w.v_slot[vb + V - 2] = -1; w.v_ctg[vb + V - 2] = (int32_t)c;
        w.v_i[vb + V - 1] = -2; w.v_j[vb + V - 1] = -2; w.v_slot[vb + V - 1] = -1; w.v_ctg[vb + V - 1] = (int32_t)c;
    }
}
AASM_DEV void kb_vfill_slot(const KCtx &k, const WS &w) {           // thread per slot
    const int64_t s = k.bid * k.nthreads + k.tid;
    if (s >= w.S) return;
    int32_t vid = -1;
    if (w.ov_ok[s] && w.ctgV[w.s_ctg[w.ov_rec[s]]] != 0) {           // (a contig rejected by the input guards has no vertices)
        const int64_t g = w.ov_rec[s], c = w.s_ctg[g];
        const int64_t b = w.rec_off[c] - w.R0, N = w.rec_off[c + 1] - w.rec_off[c], vb = w.voff[c];
        vid = (int32_t)(N + (w.ov_rank[s] - w.ov_rank[w.ov_off[b]]));      // lexicographic (i, j) (:371-372)
        const int64_t i = g - b, j = i + 1 + (s - w.ov_off[g]);
        w.v_i[vb + vid] = (int32_t)i; w.v_j[vb + vid] = (int32_t)j; w.v_slot[vb + vid] = s; w.v_ctg[vb + vid] = (int32_t)c;
    }
    w.ov_vid[s] = vid;
}

// ====================================================================================
// K3/K4  make_Graph (paf_data.cpp:531-696) as closed-form CSR rows
// ====================================================================================
// Adjacency order is significant.  Row of a vertex with current record j (= (j,j) or (i,j)):
//   [ -> dest            if j is in the last part                       (:565-595) ]
//   [ -> (j,k) pair vertices, k over j's overlap slots, ascending k     (:622-626,642-645) ]
//   [ -> (k,k) for the records k of j's part that start after qry_end_j (:613-619,637-640) ]
//   [ -> (k,k) for every record k of the next part                      (:653-695) ]
// which is exactly the emission order of the reference's loops, because inside a part the
// overlapping k (qry_str_k <= qry_end_j) precede the disjoint ones in sorted order.
// NON_SKIP_LINKABLE only truncates the two (k,k) ranges (dis_end / next_cnt, kb_nsl).
struct PartInfo { int64_t pl, pr, nr; bool last; };
AASM_DEV PartInfo part_of(const WS &w, int64_t c, int64_t b, int64_t j) {
    const int32_t *pst = w.pstart + b + c;
    const int32_t pid = w.s_pid[b + j], npp = w.np[c];
    PartInfo p;
    p.pl = pst[pid]; p.pr = pst[pid + 1];
    p.last = (pid == npp - 1);
    p.nr = p.last ? p.pr : pst[pid + 2];
    return p;
}

AASM_DEV void kb_nsl(const KCtx &k, const WS &w) {                  // thread per record (nsl only)
    const int64_t g = k.bid * k.nthreads + k.tid;
    if (g >= w.R) return;
    const int64_t c = w.s_ctg[g];
    const int64_t b = w.rec_off[c] - w.R0, N = w.rec_off[c + 1] - w.rec_off[c];
    if (N <= 1) return;
    const int64_t j = g - b;
    const PartInfo p = part_of(w, c, b, j);
    {   // disjoint range of j inside its part: break when min(qry_end seen) < qry_str_k (:606-612,630-636)
        int64_t kx = j + 1 + w.ov_cnt[g], mn = INT64_MAX;
        while (kx < p.pr) {
            if (mn < w.s_qs[b + kx]) break;
            if (w.s_qe[b + kx] < mn) mn = w.s_qe[b + kx];
            kx++;
        }
        w.dis_end[g] = (int32_t)kx;
    }
    if (j == p.pl) {  // how many records of THIS part a previous-part vertex (or src) links to (:547-551,663-668)
        int64_t kx = p.pl, mn = INT64_MAX;
        while (kx < p.pr) {
            if (mn < w.s_qs[b + kx]) break;
            if (w.s_qe[b + kx] < mn) mn = w.s_qe[b + kx];
            kx++;
        }
        w.next_cnt[g] = (int32_t)(kx - p.pl);
    }
}

struct RowPlan {
    int64_t c, b, N, V, vb;
    int32_t kind;         // 0 = record vertex (cur j), 1 = src, 2 = dest
    int64_t i, j, slot;   // (i, j); slot = -1 for (j, j)
    bool has_dest;
    int64_t ov0, ovn;     // j's overlap slots
    int64_t dis0, dis1;   // disjoint (k,k) range in j's part
    int64_t nx0, nx1;     // (k,k) range in the next part
    int64_t stq;          // lft.qry_str of this vertex
};
// (plan_row_c: the contig and its constants are the caller's - kb_graph_build's workgroup has them; two dependent round trips fewer)
AASM_DEV RowPlan plan_row_c(const WS &w, int64_t gv, int64_t c, int64_t b, int64_t N, int64_t V, int64_t vb) {
    RowPlan r;
    r.c = c; r.b = b; r.N = N; r.V = V; r.vb = vb;
    const int64_t v = gv - r.vb;
    r.i = w.v_i[gv]; r.j = w.v_j[gv]; r.slot = w.v_slot[gv];
    r.has_dest = false; r.ov0 = r.ovn = 0; r.dis0 = r.dis1 = r.nx0 = r.nx1 = 0; r.stq = 0;
    if (v == r.V - 1) { r.kind = 2; return r; }
    if (v == r.V - 2) {                                   // src -> part 0 (:540-563)
        r.kind = 1;
        const int32_t *pst = w.pstart + r.b + r.c;
        r.nx0 = pst[0];
        r.nx1 = w.nsl ? (int64_t)pst[0] + w.next_cnt[r.b + pst[0]] : (int64_t)pst[1];
        return r;
    }
    r.kind = 0;
    const int64_t gj = r.b + r.j;
    const PartInfo p = part_of(w, r.c, r.b, r.j);
    r.has_dest = p.last;
    if (w.nsl && p.last && w.s_qe[gj] < w.s_qs[r.b + r.N - 1]) r.has_dest = false;   // :572-576
    r.ov0 = w.ov_off[gj]; r.ovn = w.ov_cnt[gj];
    r.dis0 = r.j + 1 + r.ovn;
    r.dis1 = w.nsl ? (int64_t)w.dis_end[gj] : p.pr;
    if (!p.last) { r.nx0 = p.pr; r.nx1 = w.nsl ? p.pr + w.next_cnt[r.b + p.pr] : p.nr; }
    r.stq = (r.slot < 0) ? w.s_qs[gj] : w.ov_stq[r.slot];
    return r;
}
AASM_DEV RowPlan plan_row(const WS &w, int64_t gv) {
    const int64_t c = w.v_ctg[gv];
    return plan_row_c(w, gv, c, w.rec_off[c] - w.R0, w.rec_off[c + 1] - w.rec_off[c], w.ctgV[c], w.voff[c]);
}
// pair edge (.., j) -> (j, k) exists iff the pair vertex exists and lft.qry_str < rht.qry_str (:433-436)
AASM_DEV bool pair_edge_ok(const WS &w, const RowPlan &r, int64_t t) {
    const int64_t s = r.ov0 + t;
    return w.ov_vid[s] >= 0 && r.stq < w.ov_stq[s];
}

AASM_DEV void kb_row_count(const KCtx &k, const WS &w) {            // thread per vertex
    const int64_t gv = k.bid * k.nthreads + k.tid;
    if (gv >= w.VT) return;
    const RowPlan r = plan_row(w, gv);
    int32_t d = 0;
    if (r.kind == 1) d = (int32_t)(r.nx1 - r.nx0);
    else if (r.kind == 0) {
        d = r.has_dest ? 1 : 0;
        for (int64_t t = 0; t < r.ovn; t++) d += pair_edge_ok(w, r, t) ? 1 : 0;
        d += (int32_t)((r.dis1 > r.dis0 ? r.dis1 - r.dis0 : 0) + (r.nx1 - r.nx0));
    }
    w.deg[gv] = d;
}

// get_score (paf_data.cpp:449-521).  lft = this row's vertex, rht = (k,k) or the pair in slot `ps`.
struct EdgeL { int64_t qe, re, rs; int32_t chr; int32_t f; };         // what the score reads of the row's own record (the same for every edge of the row)
AASM_DEV EdgeL edge_left(const WS &w, const RowPlan &r) {
    const int64_t gl = r.b + r.j;
    EdgeL l;
    l.qe = w.s_qe[gl]; l.re = w.s_re[gl];
    l.rs = (r.slot < 0) ? w.s_rs[gl] : w.ov_str[r.slot];
    l.f = w.s_fl[gl] & 1; l.chr = w.s_chr[gl];
    return l;
}
AASM_DEV void score_edge_l(const WS &w, const EdgeL &l, int64_t b, int64_t kk, int64_t ps, int64_t &wq, int32_t &wr, uint8_t &fl) {
    const int64_t gr = b + kk;
    int64_t l_qe = l.qe, l_re = l.re;
    const int64_t l_rs = l.rs;
    int64_t r_qs, r_rs;
    const int64_t r_re = w.s_re[gr];
    if (ps >= 0) { l_qe = w.ov_peq[ps]; l_re = w.ov_per[ps]; r_qs = w.ov_stq[ps]; r_rs = w.ov_str[ps]; }   // :460-465
    else { r_qs = w.s_qs[gr]; r_rs = w.s_rs[gr]; }
    const int64_t qry_diff = r_qs - l_qe - 1;
    int64_t ref_diff = 0;
    int anom = 0;
    const uint8_t rfl = w.s_fl[gr];
    const bool lf = l.f != 0, rf = rfl & 1;
    const bool same_chr = l.chr == w.s_chr[gr];
    if (same_chr && lf == rf) {                                                     // :475-490
        const int64_t gap = lf ? r_rs - (l_re + 1) : l_re - (r_rs + 1);
        ref_diff += gap < 0 ? -gap * AASM_REF_NEGATIVE_PENALTY : gap;
        if (ref_diff > AASM_SV_BASELINE) { anom += 1; ref_diff = AASM_SV_BASELINE; }
    } else if (same_chr) {                                                          // :491-508
        anom += 1;
        ref_diff += AASM_SV_INV_PENALTY;
        const int64_t x = lf ? r_re - (l_re + 1) : r_rs - (l_rs + 1);
        ref_diff += x < 0 ? -x * AASM_REF_NEGATIVE_PENALTY : x;
        if (ref_diff > AASM_SV_BASELINE) { anom += 1; ref_diff = AASM_SV_BASELINE; }
    } else {                                                                        // :509-514
        anom += 1;
        ref_diff = AASM_SV_TRANS_PENALTY;
    }
    wq = qry_diff; wr = (int32_t)ref_diff;
    fl = edge_flags(anom, (rfl & 2) ? 1 : 0, 1);                                    // :518-519
}
AASM_DEV void score_edge(const WS &w, const RowPlan &r, int64_t kk, int64_t ps, int64_t &wq, int32_t &wr, uint8_t &fl) {
    score_edge_l(w, edge_left(w, r), r.b, kk, ps, wq, wr, fl);
}

// (ib: the in-degree counters of the row's contig - w.indeg + vb, or kb_graph_build's copy in LDS)
AASM_DEV void emit_edge(const WS &w, int64_t e, int32_t *ib, int32_t col, int64_t wq, int32_t wr, uint8_t fl) {
    w.e_col[e] = col; w.e_wq[e] = wq; w.e_wr[e] = wr; w.e_fl[e] = fl;
    atomic_add(&ib[col], (int32_t)1);
}

// fills rows: each lane owns one vertex of a 64-vertex tile (short rows); rows longer than
// LONG_ROW are then written cooperatively by the whole wave (coalesced, ballot compaction).
#define AASM_LONG_ROW 16
#define AASM_MID_ROW 96
// G lanes (the whole wave, or 16 of it for rows of up to a few dozen edges: four rows at a time) work on the row; lane = 0 .. G - 1
template <int G> AASM_DEV void fill_row_part(const WS &w, const RowPlan &r, int64_t e0, int lane, int gbase, int32_t *ib) {
    const int nl = G;
    // segments: [dest][pairs][disjoint][next]; lanes stride over each segment
    int64_t e = e0;
    if (r.kind == 1) {                                              // src (:552-561)
        for (int64_t t = r.nx0 + lane; t < r.nx1; t += nl) {
            const int64_t g = r.b + t;
            emit_edge(w, e + (t - r.nx0), ib, (int32_t)t, w.s_qs[g] * AASM_SV_FRONT_END_COEFFICIENT, 0,
                      edge_flags(0, (w.s_fl[g] & 2) ? 1 : 0, 1));
        }
        return;
    }
    if (r.has_dest) {                                               // :577-585,591
        if (lane == 0) {
            const int64_t gj = r.b + r.j;
            emit_edge(w, e, ib, (int32_t)(r.V - 1), (w.s_qt[gj] - w.s_qe[gj] - 1) * AASM_SV_FRONT_END_COEFFICIENT, 0,
                      edge_flags(0, 0, 0));
        }
        e += 1;
    }
    // pair edges need in-row compaction
    for (int64_t t0 = 0; t0 < r.ovn; t0 += nl) {
        const int64_t t = t0 + lane;
        const bool ok = t < r.ovn && pair_edge_ok(w, r, t);
        const uint64_t m = group_ballot<G>(ok, gbase);
        if (ok) {
            const int64_t s = r.ov0 + t, pos = e + popc64(m & lanemask_lt(lane));
            int64_t wq; int32_t wr; uint8_t fl;
            score_edge(w, r, r.j + 1 + t, s, wq, wr, fl);
            emit_edge(w, pos, ib, w.ov_vid[s], wq, wr, fl);
        }
        e += popc64(m);
    }
    for (int64_t t = r.dis0 + lane; t < r.dis1; t += nl) {
        int64_t wq; int32_t wr; uint8_t fl;
        score_edge(w, r, t, -1, wq, wr, fl);
        emit_edge(w, e + (t - r.dis0), ib, (int32_t)t, wq, wr, fl);
    }
    if (r.dis1 > r.dis0) e += r.dis1 - r.dis0;
    for (int64_t t = r.nx0 + lane; t < r.nx1; t += nl) {
        int64_t wq; int32_t wr; uint8_t fl;
        score_edge(w, r, t, -1, wq, wr, fl);
        emit_edge(w, e + (t - r.nx0), ib, (int32_t)t, wq, wr, fl);
    }
}
// one tile of AASM_WAVE rows [gv0, gv0 + AASM_WAVE) below gv_end; LDSI: the in-degree counters of the tile's (one) contig are `lcnt`, in LDS (else w.indeg)
template <bool LDSI> AASM_DEV void row_fill_tile(const WS &w, int lane, int64_t gv0, int64_t gv_end, int32_t *lcnt) {
    const int64_t gv = gv0 + lane;
    const bool act = gv < gv_end && (LDSI || !in_graph_class(w, w.v_ctg[gv]));
    int32_t d = act ? w.deg[gv] : 0;
    RowPlan r;
    if (act && d > 0) r = plan_row(w, gv);
    const bool small = act && d > 0 && d <= AASM_LONG_ROW;
    const bool big = act && d > AASM_LONG_ROW;
#if defined(AASM_HOST_EMUL)
    if (small || big) fill_row_part<1>(w, r, w.rowptr[gv], 0, 0, LDSI ? lcnt : w.indeg + r.vb);
#else
    // short rows: run with a single logical lane (ballot of a lone lane is its own bit)
    if (small) {
        int64_t e = w.rowptr[gv];
        int32_t *ib = LDSI ? lcnt : w.indeg + r.vb;
        if (r.kind == 1) {
            for (int64_t t = r.nx0; t < r.nx1; t++) {
                const int64_t g = r.b + t;
                emit_edge(w, e++, ib, (int32_t)t, w.s_qs[g] * AASM_SV_FRONT_END_COEFFICIENT, 0, edge_flags(0, (w.s_fl[g] & 2) ? 1 : 0, 1));
            }
        } else {
            if (r.has_dest) {
                const int64_t gj = r.b + r.j;
                emit_edge(w, e++, ib, (int32_t)(r.V - 1), (w.s_qt[gj] - w.s_qe[gj] - 1) * AASM_SV_FRONT_END_COEFFICIENT, 0, edge_flags(0, 0, 0));
            }
            for (int64_t t = 0; t < r.ovn; t++)
                if (pair_edge_ok(w, r, t)) {
                    int64_t wq; int32_t wr; uint8_t fl;
                    score_edge(w, r, r.j + 1 + t, r.ov0 + t, wq, wr, fl);
                    emit_edge(w, e++, ib, w.ov_vid[r.ov0 + t], wq, wr, fl);
                }
            for (int64_t t = r.dis0; t < r.dis1; t++) {
                int64_t wq; int32_t wr; uint8_t fl;
                score_edge(w, r, t, -1, wq, wr, fl);
                emit_edge(w, e++, ib, (int32_t)t, wq, wr, fl);
            }
            for (int64_t t = r.nx0; t < r.nx1; t++) {
                int64_t wq; int32_t wr; uint8_t fl;
                score_edge(w, r, t, -1, wq, wr, fl);
                emit_edge(w, e++, ib, (int32_t)t, wq, wr, fl);
            }
        }
    }
    // rows of 17 ... AASM_MID_ROW edges (most rows of a dense graph): four at a time, 16 lanes each
    uint64_t midmask = wave_ballot(big && d <= AASM_MID_ROW);
    while (midmask) {
        int src = -1;
        AASM_UNROLL
        for (int g = 0; g < 4; g++) {
            const int sg = midmask ? ffs64(midmask) - 1 : -1;
            if (midmask) midmask &= midmask - 1;
            if ((lane >> 4) == g) src = sg;
        }
        if (src >= 0) {
            const int64_t gv2 = gv0 + src;
            const RowPlan r2 = plan_row(w, gv2);
            fill_row_part<16>(w, r2, w.rowptr[gv2], lane & 15, lane & ~15, LDSI ? lcnt : w.indeg + r2.vb);
        }
    }
    // longer rows: all 64 lanes cooperate on one row at a time
    uint64_t bigmask = wave_ballot(big && d > AASM_MID_ROW);
    while (bigmask) {
        const int src = ffs64(bigmask) - 1;
        bigmask &= bigmask - 1;
        const int64_t gv2 = gv0 + src;
        const RowPlan r2 = plan_row(w, gv2);
        fill_row_part<AASM_WAVE>(w, r2, w.rowptr[gv2], lane, 0, LDSI ? lcnt : w.indeg + r2.vb);
    }
#endif
}
AASM_DEV void kb_row_fill(const KCtx &k, const WS &w) { row_fill_tile<false>(w, k.lane, k.bid * AASM_WAVE, w.VT, nullptr); }   // wave per 64 vertices

// The same rows, one lane per CANDIDATE edge (kb_graph_build).  With a lane per row a wave runs as many turns as its longest row
// (6-10 at a mean out-degree of 2.1, each turn two dependent round trips: the pair test, then the score's operands) for a third
// of its lanes.  Here the lanes first plan their rows, as above; a row's candidates are [-> dest][its overlap slots][the disjoint
// records of its part][the records of the next part] - every edge of the row is one of them, in row order, and only a slot can fail
// (pair_edge_ok) - and the tile's candidates are numbered through (prefix sums of the rows' counts, in LDS).  Then 64 candidates a
// turn: a lane finds its candidate's row (binary search in the prefix sums), takes that row's plan from the lane that made it,
// evaluates the candidate, and its place in the row is the number of candidates of the row that passed before it (ballot +
// a count per row carried from turn to turn).  All rows of the tile belong to the contig of `r0` (vertex gv0).
// P: AASM_WAVE + 1 words of LDS, this wave's;  src16: the source of contig-local edge e (kb_graph_build's step 3), or null.
template <bool LDSI> AASM_DEV void row_fill_tile_par(const WS &w, int lane, int64_t gv0, int64_t gv_end, int32_t *lcnt, int32_t *P, uint16_t *src16, int64_t eb, int64_t cc, int64_t cb, int64_t cN, int64_t cV, int64_t cvb) {
    const int64_t gv = gv0 + lane;
    const bool act = gv < gv_end;
    const int32_t d = act ? w.deg[gv] : 0;
    RowPlan r;
    r.c = 0; r.b = 0; r.N = 0; r.V = 0; r.vb = 0; r.kind = 2; r.i = 0; r.j = 0; r.slot = -1; r.has_dest = false;
    r.ov0 = r.ovn = 0; r.dis0 = r.dis1 = r.nx0 = r.nx1 = 0; r.stq = 0;
    if (act) r = plan_row_c(w, gv, cc, cb, cN, cV, cvb);
    int32_t ncand = 0;
    if (act && d > 0 && r.kind == 1) ncand = (int32_t)(r.nx1 - r.nx0);
    if (act && d > 0 && r.kind == 0) ncand = (r.has_dest ? 1 : 0) + (int32_t)r.ovn + (int32_t)(r.dis1 > r.dis0 ? r.dis1 - r.dis0 : 0) + (int32_t)(r.nx1 - r.nx0);
    const int32_t incl = wave_incl_add(ncand);
    wave_lds_sync();                                                 // (the tile before: its reads of P are done)
    P[lane] = incl - ncand;
    if (lane == AASM_WAVE - 1) P[AASM_WAVE] = incl;
    wave_lds_sync();
    const int32_t M = P[AASM_WAVE];
    const int32_t my_p0 = incl - ncand;
    const int64_t e_row = act ? w.rowptr[gv] : 0;
    const int32_t hd = (r.kind & 3) | (r.has_dest ? 4 : 0);
    EdgeL el;                                                        // the row's own record, once per row (not once per candidate)
    el.qe = el.re = el.rs = 0; el.chr = 0; el.f = 0;
    if (act && d > 0 && r.kind == 0) el = edge_left(w, r);
    const int32_t el_cf = (el.chr << 1) | el.f;
    int32_t carry = 0;                                               // candidates of MY row that passed in earlier turns
    for (int32_t c0 = 0; c0 < M; c0 += AASM_WAVE) {
        const int32_t g = c0 + lane;
        const bool in = g < M;
        int32_t lo = 0, hi = AASM_WAVE;                              // P[lo] <= g < P[hi]  (rows without candidates share a value: the last of them is taken, the one that has some)
        while (hi - lo > 1) { const int32_t mid = (lo + hi) >> 1; if (P[mid] <= g) lo = mid; else hi = mid; }
        const int rr = in ? lo : 0;
        // the row's plan, from the lane that made it
        RowPlan q;
        const int32_t qhd = wave_shfl_idx(hd, rr);
        q.c = 0; q.b = cb; q.N = 0; q.V = cV; q.vb = cvb; q.kind = qhd & 3; q.has_dest = (qhd & 4) != 0; q.i = 0;
        q.j = wave_shfl_idx((int32_t)r.j, rr); q.slot = wave_shfl_idx(r.slot, rr);
        q.ov0 = wave_shfl_idx(r.ov0, rr); q.ovn = wave_shfl_idx((int32_t)r.ovn, rr);
        q.dis0 = wave_shfl_idx((int32_t)r.dis0, rr); q.dis1 = wave_shfl_idx((int32_t)r.dis1, rr);
        q.nx0 = wave_shfl_idx((int32_t)r.nx0, rr); q.nx1 = wave_shfl_idx((int32_t)r.nx1, rr);
        q.stq = wave_shfl_idx(r.stq, rr);
        const int64_t q_e = wave_shfl_idx(e_row, rr);
        EdgeL ql;
        ql.qe = wave_shfl_idx(el.qe, rr); ql.re = wave_shfl_idx(el.re, rr); ql.rs = wave_shfl_idx(el.rs, rr);
        { const int32_t cf = wave_shfl_idx(el_cf, rr); ql.chr = cf >> 1; ql.f = cf & 1; }
        const int32_t q_p0 = wave_shfl_idx(my_p0, rr), q_carry = wave_shfl_idx(carry, rr);
        const int32_t jc = g - q_p0;                                 // candidate jc of its row
        // which candidate
        const int32_t n_dest = q.has_dest ? 1 : 0, n_ov = (int32_t)q.ovn, n_dis = (int32_t)(q.dis1 > q.dis0 ? q.dis1 - q.dis0 : 0);
        bool ok = in;
        int32_t col = 0;
        int64_t wq = 0; int32_t wr = 0; uint8_t fl = 0;
        if (in) {
            if (q.kind == 1) {                                       // src (:552-561)
                const int64_t t = q.nx0 + jc, gg = q.b + t;
                col = (int32_t)t; wq = w.s_qs[gg] * AASM_SV_FRONT_END_COEFFICIENT; wr = 0; fl = edge_flags(0, (w.s_fl[gg] & 2) ? 1 : 0, 1);
            } else if (jc < n_dest) {                                // :577-585,591
                const int64_t gj = q.b + q.j;
                col = (int32_t)(q.V - 1); wq = (w.s_qt[gj] - w.s_qe[gj] - 1) * AASM_SV_FRONT_END_COEFFICIENT; wr = 0; fl = edge_flags(0, 0, 0);
            } else if (jc < n_dest + n_ov) {                         // a pair (j, k)
                const int64_t t = jc - n_dest, sl = q.ov0 + t;
                col = w.ov_vid[sl];                                  // (the score's operands are fetched whether the pair passes or not: one round trip, not two)
                score_edge_l(w, ql, q.b, q.j + 1 + t, sl, wq, wr, fl);
                ok = col >= 0 && q.stq < w.ov_stq[sl];               // pair_edge_ok (:433-436)
            } else {
                const int64_t t = jc < n_dest + n_ov + n_dis ? q.dis0 + (jc - n_dest - n_ov) : q.nx0 + (jc - n_dest - n_ov - n_dis);
                score_edge_l(w, ql, q.b, t, -1, wq, wr, fl);
                col = (int32_t)t;
            }
        }
        const uint64_t okm = wave_ballot(ok);
        if (ok) {
            const int32_t first = q_p0 - c0 > 0 ? q_p0 - c0 : 0;     // my row's first lane of this turn
            const int64_t e = q_e + q_carry + popc64(okm & lanemask_lt(lane) & ~lanemask_lt(first));
            emit_edge(w, e, LDSI ? lcnt : w.indeg + q.vb, col, wq, wr, fl);
            if (src16) src16[e - eb] = (uint16_t)(gv0 - q.vb + rr);
        }
        // rows (their owner lanes) count what passed of theirs in this turn
        {
            const int32_t a0 = my_p0 - c0 > 0 ? my_p0 - c0 : 0;
            const int32_t a1 = my_p0 + ncand - c0 < AASM_WAVE ? my_p0 + ncand - c0 : AASM_WAVE;
            if (a1 > a0) carry += popc64(okm & lanemask_lt(a1) & ~lanemask_lt(a0));
        }
    }
}

// ====================================================================================
// reversed CSR (k_shortest_walks.hpp:180-183): in-list of v = its in-edges in ascending
// (source id, list position) = ascending contig-local edge id.
// ====================================================================================
// one in-edge as the sweeps read it: {source, qry weight (2 words), ref weight | flags << 24}
AASM_DEV I4 pack_in_edge(int32_t srcv, int64_t wq, int32_t wr, uint8_t fl) {
    I4 r; r.x = srcv; r.y = (int32_t)(uint32_t)(uint64_t)wq; r.z = (int32_t)((uint64_t)wq >> 32); r.w = (wr & 0xffffff) | ((int32_t)fl << 24);   // 0 <= wr <= SV_BASELINE < 2^24
    return r;
}
AASM_DEV int64_t te_wq(const I4 &p) { return (int64_t)(((uint64_t)(uint32_t)p.z << 32) | (uint32_t)p.y); }   // ... and back (the topologically ordered copy, K9)
AASM_DEV int32_t te_wr(const I4 &p) { return p.w & 0xffffff; }
AASM_DEV uint8_t te_fl(const I4 &p) { return (uint8_t)((uint32_t)p.w >> 24); }
AASM_DEV void kb_rev_fill(const KCtx &k, const WS &w) {             // thread per vertex (row loop)
    const int64_t gv = k.bid * k.nthreads + k.tid;
    if (gv >= w.VT) return;
    const int64_t c = w.v_ctg[gv], vb = w.voff[c], e_base = w.rowptr[vb];
    if (in_graph_class(w, c)) return;
    const int32_t u = (int32_t)(gv - vb);
    const int64_t r0 = w.rowptr[gv], r1 = w.rowptr[gv + 1];
    for (int64_t e0 = r0; e0 < r1; e0 += 4) {                        // four edges per round: heads -> (list start, cursor) -> the two stores
        int32_t hv[4], wr[4], cur[4];
        int64_t wq[4], rp[4];
        uint8_t fl[4];
        AASM_UNROLL
        for (int i = 0; i < 4; i++) { const bool in = e0 + i < r1; hv[i] = in ? w.e_col[e0 + i] : -1; wq[i] = in ? w.e_wq[e0 + i] : 0; wr[i] = in ? w.e_wr[e0 + i] : 0; fl[i] = in ? w.e_fl[e0 + i] : 0; }
        AASM_UNROLL
        for (int i = 0; i < 4; i++) { rp[i] = 0; cur[i] = 0; if (hv[i] >= 0) { rp[i] = w.rptr[vb + hv[i]]; cur[i] = atomic_add(&w.rcur[vb + hv[i]], (int32_t)1); } }
        AASM_UNROLL
        for (int i = 0; i < 4; i++) {
            if (hv[i] < 0) continue;
            const int64_t pos = rp[i] + cur[i];
            w.r_e[pos] = (int32_t)(e0 + i - e_base);
            w.tmp_pk[pos] = pack_in_edge(u, wq[i], wr[i], fl[i]);    // in the order the atomics landed; kb_rev_place / kb_rev_hdr move it to its place in r_pk
        }
    }
}

// The same for dense graphs (rows of tens to hundreds of edges: a thread per row reads its row alone and waits for the longest of
// 64): a wave takes AASM_WAVE consecutive rows - one contiguous run of edges - and its lanes stride over the EDGES of the run
// (the row of an edge: binary search over the run's 65 row pointers in LDS), so the edge arrays are read coalesced and the
// work is balanced.  Reversed-CSR phase of the C5 share: 7.1 -> 4.5-5.0 ms.
struct RevFillLds { int64_t ptr[AASM_WAVE_MAX + 1], vb[AASM_WAVE_MAX], eb[AASM_WAVE_MAX]; };
#define AASM_REVF_LDS_BYTES ((3 * AASM_WAVE_MAX + 1) * 8)
static_assert(sizeof(RevFillLds) <= AASM_REVF_LDS_BYTES, "LDS budget");
AASM_DEV void kb_rev_fill_w(const KCtx &k, const WS &w) {           // wave per AASM_WAVE vertices
    RevFillLds *L = (RevFillLds *)k.lds;
    const int64_t row0 = k.bid * AASM_WAVE;
    if (row0 >= w.VT) return;
    const int32_t nrows = (int32_t)((w.VT - row0 < AASM_WAVE) ? (w.VT - row0) : AASM_WAVE);
    for (int32_t t = k.lane; t <= nrows; t += AASM_WAVE) L->ptr[t] = w.rowptr[row0 + t];
    for (int32_t t = k.lane; t < nrows; t += AASM_WAVE) { const int64_t vb = w.voff[w.v_ctg[row0 + t]]; L->vb[t] = vb; L->eb[t] = w.rowptr[vb]; }
    wave_lds_sync();
    const int64_t seg0 = L->ptr[0], seg1 = L->ptr[nrows];
    for (int64_t e = seg0 + k.lane; e < seg1; e += AASM_WAVE) {
        int32_t lo = 0, hi = nrows;                                  // ptr[lo] <= e < ptr[hi]
        while (hi - lo > 1) { const int32_t mid = (lo + hi) >> 1; if (L->ptr[mid] <= e) lo = mid; else hi = mid; }
        const int64_t vb = L->vb[lo];
        const int32_t u = (int32_t)(row0 + lo - vb), hv = w.e_col[e];
        const int64_t rp = w.rptr[vb + hv];
        const int32_t cur = atomic_add(&w.rcur[vb + hv], (int32_t)1);
        w.r_e[rp + cur] = (int32_t)(e - L->eb[lo]);
        w.tmp_pk[rp + cur] = pack_in_edge(u, w.e_wq[e], w.e_wr[e], w.e_fl[e]);
    }
}

// Dense batches whose contigs have at most REV_ORD_MAXV vertices: the in-lists come out IN ORDER from one pass per contig, so
// there is nothing to put in order afterwards (kb_rev_place: an entry's rank among its list, 2.4 ms on the C5 share; the atomics
// of kb_rev_fill_w 2.2).  One wave per contig walks the contig's edges in edge order, 64 per step; an edge's place in its head's
// in-list is the number of edges with that head so far: a counter per vertex in LDS plus - heads are distinct inside a row -
// nothing else, if the rows of a step take their turns one after the other (a step of 64 edges holds ~3 rows at out-degree 21).
#define REV_ORD_U 4
struct RevOrdLds { int64_t ptr[AASM_WAVE_MAX + 1]; uint16_t cnt[REV_ORD_MAXV]; };   // (the launch for batches of contigs of <= REV_ORD_MIDV vertices declares only that many counters)
#define AASM_REVO_LDS_BYTES_V(maxv) ((AASM_WAVE_MAX + 1) * 8 + (maxv) * 2)
#define AASM_REVO_LDS_BYTES AASM_REVO_LDS_BYTES_V(REV_ORD_MAXV)
static_assert(sizeof(RevOrdLds) <= AASM_REVO_LDS_BYTES, "LDS budget");
AASM_DEV void kb_rev_fill_ord(const KCtx &k, const WS &w) {         // one wave per contig
    RevOrdLds *L = (RevOrdLds *)k.lds;
    const int64_t c = k.bid;
    const int64_t V = w.ctgV[c];
    if (V == 0) return;
    const int64_t vb = w.voff[c], e_base = w.rowptr[vb];
    for (int64_t t = k.lane; t < V; t += AASM_WAVE) L->cnt[t] = 0;
    for (int64_t row0 = 0; row0 < V; row0 += AASM_WAVE) {            // tiles of 64 rows: their edges are one contiguous run
        const int32_t nrows = (int32_t)((V - row0 < AASM_WAVE) ? (V - row0) : AASM_WAVE);
        wave_lds_sync();
        for (int32_t t = k.lane; t <= nrows; t += AASM_WAVE) L->ptr[t] = w.rowptr[vb + row0 + t];
        wave_lds_sync();
        const int64_t seg0 = L->ptr[0], seg1 = L->ptr[nrows];
        for (int64_t g0 = seg0; g0 < seg1; g0 += REV_ORD_U * AASM_WAVE) {   // REV_ORD_U steps of 64 edges per pair of global round trips
            int32_t rows[REV_ORD_U], hvs[REV_ORD_U], wrs[REV_ORD_U];
            int64_t rps[REV_ORD_U], wqs[REV_ORD_U];
            uint8_t fls[REV_ORD_U];
            AASM_UNROLL
            for (int u = 0; u < REV_ORD_U; u++) {
                const int64_t e = g0 + u * AASM_WAVE + k.lane;
                rows[u] = 0; hvs[u] = 0; wrs[u] = 0; wqs[u] = 0; fls[u] = 0;
                if (e < seg1) {
                    int32_t lo = 0, hi = nrows;                      // ptr[lo] <= e < ptr[hi]
                    while (hi - lo > 1) { const int32_t mid = (lo + hi) >> 1; if (L->ptr[mid] <= e) lo = mid; else hi = mid; }
                    rows[u] = lo;
                    hvs[u] = w.e_col[e]; wqs[u] = w.e_wq[e]; wrs[u] = w.e_wr[e]; fls[u] = w.e_fl[e];
                }
            }
            AASM_UNROLL
            for (int u = 0; u < REV_ORD_U; u++) { rps[u] = 0; if (g0 + u * AASM_WAVE + k.lane < seg1) rps[u] = w.rptr[vb + hvs[u]]; }
            AASM_UNROLL
            for (int u = 0; u < REV_ORD_U; u++) {
                const int64_t c0 = g0 + u * AASM_WAVE, e = c0 + k.lane;
                if (c0 >= seg1) break;
                const bool act = e < seg1;
                const int32_t row = rows[u], hv = hvs[u];
                // the rows of this step, first to last: the lanes of a row have distinct heads
                const int32_t row_first = uni(row), row_last = wave_bcast(row, (int)((seg1 - c0 < AASM_WAVE ? seg1 - c0 : AASM_WAVE) - 1));
                for (int32_t r = row_first; r <= row_last; r++) {
                    if (act && row == r) {
                        const int32_t at = L->cnt[hv];
                        L->cnt[hv] = (uint16_t)(at + 1);
                        const int64_t pos = rps[u] + at;
                        w.r_e[pos] = (int32_t)(e - e_base);
                        const I4 rec = pack_in_edge((int32_t)(row0 + row), wqs[u], wrs[u], fls[u]);
                        w.tmp_pk[pos] = rec; w.r_pk[pos] = rec;      // (short lists reach r_pk through kb_rev_hdr's register sort - a no-op here -, long ones are read there directly)
                    }
                    wave_lds_sync();
                }
            }
        }
    }
}

// In-lists longer than REV_REG_SORT entries (dense graphs: ~25 on average) go from the order the atomics of
// kb_rev_fill landed in (r_e = edge id, tmp_pk = record) to list order in r_pk: an entry's place is the number of
// smaller edge ids in its list (they are distinct).  A wave owns the in-lists of AASM_WAVE consecutive vertices - one
// contiguous run of slots - and its lanes stride over the ENTRIES of that run, whatever list they are in (the list
// of an entry: binary search over the run's 65 row pointers in LDS; the run's keys are staged in LDS when they fit),
// so the work is balanced and every entry is one coalesced 16-byte read and one write.  Short lists: kb_rev_hdr.
#define REV_REG_SORT 8
#define REVP_KEYS 2048
struct RevPlaceLds { int64_t ptr[AASM_WAVE_MAX + 1]; int32_t key[REVP_KEYS]; uint8_t skip[AASM_WAVE_MAX]; };
#define AASM_REVP_LDS_BYTES ((AASM_WAVE_MAX + 1) * 8 + REVP_KEYS * 4 + AASM_WAVE_MAX + 8)
static_assert(sizeof(RevPlaceLds) <= AASM_REVP_LDS_BYTES, "LDS budget");
// #{j in [a, b) : key[j] < kx}; eight independent reads per wait (a lane's rank loop over a list of a few hundred keys
// is otherwise one LDS round trip per key)
AASM_DEV int32_t rank_below(const int32_t *key, int32_t a, int32_t b, int32_t kx) {
    int32_t rank = 0, j = a;
    for (; j + 8 <= b; j += 8) {
        const int32_t k0 = key[j], k1 = key[j + 1], k2 = key[j + 2], k3 = key[j + 3], k4 = key[j + 4], k5 = key[j + 5], k6 = key[j + 6], k7 = key[j + 7];
        rank += (k0 < kx) + (k1 < kx) + (k2 < kx) + (k3 < kx) + (k4 < kx) + (k5 < kx) + (k6 < kx) + (k7 < kx);
    }
    for (; j < b; j++) rank += key[j] < kx ? 1 : 0;
    return rank;
}
AASM_DEV void kb_rev_place(const KCtx &k, const WS &w) {            // wave per AASM_WAVE vertices
    RevPlaceLds *L = (RevPlaceLds *)k.lds;
    const int64_t row0 = k.bid * AASM_WAVE;
    if (row0 >= w.VT) return;
    const int32_t nrows = (int32_t)((w.VT - row0 < AASM_WAVE) ? (w.VT - row0) : AASM_WAVE);
    for (int32_t t = k.lane; t <= nrows; t += AASM_WAVE) L->ptr[t] = w.rptr[row0 + t];
    wave_lds_sync();
    bool any_long = false;
    for (int32_t t = k.lane; t < nrows; t += AASM_WAVE) L->skip[t] = in_graph_class(w, w.v_ctg[row0 + t]) ? 1 : 0;   // (kb_graph_build's lists are in order already)
    wave_lds_sync();
    for (int32_t t = k.lane; t < nrows; t += AASM_WAVE) any_long |= !L->skip[t] && L->ptr[t + 1] - L->ptr[t] > REV_REG_SORT;
    if (!wave_any(any_long)) return;
    const int64_t seg0 = L->ptr[0], seg1 = L->ptr[nrows];
    const bool staged = seg1 - seg0 <= REVP_KEYS;
    if (staged) {
        for (int64_t i = seg0 + k.lane; i < seg1; i += AASM_WAVE) L->key[i - seg0] = w.r_e[i];
        wave_lds_sync();
    }
    if (!staged) {
        // a run that does not fit holds lists of hundreds of entries: those one at a time, all lanes on one list, so
        // that the key every lane compares with is ONE address (a broadcast read)
        for (int32_t r = 0; r < nrows; r++) {
            const int64_t p0 = L->ptr[r], len = L->ptr[r + 1] - p0;
            if (len <= REV_REG_SORT || L->skip[r]) continue;
            for (int64_t a = k.lane; a < len; a += AASM_WAVE) {
                const int32_t kx = w.r_e[p0 + a];
                const int32_t rank = rank_below(w.r_e + p0, 0, (int32_t)len, kx);
                w.r_pk[p0 + rank] = w.tmp_pk[p0 + a];
            }
        }
        return;
    }
    for (int64_t i = seg0 + k.lane; i < seg1; i += AASM_WAVE) {
        int32_t lo = 0, hi = nrows;                                  // ptr[lo] <= i < ptr[hi]
        while (hi - lo > 1) { const int32_t mid = (lo + hi) >> 1; if (L->ptr[mid] <= i) lo = mid; else hi = mid; }
        const int64_t p0 = L->ptr[lo], p1 = L->ptr[lo + 1];
        if (p1 - p0 <= REV_REG_SORT || L->skip[lo]) continue;
        const int32_t kx = L->key[i - seg0];
        const int32_t rank = rank_below(L->key, (int32_t)(p0 - seg0), (int32_t)(p1 - seg0), kx);
        w.r_pk[p0 + rank] = w.tmp_pk[i];
    }
}

// ====================================================================================
// K6  Kahn FIFO order of the reversed graph + DAG shortest-path tree in ONE sweep
//     (k_shortest_walks.hpp:132-175,184).  One wave per contig.
// ====================================================================================
// Popping v in Kahn-FIFO order and relaxing its in-edges in list order with the strict
// `d[to] > d[v] + w` test reproduces the reference's first-wins tie behaviour (hazard B2)
// without any extra key: targets inside one in-list are distinct, so the 64 lanes relax a
// whole in-list at once; newly free vertices are appended in list order by ballot+prefix.
//
// The sweep is a chain of dependent steps per popped vertex, and at 5 000 resident waves the chip is bound by
// instruction issue (profiles/r02_c3_pmc_sq.json), so a pop is written to cost few instructions and one memory
// round trip:
//  * every in-edge is ONE 16-byte record {source, qry weight (2 words), ref weight | flags << 24} (r_pk: written by
//    kb_rev_fill, put in in-list order by kb_rev_place / kb_rev_hdr), and kb_rev_hdr packs per vertex {in-list start,
//    in-degree} plus its first two records (the mean in-degree is ~2) into 48 bytes (rvh);
//  * the queue's front window lives in LDS, and an entry carries the vertex's 48-byte header and its final
//    distance - both are known to the lane that appends the vertex (it fetched the header together with its
//    relax loads) - so a pop reads nothing from global memory;
//  * the lanes work on the entry directly (lane t takes in-edge t: its record straight from LDS): nothing is
//    broadcast into scalar registers except the in-degree.
#define REVQ_N 32
struct RevEnt { I4 hdr; Dist d; I4 rec[2]; };                        // hdr = {in-list start (2 words), in-degree, vertex}
struct RevQ { RevEnt e[REVQ_N]; };
#define AASM_REV_LDS_BYTES (REVQ_N * 80)
static_assert(sizeof(RevQ) <= AASM_REV_LDS_BYTES, "LDS budget");
// the forward sweep's vertex header and both sweeps' per-vertex state (kb_rev_hdr, kb_graph_build)
// {-, head 0, head 1, anomaly weight 0 | weight 1 << 8} of the row [o0, o1)
AASM_DEV I4 fwd_hdr_heads(const WS &w, int64_t o0, int64_t o1) {
    I4 f1;
    f1.x = 0; f1.y = (o1 > o0) ? w.e_col[o0] : 0; f1.z = (o1 > o0 + 1) ? w.e_col[o0 + 1] : 0;
    f1.w = ((o1 > o0) ? (w.e_fl[o0] & 3) : 0) | (((o1 > o0 + 1) ? (w.e_fl[o0 + 1] & 3) : 0) << 8);
    return f1;
}
AASM_DEV void sweep_vertex_store(const WS &w, int64_t gv, int32_t ideg, int64_t o0, int64_t o1, const I4 &f1, int64_t vend) {   // vend: dest = vend - 1, src = vend - 2
    // the forward sweep's header: {row start (2 words), out-degree, -} + f1
    I4 f0;
    f0.x = (int32_t)(uint32_t)(uint64_t)o0; f0.y = (int32_t)((uint64_t)o0 >> 32); f0.z = (int32_t)(o1 - o0); f0.w = 0;
    w.fvh[2 * gv] = f0; w.fvh[2 * gv + 1] = f1;
    // the sweeps' per-vertex state (:139-141, paf_data.cpp:704-713): pending degrees, d = max() except d[dest] = 0, no best edge; anomaly distance -1 except src = 0
    w.cnt_tmp[gv] = (int32_t)(o1 - o0);
    w.cnt_tmp2[gv] = ideg;
    w.sp_d[gv] = (gv == vend - 1) ? dist_zero() : dist_max();
    w.sp_best[gv] = -1;
    w.an[gv] = (gv == vend - 2) ? 0 : -1;
    if (w.pend) w.pend[gv] = ideg;                                   // (kb_chain's prep wave: in-neighbours still without keys)
}
AASM_DEV void sweep_vertex_init(const WS &w, int64_t gv, int32_t ideg) {
    const int64_t o0 = w.rowptr[gv], o1 = w.rowptr[gv + 1];
    sweep_vertex_store(w, gv, ideg, o0, o1, fwd_hdr_heads(w, o0, o1), w.voff[w.v_ctg[gv] + 1]);
}

AASM_DEV void kb_rev_hdr(const KCtx &k, const WS &w) {              // thread per vertex
    const int64_t gv = k.bid * k.nthreads + k.tid;
    if (gv >= w.VT || in_graph_class(w, w.v_ctg[gv])) return;
    const int64_t r0 = w.rptr[gv], r1 = w.rptr[gv + 1];
    I4 h[3];
    for (int t = 0; t < 3; t++) { h[t].x = h[t].y = h[t].z = h[t].w = 0; }
    h[0].x = (int32_t)(uint32_t)(uint64_t)r0; h[0].y = (int32_t)((uint64_t)r0 >> 32); h[0].z = (int32_t)(r1 - r0);
    const int32_t ideg = (int32_t)(r1 - r0);
    if (ideg >= 2 && ideg <= REV_REG_SORT) {
        // kb_rev_fill left the row in the order its atomics landed: a short row (the mean in-degree is ~2) is put in
        // list order here, in registers - an entry's place is the number of smaller edge ids (they are distinct)
        int32_t key[REV_REG_SORT];
        I4 rec[REV_REG_SORT];
AASM_UNROLL
        for (int i = 0; i < REV_REG_SORT; i++) {
            key[i] = INT32_MAX;
            if (i < ideg) { key[i] = w.r_e[r0 + i]; rec[i] = w.tmp_pk[r0 + i]; } else rec[i] = h[0];
        }
AASM_UNROLL
        for (int i = 0; i < REV_REG_SORT; i++) {
            int32_t rank = 0;
AASM_UNROLL
            for (int j = 0; j < REV_REG_SORT; j++) rank += (key[j] < key[i]) ? 1 : 0;
            if (i < ideg) {
                w.r_pk[r0 + rank] = rec[i];
                if (rank == 0) h[1] = rec[i];
                if (rank == 1) h[2] = rec[i];
            }
        }
    } else if (ideg == 1) {
        h[1] = w.tmp_pk[r0];
        w.r_pk[r0] = h[1];
    } else if (ideg > REV_REG_SORT) {                                 // (kb_rev_place put the longer lists in order)
        h[1] = w.r_pk[r0];
        h[2] = w.r_pk[r0 + 1];
    }
    for (int t = 0; t < 3; t++) w.rvh[3 * gv + t] = h[t];
    sweep_vertex_init(w, gv, (int32_t)(r1 - r0));
}

// ====================================================================================
// K4 rows + reversed CSR + sweep headers of ONE contig in one workgroup (kb_graph_build)
// ====================================================================================
// Sparse batches whose contigs are all small (a 1 000-record contig: ~1 400 vertices, ~3 000 edges).  The launches this replaces -
// kb_row_fill (one global atomic per edge on indeg), the scan of indeg, kb_rev_fill (one more global atomic per edge and the
// entries in landing order through r_e / tmp_pk), kb_rev_place, kb_rev_hdr - were 1.75 ms of a 12.1 ms step at 5 000 contigs and
// a chain of six launches for a single contig.  Nothing in them crosses a contig: an in-list holds edges of its own contig and
// a contig's in-lists occupy exactly its edge range (rptr[vb] = rowptr[vb]).  So a workgroup takes a contig and keeps the
// counters in LDS:
//   1  rows as kb_row_fill writes them, a lane per candidate edge (row_fill_tile_par), the in-degree counts by LDS atomics;
//   2  in-list starts: a workgroup scan of the counts (rptr = the contig's edge base + the local start);
//   3  every edge notes its contig-local id in its head's list - LDS, landing order (a lane per edge: step 1 left the sources in LDS);
//   4  every list entry finds its place (the number of smaller edge ids in its list - they are distinct -, k_shortest_walks.hpp:180-183:
//      ascending (source, list position)) and writes its 16-byte record there, the first two of a list into the vertex header too;
//   5  the headers and the sweeps' state per vertex (kb_rev_hdr's).
// r_pk / rvh / fvh / rptr and the state arrays come out exactly as from the launches above.
#define GB_TPB 256
#define GB_U 4                           // list entries / vertices of a thread whose loads leave together (steps 4 and 5)
#define GB_MAXV 1792
#define GB_MAXE 4096                     // (both far below 65 536: edge ids and sources are 16-bit words in LDS)
#define GB_MAXV_L 3584                   // ... and the form for contigs of up to ~2 500 records: 62 KB of LDS, two workgroups per CU
#define GB_MAXE_L 8192
#define AASM_GB_LDS_BYTES_T(V, E) ((V) * 4 + ((V) + 4) * 4 + 16 * 4 + (GB_TPB / 64) * (AASM_WAVE_MAX + 1) * 4 + (E) * 4)
#define AASM_GB_LDS_BYTES AASM_GB_LDS_BYTES_T(GB_MAXV, GB_MAXE)
template <int MV, int ME> struct GbLdsT { int32_t cnt[MV]; int32_t rp[MV + 4]; int32_t aux[16]; int32_t P[GB_TPB / 64][AASM_WAVE_MAX + 1]; uint16_t ks[ME]; uint16_t src[ME]; };   // 31.8 KB: five workgroups per CU
static_assert(sizeof(GbLdsT<GB_MAXV, GB_MAXE>) <= AASM_GB_LDS_BYTES && sizeof(GbLdsT<GB_MAXV_L, GB_MAXE_L>) <= AASM_GB_LDS_BYTES_T(GB_MAXV_L, GB_MAXE_L), "LDS budget");
template <int MV, int ME>
AASM_DEV void kb_graph_build(const KCtx &k, const WS &w) {          // workgroup per contig
    const int64_t c = k.bid;
    const int64_t V = w.ctgV[c];
    if (V == 0 || w.gb_flag[c] != (MV == GB_MAXV ? 1 : 2)) return;   // (a contig of the other form, or of the separate launches)
    typedef GbLdsT<MV, ME> GbLds;
    GbLds *L = (GbLds *)k.lds;
    const int64_t vb = w.voff[c], eb = w.rowptr[vb];
    const int64_t cb = w.rec_off[c] - w.R0, cN = w.rec_off[c + 1] - w.rec_off[c];
    const int32_t E = (int32_t)(w.rowptr[vb + V] - eb);
    const int nth = k.nthreads, tid = k.tid;
    const int nw = nth / AASM_WAVE, wv = tid / AASM_WAVE;            // (the host emulation: one thread, one wave of one lane)
    for (int32_t v = tid; v < V; v += nth) L->cnt[v] = 0;
    block_barrier();
    KPROF_DECL;
    KPROF_START();
    // ---- 1: rows (paf_data.cpp:531-696), a tile of AASM_WAVE vertices per wave and turn
    for (int64_t t0 = (int64_t)wv * AASM_WAVE; t0 < V; t0 += (int64_t)nw * AASM_WAVE) row_fill_tile_par<true>(w, k.lane, vb + t0, vb + V, L->cnt, L->P[wv], L->src, eb, c, cb, cN, V, vb);
    block_barrier();                                                 // (the rows are read again below: workgroup-scope release / acquire)
    KPROF_STAMP(0);
    // ---- 2: in-list starts
    {
        const int32_t K = (int32_t)((V + nth - 1) / nth);
        const int32_t a0 = tid * K < (int32_t)V ? tid * K : (int32_t)V, a1 = a0 + K < (int32_t)V ? a0 + K : (int32_t)V;
        int32_t sum = 0;
        for (int32_t v = a0; v < a1; v++) sum += L->cnt[v];
        const int32_t incl = wave_incl_add(sum);
        if (k.lane == AASM_WAVE - 1) L->aux[wv] = incl;
        block_barrier();
        int32_t base = incl - sum;
        for (int i = 0; i < wv; i++) base += L->aux[i];
        for (int32_t v = a0; v < a1; v++) {
            const int32_t d = L->cnt[v];
            L->rp[v] = base; L->cnt[v] = 0; base += d;               // (cnt: the cursors of step 3)
            if (w.indeg) w.indeg[vb + v] = d;                        // (a batch with contigs of the separate launches: their in-list starts come from a scan over everybody's counts)
        }
        if (tid == 0) { L->rp[V] = E; w.rptr[vb + V] = eb + E; }     // (= the next contig's first start: the same value from both)
        block_barrier();
    }
    KPROF_STAMP(1);
    // ---- 3: the edges into their heads' lists
    for (int32_t el = tid; el < E; el += nth) {
        const int32_t hv = w.e_col[eb + el];
        const int32_t cur = atomic_add(&L->cnt[hv], (int32_t)1);
        L->ks[L->rp[hv] + cur] = (uint16_t)el;
    }
    block_barrier();
    KPROF_STAMP(2);
    // ---- 4: list order, records, the two header records
    // (GB_U entries of a thread at a time: their loads leave together - one round trip for the four, where a loop over single entries
    // waits for each one's loads behind the stores of the entry before)
    for (int32_t p0 = tid; p0 < E; p0 += GB_U * nth) {
        uint32_t en[GB_U]; int32_t hv[GB_U], wr[GB_U]; int64_t wq[GB_U]; uint8_t fl[GB_U];   // en: edge id | source << 16
        AASM_UNROLL
        for (int i = 0; i < GB_U; i++) { const int32_t p = p0 + i * nth; en[i] = 0u; if (p < E) { const uint32_t el = L->ks[p]; en[i] = el | ((uint32_t)L->src[el] << 16); } }
        AASM_UNROLL
        for (int i = 0; i < GB_U; i++) {
            const int64_t e = eb + (en[i] & 0xffffu);
            const bool in = p0 + i * nth < E;
            hv[i] = in ? w.e_col[e] : 0; wq[i] = in ? w.e_wq[e] : 0; wr[i] = in ? w.e_wr[e] : 0; fl[i] = in ? w.e_fl[e] : 0;
        }
        AASM_UNROLL
        for (int i = 0; i < GB_U; i++) {
            if (p0 + i * nth >= E) break;
            const uint32_t key = en[i] & 0xffffu;
            const I4 rec = pack_in_edge((int32_t)(en[i] >> 16), wq[i], wr[i], fl[i]);
            const int32_t l0 = L->rp[hv[i]], l1 = L->rp[hv[i] + 1];
            int32_t rank = 0, j = l0;
            for (; j + 4 <= l1; j += 4) {
                const uint32_t k0 = L->ks[j], k1 = L->ks[j + 1], k2 = L->ks[j + 2], k3 = L->ks[j + 3];
                rank += (k0 < key) + (k1 < key) + (k2 < key) + (k3 < key);
            }
            for (; j < l1; j++) rank += (L->ks[j] < key) ? 1 : 0;
            w.r_pk[eb + l0 + rank] = rec;
            if (rank < 2) w.rvh[3 * (vb + hv[i]) + 1 + rank] = rec;
        }
    }
    KPROF_STAMP(3);
    // ---- 5: headers + state (kb_rev_hdr)
    for (int32_t v0 = tid; v0 < V; v0 += GB_U * nth) {
        int64_t o0[GB_U], o1[GB_U];
        I4 f1[GB_U];
        AASM_UNROLL
        for (int i = 0; i < GB_U; i++) { const int32_t v = v0 + i * nth; const bool in = v < V; o0[i] = in ? w.rowptr[vb + v] : 0; o1[i] = in ? w.rowptr[vb + v + 1] : 0; }
        AASM_UNROLL
        for (int i = 0; i < GB_U; i++) f1[i] = fwd_hdr_heads(w, o0[i], o1[i]);
        AASM_UNROLL
        for (int i = 0; i < GB_U; i++) {
            const int32_t v = v0 + i * nth;
            if (v >= V) break;
            const int64_t gv = vb + v;
            const int32_t ideg = L->rp[v + 1] - L->rp[v];
            const int64_t r0 = eb + L->rp[v];
            I4 h, z;
            z.x = z.y = z.z = z.w = 0;
            h.x = (int32_t)(uint32_t)(uint64_t)r0; h.y = (int32_t)((uint64_t)r0 >> 32); h.z = ideg; h.w = 0;
            w.rptr[gv] = r0;
            w.rvh[3 * gv] = h;
            if (ideg < 1) w.rvh[3 * gv + 1] = z;
            if (ideg < 2) w.rvh[3 * gv + 2] = z;
            sweep_vertex_store(w, gv, ideg, o0[i], o1[i], f1[i], vb + V);
        }
    }
    KPROF_STAMP(4);
    KPROF_FLUSH(w.prof_gb, c, tid);
}

// Contigs per wave: the graphs are long chains (~1.4 vertices per Kahn level, in-degree ~2), so a wave that
// holds ONE contig issues every instruction of a pop for two busy lanes, and at 5 000 contigs the sweeps were
// bound by instruction issue.  A wave therefore carries AASM_WAVE / G contigs, G lanes each (G = 32 on big sparse
// batches - 16 measured the same on 5 000 contigs and slower below -, 64 on dense or small ones, which are bound by
// the chain per contig: aasm_pipeline.h picks): what used to be wave-uniform (queue head
// and tail, the popped vertex) is uniform per lane group and lives in vector registers; ballots are cut to the
// group's bits.
#ifndef AASM_SWEEP_G
#define AASM_SWEEP_G (AASM_WAVE >= 32 ? 32 : 1)
#endif
template <int G> struct SweepGrp {
    static constexpr int N = AASM_WAVE / G;                          // contigs per wave
    int g, gl;
    AASM_MEM explicit SweepGrp(int lane) : g(lane / G), gl(lane % G) {}
    AASM_MEM uint32_t bits(uint64_t m) const { return G >= 64 ? 0u : (uint32_t)((m >> (g * (G & 63))) & ((1ull << (G & 63)) - 1ull)); }
    AASM_MEM int32_t below(uint64_t m) const {                       // set bits of my group below my lane
        if (G >= 64) return popc64(m & lanemask_lt(gl));
        return __builtin_popcount(bits(m) & ((1u << gl) - 1u));
    }
    AASM_MEM int32_t count(uint64_t m) const { return G >= 64 ? popc64(m) : __builtin_popcount(bits(m)); }
};

// Progress words of one chain-class workgroup (kb_chain), in LDS.  sweep wave -> prep wave: `tail_pub` vertices of rev_order are
// final (d, best and the order entry are in memory); prep wave -> heap wave: a vertex's header words in global memory turn from
// -1 to their values; `sweep_done` / `prep_done` end the waits (1: finished, 2: gave up).
struct ChainSync { int32_t tail_pub, sweep_done, prep_done, ord_pub, ord_done, pad[3]; };   // (ord_pub: positions of the BFS order that are complete; ord_done: 1 = all of them, 2 = a header never came)
#if defined(AASM_HOST_EMUL)
AASM_DEV void st_shared_i32(int32_t *p, int32_t v) { *p = v; }
#else
AASM_DEV void st_shared_i32(int32_t *p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#endif

template <int G, bool PUB = false>
AASM_DEV void kb_rev_sweep(const KCtx &k, const WS &w, ChainSync *S = nullptr) {
    const SweepGrp<G> sg(k.lane);
    const int64_t c = k.bid * SweepGrp<G>::N + sg.g;
    const bool valid = c < w.C && (PUB || !in_chain_class(w, c));   // (the chain class sweeps inside kb_chain)
    const int64_t V = valid ? (int64_t)w.ctgV[c] : 0;
    const int64_t vb = valid ? w.voff[c] : 0;
    RevQ *Q = (RevQ *)k.lds + sg.g;                                  // queue positions [head, lds_hi)
    Dist *d = w.sp_d + vb;
    int32_t *best = w.sp_best + vb, *q = w.rev_order + vb, *cnt = w.cnt_tmp + vb;
    const I4 *rvh = w.rvh + 3 * vb;
    int32_t tail = 0, lds_hi = 0;
    for (int64_t base = 0; wave_any(base < V); base += G) {        // sources in ascending id (:139-141); kb_rev_hdr initialised cnt / d / best
        const int64_t v = base + sg.gl;
        const bool z = v < V && cnt[v] == 0;
        const uint64_t m = wave_ballot(z);
        const int32_t at = tail + sg.below(m);
        if (z) {
            q[at] = (int32_t)v;
            if (at < REVQ_N) { RevEnt *E = &Q->e[at]; I4 hd = rvh[3 * v]; hd.w = (int32_t)v; E->hdr = hd; E->d = d[v]; E->rec[0] = rvh[3 * v + 1]; E->rec[1] = rvh[3 * v + 2]; }
        }
        tail += sg.count(m);
    }
    lds_hi = tail < REVQ_N ? tail : REVQ_N;
    wave_lds_sync();
    int32_t pub = 0;                                                 // (PUB: G == 64, tail is wave-uniform)
    // (every lane stores the same word: with `if (lane == 0)` around the store inside the pop loop the compiler unswitched the whole
    // loop on the lane id - lanes != 0 first, lane 0 after them - and the sweep, whose lanes work together, stopped after two pops)
    if (PUB) { pub = uni(tail); store_drain(); st_shared_i32(&S->tail_pub, pub); }
    int32_t head = 0;
    while (wave_any(head < tail)) {
        const bool live = head < tail;
        RevEnt *E = &Q->e[head & (REVQ_N - 1)];
        if (wave_any(live && head >= lds_hi)) {                      // beyond the LDS window (wide frontiers): bring the entry in
            if (live && head >= lds_hi && sg.gl == 0) { const int32_t v = q[head]; I4 hd = rvh[3 * v]; hd.w = v; E->hdr = hd; E->d = d[v]; E->rec[0] = rvh[3 * v + 1]; E->rec[1] = rvh[3 * v + 2]; }
            wave_lds_sync();
        }
        I4 hd; hd.x = hd.y = hd.z = hd.w = 0;
        Dist dv = dist_max();
        if (live) { hd = E->hdr; dv = E->d; head++; }
        const int32_t deg = hd.z, v = hd.w;
        const int64_t r0 = (int64_t)(((uint64_t)(uint32_t)hd.y << 32) | (uint32_t)hd.x);
        const bool reach = dv.anom >= 0;                             // max() is the only distance with a negative anom here (:166)
        for (int32_t base = 0; wave_any(base < deg); base += G) {
            const int32_t t = base + sg.gl;
            const bool act = t < deg;
            bool z = false;
            int32_t u = 0;
            I4 uh0, uh1, uh2;
            Dist du = dist_max();
            uh0.x = uh0.y = uh0.z = uh0.w = 0; uh1 = uh0; uh2 = uh0;
            if (act) {
                I4 rc;                                               // (two loads kept apart: as one flat load of a selected address it waited for vmcnt(0) every time)
                if (t < 2) rc = E->rec[t];
                else { rc = w.r_pk[r0 + t]; asm volatile("" ::: "memory"); }
                u = rc.x;
                Dist wd;
                wd.qry = (int64_t)(((uint64_t)(uint32_t)rc.z << 32) | (uint32_t)rc.y); wd.ref = rc.w & 0xffffff;
                const int32_t fl = (int32_t)((uint32_t)rc.w >> 24);
                wd.anom = fl & 3; wd.qnz = (fl >> 2) & 1; wd.qtot = (fl >> 3) & 1; wd.pad = 0;
                du = d[u];
                const int32_t left = cnt[u] - 1;
                uh0 = rvh[3 * u]; uh1 = rvh[3 * u + 1]; uh2 = rvh[3 * u + 2];   // in case u becomes free now
                const Dist cand = dist_add(dv, wd);
                // cand < d[u] (:168): d[u] is max() (anom < 0) or a real distance; cand is a real distance
                const int64_t sc = cand.qry + cand.ref, su = du.qry + du.ref;
                const int32_t tc = cand.qtot ? cand.qtot : 1, tu = du.qtot ? du.qtot : 1;
                const bool better = reach & ((du.anom < 0) | (sc < su) | ((sc == su) & ((cand.anom < du.anom) | ((cand.anom == du.anom) & ((int64_t)cand.qnz * tu > (int64_t)du.qnz * tc)))));
                if (better) { du = cand; d[u] = cand; best[u] = v; }                  // :168-171
                cnt[u] = left;
                z = left == 0;
            }
            const uint64_t m = wave_ballot(z);
            const int32_t at = tail + sg.below(m);
            if (z) {
                q[at] = u;
                if (lds_hi == tail && at - head < REVQ_N - 1) { RevEnt *N = &Q->e[at & (REVQ_N - 1)]; uh0.w = u; N->hdr = uh0; N->d = du; N->rec[0] = uh1; N->rec[1] = uh2; }
            }
            const int32_t nnew = sg.count(m);
            if (lds_hi == tail) { int32_t room = REVQ_N - 1 - (tail - head); if (room > nnew) room = nnew; if (room < 0) room = 0; lds_hi += room; }   // (the slot of the entry at hand stays untouched)
            tail += nnew;
        }
        wave_lds_sync();
        // a vertex is final the moment it is appended (every out-edge relaxed): its d / best / order entry leave this wave before
        // the count that tells the prep wave about them
        if (PUB) {
            const int32_t tu = uni(tail);                            // (scalar: the publish is a uniform branch, no exec masking around the fence)
            if (tu != pub) { store_drain(); st_shared_i32(&S->tail_pub, tu); pub = tu; }
        }
    }
    if (valid && tail != (int32_t)V && sg.gl == 0) set_status(w, c, -6);   // cycle: cannot happen (:144-148)
    if (PUB) { store_drain(); st_shared_i32(&S->sweep_done, 1); }
}

// forward Kahn order (paf_data.cpp:742-746) + anomaly distance to dest.  The reference
// runs Dial's bucketed BFS on the 0/1/2 anomaly weights (k_weighted_bfs.hpp:16-37) and
// keeps only anom_dis[dest] (paf_data.cpp:715,1615); on a DAG the same scalar is the
// min-plus DP along the topological order, folded into this sweep.  Same shape as the reverse sweep.
struct FwdEnt { I4 a, b; };                                          // a = {row start (2 words), out-degree, vertex}, b = {anomaly distance, head 0, head 1, weights}
struct FwdQ { FwdEnt e[REVQ_N]; };
#define AASM_FWD_LDS_BYTES (REVQ_N * 32)
static_assert(sizeof(FwdQ) <= AASM_FWD_LDS_BYTES, "LDS budget");
template <int G>
AASM_DEV void kb_fwd_sweep(const KCtx &k, const WS &w) {
    const SweepGrp<G> sg(k.lane);
    const int64_t c = k.bid * SweepGrp<G>::N + sg.g;
    const bool valid = c < w.C;
    const int64_t V = valid ? (int64_t)w.ctgV[c] : 0;
    const int64_t vb = valid ? w.voff[c] : 0;
    FwdQ *Q = (FwdQ *)k.lds + sg.g;                                  // queue positions [head, lds_hi)
    int32_t *q = w.fwd_order + vb, *pos = w.fwd_pos + vb, *cnt = w.cnt_tmp2 + vb, *an = w.an + vb;
    const I4 *fvh = w.fvh + 2 * vb;
    int32_t tail = 0, lds_hi = 0;
    for (int64_t base = 0; wave_any(base < V); base += G) {        // kb_rev_hdr initialised cnt / an
        const int64_t v = base + sg.gl;
        const bool z = v < V && cnt[v] == 0;
        const uint64_t m = wave_ballot(z);
        if (z) {
            const int32_t t = tail + sg.below(m);
            q[t] = (int32_t)v; pos[v] = t;
            if (t < REVQ_N) { I4 fa = fvh[2 * v], fb = fvh[2 * v + 1]; fa.w = (int32_t)v; fb.x = an[v]; Q->e[t].a = fa; Q->e[t].b = fb; }
        }
        tail += sg.count(m);
    }
    lds_hi = tail < REVQ_N ? tail : REVQ_N;
    wave_lds_sync();
    int32_t head = 0;
    while (wave_any(head < tail)) {
        const bool live = head < tail;
        FwdEnt *E = &Q->e[head & (REVQ_N - 1)];
        if (wave_any(live && head >= lds_hi)) {
            if (live && head >= lds_hi && sg.gl == 0) { const int32_t u = q[head]; I4 fa = fvh[2 * u], fb = fvh[2 * u + 1]; fa.w = u; fb.x = an[u]; E->a = fa; E->b = fb; }
            wave_lds_sync();
        }
        I4 fa, fb; fa.x = fa.y = fa.z = fa.w = 0; fb = fa;
        if (live) { fa = E->a; fb = E->b; head++; }
        const int32_t deg = fa.z, au = fb.x;
        const int64_t r0 = (int64_t)(((uint64_t)(uint32_t)fa.y << 32) | (uint32_t)fa.x);
        for (int32_t base = 0; wave_any(base < deg); base += G) {
            const int32_t t = base + sg.gl;
            const bool act = t < deg;
            bool z = false;
            int32_t v = 0, av = -1;
            I4 va, vbb;
            va.x = va.y = va.z = va.w = 0; vbb = va;
            if (act) {
                int32_t wa;
                if (t < 2) { v = t == 0 ? fb.y : fb.z; wa = (fb.w >> (8 * t)) & 3; }
                else { v = w.e_col[r0 + t]; wa = w.e_fl[r0 + t] & 3; }
                av = an[v];
                const int32_t left = cnt[v] - 1;
                va = fvh[2 * v]; vbb = fvh[2 * v + 1];                // in case v becomes free now
                const int32_t nd = au + wa;
                if ((au >= 0) & ((av < 0) | (nd < av))) { av = nd; an[v] = nd; }
                cnt[v] = left;
                z = left == 0;
            }
            const uint64_t m = wave_ballot(z);
            const int32_t at = tail + sg.below(m);
            if (z) {
                q[at] = v; pos[v] = at;
                if (lds_hi == tail && at - head < REVQ_N - 1) { FwdEnt *N = &Q->e[at & (REVQ_N - 1)]; va.w = v; vbb.x = av; N->a = va; N->b = vbb; }
            }
            const int32_t nnew = sg.count(m);
            if (lds_hi == tail) { int32_t room = REVQ_N - 1 - (tail - head); if (room > nnew) room = nnew; if (room < 0) room = 0; lds_hi += room; }
            tail += nnew;
        }
        wave_lds_sync();
    }
    if (valid && sg.gl == 0 && V > 0) {
        w.anom_dest[c] = an[V - 1];
        if (tail != (int32_t)V) set_status(w, c, -6);
    }
}

// ====================================================================================
// CSR copy in forward-topological order, used by internal_shortest_path_recover (K9):
// its windows are runs of consecutive topological positions, so with this layout a window's
// vertices AND all their edges are two contiguous, coalesced reads, and edge heads are stored
// as topological positions (what the DP needs) instead of vertex ids.
// ====================================================================================
AASM_DEV void kb_topo_count(const KCtx &k, const WS &w) {           // thread per (contig, position)
    const int64_t gp = k.bid * k.nthreads + k.tid;
    if (gp >= w.VT) return;
    const int64_t vb = w.voff[w.v_ctg[gp]];
    const int32_t u = w.fwd_order[gp];
    w.tp_deg[gp] = (int32_t)(w.rowptr[vb + u + 1] - w.rowptr[vb + u]);
    w.tp_vj[gp] = w.v_j[vb + u];                                     // src / dest carry -1 / -2: never a whitelist match
}
AASM_DEV void kb_topo_fill(const KCtx &k, const WS &w) {            // wave per 64 positions
    const int64_t gp = k.bid * AASM_WAVE + k.lane;
    const bool act = gp < w.VT;
    int64_t vb = 0, r0 = 0, t0 = 0;
    int32_t deg = 0;
    if (act) {
        vb = w.voff[w.v_ctg[gp]];
        const int32_t u = w.fwd_order[gp];
        r0 = w.rowptr[vb + u]; deg = w.tp_deg[gp]; t0 = w.tp_ptr[gp];
    }
    const bool big = deg > AASM_LONG_ROW;
    if (act && !big)
        for (int32_t t = 0; t < deg; t++) {
            w.te_pk[t0 + t] = pack_in_edge(w.fwd_pos[vb + w.e_col[r0 + t]], w.e_wq[r0 + t], w.e_wr[r0 + t], w.e_fl[r0 + t]);   // (one 16-byte store: as four arrays a row of two edges dirtied four 32-byte sectors)
        }
    uint64_t bigmask = wave_ballot(big);
    while (bigmask) {
        const int src = ffs64(bigmask) - 1;
        bigmask &= bigmask - 1;
        const int64_t vb2 = wave_bcast(vb, src), r02 = wave_bcast(r0, src), t02 = wave_bcast(t0, src);
        const int32_t deg2 = wave_bcast(deg, src);
        for (int32_t t = k.lane; t < deg2; t += AASM_WAVE) {
            w.te_pk[t02 + t] = pack_in_edge(w.fwd_pos[vb2 + w.e_col[r02 + t]], w.e_wq[r02 + t], w.e_wr[r02 + t], w.e_fl[r02 + t]);
        }
    }
}

// ====================================================================================
// K7  sidetrack heaps (k_shortest_walks.hpp:191-215; leftist_heap.hpp:29-40)
// ====================================================================================
// children of u in the SP tree (:191-194), ascending id: the vertices v with best[v] == u.  Such a v has an edge
// v -> u, so it is a source in u's in-list - which is in ascending (source, position) order already; parallel edges of
// one source are neighbours there.  The list is written over the front of u's OWN in-list slots (cval is indexed
// like r_pk): no counting pass, no scan, no atomics, nothing to sort.
// (ch4: the first four children, for a caller that goes on with them - kb_k7_prep: no trip through memory for what this thread just wrote)
AASM_DEV int32_t children_vertex(const WS &w, int64_t gv, int32_t *ch4 = nullptr) {
    const int64_t vb = w.voff[w.v_ctg[gv]];
    const int32_t u = (int32_t)(gv - vb);
    const int64_t r0 = w.rptr[gv], r1 = w.rptr[gv + 1];
    int32_t n = 0, prev = -1;
    int32_t c0 = -1, c1 = -1, c2 = -1, c3 = -1;
    // (a thread has two or three in-edges and every one is two dependent loads: four edges per round, their loads issued
    // together - the kernel waits on memory 93 % of its time, not on bandwidth)
    for (int64_t t = r0; t < r1; t += 4) {
        int32_t src[4], bst[4];
        AASM_UNROLL
        for (int i = 0; i < 4; i++) src[i] = (t + i < r1) ? w.r_pk[t + i].x : -1;
        AASM_UNROLL
        for (int i = 0; i < 4; i++) bst[i] = src[i] >= 0 ? w.sp_best[vb + src[i]] : -1;
        AASM_UNROLL
        for (int i = 0; i < 4; i++) {
            if (src[i] >= 0 && src[i] != prev && bst[i] == u) {
                w.cval[r0 + n] = src[i];
                c0 = n == 0 ? src[i] : c0; c1 = n == 1 ? src[i] : c1; c2 = n == 2 ? src[i] : c2; c3 = n == 3 ? src[i] : c3;
                n++;
            }
            if (src[i] >= 0) prev = src[i];
        }
    }
    w.ccnt[gv] = n;
    if (ch4) { ch4[0] = c0; ch4[1] = c1; ch4[2] = c2; ch4[3] = c3; }
    return n;
}
AASM_DEV void kb_children(const KCtx &k, const WS &w) {              // thread per vertex
    const int64_t gv = k.bid * k.nthreads + k.tid;
    if (gv >= w.VT || in_chain_class(w, w.v_ctg[gv])) return;       // (the chain class: its prep wave does this, kb_chain)
    children_vertex(w, gv);
}
// K7 pre-pass 1, thread per vertex u (k_shortest_walks.hpp:204-210 without the insert): the sidetrack cost
// c = w + d[v] - d[u] of every out-edge that goes into u's heap - not when d[v] is max() (:204-205), and not
// the FIRST edge of the list that is u's tree edge (:207-210) - written COMPACTED to the front of u's row of
// st_cost, in list order, with the edge head v in the pad word.  The heap wave then reads a vertex's keys with
// one coalesced load (lane t = key t) instead of flag, cost and head through three dependent loads.
// A sidetrack cost has score sum >= 0 (d[u] is minimal in the CALC_SUM order, whose first key is the sum), so
// no heap key is ever max() (sum -2): K7's key test needs no sentinel handling.  Checked here.
AASM_DEV int32_t sidetrack_vertex(const WS &w, int64_t gv) {       // (returns the key count it stores in st_n)
    const int64_t c = w.v_ctg[gv], vb = w.voff[c];
    const Dist *d = w.sp_d + vb;
    const Dist du = w.sp_d[gv];
    const int32_t bu = w.sp_best[gv];
    const int64_t r0 = w.rowptr[gv], r1 = w.rowptr[gv + 1];
    const bool reach = !dist_is_max(du);
    bool seen_p = false, bad = false;
    int32_t n_ins = 0;
    for (int64_t e0 = r0; e0 < r1; e0 += 4) {                        // four edges per round: heads, then their distances and weights, together
        int32_t hv[4], wr[4];
        int64_t wq[4];
        uint8_t fl[4];
        Dist dvv[4];
        AASM_UNROLL
        for (int i = 0; i < 4; i++) { const bool in = e0 + i < r1; hv[i] = in ? w.e_col[e0 + i] : -1; wq[i] = in ? w.e_wq[e0 + i] : 0; wr[i] = in ? w.e_wr[e0 + i] : 0; fl[i] = in ? w.e_fl[e0 + i] : 0; }
        AASM_UNROLL
        for (int i = 0; i < 4; i++) dvv[i] = hv[i] >= 0 ? d[hv[i]] : dist_max();
        AASM_UNROLL
        for (int i = 0; i < 4; i++) {
            if (hv[i] < 0 || dist_is_max(dvv[i])) continue;
            Dist cc = dist_sub(dist_add(edge_dist(wq[i], wr[i], fl[i]), dvv[i]), du);
            if (!seen_p && hv[i] == bu && dist_eq(cc, dist_zero())) seen_p = true;
            else { bad |= reach && (cc.qry + cc.ref) < 0; cc.pad = hv[i]; w.st_cost[r0 + n_ins] = cc; n_ins++; }
        }
    }
    w.st_n[gv] = n_ins;
    if (bad) w.status[c] = -6;                                       // must not happen (see above)
    return n_ins;
}
// the next four vertices along best[] (final once K6 is done): the 4-hop jump record of a vertex
AASM_DEV void tnx_vertex(const WS &w, int64_t gv) {
    const int64_t vb = w.voff[w.v_ctg[gv]];
    I4 t;
    t.x = w.sp_best[gv];
    t.y = t.x >= 0 ? w.sp_best[vb + t.x] : -1;
    t.z = t.y >= 0 ? w.sp_best[vb + t.y] : -1;
    t.w = t.z >= 0 ? w.sp_best[vb + t.z] : -1;
    w.tnx[gv] = t;
}
AASM_DEV void kb_tnx(const KCtx &k, const WS &w) {                  // thread per vertex
    const int64_t gv = k.bid * k.nthreads + k.tid;
    if (gv >= w.VT || in_chain_class(w, w.v_ctg[gv]) || in_graph_class(w, w.v_ctg[gv])) return;   // (the chain class: its prep wave, kb_chain; the small contigs of a sparse batch: kb_tnx16_wg)
    tnx_vertex(w, gv);
}
AASM_DEV void kb_sidetrack(const KCtx &k, const WS &w) {
    const int64_t gv = k.bid * k.nthreads + k.tid;
    if (gv >= w.VT || in_chain_class(w, w.v_ctg[gv])) return;
    sidetrack_vertex(w, gv);
}
// The same for dense graphs (rows of tens to hundreds of edges): a wave takes AASM_WAVE consecutive rows - one contiguous run of
// edges - in chunks of 64 edges, lanes = edges.  What is sequential per row becomes ballots: the tree edge is the lowest
// candidate lane of the row's stretch of the chunk, unless an earlier chunk had it (seen[row]); a kept edge's place is the row's
// count so far (cnt[row], in LDS) plus the kept lanes of its row below it.
struct SideLds { int64_t ptr[AASM_WAVE_MAX + 1], vb[AASM_WAVE_MAX]; Dist du[AASM_WAVE_MAX]; int32_t bu[AASM_WAVE_MAX], cnt[AASM_WAVE_MAX], seen[AASM_WAVE_MAX]; };
#define AASM_SIDE_LDS_BYTES ((AASM_WAVE_MAX + 1) * 8 + AASM_WAVE_MAX * (8 + 32 + 12) + 16)
static_assert(sizeof(SideLds) <= AASM_SIDE_LDS_BYTES, "LDS budget");
AASM_DEV void kb_sidetrack_w(const KCtx &k, const WS &w) {          // wave per AASM_WAVE vertices
    SideLds *L = (SideLds *)k.lds;
    const int64_t row0 = k.bid * AASM_WAVE;
    if (row0 >= w.VT) return;
    const int32_t nrows = (int32_t)((w.VT - row0 < AASM_WAVE) ? (w.VT - row0) : AASM_WAVE);
    for (int32_t t = k.lane; t <= nrows; t += AASM_WAVE) L->ptr[t] = w.rowptr[row0 + t];
    for (int32_t t = k.lane; t < nrows; t += AASM_WAVE) {
        L->vb[t] = w.voff[w.v_ctg[row0 + t]]; L->du[t] = w.sp_d[row0 + t]; L->bu[t] = w.sp_best[row0 + t]; L->cnt[t] = 0; L->seen[t] = 0;
    }
    wave_lds_sync();
    const int64_t seg0 = L->ptr[0], seg1 = L->ptr[nrows];
    bool bad = false;
    int32_t bad_row = 0;
    for (int64_t c0 = seg0; c0 < seg1; c0 += AASM_WAVE) {
        const int64_t e = c0 + k.lane;
        const bool act = e < seg1;
        int32_t row = 0, hv = -1;
        Dist cc = dist_zero();
        bool keepable = false, cand = false;
        if (act) {
            int32_t lo = 0, hi = nrows;                              // ptr[lo] <= e < ptr[hi]
            while (hi - lo > 1) { const int32_t mid = (lo + hi) >> 1; if (L->ptr[mid] <= e) lo = mid; else hi = mid; }
            row = lo;
            hv = w.e_col[e];
            const Dist dv = w.sp_d[L->vb[row] + hv];
            if (!dist_is_max(dv)) {                                  // :204-205
                keepable = true;
                cc = dist_sub(dist_add(edge_dist(w.e_wq[e], w.e_wr[e], w.e_fl[e]), dv), L->du[row]);
                cand = hv == L->bu[row] && dist_eq(cc, dist_zero());
            }
        }
        // my row's stretch of this chunk: lanes [ls, le)
        const int64_t ps = act ? L->ptr[row] : 0, pe = act ? L->ptr[row + 1] : 0;
        const int ls = ps > c0 ? (int)(ps - c0) : 0, le = (pe - c0 < AASM_WAVE) ? (int)(pe - c0) : AASM_WAVE;
        const uint64_t rowmask = act ? (lanemask_lt(le) & ~lanemask_lt(ls)) : 0ull;
        const uint64_t mc = wave_ballot(cand) & rowmask;
        const bool tree = cand && L->seen[row] == 0 && (ffs64(mc) - 1) == k.lane;   // :207-210: the first such edge of the list
        const bool kept = keepable && !tree;
        const uint64_t mk = wave_ballot(kept) & rowmask;
        const int32_t base = act ? L->cnt[row] : 0;
        wave_lds_sync();                                             // every lane has read seen[] / cnt[] of the chunk's rows
        if (kept) {
            const Dist du = L->du[row];
            if (!dist_is_max(du) && cc.qry + cc.ref < 0) { bad = true; bad_row = row; }
            cc.pad = hv;
            w.st_cost[ps + base + popc64(mk & lanemask_lt(k.lane))] = cc;
        }
        if (act && k.lane == ls) {                                   // the first lane of a row's stretch books for the row
            if (mc) L->seen[row] = 1;
            L->cnt[row] = base + popc64(mk);
        }
        wave_lds_sync();
    }
    for (int32_t t = k.lane; t < nrows; t += AASM_WAVE) w.st_n[row0 + t] = L->cnt[t];
    if (bad) w.status[w.v_ctg[row0 + bad_row]] = -6;                 // must not happen (see kb_sidetrack)
}
// K7 pre-pass 2, thread per vertex u: what the heap wave reads per vertex, 32 bytes:
//   vhdr  = {so, #keys, #children, first child}          so = start of u's keys, relative to the contig's first edge
//   vhdr2 = {child-list start (2 words), so(first child), row length of the first child}
// and per child-list slot cinfo = {child, so(child), row length(child), -}.  A parent hands every child the place
// of its keys, so the wave fetches a vertex's header AND keys while it still works on the vertex before it
// (the queue front; in a path-like tree the first child of the vertex at hand).  What a parent knows of a child is the
// LENGTH OF ITS ROW, an upper bound of its key count (the keys sit compacted at the front of the row): the count itself is
// in the child's own header, which arrives with the keys.  So a header depends on no other vertex's pre-pass, and the child
// list, the keys and the header of a vertex are one thread's work in one launch (kb_k7_prep; round 5 - the counts used to
// come from st_n[child], a second launch).
// (returns the two header quads: kb_chain's prep wave publishes them only after everything else the vertex's step reads is in memory)
// (nch / ch4 / nkeys: what the caller's children_vertex / sidetrack_vertex just returned; nch < 0 / nkeys < 0: read from memory)
AASM_DEV void heap_hdr_vertex(const WS &w, int64_t gv, I4 &a, I4 &b, int32_t nch = -1, const int32_t *ch4 = nullptr, int32_t nkeys = 0) {
    const int64_t vb = w.voff[w.v_ctg[gv]], e_base = w.rowptr[vb];
    const int64_t c0 = w.rptr[gv], c1 = c0 + (nch >= 0 ? nch : w.ccnt[gv]);   // (kb_children: the list sits at the front of the in-list slots)
    const int32_t fc = (c1 > c0) ? (nch >= 0 ? ch4[0] : w.cval[c0]) : -1;
    a.x = (int32_t)(w.rowptr[gv] - e_base); a.y = (nch >= 0 && nkeys >= 0) ? nkeys : w.st_n[gv]; a.z = (int32_t)(c1 - c0); a.w = fc;
    b.x = (int32_t)(uint32_t)(uint64_t)c0; b.y = (int32_t)((uint64_t)c0 >> 32);
    b.z = 0; b.w = 0;
    if (fc >= 0) { const int64_t rp0 = w.rowptr[vb + fc]; b.z = (int32_t)(rp0 - e_base); b.w = (int32_t)(w.rowptr[vb + fc + 1] - rp0); }
    for (int64_t t0 = c0; t0 < c1; t0 += 4) {                        // four children per round (their gathers issued together)
        int32_t ch[4];
        int64_t rp[4], rq[4];
        AASM_UNROLL
        for (int i = 0; i < 4; i++) ch[i] = (t0 + i < c1) ? ((nch >= 0 && t0 == c0) ? ch4[i] : w.cval[t0 + i]) : -1;
        AASM_UNROLL
        for (int i = 0; i < 4; i++) { rp[i] = 0; rq[i] = 0; if (ch[i] >= 0) { rp[i] = w.rowptr[vb + ch[i]]; rq[i] = w.rowptr[vb + ch[i] + 1]; } }
        AASM_UNROLL
        for (int i = 0; i < 4; i++) {
            if (ch[i] < 0) continue;
            I4 ci; ci.x = ch[i]; ci.y = (int32_t)(rp[i] - e_base); ci.z = (int32_t)(rq[i] - rp[i]); ci.w = 0;
            w.cinfo[t0 + i] = ci;
        }
    }
}
// sixteen tree hops = four 4-hop records chained (kb_sidetrack wrote those; -1 past dest): what K9's recovery reads, one 64-byte
// record per sixteen tree edges.  It needs the 4-hop records of OTHER vertices, hence a launch of its own (side stream: only K9 waits for it)
AASM_DEV void tnx16_vertex(const WS &w, int64_t gv) {
    const int64_t vb = w.voff[w.v_ctg[gv]];
    I4 neg; neg.x = neg.y = neg.z = neg.w = -1;
    I4 *o = (I4 *)(w.tnx16 + 16 * gv);
    I4 j = w.tnx[gv];
    o[0] = j;
    AASM_UNROLL
    for (int t = 1; t < 4; t++) { const int32_t nx = j.w; j = neg; if (nx >= 0) j = w.tnx[vb + nx]; o[t] = j; }   // (a select between two structs would go through scratch memory)
}
AASM_DEV void kb_tnx16(const KCtx &k, const WS &w) {                // thread per vertex
    const int64_t gv = k.bid * k.nthreads + k.tid;
    if (gv >= w.VT || in_chain_class(w, w.v_ctg[gv]) || in_graph_class(w, w.v_ctg[gv])) return;
    tnx16_vertex(w, gv);
}
// The same records for the small contigs of a sparse batch (the class of kb_graph_build), a workgroup per contig: with the contig's tree
// (best[]) in LDS they are sixteen LDS reads per vertex - no 4-hop records in global memory in between, one launch instead of two
// (time-neutral beside the pre-pass; 0.5 GB less traffic per 5 000 x 1 000-record step).
#define TNX_TPB 256
#define AASM_TNXWG_LDS_BYTES (GB_MAXV_L * 4)
AASM_DEV void kb_tnx16_wg(const KCtx &k, const WS &w) {             // workgroup per contig
    const int64_t c = k.bid;
    const int64_t V = w.ctgV[c];
    if (V == 0 || !in_graph_class(w, c) || in_chain_class(w, c)) return;
    int32_t *Lb = (int32_t *)k.lds;
    const int64_t vb = w.voff[c];
    for (int32_t v = k.tid; v < V; v += k.nthreads) Lb[v] = w.sp_best[vb + v];
    block_barrier();
    for (int32_t v = k.tid; v < V; v += k.nthreads) {
        I4 *o = (I4 *)(w.tnx16 + 16 * (vb + v));                     // the next sixteen vertices along best[] (-1 past the root)
        int32_t x = v;
        AASM_UNROLL
        for (int t = 0; t < 4; t++) {
            I4 j;
            x = x >= 0 ? Lb[x] : -1; j.x = x;
            x = x >= 0 ? Lb[x] : -1; j.y = x;
            x = x >= 0 ? Lb[x] : -1; j.z = x;
            x = x >= 0 ? Lb[x] : -1; j.w = x;
            o[t] = j;
        }
    }
}
// child list + sidetrack keys + header of a vertex, one thread, one launch (sparse batches)
AASM_DEV void kb_k7_prep(const KCtx &k, const WS &w) {              // thread per vertex
    const int64_t gv = k.bid * k.nthreads + k.tid;
    if (gv >= w.VT || in_chain_class(w, w.v_ctg[gv])) return;
    int32_t ch4[4];
    const int32_t nch = children_vertex(w, gv, ch4);
    const int32_t nkeys = sidetrack_vertex(w, gv);
    I4 a, b;
    heap_hdr_vertex(w, gv, a, b, nch, ch4, nkeys);                   // (the count and the first four children in registers: a fifth child and on come back from cval)
    w.vhdr[gv] = a; w.vhdr2[gv] = b;
}
AASM_DEV void kb_heap_hdr(const KCtx &k, const WS &w) {
    const int64_t gv = k.bid * k.nthreads + k.tid;
    if (gv >= w.VT || in_chain_class(w, w.v_ctg[gv])) return;
    I4 a, b;
    heap_hdr_vertex(w, gv, a, b);
    w.vhdr[gv] = a; w.vhdr2[gv] = b;
}

// arena capacity per contig: an insert into a heap of s nodes allocates at most
// floor(log2(s+1)) + 2 nodes (right-spine length + the new leaf), s < #sidetracks.
AASM_DEV void kb_heap_cap(const KCtx &k, const WS &w) {             // thread per contig
    const int64_t c = k.bid * k.nthreads + k.tid;
    if (c >= w.C) return;
    const int64_t V = w.ctgV[c];
    w.mw_flag[c] = 0; w.mw_cap[c] = 0; w.mw_lg[c] = 2;
    if (w.chain_flag) w.chain_flag[c] = 0;
    w.gb_flag[c] = 0;
    if (V == 0) { w.hcap_cnt[c] = 0; return; }
    const int64_t vb = w.voff[c];
    const int64_t E = w.rowptr[vb + V] - w.rowptr[vb];
    {   // one workgroup builds the graph of a small contig of a sparse batch (kb_graph_build); everybody else: the separate launches
        const bool sparse_batch = w.rowptr[w.VT] <= 6 * w.VT;
        const int32_t f = (w.gb_off || !sparse_batch) ? 0 : (V <= GB_MAXV && E <= GB_MAXE) ? 1 : (V <= GB_MAXV_L && E <= GB_MAXE_L) ? 2 : 0;
        w.gb_flag[c] = f;
        atomic_add(&w.counters[f == 1 ? CNT_GB_S : f == 2 ? CNT_GB_L : CNT_GB_REST], (int64_t)1);
    }
    if (E > (int64_t)INT32_MAX - 64) { w.status[c] = -5; w.hcap_cnt[c] = 0; return; }   // contig-local edge ids are int32 (AASM_E_OVERFLOW)  (not in the chain class either)
    int64_t I = E - (V - 1);
    if (I < 0) I = 0;
    int lg = 0;
    for (int64_t t = I + 1; t > 1; t >>= 1) lg++;
    int64_t cap = I * (lg + 2) + 8;
    if (cap > 0x7fffff00) cap = 0x7fffff00;                          // arena indices are int32: a contig that really needs more ends with status AASM_E_OVERFLOW (kb_heap)
    w.mw_lg[c] = lg + 2;
    // dense / high-multiplicity graphs (tens of sidetracks per vertex, a wide SP tree): several waves per contig
    w.mw_flag[c] = (w.mw_mode == 1 || (w.mw_mode == 0 && I >= 6 * V && V >= 128)) ? 1 : 0;
    w.mw_cap[c] = w.mw_flag[c] ? (int32_t)cap : 0;
    w.hcap_cnt[c] = (w.mw_flag[c] && !w.mw_compact) ? 0 : (int32_t)cap;   // (a several-waves contig whose nodes stay in the provisional arena needs no final one)
    if (w.mw_flag[c]) { const int64_t slot = atomic_add(&w.counters[CNT_MW], (int64_t)1); w.mw_list[slot] = (int32_t)c; w.mw_key[slot] = (int32_t)cap; }
    if (w.chain_flag) {
        // the chain class: sparse one-wave contigs whose sweep, pre-pass and heaps run beside each other (kb_chain) - every contig of
        // a small batch (the step is the chain of its slowest contig, not throughput) or the long tail of a big one
        const int64_t N = w.rec_off[c + 1] - w.rec_off[c];
        const bool sparse_batch = w.rowptr[w.VT] <= 6 * w.VT;
        const bool in = !w.mw_flag[c] && sparse_batch && w.status[c] == 0 &&
                        (w.chain_mode == 1 || (w.chain_mode == 0 && (w.chain_all || (w.chain_minN > 0 && N >= w.chain_minN))));
        w.chain_flag[c] = in ? 1 : 0;
        if (in) { const int64_t slot = atomic_add(&w.counters[CNT_CHAIN], (int64_t)1); w.chain_list[slot] = (int32_t)c; }
    }
}

// Cooperative K7.  One wave per contig walks the SP tree in BFS order (arena index == allocation order,
// hazard B3).  What the counters of the first version said: with ~20 one-wave workgroups per
// CU the kernel was bound by the ONE scalar unit a CU's four SIMDs share (2.8 G scalar against 1.1 G vector
// instructions), and per contig by chains of dependent memory round trips.  So this version
//  * keeps per-insert work on the vector unit: the right spine of the current heap lives in registers, node j
//    in lane j, with its score sum cached; the descent is two vector compares + a ballot, the rank chain
//    (leftist_heap.hpp:36-38, a min-plus recurrence) one suffix-min over the lanes by DPP row shifts, and the
//    new nodes are built by selects in all lanes at once; the scalar unit only steers;
//  * issues NO global store inside a vertex's work: new nodes go to an LDS ring (also the read cache of the
//    pointer chase) and reach the arena in one coalesced flush at the START of the next vertex, together with
//    the finished root of the vertex before.  vmcnt counts loads and stores in issue order, so a load waits
//    for every older store: with all stores issued ahead of the prefetch loads of a vertex, waiting for those
//    loads one vertex later costs nothing extra;
//  * fetches the next vertex's header and sidetrack keys one vertex ahead (kb_heap_hdr makes it predictable):
//    the loads are issued right after the step's stores and their registers are only touched again at the END
//    of the step, where the keys are parked in an LDS slot and the header words become scalars (readfirstlane).
//    (VECTOR load results carried to the next loop iteration in registers do not survive the compiler: it copies
//    them at the loop edge and waits for them on the spot - measured: 1 500-2 000 of 4 100 cycles per vertex.)
//   spine (registers)  80-96 % of the inserts continue from exactly the heap the last insert made, and an
//                 insert's descent path is a prefix of its right spine.  A rank swap at position t sends the
//                 new spine into an old left subtree: the cache then holds positions 0..t plus the node where
//                 it continues (tail); the walk past the cached prefix (or after a root switch) chases
//                 pointers and appends what it reads to the cache.
//   ring (LDS)    the newest HEAP_RING nodes: staging for the flush + read cache for that pointer chase
//   bq (LDS)      BFS queue window: {vertex, inherited heap root, key offset, #keys}; entries beyond the
//                 window spill to global memory (wide trees)
#define HEAP_RING 64                      // ring of the several-waves kernel (per wave); the flush unit everywhere (a flush = one node per lane)
#define HEAP_QN 128
#define HEAP_KMAX 16                      // sidetrack keys of a vertex that travel through the staging slot
struct HeapStage { Dist key[HEAP_KMAX]; };   // the first keys of one vertex
// What the walk past the cached spine prefix reads (heap_read) is a node of an EARLIER insert: from the ring while it is among the
// newest `rn` nodes, else from global memory.  Measured in round 5 with stamps inside the insert (1 000-record contigs): 1.28 such steps
// per insert - the walk is NOT the rare path earlier rounds took it for, it is a quarter to a third of K7's time - of which 22 % missed
// the 64-node ring (ages 64-127: 8 %, 128-511: 8 %, older: 6 %).  A bigger ring catches them (399 -> 238 -> 79 global steps per contig
// with 128 / 512 nodes) but buys only 3-5 % of the kernel: a step costs ~700 cycles either way (its ~60 dependent instructions, not its
// load).  The chain class has LDS to spare and takes 256 nodes (18 KB a workgroup; with 512 - 30 KB - only four of its
// three-wave workgroups fit a CU, and a batch of 1 280 contigs, five per CU, took a second residency round: 6.9 ms against 6.5 with the class off); the one-wave kernel
// stays at 64 + a 128-entry queue window: with 128 + 64 (7.7 KB, on paper 20 workgroups per CU as before) C3's launch no longer fit one
// residency round and took 5.5 ms instead of 4.06.
#define HEAP_OLDN 8
template <int RING, int QN> struct HeapLdsT {
    HNode ring[RING];
    I4 bq[QN];
    HeapStage stage;                      // keys of the vertex popped next, parked here at the end of a step
    HNode bounce;                         // a node chased in global memory passes through here (heap_read)
    HNode oldn[HEAP_OLDN];                // ... and stays here until the walk is over (heap_insert)
};
#define HEAP_RING_1W 64
#define HEAP_QN_1W 128
#define HEAP_RING_CH 256
#define HEAP_QN_CH 128
#define AASM_HEAP_LDS_BYTES_T(RING, QN) ((RING) * 48 + (QN) * 16 + HEAP_KMAX * 32 + 48 + HEAP_OLDN * 48)
#define AASM_HEAP_LDS_BYTES AASM_HEAP_LDS_BYTES_T(HEAP_RING_1W, HEAP_QN_1W)
static_assert(sizeof(HeapLdsT<HEAP_RING_1W, HEAP_QN_1W>) <= AASM_HEAP_LDS_BYTES && sizeof(HeapLdsT<HEAP_RING_CH, HEAP_QN_CH>) <= AASM_HEAP_LDS_BYTES_T(HEAP_RING_CH, HEAP_QN_CH), "LDS budget");
struct Spine {
    LaneArr<NodeQ> n;          // cached nodes, position j in lane j
    LaneArr<int32_t> idx;      // their arena indices
    LaneArr<int64_t> sum;      // key.qry_score + key.ref_score of node j (first key of the CALC_SUM order)
    LaneArr<int32_t> t;        // unwinding result of position j: rank word | swapped << 24
    int32_t root, len, tail;   // heap the cache describes; cached positions; node after them (-1: the spine ends)
};
struct HeapState {
    HNode *nodes;
    HNode *ring;                   // `rn` nodes of LDS, this wave's (rn: a power of two >= HEAP_RING)
    int32_t rn;
    HNode *bounce;                 // one node of LDS, this wave's (heap_read)
    HNode *oldn;                   // HEAP_OLDN nodes of LDS, this wave's (one-wave kernel): walked nodes that have left the ring wait here for the lanes that will cache them
    int32_t alloc, flushed, cap;   // arena: next index; nodes below `flushed` are in global memory, [flushed, alloc) only in the ring
    int32_t ring_lo;               // nodes below it were never in this wave's ring (0, or the start of the vertex region being filled)
    bool ovf;
};

// where K8 and K9 find a contig's heap nodes: the final arena, or - a contig built by several waves, outside debug runs - the
// provisional one: its regions lie in BFS order of the vertices and a region is filled in allocation order, so the order of the
// indices there IS the reference's allocation order (all that K8's tie-break asks of an index); only the indices have gaps
AASM_DEV const HNode *heap_arena(const WS &w, int64_t c) { return (w.mw_flag[c] && !w.mw_compact) ? w.hprov + w.mw_off[c] : w.hnodes + w.hoff[c]; }
AASM_DEV NodeQ heap_read(const HeapState &hs, int32_t a, HNode *slot = nullptr) {   // slot: where a node from global memory is parked (default: the bounce slot)
    NodeQ n;
    const HNode *src = &hs.ring[a & (hs.rn - 1)];                                  // ds_read, lgkmcnt only
    if (!slot) slot = hs.bounce;
    if (!(a >= hs.alloc - hs.rn && a >= hs.ring_lo)) {
        // an old node, in global memory (a few per cent of the chase steps): it goes to LDS first, so that what the caller keeps
        // in its spine registers has ONE kind of source - with a global load as the other one the compiler guards every later
        // use of the spine with vmcnt(0), and the inserts wait for the next vertex's prefetch instead of running beside it
        const NodeQ g = nodeq_load(&hs.nodes[a]);
        asm volatile("" ::: "memory");                                                 // keep it a global_load (no flat access)
        nodeq_store(slot, g);
        src = slot;
    }
    n = nodeq_load(src);
    return n;
}
// node.key < key (paf_data.hpp:142-159, CALC_SUM mode) for keys that are never max() (kb_sidetrack)
AASM_DEV bool nodeq_key_lt(const NodeQ &n, const Dist &key, int64_t ksum) {
    const Dist nk = nodeq_key(n);
    const int64_t nsum = nk.qry + nk.ref;
    if (nsum != ksum && nsum != -2 && ksum != -2) return nsum < ksum;                // neither side is max() (score sum -2)
    return dist_lt<CALC_SUM_MODE>(nk, key);
}
AASM_DEV bool key_tie_lt(const NodeQ &n, const Dist &key) {                          // ... the rest of the order when the sums are equal
    const int32_t na = n.q1.x, nn = n.q1.y, nt = n.q1.z ? n.q1.z : 1, kt = key.qtot ? key.qtot : 1;
    return (na < key.anom) | ((na == key.anom) & ((int64_t)nn * (int64_t)kt > (int64_t)key.qnz * (int64_t)nt));
}
// the arena receives the ring's unflushed nodes [flushed, alloc): consecutive 48-byte nodes, one per lane
AASM_DEV void heap_flush(HeapState &hs, int lane) {
    const int32_t m = hs.alloc - hs.flushed;
    if (m > 0) {
        wave_lds_sync();
        FOR_LANE(t, m, lane) { const int32_t a = hs.flushed + t; nodeq_store(&hs.nodes[a], nodeq_load(&hs.ring[a & (hs.rn - 1)])); }
        hs.flushed = hs.alloc;
    }
}

// One persistent insert (leftist_heap.hpp:29-40) into heap `hu`; returns the new root.
template <bool UNI_CHASE, class KP>
AASM_DEV int32_t heap_insert(HeapState &hs, Spine &sp, int32_t hu, const Dist key, int32_t eu, int32_t ev, int lane, KP &kp) {
    const int64_t ksum = key.qry + key.ref;
    if (hu != sp.root) { sp.root = hu; sp.len = 0; sp.tail = hu; }                   // root switch: nothing cached yet
    // ---- descent (:30): first spine position whose key is NOT < key
    int32_t depth = -1, a_stop = -1, a_rank = 0;
    if (sp.len > 0) {
        uint64_t lt = wave_index_mask(sp.len, lane, [&](int j) { return sp.sum.at(j) < ksum; });
        const uint64_t eq = wave_index_mask(sp.len, lane, [&](int j) { return sp.sum.at(j) == ksum; });
        if (eq) lt |= wave_index_mask(sp.len, lane, [&](int j) { return sp.sum.at(j) == ksum && key_tie_lt(sp.n.at(j), key); });
        const uint64_t stop = ~lt & lanemask_lt(sp.len);
        if (stop) { depth = ffs64(stop) - 1; a_stop = LA_GET(sp.idx, depth, ); a_rank = LA_GET(sp.n, depth, .q1.w) & 0xff; }
    }
    if (depth < 0 && UNI_CHASE) {                                                                // past the cached prefix: chase and extend the cache
        depth = sp.len;
        int32_t a = sp.tail;
        const int32_t d0 = depth;
        int32_t my_a = -1, my_slot = -1;                                            // lane j in [d0, depth): the node it will cache; its place in `oldn` when it has left the ring
        int32_t nold = 0;
        if (a >= 0) wave_lds_sync();                                                // ring writes of earlier inserts before the ring reads
        while (a >= 0) {
            // The walk itself is wave-uniform: every lane reads the node at `a` (one address: an LDS broadcast), the outcome and the
            // right pointer are the same in all of them.  The lane that will cache the node only notes where it is; the nodes come
            // into the spine registers after the walk, all lanes at once - the loop carries scalars, not the sixteen spine registers
            // (round 5: the nodes from global memory too - they wait in `oldn`; written into the spine inside the loop they gave it
            // a second register set, and every step of every walk paid twenty copies between the two).
            if (depth >= AASM_WAVE_MAX - 2) { hs.ovf = true; return -1; }
            const bool in_ring = a >= hs.alloc - hs.rn && a >= hs.ring_lo;
#if defined(AASM_HOST_EMUL)
            const NodeQ n = heap_read(hs, a);
#else
            const NodeQ n = heap_read(hs, a, (!in_ring && nold < HEAP_OLDN) ? &hs.oldn[nold] : nullptr);
#endif
#if defined(AASM_KPROF) && !defined(AASM_HOST_EMUL)
            if (lane == 0) kp.acc[in_ring ? 7 : 6] += 1;                            // diagnostic: chase steps served by the ring / by global memory
#endif
            const int64_t nsum = nodeq_key(n).qry + nodeq_key(n).ref;
#if defined(AASM_HOST_EMUL)
            FOR_LANE_EQ(j, depth, lane) { sp.n.at(j) = n; sp.idx.at(j) = a; sp.sum.at(j) = nsum; }
            (void)in_ring; (void)nold; (void)my_slot;
#else
            if (lane == depth) { my_a = a; my_slot = in_ring ? -1 : (nold < HEAP_OLDN ? nold : -2); }
            if (!in_ring) nold++;
#endif
            bool lt = wave_any(nsum < ksum);                                        // (wave-uniform values: any lane = every lane)
            if (!lt && wave_any(nsum == ksum)) lt = wave_any(key_tie_lt(n, key));             // (the tie order's two 32 x 32 -> 64 multiplies only when the sums are equal: the values are wave-uniform, the branch is scalar)
            if (!lt) { a_rank = uni(n.q1.w) & 0xff; break; }                        // the stop node (the new leaf takes its place in lane `depth`)
            a = uni(n.q2.y);                                                        // ->right
            depth++;
        }
        a_stop = a;
#if defined(AASM_HOST_EMUL)
        (void)d0; (void)my_a;
#else
        if (depth > d0 && my_a >= 0) {                                              // the walked nodes, all lanes at once: from the ring, or from where heap_read parked them
            const HNode *src = my_slot >= 0 ? &hs.oldn[my_slot] : &hs.ring[my_a & (hs.rn - 1)];
            NodeQ n = nodeq_load(src);
            if (my_slot == -2) { n = nodeq_load(&hs.nodes[my_a]); asm volatile("" ::: "memory"); }   // (more than HEAP_OLDN old nodes in one walk: never on the bench's graphs)
            sp.n.r = n; sp.idx.r = my_a; sp.sum.r = nodeq_key(n).qry + nodeq_key(n).ref;
        }
#endif
    }
    if (depth < 0 && !UNI_CHASE) {                                                  // the walk lane by lane (the several-waves kernel: dense heaps, where the uniform form measured 4 % slower)
        depth = sp.len;
        int32_t a = sp.tail;
        if (a >= 0) wave_lds_sync();                                                // ring writes of earlier inserts before the ring reads
        while (a >= 0) {
            // the chased node goes straight into the registers of the lane that will cache it (lane `depth`):
            // one lane loads, compares and keeps it; only the outcome and the right pointer become scalars
            if (depth >= AASM_WAVE_MAX - 2) { hs.ovf = true; return -1; }
            bool lt = false;
            FOR_LANE_EQ(j, depth, lane) {
                const NodeQ n = heap_read(hs, a);
#if defined(AASM_KPROF) && !defined(AASM_HOST_EMUL)
                kp.acc[(a >= hs.alloc - hs.rn && a >= hs.ring_lo) ? 7 : 6] += 1;                  // diagnostic: chase steps served by the ring / by global memory
#endif
                const int64_t nsum = nodeq_key(n).qry + nodeq_key(n).ref;
                sp.n.at(j) = n; sp.idx.at(j) = a; sp.sum.at(j) = nsum;
                lt = (nsum < ksum) | ((nsum == ksum) & key_tie_lt(n, key));
            }
            if (!wave_ballot(lt)) { a_rank = LA_GET(sp.n, depth, .q1.w) & 0xff; break; }   // the stop node (its copy in lane `depth` is overwritten by the new leaf)
            a = LA_GET(sp.n, depth, .q2.y);                                         // ->right
            depth++;
        }
        a_stop = a;
    }
    if (hs.alloc + depth + 1 > hs.cap) { hs.ovf = true; return -1; }
    if (hs.alloc + depth + 1 - hs.flushed > HEAP_RING) heap_flush(hs, lane);        // a flush moves at most one node per lane (and the new nodes must not overwrite unflushed ring slots)
    // ---- rank chain (:34-38).  Bottom-up the recursion computes, with R = rank of the subtree
    // below and R_depth = 1 (the new leaf, :31):  swap iff l == null or rank(l) < R;  rank of the
    // copy R_j = 0 if l == null (the copy then has no right child) else min(rank(l), R) + 1.
    // With a_j = 0 / rank(l_j) + 1 that is R_j = min(a_j, R_{j+1} + 1), a min-plus recurrence with
    // the closed form R_j = min_{i >= j}(a_i + i) - j (a_depth = 1): one suffix-min over the lanes.
    FOR_LANE(j, depth + 1, lane) {
        const NodeQ &nj = sp.n.at(j);
        const int32_t aj = (nj.q2.x < 0) ? 0 : (((nj.q1.w >> 8) & 0xff) + 1);
        sp.t.at(j) = ((j == depth) ? 1 : aj) + j;
    }
    lane_excl_suffix_min(sp.t, depth + 1, lane);                                    // t[j] = R_{j+1} + (j + 1)
    // ---- new nodes: leaf = alloc, copy of path position j = alloc + (depth - j)  (allocation order of the recursion)
    const int32_t alloc = hs.alloc, nalloc = alloc + depth + 1;
    uint64_t swm = 0;
#if defined(AASM_HOST_EMUL)
    for (int j = 0; j <= depth; j++)
#else
    const int j = lane;
    const bool in = j <= depth;
#endif
    {
        const bool leaf = (j == depth);
        const NodeQ old = sp.n.at(j);
        const int32_t l = old.q2.x, l_rank = (old.q1.w >> 8) & 0xff, r_rank = sp.t.at(j) - (j + 1);
        const bool sw = (l < 0) | (l_rank < r_rank);                                // :36-37
        const bool has_right = sw ? (l >= 0) : true;
        const int32_t nr_rank = has_right ? (sw ? l_rank : r_rank) : 0;
        const int32_t nrank = has_right ? nr_rank + 1 : 0;                          // :38
        const int32_t ni = alloc + (depth - j), below = ni - 1;
        NodeQ n;
        n.q0.x = leaf ? (int32_t)(uint32_t)(uint64_t)key.qry : old.q0.x; n.q0.y = leaf ? (int32_t)((uint64_t)key.qry >> 32) : old.q0.y;
        n.q0.z = leaf ? (int32_t)(uint32_t)(uint64_t)key.ref : old.q0.z; n.q0.w = leaf ? (int32_t)((uint64_t)key.ref >> 32) : old.q0.w;
        n.q1.x = leaf ? key.anom : old.q1.x; n.q1.y = leaf ? key.qnz : old.q1.y; n.q1.z = leaf ? key.qtot : old.q1.z;
        n.q1.w = leaf ? (1 | (a_rank << 8)) : (nrank | ((sw ? r_rank : l_rank) << 8) | (nr_rank << 16));   // :31-32: (1, k, v, left = a, right = null)
        n.q2.x = leaf ? a_stop : (sw ? below : l); n.q2.y = leaf ? -1 : (sw ? l : below);
        n.q2.z = leaf ? eu : old.q2.z; n.q2.w = leaf ? ev : old.q2.w;
#if defined(AASM_HOST_EMUL)
        nodeq_store(&hs.ring[ni & (hs.rn - 1)], n);
        sp.n.at(j) = n; sp.idx.at(j) = ni; if (leaf) sp.sum.at(j) = ksum;
        if (!leaf && sw) swm |= 1ull << j;
#else
        if (in) {
            nodeq_store(&hs.ring[ni & (hs.rn - 1)], n);
            sp.n.at(j) = n; sp.idx.at(j) = ni; if (leaf) sp.sum.at(j) = ksum;       // position j of the NEW spine (valid up to the first swap)
        }
        swm = wave_ballot(in && !leaf && sw);
#endif
    }
    // ---- spine of the new heap: new nodes down to the first swapped position, then its old left subtree
    if (swm) { const int t = ffs64(swm) - 1; sp.len = t + 1; sp.tail = LA_GET(sp.n, t, .q2.y); }
    else { sp.len = depth + 1; sp.tail = -1; }                                      // ... or all of them and the leaf (right == null)
    sp.root = alloc + depth;
    hs.alloc = nalloc;
    return sp.root;
}

struct KProfNone { int64_t acc[8]; };   // (acc: only touched by the -DAASM_KPROF diagnostic build)
// kb_chain's heap wave: the header of vertex u is in memory (the prep wave writes the two marker words last, after everything a
// step reads through them); false: the prep wave is gone and the header never came, or 30 s passed (AASM_E_INTERNAL, not a hang)
AASM_DEV bool chain_wait_hdr(ChainSync *S, const I4 *vh, const I4 *vh2, int32_t u, int64_t patience) {
    const int64_t t0 = wave_realtime();
    int64_t guard = 0;
    for (;;) {
        const int32_t pd = uni(ld_shared_i32(&S->prep_done));       // (read before the words: a header written before the flag is seen)
        const int32_t x = uni(ld_shared_i32(&vh[u].x)), y = uni(ld_shared_i32(&vh2[u].y));
        if (x != -1 && y != -1) { wave_fence(); return true; }
        if (pd) return false;
        if ((++guard & 1023) == 0 && wave_realtime() - t0 > patience) return false;
        wave_sleep();
    }
}
template <bool CHAIN = false, int RING = HEAP_RING_1W, int QN = HEAP_QN_1W>
AASM_DEV void kb_heap(const KCtx &k, const WS &w, ChainSync *S = nullptr) {   // one wave per contig
    const int64_t c = k.bid;
    const int64_t V = w.ctgV[c];
    if (V == 0) return;
    if (!CHAIN && in_chain_class(w, c)) return;                      // (built inside kb_chain)
    HeapLdsT<RING, QN> *L = (HeapLdsT<RING, QN> *)k.lds;
    const int64_t vb = w.voff[c];
    int32_t *h = w.h_root + vb, *q = w.bq + vb;
    const I4 *vh = w.vhdr + vb, *vh2 = w.vhdr2 + vb;
    const Dist *sk = w.st_cost + w.rowptr[vb];                      // the contig's compacted sidetrack keys (kb_sidetrack)
    HeapState hs;
    hs.nodes = w.hnodes + w.hoff[c]; hs.ring = L->ring; hs.rn = RING; hs.bounce = &L->bounce; hs.oldn = L->oldn; hs.alloc = 0; hs.flushed = 0; hs.ring_lo = 0; hs.ovf = false;
    hs.cap = (int32_t)(w.hoff[c + 1] - w.hoff[c]);
    const int32_t src = (int32_t)(V - 2), dest = (int32_t)(V - 1);
    if (w.mw_flag[c]) return;                                        // wide trees: kb_heap_mw
#if !defined(AASM_HOST_EMUL)
    {   // The launch ends with its longest contig while thousands of waves share the issue slots: contigs with more
        // inserts than the batch's mean get a higher wave priority, so the tail starts ahead instead of last.
        const int64_t I = (w.rowptr[vb + V] - w.rowptr[vb]) - (V - 1);
        const int64_t a = w.avg_sidetracks > 0 ? w.avg_sidetracks : 1;
        if (8 * I > 9 * a) __builtin_amdgcn_s_setprio(3);
        else if (I > a) __builtin_amdgcn_s_setprio(2);
        else if (8 * I > 7 * a) __builtin_amdgcn_s_setprio(1);
    }
#endif
    if (k.lane == 0) w.h_cnt[c] = 0;
    if (w.status[c] != 0) return;
    if (!CHAIN && dist_is_max(w.sp_d[vb + src])) { if (k.lane == 0) set_status(w, c, -6); return; }   // :188-189: no path (must not happen)  (CHAIN: the sweep is still running - checked at the end)
    bool chain_lost = false;                                         // CHAIN: a header never came
    const int64_t patience = (CHAIN && w.chain_test == 2) ? (int64_t)100000000 : (int64_t)30 * 100000000;   // 100 MHz ticks: 30 s (test hook: 1 s)
    int32_t head = 0, tail = 0, lds_hi = 0;                          // BFS queue positions; [head, lds_hi) live in the LDS window
    Spine sp; sp.root = -2; sp.len = 0; sp.tail = -1;
#if defined(AASM_KPROF) && !defined(AASM_HOST_EMUL)
    KPROF_DECL;
    KPROF_START();
    struct KP { int64_t &t0; int64_t *acc; } kp{kp_t0, kp_acc};
#else
    KProfNone kp;
#endif
    int32_t u = dest, hu = -1, so = 0, n = 0;                        // the vertex at hand: id, inherited heap, key offset, #keys
    if (CHAIN && !chain_wait_hdr(S, vh, vh2, dest, patience)) chain_lost = true;
    else {
        const I4 a0 = vh[dest];
        so = uni(a0.x); n = uni(a0.y);
    }
    wave_lds_sync();
    int32_t pend_u = -1, pend_root = -1;                             // finished root of the previous vertex (lane 0), stored at the start of the next step
    int32_t staged = -1;                                             // the vertex whose keys are parked in L->stage and whose header words are in c_* (-1: none)
    int32_t c_nch = 0, c_fc = -1, c_sofc = 0, c_nfc = 0, c_c0lo = 0, c_c0hi = 0, c_n = 0;   // {#children, first child, its key offset / row length, child-list start}, the staged vertex's own #keys
    while (!hs.ovf && !chain_lost) {
        HeapStage *St = &L->stage;
        if (staged != u) {                                           // not staged (the root, a spilled queue entry; CHAIN: a header that was not there yet when it was prefetched): fetch now
            if (CHAIN && !chain_wait_hdr(S, vh, vh2, u, patience)) { chain_lost = true; break; }
            const I4 ha = vh[u], hb = vh2[u];
            so = uni(ha.x); c_n = uni(ha.y);                         // (what the parent handed over is the row's start and its LENGTH)
            FOR_LANE(t, (c_n < HEAP_KMAX ? c_n : HEAP_KMAX), k.lane) { const Dist kk = sk[(int64_t)so + t]; St->key[t] = kk; }   // (rare path: no need to overlap it with the header reads)
            c_nch = uni(ha.z); c_fc = uni(ha.w); c_sofc = uni(hb.z); c_nfc = uni(hb.w); c_c0lo = uni(hb.x); c_c0hi = uni(hb.y);
            wave_lds_sync();
        }
        n = c_n;                                                     // the vertex's key count: from its own header (staged with its keys, or just read)
        const int32_t nch = c_nch, fc = c_fc, so_fc = c_sofc, n_fc = c_nfc, c0lo = c_c0lo, c0hi = c_c0hi;
        KPROF_STAMP(0);                                              // context of this vertex
        // ---- every global store of this step, ahead of its loads: the root of the vertex before, the staged nodes
        if (pend_u >= 0) h[pend_u] = pend_root;                      // (lane 0 only: the pair lives in its vector registers)
        if (hs.alloc - hs.flushed >= HEAP_RING / 2) heap_flush(hs, k.lane);
        KPROF_STAMP(1);                                              // stores
        // ---- the vertex after this one: the queue front, or - the queue is empty, the tree path-like - the first child
        int32_t v2 = -1, so2 = 0, n2 = 0, hu2 = -1;
        bool peeked = false;                                         // the queue front has been read (the pop below takes it from here)
        if (head < tail) { if (head < lds_hi) { const I4 e = L->bq[head & (QN - 1)]; v2 = uni(e.x); hu2 = uni(e.y); so2 = uni(e.z); n2 = uni(e.w); peeked = true; } }
        else if (nch > 0) { v2 = fc; so2 = so_fc; n2 = n_fc; }
        I4 pa, pb;
        LaneArr<Dist> pk;
        pa.x = pa.y = pa.z = pa.w = 0; pb = pa;
#if !defined(AASM_HOST_EMUL)
        pk.r = dist_zero();
#endif
        if (v2 >= 0) {
            pa = vh[v2]; pb = vh2[v2];
            FOR_LANE(t, (n2 < HEAP_KMAX ? n2 : HEAP_KMAX), k.lane) pk.at(t) = sk[(int64_t)so2 + t];
        }
        // ... and this vertex's child slots, when its children will go through the queue (they are written into the window AFTER the
        // inserts: loaded there, the round trip was the exposed part of the children section - 10-14 % of the kernel)
        I4 cpre;
        cpre.x = cpre.y = cpre.z = cpre.w = 0;
        const int64_t c0 = (int64_t)(((uint64_t)(uint32_t)c0hi << 32) | (uint32_t)c0lo);
        if (nch > 0 && !(nch == 1 && head == tail) && k.lane < nch) cpre = w.cinfo[c0 + k.lane];
        KPROF_STAMP(2);                                              // prefetch issue
        // ---- inserts in list order (:202-211)
        for (int32_t t = 0; t < n && !hs.ovf; t++) {
            if (t >= HEAP_KMAX && (t & (HEAP_KMAX - 1)) == 0) {      // a row with more sidetracks than the slot holds (dense graphs): the next HEAP_KMAX keys replace the used ones.
                wave_lds_sync();                                     // The insert below must only ever see a key that came out of LDS: with a global load as the other source of
                FOR_LANE(j, (n - t < HEAP_KMAX ? n - t : HEAP_KMAX), k.lane) { const Dist kk = sk[(int64_t)so + t + j]; St->key[j] = kk; }   // `cc` the compiler makes EVERY insert wait for vmcnt(0),
                wave_lds_sync();                                     // i.e. for the next vertex's prefetch issued a moment earlier
            }
            const Dist cc = St->key[t & (HEAP_KMAX - 1)];
            hu = heap_insert<true>(hs, sp, hu, cc, u, cc.pad, k.lane, kp);
        }
        KPROF_STAMP(3);                                              // inserts
        pend_u = (k.lane == 0) ? u : -1; pend_root = hu;
        // ---- park the prefetched context (its loads had the whole step; this vertex's keys are used up)
        bool v2_ok = v2 >= 0;
        if (CHAIN && v2_ok && (uni(pa.x) == -1 || uni(pb.y) == -1)) v2_ok = false;   // its header was not written yet: the step that takes it waits (staged != u)
        if (v2_ok) {                                                 // (header words: straight into scalars - the loads are back by now)
            c_nch = uni(pa.z); c_fc = uni(pa.w); c_sofc = uni(pb.z); c_nfc = uni(pb.w); c_c0lo = uni(pb.x); c_c0hi = uni(pb.y); c_n = uni(pa.y);
            FOR_LANE(t, (n2 < HEAP_KMAX ? n2 : HEAP_KMAX), k.lane) St->key[t] = pk.at(t);   // (n2 = the row's length: slots past the key count hold nothing anybody reads)
        }
        staged = v2_ok ? v2 : -1;
        // ---- children adopt the heap (:213)
        if (nch == 1 && head == tail) {                              // path-like tree: the only child is next, no queue traffic
            u = fc; so = so_fc; n = n_fc;
            wave_lds_sync();
            continue;
        }
        if (nch > 0) {                                               // they enter the LDS queue window while it has room
            int32_t ncache = 0;
            if (lds_hi == tail) { ncache = QN - (tail - head); if (ncache > nch) ncache = nch; if (ncache < 0) ncache = 0; }
            for (int32_t t = k.lane; t < nch; t += AASM_WAVE) {
                I4 ci = cpre;                                        // (the first 64: fetched ahead of the inserts)
                if (t >= AASM_WAVE) ci = w.cinfo[c0 + t];
                if (t < ncache) { I4 e; e.x = ci.x; e.y = hu; e.z = ci.y; e.w = ci.z; L->bq[(tail + t) & (QN - 1)] = e; }
                else { h[ci.x] = hu; q[tail + t] = ci.x; }           // window full: spill (wide trees)
            }
            lds_hi += ncache;
            tail += nch;
        }
        wave_lds_sync();
        KPROF_STAMP(4);                                              // children
        // ---- pop
        if (head >= tail) break;
        if (peeked) { u = v2; hu = hu2; so = so2; n = n2; }
        else if (head < lds_hi) { const I4 e = L->bq[head & (QN - 1)]; u = uni(e.x); hu = uni(e.y); so = uni(e.z); n = uni(e.w); }
        else {                                                       // spilled entry (the window was full when it was pushed)
            wave_fence();
            u = uni(q[head]); hu = uni(h[u]);
            const I4 a0 = vh[u]; so = uni(a0.x); n = uni(a0.y);
        }
        head++;
        if (head >= lds_hi && head == tail) lds_hi = tail;           // spill drained: new entries go to the window again
        KPROF_STAMP(5);                                              // pop
    }
    if (pend_u >= 0) h[pend_u] = pend_root;
    heap_flush(hs, k.lane);
    KPROF_FLUSH(w.prof_heap, c, k.lane);
    if (CHAIN) {
        // :188-189 (no src -> dest path: must not happen) can only be asked once the sweep has finished
        const int64_t t0 = wave_realtime();
        int64_t guard = 0;
        while (!chain_lost && !uni(ld_shared_i32(&S->sweep_done))) {
            if ((++guard & 1023) == 0 && wave_realtime() - t0 > patience) chain_lost = true;
            wave_sleep();
        }
        wave_fence();
        if (chain_lost || uni(ld_shared_i32(&w.sp_d[vb + src].anom)) < 0) { if (k.lane == 0) { set_status(w, c, -6); w.h_cnt[c] = 0; } return; }   // (max() is the only distance with a negative anom)
    }
    if (hs.ovf) { if (k.lane == 0) set_status(w, c, -5); return; }
    if (k.lane == 0) { w.h_cnt[c] = hs.alloc; atomic_add(&w.counters[CNT_HEAPNODES], (int64_t)hs.alloc); }
}

// ====================================================================================
// The chain class: K6's sweep, K7's pre-pass and K7's heaps of ONE contig beside each other (kb_chain)
// ====================================================================================
// K6 -> K7 pre-pass -> K7 were three launches, each one wave's dependent chain per contig, each ending on the launch's longest
// contig: their times ADD (one 1 000-record contig: 0.98 + 2.18 ms; the 8 057-record contig of a heavy-tailed file: 11.3 + 21.6 ms
// while the rest of the chip idles).  Nothing in the data says they have to: a vertex's distance and tree edge are final the
// moment the sweep APPENDS it to its queue (every out-edge relaxed: k_shortest_walks.hpp:160-175), its sidetrack keys need only
// that and the distances of its heads (final earlier), and the heap of a vertex needs its parent's heap and its own keys
// (:196-215).  So a workgroup takes a contig, a wave per role (four since the BFS order has a wave of its own: chain_order, below;
// classes of more than AASM_CHAIN_ORD_MAX contigs keep the three-role form, where wave 2 is the heap wave with its own queue):
//   wave 0  the reverse sweep as before (kb_rev_sweep), which now also publishes how many vertices of rev_order are final;
//   wave 1  the pre-pass, lanes over vertices, behind it: for the vertices that became final - their sidetrack keys (what
//           kb_sidetrack does), then one count-down per out-edge on the head's `pend`; a vertex whose in-neighbours ALL have their
//           keys gets its child list and header (kb_children + kb_heap_hdr: the children of u are in-neighbours of u) - these
//           go through a work list (cq) and are published by the header's two marker words turning from -1;
//   wave 2  the BFS order of the tree: one record {vertex, parent's position, keys} per position (chain_order); it is the one that
//           waits for headers;
//   wave 3  the heaps in that order (kb_heap_ord: kb_heap's inserts, same arena, same indices) - or, three-role form, wave 2 =
//           kb_heap itself, which waits for a header only where the one-launch form could assume it.
// The graphs are local (edges join neighbouring parts, paf_data.cpp:599-694), so a vertex's in-neighbours are final soon after
// it: the heap wave - the slowest of the three - finds what it needs and the contig costs max(K6, K7) instead of the sum.
// Every wait ends: the sweep waits for nobody; the pre-pass only for the sweep (`sweep_done`); the order wave only for the
// pre-pass (`prep_done`) and the heap wave only for the order wave (`ord_done`) - each gives up with AASM_E_INTERNAL after 30 s
// rather than sit.
#define AASM_CHAIN_LDS_BYTES (AASM_REV_LDS_BYTES + AASM_HEAP_LDS_BYTES_T(HEAP_RING_CH, HEAP_QN_CH) + 32)
#define CHAIN_WAVES 4
#define CHAIN_LONG_ROW 8
struct ChainLds { RevQ rq; HeapLdsT<HEAP_RING_CH, HEAP_QN_CH> hl; ChainSync s; };
static_assert(sizeof(ChainLds) <= AASM_CHAIN_LDS_BYTES, "LDS budget");

AASM_DEV void chain_prep(const KCtx &k, const WS &w, int64_t c, ChainSync *S) {
    const int64_t vb = w.voff[c];
    const int32_t *q = w.rev_order + vb;
    int32_t *pend = w.pend + vb, *cq = w.cq + vb;
    int32_t done1 = 0, r2_head = 0, r2_tail = 0;                     // final vertices keyed so far; work list [r2_head, r2_tail)
    const int64_t t_start = wave_realtime();
    int64_t guard = 0;
    int32_t result = 1;
    for (;;) {
        const int32_t sd = uni(ld_shared_i32(&S->sweep_done));      // (before the count: once it is set the count is the last one)
        const int32_t t = uni(ld_shared_i32(&S->tail_pub));
        wave_fence();
        bool did = false;
        if (done1 < t) {
            // ---- keys of up to 64 newly final vertices, a lane each; then the count-downs along their out-edges
            const int32_t n1 = (t - done1 < AASM_WAVE) ? (t - done1) : AASM_WAVE;
            int32_t v = -1, deg = 0;
            int64_t r0 = 0;
            bool self = false;
            if (k.lane < n1) {
                v = q[done1 + k.lane];
                sidetrack_vertex(w, vb + v);
                tnx_vertex(w, vb + v);
                r0 = w.rowptr[vb + v]; deg = (int32_t)(w.rowptr[vb + v + 1] - r0);
                self = w.rptr[vb + v + 1] == w.rptr[vb + v];        // no in-neighbour (src): nothing to wait for
            }
            wave_fence();                                            // the keys are in memory before anybody is told
            {
                const uint64_t m = wave_ballot(self);
                if (self) cq[r2_tail + popc64(m & lanemask_lt(k.lane))] = v;
                r2_tail += popc64(m);
            }
            const bool longrow = deg > CHAIN_LONG_ROW;
            for (int32_t e = 0; wave_any(!longrow && e < deg); e++) {   // short rows: the lane walks its own row
                bool hit = false;
                int32_t x = -1;
                if (!longrow && e < deg) { x = w.e_col[r0 + e]; hit = atomic_add(&pend[x], (int32_t)-1) == 1; }
                const uint64_t m = wave_ballot(hit);
                if (hit) cq[r2_tail + popc64(m & lanemask_lt(k.lane))] = x;
                r2_tail += popc64(m);
            }
            uint64_t lm = wave_ballot(longrow);                      // long rows (src's, a wide part's): all lanes over the row
            while (lm) {
                const int sl = ffs64(lm) - 1;
                lm &= lm - 1;
                const int64_t r02 = wave_bcast(r0, sl);
                const int32_t deg2 = wave_bcast(deg, sl);
                for (int32_t e0 = 0; e0 < deg2; e0 += AASM_WAVE) {
                    const int32_t e = e0 + k.lane;
                    bool hit = false;
                    int32_t x = -1;
                    if (e < deg2) { x = w.e_col[r02 + e]; hit = atomic_add(&pend[x], (int32_t)-1) == 1; }
                    const uint64_t m = wave_ballot(hit);
                    if (hit) cq[r2_tail + popc64(m & lanemask_lt(k.lane))] = x;
                    r2_tail += popc64(m);
                }
            }
            done1 += n1;
            did = true;
            wave_fence();
        }
        if (r2_head < r2_tail) {
            // ---- child lists + headers of up to 64 vertices whose in-neighbours all have their keys
            const int32_t n2 = (r2_tail - r2_head < AASM_WAVE) ? (r2_tail - r2_head) : AASM_WAVE;
            int32_t u = -1, nch = 0, ch4[4] = {-1, -1, -1, -1};
            if (k.lane < n2) { u = cq[r2_head + k.lane]; nch = children_vertex(w, vb + u, ch4); }
            wave_fence();                                            // (the header pass reads a child list of more than four back)
            I4 a, b;
            a.x = a.y = a.z = a.w = 0; b = a;
            if (k.lane < n2) { heap_hdr_vertex(w, vb + u, a, b, nch, ch4, -1); tnx16_vertex(w, vb + u); }
            wave_fence();                                            // child slots, jump records: in memory before the marker words
            if (k.lane < n2 && !(w.chain_test && c == 0 && u == (int32_t)w.ctgV[c] - 1)) { w.vhdr2[vb + u] = b; w.vhdr[vb + u] = a; }   // (test hook: contig 0's root never gets its header)
            r2_head += n2;
            did = true;
            wave_fence();
        }
        if (!did) {
            if (sd && done1 >= t && r2_head == r2_tail) break;
            if ((++guard & 1023) == 0 && wave_realtime() - t_start > (int64_t)60 * 100000000) { result = 2; break; }   // (never: the sweep always ends)
            wave_sleep();
        }
    }
    store_drain();
    if (!(w.chain_test == 2 && c == 0)) st_shared_i32(&S->prep_done, result);   // (test hook 2: the heap wave of contig 0 is left to its own patience)
}

// The BFS order of a contig's SP tree by a wave of its own (kb_chain, wave 2).  A quarter of the heap wave's time per vertex was not
// inserts: waiting for the vertex's header, decoding it, fetching the child slots, pushing the children into its queue window, popping,
// the spill path of wide frontiers.  None of that needs the heaps - the order the reference allocates in is the FIFO order of the tree
// (k_shortest_walks.hpp:196-215: pop u, insert its keys, push its children in list order), and the tree is known as soon as the headers
// are.  So this wave walks the tree, up to 64 queue entries a step (their headers ready as a prefix, their children appended by a prefix
// sum of the child counts - the same order as one by one), and leaves ONE record per position: {vertex, position of its parent, start of
// its keys, #keys}.  The array is its own queue (head = next to expand, tail = next to write); `ord_pub` tells the heap wave how many
// positions are complete.  The heap wave (kb_heap_ord) reads records, keys and its parent's root - nothing else.
AASM_DEV void chain_order(const KCtx &k, const WS &w, int64_t c, ChainSync *S) {
    const int64_t V = w.ctgV[c], vb = w.voff[c];
    int32_t result = 1;
    if (V > 0 && w.status[c] == 0 && !w.mw_flag[c]) {
        I4 *q = w.bfsq + vb;
        const I4 *vh = w.vhdr + vb, *vh2 = w.vhdr2 + vb;
        const int64_t patience = (w.chain_test == 2) ? (int64_t)100000000 : (int64_t)30 * 100000000;   // as the heap wave's
        int32_t head = 0, tail = 1;
        if (k.lane == 0) { I4 r; r.x = (int32_t)(V - 1); r.y = -1; r.z = 0; r.w = 0; q[0] = r; }       // the root: dest
        wave_fence();
        while (head < tail) {
            const int32_t nb = (tail - head < AASM_WAVE) ? (tail - head) : AASM_WAVE;
            I4 rec; rec.x = -1; rec.y = -1; rec.z = 0; rec.w = 0;
            bool rdy = false;
            wave_fence();                                            // (entries other lanes wrote in the step before)
            if (k.lane < nb) {
                rec = q[head + k.lane];
                rdy = ld_shared_i32(&vh[rec.x].x) != -1 && ld_shared_i32(&vh2[rec.x].y) != -1;
            }
            const uint64_t rm = wave_ballot(rdy);
            int32_t r = (~rm == 0ull) ? AASM_WAVE : ffs64(~rm) - 1;  // the entries whose headers are there, as a prefix
            if (r == 0) {                                            // not even the first: wait for it (or learn that it will never come)
                if (!chain_wait_hdr(S, vh, vh2, uni(rec.x), patience)) { result = 2; break; }
                r = 1;
            }
            wave_fence();
            int32_t nch = 0;
            int64_t c0 = 0;
            if (k.lane < r) {
                const I4 ha = vh[rec.x], hb = vh2[rec.x];
                rec.z = ha.x; rec.w = ha.y; nch = ha.z;              // where its keys start, how many there are
                c0 = (int64_t)(((uint64_t)(uint32_t)hb.y << 32) | (uint32_t)hb.x);
            }
            const int32_t incl = wave_incl_add(nch);
            const int32_t tot = wave_bcast(incl, AASM_WAVE - 1);
            if (k.lane < r) {
                q[head + k.lane] = rec;
                int32_t at = tail + incl - nch;
                for (int32_t t = 0; t < nch; t++) {                  // (one or two children as a rule)
                    const I4 ci = w.cinfo[c0 + t];
                    I4 ch; ch.x = ci.x; ch.y = head + k.lane; ch.z = ci.y; ch.w = 0;
                    q[at + t] = ch;
                }
            }
            head += r; tail += tot;
            store_drain();                                           // the records are in memory before the count that announces them
            st_shared_i32(&S->ord_pub, head);
        }
    }
    store_drain();
    st_shared_i32(&S->ord_done, result);
}

// kb_chain's heap wave when the order comes from chain_order: for position i of the BFS order - the inherited heap is its parent's
// finished root (a ring of the newest roots in LDS, else h_root), its keys are a contiguous run - the inserts of kb_heap, same arena,
// same indices.  Record i + 2 and the keys of position i + 1 are fetched while position i's inserts run (parked at the step's end, as
// kb_heap does with its prefetch).
template <int RING, int QN>
AASM_DEV void kb_heap_ord(const KCtx &k, const WS &w, ChainSync *S) {
    const int64_t c = k.bid;
    const int64_t V = w.ctgV[c];
    if (V == 0) return;
    HeapLdsT<RING, QN> *L = (HeapLdsT<RING, QN> *)k.lds;
    const int64_t vb = w.voff[c];
    int32_t *h = w.h_root + vb;
    const Dist *sk = w.st_cost + w.rowptr[vb];
    const I4 *q = w.bfsq + vb;
    int32_t *rroot = (int32_t *)L->bq;                               // roots of the newest RN positions
    const int32_t RN = w.chain_rn > 0 ? w.chain_rn : QN * 4;         // (a power of two)
    HeapState hs;
    hs.nodes = w.hnodes + w.hoff[c]; hs.ring = L->ring; hs.rn = RING; hs.bounce = &L->bounce; hs.oldn = L->oldn; hs.alloc = 0; hs.flushed = 0; hs.ring_lo = 0; hs.ovf = false;
    hs.cap = (int32_t)(w.hoff[c + 1] - w.hoff[c]);
    const int32_t src = (int32_t)(V - 2);
    if (w.mw_flag[c]) return;
    if (k.lane == 0) w.h_cnt[c] = 0;
    if (w.status[c] != 0) return;
    bool lost = false;
    const int64_t patience = (w.chain_test == 2) ? (int64_t)100000000 : (int64_t)30 * 100000000;
    Spine sp; sp.root = -2; sp.len = 0; sp.tail = -1;
    KProfNone kp;
    HeapStage *St = &L->stage;
    int32_t i = 0;
    int32_t u = -1, p = -1, so = 0, n = 0;                           // position i: vertex, parent's position, key offset, #keys (keys in St)
    bool have = false;
    int32_t a_u = -1, a_p = -1, a_so = 0, a_n = 0;                   // position i + 1
    bool a_have = false;
    int32_t pend_u = -1, pend_root = -1;
    int32_t pub = 0;
    while (!hs.ovf) {
        if (!have) {                                                 // nothing staged (the start; the order wave was not ahead): wait for position i, fetch it now
            const int64_t t0 = wave_realtime();
            int64_t guard = 0;
            bool end = false;
            for (;;) {
                const int32_t od = uni(ld_shared_i32(&S->ord_done));    // (before the count: once it is set the count is the last one)
                pub = uni(ld_shared_i32(&S->ord_pub));
                if (i < pub) break;
                if (od) { end = true; lost = od == 2; break; }
                if ((++guard & 1023) == 0 && wave_realtime() - t0 > patience) { end = true; lost = true; break; }
                wave_sleep();
            }
            if (end) break;
            wave_fence();
            const I4 r0 = q[i];
            u = uni(r0.x); p = uni(r0.y); so = uni(r0.z); n = uni(r0.w);
            wave_lds_sync();
            FOR_LANE(t, (n < HEAP_KMAX ? n : HEAP_KMAX), k.lane) { const Dist kk = sk[(int64_t)so + t]; St->key[t] = kk; }
            wave_lds_sync();
            have = true; a_have = false;
        }
        // ---- the inherited heap: the parent's finished root
        int32_t hu = -1;
        if (p >= 0) {
            if (i - p <= RN) hu = uni(rroot[p & (RN - 1)]);
            else { wave_fence(); hu = uni(h[uni(q[p].x)]); }         // (a frontier wider than the ring: the root is in memory - stored two steps after it was made at the latest)
        }
        // ---- every global store of this step ahead of its loads (kb_heap)
        if (pend_u >= 0) h[pend_u] = pend_root;
        if (hs.alloc - hs.flushed >= HEAP_RING / 2) heap_flush(hs, k.lane);
        // ---- position i + 1 (if the order wave is not ahead this is a wait of its own), then its keys and record i + 2 behind the inserts
        pub = uni(ld_shared_i32(&S->ord_pub));
        if (!a_have && i + 1 < pub) {
            wave_fence();
            const I4 r1 = q[i + 1];
            a_u = uni(r1.x); a_p = uni(r1.y); a_so = uni(r1.z); a_n = uni(r1.w);
            a_have = true;
        }
        LaneArr<Dist> pk;
        I4 rb;
        rb.x = rb.y = rb.z = rb.w = 0;
#if !defined(AASM_HOST_EMUL)
        pk.r = dist_zero();
#endif
        const bool b_have = a_have && i + 2 < pub;
        if (a_have) { FOR_LANE(t, (a_n < HEAP_KMAX ? a_n : HEAP_KMAX), k.lane) pk.at(t) = sk[(int64_t)a_so + t]; }
        if (b_have) rb = q[i + 2];
        // ---- inserts in list order (:202-211)
        for (int32_t t = 0; t < n && !hs.ovf; t++) {
            if (t >= HEAP_KMAX && (t & (HEAP_KMAX - 1)) == 0) {
                wave_lds_sync();
                FOR_LANE(j, (n - t < HEAP_KMAX ? n - t : HEAP_KMAX), k.lane) { const Dist kk = sk[(int64_t)so + t + j]; St->key[j] = kk; }
                wave_lds_sync();
            }
            const Dist cc = St->key[t & (HEAP_KMAX - 1)];
            hu = heap_insert<true>(hs, sp, hu, cc, u, cc.pad, k.lane, kp);
        }
        pend_u = (k.lane == 0) ? u : -1; pend_root = hu;
        wave_lds_sync();
        if (k.lane == 0) rroot[i & (RN - 1)] = hu;
        // ---- park what was fetched: position i + 1 becomes the one at hand
        if (a_have) { FOR_LANE(t, (a_n < HEAP_KMAX ? a_n : HEAP_KMAX), k.lane) St->key[t] = pk.at(t); }
        wave_lds_sync();
        have = a_have; u = a_u; p = a_p; so = a_so; n = a_n;
        a_have = b_have;
        if (b_have) { a_u = uni(rb.x); a_p = uni(rb.y); a_so = uni(rb.z); a_n = uni(rb.w); }
        i++;
    }
    if (pend_u >= 0) h[pend_u] = pend_root;
    heap_flush(hs, k.lane);
    {
        // :188-189 (no src -> dest path: must not happen) can only be asked once the sweep has finished
        const int64_t t0 = wave_realtime();
        int64_t guard = 0;
        while (!lost && !uni(ld_shared_i32(&S->sweep_done))) {
            if ((++guard & 1023) == 0 && wave_realtime() - t0 > patience) lost = true;
            wave_sleep();
        }
        wave_fence();
        if (lost || uni(ld_shared_i32(&w.sp_d[vb + src].anom)) < 0) { if (k.lane == 0) { set_status(w, c, -6); w.h_cnt[c] = 0; } return; }
    }
    if (hs.ovf) { if (k.lane == 0) set_status(w, c, -5); return; }
    if (k.lane == 0) { w.h_cnt[c] = hs.alloc; atomic_add(&w.counters[CNT_HEAPNODES], (int64_t)hs.alloc); }
}

// The three roles share one kernel, hence one register allocation: with every role reading its pointers from the kernel's
// argument block the compiler loads them all into scalar registers up front, and the heap role - the critical path, 89 SGPRs on
// its own - ran with 36 of its scalars spilled to vector lanes (2.36 ms for a 1 000-record contig against 2.16 in the kernel of
// its own).  The sweep and the pre-pass have time to spare, so THEY read the argument block through a pointer the compiler
// cannot see through: their loads stay inside their branch and land in vector registers.
#if defined(AASM_HOST_EMUL)
AASM_DEV const WS &chain_role_ws(const WS &w) { return w; }
#else
// (WS is the kernel's only argument: offset 0 of the argument segment.  Taking &w instead would make the compiler copy the
// whole structure to scratch memory first.)
AASM_DEV const WS &chain_role_ws(const WS &) {
    const WS *p = (const WS *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *p;
}
#endif
// (two kernels, so that each form's heap wave gets the register allocation of ITS roles: with both forms in one kernel the heap role ran
// with 15 scalars spilled and classes of 1 000-1 280 contigs lost 3-4 %)
template <bool ORD>
AASM_DEV void kb_chain(const KCtx &k, const WS &w) {                // ORD: four waves per contig of the class (sweep, pre-pass, order, heaps); else three (the heap wave keeps its own queue)
    ChainLds *L = (ChainLds *)k.lds;
    const int64_t c = w.chain_list[k.bid];
    const int wv = k.tid / AASM_WAVE;
    if (k.tid == 0) { L->s.tail_pub = 0; L->s.sweep_done = 0; L->s.prep_done = 0; L->s.ord_pub = 0; L->s.ord_done = 0; }
    block_barrier();
    KCtx k2 = k;
    k2.bid = c; k2.tid = k.lane; k2.nthreads = AASM_WAVE;
    if (wv == 0) {
        k2.lds = (char *)&L->rq;
        kb_rev_sweep<AASM_WAVE, true>(k2, chain_role_ws(w), &L->s);
    } else if (wv == 1) {
        chain_prep(k2, chain_role_ws(w), c, &L->s);
    } else if (wv == 2) {
        if (ORD) chain_order(k2, chain_role_ws(w), c, &L->s);
        else {                                                       // (classes of more than AASM_CHAIN_ORD_MAX contigs, probes, tests: the heap wave with its own queue, no order wave)
#if !defined(AASM_HOST_EMUL)
            __builtin_amdgcn_s_setprio(1);
#endif
            k2.lds = (char *)&L->hl;
            kb_heap<true, HEAP_RING_CH, HEAP_QN_CH>(k2, w, &L->s);
        }
    } else if (ORD) {
#if !defined(AASM_HOST_EMUL)
        __builtin_amdgcn_s_setprio(1);                               // the heap wave is the contig's critical path
#endif
        k2.lds = (char *)&L->hl;
        kb_heap_ord<HEAP_RING_CH, HEAP_QN_CH>(k2, w, &L->s);
    }
}

// ====================================================================================
// K7 with SEVERAL WAVES PER CONTIG (wide SP trees: dense / high-multiplicity graphs, giant contigs)
// ====================================================================================
// Persistence means that the heaps of two vertices of which neither is an ancestor of the other share
// nothing they write: independent branches of the SP tree can be built at the same time.  What ties the
// branches together is only the ORDER of allocation - the arena index is the k-walk tie-break (hazard B3) and
// the reference allocates in BFS order of the tree.  So:
//   phase 0  (wave 0) numbers the tree's vertices in BFS order, 64 queue entries per step, and gives every
//            vertex a region of the PROVISIONAL arena sized by the per-insert bound (#keys * (floor(log2(I+1)) + 2)),
//            regions laid out in BFS order;
//   phase 1  every wave builds vertices: it keeps going down its own branch (the first child inherits the heap
//            whose right spine is in its registers) and hands the other children to the waves that wait, through
//            a ticket queue (tickets and counters in LDS, entries in global memory; a producer drains its stores
//            before it publishes);
//   phase 2  all waves compact: nodes move from their regions to the final arena in BFS order - exactly the
//            reference's allocation order - and child pointers / roots are translated (own region: arithmetic,
//            ancestor's region: binary search over the region starts).
// The result is bit-identical to the one-wave kernel's, arena indices included.
#define MW_WAVES 16                      // most waves a contig can get (the launch picks 4, 8 or 16 by how many contigs share the chip)
struct MwLds {
    int32_t q_head, q_tail, n_done, n_total, stop, n_nodes, pad0, pad1;
    HNode bounce[MW_WAVES];              // heap_read's slot, one per wave
    HNode ring[MW_WAVES][HEAP_RING];     // (a launch with fewer waves declares only its share)
};
#define AASM_MW_LDS_BYTES(waves) ((waves) * HEAP_RING * 48 + 32 + MW_WAVES * 48)
static_assert(sizeof(MwLds) <= AASM_MW_LDS_BYTES(MW_WAVES), "LDS budget");

// last position i in [0, n) with a[i] <= x (a non-decreasing, a[0] <= x)
AASM_DEV int32_t mw_last_le(const int32_t *a, int32_t n, int32_t x) {
    int32_t lo = 0, hi = n;
    while (hi - lo > 1) { const int32_t m = (lo + hi) >> 1; if (a[m] <= x) lo = m; else hi = m; }
    return lo;
}

// the launch order: the contigs of the class by their node bound, largest first (rank by counting: the pipeline asks for it only
// while the class is a few thousand contigs)
AASM_DEV void kb_mw_rank(const KCtx &k, const WS &w) {               // thread per contig of the class
    const int64_t i = k.bid * k.nthreads + k.tid;
    if (i >= w.mw_n) return;
    const int32_t ci = w.mw_list[i], key = w.mw_key[i];
    int32_t r = 0;
    int64_t j = 0;
    for (; j + 8 <= w.mw_n; j += 8) {                                // (keys and ids side by side in slot order: eight independent pairs of loads per round)
        int32_t kj[8], cj[8];
        AASM_UNROLL
        for (int u = 0; u < 8; u++) { kj[u] = w.mw_key[j + u]; cj[u] = w.mw_list[j + u]; }
        AASM_UNROLL
        for (int u = 0; u < 8; u++) r += (kj[u] > key || (kj[u] == key && cj[u] < ci)) ? 1 : 0;
    }
    for (; j < w.mw_n; j++) { const int32_t kj = w.mw_key[j], cj = w.mw_list[j]; r += (kj > key || (kj == key && cj < ci)) ? 1 : 0; }
    w.mw_sorted[r] = ci;
}

AASM_DEV void kb_heap_mw(const KCtx &k, const WS &w) {              // MW_WAVES waves per contig
    const int64_t c = (w.mw_base >= 0) ? (int64_t)w.mw_sorted[w.mw_base + k.bid] : k.bid;
    const int64_t V = w.ctgV[c];
    if (V == 0 || !w.mw_flag[c]) return;
    MwLds *L = (MwLds *)k.lds;
    const int64_t vb = w.voff[c];
    const int wv = k.tid / AASM_WAVE;                                // this wave
    int32_t *h = w.h_root + vb, *q = w.bq + vb;
    const I4 *vh = w.vhdr + vb, *vh2 = w.vhdr2 + vb;
    const Dist *sk = w.st_cost + w.rowptr[vb];
    int32_t *order = w.mw_order + vb, *rs = w.mw_rs + vb, *fb = w.mw_fb + vb, *rsv = w.mw_rsv + vb, *used = w.mw_used + vb;
    const int32_t src = (int32_t)(V - 2), dest = (int32_t)(V - 1);
    const int32_t per_insert = w.mw_lg[c];
    if (k.tid == 0) w.h_cnt[c] = 0;
    if (w.status[c] != 0) return;
    if (dist_is_max(w.sp_d[vb + src])) { if (k.tid == 0) set_status(w, c, -6); return; }   // :188-189: no path (must not happen)
#if defined(AASM_KPROF) && !defined(AASM_HOST_EMUL)
    const int64_t mwp_t0 = wave_realtime();                          // diagnostic build: the phases' wall time (100 MHz ticks) of wave 0
    int64_t mwp_t1 = 0, mwp_t2 = 0, mwp_t3 = 0;
#define MWP_STAMP(x) do { x = wave_realtime(); } while (0)
#else
#define MWP_STAMP(x) do {} while (0)
#endif
    // ---- phase 0: BFS numbering (k_shortest_walks.hpp:196-214 without the inserts) + region starts
    if (wv == 0) {
        if (k.lane == 0) order[0] = dest;
        wave_fence();
        int32_t n = 1, head = 0, rbase = 0;
        while (head < n) {
            const int32_t chunk = (n - head < AASM_WAVE) ? (n - head) : AASM_WAVE;
            int32_t v = -1, nch = 0, nin = 0;
            int64_t c0 = 0;
            if (k.lane < chunk) {
                v = order[head + k.lane];
                const I4 a = vh[v], b = vh2[v];
                nin = a.y; nch = a.z;
                c0 = (int64_t)(((uint64_t)(uint32_t)b.y << 32) | (uint32_t)b.x);
            }
            const int32_t capv = nin * per_insert, cincl = wave_incl_add(capv), nincl = wave_incl_add(nch);
            if (k.lane < chunk) { rs[head + k.lane] = rbase + cincl - capv; rsv[v] = rbase + cincl - capv; used[v] = 0; }
            for (int32_t j = 0; j < nch; j++) order[n + (nincl - nch) + j] = w.cval[c0 + j];   // children in ascending id (:191-194)
            rbase += wave_bcast(cincl, AASM_WAVE - 1);
            n += wave_bcast(nincl, AASM_WAVE - 1);
            head += chunk;
            wave_fence();
        }
        if (k.lane == 0) { L->n_total = n; L->q_head = 0; L->q_tail = 0; L->n_done = 0; L->stop = 0; L->n_nodes = 0; }
        store_drain();
    }
    block_barrier();
    MWP_STAMP(mwp_t1);
    const int32_t nv = uni(ld_shared_i32(&L->n_total));
    // ---- phase 1: the heaps, into per-vertex regions of the provisional arena
    HeapState hs;
    hs.nodes = w.hprov + w.mw_off[c]; hs.ring = L->ring[wv]; hs.rn = HEAP_RING; hs.bounce = &L->bounce[wv]; hs.oldn = nullptr; hs.alloc = 0; hs.flushed = 0; hs.ring_lo = 0; hs.cap = 0; hs.ovf = false;
    Spine sp; sp.root = -2; sp.len = 0; sp.tail = -1;
    KProfNone kp;
    for (int i_ = 0; i_ < 8; i_++) kp.acc[i_] = 0;
    int32_t u = (wv == 0) ? dest : -1, hu = -1;
    int64_t guard = 0;
    while (true) {
        if (uni(ld_shared_i32(&L->stop))) break;
        if (u < 0) {                                                 // nothing of its own left: a ticket for the shared queue
            int32_t s = 0;
            if (k.lane == 0) s = atomic_add(&L->q_head, (int32_t)1);
            s = wave_bcast(s, 0);
            const int64_t wait_t0 = wave_realtime();
            while (true) {
                if (uni(ld_shared_i32(&L->stop)) || uni(ld_shared_i32(&L->n_done)) >= nv) { u = -2; break; }
                const int32_t e = (s < V) ? uni(ld_shared_i32(&q[s])) : -1;
                if (e >= 0) { u = e; hu = uni(ld_shared_i32(&h[u])); break; }
                // (never: every wait ends with an entry or with n_done == nv.)  Bounded by wall time, not by a spin count: a wave may
                // rightly wait for as long as the others work on a path-like part of the tree, but 30 s without its ticket being
                // served means a publishing wave is gone, and the launch ends with AASM_E_INTERNAL instead of sitting for minutes
                if ((++guard & 1023) == 0 && wave_realtime() - wait_t0 > (int64_t)30 * 100000000) { u = -2; if (k.lane == 0) L->stop = 2; break; }
                wave_sleep();
            }
            if (u == -2) break;
        }
        const I4 ha = vh[u], hb = vh2[u];
        const int32_t so = uni(ha.x), n = uni(ha.y), nch = uni(ha.z), fc = uni(ha.w);
        const int64_t c0 = (int64_t)(((uint64_t)(uint32_t)uni(hb.y) << 32) | (uint32_t)uni(hb.x));
        const int32_t r0 = uni(rsv[u]);
        heap_flush(hs, k.lane);
        hs.alloc = r0; hs.flushed = r0; hs.ring_lo = r0; hs.cap = r0 + n * per_insert;
        for (int32_t base = 0; base < n && !hs.ovf; base += AASM_WAVE_MAX) {   // inserts in list order (:202-211)
            const int32_t m = (n - base < AASM_WAVE_MAX) ? (n - base) : AASM_WAVE_MAX;
            LaneArr<Dist> key;
            FOR_LANE(t, m, k.lane) key.at(t) = sk[(int64_t)so + base + t];
            for (int32_t t = 0; t < m && !hs.ovf; t++) {
                const Dist cc = la_get_dist(key, t);
                hu = heap_insert<false>(hs, sp, hu, cc, u, cc.pad, k.lane, kp);
            }
        }
        if (hs.ovf) { if (k.lane == 0) L->stop = 1; break; }
        heap_flush(hs, k.lane);
        if (k.lane == 0) { used[u] = hs.alloc - r0; h[u] = hu; }     // (the inherited root this word carried has been read)
        // children adopt the heap (:213): the first one stays with this wave, the others go to whoever waits
        if (nch >= 2) {
            for (int32_t t = 1 + k.lane; t < nch; t += AASM_WAVE) h[w.cval[c0 + t]] = hu;
            store_drain();                                           // nodes, roots: in the shared cache before the entries appear
            int32_t at = 0;
            if (k.lane == 0) at = atomic_add(&L->q_tail, nch - 1);
            at = wave_bcast(at, 0);
            for (int32_t t = 1 + k.lane; t < nch; t += AASM_WAVE) q[at + t - 1] = w.cval[c0 + t];
        }
        store_drain();
        if (k.lane == 0) atomic_add(&L->n_done, (int32_t)1);
        u = (nch > 0) ? fc : -1;
    }
    heap_flush(hs, k.lane);
    store_drain();
    block_barrier();
    MWP_STAMP(mwp_t2);
    const int32_t stop = uni(ld_shared_i32(&L->stop));
    if (stop) { if (k.tid == 0) set_status(w, c, stop == 1 ? -5 : -6); return; }
    // ---- phase 2: compaction into the final arena, BFS order = the reference's allocation order
    if (wv == 0) {
        int32_t carry = 0;
        for (int32_t base = 0; base < nv; base += AASM_WAVE) {
            const int32_t i = base + k.lane;
            const int32_t x = (i < nv) ? used[order[i]] : 0, incl = wave_incl_add(x);
            if (i < nv) fb[i] = carry + incl - x;
            carry += wave_bcast(incl, AASM_WAVE - 1);
        }
        if (k.lane == 0) L->n_nodes = carry;
        store_drain();
    }
    block_barrier();
    const int32_t H = uni(ld_shared_i32(&L->n_nodes));
    if (!w.mw_compact) {                                             // the nodes stay where they are (heap_arena): roots and child pointers are provisional indices already
        if (k.tid == 0) { w.h_cnt[c] = H; atomic_add(&w.counters[CNT_HEAPNODES], (int64_t)H); }
        return;
    }
    HNode *fin = w.hnodes + w.hoff[c];
    if (H > (int32_t)(w.hoff[c + 1] - w.hoff[c])) { if (k.tid == 0) set_status(w, c, -5); return; }
    // The region starts (the keys of every pointer translation) go to LDS when they fit in the rings' space, which is free now:
    // a node's two pointers mostly lead into ancestors' regions, i.e. two binary searches of ~12 dependent reads each - from
    // global memory they were 20 of the 44 ms a dense 1 000-record contig's block took (phase 1: 24).
    int32_t *rs_l = (int32_t *)&L->ring[0][0];
    const int nwaves = k.nthreads / AASM_WAVE;
    const bool rs_in_lds = nv <= nwaves * (int32_t)(HEAP_RING * sizeof(HNode) / sizeof(int32_t));
    if (rs_in_lds) for (int32_t i = k.tid; i < nv; i += k.nthreads) rs_l[i] = rs[i];
    block_barrier();
    auto compact = [&](const int32_t *rsk) {                        // (called once per address space of rsk, so that the LDS form reads with ds_read)
        auto translate = [&](int32_t x, int32_t i_own) -> int32_t { // provisional index -> final index
            if (x < 0) return x;
            const int32_t j = (i_own >= 0 && x >= rsk[i_own]) ? i_own : mw_last_le(rsk, i_own >= 0 ? i_own : nv, x);
            return fb[j] + (x - rsk[j]);
        };
        // a wave takes the regions wv, wv + nwaves, ... whole (no search for the region of a node), its lanes the nodes of a region
        for (int32_t ib = wv; ib < nv; ib += nwaves * AASM_WAVE) {  // (the sizes and final starts of its next 64 regions: two loads, not two per region)
            const int32_t i_l = ib + k.lane * nwaves;
            int32_t n_l = 0, f_l = 0;
            if (i_l < nv) { n_l = used[order[i_l]]; f_l = fb[i_l]; }
            for (int32_t q = 0; q < AASM_WAVE; q++) {
                const int32_t i = ib + q * nwaves;
                if (i >= nv) break;
                const int32_t n_i = wave_bcast(n_l, q), f0 = wave_bcast(f_l, q), r0 = uni(rsk[i]);
                for (int32_t t = k.lane; t < n_i; t += AASM_WAVE) {
                    HNode nd = hs.nodes[r0 + t];
                    nd.left = translate(nd.left, i); nd.right = translate(nd.right, i);
                    fin[f0 + t] = nd;
                }
            }
        }
        for (int32_t i = k.tid; i < nv; i += k.nthreads) { const int32_t v = order[i]; h[v] = translate(h[v], -1); }
    };
    if (rs_in_lds) compact(rs_l); else compact(rs);
    if (k.tid == 0) { w.h_cnt[c] = H; atomic_add(&w.counters[CNT_HEAPNODES], (int64_t)H); }
#if defined(AASM_KPROF) && !defined(AASM_HOST_EMUL)
    block_barrier();
    MWP_STAMP(mwp_t3);
    if (k.tid == 0 && w.prof_heap) { int64_t *pp = w.prof_heap + c * 8; pp[0] = mwp_t1 - mwp_t0; pp[1] = mwp_t2 - mwp_t1; pp[2] = mwp_t3 - mwp_t2; pp[3] = H; pp[4] = nv; }
    if (k.lane == 0 && w.prof_heap) { atomic_add(&w.prof_heap[c * 8 + 5], kp.acc[6]); atomic_add(&w.prof_heap[c * 8 + 6], kp.acc[7]); }   // chase steps served by global memory / by the ring, all waves
#endif
}

// ====================================================================================
// K8  k-walk enumeration (k_shortest_walks.hpp:217-249)
// ====================================================================================
// std::greater<tuple<Distance, heap_t*, int64_t>>: entries in the queue are distances of
// real walks (non-negative components), on which operator< is a strict weak order, so the
// tuple order is total and any correct min-queue pops the same sequence.  The pointer is
// replaced by the arena index (DESIGN.md "hazard B3").
//
// The queue is a wave-cooperative 8-ARY heap of 32-byte keys {score sum, anom, qul_nonzero, qul_total, heap
// node, insertion index}: everything the order needs (the query / reference split of the score is not part of
// it and waits in the candidate record, kcand, until the candidate is popped).  The 8 children of a heap node are 256
// contiguous bytes, fetched by 8 lanes in one pass and min-reduced on the score sum by three DPP steps inside
// the 8 lanes - equal sums are settled by a short scalar loop over the tied lanes - so a pop descends log8(n)
// levels (6 for the 30 001 entries of K = 10 000) instead of log2(n) = 15; the top four levels (1 + 8 + 64 +
// 512 entries, 18 KB) live in LDS.  (First version: 48-byte entries, three levels in LDS, an 8-field shuffle
// butterfly per level: 9 us per pop at K = 10 000.)
#define PQ8_LDS_N 585
#define AASM_ENUM_LDS_BYTES (PQ8_LDS_N * 32 + 32)
struct __attribute__((aligned(16))) PqK { int64_t sum; int32_t anom, qnz, qtot, node, cur, pad; };
// std::tuple order of (Distance, node, cur) for distances of real walks (never max(), all components >= 0):
// paf_data.hpp:142-159 then node index then insertion index
AASM_DEV bool pq_full_less(const Dist &a, int32_t a_node, int32_t a_cur, const Dist &b, int32_t b_node, int32_t b_cur) {
    const int64_t sa = a.qry + a.ref, sb = b.qry + b.ref;
    if (sa != sb) return sa < sb;
    if (a.anom != b.anom) return a.anom < b.anom;
    const int32_t ta = a.qtot ? a.qtot : 1, tb = b.qtot ? b.qtot : 1;
    const int64_t l = (int64_t)a.qnz * (int64_t)tb, r = (int64_t)b.qnz * (int64_t)ta;
    if (l != r) return l > r;
    if (a_node != b_node) return a_node < b_node;
    return a_cur < b_cur;
}
AASM_DEV bool pqk_less(const PqK &a, const PqK &b) {                 // the same order on heap keys
    if (a.sum != b.sum) return a.sum < b.sum;
    if (a.anom != b.anom) return a.anom < b.anom;
    const int32_t ta = a.qtot ? a.qtot : 1, tb = b.qtot ? b.qtot : 1;
    const int64_t l = (int64_t)a.qnz * (int64_t)tb, r = (int64_t)b.qnz * (int64_t)ta;
    if (l != r) return l > r;
    if (a.node != b.node) return a.node < b.node;
    return a.cur < b.cur;
}
struct Pq8 { PqK *g, *l; int32_t n; int32_t pc_idx; PqK pc; };      // pc*: last parent read by a push
AASM_DEV PqK pq8_get(const Pq8 &q, int32_t i) {
    PqK e;
    if (i < PQ8_LDS_N) e = q.l[i];
    else { e = q.g[i]; asm volatile("" ::: "memory"); }              // keep LDS and global reads apart (no flat access)
    return e;
}
AASM_DEV void pq8_set(Pq8 &q, int32_t i, const PqK &e, int lane) {
    if (lane == 0) { if (i < PQ8_LDS_N) q.l[i] = e; else q.g[i] = e; }
    if (i == q.pc_idx) q.pc_idx = -1;
}
AASM_DEV PqK pqk_uni(const PqK &k) { PqK r; r.sum = uni(k.sum); r.anom = uni(k.anom); r.qnz = uni(k.qnz); r.qtot = uni(k.qtot); r.node = uni(k.node); r.cur = uni(k.cur); r.pad = 0; return r; }
#if !defined(AASM_HOST_EMUL)
AASM_DEV PqK pqk_lane(const PqK &k, int j) {                         // the key lane j holds (wave-uniform j)
    PqK r;
    r.sum = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)((uint64_t)k.sum >> 32), j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)k.sum, j));
    r.anom = __builtin_amdgcn_readlane(k.anom, j); r.qnz = __builtin_amdgcn_readlane(k.qnz, j); r.qtot = __builtin_amdgcn_readlane(k.qtot, j);
    r.node = __builtin_amdgcn_readlane(k.node, j); r.cur = __builtin_amdgcn_readlane(k.cur, j); r.pad = 0;
    return r;
}
#endif
AASM_DEV void pq8_push(Pq8 &q, const PqK &x, int lane) {
    int32_t i = q.n++;
    while (i > 0) {
        const int32_t p = (i - 1) >> 3;
        // the <= 3 pushes of one iteration land side by side and almost always share a parent
        if (p != q.pc_idx) { q.pc = pqk_uni(pq8_get(q, p)); q.pc_idx = p; }
        const PqK pe = q.pc;
        if (!pqk_less(x, pe)) break;
        pq8_set(q, i, pe, lane);
        i = p;
    }
    pq8_set(q, i, x, lane);
    wave_lds_sync();
}
#if !defined(AASM_HOST_EMUL)
// the same order, branch-free, for keys held one per lane (the distance sums tie all the time: the K best walks of
// a contig spread over a few dozen distinct sums, so every field of the order is in play)
AASM_DEV bool pqk_less_v(const PqK &a, const PqK &b) {
    const int32_t ta = a.qtot ? a.qtot : 1, tb = b.qtot ? b.qtot : 1;
    const int64_t l = (int64_t)a.qnz * (int64_t)tb, r = (int64_t)b.qnz * (int64_t)ta;
    const int64_t ia = (int64_t)(((uint64_t)(uint32_t)a.node << 32) | (uint32_t)a.cur), ib = (int64_t)(((uint64_t)(uint32_t)b.node << 32) | (uint32_t)b.cur);
    return (a.sum < b.sum) | ((a.sum == b.sum) & ((a.anom < b.anom) | ((a.anom == b.anom) & ((l > r) | ((l == r) & (ia < ib))))));
}
template <int CTRL> AASM_DEV int32_t dpp_mov(int32_t x) { return __builtin_amdgcn_mov_dpp(x, CTRL, 0xf, 0xf, true); }
template <int CTRL> AASM_DEV void pqk_min_step(PqK &k, int32_t &slot) {    // k <- min(k, partner's k) inside a group of 8 lanes
    PqK o;
    o.sum = (int64_t)(((uint64_t)(uint32_t)dpp_mov<CTRL>((int)((uint64_t)k.sum >> 32)) << 32) | (uint32_t)dpp_mov<CTRL>((int)(uint32_t)(uint64_t)k.sum));
    o.anom = dpp_mov<CTRL>(k.anom); o.qnz = dpp_mov<CTRL>(k.qnz); o.qtot = dpp_mov<CTRL>(k.qtot); o.node = dpp_mov<CTRL>(k.node); o.cur = dpp_mov<CTRL>(k.cur); o.pad = 0;
    const int32_t oslot = dpp_mov<CTRL>(slot);
    const bool take = pqk_less_v(o, k);
    k.sum = take ? o.sum : k.sum; k.anom = take ? o.anom : k.anom; k.qnz = take ? o.qnz : k.qnz; k.qtot = take ? o.qtot : k.qtot;
    k.node = take ? o.node : k.node; k.cur = take ? o.cur : k.cur; slot = take ? oslot : slot;
}
#endif
// x takes the place of the root and sinks to where it belongs
AASM_DEV void pq8_sink_from_root(Pq8 &q, const PqK &x, int lane) {
    int32_t i = 0;
    while (true) {
        const int32_t c0 = 8 * i + 1;
        if (c0 >= q.n) break;
        const int32_t nchild = (q.n - c0 < 8) ? (q.n - c0) : 8;
        PqK best;                                                    // the smallest child, wave-uniform
        int32_t bslot;
#if defined(AASM_HOST_EMUL)
        best = pq8_get(q, c0); bslot = 0;
        for (int32_t j = 1; j < nchild; j++) { const PqK e = pq8_get(q, c0 + j); if (pqk_less(e, best)) { best = e; bslot = j; } }
#else
        PqK mine; mine.sum = INT64_MAX; mine.anom = mine.qnz = mine.qtot = mine.node = mine.cur = mine.pad = 0;
        if (lane < nchild) mine = pq8_get(q, c0 + lane);             // (the children of one node are all in LDS or all in global memory)
        int32_t slot = lane;
        pqk_min_step<0xB1>(mine, slot);                              // quad_perm [1,0,3,2]
        pqk_min_step<0x4E>(mine, slot);                              // quad_perm [2,3,0,1]
        pqk_min_step<0x141>(mine, slot);                             // row_half_mirror: lane i <-> 7 - i
        best = pqk_uni(mine); bslot = uni(slot);                     // lanes 0..7 all hold the smallest child now
#endif
        if (!pqk_less(best, x)) break;
        pq8_set(q, i, best, lane);
        i = c0 + bslot;
        wave_lds_sync();
    }
    pq8_set(q, i, x, lane);
    wave_lds_sync();
}
AASM_DEV PqK pq8_top(const Pq8 &q) { return pqk_uni(pq8_get(q, 0)); }
AASM_DEV void pq8_drop_top(Pq8 &q, int lane) {                      // the last entry replaces the root
    const PqK x = pqk_uni(pq8_get(q, --q.n));
    q.pc_idx = -1;
    if (q.n > 0) pq8_sink_from_root(q, x, lane);
}

AASM_DEV void kb_enum_heap(const KCtx &k, const WS &w) {            // one wave per contig (the d-ary heap form: host emulation, and the device cross-check of kb_enum_lsm)
    const int64_t c = k.bid;
    const int64_t V = w.ctgV[c];
    if (k.lane == 0) w.kfound[c] = 0;
    if (V == 0 || w.status[c] != 0) return;
    const int64_t vb = w.voff[c];
    const int64_t K = w.K;
    Dist *kd = w.kd + c * K;
    int32_t *klast = w.klast + c * K;
    I4 *kcand = w.kcand + 2 * c * (3 * K + 1);                       // {heap node, predecessor, qry_score} of candidate `cur` (the key carries the sum)
    Pq8 q; q.g = w.pq + c * w.pq_stride; q.l = (PqK *)k.lds; q.n = 0; q.pc_idx = -1;
    static_assert(PQ8_LDS_N * sizeof(PqK) <= AASM_ENUM_LDS_BYTES, "LDS budget");
    const HNode *nodes = heap_arena(w, c);
    const int32_t *h = w.h_root + vb;
    const int32_t src = (int32_t)(V - 2);
    const bool L0 = k.lane == 0;
    int32_t found = 0, nn = 0;
    const Dist dsrc = w.sp_d[vb + src];
    if (L0) { kd[0] = dsrc; klast[0] = -1; }                         // :217-220
    found = 1;
    const int32_t hs = uni(h[src]);
    auto emplace = [&](const Dist &dd, int32_t hp, int32_t pre) {    // :232-237
        PqK x; x.sum = uni(dd.qry + dd.ref); x.anom = uni(dd.anom); x.qnz = uni(dd.qnz); x.qtot = uni(dd.qtot); x.node = hp; x.cur = nn; x.pad = 0;
        if (L0) { I4 cd; cd.x = hp; cd.y = pre; cd.z = (int32_t)(uint32_t)(uint64_t)dd.qry; cd.w = (int32_t)((uint64_t)dd.qry >> 32); kcand[2 * (int64_t)nn] = cd; }
        pq8_push(q, x, k.lane);
        nn++;
    };
    KPROF_DECL;
    KPROF_START();
    if (hs >= 0) {                                                  // :227-228
        emplace(dist_add(dsrc, hnode_key(nodes[hs])), hs, -1);       // :239
        while (q.n > 0 && found < K) {                              // :240-248
            KPROF_STAMP(3);                                          // pushes of the previous pop
            const PqK top = pq8_top(q);
            pq8_drop_top(q, k.lane);
            KPROF_STAMP(0);                                          // pop
            const int32_t tcur = top.cur, tnode = top.node;
            wave_fence();
            const I4 tcd = kcand[2 * (int64_t)tcur];
            Dist dtop; dtop.qry = uni((int64_t)(((uint64_t)(uint32_t)tcd.w << 32) | (uint32_t)tcd.z)); dtop.ref = top.sum - dtop.qry; dtop.anom = top.anom; dtop.qnz = top.qnz; dtop.qtot = top.qtot; dtop.pad = 0;
            const HNode ch = nodes[tnode];
            const int32_t prev_of_top = uni(tcd.y);
            if (L0) { kd[found] = dtop; klast[found] = tcur; }
            found++;
            const int32_t hv = uni(h[ch.v]);
            const Dist chk = hnode_key(ch);
            const int32_t cl = uni(ch.left), cr = uni(ch.right);
            KPROF_STAMP(1);                                          // popped node + cross heap root
            if (hv >= 0) emplace(dist_add(dtop, hnode_key(nodes[hv])), hv, tcur);
            if (cl >= 0) emplace(dist_sub(dist_add(dtop, hnode_key(nodes[cl])), chk), cl, prev_of_top);
            if (cr >= 0) emplace(dist_sub(dist_add(dtop, hnode_key(nodes[cr])), chk), cr, prev_of_top);
        }
    }
#if defined(AASM_KPROF) && !defined(AASM_HOST_EMUL)
    if (w.K >= 1000) KPROF_FLUSH(w.prof_heap, c, k.lane);            // diagnostic build: K8's sections replace K7's in the dump
#endif
    if (L0) {
        w.kfound[c] = found;
        atomic_add(&w.counters[CNT_PATHS], (int64_t)found);
        atomic_add(&w.counters[CNT_PQ_PUSH], (int64_t)nn);
    }
}

// ====================================================================================
// K9  path recovery, upgrade, conversion, selection
//     (k_shortest_walks.hpp:254-290; paf_data.cpp:750-921,1489-1649)
// ====================================================================================
#if defined(AASM_HOST_EMUL)
static int64_t g_k9_stat[64];                  // emulation-only step statistics of the upgrade loop (tools/k9_steps.py)
#define K9STAT(i, n) (g_k9_stat[i] += (n))
#else
#define K9STAT(i, n) do {} while (0)
#endif
struct SelCtx {
    const WS *w;
    int64_t c, b, N, V, vb, cap;     // cap = N + 2 edge pairs per path buffer
    int32_t src, dest;
    int32_t *pathA, *pathB, *pathT, *pre2, *stamp;
    Dist *dist2;
    OutElem *cur;                    // where the conversion leaves its elements
    int64_t *cls;                    // per edge of pathA {position of its head, v_j << 2 | flags} (sel_classify_edge); lives in the bytes of `cur` until the elements are written
    int32_t epoch, last_head, last_pos;   // last_pos: topological position of last_head (-1: not known)
    int32_t res_ppos;                // after a window DP: position of the result's last-but-one vertex
    int32_t *out_dst;
    int32_t out_n, out_flushed;
    int32_t pa_base, pa_n;           // device: edges pa_base .. pa_base + pa_n - 1 of pathA are in the lanes' registers ...
    int32_t wu, wv, wpv, wvf;        // ... lane t: edge pa_base + t = (wu, wv), position of its head, v_j << 2 | flags
    int32_t cw_pa, cw_n;             // the LDS copy of the topologically ordered CSR: first position (-1: none) and how many (sel_cw_fill)
    bool err, res_lds;
    int lane;
    char *lds;
    int64_t n_ispr_e, n_ispr_v, n_path_e, n_out_e;   // byte-model counters
#if defined(AASM_KPROF) && !defined(AASM_HOST_EMUL)
    int64_t kp_t0, kp_acc[8];
#endif
};
#if defined(AASM_KPROF) && !defined(AASM_HOST_EMUL)
#define SPROF(s, i) do { __builtin_amdgcn_s_waitcnt(0); const int64_t t1_ = (int64_t)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0); (s).kp_acc[i] += t1_ - (s).kp_t0; (s).kp_t0 = t1_; } while (0)
#else
#define SPROF(s, i) do {} while (0)
#endif

// ---- LDS working set of the conversion kernels ------------------------------------------
// vmcnt is in-order over loads and stores, so a global store directly ahead of a dependent
// global load costs a full store round trip.  The conversion therefore keeps its small
// intermediate lists in LDS: a 64-edge write buffer that is flushed by one coalesced store (first in the
// block: kb_sel_recover needs nothing else), the window DP's cache + state + result.
#define ISPR_MAX_E 192
#define SEL_WIN 64
struct SelLds {
    int32_t pb_buf[2 * SEL_WIN];   // write buffer in front of pathA / pathB
    int64_t wq[ISPR_MAX_E];
    int32_t wr[ISPR_MAX_E];
    int8_t tgt[ISPR_MAX_E];
    uint8_t fl[ISPR_MAX_E];
    Dist dist[64];
    int32_t excl[64], u[64], vj[64];
    int8_t pre[64];
    uint8_t reach[64];
    int32_t res[2 * 64];           // ISPR result edges (u, v), reverse order
};
#define AASM_SEL_LDS_BYTES 6656
#define AASM_SELREC_LDS_BYTES 512
static_assert(sizeof(SelLds) <= AASM_SEL_LDS_BYTES, "LDS budget");
static_assert(offsetof(SelLds, wq) == AASM_SELREC_LDS_BYTES, "kb_sel_recover allocates the write buffer alone");
// A window DP of 64 ... ISPR_WIDE positions keeps its state in the same bytes: it OVERLAYS the block from the cache on (the LDS copy of
// the CSR and the narrow DP's arrays - the copy is given up for the call), so it needs no launch of its own.
#define ISPR_WIDE 127
struct SelWide { Dist dist[ISPR_WIDE + 1]; int32_t excl[ISPR_WIDE + 2]; int32_t vj[ISPR_WIDE + 1]; uint8_t pre[ISPR_WIDE + 1]; uint8_t reach[ISPR_WIDE + 1]; };
static_assert(offsetof(SelLds, wq) + sizeof(SelWide) <= offsetof(SelLds, res), "the wide DP state must end before the result array");

AASM_DEV void sel_out_flush(SelCtx &s) {
    SelLds *L = (SelLds *)s.lds;
    wave_lds_sync();
    const int32_t n = s.out_n - s.out_flushed;
    for (int32_t t = s.lane; t < 2 * n; t += AASM_WAVE) s.out_dst[2 * s.out_flushed + t] = L->pb_buf[t];
    s.out_flushed = s.out_n;
    wave_lds_sync();
}
AASM_DEV void sel_out_begin(SelCtx &s, int32_t *dst) { s.out_dst = dst; s.out_n = 0; s.out_flushed = 0; }
// append edge (u, v); pos_v = topological position of v (-1: not known)
AASM_DEV void sel_push(SelCtx &s, int32_t u, int32_t v, int32_t pos_v) {
    SelLds *L = (SelLds *)s.lds;
    if (s.out_n >= s.cap) { s.err = true; return; }
    const int32_t slot = s.out_n - s.out_flushed;
    if (s.lane == 0) { L->pb_buf[2 * slot] = u; L->pb_buf[2 * slot + 1] = v; }
    s.out_n++;
    s.last_head = v; s.last_pos = pos_v;
    if (s.out_n - s.out_flushed == SEL_WIN) sel_out_flush(s);
}
// the same for up to 64 edges at once: lanes l0 .. l0 + m - 1 append (u, v) of their edge, in lane order (l0, m wave-uniform, m <= SEL_WIN).
// The caller sets last_head / last_pos.
#if !defined(AASM_HOST_EMUL)
AASM_DEV void sel_push_lanes(SelCtx &s, int32_t l0, int32_t m, int32_t u, int32_t v) {
    SelLds *L = (SelLds *)s.lds;
    if (s.out_n + m > s.cap) { s.err = true; return; }
    const int32_t slot0 = s.out_n - s.out_flushed, room = SEL_WIN - slot0;
    const int32_t first = m < room ? m : room;
    const int32_t k = s.lane - l0;
    if (k >= 0 && k < first) { L->pb_buf[2 * (slot0 + k)] = u; L->pb_buf[2 * (slot0 + k) + 1] = v; }
    s.out_n += first;
    if (s.out_n - s.out_flushed == SEL_WIN) sel_out_flush(s);
    if (m > first) {
        if (k >= first && k < m) { L->pb_buf[2 * (k - first)] = u; L->pb_buf[2 * (k - first) + 1] = v; }
        s.out_n += m - first;
    }
}
#endif

// k_shortest_walks.hpp:254-290 -> pathA; returns #edges or -1
AASM_DEV int32_t sel_recover(SelCtx &s, int32_t kidx) {
    const WS &w = *s.w;
    const int64_t K = w.K;
    if (kidx < 0 || kidx >= w.kfound[s.c]) return 0;
    const I4 *kcand = w.kcand + 2 * s.c * (3 * K + 1);
    const HNode *nodes = heap_arena(w, s.c);
    int32_t ns = 0;
    int32_t cur = w.klast[s.c * K + kidx];
    while (cur != -1) {                                             // sidetrack chain, newest first
        if (ns >= s.cap) { s.err = true; return -1; }
        const I4 cd = kcand[2 * (int64_t)cur];                       // {heap node, predecessor, ...}
        const HNode nd = nodes[cd.x];
        if (s.lane == 0) { s.pathT[2 * ns] = nd.u; s.pathT[2 * ns + 1] = nd.v; }
        ns++;
        cur = cd.y;
    }
    wave_fence();
    sel_out_begin(s, s.pathA);
    int32_t idx = ns - 1, cv = s.src;
    int32_t st_u = -1, st_v = -1;                                   // next sidetrack (tail, head)
    if (idx >= 0) { st_u = uni(s.pathT[2 * idx]); st_v = uni(s.pathT[2 * idx + 1]); }
    while (cv != s.dest || idx >= 0) {
        if (s.err) return -1;
        if (idx >= 0 && cv == st_u) {
            sel_push(s, cv, st_v, -1);
            cv = st_v; idx--;
            if (idx >= 0) { st_u = uni(s.pathT[2 * idx]); st_v = uni(s.pathT[2 * idx + 1]); }
        } else {                                                     // up to sixteen tree edges per load, appended by sixteen lanes
            const int32_t *rec = w.tnx16 + 16 * (s.vb + cv);
#if defined(AASM_HOST_EMUL)
            for (int t = 0; t < 16; t++) {
                const int32_t nx = rec[t];
                if (nx < 0) { s.err = true; return -1; }
                sel_push(s, cv, nx, -1);
                cv = nx;
                if (s.err || cv == s.dest || (idx >= 0 && cv == st_u)) break;   // the outer loop decides what comes next
            }
#else
            const bool in = s.lane < 16;
            const int32_t x = in ? rec[s.lane] : -1;
            const uint64_t stop = wave_ballot(in && (x == s.dest || (idx >= 0 && x == st_u)));   // the walk ends AFTER such an edge
            const uint64_t bad = wave_ballot(in && x < 0);
            const int32_t m = stop ? ffs64(stop) : 16;
            if (bad && ffs64(bad) <= m) { s.err = true; return -1; }
            int32_t from = wave_shfl_up(x, 1, cv);
            if (s.lane == 0) from = cv;
            sel_push_lanes(s, 0, m, from, x);
            cv = wave_bcast(x, m - 1);
#endif
        }
    }
    if (s.err) return -1;
    sel_out_flush(s);
    return s.out_n;
}

// ---- per-edge classification of a recovered path (one thread per edge) ---------------------
// What the upgrade asks about EVERY edge t = (u, v) of the path, computed for all edges at once before the sequential
// steps start (it was done window by window inside them: ~8 dependent round trips per 64 edges, a fifth of a conversion):
//   * the topological position of its head, v_j of its head, whether the head is a single-record vertex (v_i == v_j);
//   * SETTLED: for (u, v) followed by (v, nv), internal_shortest_path_recover(u, nv) is decided without running it.  If nv sits
//     exactly two positions after u (v is the only vertex between them) and the graph has no edge u -> nv, then u -> v -> nv is
//     the only path of the window, whatever the whitelist (v carries it).  If it sits three positions after u there is one more
//     vertex y in the window: the same holds if there is no edge u -> nv, no u -> y -> nv, and no three-hop path through y
//     (u -> y -> v or v -> y -> nv).  The rows of the positions pos(u) ... pos(u) + 2 - u itself and the one or two vertices
//     in between - are ONE contiguous run of the topologically ordered CSR copy, whose heads are positions already: four row
//     pointers and one short scan answer all of it (windows with more than SETTLE_SCAN_MAX edges are left to the DP).
//     The step loop uses it whenever its continuation vertex is u (99 % of the time).
// Also the tp-flag marks of the un-upgraded path (paf_data.cpp:1490-1496): mark_time[r] = smallest conversion ordinal whose
// path touched record r.
// cls[t] = {position of the head, v_j << 2 | settled << 1 | single}.
#define SETTLE_SCAN_MAX 48
AASM_DEV void sel_classify_edge(const WS &w, int64_t vb, int32_t dest, const int32_t *pathA, int32_t la, int32_t t, int32_t *mark, int32_t ord, int64_t *cls) {
    const int32_t tu = pathA[2 * t], hv = pathA[2 * t + 1];
    const int32_t nvv = (t + 1 < la) ? pathA[2 * t + 3] : -1;
    const int32_t a = w.v_i[vb + hv], b = w.v_j[vb + hv];
    const int32_t pu = w.fwd_pos[vb + tu], pv = w.fwd_pos[vb + hv];
    const int32_t pnv = nvv >= 0 ? w.fwd_pos[vb + nvv] : -1;
    if (hv != dest) {                                                // (a record the main path marked needs no atomic from the tie / alt walks)
        atomic_min_i32(&mark[a], ord);                                // (fire and forget: a look at mark[] first would make it a dependent pair)
        if (b != a) atomic_min_i32(&mark[b], ord);
    }
    int32_t fl = (a == b) ? 1 : 0;
    const int32_t dpos = pnv - pu;
    if (nvv >= 0 && (dpos == 2 || dpos == 3)) {
        const int64_t *tp = w.tp_ptr + vb + pu;
        const int64_t p0 = tp[0], p1 = tp[1], p2 = tp[2], p3 = (dpos == 3) ? tp[3] : p2;
        if (p3 - p0 <= SETTLE_SCAN_MAX) {
            // rows: [p0, p1) = u, [p1, p2) = position pu + 1, [p2, p3) = position pu + 2 (dpos == 3 only)
            const bool y_first = pv == pu + 2;                       // dpos == 3: the other vertex y sits at pu + 1 (else at pu + 2)
            const int32_t py = y_first ? pu + 1 : pu + 2;
            bool u_nv = false, u_y = false, y_nv = false, y_v = false, v_y = false;
            for (int64_t e = p0; e < p3; e++) {
                const int32_t tg = w.te_pk[e].x;
                const int32_t row = e < p1 ? 0 : e < p2 ? 1 : 2;    // position pu + row
                if (row == 0) { u_nv |= tg == pnv; u_y |= tg == py; }
                else if (pu + row == py) { y_nv |= tg == pnv; y_v |= tg == pv; }
                else v_y |= tg == py;
            }
            const bool other = (dpos == 2) ? u_nv : (u_nv || (u_y && y_nv) || (y_first ? (u_y && y_v) : (v_y && y_nv)));
            if (!other) { fl |= 2; K9STAT(dpos == 2 ? 16 : 17, 1); }
        }
    }
    cls[t] = (int64_t)(((uint64_t)(((uint32_t)b << 2) | (uint32_t)fl) << 32) | (uint32_t)pv);
}

// internal_shortest_path_recover (paf_data.cpp:750-792): QRY_SCORE-mode DAG DP over the
// forward topological window [order[a], order[b)).  The result edges are left in REVERSE
// order in LDS (res_lds) or in pathT; returns their count, 0 when a == b, -1 on "must not
// happen".
//
// Targets beyond position order[b] are never expanded and never lie on the returned path,
// so relaxing them (as the reference's hash map does) is unobservable and is skipped.
// Forms (sel_ispr picks):
//  * window of <= ISPR_REG_W positions and <= 64 edges inside the LDS copy (99 % of the calls on sparse graphs): every
//    edge of the window in a lane's registers, the DP position by position with the candidates folded in lane order
//    (sel_ispr_reg, device only);
//  * window of <= 63 vertices and <= ISPR_MAX_E edges: staged in LDS from the topologically ordered CSR copy by two
//    coalesced load rounds, the DP runs on LDS state, lanes sharing one source's edges;
//  * up to ISPR_WIDE positions, any number of edges: state in LDS, the rows streamed (sel_ispr_stream);
//  * otherwise: same DP on epoch-stamped global arrays, lanes sharing one source's edges (sel_ispr_generic).
// Inside a row targets are distinct, so the lane-parallel relaxation is conflict-free and
// the sequential source order keeps the reference's strict-`<` first-wins behaviour.
AASM_DEV int32_t sel_ispr_generic(SelCtx &s, int32_t a, int32_t bd, bool wl_flag, int32_t wl, int32_t pa, int32_t pb) {
    // Windows too wide or too dense for LDS (most of a dense graph's): the same DP on epoch-stamped global arrays, indexed by
    // topological POSITION, over the topologically ordered copy of the CSR - a window's positions, rows and state are then
    // consecutive memory: the stamps of 64 positions are one load (a position nobody reached costs nothing; the ones reached
    // by a source of the same 64 are added to the mask as they are relaxed), a source's row starts at tp_ptr[position], and an
    // edge's head position comes with the edge (by vertex id it was order[i] -> stamp -> row pointers -> heads -> positions
    // -> state: five dependent round trips per position, 137 us per call on the dense C5 graphs).
    const WS &w = *s.w;
    const int32_t *order = w.fwd_order + s.vb;
    const int32_t ep = ++s.epoch;
    s.res_lds = false;
    SPROF(s, 5);
    K9STAT(10, 1);
    if (s.lane == 0) { s.dist2[pa] = dist_zero(); s.pre2[pa] = -1; s.stamp[pa] = ep; }
    wave_fence();
    for (int32_t i0 = pa; i0 < pb; i0 += AASM_WAVE) {
        const int32_t nchunk = (pb - i0 < AASM_WAVE) ? (pb - i0) : AASM_WAVE;
        uint64_t reach = wave_ballot(s.lane < nchunk && s.stamp[i0 + s.lane] == ep);   // reached positions of this chunk
        for (int32_t t = 0; t < nchunk; t++) {
            if (!((reach >> t) & 1ull)) continue;
            const int32_t i = i0 + t;
            const Dist cd = s.dist2[i];
            const int64_t r0 = uni(w.tp_ptr[s.vb + i]), r1 = uni(w.tp_ptr[s.vb + i + 1]);
            const bool u_ok = uni(w.tp_vj[s.vb + i]) == wl;          // :767-773 (src / dest carry -1 / -2: never a whitelist match)
            s.n_ispr_v++; s.n_ispr_e += r1 - r0;
            for (int64_t e0 = r0; e0 < r1; e0 += AASM_WAVE) {
                const int64_t e = e0 + s.lane;
                int32_t tg = INT32_MAX;
                bool upd = false;
                if (e < r1) {
                    const I4 pk = w.te_pk[e];
                    tg = pk.x;
                    if (tg <= pb && !(wl_flag && tg == pb && !u_ok)) {
                        const Dist nd = dist_add(cd, edge_dist(te_wq(pk), te_wr(pk), te_fl(pk)));
                        if (s.stamp[tg] != ep || dist_lt<QRY_SCORE_MODE>(nd, s.dist2[tg])) { s.dist2[tg] = nd; s.pre2[tg] = i; s.stamp[tg] = ep; upd = true; }
                    }
                }
                uint64_t m = wave_ballot(upd && tg < i0 + nchunk);   // heads inside this chunk: reached from now on
                while (m) { const int j = ffs64(m) - 1; m &= m - 1; reach |= 1ull << (wave_bcast(tg, j) - i0); }
            }
            wave_fence();
        }
    }
    if (uni(s.stamp[pb]) != ep) { s.err = true; return -1; }        // :783
    s.res_ppos = uni(s.pre2[pb]);
    int32_t n = 0, last = pb;
    while (last != pa) {
        if (n >= s.cap) { s.err = true; return -1; }
        const int32_t pv = uni(s.pre2[last]);
        if (s.lane == 0) { s.pathT[2 * n] = order[pv]; s.pathT[2 * n + 1] = (last == pb) ? bd : order[last]; }
        n++;
        last = pv;
    }
    wave_fence();
    SPROF(s, 7);                                                     // (diagnostic build: the global-state DP's share)
    return n;
}

// The same DP for windows with ANY number of edges (dense graphs: a window of 30 positions has ~600 edges) and up to ISPR_WIDE
// positions (the LDS copy is given up for the call: its arrays, or the SelWide overlay, hold the state): the state (distance, predecessor, reached) of every window position lives in LDS, the rows are not staged at
// all - they are one contiguous run of the topologically ordered copy and stream through the lanes 64 edges at a time; the
// sources a block of edges belongs to (~3 of them at out-degree 21) are relaxed one after the other from registers, so a
// source costs LDS round trips and 64 edges cost one global load.  (On global state a call took 137 us on the C5 graphs.)
struct SelStream { Dist *dist; int32_t *excl, *vj; uint8_t *pre, *reach; };   // the state arrays of one streamed DP: X->field[i] below
AASM_DEV int32_t sel_ispr_stream(SelCtx &s, const SelStream *X, int32_t a, int32_t bd, bool wl_flag, int32_t wl, int32_t pa, int32_t pb) {
    const WS &w = *s.w;
    const int32_t W = pb - pa;
    const int32_t *order = w.fwd_order + s.vb;
    s.res_lds = false;
    SPROF(s, 5);
    K9STAT(9, 1);
    wave_lds_sync();
    const int64_t e_start = uni(w.tp_ptr[s.vb + pa]);
    for (int32_t t = s.lane; t <= W; t += AASM_WAVE) { X->excl[t] = (int32_t)(w.tp_ptr[s.vb + pa + t] - e_start); X->reach[t] = (t == 0) ? 1 : 0; }
    for (int32_t t = s.lane; t < W; t += AASM_WAVE) X->vj[t] = w.tp_vj[s.vb + pa + t];
    if (s.lane == 0) { X->dist[0] = dist_zero(); X->pre[0] = 0; }
    wave_lds_sync();
    const int32_t T = uni(X->excl[W]);
    int32_t t = 0;                                                   // the source whose row the stream is in
    for (int32_t b0 = 0; b0 < T; b0 += AASM_WAVE) {
        const int32_t idx = b0 + s.lane, bend = (T - b0 < AASM_WAVE) ? T : b0 + AASM_WAVE;
        int32_t tg = -1, wr = 0;
        int64_t wq = 0;
        uint8_t fl = 0;
        if (idx < T) {
            const I4 pk = w.te_pk[e_start + idx];
            const int32_t rel = pk.x - pa;
            tg = rel > W ? -1 : rel;                                 // (targets behind the window's end are never expanded and never on the path)
            wq = te_wq(pk); wr = te_wr(pk); fl = te_fl(pk);
        }
        while (t < W) {
            const int32_t r0 = uni(X->excl[t]), r1 = uni(X->excl[t + 1]);
            if (r0 >= bend) break;                                   // the row starts behind this block
            const int32_t e0 = r0 > b0 ? r0 : b0, e1 = r1 < bend ? r1 : bend;
            if (e1 > e0) {
                // the source's state and - the heads are in registers - the state of every head of the block, in ONE LDS round trip
                const int32_t tgc = tg >= 0 ? tg : 0;
                Dist cd = X->dist[t], td = X->dist[tgc];
                int32_t sr = X->reach[t], tr = X->reach[tgc], svj = X->vj[t];
                keep_load(cd.qry); keep_load(td.qry); keep_load(sr); keep_load(tr); keep_load(svj);
                if (uni(sr)) {
                    const bool to_dest_ok = !wl_flag || (uni(svj) == wl);    // :767-773 (src / dest have vj < 0)
                    if (r0 >= b0) s.n_ispr_v++;
                    s.n_ispr_e += e1 - e0;
                    if (idx >= e0 && idx < e1 && tg >= 0 && !(tg == W && !to_dest_ok)) {
                        const Dist nd = dist_add(cd, edge_dist(wq, wr, fl));
                        if (!tr || dist_lt<QRY_SCORE_MODE>(nd, td)) { X->dist[tg] = nd; X->pre[tg] = (uint8_t)t; X->reach[tg] = 1; }
                    }
                    wave_lds_sync();
                }
            }
            if (r1 <= bend) t++; else break;                         // (the row goes on in the next block)
        }
    }
    if (!uni((int32_t)X->reach[W])) { s.err = true; return -1; }     // :783
    s.res_ppos = pa + uni((int32_t)X->pre[W]);
    int32_t n = 0, last = W;
    while (last != 0) {
        if (n >= s.cap) { s.err = true; return -1; }
        const int32_t pv = uni((int32_t)X->pre[last]);
        if (s.lane == 0) { s.pathT[2 * n] = order[pa + pv]; s.pathT[2 * n + 1] = (last == W) ? bd : order[pa + last]; }
        n++;
        last = pv;
    }
    wave_fence();
    SPROF(s, 4);                                                     // (diagnostic build: the streamed DP's share)
    return n;
}

// The LDS copy of the topologically ordered CSR that the window DPs run on is a CACHE: a refill takes up to ISPR_CW
// consecutive positions from the window's start (as many as ISPR_MAX_E edges allow), and the calls that follow - the path
// moves forward in topological order, a few positions per edge - find their windows inside it: two dependent global round
// trips per ~20 calls instead of three per call (the staging was 25-32 % of a conversion's cycles).
#define ISPR_CW 63
AASM_DEV bool sel_cw_fill(SelCtx &s, int32_t pa, int32_t pb) {      // false: the window [pa, pb] itself does not fit
    const WS &w = *s.w;
    SelLds *L = (SelLds *)s.lds;
    int32_t n = (int32_t)(s.V - pa);                                 // positions of the contig from pa on
    if (n > ISPR_CW) n = ISPR_CW;
    K9STAT(14, 1);
    wave_lds_sync();                                                 // (earlier calls have read the arrays)
    // round 1: the positions (lane t: position pa + t; lane n: the end of the last row)
#if defined(AASM_HOST_EMUL)
    const int64_t e_start = w.tp_ptr[s.vb + pa];
    int32_t m = 0;
    for (int32_t t = 0; t <= n && w.tp_ptr[s.vb + pa + t] - e_start <= ISPR_MAX_E; t++) m = t;   // row ends that still fit (the offsets ascend)
    if (m < pb - pa) { s.cw_pa = -1; return false; }
    for (int32_t t = 0; t <= m; t++) L->excl[t] = (int32_t)(w.tp_ptr[s.vb + pa + t] - e_start);
    for (int32_t t = 0; t < m; t++) { L->u[t] = w.fwd_order[s.vb + pa + t]; L->vj[t] = w.tp_vj[s.vb + pa + t]; }
    const int32_t T = L->excl[m];
#else
    int64_t ptr = 0;
    int32_t uu = 0, vv = 0;
    if (s.lane <= n) ptr = w.tp_ptr[s.vb + pa + s.lane];
    if (s.lane < n) { uu = w.fwd_order[s.vb + pa + s.lane]; vv = w.tp_vj[s.vb + pa + s.lane]; }
    const int64_t e_start = wave_bcast(ptr, 0);
    const int64_t off = ptr - e_start;
    const uint64_t fit = wave_ballot(s.lane <= n && off <= ISPR_MAX_E);   // row ends that still fit (a prefix: the offsets ascend)
    const int32_t m = 63 - __builtin_clzll(fit | 1ull);             // positions kept: rows 0 .. m - 1
    if (m < pb - pa) { s.cw_pa = -1; return false; }
    if (s.lane <= m) L->excl[s.lane] = (int32_t)off;
    if (s.lane < m) { L->u[s.lane] = uu; L->vj[s.lane] = vv; }
    const int32_t T = (int32_t)wave_bcast(off, m);
#endif
    // round 2: all their edges, one contiguous run of the topologically ordered copy
    for (int32_t idx = s.lane; idx < T; idx += AASM_WAVE) {
        const I4 pk = w.te_pk[e_start + idx];
        int32_t rel = pk.x - pa;
        if (rel > m) rel = -1;                                       // beyond every window this copy can serve
        L->tgt[idx] = (int8_t)rel; L->wq[idx] = te_wq(pk); L->wr[idx] = te_wr(pk); L->fl[idx] = te_fl(pk);
    }
    s.cw_pa = pa; s.cw_n = m;
    wave_lds_sync();
    return true;
}

// ---- the path window of the step loop ---------------------------------------------------
// Edge `it` of pathA with its classification.  Device: 64 consecutive edges live in the lanes' registers (lane t: edge
// pa_base + t) and an edge is read with v_readlane - no LDS round trip per question; a reload is one 16-byte round trip per
// lane.  The 1-lane emulation reads the two arrays directly.
struct SelEdge { int32_t u, v, pv, vj; bool single, settled; };
#if !defined(AASM_HOST_EMUL)
AASM_DEV void sel_win_load(SelCtx &s, int32_t it, int32_t la) {
    s.pa_base = it;
    s.pa_n = (la - it < AASM_WAVE) ? (la - it) : AASM_WAVE;
    int32_t u = -1, v = -1, pv = -1, vf = 0;
    if (s.lane < s.pa_n) {
        u = s.pathA[2 * (it + s.lane)]; v = s.pathA[2 * (it + s.lane) + 1];
        const int64_t c = s.cls[it + s.lane];
        pv = (int32_t)(uint32_t)(uint64_t)c; vf = (int32_t)((uint64_t)c >> 32);
    }
    s.wu = u; s.wv = v; s.wpv = pv; s.wvf = vf;
}
#endif
AASM_DEV SelEdge sel_edge(SelCtx &s, int32_t it) {                   // (device: `it` inside the loaded window)
    SelEdge e;
#if defined(AASM_HOST_EMUL)
    e.u = s.pathA[2 * it]; e.v = s.pathA[2 * it + 1];
    const int64_t c = s.cls[it];
    e.pv = (int32_t)(uint32_t)(uint64_t)c;
    const int32_t vf = (int32_t)((uint64_t)c >> 32);
#else
    const int32_t l = it - s.pa_base;
    e.u = __builtin_amdgcn_readlane(s.wu, l); e.v = __builtin_amdgcn_readlane(s.wv, l); e.pv = __builtin_amdgcn_readlane(s.wpv, l);
    const int32_t vf = __builtin_amdgcn_readlane(s.wvf, l);
#endif
    e.vj = vf >> 2; e.single = (vf & 1) != 0; e.settled = (vf & 2) != 0;
    return e;
}

// append the result of the LDS / streamed / global-state DP (reverse order in L->res or pathT), optionally without its last edge
AASM_DEV void sel_append_alt(SelCtx &s, int32_t n, bool drop_last, int32_t pb) {
    SelLds *L = (SelLds *)s.lds;
    for (int32_t t = n - 1; t >= (drop_last ? 1 : 0); t--) {
        int32_t u, v;
        if (s.res_lds) { u = uni(L->res[2 * t]); v = uni(L->res[2 * t + 1]); }
        else { u = uni(s.pathT[2 * t]); v = uni(s.pathT[2 * t + 1]); }
        sel_push(s, u, v, -1);
    }
    if (n > (drop_last ? 1 : 0)) s.last_pos = drop_last ? s.res_ppos : pb;
}

#if !defined(AASM_HOST_EMUL)
// The window DP with every edge of the window in a lane's registers (W <= ISPR_REG_W positions, <= 64 edges, inside the LDS
// copy: o = first position relative to the copy).  Lane i holds edge e0 + i of the window's contiguous run of rows: its
// source position (how many row starts lie at or before it), its head, its weight; cd = dist[source] + weight once the source
// is final.  Position p = 1 .. W becomes final by folding the candidates of its in-edges IN LANE ORDER - ascending source
// position, the order the reference relaxes them in (:761-779; the heads of one row are distinct) - with the strict `<` of
// QRY_SCORE mode: first wins.  The state is wave-uniform and never leaves registers: two LDS round trips (row starts, the
// lane's edge) + one for the result's vertex ids, against ~4 per SOURCE on LDS state.  The result goes straight into the
// write buffer, in path order.  Returns the number of result edges (before drop_last), -1 on error, -2: not this form.
#define ISPR_REG_W 16
AASM_DEV int32_t sel_ispr_reg(SelCtx &s, int32_t o, int32_t W, int32_t pa, int32_t bd, bool wl_flag, int32_t wl, bool drop_last) {
    SelLds *L = (SelLds *)s.lds;
    // row starts of positions o .. o + W (lane t <= W), relative to the first
    const int32_t xs_abs = L->excl[o + (s.lane <= W ? s.lane : W)];
    const int32_t e0 = __builtin_amdgcn_readfirstlane(xs_abs);
    const int32_t xs = xs_abs - e0;
    const int32_t ne = __builtin_amdgcn_readlane(xs, W);
    if (ne > AASM_WAVE) return -2;
    const bool have = s.lane < ne;
    const int32_t idx = e0 + (have ? s.lane : 0);
    const int32_t tc = L->tgt[idx];                                  // head relative to the copy's first position (-1: beyond the copy)
    const int64_t wq = L->wq[idx];
    const int32_t wr = L->wr[idx];
    const uint8_t fl = L->fl[idx];
    int32_t src = 0;                                                 // source position of the lane's edge, relative to the window
    for (int32_t t = 1; t < W; t++) src += (s.lane >= __builtin_amdgcn_readlane(xs, t)) ? 1 : 0;
    const int32_t tg = tc - o;
    bool valid = have && tc >= 0 && tg <= W;                         // (heads behind the window's end are never expanded and never on the path)
    if (wl_flag) {                                                   // :767-773: the hop into the target only from a vertex of record wl (src / dest have vj < 0)
        const int32_t svj = L->vj[o + src];
        valid = valid && !(tg == W && svj != wl);
    }
    const Dist ew = edge_dist(wq, wr, fl);
    Dist cd = ew;                                                    // sources at position 0: dist 0 + weight
    bool live = src == 0;
    int32_t prev = 0;                                                // lane p: predecessor position of position p
    uint32_t reach = 1u;
    for (int32_t p = 1; p <= W; p++) {
        uint64_t m = wave_ballot(valid && live && tg == p);
        if (!m) continue;
        int j = ffs64(m) - 1;
        m &= m - 1;
        LaneArr<Dist> ca; ca.r = cd;
        Dist best = la_get_dist(ca, j);
        int32_t bsrc = __builtin_amdgcn_readlane(src, j);
        while (m) {
            j = ffs64(m) - 1;
            m &= m - 1;
            const Dist c2 = la_get_dist(ca, j);
            if (dist_lt<QRY_SCORE_MODE>(c2, best)) { best = c2; bsrc = __builtin_amdgcn_readlane(src, j); }
        }
        reach |= 1u << p;
        if (s.lane == p) prev = bsrc;
        if (src == p) { cd = dist_add(best, ew); live = true; }
    }
    s.n_ispr_v += popc64((uint64_t)(reach & ((1u << W) - 1u)));
    s.n_ispr_e += popc64(wave_ballot(have && ((reach >> src) & 1u)));
    if (!((reach >> W) & 1u)) { s.err = true; return -1; }          // :783
    // back along the predecessors: lane k = the k-th edge from the END, (from, to) as window positions
    int32_t n = 0, last = W, cfrom = 0, cto = 0;
    while (last != 0) {
        const int32_t pv = __builtin_amdgcn_readlane(prev, last);
        if (s.lane == n) { cfrom = pv; cto = last; }
        n++;
        last = pv;
    }
    const int32_t m_out = drop_last ? n - 1 : n;
    if (m_out > 0) {
        const int32_t r = n - 1 - (s.lane < m_out ? s.lane : 0);     // lane k < m_out appends edge k of the path = reverse index n - 1 - k
        const int32_t pf = __shfl(cfrom, r, 64), pt = __shfl(cto, r, 64);
        const int32_t uf = L->u[o + pf];
        const int32_t ut = (pt == W) ? bd : L->u[o + (pt < W ? pt : 0)];
        sel_push_lanes(s, 0, m_out, uf, ut);
        s.last_head = __builtin_amdgcn_readlane(ut, m_out - 1);
        s.last_pos = pa + __builtin_amdgcn_readlane(pt, m_out - 1);
    }
    return n;
}
#endif

// internal_shortest_path_recover(a, bd) with the result APPENDED to the output (without its last edge if drop_last).
// pa / pb: topological positions of a / bd.  Returns the number of result edges (0 when a == bd: the caller appends the
// original edges), -1 on "must not happen".
AASM_DEV int32_t sel_ispr(SelCtx &s, int32_t a, int32_t pa, int32_t bd, int32_t pb, bool wl_flag, int32_t wl, bool drop_last) {
    K9STAT(6, 1);
    if (a == bd) { K9STAT(11, 1); return 0; }
    const int32_t W = pb - pa;
    if (W <= 0) { s.err = true; return -1; }
    int32_t n;
    if (W > ISPR_CW) {
        if (W > ISPR_WIDE) n = sel_ispr_generic(s, a, bd, wl_flag, wl, pa, pb);
        else {
            SelWide *X = (SelWide *)(s.lds + offsetof(SelLds, wq));  // over the LDS copy and the narrow DP's arrays
            s.cw_pa = -1;
            const SelStream st{X->dist, X->excl, X->vj, X->pre, X->reach};
            n = sel_ispr_stream(s, &st, a, bd, wl_flag, wl, pa, pb);
        }
        if (n > 0) sel_append_alt(s, n, drop_last, pb);
        return n;
    }
    SPROF(s, 5);
    SelLds *L = (SelLds *)s.lds;
    if (!(s.cw_pa >= 0 && pa >= s.cw_pa && pb <= s.cw_pa + s.cw_n) && !sel_cw_fill(s, pa, pb)) {   // too many edges for the copy (which is given up: cw_pa = -1)
        const SelStream st{L->dist, L->excl, L->vj, (uint8_t *)L->pre, L->reach};
        n = sel_ispr_stream(s, &st, a, bd, wl_flag, wl, pa, pb);
        if (n > 0) sel_append_alt(s, n, drop_last, pb);
        return n;
    }
    const int32_t o = pa - s.cw_pa;                                  // window position t = cached position o + t
    SPROF(s, 2);                                                     // ISPR staging
#if !defined(AASM_HOST_EMUL)
    if (W <= ISPR_REG_W) {
        n = sel_ispr_reg(s, o, W, pa, bd, wl_flag, wl, drop_last);
        if (n != -2) { SPROF(s, 3); return n; }
    }
#endif
    K9STAT(8, 1); K9STAT(15, W);
    for (int32_t t = s.lane; t <= W; t += AASM_WAVE) L->reach[t] = (t == 0) ? 1 : 0;
    if (s.lane == 0) { L->dist[0] = dist_zero(); L->pre[0] = -1; }
    wave_lds_sync();
    // DP over the window, source by source
    for (int32_t t = 0; t < W; t++) {
        if (!uni((int32_t)L->reach[t])) continue;
        const Dist cd = L->dist[t];
        const int32_t e0 = uni(L->excl[o + t]), e1 = uni(L->excl[o + t + 1]);
        s.n_ispr_v++; s.n_ispr_e += e1 - e0;
        const bool to_dest_ok = !wl_flag || (uni(L->vj[o + t]) == wl);   // :767-773 (src / dest have vj < 0)
        for (int32_t idx = e0 + s.lane; idx < e1; idx += AASM_WAVE) {
            const int32_t tc = L->tgt[idx];
            const int32_t tg = tc - o;
            if (tc < 0 || tg > W || (tg == W && !to_dest_ok)) continue;   // (targets behind the window's end are never expanded and never on the path)
            const Dist nd = dist_add(cd, edge_dist(L->wq[idx], L->wr[idx], L->fl[idx]));
            if (!L->reach[tg] || dist_lt<QRY_SCORE_MODE>(nd, L->dist[tg])) { L->dist[tg] = nd; L->pre[tg] = (int8_t)t; L->reach[tg] = 1; }
        }
        wave_lds_sync();
    }
    SPROF(s, 3);                                                     // ISPR DP on LDS
    if (!uni((int32_t)L->reach[W])) { s.err = true; return -1; }     // :783
    s.res_ppos = pa + uni((int32_t)L->pre[W]);
    n = 0;
    int32_t last = W;
    while (last != 0) {
        if (n >= 64) { s.err = true; return -1; }
        const int32_t pv = uni((int32_t)L->pre[last]);
        if (s.lane == 0) { L->res[2 * n] = L->u[o + pv]; L->res[2 * n + 1] = (last == W) ? bd : L->u[o + last]; }
        n++;
        last = pv;
    }
    s.res_lds = true;
    wave_lds_sync();
    SPROF(s, 4);                                                     // ISPR backtrack
    sel_append_alt(s, n, drop_last, pb);
    return n;
}

// upgrade_edge_path_with_alt_path (paf_data.cpp:795-921): pathA[la] (classified: s.cls) -> pathB; returns lb
AASM_DEV int32_t sel_upgrade(SelCtx &s, int32_t la) {
    const WS &w = *s.w;
    sel_out_begin(s, s.pathB);
    s.pa_base = 0; s.pa_n = 0;
    s.last_head = -1; s.last_pos = -1;
    const int32_t pos_src = uni(w.fwd_pos[s.vb + s.src]);
    for (int32_t it = 0; it < la && !s.err; ++it) {
#if !defined(AASM_HOST_EMUL)
        // the window holds edge `it` and, if there is one, its successor
        if (it < s.pa_base || it >= s.pa_base + s.pa_n || (it + 1 == s.pa_base + s.pa_n && it + 1 < la)) { SPROF(s, 5); sel_win_load(s, it, la); SPROF(s, 7); }
#endif
        const SelEdge e = sel_edge(s, it);
        // A run of edges whose single-record heads are all settled (the other heads are pushed as they are
        // anyway, whatever the continuation vertex: :866-873) comes out of the steps below exactly as it went in - whether a
        // step takes one edge (:812-833) or two (:834-843) - as long as the continuation vertex is the edge tail at every
        // settled head, which it stays inside such a run.  The part of the run inside the window is copied in one piece.
        if (it > 0 && e.v != s.dest && (!e.single || (e.settled && s.last_head == e.u))) {
#if defined(AASM_HOST_EMUL)
            int32_t run = 0;
            for (int32_t t = it; t < la && run < 64; t++, run++) {
                const SelEdge x = sel_edge(s, t);
                if ((x.single && !x.settled) || x.v == s.dest) break;
            }
            if (run >= 2) {
                for (int32_t t = it; t < it + run; t++) { const SelEdge x = sel_edge(s, t); sel_push(s, x.u, x.v, x.pv); }
                K9STAT(2, run);
                it += run - 1;
                continue;
            }
#else
            const int32_t l = it - s.pa_base;
            const uint64_t stop = wave_ballot(s.lane >= l && s.lane < s.pa_n && ((((s.wvf & 1) != 0) && !(s.wvf & 2)) || s.wv == s.dest));   // edges the steps have to look at
            const int32_t first = stop ? ffs64(stop) - 1 : s.pa_n;
            const int32_t run = first - l;
            if (run >= 2) {
                sel_push_lanes(s, l, run, s.wu, s.wv);
                s.last_head = __builtin_amdgcn_readlane(s.wv, l + run - 1);
                s.last_pos = __builtin_amdgcn_readlane(s.wpv, l + run - 1);
                it += run - 1;
                continue;
            }
#endif
        }
        const bool from_src = (e.u == s.src);
        K9STAT(3, 1);
        if (from_src || e.v != s.dest) {
            int32_t start, start_pos;
            if (from_src) { start = e.u; start_pos = pos_src; }       // :804
            else {
                if (s.out_n == 0) { s.err = true; break; }
                if (!e.single) { K9STAT(4, 1); sel_push(s, e.u, e.v, e.pv); continue; }      // :866-873
                start = s.last_head;                                 // continuation_src (:863)
                start_pos = s.last_pos;
            }
            if (start != e.u) K9STAT(12, 1);
            const int32_t y = e.vj;
            if (it + 1 >= la) { s.err = true; break; }
            const SelEdge ne = sel_edge(s, it + 1);
            const bool nv_single = (ne.v == s.dest) || ne.single;
            const bool known = e.settled && start == e.u;            // the call would return (u, v), (v, nv): see sel_classify_edge
            if (known) K9STAT(5, 1);
            if (!known && start_pos < 0) start_pos = uni(w.fwd_pos[s.vb + start]);
            if (nv_single) {                                         // :812-833 / :879-899
                if (known) sel_push(s, e.u, e.v, e.pv);              // the result without its last edge
                else {
                    const int32_t n = sel_ispr(s, start, start_pos, ne.v, ne.pv, true, y, true);
                    if (n < 0) break;
                    if (n == 0) sel_push(s, e.u, e.v, e.pv);
                }
            } else {                                                 // :834-843 / :900-909
                if (known) { sel_push(s, e.u, e.v, e.pv); sel_push(s, ne.u, ne.v, ne.pv); }
                else {
                    const int32_t n = sel_ispr(s, start, start_pos, ne.v, ne.pv, false, -1, false);
                    if (n < 0) break;
                    if (n == 0) { sel_push(s, e.u, e.v, e.pv); sel_push(s, ne.u, ne.v, ne.pv); }
                }
                ++it;
            }
        } else {                                                     // v == dest (:845-858)
            K9STAT(13, 1);
            if (s.out_n == 0) { s.err = true; break; }
            int32_t start_pos = s.last_pos;
            if (start_pos < 0) start_pos = uni(w.fwd_pos[s.vb + s.last_head]);
            const int32_t n = sel_ispr(s, s.last_head, start_pos, e.v, e.pv, false, -1, false);
            if (n < 0) break;
        }
    }
    sel_out_flush(s);
    return s.out_n;
}

AASM_DEV OutElem out_from_rec(const WS &w, int64_t g) {             // PafOutputData(rec), paf_data.hpp:101-104
    OutElem o; o.qs = w.s_qs[g]; o.qe = w.s_qe[g]; o.rs = w.s_rs[g]; o.re = w.s_re[g]; o.ctg_index = w.s_orig[g]; o.is_alt = 0;
    return o;
}

// edge_path_to_paf_path (paf_data.cpp:1489-1568), second half: the classified pathA[la] -> upgrade -> cur_out; returns
// #elements, coverage in cov.
// The tp flag (:1560-1566) depends on every path converted BEFORE this one (hazard B5):
// mark_time[r] = smallest conversion ordinal whose un-upgraded path touched record r (sel_classify_edge), and an
// element of conversion `ord` is_alt iff mark_time[r] > ord.  This leaves the SORTED
// record index in is_alt; copy_resolved() turns it into the flag.
// element t of the upgraded path pathB[lb] (:1503-1557 as a map over path edges): edge t = (u, v) emits record cur(v) (nothing for
// v == dest, which is the last edge); a pair vertex v = (x, y) clips the start of its own element to edited_loc_str[x][y] and the
// END of the previous element to edited_loc_pre_end[x][y], i.e. element t takes its end clip from edge t + 1.  Returns the
// element's coverage term (get_total_coverage, :1571-1579), -1 for a path that passes through dest (must not happen).
AASM_DEV int64_t sel_emit_elem(const WS &w, int64_t vb, int64_t b, int32_t dest, const int32_t *pathB, int32_t t, OutElem *out) {
    const int32_t v = pathB[2 * t + 1], nv = pathB[2 * (t + 1) + 1];
    if (v == dest) return -1;
    const int32_t y1 = w.v_i[vb + v], y2 = w.v_j[vb + v];
    OutElem o = out_from_rec(w, b + y2);
    o.is_alt = y2;                                                   // resolved by copy_resolved()
    if (y1 != y2) { const int64_t sl = w.v_slot[vb + v]; o.qs = w.ov_stq[sl]; o.rs = w.ov_str[sl]; }
    if (nv != dest && w.v_i[vb + nv] != w.v_j[vb + nv]) { const int64_t sl = w.v_slot[vb + nv]; o.qe = w.ov_peq[sl]; o.re = w.ov_per[sl]; }
    out[t] = o;
    return (o.qe - o.qs) + (o.re > o.rs ? o.re - o.rs : o.rs - o.re);
}
AASM_DEV int32_t sel_upgrade_emit(SelCtx &s, int32_t la, int64_t &cov) {
    const WS &w = *s.w;
    OutElem *out = s.cur;
    cov = 0;
    const int32_t lb = sel_upgrade(s, la);                           // :1500-1501
    wave_fence();
    SPROF(s, 5);                                                     // upgrade: control + appends (ISPR parts stamped inside)
    if (s.err) return 0;
    s.n_path_e += la + lb;
    // (From here on the bytes of s.cls belong to the elements.)
    if (lb < 2 || s.pathB[2 * (lb - 1) + 1] != s.dest) { s.err = true; return 0; }
    const int32_t n = lb - 1;
    if (n > s.N) { s.err = true; return 0; }
    int64_t part = 0;
    bool bad = false;
    for (int32_t t = s.lane; t < n; t += AASM_WAVE) {
        const int64_t cv = sel_emit_elem(w, s.vb, s.b, s.dest, s.pathB, t, out);
        if (cv < 0) bad = true; else part += cv;
    }
    if (wave_ballot(bad)) { s.err = true; return 0; }
    cov = wave_sum(part);
    SPROF(s, 6);                                                     // conversion
    if (s.lane == 0) atomic_add(&w.counters[CNT_CONVERTED], (int64_t)1);
    s.n_out_e += n;
    return n;
}

// the whole conversion of walk k by ONE wave (kb_select; the parallel form runs the three stages as three kernels)
AASM_DEV int32_t sel_convert(SelCtx &s, int32_t kidx, int32_t ord, int64_t &cov) {
    const WS &w = *s.w;
    cov = 0;
    SPROF(s, 7);
    const int32_t la = sel_recover(s, kidx);
    if (la <= 0) { s.err = true; return 0; }
    K9STAT(0, 1); K9STAT(1, la);
    wave_fence();
    SPROF(s, 0);                                                     // recover
    for (int32_t t = s.lane; t < la; t += AASM_WAVE) sel_classify_edge(w, s.vb, s.dest, s.pathA, la, t, w.mark_time + s.b, ord, s.cls);
    wave_fence();
    SPROF(s, 1);                                                     // classification + marking
    return sel_upgrade_emit(s, la, cov);
}

AASM_DEV void copy_resolved(OutElem *dst, const OutElem *src, int32_t n, const int32_t *mark, int32_t ord, int lane) {
    for (int32_t t = lane; t < n; t += AASM_WAVE) { OutElem o = src[t]; o.is_alt = (mark[o.is_alt] > ord) ? 1 : 0; dst[t] = o; }
}
AASM_DEV bool dist_sel_equal(const Dist &a, const Dist &b) { return a.qry + a.ref == b.qry + b.ref && a.anom == b.anom; }   // :1581-1583

// ---- K9 as plan -> convert -> final ----------------------------------------------------
// WHICH paths get converted, and in what order, is a pure function of the K distances
// (paf_data.cpp:1596-1649: tie run, then alt candidates), so the expensive conversions
// (recover + upgrade + clip) run as independent waves and the per-contig tail of the
// sequential form disappears.  kinds: 0 main, 1 tie, 2 alt (new best ratio), 3 alt (equal).
AASM_DEV bool sel_has_graph(const WS &w, int64_t c) {
    return (w.rec_off[c + 1] - w.rec_off[c]) > 1 && w.status[c] == 0 && w.kfound[c] > 0;
}
// One wave per contig scans the K distances 64 at a time (ballot): the tie run is the maximal
// prefix of paths equal to the best one (:1596-1611); alt candidates are the paths with fewer
// anomalies (:1613-1649), visited in index order because the "best ratio so far" is sequential.
// fill == false only counts; fill == true writes the conversion records at conv_off[c].
#define SEL_PLAN_KEEP 64
AASM_DEV int32_t sel_plan_wave(const KCtx &k, const WS &w, int64_t c, bool fill) {
    const int32_t found = w.kfound[c];
    const Dist *kd = w.kd + c * (int64_t)w.K;
    const Dist mind = kd[0];
    const int64_t j0 = fill ? w.conv_off[c] : 0;
    const int32_t N = (int32_t)(w.rec_off[c + 1] - w.rec_off[c]), V = w.ctgV[c];
    int32_t *keep = w.plan_kk + c * (int64_t)(2 * SEL_PLAN_KEEP);
    auto put = [&](int32_t ord, int32_t kidx, int32_t kind) {       // the counting pass remembers the first SEL_PLAN_KEEP, the filling pass writes the records
        if (!fill) { if (ord < SEL_PLAN_KEEP) { keep[2 * ord] = kidx; keep[2 * ord + 1] = kind; } return; }
        const int64_t j = j0 + ord;
        w.cv_ctg[j] = (int32_t)c; w.cv_k[j] = kidx; w.cv_ord[j] = ord; w.cv_kind[j] = kind; w.cv_szr[j] = N + 2; w.cv_szv[j] = V;
    };
    int32_t n = 0;
    if (k.lane == 0) put(0, 0, 0);
    n = 1;
    for (int32_t base = 1; base < found; base += AASM_WAVE) {       // tie run
        const int32_t idx = base + k.lane;
        const bool in = idx < found;
        const bool eq = in && dist_sel_equal(mind, kd[idx]);
        const uint64_t m = wave_ballot(eq), full = wave_ballot(in);
        const int lead = (~m == 0ull) ? 64 : (ffs64(~m) - 1);
        if (k.lane < lead) put(n + k.lane, idx, 1);
        n += lead;
        if (lead < popc64(full)) break;
    }
    if (found >= 2 && mind.anom != w.anom_dest[c]) {
        int64_t ans_up = 0, ans_down = 0;
        int32_t ans_idx = -1;
        Dist ans_d = mind;
        for (int32_t base0 = 1; base0 < found; base0 += 4 * AASM_WAVE) {   // four loads of 64 distances in flight per round; the candidates are then taken
            LaneArr<Dist> dl[4];                                            // from the lanes' registers, not re-read one by one (K = 10 000: thousands of them per contig)
            AASM_UNROLL
            for (int j = 0; j < 4; j++) {
                const int32_t idx = base0 + j * AASM_WAVE + k.lane;
                Dist x = dist_zero(); x.anom = INT32_MAX;
                if (idx < found) x = kd[idx];
                dl[j].at(0) = x;
            }
            AASM_UNROLL
            for (int j = 0; j < 4; j++) {
                const int32_t base = base0 + j * AASM_WAVE;
                uint64_t m = wave_ballot(dl[j].at(0).anom < mind.anom);
                while (m) {
                    const int bit = ffs64(m) - 1;
                    m &= m - 1;
                    const int32_t i = base + bit;
                    const Dist dd = la_get_dist(dl[j], bit);
                    const int64_t up = (dd.qry + dd.ref) - (mind.qry + mind.ref), down = (int64_t)mind.anom - dd.anom;
                    int32_t kind = -1;
                    if (ans_idx == -1 || up * ans_down < down * ans_up) { ans_up = up; ans_down = down; ans_idx = i; ans_d = dd; kind = 2; }
                    else if (dist_sel_equal(dd, ans_d)) kind = 3;
                    if (kind >= 0) { if (k.lane == 0) put(n, i, kind); n++; }
                }
            }
        }
    }
    return n;
}
AASM_DEV void kb_sel_plan(const KCtx &k, const WS &w) {             // one wave per contig
    const int64_t c = k.bid;
    const int32_t n = sel_has_graph(w, c) ? sel_plan_wave(k, w, c, false) : 0;
    if (k.lane == 0) w.nconv[c] = n;
}
AASM_DEV void kb_sel_planfill(const KCtx &k, const WS &w) {         // one wave per contig
    const int64_t c = k.bid;
    if (!sel_has_graph(w, c)) return;
    const int32_t n = w.nconv[c];
    if (n > SEL_PLAN_KEEP) { sel_plan_wave(k, w, c, true); return; } // (more conversions than the counting pass kept: walk the K distances again)
    const int32_t N = (int32_t)(w.rec_off[c + 1] - w.rec_off[c]), V = w.ctgV[c];
    const int32_t *keep = w.plan_kk + c * (int64_t)(2 * SEL_PLAN_KEEP);
    const int64_t j0 = w.conv_off[c];
    for (int32_t t = k.lane; t < n; t += AASM_WAVE) {
        const int64_t j = j0 + t;
        w.cv_ctg[j] = (int32_t)c; w.cv_k[j] = keep[2 * t]; w.cv_ord[j] = t; w.cv_kind[j] = keep[2 * t + 1]; w.cv_szr[j] = N + 2; w.cv_szv[j] = V;
    }
}

AASM_DEV void sel_ctx_init(SelCtx &s, const KCtx &k, const WS &w, int64_t c) {
    const int64_t gb = w.rec_off[c];
    s.w = &w; s.c = c; s.b = gb - w.R0; s.N = w.rec_off[c + 1] - gb; s.V = w.ctgV[c]; s.vb = w.voff[c]; s.cap = s.N + 2;
    s.src = (int32_t)(s.V - 2); s.dest = (int32_t)(s.V - 1);
    s.epoch = 0; s.last_head = -1; s.last_pos = -1; s.res_ppos = -1; s.err = false; s.res_lds = false; s.out_dst = nullptr; s.out_n = s.out_flushed = 0;
    s.pa_base = 0; s.pa_n = 0; s.wu = s.wv = s.wpv = s.wvf = 0; s.cls = nullptr;
    s.lane = k.lane; s.lds = k.lds; s.n_ispr_e = s.n_ispr_v = s.n_path_e = s.n_out_e = 0;
    s.cw_pa = -1; s.cw_n = 0;
}
AASM_DEV void sel_flush_counters(const SelCtx &s, const WS &w) {
    if (s.lane == 0) {
        atomic_add(&w.counters[CNT_ISPR_E], s.n_ispr_e); atomic_add(&w.counters[CNT_ISPR_V], s.n_ispr_v);
        atomic_add(&w.counters[CNT_PATH_E], s.n_path_e); atomic_add(&w.counters[CNT_OUT_E], s.n_out_e);
    }
}

// The conversion of one planned walk as three launches: (1) recovery, a chain of dependent loads, one wave per conversion
// with nothing but the write buffer in LDS; (2) classification + marking, a thread per path edge; (3) the sequential upgrade
// steps + the elements, one wave per conversion.
AASM_DEV void sel_conv_setup(SelCtx &s, const KCtx &k, const WS &w, int64_t j) {
    sel_ctx_init(s, k, w, w.cv_ctg[j]);
    const int64_t ro = w.cv_roff[j], vo = w.cv_voff[j];
    s.pathA = w.cv_path + 6 * ro; s.pathB = s.pathA + 2 * s.cap; s.pathT = s.pathB + 2 * s.cap;
    s.pre2 = w.cv_pre2 + vo; s.stamp = w.cv_stamp + vo; s.dist2 = w.cv_dist2 + vo;
    s.cur = w.cv_out + ro; s.cls = (int64_t *)s.cur;
}
AASM_DEV void kb_sel_recover(const KCtx &k, const WS &w) {          // one wave per conversion
    const int64_t j = k.bid;
    SelCtx s;
    sel_conv_setup(s, k, w, j);
    const int32_t la = sel_recover(s, w.cv_k[j]);
    if (k.lane == 0) w.cv_la[j] = (la <= 0 || s.err) ? -1 : la;
}
AASM_DEV void kb_sel_classify(const KCtx &k, const WS &w) {         // one workgroup per conversion, a thread per path edge
    const int64_t j = k.bid;
    const int32_t la = w.cv_la[j];
    if (la <= 0) return;
    const int64_t c = w.cv_ctg[j], vb = w.voff[c], ro = w.cv_roff[j];
    const int32_t dest = w.ctgV[c] - 1;
    const int64_t cap = (w.rec_off[c + 1] - w.rec_off[c]) + 2;
    (void)cap;
    int32_t *mark = w.mark_time + (w.rec_off[c] - w.R0);
    for (int32_t t = k.tid; t < la; t += k.nthreads) sel_classify_edge(w, vb, dest, w.cv_path + 6 * ro, la, t, mark, w.cv_ord[j], (int64_t *)(w.cv_out + ro));
}
AASM_DEV void kb_sel_convert(const KCtx &k, const WS &w) {          // one wave per conversion
    const int64_t j = k.bid;
    const int64_t c = w.cv_ctg[j];
    SelCtx s;
    sel_conv_setup(s, k, w, j);
#if defined(AASM_KPROF) && !defined(AASM_HOST_EMUL)
    for (int i_ = 0; i_ < 8; i_++) s.kp_acc[i_] = 0;
    __builtin_amdgcn_s_waitcnt(0); s.kp_t0 = (int64_t)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0);
#endif
    int64_t cov = 0;
    int32_t n = 0;
    const int32_t la = w.cv_la[j];
    if (la <= 0) s.err = true;
    else { K9STAT(0, 1); K9STAT(1, la); n = sel_upgrade_emit(s, la, cov); }
    if (k.lane == 0) { w.cv_n[j] = n; w.cv_cov[j] = cov; w.cv_err[j] = s.err ? 1 : 0; }
#if defined(AASM_KPROF) && !defined(AASM_HOST_EMUL)
    if (k.lane == 0 && w.cv_ord[j] == 0) for (int i_ = 0; i_ < 8; i_++) w.prof_sel[c * 8 + i_] = s.kp_acc[i_];
#endif
    (void)c;
    sel_flush_counters(s, w);
}

AASM_DEV void sel_all_append(const KCtx &k, const WS &w, int64_t c, const OutElem *src, int32_t n, const int32_t *mark, int32_t ord) {
    int64_t off = 0, r = 0;
    if (k.lane == 0) { off = (int64_t)atomic_add(&w.counters[CNT_POOL], (int64_t)n); r = (int64_t)atomic_add(&w.counters[CNT_AR], (int64_t)1); }
    off = wave_bcast(off, 0); r = wave_bcast(r, 0);
    if (off + n <= w.pool_cap && r < w.ar_cap) {
        copy_resolved(w.pool + off, src, n, mark, ord, k.lane);
        if (k.lane == 0) { w.ar_ctg[r] = (int32_t)c; w.ar_gen[r] = w.all_gen[c]; w.ar_seq[r] = w.all_seq[c]; w.ar_off[r] = off; w.ar_len[r] = n; }
    } else if (k.lane == 0) {
        atomic_add(&w.counters[CNT_OVF], (int64_t)1);               // demand keeps being counted; the host re-runs with an exact-size pool
    }
    if (k.lane == 0) w.all_seq[c] += 1;
    wave_fence();
}

AASM_DEV void kb_sel_final(const KCtx &k, const WS &w) {            // one wave per contig
    const int64_t c = k.bid;
    const int64_t gb = w.rec_off[c], N = w.rec_off[c + 1] - gb, b = gb - w.R0;
    const bool L0 = (k.lane == 0);
    if (L0) { w.main_len[c] = 0; w.alt_len[c] = 0; }
    if (N <= 0) return;
    if (N == 1) {                                                   // paf_data.cpp:235-239
        OutElem o; o.qs = w.in_qs[gb]; o.qe = w.in_qe[gb]; o.rs = w.in_rs[gb]; o.re = w.in_re[gb]; o.ctg_index = 0; o.is_alt = 0;
        if (L0) { w.main_out[b] = o; w.main_len[c] = 1; }
        return;
    }
    if (w.status[c] != 0) return;
    const int32_t nc = w.nconv[c];
    if (nc <= 0) { if (L0) set_status(w, c, -6); return; }                  // :732
    const int64_t j0 = w.conv_off[c];
    for (int32_t t = 0; t < nc; t++) if (w.cv_err[j0 + t]) { if (L0) set_status(w, c, -6); return; }
    const int32_t *mark = w.mark_time + b;
    // main + .all: the list is cleared whenever a strictly better coverage appears (:1603-1609)
    int32_t main_t = 0, alt_t = -1;
    int64_t max_cov = w.cv_cov[j0];
    int32_t ties_end = 1;
    while (ties_end < nc && w.cv_kind[j0 + ties_end] == 1) ties_end++;
    for (int32_t t = 1; t < ties_end; t++) if (w.cv_cov[j0 + t] > max_cov) { max_cov = w.cv_cov[j0 + t]; main_t = t; }
    for (int32_t t = main_t + 1; t < ties_end; t++)
        if (w.cv_cov[j0 + t] == max_cov) sel_all_append(k, w, c, w.cv_out + w.cv_roff[j0 + t], w.cv_n[j0 + t], mark, t);
    int64_t alt_cov = -1;                                           // :1613-1649
    for (int32_t t = ties_end; t < nc; t++) {
        const int32_t kind = w.cv_kind[j0 + t];
        if (kind == 2) { alt_t = t; alt_cov = w.cv_cov[j0 + t]; }
        else if (kind == 3 && w.cv_cov[j0 + t] > alt_cov) { alt_t = t; alt_cov = w.cv_cov[j0 + t]; }
    }
    copy_resolved(w.main_out + b, w.cv_out + w.cv_roff[j0 + main_t], w.cv_n[j0 + main_t], mark, main_t, k.lane);
    if (L0) w.main_len[c] = w.cv_n[j0 + main_t];
    if (alt_t >= 0) {
        copy_resolved(w.alt_out + b, w.cv_out + w.cv_roff[j0 + alt_t], w.cv_n[j0 + alt_t], mark, alt_t, k.lane);
        if (L0) w.alt_len[c] = w.cv_n[j0 + alt_t];
    }
}

AASM_DEV void kb_select(const KCtx &k, const WS &w) {               // one wave per contig
    // Control flow is wave-uniform: every lane runs the same scalar program (uniform loads;
    // identical stores), and the lanes fan out inside sel_ispr and the output copies.
    const int64_t c = k.bid;
    const int64_t gb = w.rec_off[c], N = w.rec_off[c + 1] - gb, b = gb - w.R0;
    const bool L0 = (k.lane == 0);
    if (L0) { w.main_len[c] = 0; w.alt_len[c] = 0; }
    if (N <= 0) return;
    if (N == 1) {                                                   // paf_data.cpp:235-239
        OutElem o; o.qs = w.in_qs[gb]; o.qe = w.in_qe[gb]; o.rs = w.in_rs[gb]; o.re = w.in_re[gb]; o.ctg_index = 0; o.is_alt = 0;
        if (L0) { w.main_out[b] = o; w.main_len[c] = 1; }
        return;
    }
    if (w.status[c] != 0) return;
    const int32_t found = w.kfound[c];
    if (found <= 0) { if (L0) set_status(w, c, -6); return; }               // :732
    SelCtx s;
    sel_ctx_init(s, k, w, c);
    const int64_t pb = 2 * (b + 2 * c);                             // (N+2) pairs per contig
    s.pathA = w.pathA + pb; s.pathB = w.pathB + pb; s.pathT = w.pathT + pb;
    s.pre2 = w.pre2 + s.vb; s.stamp = w.stamp + s.vb; s.dist2 = w.dist2 + s.vb;
#if defined(AASM_KPROF) && !defined(AASM_HOST_EMUL)
    __builtin_amdgcn_s_waitcnt(0);
    const int64_t kp_mt0 = (int64_t)__builtin_amdgcn_s_memtime(), kp_rt0 = (int64_t)__builtin_amdgcn_s_memrealtime();
    for (int i_ = 0; i_ < 8; i_++) s.kp_acc[i_] = 0;
    __builtin_amdgcn_s_waitcnt(0); s.kp_t0 = (int64_t)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0);
#endif
    s.cur = w.cur_out + b; s.cls = (int64_t *)s.cur;
    const int32_t *mark = w.mark_time + b;
    int32_t ord = 0;                                                // conversion ordinal (hazard B5)
    const Dist *kd = w.kd + c * (int64_t)w.K;
    OutElem *cur = w.cur_out + b, *mo = w.main_out + b, *ao = w.alt_out + b;
    const Dist mind = kd[0];                                        // :1585
    int64_t cov = 0, max_cov = 0;
    int32_t n = sel_convert(s, 0, ord, cov);                        // :1589-1593
    if (s.err) { if (L0) set_status(w, c, -6); return; }
    max_cov = cov;
    copy_resolved(mo, cur, n, mark, ord, k.lane);
    if (L0) w.main_len[c] = n;
    for (int32_t idx = 1; idx < found; idx++) {                     // ties, :1596-1611
        const Dist dd = kd[idx];
        if (!(mind.qry + mind.ref == dd.qry + dd.ref && mind.anom == dd.anom)) break;
        n = sel_convert(s, idx, ++ord, cov);
        if (s.err) { if (L0) set_status(w, c, -6); return; }
        if (cov > max_cov) {
            max_cov = cov;
            copy_resolved(mo, cur, n, mark, ord, k.lane);
            if (L0) { w.main_len[c] = n; w.all_gen[c] += 1; }       // paf_ctg_max_out.clear()
            wave_fence();
        } else if (cov == max_cov) {
            sel_all_append(k, w, c, cur, n, mark, ord);
        }
    }
    max_cov = -1;                                                   // alt path, :1613-1649
    if (found >= 2 && mind.anom != w.anom_dest[c]) {
        int64_t ans_up = 0, ans_down = 0;
        int32_t ans_idx = -1;
        for (int32_t i = 1; i < found; i++) {
            const Dist dd = kd[i];
            if (dd.anom >= mind.anom) continue;
            const int64_t up = (dd.qry + dd.ref) - (mind.qry + mind.ref);
            const int64_t down = (int64_t)mind.anom - dd.anom;
            if (ans_idx == -1 || up * ans_down < down * ans_up) {
                ans_up = up; ans_down = down; ans_idx = i;
                n = sel_convert(s, i, ++ord, cov);
                if (s.err) { if (L0) set_status(w, c, -6); return; }
                max_cov = cov;
                copy_resolved(ao, cur, n, mark, ord, k.lane);
                if (L0) w.alt_len[c] = n;
            } else {
                const Dist da = kd[ans_idx];
                if (dd.qry + dd.ref == da.qry + da.ref && dd.anom == da.anom) {
                    n = sel_convert(s, i, ++ord, cov);
                    if (s.err) { if (L0) set_status(w, c, -6); return; }
                    if (cov > max_cov) {
                        max_cov = cov;
                        copy_resolved(ao, cur, n, mark, ord, k.lane);
                        if (L0) w.alt_len[c] = n;
                    }
                }
            }
        }
    }
#if defined(AASM_KPROF) && !defined(AASM_HOST_EMUL)
    {   // slot 6/7: elapsed shader cycles and elapsed 100 MHz ticks of this wave -> effective clock
        __builtin_amdgcn_s_waitcnt(0);
        const int64_t mt1 = (int64_t)__builtin_amdgcn_s_memtime(), rt1 = (int64_t)__builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0);
        s.kp_acc[6] = mt1 - kp_mt0; s.kp_acc[7] = rt1 - kp_rt0;
    }
    if (k.lane == 0) for (int i_ = 0; i_ < 8; i_++) w.prof_sel[c * 8 + i_] = s.kp_acc[i_];
#endif
    sel_flush_counters(s, w);
}

// compaction of main/alt into ragged arrays: one wave per contig
AASM_DEV void kb_gather_out(const KCtx &k, const WS &w) {
    const int64_t c = k.bid;
    const int64_t b = w.rec_off[c] - w.R0;
    const int64_t nm = w.main_len[c], na = w.alt_len[c], om = w.main_off[c], oa = w.alt_off[c];
    for (int64_t t = k.lane; t < nm; t += AASM_WAVE) w.main_c[om + t] = w.main_out[b + t];
    for (int64_t t = k.lane; t < na; t += AASM_WAVE) w.alt_c[oa + t] = w.alt_out[b + t];
}

}  // namespace aasm
