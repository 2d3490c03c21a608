// aasm_pipeline.h -- launch sequence of the per-contig path-inference pipeline.
//
// Templated on a Backend that provides memory, launches, scans and scalar read-back:
//   * GpuBackend (aasm_gpu.hip): HIP stream, pooled device arena, HIP-event phase timers.
//   * tests/host_emul/emul.cpp: malloc + loops, for CPU-side logic tests only.
// Sizes are data dependent (slots S, vertices VT, edges ET, heap arena HT), so the
// sequence is count -> exclusive scan -> read total -> allocate -> fill, with five small
// device->host scalar reads per batch.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/alignasm_amd.h"
#include "aasm_kernels.h"
#include "aasm_enum.h"

namespace aasm {

enum Kern {
    KN_CS_RANGES, KN_SORT, KN_SORT_RANK, KN_SORT_FIX, KN_GATHER_PARTS, KN_OV_COUNT, KN_OV_MERGE, KN_VCOUNT, KN_VFILL_REC, KN_VFILL_SLOT,
    KN_NSL, KN_ROW_COUNT, KN_ROW_FILL, KN_GRAPH, KN_GRAPH_L, KN_REV_FILL, KN_REV_FILL_W, KN_REV_FILL_ORD, KN_REV_FILL_ORD_S, KN_SORT_ROWS_REV, KN_REV_HDR, KN_REV_SWEEP, KN_FWD_SWEEP, KN_REV_SWEEP_G, KN_FWD_SWEEP_G,
    KN_CHILDREN, KN_HEAP_CAP, KN_SIDETRACK, KN_SIDETRACK_W, KN_HEAP_HDR, KN_HEAP, KN_HEAP_MW, KN_HEAP_MW8, KN_HEAP_MW16, KN_MW_RANK, KN_ENUM, KN_ENUM_S, KN_ENUM_HEAP, KN_SELECT, KN_GATHER_OUT, KN_TOPO_COUNT, KN_TOPO_FILL,
    KN_SEL_PLAN, KN_SEL_PLANFILL, KN_SEL_RECOVER, KN_SEL_CLASSIFY, KN_SEL_CONVERT, KN_SEL_FINAL, KN_CHAIN, KN_CHAIN3, KN_K7_PREP, KN_TNX, KN_TNX16, KN_TNX16_WG
};

// dispatch a kernel body (used verbatim by both backends)
AASM_DEV void run_kernel_body(int kn, const KCtx &k, const WS &w) {
    switch (kn) {
        case KN_CS_RANGES: kb_cs_ranges(k, w); break;
        case KN_SORT: kb_sort(k, w); break;
        case KN_SORT_RANK: kb_sort_rank(k, w); break;
        case KN_SORT_FIX: kb_sort_fix(k, w); break;
        case KN_GATHER_PARTS: kb_gather_parts(k, w); break;
        case KN_OV_COUNT: kb_ov_count(k, w); break;
        case KN_OV_MERGE: kb_ov_merge(k, w); break;
        case KN_VCOUNT: kb_vcount(k, w); break;
        case KN_VFILL_REC: kb_vfill_rec(k, w); break;
        case KN_VFILL_SLOT: kb_vfill_slot(k, w); break;
        case KN_NSL: kb_nsl(k, w); break;
        case KN_ROW_COUNT: kb_row_count(k, w); break;
        case KN_ROW_FILL: kb_row_fill(k, w); break;
        case KN_GRAPH: kb_graph_build<GB_MAXV, GB_MAXE>(k, w); break;
        case KN_GRAPH_L: kb_graph_build<GB_MAXV_L, GB_MAXE_L>(k, w); break;
        case KN_REV_FILL: kb_rev_fill(k, w); break;
        case KN_REV_FILL_W: kb_rev_fill_w(k, w); break;
        case KN_REV_FILL_ORD: case KN_REV_FILL_ORD_S: kb_rev_fill_ord(k, w); break;
        case KN_SORT_ROWS_REV: kb_rev_place(k, w); break;
        case KN_REV_HDR: kb_rev_hdr(k, w); break;
        case KN_REV_SWEEP: kb_rev_sweep<AASM_WAVE>(k, w); break;
        case KN_FWD_SWEEP: kb_fwd_sweep<AASM_WAVE>(k, w); break;
        case KN_REV_SWEEP_G: kb_rev_sweep<AASM_SWEEP_G>(k, w); break;
        case KN_FWD_SWEEP_G: kb_fwd_sweep<AASM_SWEEP_G>(k, w); break;
        case KN_CHILDREN: kb_children(k, w); break;
        case KN_HEAP_CAP: kb_heap_cap(k, w); break;
        case KN_SIDETRACK: kb_sidetrack(k, w); break;
        case KN_SIDETRACK_W: kb_sidetrack_w(k, w); break;
        case KN_HEAP_HDR: kb_heap_hdr(k, w); break;
        case KN_HEAP: kb_heap<false, HEAP_RING_1W, HEAP_QN_1W>(k, w); break;
        case KN_HEAP_MW: case KN_HEAP_MW8: case KN_HEAP_MW16: kb_heap_mw(k, w); break;
        case KN_MW_RANK: kb_mw_rank(k, w); break;
#if defined(AASM_HOST_EMUL)
        case KN_ENUM: case KN_ENUM_S: case KN_ENUM_HEAP: kb_enum_heap(k, w); break;
#else
        case KN_ENUM: kb_enum_lsm<64>(k, w); break;
        case KN_ENUM_S: kb_enum_lsm<EQ_FSMALL>(k, w); break;
        case KN_ENUM_HEAP: kb_enum_heap(k, w); break;
#endif
        case KN_SELECT: kb_select(k, w); break;
        case KN_GATHER_OUT: kb_gather_out(k, w); break;
        case KN_TOPO_COUNT: kb_topo_count(k, w); break;
        case KN_TOPO_FILL: kb_topo_fill(k, w); break;
        case KN_SEL_PLAN: kb_sel_plan(k, w); break;
        case KN_SEL_PLANFILL: kb_sel_planfill(k, w); break;
        case KN_SEL_RECOVER: kb_sel_recover(k, w); break;
        case KN_SEL_CLASSIFY: kb_sel_classify(k, w); break;
        case KN_SEL_CONVERT: kb_sel_convert(k, w); break;
        case KN_SEL_FINAL: kb_sel_final(k, w); break;
        case KN_CHAIN: kb_chain<true>(k, w); break;
        case KN_CHAIN3: kb_chain<false>(k, w); break;
        case KN_K7_PREP: kb_k7_prep(k, w); break;
        case KN_TNX: kb_tnx(k, w); break;
        case KN_TNX16: kb_tnx16(k, w); break;
        case KN_TNX16_WG: kb_tnx16_wg(k, w); break;
        default: break;
    }
}

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
#ifndef AASM_CHAIN_ORD_MAX
#define AASM_CHAIN_ORD_MAX 1024         // contigs of the chain class up to which its workgroups run the order wave (four four-wave workgroups per CU; measured: 1 000 contigs 4.28 against 4.40 ms, 1 150 contigs 5.3 against 4.58)
#endif
#define AASM_CHAIN_SMALL_BATCH 1536      // contigs: up to here every contig of a sparse batch is in the class.  1 280 workgroups are resident at once (256 CUs x 5 three-wave workgroups; the kernels are held to 96 VGPRs - 5 waves / SIMD - so that the forward sweep beside them finds slots: at 4 waves / SIMD a 1 250-contig class took a second round, 5.55 against 4.64 ms); a partial second round still beats the three launches up to ~1 600 contigs (measured: 1 400 contigs 4.85 against 6.3 ms, 1 500: 5.9 / 6.5, 1 600: 6.55 / 6.64, 1 800: 6.7 / 6.8, 2 200: 7.4 / 7.2)

#ifndef AASM_GROUPED_MIN
#define AASM_GROUPED_MIN 2560
#endif
struct PipelineSizes { int64_t C = 0, R = 0, S = 0, VT = 0, ET = 0, HT = 0, bad_record = -1; };

// Runs the pipeline for contigs [0, C) described by `in` (device pointers; ctg_rec_off
// already offset to the chunk).  Leaves all intermediates in the backend's arena and
// returns the filled WS (so fetch / debug can read them).
template <class B>
int run_pipeline(B &be, const aasm_batch_in &in, const aasm_opts &opts, WS &w, PipelineSizes &sz) {
    std::memset(&w, 0, sizeof(w));
    const int64_t C = in.n_contigs;
    int64_t rr[2];
    be.read_i64s({in.ctg_rec_off, in.ctg_rec_off + C}, rr);
    const int64_t R0 = rr[0], R1 = rr[1];
    const int64_t R = R1 - R0;
    w.C = C; w.R = R; w.R0 = R0;
    w.K = opts.max_paths > 0 ? opts.max_paths : 10000;
    w.nsl = opts.non_skip_linkable ? 1 : 0;
    w.xcd_map = (opts.reserved[0] & 32) ? 0 : 1;                      // (bit 5, probes: blocks take work items in grid order)
    w.sort_depth_test = (opts.reserved[2] >> 8) & 0xff;               // test hook: depth limit of kb_sort_fix's introsort
    // the chain class (kb_chain: sweep, pre-pass and heaps of a contig beside each other).  reserved[0] bit 6: every sparse contig,
    // bit 7: none; else by shape: every contig of a batch too small to fill the chip (its step is its slowest contig's chain), and
    // the long tail of a big one (contigs of >= 4x the mean and >= 2048 records: each a chain many times the batch's own step)
    // (both bits, tests: the contigs of at least the batch's mean size - a class that is part of the batch whatever its shape, so that
    // the class's workgroups and the three launches of the others run beside each other)
    w.chain_mode = (opts.reserved[0] & 192) == 192 ? 0 : (opts.reserved[0] & 64) ? 1 : (opts.reserved[0] & 128) ? 2 : 0;
    w.chain_test = (opts.reserved[2] & 16) ? 2 : (opts.reserved[2] & 8) ? 1 : 0;
    w.chain_rn = (opts.reserved[2] & 64) ? 4 : 0;                     // (bit 6, tests: a 4-entry ring of roots in the chain class's heap wave)
    w.chain_ord = (opts.reserved[2] & 32) ? 0 : 1;                    // (bit 5: the heap wave of the chain class keeps its own BFS queue - probes, tests; decided below by the size of the class)
    w.chain_all = ((opts.reserved[0] & 192) != 192 && C <= AASM_CHAIN_SMALL_BATCH) ? 1 : 0;
    w.chain_minN = (int32_t)std::min<int64_t>(std::max<int64_t>(2048, 4 * (R / std::max<int64_t>(C, 1))), INT32_MAX);
    if ((opts.reserved[0] & 192) == 192) w.chain_minN = (int32_t)std::min<int64_t>(std::max<int64_t>(1, cdiv(R, std::max<int64_t>(C, 1))), INT32_MAX);
    w.rec_off = in.ctg_rec_off; w.in_qs = in.qry_str; w.in_qe = in.qry_end; w.in_rs = in.ref_str; w.in_re = in.ref_end;
    w.in_qt = in.qry_total; w.in_chr = in.ref_chr; w.in_fwd = in.aln_fwd; w.in_mq = in.map_qul;
    w.in_rng_off = in.rec_rng_off; w.rql = in.rng_qry_l; w.rqr = in.rng_qry_r; w.rrl = in.rng_ref_l; w.rng_stride = 1;
    sz.C = C; sz.R = R;
    if (C <= 0 || R <= 0) return AASM_E_INVAL;

#define A(field, type, n, name) w.field = (type *)be.alloc(name, sizeof(type) * (size_t)((n) > 0 ? (n) : 1))
#define AZ(field, type, n, name) do { A(field, type, n, name); be.zero_alloc(w.field, sizeof(type) * (size_t)((n) > 0 ? (n) : 1)); } while (0)
// out of device memory -> AASM_E_NOMEM (the caller may split the contig range); any other HIP failure (launch,
// memset, scan, read-back) -> AASM_E_HIP with the original error text, never retried
#define CHECK_ALLOC() do { if (be.oom()) return AASM_E_NOMEM; if (be.failed()) return AASM_E_HIP; } while (0)

    AZ(status, int32_t, C, "status");
    AZ(prof_heap, int64_t, C * 8, "prof_heap"); AZ(prof_sel, int64_t, C * 8, "prof_sel"); AZ(prof_gb, int64_t, C * 8, "prof_gb");
    AZ(counters, int64_t, CNT_N, "counters");
    // every zero-initialised per-contig array of the pipeline, one behind the other: one fill for all of them (they used to be zeroed where
    // they were first needed: five more fill dispatches per step)
    AZ(dupflag, int32_t, C, "dupflag");
    AZ(main_len, int32_t, C, "main_len"); AZ(alt_len, int32_t, C, "alt_len"); AZ(all_gen, int32_t, C, "all_gen"); AZ(all_seq, int32_t, C, "all_seq");
    AZ(kfound, int32_t, C, "kfound"); AZ(anom_dest, int32_t, C, "anom_dest"); AZ(h_cnt, int32_t, C, "h_cnt");
    AZ(nconv, int32_t, C, "nconv");

    // ---- K0 (optional): match ranges from the cs tags, on the device
    if (!in.rng_qry_l && in.cs_text && in.rec_cs_off) {
        if (R > INT32_MAX) return AASM_E_INVAL;
        int64_t gg[2];
        be.read_i64s({in.rec_rng_off + R0, in.rec_rng_off + R1}, gg);
        const int64_t G0 = gg[0], G1 = gg[1];
        be.phase_begin(AASM_PH_CS);
        w.cs_text = in.cs_text; w.cs_off = in.rec_cs_off;
        A(rng_rec, int64_t, 4 * (G1 - G0), "rng_rec"); A(cs_bad, int32_t, 2, "cs_bad");
        CHECK_ALLOC();
        w.rng_rec -= 4 * G0;                                         // indexed with the batch's own range offsets
        be.zero(w.cs_bad, 8);
        be.launch(KN_CS_RANGES, cdiv(R, 256), 256, w);
        be.phase_end(AASM_PH_CS);
        const int32_t badv = (int32_t)(uint32_t)(uint64_t)be.read_i64((const int64_t *)w.cs_bad);   // 0 = none, else record - INT32_MAX
        if (badv != 0) { sz.bad_record = R0 + ((int64_t)badv + INT32_MAX); return AASM_E_PARSE; }
        w.rql = w.rng_rec; w.rqr = w.rng_rec + 1; w.rrl = w.rng_rec + 2; w.rng_stride = 4;
    } else if (!in.rng_qry_l) return AASM_E_INVAL;

    // ---- K1 sort + parts
    be.phase_begin(AASM_PH_SORT);
    A(perm, int32_t, R, "perm"); A(np, int32_t, C, "np"); A(pstart, int32_t, R + C, "pstart");
    A(s_qs, int64_t, R, "s_qs"); A(s_qe, int64_t, R, "s_qe"); A(s_rs, int64_t, R, "s_rs"); A(s_re, int64_t, R, "s_re");
    A(s_qt, int64_t, R, "s_qt"); A(s_rb, int64_t, R, "s_rb"); A(s_rn, int32_t, R, "s_rn"); A(s_chr, int32_t, R, "s_chr");
    A(s_orig, int32_t, R, "s_orig"); A(s_ctg, int32_t, R, "s_ctg"); A(s_pid, int32_t, R, "s_pid"); A(s_fl, uint8_t, R, "s_fl");
    CHECK_ALLOC();
    if (opts.reserved[2] & 1) be.launch(KN_SORT, C, 4096, w);       // test hook: an invalid launch configuration (block size > 1024)
    be.launch(KN_SORT, C, 256, w);
    be.launch(KN_SORT_RANK, cdiv(R, 256), 256, w);
    be.launch(KN_SORT_FIX, C, AASM_WAVE, w);
    be.launch(KN_GATHER_PARTS, C, AASM_WAVE, w);
    be.phase_end(AASM_PH_SORT);

    // ---- K2 overlap slots
    be.phase_begin(AASM_PH_PAIRS);
    A(ov_cnt, int32_t, R, "ov_cnt"); A(ov_off, int64_t, R + 1, "ov_off");
    CHECK_ALLOC();
    be.launch(KN_OV_COUNT, cdiv(R, 256), 256, w);
    be.scan_i32(w.ov_cnt, R, w.ov_off);
    const int64_t S = be.read_i64(w.ov_off + R);
    w.S = S; sz.S = S;
    A(ov_rec, int32_t, S, "ov_rec"); A(ov_vid, int32_t, S, "ov_vid"); A(ov_rank, int64_t, S + 1, "ov_rank");
    A(ov_peq, int64_t, S, "ov_peq"); A(ov_per, int64_t, S, "ov_per"); A(ov_stq, int64_t, S, "ov_stq"); A(ov_str, int64_t, S, "ov_str");
    A(ov_ok, uint8_t, S, "ov_ok");
    CHECK_ALLOC();
    if (S > 0) be.launch(KN_OV_MERGE, cdiv(S, 256), 256, w);
    be.scan_u8(w.ov_ok, S, w.ov_rank);
    A(ctgV, int32_t, C, "ctgV"); A(voff, int64_t, C + 1, "voff");
    CHECK_ALLOC();
    be.launch(KN_VCOUNT, cdiv(C, 256), 256, w);
    be.scan_i32(w.ctgV, C, w.voff);
    const int64_t VT = be.read_i64(w.voff + C);
    w.VT = VT; sz.VT = VT;
    A(v_i, int32_t, VT, "v_i"); A(v_j, int32_t, VT, "v_j"); A(v_ctg, int32_t, VT, "v_ctg"); A(v_slot, int64_t, VT, "v_slot");
    CHECK_ALLOC();
    be.launch(KN_VFILL_REC, cdiv(R, 256), 256, w);
    if (S > 0) be.launch(KN_VFILL_SLOT, cdiv(S, 256), 256, w);
    be.phase_end(AASM_PH_PAIRS);

    // main/alt outputs exist even when no contig has a graph (all single-record contigs)
    A(cur_out, OutElem, R, "cur_out"); A(main_out, OutElem, R, "main_out"); A(alt_out, OutElem, R, "alt_out");
    A(main_off, int64_t, C + 1, "main_off"); A(alt_off, int64_t, C + 1, "alt_off");
    w.pool_cap = R + 1024; w.ar_cap = C + R / 4 + 1024;
    A(pool, OutElem, w.pool_cap, "pool");
    A(ar_ctg, int32_t, w.ar_cap, "ar_ctg"); A(ar_gen, int32_t, w.ar_cap, "ar_gen"); A(ar_seq, int32_t, w.ar_cap, "ar_seq");
    A(ar_len, int32_t, w.ar_cap, "ar_len"); A(ar_off, int64_t, w.ar_cap, "ar_off");
    CHECK_ALLOC();

    if (VT > 0) {
        // ---- K3/K4 CSR
        be.phase_begin(AASM_PH_EDGES);
        if (w.nsl) {
            A(dis_end, int32_t, R, "dis_end"); AZ(next_cnt, int32_t, R, "next_cnt");
            CHECK_ALLOC();
            be.launch(KN_NSL, cdiv(R, 256), 256, w);
        }
        A(deg, int32_t, VT, "deg"); A(rowptr, int64_t, VT + 1, "csr_rowptr");
        // (heap arena sizing + classes need only V and E per contig: sized here, so that their read-back shares the edges' one)
        A(hcap_cnt, int32_t, C, "hcap_cnt"); A(hoff, int64_t, C + 1, "hoff");
        A(mw_flag, int32_t, C, "mw_flag"); A(mw_lg, int32_t, C, "mw_lg"); A(mw_cap, int32_t, C, "mw_cap"); A(mw_off, int64_t, C + 1, "mw_off"); A(mw_list, int32_t, C, "mw_list"); A(mw_sorted, int32_t, C, "mw_sorted"); A(mw_key, int32_t, C, "mw_key");
        A(chain_flag, int32_t, C, "chain_flag"); A(chain_list, int32_t, C, "chain_list"); A(gb_flag, int32_t, C, "gb_flag");
        w.gb_off = (opts.reserved[0] & 0x10000) ? 1 : 0;                 // (bit 16: every contig by the separate launches, for tests and probes)
        w.mw_mode = (opts.reserved[0] & 2) ? 1 : (opts.reserved[0] & 4) ? 2 : 0;
        w.mw_compact = opts.keep_debug ? 1 : 0;   // debug runs compare arena indices with the reference's allocation order
        CHECK_ALLOC();
        be.launch(KN_ROW_COUNT, cdiv(VT, 256), 256, w);
        be.scan_i32(w.deg, VT, w.rowptr);
        be.launch(KN_HEAP_CAP, cdiv(C, 256), 256, w);
        be.scan_i32_pair(w.hcap_cnt, w.hoff, w.mw_cap, w.mw_off, C);
        int64_t et_mv[12];
        be.read_i64s({w.rowptr + VT, w.counters + CNT_MAXV, w.hoff + C, w.mw_off + C, w.counters + CNT_MW, w.counters + CNT_MAXN, w.counters + CNT_CHAIN, w.counters + CNT_GB_S, w.counters + CNT_GB_L, w.counters + CNT_GB_REST}, et_mv);
        const int64_t ET = et_mv[0], MAXV = et_mv[1];                 // (MAXV: most vertices of one contig)
        const int64_t GB_S = et_mv[7], GB_L = et_mv[8], GB_REST = et_mv[9];   // contigs whose graph one workgroup builds (kb_graph_build, two forms) / the others
        const int64_t hh[4] = {et_mv[2], et_mv[3], et_mv[4], et_mv[5]};
        const int64_t NCHAIN = et_mv[6];
        // the class's BFS order from a wave of its own while four waves a contig fit the chip beside the forward sweep (four workgroups per CU);
        // beyond that the three-role kernel, where the heap wave keeps its own queue
        if (NCHAIN > AASM_CHAIN_ORD_MAX) w.chain_ord = 0;
        w.ET = ET; sz.ET = ET;
        A(e_col, int32_t, ET, "csr_col"); A(e_wq, int64_t, ET, "csr_w_qry"); A(e_wr, int32_t, ET, "csr_w_ref"); A(e_fl, uint8_t, ET, "csr_w_flags");
        A(rptr, int64_t, VT + 1, "rptr"); A(r_pk, I4, ET, "r_pk");
        A(rvh, I4, 3 * VT, "rvh"); A(fvh, I4, 2 * VT, "fvh");
        A(sp_d, Dist, VT, "sp_d"); A(sp_best, int32_t, VT, "sp_best"); A(cnt_tmp, int32_t, VT, "cnt_tmp"); A(cnt_tmp2, int32_t, VT, "cnt_tmp2"); A(an, int32_t, VT, "an");
        if (NCHAIN > 0) { A(pend, int32_t, VT, "pend"); A(cq, int32_t, VT, "cq"); A(bfsq, I4, VT, "bfsq"); }
        // sparse, every contig small: rows, reversed CSR and the sweeps' headers of a contig by ONE workgroup (kb_graph_build)
        // the small contigs of a sparse batch: rows, reversed CSR and the sweeps' headers of a contig by ONE workgroup (kb_graph_build: kb_heap_cap
        // picked them); the others - dense batches, contigs of more than GB_MAXV_L vertices or GB_MAXE_L edges - by the separate launches, which
        // leave the contigs of the class alone
        w.indeg = nullptr;
        if (GB_REST > 0) { AZ(indeg, int32_t, VT, "indeg"); AZ(rcur, int32_t, VT, "rcur"); }   // (two fills in one)
        CHECK_ALLOC();
        if (GB_S > 0) be.launch(KN_GRAPH, C, GB_TPB, w);
        if (GB_L > 0) be.launch(KN_GRAPH_L, C, GB_TPB, w);
        if (GB_REST == 0) {
            be.phase_end(AASM_PH_EDGES);
            be.phase_begin(AASM_PH_REVCSR);
            be.phase_end(AASM_PH_REVCSR);
        } else {
        be.launch(KN_ROW_FILL, cdiv(VT, AASM_WAVE), AASM_WAVE, w);
        be.phase_end(AASM_PH_EDGES);

        // ---- reversed CSR
        be.phase_begin(AASM_PH_REVCSR);
        A(r_e, int32_t, ET, "r_e"); A(tmp_pk, I4, ET, "tmp_pk");
        CHECK_ALLOC();
        be.scan_i32(w.indeg, VT, w.rptr);
        // dense, no giant contig: the in-lists in order from one pass per contig (every contig of <= REV_ORD_MIDV vertices - C5's have 2 500 -: the
        // launch with 6.7 KB of LDS counters instead of 25 KB, i.e. 6 waves per SIMD instead of 2)
        if (ET > 6 * VT && MAXV <= REV_ORD_MAXV) be.launch(MAXV <= REV_ORD_MIDV ? KN_REV_FILL_ORD_S : KN_REV_FILL_ORD, C, AASM_WAVE, w);
        else {
            if (ET > 6 * VT) be.launch(KN_REV_FILL_W, cdiv(VT, AASM_WAVE), AASM_WAVE, w);   // dense: lanes over the edges of 64 rows
            else be.launch(KN_REV_FILL, cdiv(VT, 256), 256, w);
            be.launch(KN_SORT_ROWS_REV, cdiv(VT, AASM_WAVE), AASM_WAVE, w);
        }
        be.launch(KN_REV_HDR, cdiv(VT, 256), 256, w);
        be.phase_end(AASM_PH_REVCSR);
        }

        // ---- K6 / K5 sweeps.  The forward sweep + the topologically ordered CSR copy only feed
        // K9, so they run on a second stream beside rev_sweep -> heaps -> enumeration.
        A(rev_order, int32_t, VT, "rev_order"); A(fwd_order, int32_t, VT, "fwd_order"); A(fwd_pos, int32_t, VT, "fwd_pos");
        A(tp_deg, int32_t, VT, "tp_deg"); A(tp_vj, int32_t, VT, "tp_vj"); A(tp_ptr, int64_t, VT + 1, "tp_ptr");
        A(te_pk, I4, ET, "te_pk");
        CHECK_ALLOC();
        // big sparse batches (mean degree <= 6, thousands of contigs: bound by instruction issue): two contigs per wave,
        // AASM_SWEEP_G lanes each; dense ones and small batches (bound by the chain per contig): a wave per contig
        const bool grouped = ET <= 6 * VT && C >= AASM_GROUPED_MIN;            // (few contigs: a wave each - the scalar-uniform variant has the shorter chain per pop)
        const int64_t sweep_n = AASM_WAVE / AASM_SWEEP_G;
        auto side_work = [&]() {
            be.fork();                                               // side stream waits for everything enqueued so far
            be.use_side(true);
            be.phase_begin(AASM_PH_FWD);
            if (grouped) be.launch(KN_FWD_SWEEP_G, cdiv(C, sweep_n), AASM_WAVE, w);
            else be.launch(KN_FWD_SWEEP, C, AASM_WAVE, w);
            be.phase_end(AASM_PH_FWD);
            be.phase_begin(AASM_PH_TOPO);
            be.launch(KN_TOPO_COUNT, cdiv(VT, 256), 256, w);
            be.scan_i32(w.tp_deg, VT, w.tp_ptr);
            be.launch(KN_TOPO_FILL, cdiv(VT, AASM_WAVE), AASM_WAVE, w);
            be.phase_end(AASM_PH_TOPO);
            be.use_side(false);
        };
        // (Measured in round 4, same box: started after the reverse sweep instead, or beside K7 only, the reverse sweep drops to 1.98 ms
        // but K7 beside it rises 4.10 -> 4.6 / 5.0 ms and the step 12.60 -> 12.99 / 12.91: every chain kernel is short of issue slots.)
        side_work();
        // ---- K7's arrays (the chain class fills them while its sweep still runs, so they exist before any sweep starts)
        A(ccnt, int32_t, VT, "ccnt"); A(cval, int32_t, ET, "cval");
        A(st_cost, Dist, ET, "st_cost"); A(st_n, int32_t, VT, "st_n"); A(vhdr, I4, VT, "vhdr"); A(vhdr2, I4, VT, "vhdr2"); A(cinfo, I4, ET, "cinfo"); A(tnx, I4, VT, "tnx"); A(tnx16, int32_t, 16 * VT, "tnx16");
        const int64_t HT = hh[0], HTM = hh[1], NMW = HTM > 0 ? hh[2] : 0;
        sz.HT = HT;
        w.avg_sidetracks = (int32_t)std::min<int64_t>((ET - VT + C) / (C > 0 ? C : 1), INT32_MAX);
        A(hnodes, HNode, HT, "hnodes"); A(h_root, int32_t, VT, "h_root"); A(bq, int32_t, VT, "bq");
        A(hprov, HNode, HTM, "hprov");
        A(mw_order, int32_t, VT, "mw_order"); A(mw_rs, int32_t, VT, "mw_rs"); A(mw_fb, int32_t, VT, "mw_fb"); A(mw_rsv, int32_t, VT, "mw_rsv"); A(mw_used, int32_t, VT, "mw_used");
        CHECK_ALLOC();
        be.fill_ff(w.h_root, sizeof(int32_t) * (size_t)VT);
        be.fill_ff(w.bq, sizeof(int32_t) * (size_t)VT);
        if (NCHAIN > 0) {
            // the class's workgroups (three waves a contig) on a stream of their own, beside the forward sweep and - when the class is
            // only the batch's long tail - beside the three launches of everybody else
            be.fill_ff(w.vhdr, sizeof(I4) * (size_t)VT);             // the marker words the heap wave waits on
            be.fill_ff(w.vhdr2, sizeof(I4) * (size_t)VT);
            be.fork2();
            be.use_side2(true);
            be.phase_begin(AASM_PH_CHAIN);
            if (w.chain_ord) be.launch(KN_CHAIN, NCHAIN, AASM_WAVE * CHAIN_WAVES, w);
            else be.launch(KN_CHAIN3, NCHAIN, AASM_WAVE * (CHAIN_WAVES - 1), w);   // (no order wave: three waves a contig, five workgroups a CU; 1 250 contigs 4.6 ms where four-wave workgroups took 6.0)
            be.phase_end(AASM_PH_CHAIN);
            be.use_side2(false);
        }
        if (NCHAIN < C) {
        be.phase_begin(AASM_PH_SPTREE);
        if (grouped) be.launch(KN_REV_SWEEP_G, cdiv(C, sweep_n), AASM_WAVE, w);
        else be.launch(KN_REV_SWEEP, C, AASM_WAVE, w);
        be.phase_end(AASM_PH_SPTREE);

        // ---- K7 heaps
        // the jump records of K9's recovery (the next 4 and the next 16 vertices along best[]) need nothing but the tree: they go to the
        // side stream, which finishes its topological copy about when the reverse sweep ends, and run beside the heap pre-pass
        be.fork_again();
        be.use_side(true);
        if (GB_S + GB_L > 0) be.launch(KN_TNX16_WG, C, TNX_TPB, w);  // (the small contigs of a sparse batch - kb_graph_build's class: sixteen hops through the tree in LDS)
        if (GB_REST > 0) {
            be.launch(KN_TNX, cdiv(VT, 256), 256, w);
            be.launch(KN_TNX16, cdiv(VT, 256), 256, w);
        }
        be.use_side(false);
        be.phase_begin(AASM_PH_HEAP_PREP);
        if (ET > 6 * VT) {
            be.launch(KN_CHILDREN, cdiv(VT, 256), 256, w);
            be.launch(KN_SIDETRACK_W, cdiv(VT, AASM_WAVE), AASM_WAVE, w);   // dense: lanes over the edges of 64 rows
            be.launch(KN_HEAP_HDR, cdiv(VT, 256), 256, w);
        } else be.launch(KN_K7_PREP, cdiv(VT, 256), 256, w);         // child list + keys + header of a vertex: one thread, one launch
        be.phase_end(AASM_PH_HEAP_PREP);
        // Nothing may START beside the heap kernel: all its workgroups are resident for the whole launch, so whatever share of the CUs a
        // second queue holds while they are dealt out skews their placement for good (measured: the 0.2 ms jump-record kernel started
        // beside it cost it 1.8 ms, 4.05 -> 5.8; the forward sweep beside it 0.5-0.9 ms in round 4).  The side stream is done by now.
        be.join();
        be.phase_begin(AASM_PH_HEAP);
        be.launch(KN_HEAP, C, AASM_WAVE, w);
        // contigs of the wide-tree class (kb_heap skips them): 16, 8 or 4 waves each, by how many of them share the chip's ~8 k wave slots
        w.mw_n = (int32_t)NMW; w.mw_base = -1;
        if (HTM > 0) {
            const int hook = (opts.reserved[0] >> 8) & 0xff;         // probes: 1 = input order; 4 / 8 / 16 = that many waves per contig
            const int mw = (hook == 4 || hook == 8 || hook == 16) ? hook : NMW * 16 <= 6144 ? 16 : NMW * 8 <= 6144 ? 8 : 4;
            const int kn = mw == 16 ? KN_HEAP_MW16 : mw == 8 ? KN_HEAP_MW8 : KN_HEAP_MW;
            // One block per contig of the class, the contig with the largest node bound first: block times of a dense batch go with the
            // node count (C5 share: mean 25 ms, longest 42), and in input order the heavy ones land on the CUs as they come - clumps of them
            // share a CU's issue slots and the launch ends with such a clump.  Largest first deals every CU a spread of weights and
            // starts the longest chains first: C5 share 35.5 -> 30.0 ms, 700 contigs 27.3 -> 25.5, 400 x 1 500 records 25.1 -> 22.8.
            // opts.reserved[0] bits 8-15 == 1 (probes): input order, a block per contig of the batch.
            const bool by_list = hook != 1 && NMW >= 2 && NMW <= 32768;  // (ranked by counting: NMW^2 compares)
            if (by_list) {
                be.launch(KN_MW_RANK, cdiv(NMW, 256), 256, w);
                w.mw_base = 0;
                { const int32_t xm = w.xcd_map; w.xcd_map = 0; be.launch(kn, NMW, AASM_WAVE * mw, w); w.xcd_map = xm; }   // (its order is its own)
                w.mw_base = -1;
            } else be.launch(kn, C, AASM_WAVE * mw, w);
        }
        be.phase_end(AASM_PH_HEAP);
        }                                                            // (NCHAIN < C)
        be.join2();                                                  // the chain class's heaps, before anybody enumerates

        // ---- K8 enumeration
        const int64_t K = w.K;
        A(kd, Dist, C * K, "kd"); A(klast, int32_t, C * K, "klast");
        // queue form: sorted front + sorted runs (aasm_enum.h) unless a contig is too long for its packed ratio key; reserved[0] bit 3
        // (tests) forces the d-ary heap form, which is also what the 1-lane host emulation runs
        const bool enum_heap = B::host_emulation || (opts.reserved[0] & 8) != 0 || hh[3] > AASM_ENUM_MAX_N;
        w.pq_stride = enum_heap ? 3 * K + 1 : enum_stride(K);
        A(kcand, I4, 2 * C * (3 * K + 1), "kcand"); A(pq, PqK, C * w.pq_stride, "pq");
        CHECK_ALLOC();
        be.phase_begin(AASM_PH_ENUM);
        // more contigs than the 64-entry front keeps resident (14 waves per CU = 3 584): the 40-entry front (20 per CU = 5 120) runs them
        // in one residency round (round 4, with the far tier: 5 000 contigs 23.85 -> 23.1 ms; round 3, without it, the extra refills and
        // flushes of the small front cost more than the second round: 30.2 vs 31.5 ms); beyond 5 120 both take a second round
        be.launch(enum_heap ? KN_ENUM_HEAP : ((C > 14 * 256 && C <= 20 * 256 && K > 21) || (opts.reserved[0] & 16)) ? KN_ENUM_S : KN_ENUM, C, AASM_WAVE, w);
        be.phase_end(AASM_PH_ENUM);
    }

    be.join();                                                       // forward order / topo copy ready
    // ---- K9 selection (also emits the N == 1 contigs)
    // Default: plan -> one wave per converted path -> per-contig final pick.  Falls back to the
    // sequential one-wave-per-contig kernel when the per-conversion scratch would not fit
    // (tie-heavy inputs at large K) or when opts.reserved[0] bit 0 asks for it (tests).
    A(mark_time, int32_t, R, "mark_time");
    A(conv_off, int64_t, C + 1, "conv_off"); A(plan_kk, int32_t, C * 2 * SEL_PLAN_KEEP, "plan_kk");
    CHECK_ALLOC();
    be.fill_byte(w.mark_time, 0x7F, sizeof(int32_t) * (size_t)R);
    bool sequential = (opts.reserved[0] & 1) != 0;
    int64_t NCONV = 0, SR = 0, SV = 0;
    bool nm_ready = false;                                           // the output totals have been read with the pick's pool demand
    int64_t nm[2] = {0, 0};
    if (!sequential) {
        be.phase_begin(AASM_PH_MISC);
        if (VT > 0) be.launch(KN_SEL_PLAN, C, AASM_WAVE, w);
        be.scan_i32(w.nconv, C, w.conv_off);
        NCONV = be.read_i64(w.conv_off + C);
        w.NCONV = NCONV;
        if (NCONV > 0) {
            A(cv_ctg, int32_t, NCONV, "cv_ctg"); A(cv_k, int32_t, NCONV, "cv_k"); A(cv_ord, int32_t, NCONV, "cv_ord"); A(cv_kind, int32_t, NCONV, "cv_kind");
            A(cv_szr, int32_t, NCONV, "cv_szr"); A(cv_szv, int32_t, NCONV, "cv_szv"); A(cv_roff, int64_t, NCONV + 1, "cv_roff"); A(cv_voff, int64_t, NCONV + 1, "cv_voff");
            AZ(cv_n, int32_t, NCONV, "cv_n"); AZ(cv_err, int32_t, NCONV, "cv_err"); AZ(cv_cov, int64_t, NCONV, "cv_cov"); A(cv_la, int32_t, NCONV, "cv_la");
            CHECK_ALLOC();
            be.launch(KN_SEL_PLANFILL, C, AASM_WAVE, w);
            be.scan_i32_pair(w.cv_szr, w.cv_roff, w.cv_szv, w.cv_voff, NCONV);
            int64_t sv[2];
            be.read_i64s({w.cv_roff + NCONV, w.cv_voff + NCONV}, sv);
            SR = sv[0]; SV = sv[1];
            const int64_t bytes = SR * (24 + (int64_t)sizeof(OutElem)) + SV * ((int64_t)sizeof(Dist) + 8);
            if (bytes > ((int64_t)24 << 30)) sequential = true;
        }
        be.phase_end(AASM_PH_MISC);
    }
    if (!sequential) {
        if (NCONV > 0) {
            A(cv_path, int32_t, 6 * SR, "cv_path"); A(cv_out, OutElem, SR, "cv_out");
            A(cv_dist2, Dist, SV, "cv_dist2"); A(cv_pre2, int32_t, SV, "cv_pre2"); AZ(cv_stamp, int32_t, SV, "cv_stamp");
            CHECK_ALLOC();
            be.phase_begin(AASM_PH_SELECT);
            be.launch(KN_SEL_RECOVER, NCONV, AASM_WAVE, w);
            be.launch(KN_SEL_CLASSIFY, NCONV, 256, w);
            be.launch(KN_SEL_CONVERT, NCONV, AASM_WAVE, w);
            be.phase_end(AASM_PH_SELECT);
        }
        be.phase_begin(AASM_PH_FINAL);
        be.launch(KN_SEL_FINAL, C, AASM_WAVE, w);
        // the output lengths are final with the pick: their offsets are scanned at once, and ONE read-back brings the pool demand and the
        // two totals (it was a wait for the demand, then the scans, then a wait for the totals)
        be.scan_i32_pair(w.main_len, w.main_off, w.alt_len, w.alt_off, C);
        int64_t np2[4];
        be.read_i64s({w.counters + CNT_POOL, w.counters + CNT_AR, w.main_off + C, w.alt_off + C}, np2);
        const int64_t need_pool = np2[0], need_ar = np2[1];
        nm_ready = true; nm[0] = np2[2]; nm[1] = np2[3];
        if (need_pool > w.pool_cap || need_ar > w.ar_cap) {         // .all pool overflow: exact-size re-run of the pick only
            w.pool_cap = need_pool + 16; w.ar_cap = need_ar + 16;
            A(pool, OutElem, w.pool_cap, "pool");
            A(ar_ctg, int32_t, w.ar_cap, "ar_ctg"); A(ar_gen, int32_t, w.ar_cap, "ar_gen"); A(ar_seq, int32_t, w.ar_cap, "ar_seq");
            A(ar_len, int32_t, w.ar_cap, "ar_len"); A(ar_off, int64_t, w.ar_cap, "ar_off");
            CHECK_ALLOC();
            be.zero(w.all_seq, sizeof(int32_t) * (size_t)C);
            be.zero(w.counters + CNT_POOL, sizeof(int64_t)); be.zero(w.counters + CNT_AR, sizeof(int64_t)); be.zero(w.counters + CNT_OVF, sizeof(int64_t));
            be.launch(KN_SEL_FINAL, C, AASM_WAVE, w);
            nm_ready = false;                                        // (the same lengths again, but keep the one code path: scanned and read below)
        }
        be.phase_end(AASM_PH_FINAL);
    } else {
        A(pathA, int32_t, 2 * (R + 2 * C), "pathA"); A(pathB, int32_t, 2 * (R + 2 * C), "pathB"); A(pathT, int32_t, 2 * (R + 2 * C), "pathT");
        A(pre2, int32_t, VT, "pre2"); AZ(stamp, int32_t, VT, "stamp"); A(dist2, Dist, VT, "dist2");
        CHECK_ALLOC();
        be.phase_begin(AASM_PH_SELECT);
        be.launch(KN_SELECT, C, AASM_WAVE, w);
        be.phase_end(AASM_PH_SELECT);
        int64_t np2[2];
        be.read_i64s({w.counters + CNT_POOL, w.counters + CNT_AR}, np2);
        const int64_t need_pool = np2[0], need_ar = np2[1];
        if (need_pool > w.pool_cap || need_ar > w.ar_cap) {         // .all pool overflow (tie-heavy inputs): one exact-size re-run
            w.pool_cap = need_pool + 16; w.ar_cap = need_ar + 16;
            A(pool, OutElem, w.pool_cap, "pool");
            A(ar_ctg, int32_t, w.ar_cap, "ar_ctg"); A(ar_gen, int32_t, w.ar_cap, "ar_gen"); A(ar_seq, int32_t, w.ar_cap, "ar_seq");
            A(ar_len, int32_t, w.ar_cap, "ar_len"); A(ar_off, int64_t, w.ar_cap, "ar_off");
            CHECK_ALLOC();
            be.zero(w.stamp, sizeof(int32_t) * (size_t)(VT > 0 ? VT : 1));
            be.fill_byte(w.mark_time, 0x7F, sizeof(int32_t) * (size_t)R);
            be.zero(w.all_gen, sizeof(int32_t) * (size_t)C); be.zero(w.all_seq, sizeof(int32_t) * (size_t)C);
            be.zero(w.counters + CNT_POOL, sizeof(int64_t)); be.zero(w.counters + CNT_AR, sizeof(int64_t));
            be.zero(w.counters + CNT_CONVERTED, sizeof(int64_t)); be.zero(w.counters + CNT_OVF, sizeof(int64_t));
            be.zero(w.counters + CNT_ISPR_E, sizeof(int64_t)); be.zero(w.counters + CNT_ISPR_V, sizeof(int64_t));
            be.zero(w.counters + CNT_PATH_E, sizeof(int64_t)); be.zero(w.counters + CNT_OUT_E, sizeof(int64_t));
            be.phase_begin(AASM_PH_MISC);
            be.launch(KN_SELECT, C, AASM_WAVE, w);
            be.phase_end(AASM_PH_MISC);
        }
    }

    // ---- output compaction
    be.phase_begin(AASM_PH_GATHER);
    if (!nm_ready) {
        be.scan_i32_pair(w.main_len, w.main_off, w.alt_len, w.alt_off, C);
        be.read_i64s({w.main_off + C, w.alt_off + C}, nm);
    }
    const int64_t NM = nm[0], NA = nm[1];
    A(main_c, OutElem, NM, "main_c"); A(alt_c, OutElem, NA, "alt_c");
    CHECK_ALLOC();
    be.launch(KN_GATHER_OUT, C, AASM_WAVE, w);
    be.phase_end(AASM_PH_GATHER);
#undef A
#undef AZ
#undef CHECK_ALLOC
    return AASM_OK;
}

// Pack device results into the ragged host structure (shared by both backends).
template <class B>
int fetch_results(B &be, const WS &w, const PipelineSizes &sz, aasm_batch_out *out) {
    std::memset(out, 0, sizeof(*out));
    const int64_t C = w.C;
    out->n_contigs = C;
    out->main_off = (int64_t *)calloc(C + 1, 8);
    out->alt_off = (int64_t *)calloc(C + 1, 8);
    out->all_path_off = (int64_t *)calloc(C + 1, 8);
    out->ctg_status = (int32_t *)calloc(C + 1, 4);
    be.d2h(out->main_off, w.main_off, (C + 1) * 8);
    be.d2h(out->alt_off, w.alt_off, (C + 1) * 8);
    be.d2h(out->ctg_status, w.status, C * 4);
    const int64_t NM = out->main_off[C], NA = out->alt_off[C];
    out->main_elems = (aasm_out_elem *)malloc((NM + 1) * sizeof(aasm_out_elem));      // (every element is overwritten: no zero fill of 100s of MB)
    out->alt_elems = (aasm_out_elem *)malloc((NA + 1) * sizeof(aasm_out_elem));
    static_assert(sizeof(aasm_out_elem) == sizeof(OutElem), "layout");
    if (NM) be.d2h_big(out->main_elems, w.main_c, NM * sizeof(OutElem));
    if (NA) be.d2h_big(out->alt_elems, w.alt_c, NA * sizeof(OutElem));
    int64_t cnt[CNT_N];
    be.d2h(cnt, w.counters, sizeof(cnt));
    // .all paths: keep records of the final generation, ordered by (contig, seq)
    int64_t nar = cnt[CNT_AR] < w.ar_cap ? cnt[CNT_AR] : w.ar_cap;
    int64_t npool = cnt[CNT_POOL] < w.pool_cap ? cnt[CNT_POOL] : w.pool_cap;
    std::vector<int32_t> ar_ctg(nar), ar_gen(nar), ar_seq(nar), ar_len(nar), gen(C);
    std::vector<int64_t> ar_off(nar);
    std::vector<OutElem> pool(npool);
    if (nar) {
        be.d2h(ar_ctg.data(), w.ar_ctg, nar * 4); be.d2h(ar_gen.data(), w.ar_gen, nar * 4); be.d2h(ar_seq.data(), w.ar_seq, nar * 4);
        be.d2h(ar_len.data(), w.ar_len, nar * 4); be.d2h(ar_off.data(), w.ar_off, nar * 8);
        be.d2h(pool.data(), w.pool, npool * sizeof(OutElem));
    }
    be.d2h(gen.data(), w.all_gen, C * 4);
    std::vector<std::vector<std::pair<int32_t, int64_t>>> per(C);   // (seq, record)
    for (int64_t r = 0; r < nar; r++) {
        const int32_t c = ar_ctg[r];
        if (c < 0 || c >= C || ar_gen[r] != gen[c]) continue;
        if (ar_off[r] + ar_len[r] > npool) continue;
        per[c].push_back({ar_seq[r], r});
    }
    int64_t np = 0, ne = 0;
    for (int64_t c = 0; c < C; c++) {
        std::sort(per[c].begin(), per[c].end());
        np += (int64_t)per[c].size();
        for (auto &x : per[c]) ne += ar_len[x.second];
        out->all_path_off[c + 1] = np;
    }
    out->n_all_paths = np;
    out->all_elem_off = (int64_t *)calloc(np + 1, 8);
    out->all_elems = (aasm_out_elem *)calloc(ne + 1, sizeof(aasm_out_elem));
    int64_t ip = 0, ie = 0;
    for (int64_t c = 0; c < C; c++)
        for (auto &x : per[c]) {
            std::memcpy(out->all_elems + ie, pool.data() + ar_off[x.second], sizeof(OutElem) * (size_t)ar_len[x.second]);
            ie += ar_len[x.second];
            out->all_elem_off[++ip] = ie;
        }
    // statistics
    aasm_stats &st = out->stats;
    st.n_vertices = sz.VT; st.n_edges = sz.ET;
    st.n_heap_nodes = cnt[CNT_HEAPNODES]; st.n_paths_found = cnt[CNT_PATHS]; st.n_paths_converted = cnt[CNT_CONVERTED];
    st.n_unconnectable = cnt[CNT_UNCONN]; st.range_steps = cnt[CNT_RANGE_STEPS];
    st.ispr_edges = cnt[CNT_ISPR_E]; st.ispr_vertices = cnt[CNT_ISPR_V]; st.path_edges = cnt[CNT_PATH_E]; st.out_elems = cnt[CNT_OUT_E];
    st.pq_pushes = cnt[CNT_PQ_PUSH];
    std::vector<int32_t> ctgV(C);
    be.d2h(ctgV.data(), w.ctgV, C * 4);
    std::vector<int64_t> roff(C + 1);
    be.d2h(roff.data(), w.rec_off, (C + 1) * 8);
    for (int64_t c = 0; c < C; c++) {
        const int64_t N = roff[c + 1] - roff[c];
        if (N == 1) st.n_single++;
        if (ctgV[c] > 0) st.n_pairs += ctgV[c] - 2 - N;
        if (out->ctg_status[c] != 0) st.n_internal_errors++;
    }
    return AASM_OK;
}

}  // namespace aasm
