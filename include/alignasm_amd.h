/*
 * alignasm_amd.h -- C-ABI of the MI355X-native per-contig path-inference solver.
 *
 * Drop-in boundary for ONE hot path of ACCtools/alignasm: the per-contig solver
 *
 *     void solve_ctg_read(std::vector<PafReadData>& in,
 *                         std::vector<PafOutputData>& out,
 *                         std::vector<PafOutputData>& alt_out,
 *                         std::vector<std::vector<PafOutputData>>& max_out)
 *
 * (reference: src/paf_data.hpp:193 declaration, src/paf_data.cpp:223 definition,
 *  call sites src/alignasm.cpp:357,373,391).  The reference has no FFI layer; this
 * header is the FFI a maintainer would bind (INTEGRATION.md shows the glue).
 * The GPU wants many contigs per launch, so the unit is a BATCH of contigs in flat
 * SoA arrays with per-contig offsets instead of one std::vector per call.
 *
 * Conventions (all taken from the reference):
 *  - every interval is CLOSED [str, end], 0-based        (src/alignasm.cpp:141-151)
 *  - for '-' strand records ref_str > ref_end: ref_str is the reference position of
 *    qry_str                                              (src/alignasm.cpp:155-159)
 *  - records of one contig are given in INPUT order; position inside the contig is
 *    the reference's PafReadData::ctg_index               (src/alignasm.cpp:138)
 *  - match ranges are what get_overlap_range() produces   (src/paf_data.cpp:90-123):
 *    one (qry_l, qry_r, ref_l) triple per ':' op of the cs tag, query-oriented;
 *    the reference-side right end is ref_l + (qry_r-qry_l)*step and is not passed.
 *
 * No exceptions cross this ABI; every entry point returns 0 or a negative AASM_E_*.
 * All arithmetic on the path is int64 / int32 integer work (no floating point).
 */
#ifndef ALIGNASM_AMD_H
#define ALIGNASM_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AASM_ABI_VERSION 3

/* error codes */
#define AASM_OK              0
#define AASM_E_INVAL        -1   /* bad argument / inconsistent offsets            */
#define AASM_E_NODEVICE     -2   /* no HIP device / HIP runtime error at init      */
#define AASM_E_HIP          -3   /* HIP runtime error during a solve               */
#define AASM_E_NOMEM        -4   /* device or host allocation failed               */
#define AASM_E_OVERFLOW     -5   /* input outside the supported range (coordinates >= 2^40, a contig of >= 2^31
                                    records / vertices / heap nodes) or an internal pool overflow        */
#define AASM_E_INTERNAL     -6   /* "must not happen" state of the reference hit   */
#define AASM_E_PARSE        -7   /* PAF / cs tag parse error (host codec)          */
#define AASM_E_IO           -8

/* ---- input: a batch of contigs -------------------------------------------------
 * Mirrors the solver-relevant fields of PafReadData (src/paf_data.hpp:51-67).
 * All pointers are HOST pointers for aasm_solve_batch() and DEVICE pointers for
 * aasm_solve_device().                                                           */
typedef struct aasm_batch_in {
    int64_t n_contigs;
    int64_t n_records;            /* == ctg_rec_off[n_contigs]                     */
    int64_t n_ranges;             /* == rec_rng_off[n_records]                     */
    const int64_t *ctg_rec_off;   /* [n_contigs+1] record offsets, non-decreasing  */
    const int64_t *qry_str;       /* [n_records]  PafReadData::qry_str             */
    const int64_t *qry_end;       /* [n_records]  PafReadData::qry_end (closed)    */
    const int64_t *ref_str;       /* [n_records]  ref position of qry_str          */
    const int64_t *ref_end;       /* [n_records]  ref position of qry_end          */
    const int64_t *qry_total;     /* [n_records]  PafReadData::qry_total_length    */
    const int32_t *ref_chr;       /* [n_records]  dense reference-name id          */
    const uint8_t *aln_fwd;       /* [n_records]  1 = '+', 0 = '-'                 */
    const uint8_t *map_qul;       /* [n_records]  mapping quality                  */
    const int64_t *rec_rng_off;   /* [n_records+1] match-range offsets             */
    const int64_t *rng_qry_l;     /* [n_ranges]   qry_overlap_range[k].first       */
    const int64_t *rng_qry_r;     /* [n_ranges]   qry_overlap_range[k].second      */
    const int64_t *rng_ref_l;     /* [n_ranges]   ref_overlap_range[k].first       */
    /* Alternative to the three rng_* arrays (ABI 2): when rng_qry_l is NULL and cs_text is
     * not, the device derives the match ranges itself from the records' short-form cs tags
     * (get_overlap_range, paf_data.cpp:90-123, kernel aasm_k0_cs_ranges).  rec_rng_off and
     * n_ranges must still be given: a record's range count is its number of ':' operations. */
    const char    *cs_text;       /* cs tags back to back, each starting "cs:Z:"    */
    const int64_t *rec_cs_off;    /* [n_records+1] offsets into cs_text            */
} aasm_batch_in;

/* ---- options --------------------------------------------------------------------*/
typedef struct aasm_opts {
    int32_t max_paths;         /* MAX_PATH_COUNT, src/paf_data.cpp:729; 0 -> 10000   */
    int32_t non_skip_linkable; /* global NON_SKIP_LINKABLE, src/paf_data.hpp:12      */
    int32_t device;            /* HIP device ordinal                                  */
    int32_t collect_timing;    /* 1: bracket every kernel with HIP events            */
    int32_t keep_debug;        /* 1: keep device intermediates for aasm_debug_fetch   */
    int32_t reserved[3];       /* test hooks, 0 in production:
                                * [0] bit 0: force the sequential selection kernel; bit 1: build every contig's heaps with the
                                *            several-waves-per-contig kernel; bit 2: none of them (default: by graph density);
                                *            bit 3: K8 on the d-ary heap queue (cross-check of the default sorted-front / sorted-runs queue);
                                *            bit 4: K8 with the 40-entry front it takes for batches of more than 3 584 contigs
                                *            bit 5: workgroups take their work items in grid order (default: XCD by XCD, aasm_gpu.hip)
                                *            bit 6: every sparse contig in the chain class (aasm_k67_chain: sweep, pre-pass and heaps of a contig
                                *            beside each other in one workgroup); bit 7: none; both: the contigs of at least the batch's mean size (default: small
                                *            batches and the long tail of big ones)
                                *            bits 8-15: 1 = the several-waves heap kernel launched in input order, a block per contig of
                                *            the batch (default: a block per contig of its class, largest node bound first); 4 / 8 / 16 =
                                *            that many waves per contig of the class (default: by how many contigs share the chip)
                                *            bit 16: rows, reversed CSR and sweep headers by the separate launches even where every contig is small
                                *            enough for one workgroup to build its graph (aasm_k46_graph)
                                * [1] > 0:   pretend that contig ranges longer than this do not fit in device
                                *            memory (exercises the range split of aasm_solve_batch)
                                * [2] bit 0: inject one failing kernel launch (must surface as AASM_E_HIP);
                                *     bit 1: aasm_solve_batch_multi wraps device ordinals around the devices that exist;
                                *     bit 2: the next scan finds its ticket counter as an aborted launch would leave it (its look-back
                                *            must give up after 10 s with AASM_E_HIP, and the next solve on the context must succeed);
                                *     bit 3: the chain class's pre-pass wave of contig 0 never publishes the root's header (the contig must end with
                                *            AASM_E_INTERNAL, nothing may hang); bit 4: ... and never reports that it is done (the heap wave gives up
                                *            after 1 s instead of its 30 s);
                                *     bit 5: the chain class's heap wave keeps its own BFS queue (default: the order comes from a wave of its
                                *            own while the class has at most 896 contigs); bit 6: its ring of parents' roots has 4 entries, not 512;
                                *     bits 8-15: d + 1 = the sort replay of duplicate-key contigs takes its heap sort
                                *            fallback after d partition levels instead of 2 lg N                          */
} aasm_opts;

/* ---- output ---------------------------------------------------------------------
 * One element == one PafOutputData (src/paf_data.hpp:90-105).                      */
typedef struct aasm_out_elem {
    int64_t edited_qry_str, edited_qry_end;
    int64_t edited_ref_str, edited_ref_end;
    int32_t ctg_index;         /* index of the record inside its contig (input order) */
    int32_t is_alt_path;       /* tp:A:S if 1 (src/alignasm.cpp:438)                   */
} aasm_out_elem;

#define AASM_N_PHASES 16
typedef struct aasm_stats {
    int64_t n_vertices;        /* sum over contigs of V = N + P + 2                   */
    int64_t n_pairs;           /* sum of P (overlap vertices)                          */
    int64_t n_edges;           /* sum of E = get_edge_count(graph)                     */
    int64_t n_heap_nodes;      /* persistent leftist-heap nodes allocated              */
    int64_t n_paths_found;     /* sum of distances.size()                              */
    int64_t n_paths_converted; /* calls of edge_path_to_paf_path                       */
    int64_t n_unconnectable;   /* overlap pairs with no cut (paf_data.cpp:373-375)     */
    int64_t n_internal_errors; /* contigs that hit a must-not-happen state             */
    int64_t n_single;          /* contigs with one record (paf_data.cpp:235-239)       */
    int64_t range_steps;       /* match-range entries visited by the merges            */
    int64_t device_bytes;      /* peak device workspace                                */
    int64_t ispr_edges;        /* K9: edges relaxed by internal_shortest_path_recover  */
    int64_t ispr_vertices;     /* K9: window vertices expanded                         */
    int64_t path_edges;        /* K9: edges of recovered + upgraded paths              */
    int64_t out_elems;         /* K9: PafOutputData elements produced by conversions   */
    int64_t pq_pushes;         /* K8: priority-queue pushes                            */
    float   phase_ms[AASM_N_PHASES];  /* per-phase kernel time (HIP events), ms       */
    float   total_ms;                 /* whole device pipeline, ms                    */
    float   reserved_f[3];
} aasm_stats;

/* phase ids for aasm_stats::phase_ms */
enum {
    AASM_PH_SORT = 0,     /* K1 sort + parts (3 kernels)                  */
    AASM_PH_PAIRS,        /* K2 overlap slots + cut merge + vertex ids    */
    AASM_PH_EDGES,        /* K3/K4 CSR build + scores                     */
    AASM_PH_REVCSR,       /* reversed CSR                                 */
    AASM_PH_SPTREE,       /* K6 aasm_k6_rev_sweep alone                   */
    AASM_PH_FWD,          /* K5/K6 aasm_k5_fwd_sweep alone                */
    AASM_PH_HEAP,         /* K7 aasm_k7_heap alone                        */
    AASM_PH_ENUM,         /* K8 aasm_k8_enum alone                        */
    AASM_PH_SELECT,       /* K9 aasm_k9_select alone                      */
    AASM_PH_GATHER,       /* output compaction                            */
    AASM_PH_HEAP_PREP,    /* SP-tree children CSR + arena sizing          */
    AASM_PH_TOPO,         /* topologically ordered CSR copy for K9        */
    AASM_PH_MISC,
    AASM_PH_CS,           /* K0 aasm_k0_cs_ranges alone (only with cs_text input) */
    AASM_PH_FINAL,        /* K9 per-contig final pick (aasm_k9_sel_final)          */
    AASM_PH_CHAIN         /* aasm_k67_chain: K6 sweep + K7 pre-pass + K7 heaps of the chain class, beside each other */
};

/* Ragged result of a batch: three lists per contig, exactly the three output
 * vectors of solve_ctg_read.  `all` is a list of paths per contig.
 * Arrays are malloc'ed by the library, released by aasm_free_out().              */
typedef struct aasm_batch_out {
    int64_t n_contigs;
    int64_t *main_off;      /* [n_contigs+1] into main_elems                       */
    int64_t *alt_off;       /* [n_contigs+1] into alt_elems                        */
    int64_t *all_path_off;  /* [n_contigs+1] into all_elem_off (paths per contig)  */
    int64_t *all_elem_off;  /* [n_all_paths+1] into all_elems                      */
    aasm_out_elem *main_elems;
    aasm_out_elem *alt_elems;
    aasm_out_elem *all_elems;
    int64_t n_all_paths;
    int32_t *ctg_status;    /* [n_contigs] 0 ok, <0 AASM_E_* for that contig       */
    aasm_stats stats;
} aasm_batch_out;

/* ---- entry points -----------------------------------------------------------------*/

/* library / device probe.  Returns AASM_OK when a gfx950-class HIP device is usable. */
int  aasm_abi_version(void);
int  aasm_device_count(void);
int  aasm_init(int device);
const char *aasm_last_error(void);

/* solve_ctg_read over a batch (replaces the dispatch loop src/alignasm.cpp:346-397).
 * Host pointers in, host arrays out (malloc'ed into *out).                           */
int  aasm_solve_batch(const aasm_batch_in *in, const aasm_opts *opts, aasm_batch_out *out);
/* The same for contigs [c0, c1) of the batch (out then covers c1 - c0 contigs): a caller that streams a file solves one
 * range while it writes the rows of the range before (aasm_writer_*).  The batch is validated with the range c0 = 0.   */
int  aasm_solve_batch_range(const aasm_batch_in *in, int64_t c0, int64_t c1, const aasm_opts *opts, aasm_batch_out *out);

/* Contig-sharded solve across n_devices GPUs of one node (devices opts->device .. +n-1):
 * static per-contig partition, one host thread + stream per device, outputs concatenated
 * in contig order; no collective (contigs are independent, src/alignasm.cpp:351-359).    */
int  aasm_solve_batch_multi(const aasm_batch_in *in, const aasm_opts *opts, int n_devices, aasm_batch_out *out);

/* The partition aasm_solve_batch_multi uses, for callers that run one process per GPU (bench.py, MPI-style
 * launchers): cost[c] = estimated GPU cost of contig c (records + a graph-density term from the part sizes);
 * cuts[0..n_shards] = cut points of the contiguous partition whose fullest block is as light as a contiguous partition
 * allows (cuts[0] = 0, cuts[n] = n_contigs).  aasm_partition_costs: the same cut for caller-supplied costs.
 * Host pointers; only ctg_rec_off, qry_str and qry_end are read.                                       */
int  aasm_contig_costs(const aasm_batch_in *in, double *cost);
int  aasm_partition_contigs(const aasm_batch_in *in, int n_shards, int64_t *cuts);
int  aasm_partition_costs(const double *cost, int64_t n_contigs, int n_shards, int64_t *cuts);

/* The solver's generic single-source shortest paths, dijkstra() (src/k_shortest_walks.hpp:69-87), over a batch of
 * graphs that may contain cycles: graph g owns vertices [g_voff[g], g_voff[g+1]) (local ids 0..), rowptr is one CSR
 * row-pointer array over all vertices, col holds LOCAL head ids, w5 five int64 per edge {qry_score, ref_score, anom,
 * qul_nonzero, qul_total}, src the local source per graph.  d5 (5 int64 per vertex; unreachable = PafDistance::max()
 * = {-1,-1,-1,-1,0}) and prev (-1 = none) are exactly the reference's return values.  The reference's CLI never
 * calls dijkstra (is_dag = true, paf_data.cpp:728); this entry exists because the solver class offers it.  */
int  aasm_sssp_dijkstra(int64_t n_graphs, const int64_t *g_voff, const int64_t *rowptr, const int32_t *col, const int64_t *w5,
                        const int32_t *src, int64_t *d5, int32_t *prev, int device);

/* Dial's bucketed BFS, k_weighted_bfs() (src/k_weighted_bfs.hpp:16-37; the solver runs it with lim = 2 on the anomaly weights,
 * src/paf_data.cpp:704-713), over a batch of digraphs that may contain cycles and parallel edges: same graph layout as
 * aasm_sssp_dijkstra, cost one int32 per edge in 0 .. lim (lim <= 7), src the local source per graph.  dist (-1 = unreachable)
 * and pre (-1 = none; LOCAL vertex ids) are exactly the vectors the reference fills - pre depends on the LIFO order inside a
 * bucket, which the kernel keeps (buckets staged in LDS, pushes compacted per bucket by ballot + prefix count). */
int  aasm_sssp_dial(int64_t n_graphs, const int64_t *g_voff, const int64_t *rowptr, const int32_t *col, const int32_t *cost,
                    const int32_t *src, int lim, int64_t *dist, int64_t *pre, int device);

/* Same, with the batch already resident in device memory (in->pointers are device
 * pointers; in->ctg_rec_off / rec_rng_off too).  `stream` is a hipStream_t (or NULL).
 * The device result stays resident in an opaque handle until fetched/freed.          */
typedef struct aasm_result aasm_result;
int  aasm_solve_device(const aasm_batch_in *dev_in, const aasm_opts *opts, void *stream,
                       aasm_result **res);
int  aasm_result_stats(const aasm_result *res, aasm_stats *stats);
int  aasm_result_fetch(aasm_result *res, aasm_batch_out *out);   /* D2H + ragged pack */
void aasm_result_free(aasm_result *res);
void aasm_free_out(aasm_batch_out *out);

/* Upload a host batch once and solve it repeatedly (benchmark path: inputs resident in
 * HBM before the timed region).  dev_view receives device pointers for aasm_solve_device. */
typedef struct aasm_upload aasm_upload;
int  aasm_upload_batch(const aasm_batch_in *host_in, int device, aasm_upload **up, aasm_batch_in *dev_view);
void aasm_upload_free(aasm_upload *up);

/* Start-up helper for a fresh process (no counterpart in the reference: its state is the CPU heap): creates the device
 * context and grows the workspace arena to `bytes` (capped at half of the free device memory) so that the first solve
 * does not pay HIP's start-up and the arena's hipMalloc calls.  Meant to run on a thread of its own while the caller
 * still reads its input (the CLI does: ~3.7 bytes of arena per byte of PAF text).  AASM_E_NOMEM is harmless here. */
int  aasm_reserve_workspace(int device, int64_t bytes);

/* Debug/parity hook: copy a named device intermediate of a result solved with
 * opts.keep_debug=1 (names listed in DESIGN.md; e.g. "perm", "csr_col", "sp_d").
 * Call with dst==NULL to get the byte size.                                           */
int64_t aasm_debug_fetch(aasm_result *res, const char *name, void *dst, int64_t dst_bytes);
/* Process-wide diagnostic counters (tests / tuning): "range_splits" (contig ranges halved after an
 * out-of-memory), "device_mallocs" (hipMalloc calls of the arenas), "stream_syncs" (host waits on a
 * pipeline stream).  Unknown name: -1.                                                         */
int64_t aasm_debug_counter(const char *name);
/* Test entry for row T1: the device's PafDistance predicates (paf_data.hpp:142-168) on n pairs of
 * {qry, ref, anom, qul_nonzero, qul_total} tuples.  out[i] bit 0: a < b in CALC_SUM mode, bit 1: a < b in
 * QRY_SCORE mode, bit 2: a == b, bit 3: K7's node-key test, bit 4: K8's queue order (equal node / index). */
int  aasm_debug_predicates(const int64_t *a, const int64_t *b, int64_t n, uint8_t *out, int device);
/* Test entry for hazard B1: K1's replay of libstdc++'s std::sort (paf_data.cpp:241-246 sorts with an unstable sort, so the
 * order of records with equal (qry_str, qry_end) is whatever that algorithm leaves) alone, on arbitrary keys.
 * rec_off[n_contigs + 1] starts at 0; perm_out[rec_off[c] + r] = the input index, relative to contig c, that ends at
 * sorted position r.  depth_test: 0, or d + 1 for a depth limit of d partition levels (as aasm_opts.reserved[2] bits 8-15). */
int  aasm_debug_sort_replay(const int64_t *rec_off, int64_t n_contigs, const int64_t *qs, const int64_t *qe, int32_t *perm_out,
                            int depth_test, int device);

/* ---- host-side codec + file contract (reference: src/paf_data.cpp:19-220,
 *      src/alignasm.cpp:76-183,398-490).  Implemented in host C++.                  */
typedef struct aasm_paf aasm_paf;    /* parsed PAF file: names, records, cs strings   */

int  aasm_paf_read(const char *path, aasm_paf **paf);              /* alignasm.cpp:76-183 */
/* Host threads of the PAF reader and the output writers (row-parallel; results do not depend
 * on it).  The reference's -t/--thread (alignasm.cpp:45-49,346-352) sizes the TBB arena that
 * runs solve_ctg_read; here that work is on the GPU and -t sizes the host codec instead.
 * 0 = the CPUs this process may use - hardware threads cut to the affinity mask and the cgroup CPU
 * quota, at most 64 (default).  Returns the previous setting. */
int  aasm_set_host_threads(int n);
int  aasm_paf_parse_mem(const char *text, int64_t len, aasm_paf **paf);
/* Reader flags.  AASM_READ_DEVICE_RANGES: do not build the match ranges on the host; rows are
 * only indexed and their ':' operations counted, aasm_paf_batch() then hands out cs_text /
 * rec_cs_off with NULL rng_* pointers and the solver parses the cs tags on the GPU (a malformed
 * tag is then reported by the solve call, AASM_E_PARSE, instead of by the reader).           */
#define AASM_READ_DEVICE_RANGES 1
int  aasm_paf_read_opts(const char *path, int flags, aasm_paf **paf);
int  aasm_paf_parse_mem_opts(const char *text, int64_t len, int flags, aasm_paf **paf);
/* --alt merge of a second PAF of sub-contig re-alignments (alignasm.cpp:186-332) */
int  aasm_paf_merge_alt(aasm_paf *paf, const char *alt_path, double alt_baseline);
int  aasm_paf_merge_alt_mem(aasm_paf *paf, const char *text, int64_t len, double alt_baseline);
void aasm_paf_free(aasm_paf *paf);
int  aasm_paf_batch(const aasm_paf *paf, aasm_batch_in *view);      /* borrowed pointers  */
int64_t aasm_paf_n_contigs(const aasm_paf *paf);
/* write <stem>.aln.paf / .aln.alt.paf / .aln.all.paf (alignasm.cpp:407-490)          */
int  aasm_paf_write_outputs(const aasm_paf *paf, const aasm_batch_out *out,
                            const char *main_path, const char *alt_path, const char *all_path);
/* The same in pieces: the three files are opened once (under temporary names), receive the rows of consecutive contig
 * ranges in order (out = the result of contigs [contig0, contig0 + out->n_contigs)), and take their final names at
 * aasm_writer_close(w, 1); close(w, 0), or any failed append, removes them.                                        */
typedef struct aasm_writer aasm_writer;
int  aasm_writer_open(const char *main_path, const char *alt_path, const char *all_path, aasm_writer **w);
int  aasm_writer_append(aasm_writer *w, const aasm_paf *paf, const aasm_batch_out *out, int64_t contig0);
int  aasm_writer_close(aasm_writer *w, int commit);
/* get_overlap_range (paf_data.cpp:90): returns #ranges or <0; arrays may be NULL      */
int64_t aasm_cs_match_ranges(const char *cs, int64_t cs_len, int aln_fwd,
                             int64_t qry_str, int64_t qry_end, int64_t ref_str, int64_t ref_end,
                             int64_t *qry_l, int64_t *qry_r, int64_t *ref_l, int64_t cap);
/* get_edited_paf_data (paf_data.cpp:125): re-cut a cs tag; returns length written     */
int64_t aasm_cs_edit(const char *cs, int64_t cs_len, int aln_fwd,
                     int64_t qry_str, int64_t qry_end,
                     int64_t e_qry_str, int64_t e_qry_end, int64_t e_ref_str, int64_t e_ref_end,
                     char *out_cs, int64_t cap, int32_t *mat_num, int32_t *aln_len, int32_t *is_cut);

/* ---- synthetic PAF generator (SURVEY.md Appendix C spec; own code) -----------------*/
typedef struct aasm_synth_cfg {
    int64_t n_contigs;
    int64_t recs_per_contig;   /* fixed size, or the mean when heavy_tail=1            */
    uint64_t seed;
    int32_t dense;             /* 0 sparse, 1 dense/high-multiplicity                  */
    int32_t heavy_tail;        /* log-normal contig sizes                              */
    int32_t dup_every;         /* >0: duplicate every n-th record on another chr (ties)*/
    int32_t reserved;          /* bit 0: shuffle records inside a contig; bit 1: no cs tags; bit 2: records only (no match
                                  ranges, no cs: the form the contig cost model reads)                            */
} aasm_synth_cfg;
int  aasm_synth_paf(const aasm_synth_cfg *cfg, aasm_paf **paf);     /* full PAF w/ cs  */
/* contigs [first, first + count) of that file: a contig has its own PRNG stream, so the range equals the same contigs of the
   whole (BASELINE configs[3] / [4]: a rank of a contig-sharded run generates only its own block of the one file)            */
int  aasm_synth_paf_range(const aasm_synth_cfg *cfg, int64_t first, int64_t count, aasm_paf **paf);
int  aasm_paf_to_text(const aasm_paf *paf, char **text, int64_t *len); /* free()       */
int  aasm_paf_save(const aasm_paf *paf, const char *path);          /* the same text written to a file (no 2 GiB limit on the caller's side) */

#ifdef __cplusplus
}
#endif
#endif /* ALIGNASM_AMD_H */
