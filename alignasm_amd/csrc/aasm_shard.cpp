// aasm_shard.cpp -- static per-contig partition of one batch over the GPUs of a node.
//
// The reference runs one TBB task per contig over ONE input (src/alignasm.cpp:346-361); here the
// contig list is cut into one contiguous block per device.  No collective is involved: contigs are
// independent, every device solves its block, the host concatenates the outputs in contig order.
//
// Cost model (a*N + c*E_est).  What a contig costs on the GPU is set by its graph,
// not by its record count: the dense C5 contigs take ~40x the time of sparse C3 contigs of the same
// N = 1000, because inside a part (a maximal chain of query-overlapping records,
// paf_data.cpp:248-261) every record links to every later disjoint record (:598-651), so E grows
// with the SQUARE of the part size, and the sidetrack heaps (K7, the dominant kernel) grow like
// E * log E.  E is only known on the device after K4's count pass, but its driver, the part sizes,
// costs one host sort per contig:
//     E_est = sum over parts p of  B_p * (B_p - 1) / 2  +  B_p * B_{p+1}      (intra-part + next-part links)
//     cost  = N + 0.25 * E_est * log2(2 + E_est / N)
// (sparse synthetic contigs: E_est/N ~ 2, dense ones ~ 50: cost ratio ~ 30x, measured time ratio ~ 40x).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

#include "aasm_paf.hpp"

namespace aasm {

static double contig_cost(const int64_t *qs, const int64_t *qe, int64_t N, std::vector<std::pair<int64_t, int64_t>> &tmp) {
    if (N <= 1) return 1.0;
    tmp.resize((size_t)N);
    for (int64_t i = 0; i < N; i++) tmp[(size_t)i] = {qs[i], qe[i]};
    std::sort(tmp.begin(), tmp.end());
    double e_est = 0.0;
    int64_t part_end = -1, B = 0, prevB = 0;
    auto close_part = [&]() { e_est += 0.5 * (double)B * (double)(B - 1) + (double)prevB * (double)B; prevB = B; B = 0; };
    for (int64_t i = 0; i < N; i++) {
        if (part_end < tmp[(size_t)i].first && B > 0) close_part();
        B++;
        part_end = std::max(part_end, tmp[(size_t)i].second);
    }
    close_part();
    return (double)N + 0.25 * e_est * std::log2(2.0 + e_est / (double)N);
}

void contig_costs(const aasm_batch_in *in, double *cost) {
    const int64_t C = in->n_contigs;
    int T = host_threads();
    if (in->n_records < (1 << 16)) T = 1;
    std::vector<std::thread> th;
    auto work = [&](int t) {
        std::vector<std::pair<int64_t, int64_t>> tmp;
        for (int64_t c = C * t / T; c < C * (t + 1) / T; c++) {
            const int64_t r0 = in->ctg_rec_off[c];
            cost[c] = contig_cost(in->qry_str + r0, in->qry_end + r0, in->ctg_rec_off[c + 1] - r0, tmp);
        }
    };
    for (int t = 1; t < T; t++) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
}

// cut points [0 = c_0 < c_1 < ... < c_n = C] of the contiguous partition with the SMALLEST possible fullest block
// (the linear partition problem): bisection on the block load L, feasibility by the greedy sweep "close a block when
// the next contig would push it over L".  Every block gets at least one contig.  On files with >= 50 contigs per block
// the fullest block stays within a few percent of what longest-processing-time-first bin packing reaches without the
// contiguity constraint (tests/test_sharding.py), and contiguous blocks need no gather of the inputs and no scatter
// of the outputs.
static int sweep_blocks(const std::vector<double> &pre, int64_t C, double L, int n_shards, int64_t *cuts) {
    int used = 0;
    int64_t c = 0;
    while (c < C) {
        if (used == n_shards) return n_shards + 1;                  // does not fit
        // furthest end e > c with pre[e] - pre[c] <= L (at least one contig), leaving one contig for every later block
        int64_t e = std::upper_bound(pre.begin() + c + 1, pre.end(), pre[(size_t)c] + L) - pre.begin() - 1;
        if (e <= c) e = c + 1;
        if (cuts) cuts[used + 1] = e;
        c = e; used++;
    }
    return used;
}
void partition_by_cost(const double *cost, int64_t C, int n_shards, int64_t *cuts) {
    std::vector<double> pre((size_t)C + 1, 0.0);
    double big = 0.0;
    for (int64_t c = 0; c < C; c++) { pre[(size_t)c + 1] = pre[(size_t)c] + cost[c]; big = std::max(big, cost[c]); }
    double lo = std::max(big, pre[(size_t)C] / n_shards), hi = pre[(size_t)C];
    for (int it = 0; it < 60 && hi - lo > 1e-9 * hi; it++) {
        const double mid = 0.5 * (lo + hi);
        if (sweep_blocks(pre, C, mid, n_shards, nullptr) <= n_shards) hi = mid; else lo = mid;
    }
    std::vector<int64_t> tmp((size_t)n_shards + 2, 0);
    const int used = sweep_blocks(pre, C, hi, n_shards, tmp.data());
    cuts[0] = 0;
    for (int d = 1; d <= used; d++) cuts[d] = tmp[(size_t)d];
    // fewer blocks than shards (a few huge contigs): split the blocks with the most contigs until every shard has one
    int n = used;
    while (n < n_shards) {
        int best = -1; int64_t bl = 1;
        for (int d = 0; d < n; d++) if (cuts[d + 1] - cuts[d] > bl) { bl = cuts[d + 1] - cuts[d]; best = d; }
        if (best < 0) break;                                         // (C >= n_shards is the caller's precondition)
        // cut block `best` where its two halves are closest in cost
        const double half = 0.5 * (pre[(size_t)cuts[best]] + pre[(size_t)cuts[best + 1]]);
        int64_t m = std::lower_bound(pre.begin() + cuts[best] + 1, pre.begin() + cuts[best + 1], half) - pre.begin();
        if (m >= cuts[best + 1]) m = cuts[best + 1] - 1;
        if (m <= cuts[best]) m = cuts[best] + 1;
        for (int d = n; d > best; d--) cuts[d + 1] = cuts[d];
        cuts[best + 1] = m;
        n++;
    }
    cuts[n_shards] = C;
}

}  // namespace aasm

extern "C" {

int aasm_contig_costs(const aasm_batch_in *in, double *cost) {
    if (!in || !cost || in->n_contigs <= 0 || !in->ctg_rec_off || !in->qry_str || !in->qry_end) return AASM_E_INVAL;
    aasm::contig_costs(in, cost);
    return AASM_OK;
}

int aasm_partition_costs(const double *cost, int64_t n_contigs, int n_shards, int64_t *cuts) {
    if (!cost || !cuts || n_shards < 1 || n_contigs < n_shards) return AASM_E_INVAL;
    aasm::partition_by_cost(cost, n_contigs, n_shards, cuts);
    return AASM_OK;
}

int aasm_partition_contigs(const aasm_batch_in *in, int n_shards, int64_t *cuts) {
    if (!in || !cuts || n_shards < 1 || in->n_contigs < n_shards) return AASM_E_INVAL;
    std::vector<double> cost((size_t)in->n_contigs);
    int rc = aasm_contig_costs(in, cost.data());
    if (rc != AASM_OK) return rc;
    aasm::partition_by_cost(cost.data(), in->n_contigs, n_shards, cuts);
    return AASM_OK;
}

}  // extern "C"
