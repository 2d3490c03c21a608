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

// cut points [0 = c_0 < c_1 < ... < c_n = C] of a contiguous partition balanced on the cost prefix sums
void partition_by_cost(const double *cost, int64_t C, int n_shards, int64_t *cuts) {
    std::vector<double> pre((size_t)C + 1, 0.0);
    for (int64_t c = 0; c < C; c++) pre[(size_t)c + 1] = pre[(size_t)c] + cost[c];
    cuts[0] = 0; cuts[n_shards] = C;
    for (int d = 1; d < n_shards; d++) {
        const double target = pre[(size_t)C] * d / n_shards;
        int64_t c = std::lower_bound(pre.begin(), pre.end(), target) - pre.begin();
        // the cut that leaves the smaller imbalance: before or after the contig that straddles the target
        if (c > 0 && c <= C && target - pre[(size_t)c - 1] < pre[(size_t)c] - target) c--;
        if (c <= cuts[d - 1]) c = cuts[d - 1] + 1;
        if (c > C - (n_shards - d)) c = C - (n_shards - d);
        cuts[d] = c;
    }
}

}  // namespace aasm

extern "C" {

int aasm_contig_costs(const aasm_batch_in *in, double *cost) {
    if (!in || !cost || in->n_contigs <= 0 || !in->ctg_rec_off || !in->qry_str || !in->qry_end) return AASM_E_INVAL;
    aasm::contig_costs(in, cost);
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
