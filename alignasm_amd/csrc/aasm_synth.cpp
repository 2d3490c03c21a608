// aasm_synth.cpp -- deterministic synthetic PAF generator (own code).
//
// The reference ships no sample data (*.paf is git-ignored, /root/reference/.gitignore:7).
// Distribution follows SURVEY.md Appendix C / section 8(d):
//   record length U[2000,30000]; cs = alternating ":n" (n in U[20,400]) and one of
//   "*ac" (50 %), "+a{1..5}" (25 %), "-c{1..5}" (25 %), forced to end on a ':' op;
//   placement "sparse": 25 % partial overlap of 50-3000 bp, 10 % heavy overlap or
//   containment, 65 % gap 0-5000; "dense": 60 % advance U[100,len/3], else jump ~len;
//   3 % random-chromosome hit, 3 % strand flip; ref cursor follows with jitter
//   U[-500,5000]; qry_total = max end + U[0,5000]; mapq in {0,0,10,30,60,60,60};
//   22 reference names, ref_total 250,000,000.
// PRNG: xoshiro256** seeded through splitmix64 (one stream per contig, so contigs can be
// generated in parallel and a shard of a file equals the same contigs of the whole).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>

#include "aasm_paf.hpp"

namespace {

struct Rng {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t &x) {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    explicit Rng(uint64_t seed) { for (auto &v : s) v = splitmix(seed); }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    // uniform integer in [lo, hi] (inclusive); modulo bias is irrelevant here
    int64_t uni(int64_t lo, int64_t hi) { return hi <= lo ? lo : lo + (int64_t)(next() % (uint64_t)(hi - lo + 1)); }
    double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

struct CtgOut {
    std::vector<int64_t> qs, qe, rs, re, rng_off, ql, qr, rl, cs_off;
    std::vector<int32_t> chr, mat, aln;
    std::vector<uint8_t> fwd, mq;
    std::string cs;
    int64_t qtot = 0;
};

static const int kMapq[7] = {0, 0, 10, 30, 60, 60, 60};

void gen_contig(const aasm_synth_cfg &cfg, int64_t c, int64_t nrec, bool want_cs, CtgOut &o) {
    Rng rng(cfg.seed * 0x100000001B3ull + (uint64_t)c * 0x9E3779B97F4A7C15ull + 12345);
    int64_t pos = rng.uni(0, 5000);
    const int32_t home = (int32_t)rng.uni(0, 21);
    int64_t rpos = rng.uni(1000000, 50000000);
    const bool strand = true;
    int64_t max_end = 0;
    o.rng_off.push_back(0);
    o.cs_off.push_back(0);
    std::vector<int64_t> oq, olen, oref;         // per ':' op: query offset, length, ref offset (cs order)
    char buf[32];
    for (int64_t i = 0; i < nrec; i++) {
        const int64_t ql = rng.uni(2000, 30000);
        // ---- cs ops (in cs / reference-forward order)
        oq.clear(); olen.clear(); oref.clear();
        const size_t cs_begin = o.cs.size();
        if (want_cs) o.cs += "cs:Z:";
        int64_t q = 0, r = 0, m = 0, a = 0;
        bool last_match = false;
        while (q < ql) {
            const int64_t n = std::min(ql - q, rng.uni(20, 400));
            oq.push_back(q); olen.push_back(n); oref.push_back(r);
            if (want_cs) { int k = std::snprintf(buf, sizeof buf, ":%lld", (long long)n); o.cs.append(buf, k); }
            q += n; r += n; m += n; a += n; last_match = true;
            if (q >= ql) break;
            const double t = rng.unit();
            if (t < 0.5) { if (want_cs) o.cs += "*ac"; q += 1; r += 1; a += 1; last_match = false; }
            else if (t < 0.75) {
                const int64_t k = std::min(ql - q - 1, rng.uni(1, 5));
                if (k > 0) { if (want_cs) { o.cs += '+'; o.cs.append((size_t)k, 'a'); } q += k; a += k; last_match = false; }
            } else {
                const int64_t k = rng.uni(1, 5);
                if (want_cs) { o.cs += '-'; o.cs.append((size_t)k, 'c'); }
                r += k; a += k; last_match = false;
            }
        }
        if (!last_match) {
            oq.push_back(q); olen.push_back(1); oref.push_back(r);
            if (want_cs) o.cs += ":1";
            q += 1; r += 1; m += 1; a += 1;
        }
        (void)cs_begin;
        // ---- placement on the reference
        const double u = rng.unit();
        int32_t ch = home; bool st = strand; int64_t rs = rpos;
        if (u < 0.03) { ch = (int32_t)rng.uni(0, 21); rs = rng.uni(1000000, 50000000); }
        else if (u < 0.06) st = !strand;
        const int32_t mapq = kMapq[rng.uni(0, 6)];
        auto emit = [&](int32_t chr_id, int64_t ref_start, bool fwd) {
            // PAF columns: [pos, pos+q) on the query, [ref_start, ref_start+r) on the reference.
            const int64_t qs = pos, qe = pos + q - 1;
            int64_t r_s = ref_start, r_e = ref_start + r - 1;
            if (!fwd) std::swap(r_s, r_e);             // ref_str = ref position of qry_str
            o.qs.push_back(qs); o.qe.push_back(qe); o.rs.push_back(r_s); o.re.push_back(r_e);
            o.chr.push_back(chr_id); o.mat.push_back((int32_t)m); o.aln.push_back((int32_t)a);
            o.fwd.push_back(fwd ? 1 : 0); o.mq.push_back((uint8_t)mapq);
            // match ranges in QUERY orientation (get_overlap_range, paf_data.cpp:90-123):
            // '+' walks the cs ops forward; '-' walks them backward with ref descending.
            const int64_t nops = (int64_t)oq.size();
            if (fwd) {
                for (int64_t k = 0; k < nops; k++) {
                    o.ql.push_back(qs + oq[k]); o.qr.push_back(qs + oq[k] + olen[k] - 1); o.rl.push_back(ref_start + oref[k]);
                }
            } else {
                // cs op k covers ref [ref_start+oref[k], +len) and, reversed, query
                // [qs + (q - oq[k] - len), ...]; its query-first base maps to the ref LAST base.
                for (int64_t k = nops - 1; k >= 0; k--) {
                    const int64_t qoff = q - oq[k] - olen[k];
                    o.ql.push_back(qs + qoff); o.qr.push_back(qs + qoff + olen[k] - 1);
                    o.rl.push_back(ref_start + oref[k] + olen[k] - 1);
                }
            }
            o.rng_off.push_back((int64_t)o.ql.size());
            max_end = std::max(max_end, pos + q);
        };
        emit(ch, rs, st);
        if (want_cs) o.cs_off.push_back((int64_t)o.cs.size());
        if (cfg.dup_every > 0 && (i % cfg.dup_every) == (cfg.dup_every - 1)) {
            // tie maker: same query interval and cs on another chromosome
            int32_t ch2 = (int32_t)((ch + 1 + rng.uni(0, 20)) % 22);
            int64_t rs2 = rng.uni(1000000, 50000000);
            if (want_cs) {
                const int64_t b = o.cs_off[o.cs_off.size() - 2], e = o.cs_off.back();
                o.cs.append(o.cs, (size_t)b, (size_t)(e - b));
            }
            emit(ch2, rs2, st);
            if (want_cs) o.cs_off.push_back((int64_t)o.cs.size());
        }
        // ---- advance the query cursor
        const double step = rng.unit();
        if (cfg.dense) {
            if (step < 0.6) pos += rng.uni(100, std::max<int64_t>(101, q / 3));
            else pos += q + rng.uni(-(q / 4), 2000);
        } else {
            if (step < 0.25) pos += q - rng.uni(50, std::min<int64_t>(q - 1, 3000));
            else if (step < 0.35) pos += rng.uni(10, q / 2);
            else pos += q + rng.uni(0, 5000);
        }
        if (ch == home && st == strand) rpos = rs + r + rng.uni(-500, 5000);
        else rpos = rpos + q;
    }
    o.qtot = max_end + rng.uni(0, 5000);
    if (cfg.reserved & 1) {
        // shuffle record order inside the contig (PAF input order is arbitrary)
        const int64_t n = (int64_t)o.qs.size();
        std::vector<int64_t> perm(n);
        for (int64_t i = 0; i < n; i++) perm[i] = i;
        for (int64_t i = n - 1; i > 0; i--) std::swap(perm[i], perm[rng.uni(0, i)]);
        CtgOut s;
        s.qtot = o.qtot; s.rng_off.push_back(0); s.cs_off.push_back(0);
        for (int64_t k = 0; k < n; k++) {
            const int64_t i = perm[k];
            s.qs.push_back(o.qs[i]); s.qe.push_back(o.qe[i]); s.rs.push_back(o.rs[i]); s.re.push_back(o.re[i]);
            s.chr.push_back(o.chr[i]); s.mat.push_back(o.mat[i]); s.aln.push_back(o.aln[i]);
            s.fwd.push_back(o.fwd[i]); s.mq.push_back(o.mq[i]);
            for (int64_t t = o.rng_off[i]; t < o.rng_off[i + 1]; t++) { s.ql.push_back(o.ql[t]); s.qr.push_back(o.qr[t]); s.rl.push_back(o.rl[t]); }
            s.rng_off.push_back((int64_t)s.ql.size());
            if (want_cs) { s.cs.append(o.cs, (size_t)o.cs_off[i], (size_t)(o.cs_off[i + 1] - o.cs_off[i])); s.cs_off.push_back((int64_t)s.cs.size()); }
        }
        o = std::move(s);
    }
}

}  // namespace

// Contigs [first, first + count) of the file `cfg` describes (a contig has its own PRNG stream, so a range of the file equals
// the same contigs of the whole; names keep the file's numbering).  cfg->reserved bit 2: records only - no match ranges, no cs
// tags (what the contig cost model reads: a rank of a sharded run cuts the WHOLE file from this form, then generates its block).
extern "C" int aasm_synth_paf_range(const aasm_synth_cfg *cfg, int64_t first, int64_t count, aasm_paf **out) {
    if (!cfg || !out || cfg->n_contigs <= 0 || cfg->recs_per_contig <= 0) return AASM_E_INVAL;
    if (first < 0 || count <= 0 || first > cfg->n_contigs - count) return AASM_E_INVAL;
    const bool recs_only = (cfg->reserved & 4) != 0;
    const bool want_cs = !(cfg->reserved & 2) && !recs_only;
    const int64_t C = cfg->n_contigs;
    // contig sizes
    std::vector<int64_t> sizes(C, cfg->recs_per_contig);
    if (cfg->heavy_tail) {
        Rng rng(cfg->seed ^ 0xABCDEF12345ull);
        std::vector<double> raw(C);
        double tot = 0;
        for (int64_t c = 0; c < C; c++) {
            double u1 = std::max(rng.unit(), 1e-12), u2 = rng.unit();
            double z = std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
            double v = std::exp(std::log(600.0) + z);
            v = std::min(std::max(v, 1.0), 8000.0);
            raw[c] = v; tot += v;
        }
        const double scale = (double)(cfg->recs_per_contig * C) / tot;
        for (int64_t c = 0; c < C; c++) sizes[c] = std::max<int64_t>(1, (int64_t)std::llround(raw[c] * scale));
    }
    std::vector<CtgOut> parts(count);
    unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (count < 8) nt = 1;
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t] {
                for (int64_t c = t; c < count; c += nt) {
                    CtgOut &p = parts[c];
                    gen_contig(*cfg, first + c, sizes[first + c], want_cs, p);
                    if (recs_only) {                                 // keep the records, drop the ranges
                        std::vector<int64_t>().swap(p.ql); std::vector<int64_t>().swap(p.qr); std::vector<int64_t>().swap(p.rl);
                        std::fill(p.rng_off.begin(), p.rng_off.end(), 0);
                    }
                }
            });
        for (auto &x : th) x.join();
    }
    aasm_paf *paf = new aasm_paf();
    paf->has_cs = want_cs;
    for (int i = 1; i <= 22; i++) paf->chr_name.push_back("chr" + std::to_string(i));
    int64_t R = 0, NR = 0; size_t CS = 0;
    for (auto &p : parts) { R += (int64_t)p.qs.size(); NR += (int64_t)p.ql.size(); CS += p.cs.size(); }
    paf->qry_str.reserve(R); paf->qry_end.reserve(R); paf->ref_str.reserve(R); paf->ref_end.reserve(R);
    paf->qry_total.reserve(R); paf->ref_total.reserve(R); paf->ref_chr.reserve(R); paf->mat_num.reserve(R);
    paf->aln_len.reserve(R); paf->row_index.reserve(R); paf->aln_fwd.reserve(R); paf->map_qul.reserve(R);
    paf->cord_type.reserve(R); paf->rec_rng_off.reserve(R + 1); paf->cs_off.reserve(R + 1);
    paf->rng_qry_l.reserve(NR); paf->rng_qry_r.reserve(NR); paf->rng_ref_l.reserve(NR);
    paf->cs_pool.reserve(CS);
    paf->ctg_rec_off.push_back(0); paf->rec_rng_off.push_back(0); paf->cs_off.push_back(0);
    char name[32];
    int32_t row = 0;
    for (int64_t c = 0; c < count; c++) {
        CtgOut &p = parts[c];
        std::snprintf(name, sizeof name, "ptg%06lldl", (long long)(first + c));
        paf->ctg_name.push_back(name);
        const int64_t n = (int64_t)p.qs.size();
        const int64_t rng_base = (int64_t)paf->rng_qry_l.size();
        const int64_t cs_base = (int64_t)paf->cs_pool.size();
        for (int64_t i = 0; i < n; i++) {
            paf->qry_str.push_back(p.qs[i]); paf->qry_end.push_back(p.qe[i]);
            paf->ref_str.push_back(p.rs[i]); paf->ref_end.push_back(p.re[i]);
            paf->qry_total.push_back(p.qtot); paf->ref_total.push_back(250000000);
            paf->ref_chr.push_back(p.chr[i]); paf->mat_num.push_back(p.mat[i]); paf->aln_len.push_back(p.aln[i]);
            paf->row_index.push_back(row++); paf->cord_type.push_back(0);
            paf->aln_fwd.push_back(p.fwd[i]); paf->map_qul.push_back(p.mq[i]);
            paf->rec_rng_off.push_back(rng_base + p.rng_off[i + 1]);
            paf->cs_off.push_back(want_cs ? cs_base + p.cs_off[i + 1] : 0);
        }
        paf->rng_qry_l.insert(paf->rng_qry_l.end(), p.ql.begin(), p.ql.end());
        paf->rng_qry_r.insert(paf->rng_qry_r.end(), p.qr.begin(), p.qr.end());
        paf->rng_ref_l.insert(paf->rng_ref_l.end(), p.rl.begin(), p.rl.end());
        if (want_cs) paf->cs_pool.insert(paf->cs_pool.end(), p.cs.begin(), p.cs.end());
        paf->ctg_rec_off.push_back((int64_t)paf->qry_str.size());
        CtgOut().qs.swap(p.qs);   // release per-contig scratch early
        p = CtgOut();
    }
    *out = paf;
    return AASM_OK;
}

extern "C" int aasm_synth_paf(const aasm_synth_cfg *cfg, aasm_paf **out) {
    if (!cfg) return AASM_E_INVAL;
    return aasm_synth_paf_range(cfg, 0, cfg->n_contigs, out);
}
