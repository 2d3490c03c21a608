// alignasm_main.cpp -- `alignasm <input.paf>` command line, MI355X build.
//
// Keeps the reference's CLI surface and file contract (src/alignasm.cpp:30-75,487-490):
//   alignasm PAF_LOC [-t THREAD] [-a PAF_ALT_LOC] [-b ALT_BASELINE] [--non_skip_linkable]
//   -> <stem>.aln.paf, <stem>.aln.alt.paf, <stem>.aln.all.paf next to the input,
// and adds --max-paths K (MAX_PATH_COUNT, default 10000), --gpus N, --device D.
// -t sizes the host side (row-parallel PAF reader and writers; default: all hardware threads);
// the per-contig parallelism the reference drives with it now lives on the GPU.
// --timing prints the wall time of read / solve / write to stderr.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <filesystem>
#include <condition_variable>
#include <mutex>
#include <algorithm>
#include <thread>
#include <vector>
#include <iostream>
#include <string>

#include "../../include/alignasm_amd.h"

static void usage(std::ostream &os) {
    os << "Usage: alignasm [--help] [--version] [--thread THREAD] [--alt PAF_ALT_LOC] [--alt_baseline ALT_BASELINE] "
          "[--non_skip_linkable] [--max-paths K] [--gpus N] [--device D] [--timing] [--host-ranges] PAF_LOC\n\n"
          "Positional arguments:\n  PAF_LOC              Location of PAF file [required]\n\n"
          "Optional arguments:\n  -h, --help           shows help message and exits\n  -v, --version        prints version information and exits\n"
          "  -t, --thread THREAD  Number of host threads for reading / writing PAF [default: all]\n"
          "  -a, --alt PAF_ALT_LOC  Location of alternative PAF file\n"
          "  -b, --alt_baseline ALT_BASELINE  Baseline for coverage of alternative PAF file [default: 0.5]\n"
          "  --non_skip_linkable  no edge a -> b when a -> c -> b exists\n"
          "  --max-paths K        paths enumerated per contig (reference constant MAX_PATH_COUNT) [default: 10000]\n"
          "  --gpus N             shard contigs over N GPUs of this node [default: 1]\n"
          "  --device D           first HIP device ordinal [default: 0]\n"
          "  --timing             print read / solve / write wall time to stderr\n"
          "  --host-ranges        build the cs match ranges in the reader instead of on the GPU\n";
}

int main(int argc, char **argv) {
    std::string paf_loc, alt_loc;
    aasm_opts opts;
    std::memset(&opts, 0, sizeof(opts));
    opts.max_paths = 10000;
    int gpus = 1;
    double alt_baseline = 0.5;
    bool bad = false, use_alt = false, timing = false, host_ranges = false;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto need = [&](const char *what) -> const char * {
            if (i + 1 >= argc) { std::cerr << what << ": missing value\n"; bad = true; return "0"; }
            return argv[++i];
        };
        if (a == "-h" || a == "--help") { usage(std::cout); return 0; }
        else if (a == "-v" || a == "--version") { std::cout << "0.1.0\n"; return 0; }
        else if (a == "-t" || a == "--thread") aasm_set_host_threads(std::atoi(need("--thread")));
        else if (a == "--timing") timing = true;
        else if (a == "--host-ranges") host_ranges = true;
        else if (a == "-a" || a == "--alt") alt_loc = need("--alt");
        else if (a == "-b" || a == "--alt_baseline") alt_baseline = std::atof(need("--alt_baseline"));
        else if (a == "--non_skip_linkable") opts.non_skip_linkable = 1;
        else if (a == "--max-paths") opts.max_paths = std::atoi(need("--max-paths"));
        else if (a == "--gpus") gpus = std::atoi(need("--gpus"));
        else if (a == "--device") opts.device = std::atoi(need("--device"));
        else if (!a.empty() && a[0] == '-') { std::cerr << "Unknown argument: " << a << "\n"; bad = true; }
        else if (paf_loc.empty()) paf_loc = a;
        else { std::cerr << "Maximum number of positional arguments exceeded\n"; bad = true; }
    }
    if (bad || paf_loc.empty()) { usage(std::cerr); return 1; }                 // alignasm.cpp:59-65
    std::filesystem::path p{paf_loc};
    if (p.extension() != ".paf") {                                              // :67-72
        std::cerr << "Wrong PAF file : " << std::filesystem::absolute(p);
        usage(std::cerr);
        return 1;
    }
    if (!alt_loc.empty()) {                                                     // :186-201
        if (std::filesystem::path(alt_loc).extension() != ".paf") {
            std::cerr << "Wrong PAF file : " << std::filesystem::absolute(alt_loc);
            usage(std::cerr);
            return 1;
        }
        std::error_code ec;
        auto sz = std::filesystem::file_size(alt_loc, ec);
        use_alt = !(!ec && sz == 0);                                            // empty file == no --alt
    }
    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const auto t0 = clk::now();
    // While the reader works: HIP start-up + the workspace arena of every GPU (sized from the file: ~3.7 bytes of arena
    // per byte of PAF text on whole-genome input; too small just means the solve allocates the rest itself).
    std::vector<std::thread> warm;
    struct JoinAll { std::vector<std::thread> &v; ~JoinAll() { for (auto &t : v) if (t.joinable()) t.join(); } } join_warm{warm};   // (error exits below)
    {
        std::error_code ec;
        double text = (double)std::filesystem::file_size(p, ec);
        if (ec) text = 0;
        if (use_alt) { const auto s2 = std::filesystem::file_size(alt_loc, ec); if (!ec) text += (double)s2; }
        const int64_t per_gpu = (int64_t)(text * 3.7 / (gpus > 0 ? gpus : 1)) + ((int64_t)256 << 20);
        for (int g = 0; g < (gpus > 0 ? gpus : 1); g++) warm.emplace_back([=] { (void)aasm_reserve_workspace(opts.device + g, per_gpu); });
    }
    aasm_paf *paf = nullptr;
    // the reader only indexes the rows; the cs tags are turned into match ranges on the GPU
    int rc = aasm_paf_read_opts(std::filesystem::absolute(p).c_str(), host_ranges ? 0 : AASM_READ_DEVICE_RANGES, &paf);
    if (rc != AASM_OK) { std::cerr << aasm_last_error() << "\n"; return 1; }    // e.g. "Missing cs:Z tag ..." (:165-168)
    if (use_alt) {
        rc = aasm_paf_merge_alt(paf, std::filesystem::absolute(alt_loc).c_str(), alt_baseline);
        if (rc != AASM_OK) { std::cerr << aasm_last_error() << "\n"; aasm_paf_free(paf); return 1; }
    }
    std::cout << "File read complete" << std::endl;                              // :340
    aasm_batch_in view;
    aasm_paf_batch(paf, &view);
    std::cout << "Analyze PAF " << view.n_contigs << " data in parallel" << std::endl;   // :349
    const auto t1 = clk::now();                                                 // (the warm-up threads are not waited for: the upload runs beside them, the solve takes the context's lock after them)
    auto ap = std::filesystem::absolute(p);
    auto f_main = ap; f_main.replace_extension(".aln.paf");
    auto f_alt = ap; f_alt.replace_extension(".aln.alt.paf");
    auto f_all = ap; f_all.replace_extension(".aln.all.paf");
    double upload_s = 0, device_s = 0, fetch_s = 0, solve_busy_s = 0, write_busy_s = 0;
    int64_t n_internal = 0;
    clk::time_point t2 = t1;
    if (gpus <= 1 && view.n_contigs >= 64 && view.n_records >= (1 << 16)) {
        // ---- one GPU: the file goes through in contig ranges of about equal record counts; while the GPU solves range k
        //      (upload -> K0 .. K9 -> fetch) the host threads format and write the rows of range k - 1.  The three outputs
        //      are appended to in contig order (aasm_writer_*), so the bytes are those of the one-piece path.
        const int n_chunks = (int)std::min<int64_t>(8, std::max<int64_t>(2, view.n_records / (1 << 19)));
        std::vector<int64_t> cut(n_chunks + 1, view.n_contigs);
        cut[0] = 0;
        for (int k = 1; k < n_chunks; k++) {
            const int64_t want = view.n_records * k / n_chunks;
            int64_t c = std::lower_bound(view.ctg_rec_off, view.ctg_rec_off + view.n_contigs + 1, want) - view.ctg_rec_off;
            cut[k] = std::min<int64_t>(std::max<int64_t>(c, cut[k - 1] + 1), view.n_contigs - (n_chunks - k));
        }
        aasm_writer *wr = nullptr;
        rc = aasm_writer_open(f_main.c_str(), f_alt.c_str(), f_all.c_str(), &wr);
        if (rc != AASM_OK) { std::cerr << "alignasm: writing outputs failed: " << aasm_last_error() << "\n"; aasm_paf_free(paf); return 3; }
        std::vector<aasm_batch_out> outs(n_chunks);
        std::vector<int> rcs(n_chunks, AASM_OK);
        std::vector<std::string> errs(n_chunks);
        std::mutex mu;
        std::condition_variable cv;
        int solved = 0;                                                         // ranges [0, solved) are ready for the writer
        bool stop = false;
        std::thread solver([&] {
            for (int k = 0; k < n_chunks; k++) {
                { std::lock_guard<std::mutex> lk(mu); if (stop) break; }
                const auto a0 = clk::now();
                std::memset(&outs[k], 0, sizeof(outs[k]));
                rcs[k] = aasm_solve_batch_range(&view, cut[k], cut[k + 1], &opts, &outs[k]);
                if (rcs[k] != AASM_OK) errs[k] = aasm_last_error();
                else { upload_s += outs[k].stats.reserved_f[0] / 1e3; device_s += outs[k].stats.reserved_f[2] / 1e3; fetch_s += outs[k].stats.reserved_f[1] / 1e3; n_internal += outs[k].stats.n_internal_errors; }
                solve_busy_s += secs(a0, clk::now());
                { std::lock_guard<std::mutex> lk(mu); solved = k + 1; }
                cv.notify_all();
                if (rcs[k] != AASM_OK) break;
            }
        });
        bool said_write = false;
        int wrc = AASM_OK, src_ = AASM_OK;
        std::string fail;
        for (int k = 0; k < n_chunks; k++) {
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return solved > k; }); }
            if (rcs[k] != AASM_OK) { src_ = rcs[k]; fail = errs[k]; break; }
            if (k == n_chunks - 1) t2 = clk::now();
            if (!said_write) { std::cout << "Write output PAF file" << std::endl; said_write = true; }   // :487
            const auto a0 = clk::now();
            wrc = aasm_writer_append(wr, paf, &outs[k], cut[k]);
            write_busy_s += secs(a0, clk::now());
            if (wrc != AASM_OK) { fail = aasm_last_error(); break; }
        }
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        solver.join();
        const int crc = aasm_writer_close(wr, src_ == AASM_OK && wrc == AASM_OK ? 1 : 0);
        if (src_ == AASM_E_PARSE) { std::cerr << fail << "\n"; aasm_paf_free(paf); return 1; }             // malformed cs tag, found by the device parser
        if (src_ != AASM_OK) { std::cerr << "alignasm: solver failed (" << src_ << "): " << fail << "\n"; aasm_paf_free(paf); return 2; }
        if (n_internal) std::cerr << "alignasm: " << n_internal << " contig(s) hit an internal error state\n";
        rc = wrc != AASM_OK ? wrc : crc;
        if (rc != AASM_OK) std::cerr << "alignasm: writing outputs failed: " << (wrc != AASM_OK ? fail : std::string(aasm_last_error())) << "\n";
    } else {
        aasm_batch_out out;
        rc = aasm_solve_batch_multi(&view, &opts, gpus, &out);
        t2 = clk::now();
        if (rc == AASM_E_PARSE) { std::cerr << aasm_last_error() << "\n"; aasm_paf_free(paf); return 1; }       // malformed cs tag, found by the device parser
        if (rc != AASM_OK) { std::cerr << "alignasm: solver failed (" << rc << "): " << aasm_last_error() << "\n"; aasm_paf_free(paf); return 2; }
        if (out.stats.n_internal_errors) std::cerr << "alignasm: " << out.stats.n_internal_errors << " contig(s) hit an internal error state\n";
        upload_s = out.stats.reserved_f[0] / 1e3; device_s = out.stats.reserved_f[2] / 1e3; fetch_s = out.stats.reserved_f[1] / 1e3;
        solve_busy_s = secs(t1, t2);
        std::cout << "Write output PAF file" << std::endl;                           // :487
        rc = aasm_paf_write_outputs(paf, &out, f_main.c_str(), f_alt.c_str(), f_all.c_str());
        if (rc != AASM_OK) std::cerr << "alignasm: writing outputs failed: " << aasm_last_error() << "\n";
        write_busy_s = secs(t2, clk::now());
    }
    const auto t3 = clk::now();
    if (timing)
        std::cerr << "alignasm timing: records " << view.n_records << " contigs " << view.n_contigs << " read_s " << secs(t0, t1) << " solve_s " << solve_busy_s
                  << " (upload " << upload_s << " device " << device_s << " fetch " << fetch_s << ") write_s " << write_busy_s
                  << " overlap_s " << (solve_busy_s + write_busy_s - secs(t1, t3)) << " total_s " << secs(t0, t3) << "\n";
    // the process ends here: the GBs of parsed text and results go back to the OS in one piece
    // instead of vector by vector (1.7 s of page freeing for a 5M-record file)
    std::cout.flush(); std::cerr.flush();
    for (auto &th : warm) if (th.joinable()) th.join();
    std::_Exit(rc == AASM_OK ? 0 : 3);
}
