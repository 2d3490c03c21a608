// Diagnostic: T producer threads make 1 MB chunks at about 1 GB/s each (a byte loop over a private source, like the row
// formatter) for ONE output file; who should write them?   g++ -O2 -pthread tools/write_probe2.cpp -o /tmp/wp2 && /tmp/wp2 <dir> [GB]
//   own     every producer pwrites its own chunk (offsets chained), all at once on the inode
//   mutex   the same, one pwrite at a time behind a user-space mutex (waiters sleep instead of spinning on the inode's rwsem)
//   single  one writer thread takes the chunks in order
#include <fcntl.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static const size_t CH = 1 << 20;
static void produce(char *dst, const char *src, size_t n) {          // ~1 GB/s: a dependent byte transform
    unsigned acc = 7;
    for (size_t i = 0; i < n; i++) { acc = acc * 31 + (unsigned char)src[i]; dst[i] = (char)('a' + (acc & 15)); }
}
int main(int argc, char **argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const size_t total = (size_t)((argc > 2 ? atof(argv[2]) : 2.0) * (1ull << 30));
    const int64_t NCH = (int64_t)(total / CH);
    for (const char *mode : {"none", "own", "mutex", "single"})
        for (int T : {1, 4, 8, 16, 32}) {
            const std::string path = dir + "/write_probe2.bin";
            unlink(path.c_str());
            int fd = open(path.c_str(), O_CREAT | O_WRONLY | O_TRUNC, 0644);
            if (fd < 0) return 1;
            std::vector<std::vector<char>> src(T, std::vector<char>(8 << 20, 'q')), buf(2 * T, std::vector<char>(CH));
            std::mutex mu, wmu;
            std::condition_variable cv_ready, cv_free;
            std::vector<char> ready(NCH + 1, 0);
            int64_t written = 0;
            const bool single = !strcmp(mode, "single"), own = !strcmp(mode, "own"), mtx = !strcmp(mode, "mutex");
            const double t0 = now();
            std::vector<std::thread> th;
            if (single)
                th.emplace_back([&] {
                    for (int64_t k = 0; k < NCH; k++) {
                        { std::unique_lock<std::mutex> lk(mu); cv_ready.wait(lk, [&] { return ready[k] != 0; }); }
                        if (pwrite(fd, buf[((k / T) & 1) * T + k % T].data(), CH, (off_t)(k * CH)) != (ssize_t)CH) abort();
                        std::lock_guard<std::mutex> lk(mu); written = k + 1; cv_free.notify_all();
                    }
                });
            for (int t = 0; t < T; t++)
                th.emplace_back([&, t] {
                    for (int64_t k = t; k < NCH; k += T) {
                        char *b = buf[((k / T) & 1) * T + t].data();
                        if (single) { std::unique_lock<std::mutex> lk(mu); cv_free.wait(lk, [&] { return written + 2 * (int64_t)T > k; }); }
                        produce(b, src[t].data() + ((k * 4099) % (7 << 20)), CH);
                        if (single) { std::lock_guard<std::mutex> lk(mu); ready[k] = 1; cv_ready.notify_all(); }
                        else if (own) { if (pwrite(fd, b, CH, (off_t)(k * CH)) != (ssize_t)CH) abort(); }
                        else if (mtx) { std::lock_guard<std::mutex> lk(wmu); if (pwrite(fd, b, CH, (off_t)(k * CH)) != (ssize_t)CH) abort(); }
                    }
                });
            for (auto &x : th) x.join();
            const double dt = now() - t0;
            close(fd);
            printf("%-7s T=%2d: %.3f s = %.2f GB/s\n", mode, T, dt, total / dt / 1e9);
            fflush(stdout);
        }
    unlink((dir + "/write_probe2.bin").c_str());
    return 0;
}
