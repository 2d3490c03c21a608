// Diagnostic: how fast can T threads put one 3 GB file into the page cache of this box's filesystem?
//   g++ -O2 -pthread tools/write_probe.cpp -o /tmp/write_probe && /tmp/write_probe <dir> [GB]
#include <fcntl.h>
#include <unistd.h>
#include <sys/stat.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const size_t total = (size_t)((argc > 2 ? atof(argv[2]) : 3.0) * (1ull << 30));
    const size_t chunk = 8ull << 20;
    char *src = (char *)aligned_alloc(4096, chunk);
    memset(src, 'x', chunk);
    struct Mode { const char *name; int flags; bool prealloc; bool direct; };
    const Mode modes[] = {{"pwrite", 0, false, false}, {"fallocate+pwrite", 0, true, false}, {"ftruncate+pwrite", 0, false, false}, {"O_DIRECT+fallocate", O_DIRECT, true, true}};
    for (int mi = 0; mi < 4; mi++)
        for (int T : {1, 4, 16, 64}) {
            const Mode &m = modes[mi];
            const std::string path = dir + "/write_probe.bin";
            unlink(path.c_str());
            const double t0 = now();
            int fd = open(path.c_str(), O_CREAT | O_WRONLY | O_TRUNC | m.flags, 0644);
            if (fd < 0) { printf("%-20s T=%2d: open failed (%s)\n", m.name, T, strerror(errno)); continue; }
            if (m.prealloc && posix_fallocate(fd, 0, (off_t)total) != 0) { printf("%-20s: fallocate failed\n", m.name); close(fd); continue; }
            if (mi == 2 && ftruncate(fd, (off_t)total) != 0) { close(fd); continue; }
            const double t1 = now();
            std::vector<std::thread> th;
            bool fail = false;
            const size_t nchunks = total / chunk;
            for (int t = 0; t < T; t++)
                th.emplace_back([&, t] {
                    for (size_t c = t; c < nchunks; c += T)
                        if (pwrite(fd, src, chunk, (off_t)(c * chunk)) != (ssize_t)chunk) { fail = true; return; }
                });
            for (auto &x : th) x.join();
            const double t2 = now();
            close(fd);
            printf("%-20s T=%2d: setup %.3f s, write %.3f s = %.2f GB/s%s\n", m.name, T, t1 - t0, t2 - t1, total / (t2 - t1) / 1e9, fail ? " (FAILED)" : "");
            fflush(stdout);
        }
    unlink((dir + "/write_probe.bin").c_str());
    return 0;
}
