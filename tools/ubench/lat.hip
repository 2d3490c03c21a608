// Scratch microbenchmark (not product code): latencies a lone wave sees on MI355X.
//   hipcc --offload-arch=gfx950 -O3 -o lat lat.hip && ./lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_fill(int *p, size_t n, int stride) {   // pointer-chase ring: p[i] = (i + stride) % n, written by a previous kernel
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (int)((i + stride) % n);
}
// one wave per block; every wave chases its own region
__global__ void __launch_bounds__(64) k_chase(const int *p, size_t region, int steps, long long *out) {
    const int *q = p + (size_t)blockIdx.x * region;
    int idx = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; s++) { idx = __builtin_amdgcn_readfirstlane(q[idx]); }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = idx; }
}
// store then wait for it (vmcnt(0)), repeated
__global__ void __launch_bounds__(64) k_store_drain(int *p, size_t region, int steps, int lanes, long long *out) {
    int *q = p + (size_t)blockIdx.x * region;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; s++) {
        if ((int)threadIdx.x < lanes) q[(size_t)s * 64 + threadIdx.x] = s;
        __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) expcnt(0) lgkmcnt(0)
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x * 2] = t1 - t0;
}
// store, then an independent load of a DIFFERENT (warm) line, wait for the load only (compiler emits vmcnt(0) anyway)
__global__ void __launch_bounds__(64) k_store_then_load(int *p, const int *warm, size_t region, int steps, long long *out) {
    int *q = p + (size_t)blockIdx.x * region;
    int acc = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; s++) {
        q[(size_t)s * 64 + threadIdx.x] = s;
        acc += __builtin_amdgcn_readfirstlane(warm[(acc & 1) + blockIdx.x * 64]);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = acc; }
}
__global__ void __launch_bounds__(64) k_lds(int steps, long long *out) {
    __shared__ int sh[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) sh[i] = (i * 7 + 1) & 1023;
    __syncthreads();
    int idx = threadIdx.x & 1;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; s++) idx = __builtin_amdgcn_readfirstlane(sh[idx]);
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = idx; }
}
__global__ void __launch_bounds__(64) k_alu(int steps, long long *out, int mode) {
    int x = threadIdx.x, y = blockIdx.x;   // y uniform -> SALU chain; x divergent -> VALU chain
    long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0) { for (int s = 0; s < steps; s++) { x = x * 3 + 1; x ^= x >> 3; x = x * 5 + 7; x ^= x >> 2; } }
    else { for (int s = 0; s < steps; s++) { y = y * 3 + 1; y ^= y >> 3; y = y * 5 + 7; y ^= y >> 2; } }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = x + y; }
}
static double mean(const std::vector<long long> &v, int n) { double s = 0; for (int i = 0; i < n; i++) s += (double)v[2 * i]; return s / n; }
int main() {
    const int steps = 2000;
    for (int waves : {1, 256, 625, 5000}) {
        const size_t region = 1 << 16;                       // ints per wave (256 KB): lines never revisited within a chase of stride 1024+
        int *p; long long *out; int *warm;
        CK(hipMalloc(&p, (size_t)waves * region * 4)); CK(hipMalloc(&out, (size_t)waves * 16)); CK(hipMalloc(&warm, (size_t)waves * 256 + 1024));
        CK(hipMemset(warm, 0, (size_t)waves * 256 + 1024));
        std::vector<long long> h((size_t)waves * 2);
        auto fetch = [&]() { CK(hipDeviceSynchronize()); CK(hipMemcpy(h.data(), out, (size_t)waves * 16, hipMemcpyDeviceToHost)); };
        // (a) dependent global loads, data written by the previous kernel, every step a new 4 KB-distant line (L2-cold after the kernel boundary)
        k_fill<<<(unsigned)((waves * region + 255) / 256), 256>>>(p, (size_t)waves * region, 0); CK(hipDeviceSynchronize());
        for (size_t w = 0; w < (size_t)waves; w++) ;  // (ring is global; per-wave rings below)
        {   // per-wave ring with stride 1031 ints
            std::vector<int> hp(region);
            for (size_t i = 0; i < region; i++) hp[i] = (int)((i + 1031) % region);
            for (int w = 0; w < waves && w < 8; w++) CK(hipMemcpy(p + (size_t)w * region, hp.data(), region * 4, hipMemcpyHostToDevice));
            for (int w = 8; w < waves; w++) CK(hipMemcpyAsync(p + (size_t)w * region, p, region * 4, hipMemcpyDeviceToDevice));
            CK(hipDeviceSynchronize());
        }
        k_chase<<<waves, 64>>>(p, region, steps, out); fetch();
        double cold = mean(h, waves) / steps;
        k_chase<<<waves, 64>>>(p, region, steps, out); fetch();   // same lines again: L2 / MALL resident now?
        double again = mean(h, waves) / steps;
        // small ring: 64 lines -> L2/L1 hits
        {   std::vector<int> hp(region, 0);
            for (int i = 0; i < 64; i++) hp[i * 32] = ((i + 1) % 64) * 32;
            for (int w = 0; w < waves && w < 8; w++) CK(hipMemcpy(p + (size_t)w * region, hp.data(), region * 4, hipMemcpyHostToDevice));
            for (int w = 8; w < waves; w++) CK(hipMemcpyAsync(p + (size_t)w * region, p, region * 4, hipMemcpyDeviceToDevice));
            CK(hipDeviceSynchronize()); }
        k_chase<<<waves, 64>>>(p, region, steps, out); fetch();
        double hot = mean(h, waves) / steps;
        k_store_drain<<<waves, 64>>>(p, region, 1000, 64, out); fetch(); double st64 = mean(h, waves) / 1000;
        k_store_drain<<<waves, 64>>>(p, region, 1000, 1, out); fetch(); double st1 = mean(h, waves) / 1000;
        k_store_then_load<<<waves, 64>>>(p, warm, region, 1000, out); fetch(); double stld = mean(h, waves) / 1000;
        k_lds<<<waves, 64>>>(steps, out); fetch(); double lds = mean(h, waves) / steps;
        k_alu<<<waves, 64>>>(steps, out, 0); fetch(); double valu = mean(h, waves) / steps / 7;
        k_alu<<<waves, 64>>>(steps, out, 1); fetch(); double salu = mean(h, waves) / steps / 7;
        printf("waves %5d | dependent load: first touch %.0f, again %.0f, 64-line ring %.0f cyc | store+drain: 64 lanes %.0f, 1 lane %.0f | store then warm load %.0f | LDS dependent read %.0f | VALU dep op %.1f, SALU dep op %.1f\n",
               waves, cold, again, hot, st64, st1, stld, lds, valu, salu);
        CK(hipFree(p)); CK(hipFree(out)); CK(hipFree(warm));
    }
    return 0;
}
