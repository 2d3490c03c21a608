// Scratch microbenchmark: dependent loads that hop between N far-apart arrays (TLB reach of a lone wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__global__ void __launch_bounds__(64) k_hop(const int *base, size_t arr_stride_ints, int narr, size_t wave_stride_ints, int steps, long long *out) {
    const int *q = base + (size_t)blockIdx.x * wave_stride_ints;
    int idx = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; s++) {
        const int a = s % narr;
        idx = __builtin_amdgcn_readfirstlane(q[(size_t)a * arr_stride_ints + (size_t)((idx + s * 1031) & 0x3fff)]);   // the data is all zero: the dependence is what matters
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = idx; }
}
int main() {
    const size_t total = (size_t)12 << 30;          // 12 GB of zeros
    int *p; long long *out;
    CK(hipMalloc(&p, total)); CK(hipMemset(p, 0, total)); CK(hipMalloc(&out, 5000 * 16));
    std::vector<long long> h(10000);
    for (int waves : {1, 625, 5000})
        for (int narr : {1, 2, 4, 8, 16}) {
            const size_t arr_stride = ((size_t)700 << 20) / 4;            // arrays 700 MB apart
            const size_t wave_stride = ((size_t)136 << 10) / 4;           // every wave its own 136 KB slice inside each array
            CK(hipDeviceSynchronize());
            k_hop<<<waves, 64>>>(p, arr_stride, narr, wave_stride, 1000, out);
            CK(hipDeviceSynchronize()); CK(hipMemcpy(h.data(), out, (size_t)waves * 16, hipMemcpyDeviceToHost));
            double s = 0; for (int i = 0; i < waves; i++) s += (double)h[2 * i];
            printf("waves %5d arrays %2d : %.0f cycles per dependent load\n", waves, narr, s / waves / 1000);
        }
    return 0;
}
