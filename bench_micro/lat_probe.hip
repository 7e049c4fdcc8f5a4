// latency probes on one workgroup (what bounds the l x l kernels of the PCA iteration): dependent f64 FMA chain, v_rsq_f64 chain,
// LDS write -> barrier -> read round trip with 6 waves; reports ns per step (s_memrealtime, 100 MHz) and shader clocks (s_memtime)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(384) void k(double* out, long long* t, int n, int mode) {
    __shared__ double s[512];
    double x = 1.0 + threadIdx.x * 1e-9, y = 0.5;
    s[threadIdx.x] = x;
    __syncthreads();
    const long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    if (mode == 0) for (int i = 0; i < n; ++i) x = __builtin_fma(x, y, 0.25);
    if (mode == 1) for (int i = 0; i < n; ++i) x = __builtin_amdgcn_rsq(x + 1.0);
    if (mode == 2) for (int i = 0; i < n; ++i) {
        s[(threadIdx.x + 1) & 383] = x;
        __syncthreads();
        x = s[threadIdx.x] + 1.0;
    }
    if (mode == 3) for (int i = 0; i < n; ++i) { __builtin_amdgcn_s_barrier(); }
    const long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) { t[0] = c1 - c0; t[1] = r1 - r0; }
}
int main() {
    double* d; long long* t;
    (void)hipMalloc(&d, 4096); (void)hipMalloc(&t, 64);
    const char* names[] = {"dependent f64 fma", "dependent v_rsq_f64 + add", "LDS write, barrier, read (6 waves)", "s_barrier only (6 waves)"};
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 4; ++mode) {
        const int n = 20000;
        k<<<1, 384>>>(d, t, n, mode);
        long long h[2];
        (void)hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
        printf("%-40s %8.1f ns/step  %8.1f memtime ticks/step\n", names[mode], h[1] * 10.0 / n, (double)h[0] / n);
    }
    return 0;
}
