// latency probes on one workgroup (what the l x l kernels of the PCA iteration are made of, gfx950): dependent f64 / f32 chains
// unrolled 64 deep — a loop of ONE operation measures its own taken scalar branch (~13 ns), which an earlier form of this probe
// reported as "32 clocks per dependent FMA" —, LDS write -> barrier -> read and a bare s_barrier with 6 waves (these two keep
// the loop: subtract ~13 ns).  s_memrealtime, 100 MHz.  Measured: f64 FMA 5.6 clk, f32 FMA the same, f64 mul 4.5, rsq / rcp +
// add 24; LDS round trip 57-70 ns; barrier 18 ns.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(double* out, long long* t, int n) {
    double x = 1.0 + threadIdx.x * 1e-9, y = 0.5, a = 1.1, b = 1.2, c = 1.3;
    float xf = 1.0f + threadIdx.x * 1e-6f, yf = 0.5f;
    const long long r0 = wall_clock64();
    // 64 operations per loop iteration: the loop's own scalar branch (~13 ns!) is 1/64 of an iteration
    for (int i = 0; i < n / 64; ++i) {
#pragma unroll
        for (int u = 0; u < 64; ++u) {
            if (MODE == 0) x = __builtin_fma(x, y, 0.25);
            if (MODE == 1) xf = __builtin_fmaf(xf, yf, 0.25f);
            if (MODE == 2) { x = __builtin_fma(x, y, 0.25); a = __builtin_fma(a, y, 0.25); b = __builtin_fma(b, y, 0.25); c = __builtin_fma(c, y, 0.25); }
            if (MODE == 3) x = x * y;
            if (MODE == 4) x = __builtin_amdgcn_rsq(x + 1.5);
            if (MODE == 5) x = __builtin_amdgcn_rcp(x + 1.5);
        }
    }
    __shared__ double sm[1024];
    if (MODE == 6) { sm[threadIdx.x] = x; __syncthreads();
        for (int i = 0; i < n / 8; ++i) { sm[(threadIdx.x + 1) % blockDim.x] = x; __syncthreads(); x = sm[threadIdx.x] + 1.0; __syncthreads(); } }
    if (MODE == 7) for (int i = 0; i < n / 8; ++i) __builtin_amdgcn_s_barrier();
    const long long r1 = wall_clock64();
    out[threadIdx.x] = x + a + b + c + xf;
    if (threadIdx.x == 0) t[0] = r1 - r0;
}
template <int MODE>
void run(const char* name, int threads, double* d, long long* t) {
    const int n = 64 * 2000;
    k<MODE><<<1, threads>>>(d, t, n);
    long long h;
    (void)hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    const double steps = MODE >= 6 ? n / 8 : n;
    printf("%-46s %4d threads: %7.2f ns/step = %6.1f clk at 2.4 GHz\n", name, threads, h * 10.0 / steps, h * 10.0 / steps * 2.4);
}
int main() {
    double* d; long long* t;
    (void)hipMalloc(&d, 8192); (void)hipMalloc(&t, 64);
    for (int threads : {64, 256, 512, 1024}) {
        run<0>("dependent f64 fma", threads, d, t);
        run<1>("dependent f32 fma", threads, d, t);
        run<2>("4 independent f64 fma chains (per iteration)", threads, d, t);
        run<3>("dependent f64 mul", threads, d, t);
        run<4>("dependent v_rsq_f64 + add", threads, d, t);
        run<5>("dependent v_rcp_f64 + add", threads, d, t);
    }
    run<6>("LDS write, barrier, read, barrier (loop)", 384, d, t);
    run<7>("s_barrier only (loop)", 384, d, t);
    return 0;
}
