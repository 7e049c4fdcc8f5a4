// precision of the f64 hardware seeds v_rsq_f64 / v_rcp_f64 on gfx950 (what the Jacobi rotation's refinement starts from)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(double* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = ldexp(1.0 + (double)i / n * 3.0, (i % 41) - 20);
    double y = __builtin_amdgcn_rsq(x), r = __builtin_amdgcn_rcp(x);
    out[2 * i] = fabs(y * sqrt(x) - 1.0);
    out[2 * i + 1] = fabs(r * x - 1.0);
}
int main() {
    const int n = 1 << 20;
    double* d;
    hipMalloc(&d, 2 * n * sizeof(double));
    k<<<n / 256, 256>>>(d, n);
    double* h = new double[2 * n];
    hipMemcpy(h, d, 2 * n * sizeof(double), hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0;
    for (int i = 0; i < n; ++i) { m0 = fmax(m0, h[2 * i]); m1 = fmax(m1, h[2 * i + 1]); }
    printf("v_rsq_f64 max rel err %.3e (2^%.1f)   v_rcp_f64 max rel err %.3e (2^%.1f)\n", m0, log2(m0), m1, log2(m1));
    return 0;
}
