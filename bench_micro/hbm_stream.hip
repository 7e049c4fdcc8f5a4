// Microbenchmark: what a streaming pass can reach on this MI355X (read-only sum, copy, in-place update).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_read(const float4* __restrict__ a, size_t n, float* out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = a[i]; s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) *out = s;
}
__global__ void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_inplace(float4* __restrict__ a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = a[i]; v.x *= 1.0001f; v.y *= 1.0001f; v.z *= 1.0001f; v.w *= 1.0001f; a[i] = v;
    }
}
template <typename F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipEventRecord(a); for (int i = 0; i < 5; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main() {
    const size_t bytes = 4370000000ull / 16 * 16, n = bytes / 16;
    float4 *a, *b; float* o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    for (int grid : {2048, 8192, 32768}) {
        float t1 = timeit([&] { k_read<<<grid, 256>>>(a, n, o); });
        float t2 = timeit([&] { k_copy<<<grid, 256>>>(a, b, n); });
        float t3 = timeit([&] { k_inplace<<<grid, 256>>>(a, n); });
        printf("grid %6d: read %.3f ms (%.2f TB/s) | copy %.3f ms (%.2f TB/s r+w) | in-place %.3f ms (%.2f TB/s r+w)\n", grid,
               t1, bytes / t1 / 1e9, t2, 2.0 * bytes / t2 / 1e9, t3, 2.0 * bytes / t3 / 1e9);
    }
    return 0;
}
