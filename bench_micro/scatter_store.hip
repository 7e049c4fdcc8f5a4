// Microbenchmark: the request rate of scattered small stores — k_bucket writes 1.1e8 12-byte owner records per launch, each lane its
// own record into one of ~250 runs of a 470 KB region per 512-cell block (L2-resident: the lines are combined there), and runs at 0.2
// store requests per clock and CU (profiles/r04_knockouts.md).  Here: every lane of 2 x 1024 threads per CU stores BYTES bytes at a
// pseudo-random slot of its workgroup's own region, 8 stores in flight.
// Build: hipcc --offload-arch=gfx950 -O3 -o scatter_store scatter_store.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
struct R12 { uint32_t a, b, c; };
template <int BYTES, int MODE /* 0: random slot per lane; 1: lanes of a wave write CONSECUTIVE slots from a random start */>
__global__ __launch_bounds__(1024) void k(char* buf, uint32_t slots_per_wg, int iters) {
    char* base = buf + (size_t)blockIdx.x * slots_per_wg * BYTES;
    const uint32_t t = blockIdx.x * 1024u + threadIdx.x, lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            uint32_t s;
            if (MODE == 0) s = hash32(t * 2654435761u + (uint32_t)(it * 8 + u) * 40503u) % slots_per_wg;
            else s = (hash32((t >> 6) * 2654435761u + (uint32_t)(it * 8 + u) * 40503u) % (slots_per_wg - 64)) + lane;
            if (BYTES == 12) *reinterpret_cast<R12*>(base + (size_t)s * 12) = R12{s, t, (uint32_t)it};
            else if (BYTES == 4) *reinterpret_cast<uint32_t*>(base + (size_t)s * 4) = s;
            else *reinterpret_cast<uint4*>(base + (size_t)s * 16) = uint4{s, t, (uint32_t)it, 0u};
        }
    }
}
int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int n_cus = prop.multiProcessorCount; const double clk = prop.clockRate * 1e3;
    const uint32_t slots = 40000;                       // ~470 KB of 12-byte records per workgroup
    char* d; CK(hipMalloc(&d, (size_t)n_cus * 2 * slots * 16));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](auto kern, const char* name, int bytes) {
        const int iters = 400, blocks = n_cus * 2;
        kern<<<blocks, 1024>>>(d, slots, 10);
        CK(hipEventRecord(a)); kern<<<blocks, 1024>>>(d, slots, iters); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double stores = (double)blocks * 1024 * iters * 8;
        printf("%-44s %8.3f ms  %6.3f lane-stores per clk and CU  %7.1f GB/s  (1.1e8 stores: %.2f ms)\n", name, ms, stores / (ms * 1e-3 * clk) / n_cus,
               stores * bytes / ms * 1e-6, ms * 1.1e8 / stores);
    };
    run(k<12, 0>, "12-byte records, random slot per lane", 12);
    run(k<16, 0>, "16-byte records, random slot per lane", 16);
    run(k<4, 0>, "4-byte words, random slot per lane", 4);
    run(k<12, 1>, "12-byte records, 64 consecutive slots per wave", 12);
    CK(hipGetLastError());
    return 0;
}
