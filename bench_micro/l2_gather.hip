// Microbenchmark: what a CU gets out of L2 (and beyond) with the Gram kernel's operand fetch — a buffer load whose first LEN lanes
// read LEN consecutive 8-byte entries from a pseudo-random 8-byte-aligned offset of a region (k_gram_stripes: one ~36-entry row
// suffix per owner record, range-as-predicate buffer loads, 8 of them in flight per wave, 16 waves per CU) — against the region size:
// 2 MB (every XCD's L2 holds it), 64 MB (Infinity Cache), 2 GB (HBM).  Reports GB/s of REQUESTED bytes and 64-byte line requests per
// clock and CU (the unit of profiles/r04_pmc_gram.md: the Gram kernel runs at 0.124).
// Build: hipcc --offload-arch=gfx950 -O3 -o l2_gather l2_gather.hip          Run: ./l2_gather
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

typedef unsigned int u2 __attribute__((ext_vector_type(2)));

// `len` lanes of a wave read `len` consecutive entries starting at entry `pos` (lanes >= len are out of the buffer's range: no fetch)
__device__ __forceinline__ u2 gather(const uint64_t* base, uint32_t pos, uint32_t len, int lane) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t*>(base + pos), (short)0, (int)(len * 8u), 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b64(rs, lane * 8, 0, 0);
}

constexpr int kUnroll = 8;
// RUNS runs of 64 / RUNS lanes each at its own random offset, `len` (<= 64 / RUNS) lanes of every run active (exec-masked global
// loads with per-lane addresses): what a load instruction costs when it serves 2 or 4 short suffixes at once
template <int RUNS>
__global__ __launch_bounds__(512, 2) void k_runs(const uint64_t* __restrict__ buf, uint32_t n_entries, uint32_t len, int iters, uint32_t* out) {
    const int lane = threadIdx.x & 63;
    constexpr int W = 64 / RUNS;
    const int run = lane / W, off = lane % W;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        u2 v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            uint32_t pos = hash32((wave * RUNS + run) * 2654435761u + (uint32_t)(it * kUnroll + u) * 40503u) & (n_entries - 1);
            if (pos + 64 > n_entries) pos = n_entries - 64;
            v[u] = u2{0u, 0u};
            if ((uint32_t)off < len) v[u] = *reinterpret_cast<const u2*>(buf + pos + off);
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) acc ^= v[u].x + v[u].y;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ __launch_bounds__(512, 2) void k(const uint64_t* __restrict__ buf, uint32_t n_entries /* power of two */, uint32_t len, int iters,
                                            uint32_t* out) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        u2 v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            uint32_t pos = hash32(wave * 2654435761u + (uint32_t)(it * kUnroll + u) * 40503u) & (n_entries - 1);
            pos = __builtin_amdgcn_readfirstlane(pos);
            if (pos + 64 > n_entries) pos = n_entries - 64;
            v[u] = gather(buf, pos, len, lane);
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) acc ^= v[u].x + v[u].y;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;
    const size_t max_bytes = (size_t)2 << 30;
    uint64_t* d_buf;
    uint32_t* d_out;
    CK(hipMalloc(&d_buf, max_bytes));
    CK(hipMalloc(&d_out, 64));
    CK(hipMemset(d_buf, 1, max_bytes));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    printf("%d CUs at %.2f GHz; 2 workgroups of 8 waves per CU, %d gathers in flight per wave\n", n_cus, clk * 1e-9, kUnroll);
    printf("%-10s %-6s %12s %14s %22s\n", "region", "lanes", "GB/s asked", "lines/clk/CU", "ms per 1e8 gathers");
    const size_t regions[] = {(size_t)2 << 20, (size_t)64 << 20, (size_t)2 << 30};
    const uint32_t lens[] = {64, 36, 16};
    for (size_t rg : regions)
        for (uint32_t len : lens) {
            const uint32_t n_entries = (uint32_t)(rg / 8);
            const int iters = 2000, blocks = n_cus * 2;
            k<<<blocks, 512>>>(d_buf, n_entries, len, 50, d_out);
            CK(hipEventRecord(a));
            k<<<blocks, 512>>>(d_buf, n_entries, len, iters, d_out);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            const double gathers = (double)blocks * 8 * iters * kUnroll;
            const double bytes = gathers * len * 8.0;
            // 64-byte lines a LEN-entry run at a random 8-byte offset touches: (len * 8 + 56) / 64 on average
            const double lines = gathers * (len * 8.0 + 56.0) / 64.0;
            printf("%-10s %-6u %12.0f %14.3f %22.2f\n", rg == ((size_t)2 << 20) ? "2 MB" : rg == ((size_t)64 << 20) ? "64 MB" : "2 GB", len,
                   bytes / ms * 1e-6, lines / (ms * 1e-3 * clk) / n_cus, ms * 1e8 / gathers);
        }
    printf("\nexec-masked global loads, RUNS runs per instruction (2 MB region: L2 hits)\n%-6s %-6s %12s %24s\n", "runs", "lanes", "GB/s asked", "ms per 1e8 instructions");
    auto runs = [&](auto kern, int R, uint32_t len) {
        const uint32_t n_entries = (uint32_t)(((size_t)2 << 20) / 8);
        const int iters = 2000, blocks = n_cus * 2;
        kern<<<blocks, 512>>>(d_buf, n_entries, len, 50, d_out);
        CK(hipEventRecord(a));
        kern<<<blocks, 512>>>(d_buf, n_entries, len, iters, d_out);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        const double instr = (double)blocks * 8 * iters * kUnroll;
        printf("%-6d %-6u %12.0f %24.2f\n", R, len, instr * R * len * 8.0 / ms * 1e-6, ms * 1e8 / instr);
    };
    runs(k_runs<1>, 1, 64); runs(k_runs<1>, 1, 36); runs(k_runs<1>, 1, 16);
    runs(k_runs<2>, 2, 32); runs(k_runs<2>, 2, 18);
    runs(k_runs<4>, 4, 16); runs(k_runs<4>, 4, 9);
    return 0;
}
