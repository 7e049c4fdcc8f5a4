// Microbenchmark: what rocprofv3's FETCH_SIZE reports for the access patterns of this repo's kernels, on KNOWN byte counts (VERDICT r5
// item 4: "calibrate FETCH_SIZE on the kernel's own access pattern before the next decision").  Each pattern is its own kernel
// symbol, launched once over a 4 GB region (16x the Infinity Cache: every first touch of a line goes to HBM), each address touched once:
//   k_stream      16 bytes per lane, consecutive lanes consecutive addresses (the guide's calibration case: FETCH_SIZE = 1/2 of the bytes)
//   k_gather<36>  one run of 36 consecutive 8-byte entries per wave-instruction at a random 8-byte-aligned offset, 8 bytes per lane
//                 (the Gram kernel of rounds 2-4: one row suffix per load)
//   k_gather<64>  the same with full 64-lane runs
//   k_pairs<18>   TWO runs per instruction (lanes 0-31 / 32-63), 16 bytes per lane, 18 lanes of each half active = 288 bytes per run
//                 (the Gram kernel since round 4: two owner records per global_load_dwordx4)
//   k_quads       four 16-byte pieces at a 64-byte stride per quad (the forward SpMM's chunk load: 64 bytes per lane in four loads)
// Runs are placed at (random slot) * 1024 + (random 8-byte offset below 512), slots drawn without replacement (a bijective hash):
// no two runs share a line, so the bytes any cache level must bring in are known: per run, the 64-byte sectors / 128-byte lines it
// touches.  Usage (GPU box):  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o c -- ./fetch_calib ; summarize with
// profiles/summarize_pmc.py.  The program prints the expected bytes per kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr uint32_t kSlotBits = 22;                        // 4 Mi slots of 1 KiB = 4 GiB
__device__ __host__ inline uint32_t perm22(uint32_t x) {  // a bijection of [0, 2^22): odd multiplier + xorshift, twice
    const uint32_t M = (1u << kSlotBits) - 1;
    x = (x * 0x2c9277b5u) & M; x ^= x >> 11; x = (x * 0x1b873593u) & M; x ^= x >> 13; x = (x * 0x9e3779b1u) & M;
    return x & M;
}
__device__ __host__ inline uint32_t off8(uint32_t x) {    // 8-byte-aligned offset below 512
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15;
    return (x & 63u) * 8u;
}

__global__ __launch_bounds__(256) void k_stream(const uint4* __restrict__ buf, size_t n16, uint32_t* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i < n16; i += stride) {
        const uint4 v = buf[i];
        acc ^= v.x + v.y + v.z + v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int LEN>
__global__ __launch_bounds__(256) void k_gather(const char* __restrict__ buf, uint32_t n_runs, uint32_t* out) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    uint32_t acc = 0;
    for (uint32_t r = wave; r < n_runs; r += n_waves) {
        const char* p = buf + (size_t)perm22(r) * 1024 + off8(r);
        if (lane < LEN) {
            const uint2 v = *reinterpret_cast<const uint2*>(p + lane * 8);
            acc ^= v.x + v.y;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int LANES>
__global__ __launch_bounds__(256) void k_pairs(const char* __restrict__ buf, uint32_t n_runs, uint32_t* out) {
    const int lane = threadIdx.x & 63, half = lane >> 5, l = lane & 31;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    uint32_t acc = 0;
    for (uint32_t r = 2 * wave; r + 1 < n_runs; r += 2 * n_waves) {
        const uint32_t rr = r + half;
        const char* p = buf + (size_t)perm22(rr) * 1024 + off8(rr);
        if (l < LANES) {
            uint4 v;
            __builtin_memcpy(&v, p + l * 16, 16);        // 8-byte aligned 16-byte load
            acc ^= v.x + v.y + v.z + v.w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_quads(const char* __restrict__ buf, uint32_t n_runs /* 256-byte runs, one per quad */, uint32_t* out) {
    const int lane = threadIdx.x & 63, quad = lane >> 2, w4 = lane & 3;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    uint32_t acc = 0;
    for (uint32_t r = 16 * wave; r + 15 < n_runs; r += 16 * n_waves) {
        const uint32_t rr = r + quad;
        const char* p = buf + (size_t)perm22(rr) * 1024 + off8(rr) + 64 * w4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint4 v;
            __builtin_memcpy(&v, p + 16 * i, 16);
            acc ^= v.x + v.y + v.z + v.w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

static void expect(const char* name, uint32_t n_runs, uint32_t run_bytes) {
    // sectors / lines a run of run_bytes at offset off8(r) of a 1 KiB-aligned slot touches
    double s64 = 0, l128 = 0;
    for (uint32_t r = 0; r < n_runs; ++r) {
        const uint32_t o = off8(r);
        s64 += (double)((o + run_bytes - 1) / 64 - o / 64 + 1);
        l128 += (double)((o + run_bytes - 1) / 128 - o / 128 + 1);
    }
    printf("%-14s runs %9u  asked %8.1f MB  64-B sectors %8.1f MB  128-B lines %8.1f MB\n", name, n_runs, n_runs * (double)run_bytes * 1e-6,
           s64 * 64e-6, l128 * 128e-6);
}

int main() {
    const size_t bytes = (size_t)1 << 32;
    char* d_buf;
    uint32_t* d_out;
    CK(hipMalloc(&d_buf, bytes + 4096));
    CK(hipMalloc(&d_out, 64));
    CK(hipMemset(d_buf, 1, bytes + 4096));
    CK(hipDeviceSynchronize());
    const uint32_t n_runs = 1u << 21;                     // 2 Mi runs (half of the slots): 0.6-1 GB per kernel
    const int blocks = 256 * 8;
    k_stream<<<blocks, 256>>>(reinterpret_cast<const uint4*>(d_buf), ((size_t)1 << 30) / 16, d_out);     // 1 GiB
    k_gather<36><<<blocks, 256>>>(d_buf, n_runs, d_out);
    k_gather<64><<<blocks, 256>>>(d_buf, n_runs, d_out);
    k_pairs<18><<<blocks, 256>>>(d_buf, n_runs, d_out);
    k_quads<<<blocks, 256>>>(d_buf, n_runs, d_out);
    CK(hipDeviceSynchronize());
    printf("%-14s asked %8.1f MB (= sectors = lines)\n", "k_stream", 1073.7);
    expect("k_gather<36>", n_runs, 288);
    expect("k_gather<64>", n_runs, 512);
    expect("k_pairs<18>", n_runs, 288);
    expect("k_quads", n_runs, 256);
    return 0;
}
