// Microbenchmark: rate of f64 LDS atomic adds as a function of the address pattern (MI355X, gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o lds_atomic_banks lds_atomic_banks.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(1024) void k(int iters, double* out) {
    extern __shared__ double acc[];
    for (int e = threadIdx.x; e < 16384 + 256; e += 1024) acc[e] = 0.0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, tid = threadIdx.x + blockIdx.x * 1024;
    for (int it = 0; it < iters; ++it) {
        const uint32_t h = hash32(tid * 2654435761u + it * 40503u);
        uint32_t idx;
        if (MODE == 0) idx = (it * 64 + lane + (threadIdx.x >> 6) * 1024) & 16383;       // 64 consecutive doubles
        else if (MODE == 1) idx = h & 16383;                                              // random
        else if (MODE == 2) idx = ((h & 16383) & ~31u) | (lane & 31);                     // random row, bank pair = lane & 31
        else if (MODE == 3) idx = ((h & 16383) & ~63u) | lane;                            // random 512-B row, lane-ordered
        else if (MODE == 4) idx = ((h & 16383) & ~15u) | (lane & 15);                     // 16-lane groups: random 128-B segment each
        else if (MODE == 8) idx = (h & 16383) & ~1u;                                      // pairs of lanes may share: ~random, even only
        else if (MODE == 9) idx = hash32((tid >> 1) * 2654435761u + it * 40503u) & 16383; // lanes 2m, 2m+1 hit the SAME address
        else if (MODE == 10) idx = hash32((tid >> 2) * 2654435761u + it * 40503u) & 16383; // 4 lanes per address
        else idx = (h & 16383);
        if (MODE == 5) {                                                                  // random, u64 integer add
            atomicAdd(reinterpret_cast<unsigned long long*>(acc) + idx, (unsigned long long)h);
        } else if (MODE == 6) {                                                           // random, 32-bit integer add
            atomicAdd(reinterpret_cast<unsigned int*>(acc) + (h & 32767), h);
        } else if (MODE == 15) {                                                          // random, f32 add
            atomicAdd(reinterpret_cast<float*>(acc) + (h & 32767), 1.0f + lane);
        } else if (MODE == 16) {                                                          // random, f32 add, 32 lanes active
            if (lane & 1) atomicAdd(reinterpret_cast<float*>(acc) + (h & 32767), 1.0f + lane);
        } else if (MODE == 7) {                                                           // random, non-atomic f64 store
            acc[idx] = (double)h;
        } else if (MODE == 11) {                                                          // random, 32 of 64 lanes active
            if (lane & 1) atomicAdd(&acc[idx], 1.0 + lane);
        } else if (MODE == 12) {                                                          // random, 16 of 64 lanes active
            if ((lane & 3) == 0) atomicAdd(&acc[idx], 1.0 + lane);
        } else if (MODE == 13) {                                                          // random, 8 of 64 lanes active
            if ((lane & 7) == 0) atomicAdd(&acc[idx], 1.0 + lane);
        } else if (MODE == 14) {                                                          // 2 dependent-free b64 reads + random atomic
            const double x = acc[16384 + ((h >> 14) & 255)], y = acc[16384 + ((h >> 22) & 255)];
            atomicAdd(&acc[idx], x + y + 1.0);
        } else {
            atomicAdd(&acc[idx], 1.0 + lane);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc[5] + acc[77];
}

template <int MODE>
void run(const char* name, double* d_out) {
    const int iters = 20000, blocks = 256;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 1024, 131072>>>(100, d_out);
    hipEventRecord(a);
    k<MODE><<<blocks, 1024, 131072>>>(iters, d_out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double lane_ops = (double)iters * 1024 * blocks;
    printf("%-58s %8.3f ms  %6.2f lane-ops/clk/CU (at 2.4 GHz, 256 CUs)\n", name, ms, lane_ops / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
    double* d_out; hipMalloc(&d_out, 256 * 8);
    run<0>("f64 add, 64 consecutive doubles per wave", d_out);
    run<1>("f64 add, random over 16384 doubles", d_out);
    run<2>("f64 add, random row, bank pair = lane & 31", d_out);
    run<3>("f64 add, random 512-B row, lanes in order", d_out);
    run<4>("f64 add, 16-lane groups on random 128-B segments", d_out);
    run<5>("u64 add, random", d_out);
    run<6>("u32 add, random over 32768 words", d_out);
    run<7>("f64 plain store, random", d_out);
    run<15>("f32 add, random over 32768 words", d_out);
    run<16>("f32 add, random, 32 of 64 lanes active (x2)", d_out);
    run<9>("f64 add, random, lane pairs on the SAME address", d_out);
    run<10>("f64 add, random, 4 lanes per address", d_out);
    run<11>("f64 add, random, 32 of 64 lanes active (rate per ACTIVE lane x2)", d_out);
    run<12>("f64 add, random, 16 of 64 lanes active (x4)", d_out);
    run<13>("f64 add, random, 8 of 64 lanes active (x8)", d_out);
    run<14>("2 x ds_read_b64 (random in 2 KiB) + random f64 add", d_out);
    return 0;
}
