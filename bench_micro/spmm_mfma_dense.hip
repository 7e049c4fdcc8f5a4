// Microbenchmark (round 4): the forward SpMM  Y = A P  (A: N x k HVG-compacted, ~72 of k = 2000 entries per
// cell; P: k x 64 f32 panel) as a DENSIFIED-TILE MFMA kernel — a wave densifies a 16-cell x 256-gene tile of A into LDS and
// multiplies it with the 256 x 64 panel block on v_mfma_f32_16x16x4_f32 — next to a plain gather kernel on the same data.
// The point measured: at 3.6 % density the dense formulation spends 2 N k 64 = 3.3e11 flops on 9.4e9 useful ones (28x), and
// the f32 matrix cores (256 flop / clk / CU) need > 0.9 ms for that at c3 whatever the memory system does.
// Build: hipcc --offload-arch=gfx950 -O3 -o spmm_mfma_dense spmm_mfma_dense.hip        Run: ./spmm_mfma_dense [n_cells]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Entry { int32_t j; float v; };
constexpr int K = 2000, L = 64, KB = 256;                 // genes, panel columns, genes per staged panel block
constexpr int NB = (K + KB - 1) / KB;
typedef float f4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// rows of m = 48 .. 96 entries, stratified sorted columns (like the bench generator)
__global__ void k_fill(const int64_t* ptr, uint64_t n_rows, Entry* e) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const int64_t lo = ptr[r];
    const int m = (int)(ptr[r + 1] - lo);
    for (int s = 0; s < m; ++s) {
        const int b0 = (int)((int64_t)s * K / m), b1 = (int)((int64_t)(s + 1) * K / m);
        const uint64_t h = mix(r * 1315423911ull + s);
        e[lo + s].j = b0 + (int)(h % (uint64_t)(b1 - b0));
        e[lo + s].v = 0.25f + (float)((h >> 40) & 1023) * (1.0f / 256.0f);
    }
}

// ---- reference: one wave per cell, lane = panel column, the panel straight from global memory (L2) ----
__global__ __launch_bounds__(256) void k_gather(const int64_t* __restrict__ ptr, const Entry* __restrict__ e, uint64_t n_rows,
                                                const float* __restrict__ P, float* __restrict__ Y) {
    const uint64_t w = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / 64;
    const int lane = threadIdx.x & 63;
    if (w >= n_rows) return;
    float acc = 0.f;
    for (int64_t p = ptr[w]; p < ptr[w + 1]; ++p) acc += e[p].v * P[(size_t)e[p].j * L + lane];
    Y[w * L + lane] = acc;
}

// ---- the densified-tile MFMA variant -----------------------------------------------------------------
// Workgroup = 4 waves = 64 cells; per 256-gene block: the 256 x 64 panel block staged in LDS (64 KiB), each wave's 16 x 256
// tile of A zeroed and filled from the rows' segments (sorted rows: a cursor per row), then 64 K-steps of
// (1 A read + 4 B reads + 4 MFMA 16x16x4): accumulators 4 x f4 per lane = the wave's 16 x 64 output tile.
__global__ __launch_bounds__(256) void k_mfma(const int64_t* __restrict__ ptr, const Entry* __restrict__ e, uint64_t n_rows,
                                              const float* __restrict__ P, float* __restrict__ Y, int densify) {
    extern __shared__ float lds[];
    float* sP = lds;                                   // KB x 64
    float* sA = lds + KB * L + (threadIdx.x / 64) * 16 * (KB + 4);      // 16 x (KB + 4) per wave (padded rows: no bank conflicts)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t row0 = ((uint64_t)blockIdx.x * 4 + wave) * 16;
    // lane r < 16: cursor / end of row row0 + r
    int64_t cur = 0, end = 0;
    if (lane < 16 && row0 + lane < n_rows) { cur = ptr[row0 + lane]; end = ptr[row0 + lane + 1]; }
    f4 acc[4] = {f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}};
    const int li = lane & 15, lk = lane >> 4;
    for (int gb = 0; gb < NB; ++gb) {
        const int g0 = gb * KB, g1 = g0 + KB < K ? g0 + KB : K;
        __syncthreads();
        for (int x = threadIdx.x; x < KB * L / 4; x += 256) {
            const int g = g0 + x / (L / 4);
            reinterpret_cast<float4*>(sP)[x] = g < K ? reinterpret_cast<const float4*>(P)[(size_t)g * (L / 4) + x % (L / 4)]
                                                     : float4{0, 0, 0, 0};
        }
        for (int x = lane; x < 16 * (KB + 4); x += 64) sA[x] = 0.f;
        __syncthreads();
        if (densify) {
            for (int r = 0; r < 16; ++r) {
                const int64_t c = __shfl(cur, r, 64), en = __shfl(end, r, 64);
                const int64_t p = c + lane;
                Entry x{K, 0.f};
                if (p < en) x = e[p];
                const bool in = x.j < g1;                                  // (sorted: the segment is a prefix of what is left)
                if (in) sA[r * (KB + 4) + (x.j - g0)] = x.v;
                const int n_in = __popcll(__ballot(in));
                if (lane == r) cur += n_in;
            }
        }
        __syncthreads();
        for (int kk = 0; kk < KB; kk += 4) {
            const float a = sA[li * (KB + 4) + kk + lk];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float b = sP[(kk + lk) * L + 16 * t + li];
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
            }
        }
    }
    // C/D layout of the f32 16x16x4 MFMA: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const uint64_t r = row0 + 4 * lk + v;
            if (r < n_rows) Y[r * L + 16 * t + li] = acc[t][v];
        }
}

int main(int argc, char** argv) {
    const uint64_t N = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1300000ull;
    std::vector<int64_t> hp(N + 1);
    hp[0] = 0;
    for (uint64_t r = 0; r < N; ++r) hp[r + 1] = hp[r] + 48 + (int64_t)(mix(r) % 49);
    const int64_t nnz = hp[N];
    int64_t* d_ptr; Entry* d_e; float *d_P, *d_Y0, *d_Y1;
    CK(hipMalloc(&d_ptr, (N + 1) * 8));
    CK(hipMalloc(&d_e, (nnz + 64) * sizeof(Entry)));
    CK(hipMalloc(&d_P, (size_t)K * L * 4));
    CK(hipMalloc(&d_Y0, N * L * 4));
    CK(hipMalloc(&d_Y1, N * L * 4));
    CK(hipMemcpy(d_ptr, hp.data(), (N + 1) * 8, hipMemcpyHostToDevice));
    std::vector<float> hP((size_t)K * L);
    for (size_t i = 0; i < hP.size(); ++i) hP[i] = (float)((int)(mix(i) % 2001) - 1000) * 1e-3f;
    CK(hipMemcpy(d_P, hP.data(), hP.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_e, 0, (nnz + 64) * sizeof(Entry)));
    k_fill<<<(unsigned)((N + 255) / 256), 256>>>(d_ptr, N, d_e);
    CK(hipDeviceSynchronize());
    const size_t lds = (size_t)(KB * L + 4 * 16 * (KB + 4)) * 4;
    CK(hipFuncSetAttribute((const void*)k_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto time = [&](auto&& f, int reps) {
        f();
        CK(hipEventRecord(a));
        for (int i = 0; i < reps; ++i) f();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        return ms / reps;
    };
    const float t_g = time([&] { k_gather<<<(unsigned)((N * 64 + 255) / 256), 256>>>(d_ptr, d_e, N, d_P, d_Y0); }, 5);
    const float t_m = time([&] { k_mfma<<<(unsigned)((N + 63) / 64), 256, lds>>>(d_ptr, d_e, N, d_P, d_Y1, 1); }, 5);
    CK(hipGetLastError());
    std::vector<float> y0(1 << 20), y1(1 << 20);
    CK(hipMemcpy(y0.data(), d_Y0, y0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(y1.data(), d_Y1, y1.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    for (size_t i = 0; i < y0.size(); ++i) { worst = fmax(worst, fabs((double)y0[i] - y1[i])); scale = fmax(scale, fabs((double)y0[i])); }
    const float t_n = time([&] { k_mfma<<<(unsigned)((N + 63) / 64), 256, lds>>>(d_ptr, d_e, N, d_P, d_Y1, 0); }, 5);
    const double dense_flops = 2.0 * (double)N * (NB * KB) * L, useful = 2.0 * (double)nnz * L;
    printf("cells %llu, kept entries %lld (%.1f per cell), k = %d, %d panel columns\n", (unsigned long long)N, (long long)nnz,
           (double)nnz / N, K, L);
    printf("plain gather kernel (wave per cell, panel from L2)      : %8.3f ms\n", t_g);
    printf("densified 16 x 256 tiles on v_mfma_f32_16x16x4_f32      : %8.3f ms  (%.1f TFLOP/s dense, %.2f useful; dense / useful flops = %.1fx)\n",
           t_m, dense_flops / t_m * 1e-9, useful / t_m * 1e-9, dense_flops / useful);
    printf("... the same without the densification (MFMA + LDS only): %8.3f ms\n", t_n);
    printf("max |gather - mfma| over the first 16384 cells: %.3e (largest |y| %.3e)\n", worst, scale);
    return 0;
}
