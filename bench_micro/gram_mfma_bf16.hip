// Microbenchmark (round 5): the DENSE-GENE block of the sparse Gram matrix on the bf16 matrix cores.
//
// Under skewed gene densities the selected genes are the dense ones and the ~256 densest of them (present in ~20 % of the cells)
// carry ~45 % of the Gram kernel's scalar products: 7 % in their own 256 x 256 block and 38 % in the 256 x 1744 cross block with
// the other selected genes (~6 % dense).  Those blocks as DENSE products on v_mfma_f32_16x16x32_bf16, every f32 value split into
// bf16 hi + lo (hi hi + hi lo + lo hi: 16 mantissa bits, f32 accumulation):
//
//   workgroup = (chunk of cells, block of 256 "other" genes); per step of 32 cells it densifies the cells' entries of the 256
//   dense genes and of its 256-gene block into LDS ([gene][cell] bf16, hi and lo: the A / B fragment of the 16x16x32 MFMA is 8
//   consecutive cells of one gene, one 16-byte LDS read) and adds D^T S into a 256 x 256 f32 tile held in registers (8 waves x
//   (64 x 128): 128 accumulator registers per lane), 3 MFMAs per 16 x 16 tile and step.
//
// What it measures: the time of that dense formulation for N cells x 7 blocks (= the 256 x 1792 cross block) with the
// densification and without it (the MFMA + LDS-read floor), and the error of the split against f64 on a small case.
// Build: hipcc --offload-arch=gfx950 -O3 -o gram_mfma_bf16 gram_mfma_bf16.hip        Run: ./gram_mfma_bf16 [n_cells]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int GB = 256;                  // genes per block
constexpr int KC = 32;                   // cells per step (the MFMA's contraction length)
constexpr int LDG = KC + 8;              // bf16 per gene row in LDS (80 bytes: 16 lanes x 80 B spread over the banks)
constexpr int kWaves = 8;
typedef short bf8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

struct Entry { uint16_t j; uint16_t pad; float v; };

__host__ __device__ inline uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__host__ __device__ inline uint16_t bf16_rne(float f) {
    uint32_t u;
#ifdef __HIP_DEVICE_COMPILE__
    u = __float_as_uint(f);
#else
    memcpy(&u, &f, 4);
#endif
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__host__ __device__ inline float bf16_to_f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(u);
#else
    float f; memcpy(&f, &u, 4); return f;
#endif
}

// block b (0 = the dense genes, 1 .. = the other blocks): cell r has each gene with probability dens (hash), value in [0.7, 9)
__global__ void k_count(uint64_t n, int b, float dens, int* cnt) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int c = 0;
    for (int g = 0; g < GB; ++g) c += (mix((r * 8191 + b) * 977 + g) & 0xffffff) < (uint32_t)(dens * 16777216.f);
    cnt[r] = c;
}
__global__ void k_fill(uint64_t n, int b, float dens, const int64_t* ptr, Entry* e) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int64_t p = ptr[r];
    for (int g = 0; g < GB; ++g) {
        const uint64_t h = mix((r * 8191 + b) * 977 + g);
        if ((h & 0xffffff) < (uint32_t)(dens * 16777216.f)) {
            e[p].j = (uint16_t)g;
            e[p].pad = 0;
            e[p].v = 0.7f + (float)((h >> 24) & 0xffff) * (8.3f / 65536.f);
            ++p;
        }
    }
}

// G_part[chunk][b][256][256] += D^T S_b over the chunk's cells.  `densify` = 0: the LDS tiles are filled once (timing floor).
__global__ __launch_bounds__(kWaves * 64) void k_gram_dense(const int64_t* __restrict__ ptrD, const Entry* __restrict__ eD,
                                                             const int64_t* __restrict__ ptrS, const Entry* __restrict__ eS,
                                                             uint64_t n_cells, uint64_t cells_per_chunk, uint64_t block_stride_ptr,
                                                             uint64_t block_stride_e, float* __restrict__ Gp, int densify) {
    extern __shared__ uint16_t lds[];          // [4][GB][LDG]: D hi, D lo, S hi, S lo
    uint16_t* const Dh = lds, * const Dl = lds + GB * LDG, * const Sh = lds + 2 * GB * LDG, * const Sl = lds + 3 * GB * LDG;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.y;
    const int64_t* pS = ptrS + (uint64_t)b * block_stride_ptr;
    const Entry* enS = eS + (uint64_t)b * block_stride_e;
    const uint64_t c0 = (uint64_t)blockIdx.x * cells_per_chunk, c1 = c0 + cells_per_chunk < n_cells ? c0 + cells_per_chunk : n_cells;
    const int wr = wv >> 1, wc = wv & 1;        // 64 rows (dense genes) x 128 columns of the 256 x 256 tile
    f4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    const int fi = lane & 15, fk = (lane >> 4) * 8;
    for (uint64_t s0 = c0; s0 < c1; s0 += KC) {
        if (densify || s0 == c0) {
            __syncthreads();
            for (int e = tid; e < 4 * GB * LDG / 8; e += kWaves * 64) reinterpret_cast<uint4*>(lds)[e] = uint4{0u, 0u, 0u, 0u};
            __syncthreads();
            // 16 threads per cell: the cell's entries of the dense block, then of this block
            const int cell = tid >> 4, sub = tid & 15;
            const uint64_t r = s0 + cell;
            if (r < c1) {
                for (int64_t p = ptrD[r] + sub; p < ptrD[r + 1]; p += 16) {
                    const Entry x = eD[p];
                    const uint16_t h = bf16_rne(x.v);
                    Dh[x.j * LDG + cell] = h;
                    Dl[x.j * LDG + cell] = bf16_rne(x.v - bf16_to_f(h));
                }
                for (int64_t p = pS[r] + sub; p < pS[r + 1]; p += 16) {
                    const Entry x = enS[p];
                    const uint16_t h = bf16_rne(x.v);
                    Sh[x.j * LDG + cell] = h;
                    Sl[x.j * LDG + cell] = bf16_rne(x.v - bf16_to_f(h));
                }
            }
            __syncthreads();
        }
        // A fragment: lane holds A[i = lane & 15][k = 8 (lane >> 4) .. + 7] = 8 consecutive cells of dense gene 64 wr + 16 ti + i
        bf8 ah[4], al[4];
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            const int g = 64 * wr + 16 * ti + fi;
            ah[ti] = *reinterpret_cast<const bf8*>(Dh + g * LDG + fk);
            al[ti] = *reinterpret_cast<const bf8*>(Dl + g * LDG + fk);
        }
#pragma unroll
        for (int tj = 0; tj < 8; ++tj) {
            const int g = 128 * wc + 16 * tj + fi;
            const bf8 bh = *reinterpret_cast<const bf8*>(Sh + g * LDG + fk);
            const bf8 bl = *reinterpret_cast<const bf8*>(Sl + g * LDG + fk);
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) {
                acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ti], bh, acc[ti][tj], 0, 0, 0);
                acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ti], bl, acc[ti][tj], 0, 0, 0);
                acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[ti], bh, acc[ti][tj], 0, 0, 0);
            }
        }
    }
    // C/D: col = lane & 15, row = (lane >> 4) * 4 + reg
    float* out = Gp + ((uint64_t)blockIdx.x * gridDim.y + b) * GB * GB;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 8; ++tj)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                out[(64 * wr + 16 * ti + (lane >> 4) * 4 + v) * GB + 128 * wc + 16 * tj + (lane & 15)] = acc[ti][tj][v];
}

static void scan_host(const std::vector<int>& c, std::vector<int64_t>& p) {
    p.resize(c.size() + 1);
    p[0] = 0;
    for (size_t i = 0; i < c.size(); ++i) p[i + 1] = p[i] + c[i];
}

int main(int argc, char** argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1300000ull;
    const int n_blocks = 7;                       // 7 x 256 = 1792 "other" genes
    const float dens_d = 0.20f, dens_s = 0.06f;
    // matrices: block 0 = dense genes; blocks 1 .. 7 = the others (each its own ptr / entry array at a fixed stride)
    int* d_cnt;
    CK(hipMalloc(&d_cnt, n * sizeof(int)));
    std::vector<int> cnt(n);
    std::vector<int64_t> ptr;
    auto make = [&](int b, float dens, int64_t** d_ptr, Entry** d_e, uint64_t cap_e) -> uint64_t {
        hipLaunchKernelGGL(k_count, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, n, b, dens, d_cnt);
        CK(hipMemcpy(cnt.data(), d_cnt, n * sizeof(int), hipMemcpyDeviceToHost));
        scan_host(cnt, ptr);
        if ((uint64_t)ptr[n] > cap_e) { printf("entry capacity\n"); exit(1); }
        CK(hipMemcpy(*d_ptr, ptr.data(), (n + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, n, b, dens, *d_ptr, *d_e);
        CK(hipDeviceSynchronize());
        return (uint64_t)ptr[n];
    };
    const uint64_t capD = (uint64_t)(n * GB * dens_d * 1.1) + 1024, capS = (uint64_t)(n * GB * dens_s * 1.15) + 1024;
    int64_t *pD, *pS;
    Entry *eD, *eS;
    CK(hipMalloc(&pD, (n + 1) * 8));
    CK(hipMalloc(&eD, capD * sizeof(Entry)));
    CK(hipMalloc(&pS, (uint64_t)n_blocks * (n + 1) * 8));
    CK(hipMalloc(&eS, (uint64_t)n_blocks * capS * sizeof(Entry)));
    const uint64_t nD = make(0, dens_d, &pD, &eD, capD);
    uint64_t nS = 0;
    std::vector<int64_t> ptrS0;
    for (int b = 0; b < n_blocks; ++b) {
        int64_t* pp = pS + (uint64_t)b * (n + 1);
        Entry* ee = eS + (uint64_t)b * capS;
        nS += make(b + 1, dens_s, &pp, &ee, capS);
        if (b == 0) ptrS0 = ptr;
    }
    printf("cells %llu: dense block %.1f entries / cell, other genes %.1f / cell (7 x 256)\n", (unsigned long long)n, (double)nD / n,
           (double)nS / n);
    int n_cu = 256;
    { hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); n_cu = pr.multiProcessorCount; }
    const int chunks = (n_cu + n_blocks - 1) / n_blocks;             // ~ one workgroup per CU
    const uint64_t cpc = ((n + chunks - 1) / chunks + KC - 1) / KC * KC;
    float* Gp;
    CK(hipMalloc(&Gp, (uint64_t)chunks * n_blocks * GB * GB * sizeof(float)));
    const size_t ldsb = 4 * GB * LDG * sizeof(uint16_t);
    CK(hipFuncSetAttribute((const void*)k_gram_dense, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0));
    CK(hipEventCreate(&t1));
    for (int densify = 1; densify >= 0; --densify) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(t0));
            hipLaunchKernelGGL(k_gram_dense, dim3(chunks, n_blocks), dim3(kWaves * 64), ldsb, 0, pD, eD, pS, eS, n, cpc, n + 1, capS, Gp,
                               densify);
            CK(hipEventRecord(t1));
            CK(hipEventSynchronize(t1));
            float ms;
            CK(hipEventElapsedTime(&ms, t0, t1));
            if (rep && ms < best) best = ms;
        }
        const double flop = 2.0 * 3.0 * (double)n * GB * GB * n_blocks;
        printf("%s: %.3f ms  (%.0f TFLOP/s of bf16 MFMA on the 3-product dense formulation; useful products %.2e)\n",
               densify ? "densify + MFMA" : "MFMA + LDS reads only (tiles filled once)", best, flop / best / 1e9,
               (double)nD / n * (double)nS / n * n);
    }
    // accuracy of the split on the first chunk's block 0 against f64 (host), the kernel run with densification
    hipLaunchKernelGGL(k_gram_dense, dim3(chunks, n_blocks), dim3(kWaves * 64), ldsb, 0, pD, eD, pS, eS, n, cpc, n + 1, capS, Gp, 1);
    CK(hipDeviceSynchronize());
    {
        const uint64_t nc = cpc < n ? cpc : n;
        std::vector<int64_t> hpD(nc + 1), hpS(nc + 1);
        CK(hipMemcpy(hpD.data(), pD, (nc + 1) * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hpS.data(), pS, (nc + 1) * 8, hipMemcpyDeviceToHost));
        std::vector<Entry> hD(hpD[nc]), hS(hpS[nc]);
        CK(hipMemcpy(hD.data(), eD, hD.size() * sizeof(Entry), hipMemcpyDeviceToHost));
        CK(hipMemcpy(hS.data(), eS, hS.size() * sizeof(Entry), hipMemcpyDeviceToHost));
        std::vector<double> ref((size_t)GB * GB, 0.0);
        for (uint64_t r = 0; r < nc; ++r)
            for (int64_t a = hpD[r]; a < hpD[r + 1]; ++a)
                for (int64_t c = hpS[r]; c < hpS[r + 1]; ++c) ref[(size_t)hD[a].j * GB + hS[c].j] += (double)hD[a].v * (double)hS[c].v;
        std::vector<float> got((size_t)GB * GB);
        CK(hipMemcpy(got.data(), Gp, got.size() * sizeof(float), hipMemcpyDeviceToHost));
        double worst = 0.0, sum2 = 0.0, ref2 = 0.0;
        for (size_t i = 0; i < ref.size(); ++i) {
            if (ref[i] > 0) worst = fmax(worst, fabs(got[i] - ref[i]) / ref[i]);
            sum2 += (got[i] - ref[i]) * (got[i] - ref[i]);
            ref2 += ref[i] * ref[i];
        }
        printf("accuracy over %llu cells (256 x 256 entries): worst relative error %.2e, Frobenius %.2e  [hi hi + hi lo + lo hi, f32 accumulation]\n",
               (unsigned long long)nc, worst, sqrt(sum2 / ref2));
    }
    return 0;
}
