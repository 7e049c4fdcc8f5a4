// Microbenchmark: what the (row block, gene tile) segment walk of k_gene_moments can read at.  The matrix: N rows of ROWLEN entries
// (16-bit column index + f32 value, two arrays), every row cut into 3 gene-tile segments at jittered positions.  A workgroup of the
// walk owns (a block of rows, ONE tile): it reads every third ~1.7 KB piece of the two arrays.
//   flat     : waves stream the arrays front to back (8-byte index + 16-byte value loads per lane): the ceiling
//   segment  : workgroup = (row block, tile), 16 waves, a wave per row segment, 4 entries per lane, 4-aligned start (the round-3 walk)
//   batch    : the same with 3 row segments laid end to end in one slot space and 4 steps of loads in flight (round 4's walk)
// each with and without the in-place 16-byte store of the values, at 1 workgroup per CU (148 KB of LDS held, like the kernel) or 2.
// Build: hipcc --offload-arch=gfx950 -O3 -o segment_read segment_read.hip        Run: ./segment_read [n_rows]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ROWLEN = 840, NT = 3;
__host__ __device__ inline uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// segment t of row r: entries [cut(r, t), cut(r, t + 1)) of the row
__host__ __device__ inline int cut(uint32_t r, int t) {
    if (t <= 0) return 0;
    if (t >= NT) return ROWLEN;
    return t * (ROWLEN / NT) + (int)(hash32(r * 3u + (uint32_t)t) % 65u) - 32;
}

__global__ __launch_bounds__(1024) void k_flat(const uint2* __restrict__ idx, float4* vals, uint64_t n_chunks, int store, float* out) {
    float acc = 0.f;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += stride) {
        const uint2 i2 = idx[c];
        float4 v = vals[c];
        acc += v.x + v.y + v.z + v.w + (float)(i2.x ^ i2.y);
        if (store) { v.x *= 1.0001f; v.y *= 1.0001f; v.z *= 1.0001f; v.w *= 1.0001f; vals[c] = v; }
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int MODE /* 0: one segment per wave visit, 1: batches of 3 segments / 4 steps in flight */>
__global__ __launch_bounds__(1024) void k_seg(const uint16_t* __restrict__ idx, float* vals, uint32_t n_rows, uint32_t rows_per_block, int store,
                                              float* out) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = 0.f;
    const int tile = blockIdx.x % NT;
    const uint32_t rb = blockIdx.x / NT;
    const uint32_t r0 = rb * rows_per_block, r1 = min(r0 + rows_per_block, n_rows);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    auto chunk = [&](uint64_t e0, int rel, int len) {      // entries e0 .. e0 + 3 (4-aligned), `rel` = position of e0 in the segment
        const uint2 i2 = *reinterpret_cast<const uint2*>(idx + e0);
        float4 v = *reinterpret_cast<const float4*>(vals + e0);
        acc += v.x + v.y + v.z + v.w + (float)(i2.x ^ i2.y);
        if (store && rel >= 0 && rel + 3 < len) {
            v.x *= 1.0001f; v.y *= 1.0001f; v.z *= 1.0001f; v.w *= 1.0001f;
            *reinterpret_cast<float4*>(vals + e0) = v;
        }
    };
    if (MODE == 0) {
        for (uint32_t r = r0 + wave; r < r1; r += 16) {
            const uint64_t lo = (uint64_t)r * ROWLEN + cut(r, tile), hi = (uint64_t)r * ROWLEN + cut(r, tile + 1);
            const uint64_t b0 = lo & ~3ull;
            const int len = (int)(hi - lo), a = (int)(lo - b0);
            for (int s = lane; 4 * s < a + len; s += 64) chunk(b0 + 4 * s, 4 * s - a, len);
        }
    } else {
        for (uint32_t rbase = r0 + wave; rbase < r1; rbase += 48) {
            uint64_t b0[3];
            int a[3], len[3], S[4];
            S[0] = 0;
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const uint32_t r = rbase + 16 * u;
                uint64_t lo = 0, hi = 0;
                if (r < r1) { lo = (uint64_t)r * ROWLEN + cut(r, tile); hi = (uint64_t)r * ROWLEN + cut(r, tile + 1); }
                b0[u] = lo & ~3ull;
                a[u] = (int)(lo - b0[u]);
                len[u] = (int)(hi - lo);
                S[u + 1] = S[u] + ((a[u] + len[u] + 3) >> 2);
            }
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int slot = st * 64 + lane;
                if (slot < S[3]) {
                    const int u = slot >= S[2] ? 2 : slot >= S[1] ? 1 : 0;
                    const int s = slot - (u == 2 ? S[2] : u == 1 ? S[1] : 0);
                    const uint64_t bb = u == 2 ? b0[2] : u == 1 ? b0[1] : b0[0];
                    const int aa = u == 2 ? a[2] : u == 1 ? a[1] : a[0], ll = u == 2 ? len[2] : u == 1 ? len[1] : len[0];
                    chunk(bb + 4 * s, 4 * s - aa, ll);
                }
            }
            for (int slot = 256 + lane; slot < S[3]; slot += 64) {
                const int u = slot >= S[2] ? 2 : slot >= S[1] ? 1 : 0;
                const int s = slot - (u == 2 ? S[2] : u == 1 ? S[1] : 0);
                const uint64_t bb = u == 2 ? b0[2] : u == 1 ? b0[1] : b0[0];
                const int aa = u == 2 ? a[2] : u == 1 ? a[1] : a[0], ll = u == 2 ? len[2] : u == 1 ? len[1] : len[0];
                chunk(bb + 4 * s, 4 * s - aa, ll);
            }
        }
    }
    if (acc == 12345.678f) out[0] = acc + lds[0];
}

int main(int argc, char** argv) {
    const uint32_t N = argc > 1 ? (uint32_t)strtoul(argv[1], nullptr, 10) : 1300000u;
    const uint64_t nnz = (uint64_t)N * ROWLEN;
    uint16_t* d_idx; float *d_vals, *d_out;
    CK(hipMalloc(&d_idx, nnz * 2 + 64)); CK(hipMalloc(&d_vals, nnz * 4 + 64)); CK(hipMalloc(&d_out, 64));
    CK(hipMemset(d_idx, 1, nnz * 2 + 64)); CK(hipMemset(d_vals, 0, nnz * 4 + 64));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const double gb_r = (double)nnz * 6e-9, gb_w = (double)nnz * 4e-9;
    auto report = [&](const char* name, float ms, int store) {
        printf("%-44s %7.3f ms  %6.2f TB/s read%s\n", name, ms, gb_r / ms, store ? " (+ the values written back)" : "");
    };
    auto time = [&](auto&& f) {
        f();
        CK(hipEventRecord(a));
        for (int i = 0; i < 3; ++i) f();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        return ms / 3;
    };
    (void)gb_w;
    for (int store = 0; store < 2; ++store) {
        report("flat stream, 2048 workgroups", time([&] { k_flat<<<2048, 1024>>>((const uint2*)d_idx, (float4*)d_vals, nnz / 4, store, d_out); }), store);
        for (int per_cu = 1; per_cu <= 2; ++per_cu) {
            const size_t lds = per_cu == 1 ? 151392 : 65536;
            CK(hipFuncSetAttribute((const void*)k_seg<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            CK(hipFuncSetAttribute((const void*)k_seg<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const uint32_t n_rb = 341, rpb = (N + n_rb - 1) / n_rb;
            char nm[96];
            snprintf(nm, sizeof nm, "segment walk, %d workgroup(s) per CU", per_cu);
            report(nm, time([&] { k_seg<0><<<n_rb * NT, 1024, lds>>>(d_idx, d_vals, N, rpb, store, d_out); }), store);
            snprintf(nm, sizeof nm, "batches of 3 segments, %d workgroup(s) per CU", per_cu);
            report(nm, time([&] { k_seg<1><<<n_rb * NT, 1024, lds>>>(d_idx, d_vals, N, rpb, store, d_out); }), store);
        }
    }
    CK(hipGetLastError());
    return 0;
}
