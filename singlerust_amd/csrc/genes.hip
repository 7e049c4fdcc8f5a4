// genes.hip — per-gene (column-direction) passes over the row-major CSR, and the statistics
// entry points of the C ABI.
//
// Per-gene accumulation from a CSR is a scatter: global atomics (one per non-zero, 3.3e9
// for the 1.3M-cell config) are two orders of magnitude too slow, so the accumulators are
// privatised in LDS.  3 accumulators x G genes do not fit 160 KiB, hence GENE TILING: the
// gene axis is cut into tiles of <= 8000 genes (20 B of LDS each: u32 count + f64 sum + f64
// sum of squares); because column indices are sorted inside a row, a tile's entries are one
// contiguous segment of every row, found once per sparsity pattern by binary search
// (k_tile_ptr) and cached on the matrix.  A 1024-thread workgroup (one per CU: it owns the
// whole LDS) takes one (gene tile, row block), walks the row segments one wave per row with
// coalesced loads of indices/values, accumulates with LDS atomics, and flushes its tile to
// a per-block partial buffer; a second tiny kernel sums the partials in fixed order.
//
// Algorithmic bytes: nnz*(4 + s_v) + (N+1)*8 + G*24.
#include "common.hpp"

#include <algorithm>
#include <cmath>

namespace srx {

constexpr int kMaxTileGenes = 8000;       // 8000 * 20 B = 160000 B <= 163840 B of LDS
constexpr int kMomThreads = 1024;

// lower_bound of `bound` inside each row's sorted column list, for the interior tile cuts.
__global__ void k_tile_ptr(const int64_t* __restrict__ indptr, const int32_t* __restrict__ idx,
                           uint64_t n_rows, int n_tiles, int tile_genes, int64_t* __restrict__ tp) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t total = (uint64_t)(n_tiles - 1) * n_rows;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        uint64_t t = i / n_rows + 1, r = i % n_rows;
        int32_t bound = (int32_t)(t * (uint64_t)tile_genes);
        int64_t lo = indptr[r], hi = indptr[r + 1];
        while (lo < hi) {
            int64_t mid = lo + ((hi - lo) >> 1);
            if (idx[mid] < bound) lo = mid + 1; else hi = mid;
        }
        tp[i] = lo;
    }
}

__device__ __forceinline__ void seg_bounds(const int64_t* __restrict__ indptr, const int64_t* __restrict__ tp,
                                           uint64_t n_rows, int n_tiles, int tile, uint64_t r,
                                           int64_t& lo, int64_t& hi) {
    lo = tile == 0 ? indptr[r] : tp[(uint64_t)(tile - 1) * n_rows + r];
    hi = tile == n_tiles - 1 ? indptr[r + 1] : tp[(uint64_t)tile * n_rows + r];
}

// The three column passes of the reference — histogram (csr.rs:29-36), scatter-add of x
// (csr.rs:94-100) and of x^2 (csr.rs:175-178) — in ONE walk.
template <typename T, typename I>
__global__ __launch_bounds__(kMomThreads) void k_gene_moments(
    const int64_t* __restrict__ indptr, const int64_t* __restrict__ tp, const I* __restrict__ idx,
    const T* __restrict__ vals, uint64_t n_rows, uint64_t n_cols, int n_tiles, int tile_genes,
    uint64_t rows_per_block, uint32_t* __restrict__ part_cnt, double* __restrict__ part_sum,
    double* __restrict__ part_sq) {
    extern __shared__ double lds[];
    double* s_sum = lds;
    double* s_sq = lds + tile_genes;
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(lds + 2 * tile_genes);
    for (int g = threadIdx.x; g < tile_genes; g += kMomThreads) { s_sum[g] = 0.0; s_sq[g] = 0.0; s_cnt[g] = 0u; }
    __syncthreads();

    const int tile = blockIdx.x % n_tiles;
    const uint64_t rb = blockIdx.x / n_tiles;
    const int32_t gbase = tile * tile_genes;
    const uint64_t r0 = rb * rows_per_block;
    const uint64_t r1 = r0 + rows_per_block < n_rows ? r0 + rows_per_block : n_rows;
    const int lane = lane_id();
    const int wave = threadIdx.x / kWave;
    constexpr int kWaves = kMomThreads / kWave;

    // 4 consecutive entries per lane: one 16-byte load of the indices and one (f32) or two (f64)
    // of the values, from the 16-byte boundary at or before the segment start; entries outside
    // [lo, hi) are masked (the arrays are padded by 16 entries)
    for (uint64_t r = r0 + wave; r < r1; r += kWaves) {
        int64_t lo, hi;
        seg_bounds(indptr, tp, n_rows, n_tiles, tile, r, lo, hi);
        const int64_t base = lo & ~(int64_t)3;
        for (int64_t e0 = base + 4 * lane; e0 < hi; e0 += 4 * kWave) {
            int gg[4];
            if constexpr (sizeof(I) == 4) {
                const int4 g4 = *reinterpret_cast<const int4*>(idx + e0);
                gg[0] = g4.x; gg[1] = g4.y; gg[2] = g4.z; gg[3] = g4.w;
            } else {                                   // four 16-bit indices in one 8-byte load
                const uint2 g2 = *reinterpret_cast<const uint2*>(idx + e0);
                gg[0] = (int)(g2.x & 0xffffu); gg[1] = (int)(g2.x >> 16);
                gg[2] = (int)(g2.y & 0xffffu); gg[3] = (int)(g2.y >> 16);
            }
            T v[4];
            if constexpr (sizeof(T) == 4) {
                const float4 t4 = *reinterpret_cast<const float4*>(vals + e0);
                v[0] = t4.x; v[1] = t4.y; v[2] = t4.z; v[3] = t4.w;
            } else {
                const double2 a2 = *reinterpret_cast<const double2*>(vals + e0);
                const double2 b2 = *reinterpret_cast<const double2*>(vals + e0 + 2);
                v[0] = a2.x; v[1] = a2.y; v[2] = b2.x; v[3] = b2.y;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t pos = e0 + j;
                if (pos >= lo && pos < hi) {
                    const int32_t g0 = gg[j] - gbase;
                    const double x0 = (double)v[j];
                    __hip_atomic_fetch_add(&s_cnt[g0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(&s_sum[g0], x0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(&s_sq[g0], x0 * x0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }
    __syncthreads();
    for (int g = threadIdx.x; g < tile_genes; g += kMomThreads) {
        uint64_t gene = (uint64_t)gbase + g;
        if (gene < n_cols) {
            part_cnt[rb * n_cols + gene] = s_cnt[g];
            part_sum[rb * n_cols + gene] = s_sum[g];
            part_sq[rb * n_cols + gene] = s_sq[g];
        }
    }
}

// Fixed-order sum of the per-row-block partials -> packed f64 [cnt | sum | sq | n_rows].
__global__ void k_moments_reduce(const uint32_t* __restrict__ part_cnt, const double* __restrict__ part_sum,
                                 const double* __restrict__ part_sq, uint64_t n_cols, uint64_t n_blocks,
                                 uint64_t n_rows, double* __restrict__ packed) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) packed[3 * n_cols] = (double)n_rows;
    if (j >= n_cols) return;
    uint64_t c = 0;
    double s = 0.0, q = 0.0;
    for (uint64_t b = 0; b < n_blocks; ++b) {
        c += part_cnt[b * n_cols + j];
        s += part_sum[b * n_cols + j];
        q += part_sq[b * n_cols + j];
    }
    packed[j] = (double)c;
    packed[n_cols + j] = s;
    packed[2 * n_cols + j] = q;
}

__global__ void k_moments_unpack(const double* __restrict__ packed, uint64_t n_cols, uint64_t* __restrict__ cnt,
                                 double* __restrict__ sum, double* __restrict__ sq) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_cols) return;
    cnt[j] = (uint64_t)packed[j];
    sum[j] = packed[n_cols + j];
    sq[j] = packed[2 * n_cols + j];
}

// ---- per-gene min / max (csr.rs:212-221) ------------------------------------------------------
// Order-preserving map double -> u64 so LDS integer atomics (ds_min_u64 / ds_max_u64) apply.
__device__ __host__ __forceinline__ uint64_t f64_key(double x) {
    uint64_t b;
    memcpy(&b, &x, 8);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __host__ __forceinline__ double key_f64(uint64_t k) {
    uint64_t b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    double x;
    memcpy(&x, &b, 8);
    return x;
}

template <typename T>
__global__ __launch_bounds__(kMomThreads) void k_gene_minmax(
    const int64_t* __restrict__ indptr, const int64_t* __restrict__ tp, const int32_t* __restrict__ idx,
    const T* __restrict__ vals, uint64_t n_rows, uint64_t n_cols, int n_tiles, int tile_genes,
    uint64_t rows_per_block, unsigned long long* __restrict__ g_min, unsigned long long* __restrict__ g_max) {
    extern __shared__ double lds[];
    unsigned long long* s_min = reinterpret_cast<unsigned long long*>(lds);
    unsigned long long* s_max = s_min + tile_genes;
    const unsigned long long kInf = f64_key(INFINITY), kNinf = f64_key(-INFINITY);
    for (int g = threadIdx.x; g < tile_genes; g += kMomThreads) { s_min[g] = kInf; s_max[g] = kNinf; }
    __syncthreads();
    const int tile = blockIdx.x % n_tiles;
    const uint64_t rb = blockIdx.x / n_tiles;
    const int32_t gbase = tile * tile_genes;
    const uint64_t r0 = rb * rows_per_block;
    const uint64_t r1 = r0 + rows_per_block < n_rows ? r0 + rows_per_block : n_rows;
    const int lane = lane_id();
    const int wave = threadIdx.x / kWave;
    constexpr int kWaves = kMomThreads / kWave;
    for (uint64_t r = r0 + wave; r < r1; r += kWaves) {
        int64_t lo, hi;
        seg_bounds(indptr, tp, n_rows, n_tiles, tile, r, lo, hi);
        for (int64_t p = lo + lane; p < hi; p += kWave) {
            double x = (double)vals[p];
            if (x != x) continue;                      // f64::min/max ignore a NaN operand
            int32_t g = idx[p] - gbase;
            unsigned long long k = f64_key(x);
            atomicMin(&s_min[g], k);
            atomicMax(&s_max[g], k);
        }
    }
    __syncthreads();
    for (int g = threadIdx.x; g < tile_genes; g += kMomThreads) {
        uint64_t gene = (uint64_t)gbase + g;
        if (gene < n_cols) {
            if (s_min[g] != kInf) atomicMin(&g_min[gene], s_min[g]);
            if (s_max[g] != kNinf) atomicMax(&g_max[gene], s_max[g]);
        }
    }
}

__global__ void k_fill_u64(unsigned long long* p, uint64_t n, unsigned long long v) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// normalize_total(Column) on CSR: scale/mod.rs:141-173, values[j] *= scale[col].
template <typename T>
__global__ void k_col_scale(const int32_t* __restrict__ idx, T* __restrict__ vals, uint64_t nnz,
                            const double* __restrict__ sum, double target) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < nnz; i += stride) {
        double s = sum[idx[i]];
        double sc = s == 0.0 ? 0.0 : target / s;      // scale/mod.rs:93-99
        vals[i] = (T)((double)vals[i] * sc);
    }
}

static void tile_geometry(const srx_mat* m, int& n_tiles, int& tile_genes) {
    uint64_t G = m->n_cols ? m->n_cols : 1;
    n_tiles = (int)((G + kMaxTileGenes - 1) / kMaxTileGenes);
    tile_genes = (int)((G + n_tiles - 1) / n_tiles);
}

int32_t launch_tile_ptr(srx_ctx* ctx, const int64_t* indptr, const int32_t* idx, uint64_t n_rows, int n_tiles,
                        int tile_genes, int64_t* tp) {
    uint64_t total = (uint64_t)(n_tiles - 1) * n_rows;
    uint64_t g = (total + 255) / 256;
    if (g < 1) g = 1;
    if (g > 65535) g = 65535;
    hipLaunchKernelGGL(k_tile_ptr, dim3((unsigned)g), dim3(256), 0, ctx->stream, indptr, idx, n_rows, n_tiles,
                       tile_genes, tp);
    SRX_HIP(ctx, hipGetLastError());
    return SRX_OK;
}

// d_indices -> 16-bit mirror (entries past nnz: 0, the arrays are padded for the vector walks)
__global__ void k_narrow16(const int32_t* __restrict__ idx, uint64_t nnz, uint64_t n_out, uint16_t* __restrict__ out) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; e < n_out; e += stride) out[e] = e < nnz ? (uint16_t)idx[e] : (uint16_t)0;
}

int32_t ensure_tiles(srx_mat* m) {
    srx_ctx* ctx = m->ctx;
    if (m->n_tiles) return SRX_OK;
    int nt, tg;
    tile_geometry(m, nt, tg);
    if (nt > 1) {
        SRX_HIP(ctx, hipMalloc((void**)&m->d_tile_ptr, (size_t)(nt - 1) * (m->n_rows ? m->n_rows : 1) * sizeof(int64_t)));
        SRX_TRY(launch_tile_ptr(ctx, m->d_indptr, m->d_indices, m->n_rows, nt, tg, m->d_tile_ptr));
    }
    if (m->n_cols <= 65536 && !m->d_idx16) {
        SRX_HIP(ctx, hipMalloc((void**)&m->d_idx16, (m->nnz + 16) * sizeof(uint16_t)));
        uint64_t g = (m->nnz + 16 + 1023) / 1024;
        if (g < 1) g = 1;
        if (g > 65535) g = 65535;
        hipLaunchKernelGGL(k_narrow16, dim3((unsigned)g), dim3(256), 0, ctx->stream, m->d_indices, m->nnz, m->nnz + 16,
                           m->d_idx16);
        SRX_HIP(ctx, hipGetLastError());
    }
    m->n_tiles = nt;
    m->tile_genes = tg;
    return SRX_OK;
}

static void block_geometry(const srx_mat* m, uint64_t& n_blocks, uint64_t& rows_per_block) {
    uint64_t want = (uint64_t)(2 * m->ctx->n_cus) / (uint64_t)m->n_tiles;
    if (want < 1) want = 1;
    uint64_t by_rows = (m->n_rows + 63) / 64;   // at least ~64 rows per block
    if (by_rows < 1) by_rows = 1;
    n_blocks = want < by_rows ? want : by_rows;
    rows_per_block = (m->n_rows + n_blocks - 1) / n_blocks;
    if (rows_per_block < 1) rows_per_block = 1;
}

int32_t ensure_moments(srx_mat* m) {
    srx_ctx* ctx = m->ctx;
    if (m->moments_version == m->version) return SRX_OK;
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    SRX_TRY(ensure_tiles(m));
    const uint64_t G = m->n_cols;
    if (!m->d_cnt) {
        SRX_HIP(ctx, hipMalloc((void**)&m->d_cnt, (G ? G : 1) * sizeof(uint64_t)));
        SRX_HIP(ctx, hipMalloc((void**)&m->d_sum, (G ? G : 1) * sizeof(double)));
        SRX_HIP(ctx, hipMalloc((void**)&m->d_sq, (G ? G : 1) * sizeof(double)));
    }
    uint64_t nb, rpb;
    block_geometry(m, nb, rpb);
    uint32_t* p_cnt;
    double *p_sum, *p_sq, *packed;
    SRX_TRY(scratch(ctx, "mom_part_cnt", nb * (G ? G : 1) * sizeof(uint32_t), (void**)&p_cnt));
    SRX_TRY(scratch(ctx, "mom_part_sum", nb * (G ? G : 1) * sizeof(double), (void**)&p_sum));
    SRX_TRY(scratch(ctx, "mom_part_sq", nb * (G ? G : 1) * sizeof(double), (void**)&p_sq));
    SRX_TRY(scratch(ctx, "mom_packed", (3 * G + 1) * sizeof(double), (void**)&packed));
    const size_t lds = (size_t)m->tile_genes * 20;
    // s_i = 2 when the 16-bit index mirror exists (n_cols <= 65536), 4 otherwise
    const double bytes = (double)m->nnz * ((m->n_cols <= 65536 ? 2.0 : 4.0) + val_bytes(m)) + (double)(m->n_rows + 1) * 8.0 +
                         (double)G * 24.0;
    {
        ProfScope ps(ctx, SRX_K_MOMENTS, bytes);
        dim3 grid((unsigned)(nb * m->n_tiles));
        auto launch = [&](auto kern, const auto* idxp, const auto* valp) -> int32_t {
            SRX_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, grid, dim3(kMomThreads), lds, ctx->stream, m->d_indptr, m->d_tile_ptr, idxp, valp,
                               m->n_rows, G, m->n_tiles, m->tile_genes, rpb, p_cnt, p_sum, p_sq);
            return SRX_OK;
        };
        if (is_f32(m)) {
            if (m->d_idx16) SRX_TRY(launch(k_gene_moments<float, uint16_t>, (const uint16_t*)m->d_idx16, (const float*)m->d_values));
            else SRX_TRY(launch(k_gene_moments<float, int32_t>, (const int32_t*)m->d_indices, (const float*)m->d_values));
        } else {
            if (m->d_idx16) SRX_TRY(launch(k_gene_moments<double, uint16_t>, (const uint16_t*)m->d_idx16, (const double*)m->d_values));
            else SRX_TRY(launch(k_gene_moments<double, int32_t>, (const int32_t*)m->d_indices, (const double*)m->d_values));
        }
        hipLaunchKernelGGL(k_moments_reduce, dim3((unsigned)((G + 255) / 256 + 1)), dim3(256), 0, ctx->stream, p_cnt,
                           p_sum, p_sq, G, nb, m->n_rows, packed);
    }
    SRX_HIP(ctx, hipGetLastError());
    // one all-reduce for (cnt, sum, sumsq, N) across the row shards
    SRX_TRY(allreduce_f64(ctx, packed, 3 * G + 1));
    hipLaunchKernelGGL(k_moments_unpack, dim3((unsigned)((G + 255) / 256 + 1)), dim3(256), 0, ctx->stream, packed, G,
                       m->d_cnt, m->d_sum, m->d_sq);
    SRX_HIP(ctx, hipGetLastError());
    if (ctx->comm) {
        double ng = 0.0;
        SRX_TRY(d2h(ctx, &ng, packed + 3 * G, sizeof(double)));
        m->n_rows_global = (uint64_t)ng;
    } else {
        m->n_rows_global = m->n_rows;
    }
    m->moments_version = m->version;
    return SRX_OK;
}

static int32_t fetch_moments(srx_mat* m, std::vector<uint64_t>& cnt, std::vector<double>& sum, std::vector<double>& sq) {
    SRX_TRY(ensure_moments(m));
    const uint64_t G = m->n_cols;
    cnt.resize(G); sum.resize(G); sq.resize(G);
    SRX_TRY(d2h(m->ctx, cnt.data(), m->d_cnt, G * sizeof(uint64_t)));
    SRX_TRY(d2h(m->ctx, sum.data(), m->d_sum, G * sizeof(double)));
    SRX_TRY(d2h(m->ctx, sq.data(), m->d_sq, G * sizeof(double)));
    return SRX_OK;
}

// csr.rs:179-185, evaluated in the reference's operation order (no FMA contraction).
#pragma clang fp contract(off)
void finalize_variance(const uint64_t* cnt, const double* sum, const double* sq, uint64_t G, double* out) {
    for (uint64_t j = 0; j < G; ++j) {
        out[j] = 0.0;
        if (cnt[j] > 0) {
            double c = (double)(uint32_t)cnt[j];
            double mean = sum[j] / c;
            out[j] = sq[j] / c - mean * mean;
        }
    }
}

int32_t gene_variances(srx_mat* m, std::vector<double>& var) {
    std::vector<uint64_t> cnt;
    std::vector<double> sum, sq;
    SRX_TRY(fetch_moments(m, cnt, sum, sq));
    var.resize(m->n_cols);
    finalize_variance(cnt.data(), sum.data(), sq.data(), m->n_cols, var.data());
    return SRX_OK;
}

// dim_red/mod.rs:135-140.  Rust's sort_by is stable; comparator b.partial_cmp(a) = descending.
int32_t select_hvg_host(srx_ctx* ctx, const std::vector<double>& var, uint64_t n, std::vector<uint64_t>& out) {
    for (double v : var)
        if (v != v) return fail(ctx, SRX_E_NAN, "NaN gene variance: called `Option::unwrap()` on a `None` value (partial_cmp)");
    // stable descending order == the strict total order (variance desc, gene index asc), so a
    // partial sort of the first n under that order gives exactly the stable sort's prefix
    std::vector<uint64_t> order(var.size());
    for (uint64_t j = 0; j < order.size(); ++j) order[j] = j;
    uint64_t take = n < order.size() ? n : order.size();
    auto before = [&](uint64_t a, uint64_t b) { return var[a] > var[b] || (var[a] == var[b] && a < b); };
    std::nth_element(order.begin(), order.begin() + (take ? take - 1 : 0), order.end(), before);
    std::sort(order.begin(), order.begin() + take, before);
    out.assign(order.begin(), order.begin() + take);
    return SRX_OK;
}

int32_t row_number(srx_mat* m, uint32_t* out);
int32_t row_stat(srx_mat* m, int which, double* out0, double* out1);

}  // namespace srx

using namespace srx;

extern "C" {

int32_t srx_gene_moments(srx_mat* m, uint64_t* cnt, double* sum, double* sumsq) {
    if (!m) return fail(nullptr, SRX_E_ARG, "null matrix");
    SRX_TRY(ensure_moments(m));
    const uint64_t G = m->n_cols;
    if (cnt) SRX_TRY(d2h(m->ctx, cnt, m->d_cnt, G * sizeof(uint64_t)));
    if (sum) SRX_TRY(d2h(m->ctx, sum, m->d_sum, G * sizeof(double)));
    if (sumsq) SRX_TRY(d2h(m->ctx, sumsq, m->d_sq, G * sizeof(double)));
    return SRX_OK;
}

int32_t srx_compute_number(srx_mat* m, int32_t direction, uint32_t* out) {
    if (!m || !out) return fail(m ? m->ctx : nullptr, SRX_E_ARG, "null argument");
    SRX_HIP(m->ctx, hipSetDevice(m->ctx->device));
    if (direction == SRX_ROW) return row_number(m, out);
    if (direction != SRX_COLUMN) return fail(m->ctx, SRX_E_ARG, "bad direction %d", direction);
    std::vector<uint64_t> cnt(m->n_cols);
    SRX_TRY(srx_gene_moments(m, cnt.data(), nullptr, nullptr));
    for (uint64_t j = 0; j < m->n_cols; ++j) out[j] = (uint32_t)cnt[j];   // reference counts in u32
    return SRX_OK;
}

int32_t srx_compute_sum(srx_mat* m, int32_t direction, double* out) {
    if (!m || !out) return fail(m ? m->ctx : nullptr, SRX_E_ARG, "null argument");
    SRX_HIP(m->ctx, hipSetDevice(m->ctx->device));
    if (direction == SRX_ROW) return row_stat(m, 0, out, nullptr);
    if (direction != SRX_COLUMN) return fail(m->ctx, SRX_E_ARG, "bad direction %d", direction);
    return srx_gene_moments(m, nullptr, out, nullptr);
}

int32_t srx_compute_variance(srx_mat* m, int32_t direction, double* out) {
    if (!m || !out) return fail(m ? m->ctx : nullptr, SRX_E_ARG, "null argument");
    SRX_HIP(m->ctx, hipSetDevice(m->ctx->device));
    if (direction == SRX_ROW) return row_stat(m, 1, out, nullptr);
    if (direction != SRX_COLUMN) return fail(m->ctx, SRX_E_ARG, "bad direction %d", direction);
    std::vector<double> var;
    SRX_TRY(gene_variances(m, var));
    memcpy(out, var.data(), var.size() * sizeof(double));
    return SRX_OK;
}

int32_t srx_compute_std_dev(srx_mat* m, int32_t direction, double* out) {
    SRX_TRY(srx_compute_variance(m, direction, out));
    uint64_t n = direction == SRX_ROW ? m->n_rows : m->n_cols;
    for (uint64_t i = 0; i < n; ++i) out[i] = std::sqrt(out[i]);   // csr.rs:227
    return SRX_OK;
}

int32_t srx_compute_min_max(srx_mat* m, int32_t direction, double* mn, double* mx) {
    if (!m || !mn || !mx) return fail(m ? m->ctx : nullptr, SRX_E_ARG, "null argument");
    srx_ctx* ctx = m->ctx;
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    if (direction == SRX_ROW) return row_stat(m, 2, mn, mx);
    if (direction != SRX_COLUMN) return fail(ctx, SRX_E_ARG, "bad direction %d", direction);
    SRX_TRY(ensure_tiles(m));
    const uint64_t G = m->n_cols;
    unsigned long long* d;
    SRX_TRY(scratch(ctx, "minmax_keys", 2 * (G ? G : 1) * sizeof(unsigned long long), (void**)&d));
    unsigned g1 = (unsigned)((G + 255) / 256 + 1);
    hipLaunchKernelGGL(k_fill_u64, dim3(g1), dim3(256), 0, ctx->stream, d, G, (unsigned long long)f64_key(INFINITY));
    hipLaunchKernelGGL(k_fill_u64, dim3(g1), dim3(256), 0, ctx->stream, d + G, G, (unsigned long long)f64_key(-INFINITY));
    uint64_t nb, rpb;
    block_geometry(m, nb, rpb);
    const size_t lds = (size_t)m->tile_genes * 16;
    dim3 grid((unsigned)(nb * m->n_tiles));
    if (is_f32(m)) {
        SRX_HIP(ctx, hipFuncSetAttribute((const void*)k_gene_minmax<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_gene_minmax<float>), grid, dim3(kMomThreads), lds, ctx->stream, m->d_indptr, m->d_tile_ptr,
                           m->d_indices, (const float*)m->d_values, m->n_rows, G, m->n_tiles, m->tile_genes, rpb, d, d + G);
    } else {
        SRX_HIP(ctx, hipFuncSetAttribute((const void*)k_gene_minmax<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_gene_minmax<double>), grid, dim3(kMomThreads), lds, ctx->stream, m->d_indptr, m->d_tile_ptr,
                           m->d_indices, (const double*)m->d_values, m->n_rows, G, m->n_tiles, m->tile_genes, rpb, d, d + G);
    }
    SRX_HIP(ctx, hipGetLastError());
    std::vector<unsigned long long> keys(2 * G);
    SRX_TRY(d2h(ctx, keys.data(), d, 2 * G * sizeof(unsigned long long)));
    for (uint64_t j = 0; j < G; ++j) { mn[j] = key_f64(keys[j]); mx[j] = key_f64(keys[G + j]); }
    return SRX_OK;
}

int32_t srx_normalize_total_inplace(srx_mat* m, double target_sum, int32_t direction) {
    if (!m) return fail(nullptr, SRX_E_ARG, "null matrix");
    srx_ctx* ctx = m->ctx;
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    if (direction == SRX_ROW) return launch_normalize(m, target_sum, true, false);
    if (direction != SRX_COLUMN) return fail(ctx, SRX_E_ARG, "bad direction %d", direction);
    SRX_TRY(ensure_moments(m));
    uint64_t g = (m->nnz + 255) / 256;
    if (g < 1) g = 1;
    if (g > 8192) g = 8192;
    if (is_f32(m))
        hipLaunchKernelGGL((k_col_scale<float>), dim3((unsigned)g), dim3(256), 0, ctx->stream, m->d_indices,
                           (float*)m->d_values, m->nnz, m->d_sum, target_sum);
    else
        hipLaunchKernelGGL((k_col_scale<double>), dim3((unsigned)g), dim3(256), 0, ctx->stream, m->d_indices,
                           (double*)m->d_values, m->nnz, m->d_sum, target_sum);
    SRX_HIP(ctx, hipGetLastError());
    m->dtype = SRX_F64;
    touch(m);
    return SRX_OK;
}

int32_t srx_select_hvg(srx_mat* m, uint64_t n, uint64_t* idx_out, uint64_t* n_out) {
    if (!m || !n_out) return fail(m ? m->ctx : nullptr, SRX_E_ARG, "null argument");
    SRX_HIP(m->ctx, hipSetDevice(m->ctx->device));
    std::vector<double> var;
    SRX_TRY(gene_variances(m, var));
    std::vector<uint64_t> sel;
    SRX_TRY(select_hvg_host(m->ctx, var, n, sel));
    if (idx_out) memcpy(idx_out, sel.data(), sel.size() * sizeof(uint64_t));
    *n_out = sel.size();
    return SRX_OK;
}

}  // extern "C"
