// rows.hip — per-cell (row-direction) kernels: nnz counts, row sums, the fused
// normalise + log1p pass, row variance, row min/max.
//
// All of them are one-wave-per-row CSR walks: the 64 lanes read consecutive values of the
// row (coalesced 256 B per instruction, 16 independent loads in flight per lane), reduce
// with wavefront shuffles in f64, and — for the in-place ops — write the row back from
// registers, so a row is read from HBM exactly once.
//
// Algorithmic bytes: fused normalise+log1p = nnz*2*s_v + (N+1)*8.
#include "common.hpp"
#include "log1p64.hpp"

namespace srx {

constexpr int kRowCache = 32;  // values per lane kept in registers: rows up to 2048 nnz are read from HBM once
                               // (16 left 26 % of the c3 rows — log-normal sizes, mean 840 — in the scalar tail loop)

// 16-byte vector of row values: 4 x f32 or 2 x f64.
template <typename T>
struct alignas(16) RowVec {
    T x[16 / sizeof(T)];
};

// Fused: s_i = sum_row f64(v); scale_i = (s_i == 0) ? 0 : target / s_i; v = v * scale_i
// (scale/mod.rs:9-15,66-73: one division, one multiply — not v*target/s); then optionally
// v = ln_1p(v) (transform/mod.rs:38-47).
// One wave per row.  The row is read with 16-byte loads from the 16-byte boundary at or before
// its first entry (elements outside [lo, hi) are masked; the arrays are padded by 16 entries),
// 16 values per lane stay in registers between the reduction and the write-back, so a row is
// read from HBM exactly once; interior vectors are written back with 16-byte stores.
// PRECISE (with NORM and LOG): y = ln_1p(f64(v) * scale) evaluated in f64 and rounded ONCE to the storage type — the
// value the reference's two calls produce (scale/mod.rs:66-83 promotes to f64, transform/mod.rs:38-42), and the value
// the pipeline's moments / compaction passes computed on the fly from the raw matrix before this write-back.
template <typename T, bool NORM, bool LOG, bool PRECISE = false>
__global__ __launch_bounds__(256) void k_row_pass(const int64_t* __restrict__ indptr, T* __restrict__ vals,
                                                  uint64_t n_rows, double target,
                                                  double* __restrict__ row_sum_out) {
    __shared__ Log1pTabEntry s_tab[PRECISE ? 128 : 1];
    if constexpr (PRECISE) {
        stage_log1p_table(s_tab);
        __syncthreads();
    }
    constexpr int V = 16 / sizeof(T);
    constexpr int NV = kRowCache / V;
    using Vec = RowVec<T>;
    const uint64_t wave = global_wave_id();
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) / kWave;
    const int lane = lane_id();
    for (uint64_t r = wave; r < n_rows; r += n_waves) {
        const int64_t lo = indptr[r], hi = indptr[r + 1];
        const int64_t base = lo & ~(int64_t)(V - 1);
        const int64_t tail = base + (int64_t)NV * kWave * V;        // first element not covered by the cache
        Vec c[NV];
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            const int64_t e0 = base + ((int64_t)t * kWave + lane) * V;
            if (e0 < hi) c[t] = *reinterpret_cast<const Vec*>(vals + e0);
            else {
#pragma unroll
                for (int j = 0; j < V; ++j) c[t].x[j] = T(0);
            }
        }
        if (NORM) {
#pragma unroll
            for (int t = 0; t < NV; ++t) {
                const int64_t e0 = base + ((int64_t)t * kWave + lane) * V;
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const int64_t pos = e0 + j;
                    s += (pos >= lo && pos < hi) ? (double)c[t].x[j] : 0.0;
                }
            }
            for (int64_t p = tail + lane; p < hi; p += kWave) s += (double)vals[p];
            s = wave_sum(s);
            if (row_sum_out && lane == 0) row_sum_out[r] = s;
        }
        const double scale = NORM ? (s == 0.0 ? 0.0 : target / s) : 1.0;
        double row_table = 0.0;
        if constexpr (PRECISE) row_table = xf_row_table(scale, s_tab);
        auto f = [&](T v) -> T {
            if constexpr (PRECISE) return (T)xf_apply<T>(v, scale, row_table, s_tab);      // every lane active (shuffle)
            T x = NORM ? (T)((double)v * scale) : v;
            return LOG ? apply_log1p<T>(x) : x;
        };
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            const int64_t e0 = base + ((int64_t)t * kWave + lane) * V;
            Vec o;
            if constexpr (PRECISE) {                       // outside the lane predicate: f() shuffles
                if (__any(e0 < hi)) {
#pragma unroll
                    for (int j = 0; j < V; ++j) o.x[j] = f(c[t].x[j]);
                }
            }
            if (e0 < hi && e0 + V > lo) {
                if constexpr (!PRECISE) {
#pragma unroll
                    for (int j = 0; j < V; ++j) o.x[j] = f(c[t].x[j]);
                }
                if (e0 >= lo && e0 + V <= hi) {
                    *reinterpret_cast<Vec*>(vals + e0) = o;
                } else {
#pragma unroll
                    for (int j = 0; j < V; ++j) {
                        const int64_t pos = e0 + j;
                        if (pos >= lo && pos < hi) vals[pos] = o.x[j];
                    }
                }
            }
        }
        if constexpr (PRECISE) {
            for (int64_t p0 = tail; p0 < hi; p0 += kWave) {
                const int64_t p = p0 + lane;
                const T y = f(p < hi ? vals[p] : T(0));
                if (p < hi) vals[p] = y;
            }
        } else {
            for (int64_t p = tail + lane; p < hi; p += kWave) vals[p] = f(vals[p]);
        }
    }
}

// compute_sum(Row): csr.rs:87-93.
template <typename T>
__global__ __launch_bounds__(256) void k_row_sum(const int64_t* __restrict__ indptr, const T* __restrict__ vals,
                                                 uint64_t n_rows, double* __restrict__ out) {
    const uint64_t wave = global_wave_id();
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) / kWave;
    const int lane = lane_id();
    for (uint64_t r = wave; r < n_rows; r += n_waves) {
        const int64_t lo = indptr[r], hi = indptr[r + 1];
        double s = 0.0;
        if constexpr (sizeof(T) == 4) {
            // four consecutive values per lane and load (16 bytes at the row's own 4-byte alignment: fine on gfx950) — one value per
            // lane was 17M load instructions per launch at c3, ~0.5 ms of the memory pipeline's instruction rate alone.  (f32
            // values summed in f64: the order does not show unless a row spans more than 2^29 in magnitude.)
            typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
            const int64_t n4 = (hi - lo) >> 2;
            int64_t q = lane;
            for (; q + kWave < n4; q += 2 * kWave) {
                const f4u a = *reinterpret_cast<const f4u*>(vals + lo + 4 * q), b = *reinterpret_cast<const f4u*>(vals + lo + 4 * (q + kWave));
                s += ((double)a.x + (double)a.y) + ((double)a.z + (double)a.w);
                s += ((double)b.x + (double)b.y) + ((double)b.z + (double)b.w);
            }
            if (q < n4) {
                const f4u a = *reinterpret_cast<const f4u*>(vals + lo + 4 * q);
                s += ((double)a.x + (double)a.y) + ((double)a.z + (double)a.w);
            }
            const int64_t t = lo + 4 * n4 + lane;
            if (t < hi) s += (double)vals[t];
        } else {
            int64_t p = lo + lane;
            for (; p + 3 * kWave < hi; p += 4 * kWave) {
                T a = vals[p], b = vals[p + kWave], c = vals[p + 2 * kWave], d = vals[p + 3 * kWave];
                s += (double)a; s += (double)b; s += (double)c; s += (double)d;
            }
            for (; p < hi; p += kWave) s += (double)vals[p];
        }
        s = wave_sum(s);
        if (lane == 0) out[r] = s;
    }
}

// compute_variance(Row): csr.rs:158-171 — mean = sum/cnt, sum((v-mean)^2)/cnt; 0/0 = NaN
// for an empty row (kept).
template <typename T>
__global__ __launch_bounds__(256) void k_row_var(const int64_t* __restrict__ indptr, const T* __restrict__ vals,
                                                 uint64_t n_rows, double* __restrict__ out) {
    const uint64_t wave = global_wave_id();
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) / kWave;
    const int lane = lane_id();
    for (uint64_t r = wave; r < n_rows; r += n_waves) {
        const int64_t lo = indptr[r], hi = indptr[r + 1];
        double s = 0.0;
        for (int64_t p = lo + lane; p < hi; p += kWave) s += (double)vals[p];
        s = wave_sum(s);
        const double cnt = (double)(uint32_t)(hi - lo);
        const double mean = s / cnt;
        double a = 0.0;
        for (int64_t p = lo + lane; p < hi; p += kWave) {
            double d = (double)vals[p] - mean;
            a += d * d;
        }
        a = wave_sum(a);
        if (lane == 0) out[r] = a / cnt;
    }
}

// compute_qc_variables, the per-cell half (statistics/mod.rs:48-72): number (csr.rs:24-27), sum (:87-93) and
// variance (:158-171: mean = sum/cnt, sum((v - mean)^2)/cnt, 0/0 = NaN for an empty row) of every row in ONE
// pass — the row stays in registers between the two reductions (rows longer than the cache re-read the tail).
template <typename T>
__global__ __launch_bounds__(256) void k_row_qc(const int64_t* __restrict__ indptr, const T* __restrict__ vals,
                                                uint64_t n_rows, uint32_t* __restrict__ num, double* __restrict__ sum,
                                                double* __restrict__ var) {
    constexpr int V = 16 / sizeof(T);
    constexpr int NV = kRowCache / V;
    using Vec = RowVec<T>;
    const uint64_t wave = global_wave_id();
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) / kWave;
    const int lane = lane_id();
    for (uint64_t r = wave; r < n_rows; r += n_waves) {
        const int64_t lo = indptr[r], hi = indptr[r + 1];
        const int64_t base = lo & ~(int64_t)(V - 1);
        const int64_t tail = base + (int64_t)NV * kWave * V;
        Vec c[NV];
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            const int64_t e0 = base + ((int64_t)t * kWave + lane) * V;
            if (e0 < hi) c[t] = *reinterpret_cast<const Vec*>(vals + e0);
            else {
#pragma unroll
                for (int j = 0; j < V; ++j) c[t].x[j] = T(0);
            }
        }
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            const int64_t e0 = base + ((int64_t)t * kWave + lane) * V;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const int64_t pos = e0 + j;
                s += (pos >= lo && pos < hi) ? (double)c[t].x[j] : 0.0;
            }
        }
        for (int64_t p = tail + lane; p < hi; p += kWave) s += (double)vals[p];
        s = wave_sum(s);
        const double cnt = (double)(uint32_t)(hi - lo);
        const double mean = s / cnt;
        double a = 0.0;
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            const int64_t e0 = base + ((int64_t)t * kWave + lane) * V;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const int64_t pos = e0 + j;
                const double d = (double)c[t].x[j] - mean;
                a += (pos >= lo && pos < hi) ? d * d : 0.0;
            }
        }
        for (int64_t p = tail + lane; p < hi; p += kWave) {
            const double d = (double)vals[p] - mean;
            a += d * d;
        }
        a = wave_sum(a);
        if (lane == 0) {
            num[r] = (uint32_t)(hi - lo);
            sum[r] = s;
            var[r] = a / cnt;
        }
    }
}

// compute_min_max(Row): csr.rs:200-210; f64::min/max skip NaN operands.
template <typename T>
__global__ __launch_bounds__(256) void k_row_minmax(const int64_t* __restrict__ indptr, const T* __restrict__ vals,
                                                    uint64_t n_rows, double* __restrict__ mn,
                                                    double* __restrict__ mx) {
    const uint64_t wave = global_wave_id();
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) / kWave;
    const int lane = lane_id();
    for (uint64_t r = wave; r < n_rows; r += n_waves) {
        const int64_t lo = indptr[r], hi = indptr[r + 1];
        double a = INFINITY, b = -INFINITY;
        for (int64_t p = lo + lane; p < hi; p += kWave) {
            double x = (double)vals[p];
            a = fmin(a, x);
            b = fmax(b, x);
        }
        a = wave_min(a);
        b = wave_max(b);
        if (lane == 0) { mn[r] = a; mx[r] = b; }
    }
}

// compute_number(Row): csr.rs:21-28.
__global__ void k_row_number(const int64_t* __restrict__ indptr, uint64_t n_rows, uint32_t* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n_rows; i += stride) out[i] = (uint32_t)(indptr[i + 1] - indptr[i]);
}

// In-place x <- ln_1p(x * target / row_sum) at the storage precision (the table-driven f64 logarithm, rounded once for f32
// storage: `xf_stored`) with the row sums the pipeline's first pass left in `row_sum`.  The resident pipeline no longer needs
// it — its moments pass stores the transformed values itself — it is the write-back of the SRX_WB_SIDE=1 order (the
// round-2 arrangement: on a side stream beside the iteration).  No reduction, so nothing makes a wave wait for a whole row: 16-byte loads, four
// in flight per lane, transformed and stored as they arrive.  (k_row_pass<T, true, true> re-sums every row first: 13 GB/s
// per CU at f32, its time proportional to the CUs it was given — 3.0 ms beside the iteration on 224 CUs.)
constexpr int kApplyUnroll = 4;
template <typename T>
__global__ __launch_bounds__(256) void k_row_apply(const int64_t* __restrict__ indptr, T* __restrict__ vals, uint64_t n_rows,
                                                   double target, const double* __restrict__ row_sum) {
    using Vec = RowVec<T>;
    constexpr int V = 16 / sizeof(T);
    __shared__ Log1pTabEntry s_tab[128];
    stage_log1p_table(s_tab);
    __syncthreads();
    const uint64_t wave = global_wave_id();
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) / kWave;
    const int lane = lane_id();
    for (uint64_t r = wave; r < n_rows; r += n_waves) {
        const int64_t lo = indptr[r], hi = indptr[r + 1];
        const double s = row_sum[r];
        const double scale = s == 0.0 ? 0.0 : target / s;           // scale/mod.rs:9-15
        for (int64_t b0 = (lo & ~(int64_t)(V - 1)) + V * lane; b0 < hi; b0 += (int64_t)kApplyUnroll * kWave * V) {
            Vec c[kApplyUnroll];
#pragma unroll
            for (int t = 0; t < kApplyUnroll; ++t) {
                const int64_t e0 = b0 + (int64_t)t * kWave * V;
                if (e0 < hi) c[t] = *reinterpret_cast<const Vec*>(vals + e0);       // the array is padded: a vector may start before lo or end past hi
            }
#pragma unroll
            for (int t = 0; t < kApplyUnroll; ++t) {
                const int64_t e0 = b0 + (int64_t)t * kWave * V;
                if (e0 >= hi) continue;
                Vec o;
#pragma unroll
                for (int j = 0; j < V; ++j) o.x[j] = xf_stored(c[t].x[j], scale, s_tab);
                if (e0 >= lo && e0 + V <= hi) {
                    *reinterpret_cast<Vec*>(vals + e0) = o;
                } else {
#pragma unroll
                    for (int j = 0; j < V; ++j)
                        if (e0 + j >= lo && e0 + j < hi) vals[e0 + j] = o.x[j];
                }
            }
        }
    }
}

static int row_grid(const srx_mat* m) {
    uint64_t want = (m->n_rows + 3) / 4;  // 4 waves per 256-thread block, one row per wave
    uint64_t cap = (uint64_t)m->ctx->n_cus * 8;      // 4 .. 32 blocks per CU all give 2.26-2.30 ms at c3
    if (want < 1) want = 1;
    return (int)(want < cap ? want : cap);
}

// `stream` / `precise`: the pipeline's write-back (k_row_pass<.., PRECISE>) runs on the context's side stream
int32_t launch_normalize(srx_mat* m, double target, bool do_norm, bool do_log, hipStream_t stream, bool precise, int wgs_per_cu) {
    srx_ctx* ctx = m->ctx;
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    if (!stream) stream = ctx->stream;
    if (do_norm && !m->d_row_sum) SRX_HIP(ctx, dev_malloc(ctx, (void**)&m->d_row_sum, (m->n_rows ? m->n_rows : 1) * sizeof(double)));
    int g = row_grid(m);
    if (wgs_per_cu > 0 && g > ctx->n_cus * wgs_per_cu) g = ctx->n_cus * wgs_per_cu;
    const double bytes = (double)m->nnz * 2.0 * val_bytes(m) + (double)(m->n_rows + 1) * 8.0;
    {
        ProfScope ps(ctx, SRX_K_NORMALIZE, bytes, stream);
#define SRX_LAUNCH_ROW(T, N, L, P)                                                                  \
    hipLaunchKernelGGL((k_row_pass<T, N, L, P>), dim3(g), dim3(256), 0, stream, m->d_indptr,         \
                       (T*)m->d_values, m->n_rows, target, (N) ? m->d_row_sum : (double*)nullptr)
        if (is_f32(m)) {
            if (do_norm && do_log && precise) SRX_LAUNCH_ROW(float, true, true, true);
            else if (do_norm && do_log) SRX_LAUNCH_ROW(float, true, true, false);
            else if (do_norm) SRX_LAUNCH_ROW(float, true, false, false);
            else if (do_log) SRX_LAUNCH_ROW(float, false, true, false);
        } else {
            if (do_norm && do_log && precise) SRX_LAUNCH_ROW(double, true, true, true);
            else if (do_norm && do_log) SRX_LAUNCH_ROW(double, true, true, false);
            else if (do_norm) SRX_LAUNCH_ROW(double, true, false, false);
            else if (do_log) SRX_LAUNCH_ROW(double, false, true, false);
        }
#undef SRX_LAUNCH_ROW
    }
    SRX_HIP(ctx, hipGetLastError());
    // logical dtype bookkeeping: scale/mod.rs:74-83 (-> F64), transform/mod.rs:43-55
    if (do_norm) m->dtype = SRX_F64;
    if (do_log && m->dtype != SRX_F32) m->dtype = SRX_F64;
    touch(m);
    return SRX_OK;
}

int32_t launch_row_apply(srx_mat* m, double target, hipStream_t stream) {
    srx_ctx* ctx = m->ctx;
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    if (!m->d_row_sum) return fail(ctx, SRX_E_ARG, "row write-back without row sums");
    if (!stream) stream = ctx->stream;
    {
        ProfScope ps(ctx, SRX_K_NORMALIZE, (double)m->nnz * 2.0 * val_bytes(m) + (double)(m->n_rows + 1) * 8.0 + (double)m->n_rows * 8.0, stream);
        if (is_f32(m))
            hipLaunchKernelGGL((k_row_apply<float>), dim3(row_grid(m)), dim3(256), 0, stream, m->d_indptr, (float*)m->d_values,
                               m->n_rows, target, m->d_row_sum);
        else
            hipLaunchKernelGGL((k_row_apply<double>), dim3(row_grid(m)), dim3(256), 0, stream, m->d_indptr, (double*)m->d_values,
                               m->n_rows, target, m->d_row_sum);
    }
    SRX_HIP(ctx, hipGetLastError());
    m->dtype = SRX_F64;       // logical dtype bookkeeping as in launch_normalize: scale/mod.rs:74-83 (-> F64), transform/mod.rs:43-55
    touch(m);
    return SRX_OK;
}

// Row sums of the current values into m->d_row_sum (the first step of the pipeline: the per-gene passes apply the
// normalise + log1p transform on the fly from them, the in-place write-back follows on the side stream).
int32_t launch_row_sums(srx_mat* m) {
    srx_ctx* ctx = m->ctx;
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    if (!m->d_row_sum) SRX_HIP(ctx, dev_malloc(ctx, (void**)&m->d_row_sum, (m->n_rows ? m->n_rows : 1) * sizeof(double)));
    const int g = row_grid(m);
    ProfScope ps(ctx, SRX_K_ROWSUM, (double)m->nnz * val_bytes(m) + (double)(m->n_rows + 1) * 8.0 + (double)m->n_rows * 8.0);
    if (is_f32(m))
        hipLaunchKernelGGL((k_row_sum<float>), dim3(g), dim3(256), 0, ctx->stream, m->d_indptr, (const float*)m->d_values,
                           m->n_rows, m->d_row_sum);
    else
        hipLaunchKernelGGL((k_row_sum<double>), dim3(g), dim3(256), 0, ctx->stream, m->d_indptr, (const double*)m->d_values,
                           m->n_rows, m->d_row_sum);
    SRX_HIP(ctx, hipGetLastError());
    return SRX_OK;
}

// per-cell (number, sum, variance) of compute_qc_variables in one pass; any output may be null
int32_t row_qc(srx_mat* m, uint32_t* num, double* sum, double* var) {
    srx_ctx* ctx = m->ctx;
    const uint64_t N = m->n_rows ? m->n_rows : 1;
    uint32_t* d_num;
    double* d_f;
    SRX_TRY(scratch(ctx, "row_u32", N * sizeof(uint32_t), (void**)&d_num));
    SRX_TRY(scratch(ctx, "row_f64", 2 * N * sizeof(double), (void**)&d_f));
    const int g = row_grid(m);
    if (is_f32(m))
        hipLaunchKernelGGL((k_row_qc<float>), dim3(g), dim3(256), 0, ctx->stream, m->d_indptr, (const float*)m->d_values,
                           m->n_rows, d_num, d_f, d_f + N);
    else
        hipLaunchKernelGGL((k_row_qc<double>), dim3(g), dim3(256), 0, ctx->stream, m->d_indptr, (const double*)m->d_values,
                           m->n_rows, d_num, d_f, d_f + N);
    SRX_HIP(ctx, hipGetLastError());
    if (num) SRX_TRY(d2h(ctx, num, d_num, m->n_rows * sizeof(uint32_t)));
    if (sum) SRX_TRY(d2h(ctx, sum, d_f, m->n_rows * sizeof(double)));
    if (var) SRX_TRY(d2h(ctx, var, d_f + N, m->n_rows * sizeof(double)));
    return SRX_OK;
}

}  // namespace srx

using namespace srx;

extern "C" {

int32_t srx_log1p_inplace(srx_mat* m) {
    if (!m) return fail(nullptr, SRX_E_ARG, "null matrix");
    if (m->dtype != SRX_F32) SRX_TRY(promote_to_f64(m));      // transform/mod.rs:43-55: F32 stays F32, anything else -> F64
    return launch_normalize(m, 0.0, false, true);
}

int32_t srx_normalize_log1p_inplace(srx_mat* m, double target_sum, double* row_sums_out) {
    if (!m) return fail(nullptr, SRX_E_ARG, "null matrix");
    if (m->csc) {                                // cells are the stored columns: the two reference calls, one after the other
        if (row_sums_out) SRX_TRY(srx_compute_sum(m, SRX_ROW, row_sums_out));
        SRX_TRY(srx_normalize_total_inplace(m, target_sum, SRX_ROW));
        return srx_log1p_inplace(m);
    }
    SRX_TRY(promote_to_f64(m));
    SRX_TRY(launch_normalize(m, target_sum, true, true));
    if (row_sums_out) SRX_TRY(d2h(m->ctx, row_sums_out, m->d_row_sum, m->n_rows * sizeof(double)));
    return SRX_OK;
}

}  // extern "C"

namespace srx {

// Row-direction halves of the statistics entry points (genes.hip holds the Column halves and
// the extern "C" wrappers).
int32_t row_number(srx_mat* m, uint32_t* out) {
    srx_ctx* ctx = m->ctx;
    uint32_t* d;
    SRX_TRY(scratch(ctx, "row_u32", (m->n_rows ? m->n_rows : 1) * sizeof(uint32_t), (void**)&d));
    uint64_t g = (m->n_rows + 255) / 256;
    if (g < 1) g = 1;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_row_number, dim3((int)g), dim3(256), 0, ctx->stream, m->d_indptr, m->n_rows, d);
    SRX_HIP(ctx, hipGetLastError());
    return d2h(ctx, out, d, m->n_rows * sizeof(uint32_t));
}

int32_t row_stat(srx_mat* m, int which, double* out0, double* out1) {
    srx_ctx* ctx = m->ctx;
    double* d;
    SRX_TRY(scratch(ctx, "row_f64", 2 * (m->n_rows ? m->n_rows : 1) * sizeof(double), (void**)&d));
    double* d1 = d + m->n_rows;
    const int g = row_grid(m);
#define SRX_ROW_K(K, ...)                                                                          \
    do {                                                                                           \
        if (is_f32(m))                                                                             \
            hipLaunchKernelGGL((K<float>), dim3(g), dim3(256), 0, ctx->stream, m->d_indptr,         \
                               (const float*)m->d_values, m->n_rows, __VA_ARGS__);                  \
        else                                                                                       \
            hipLaunchKernelGGL((K<double>), dim3(g), dim3(256), 0, ctx->stream, m->d_indptr,        \
                               (const double*)m->d_values, m->n_rows, __VA_ARGS__);                 \
    } while (0)
    if (which == 0) SRX_ROW_K(k_row_sum, d);
    else if (which == 1) SRX_ROW_K(k_row_var, d);
    else SRX_ROW_K(k_row_minmax, d, d1);
#undef SRX_ROW_K
    SRX_HIP(ctx, hipGetLastError());
    SRX_TRY(d2h(ctx, out0, d, m->n_rows * sizeof(double)));
    if (which == 2) SRX_TRY(d2h(ctx, out1, d1, m->n_rows * sizeof(double)));
    return SRX_OK;
}

}  // namespace srx
