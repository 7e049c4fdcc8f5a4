// synth.hip — synthetic count matrices generated in HBM (see include/srx_synth.h).
// Every entry is a pure integer function of (seed, global row, slot), shared between the
// device kernel and the host reference generator, so both are bit-identical.
#include <cmath>

#include "../../include/srx_synth.h"
#include "common.hpp"

namespace srx {

struct SynthConst {
    uint64_t seed;
    uint64_t G, Gv;          // real / virtual gene axis length
    uint32_t T, M, B, VB, skew;
    uint32_t thr[SRX_SYNTH_MAX_TYPES];   // cumulative type thresholds on a 32-bit uniform
};

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__host__ __device__ __forceinline__ uint64_t hash2(uint64_t seed, uint64_t i, uint64_t s) {
    return mix64(mix64(seed ^ mix64(i)) + s);
}
constexpr uint64_t kTagType = (1ull << 40) + 0, kTagEmpty = (1ull << 40) + 1, kTagZ1 = (1ull << 40) + 2,
                   kTagZ2 = (1ull << 40) + 3;

__host__ __device__ __forceinline__ uint32_t row_type(const SynthConst& c, uint64_t row) {
    uint32_t u = (uint32_t)(hash2(c.seed, row, kTagType) >> 32);
    uint32_t t = 0;
    while (t + 1 < c.T && u > c.thr[t]) ++t;
    return t;
}

// virtual position -> real gene for a cell of type t (its marker genes occupy B slots each)
__host__ __device__ __forceinline__ uint64_t vmap(const SynthConst& c, uint32_t t, uint64_t p) {
    uint64_t lo = (uint64_t)t * c.M;
    if (p < lo) return p;
    if (p < lo + (uint64_t)c.B * c.M) return lo + (p - lo) / c.B;
    return p - (uint64_t)(c.B - 1) * c.M;
}

__host__ __device__ __forceinline__ void synth_entry(const SynthConst& c, uint64_t row, uint32_t t, uint64_t r,
                                                     uint64_t s, uint64_t& col, uint32_t& val) {
    uint64_t b0 = (s * c.Gv) / r, b1 = ((s + 1) * c.Gv) / r;
    if (c.skew) {
        // quadratic strata: (Gv - r B) (s/r)^2 + s B — strictly increasing, every stratum >= B virtual slots wide (so it
        // holds a whole gene after vmap, like the uniform strata), the last one ends at Gv
        const uint64_t span = c.Gv - r * c.B;
        b0 = (span * s * s) / (r * r) + s * c.B;
        b1 = (span * (s + 1) * (s + 1)) / (r * r) + (s + 1) * c.B;
    }
    uint64_t g0 = vmap(c, t, b0);
    uint64_t g1 = vmap(c, t, b1);
    uint64_t h = hash2(c.seed, row, s);
    col = g0 + (h >> 32) % (g1 - g0);
    uint32_t low = (uint32_t)(h & 0xFFFFu) | 0x8000u;
    uint32_t tz = 0;
    while (!(low & 1u)) { low >>= 1; ++tz; }       // Geometric(1/2), capped at 15
    uint32_t v = 1 + tz;
    uint64_t lo = (uint64_t)t * c.M;
    if (c.T > 1 && col >= lo && col < lo + c.M) v *= c.VB;
    val = v;
}

static int32_t make_const(const srx_synth_params* p, SynthConst& c) {
    if (!p) return fail(nullptr, SRX_E_ARG, "null synth params");
    if (p->n_types < 1 || p->n_types > SRX_SYNTH_MAX_TYPES || p->expr_boost < 1 || p->value_boost < 1)
        return fail(nullptr, SRX_E_ARG, "synth: bad type/boost parameters");
    if ((uint64_t)p->n_types * p->marker_genes > p->n_cols)
        return fail(nullptr, SRX_E_ARG, "synth: n_types*marker_genes exceeds n_cols");
    c.seed = p->seed;
    c.G = p->n_cols;
    c.T = p->n_types;
    c.M = p->n_types > 1 ? p->marker_genes : 0;
    c.B = p->n_types > 1 ? p->expr_boost : 1;
    c.VB = p->value_boost;
    c.skew = p->skew ? 1u : 0u;
    c.Gv = c.G + (uint64_t)(c.B - 1) * c.M;
    double tot = 0.0, w = 1.0;
    for (uint32_t t = 0; t < c.T; ++t) { tot += w; w *= p->type_decay; }
    double acc = 0.0;
    w = 1.0;
    for (uint32_t t = 0; t < SRX_SYNTH_MAX_TYPES; ++t) {
        if (t < c.T) {
            acc += w;
            w *= p->type_decay;
            double x = acc / tot * 4294967296.0;
            c.thr[t] = x >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)x;
        } else {
            c.thr[t] = 0xFFFFFFFFu;
        }
    }
    c.thr[c.T - 1] = 0xFFFFFFFFu;
    return SRX_OK;
}

static uint64_t row_nnz(const srx_synth_params* p, const SynthConst& c, uint64_t row) {
    if (hash2(c.seed, row, kTagEmpty) % 10000ull == 0) return 0;
    double u1 = ((double)(hash2(c.seed, row, kTagZ1) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    double u2 = ((double)(hash2(c.seed, row, kTagZ2) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    double z = std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    double s = p->lib_sigma;
    double r = std::rint(p->density * (double)c.G * std::exp(s * z - 0.5 * s * s));
    double rmax = (double)(c.Gv / c.B);          // keeps every stratum non-empty after vmap
    if (r < 0.0) r = 0.0;
    if (r > rmax) r = rmax;
    return (uint64_t)r;
}

template <typename T>
__global__ __launch_bounds__(256) void k_synth_fill(SynthConst c, uint64_t row_begin, uint64_t n_rows,
                                                    const int64_t* __restrict__ indptr, int32_t* __restrict__ idx,
                                                    T* __restrict__ vals) {
    const uint64_t wave = global_wave_id();
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) / kWave;
    const int lane = lane_id();
    for (uint64_t lr = wave; lr < n_rows; lr += n_waves) {
        const uint64_t row = row_begin + lr;
        const int64_t lo = indptr[lr];
        const uint64_t r = (uint64_t)(indptr[lr + 1] - lo);
        if (r == 0) continue;
        const uint32_t t = row_type(c, row);
        for (uint64_t s = lane; s < r; s += kWave) {
            uint64_t col;
            uint32_t v;
            synth_entry(c, row, t, r, s, col, v);
            idx[lo + s] = (int32_t)col;
            vals[lo + s] = (T)v;
        }
    }
}

}  // namespace srx

using namespace srx;

extern "C" {

void srx_synth_defaults(srx_synth_params* p, uint64_t seed, uint64_t n_rows_global, uint64_t n_cols,
                        double density) {
    if (!p) return;
    p->seed = seed;
    p->n_rows_global = n_rows_global;
    p->n_cols = n_cols;
    p->density = density;
    p->lib_sigma = 0.3;
    p->type_decay = 0.98;
    p->n_types = 52;
    p->marker_genes = 40;
    p->expr_boost = 12;
    p->value_boost = 8;
    p->skew = 0;
    p->reserved_ = 0;
    // small matrices: shrink the planted structure to fit
    while ((uint64_t)p->n_types * p->marker_genes * 4 > n_cols && p->marker_genes > 4) p->marker_genes /= 2;
    while ((uint64_t)p->n_types * p->marker_genes * 4 > n_cols && p->n_types > 2) p->n_types /= 2;
}

int32_t srx_synth_indptr(const srx_synth_params* p, uint64_t row_begin, uint64_t row_end, uint64_t* indptr_out) {
    SynthConst c;
    SRX_TRY(make_const(p, c));
    if (!indptr_out || row_end < row_begin) return fail(nullptr, SRX_E_ARG, "synth: bad row range");
    uint64_t acc = 0;
    indptr_out[0] = 0;
    for (uint64_t i = row_begin; i < row_end; ++i) {
        acc += row_nnz(p, c, i);
        indptr_out[i - row_begin + 1] = acc;
    }
    return SRX_OK;
}

int32_t srx_synth_fill_host(const srx_synth_params* p, uint64_t row_begin, uint64_t row_end, const uint64_t* indptr,
                            uint64_t* indices_out, float* values_out) {
    SynthConst c;
    SRX_TRY(make_const(p, c));
    if (!indptr || !indices_out || !values_out) return fail(nullptr, SRX_E_ARG, "synth: null buffer");
    for (uint64_t i = row_begin; i < row_end; ++i) {
        uint64_t lo = indptr[i - row_begin], r = indptr[i - row_begin + 1] - lo;
        uint32_t t = row_type(c, i);
        for (uint64_t s = 0; s < r; ++s) {
            uint64_t col;
            uint32_t v;
            synth_entry(c, i, t, r, s, col, v);
            indices_out[lo + s] = col;
            values_out[lo + s] = (float)v;
        }
    }
    return SRX_OK;
}

int32_t srx_synth_generate(srx_ctx* ctx, const srx_synth_params* p, uint64_t row_begin, uint64_t row_end,
                           int32_t dtype, int32_t store, srx_mat** out) {
    if (!ctx || !out) return fail(ctx, SRX_E_ARG, "synth: null argument");
    *out = nullptr;
    SynthConst c;
    SRX_TRY(make_const(p, c));
    if (row_end < row_begin || row_end > p->n_rows_global) return fail(ctx, SRX_E_ARG, "synth: bad row range");
    const uint64_t n = row_end - row_begin;
    std::vector<uint64_t> indptr(n + 1);
    SRX_TRY(srx_synth_indptr(p, row_begin, row_end, indptr.data()));
    srx_mat* m = nullptr;
    SRX_TRY(srx_matrix_alloc(ctx, n, p->n_cols, indptr[n], dtype, store, &m));
    m->row_offset = row_begin;
    hipError_t e = hipMemcpy(m->d_indptr, indptr.data(), (n + 1) * sizeof(int64_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        srx_matrix_free(m);
        return fail(ctx, SRX_E_HIP, "synth: H2D indptr: %s", hipGetErrorString(e));
    }
    uint64_t g = (n + 3) / 4;
    if (g < 1) g = 1;
    if (g > (uint64_t)ctx->n_cus * 8) g = (uint64_t)ctx->n_cus * 8;
    if (m->store == SRX_STORE_F32)
        hipLaunchKernelGGL((k_synth_fill<float>), dim3((unsigned)g), dim3(256), 0, ctx->stream, c, row_begin, n,
                           m->d_indptr, m->d_indices, (float*)m->d_values);
    else
        hipLaunchKernelGGL((k_synth_fill<double>), dim3((unsigned)g), dim3(256), 0, ctx->stream, c, row_begin, n,
                           m->d_indptr, m->d_indices, (double*)m->d_values);
    e = hipGetLastError();
    // the handle leaves in the state srx_matrix_upload leaves one in: the 16-bit index mirror and the gene-tile cuts are made where
    // the indices are written (the upload's H2D workers narrow on the host side of the link and tiles_from_idx16 cuts the tiles
    // before the upload returns) — not by the first statistics call on the handle
    int32_t rc = e == hipSuccess ? ensure_tiles(m) : SRX_OK;
    if (e == hipSuccess && rc == SRX_OK) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess || rc != SRX_OK) {
        srx_matrix_free(m);
        return e != hipSuccess ? fail(ctx, SRX_E_HIP, "synth fill kernel: %s", hipGetErrorString(e)) : rc;
    }
    *out = m;
    return SRX_OK;
}

}  // extern "C"
