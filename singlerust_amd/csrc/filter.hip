// filter.hip — QC filters and CSR subsetting on the device (the step right BEFORE the
// hot path).  filter_cells / filter_genes of the reference (src/memory/processing/mod.rs:86-146, :245-299):
// counts / sums of the rows or columns (already device passes: indptr differences, k_row_sum, the gene moments)
// -> thresholds (Absolute: on the nnz counts; Relative: linear-interpolated quantiles of the SUMS,
// mod.rs:148-174 — ndarray-stats 0.5.1 `interpolate::Linear`: lower + (higher - lower) * fract(p (n - 1)))
// -> boolean mask (the nine FlexValue combinations of mod.rs:33-84) -> subset.  The masks are N or G booleans
// and the two order statistics are an nth_element: host work on vectors the statistics calls return anyway.
// The subset (anndata `subset`, third-party in the reference) is a stream compaction on the device: kept
// entries per kept row (wave ballot), exclusive scan, fill; rows / columns keep their order.
#include "common.hpp"

namespace srx {

int32_t row_number(srx_mat* m, uint32_t* out);
int32_t row_stat(srx_mat* m, int which, double* out0, double* out1);
int32_t scan_exclusive(srx_ctx* ctx, const int64_t* d_in, uint64_t n, int64_t* d_out, int64_t** total_dev);

// kept entries of every OLD row (0 for dropped rows); colmap == nullptr keeps every column
__global__ __launch_bounds__(256) void k_subset_count(const int64_t* __restrict__ indptr, const int32_t* __restrict__ idx,
                                                      const uint8_t* __restrict__ rowmask,
                                                      const int32_t* __restrict__ colmap, uint64_t n_rows,
                                                      int64_t* __restrict__ counts) {
    const uint64_t wave = global_wave_id();
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) / kWave;
    const int lane = lane_id();
    for (uint64_t r = wave; r < n_rows; r += n_waves) {
        int64_t c = 0;
        if (!rowmask || rowmask[r]) {
            const int64_t lo = indptr[r], hi = indptr[r + 1];
            if (!colmap) {
                c = hi - lo;
            } else {
                int cc = 0;
                for (int64_t p = lo + lane; p < hi; p += kWave) cc += colmap[idx[p]] >= 0;
                c = wave_sum(cc);
            }
        }
        if (lane == 0) counts[r] = c;
    }
}

// new row id of every kept old row = exclusive scan of the row mask (as int64 so that scan_exclusive serves)
__global__ void k_mask_to_i64(const uint8_t* __restrict__ mask, uint64_t n, int64_t* __restrict__ out) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; e < n; e += stride) out[e] = mask ? (mask[e] ? 1 : 0) : 1;
}

template <typename T>
__global__ __launch_bounds__(256) void k_subset_fill(const int64_t* __restrict__ indptr, const int32_t* __restrict__ idx,
                                                     const T* __restrict__ vals, const uint8_t* __restrict__ rowmask,
                                                     const int32_t* __restrict__ colmap, uint64_t n_rows,
                                                     const int64_t* __restrict__ off_old /* n_rows + 1 */,
                                                     const int64_t* __restrict__ new_id /* n_rows + 1 */,
                                                     int64_t* __restrict__ out_ptr, int32_t* __restrict__ out_idx,
                                                     T* __restrict__ out_vals) {
    const uint64_t wave = global_wave_id();
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) / kWave;
    const int lane = lane_id();
    if (wave == 0 && lane == 0) out_ptr[new_id[n_rows]] = off_old[n_rows];       // the closing pointer
    for (uint64_t r = wave; r < n_rows; r += n_waves) {
        if (rowmask && !rowmask[r]) continue;
        const int64_t lo = indptr[r], hi = indptr[r + 1];
        int64_t o = off_old[r];
        if (lane == 0) out_ptr[new_id[r]] = o;
        for (int64_t base = lo; base < hi; base += kWave) {
            const int64_t p = base + lane;
            int32_t c = -1;
            if (p < hi) c = colmap ? colmap[idx[p]] : idx[p];
            const unsigned long long m = __ballot(c >= 0);
            if (c >= 0) {
                const int pos = __popcll(m & ((1ull << lane) - 1ull));
                out_idx[o + pos] = c;
                out_vals[o + pos] = vals[p];
            }
            o += __popcll(m);
        }
    }
}

static int32_t subset_device(srx_mat* m, const uint8_t* row_mask, const uint8_t* col_mask, srx_mat** out) {
    srx_ctx* ctx = m->ctx;
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t N = m->n_rows, G = m->n_cols;
    uint64_t n_keep_rows = N, n_keep_cols = G;
    uint8_t* d_rowmask = nullptr;
    int32_t* d_colmap = nullptr;
    if (row_mask) {
        n_keep_rows = 0;
        for (uint64_t i = 0; i < N; ++i) n_keep_rows += row_mask[i] ? 1 : 0;
        SRX_TRY(scratch(ctx, "sub_rowmask", N ? N : 1, (void**)&d_rowmask));
        SRX_TRY(h2d(ctx, d_rowmask, row_mask, N));
    }
    if (col_mask) {
        std::vector<int32_t> colmap(G, -1);
        n_keep_cols = 0;
        for (uint64_t j = 0; j < G; ++j)
            if (col_mask[j]) colmap[j] = (int32_t)n_keep_cols++;
        SRX_TRY(scratch(ctx, "sub_colmap", (G ? G : 1) * sizeof(int32_t), (void**)&d_colmap));
        SRX_TRY(h2d(ctx, d_colmap, colmap.data(), G * sizeof(int32_t)));
    }
    int64_t *d_counts, *d_off, *d_flag, *d_newid, *d_total;
    SRX_TRY(scratch(ctx, "sub_counts", (N ? N : 1) * sizeof(int64_t), (void**)&d_counts));
    SRX_TRY(scratch(ctx, "sub_off", (N + 1) * sizeof(int64_t), (void**)&d_off));
    SRX_TRY(scratch(ctx, "sub_flag", (N ? N : 1) * sizeof(int64_t), (void**)&d_flag));
    SRX_TRY(scratch(ctx, "sub_newid", (N + 1) * sizeof(int64_t), (void**)&d_newid));
    uint64_t gw = (N + 3) / 4;
    if (gw < 1) gw = 1;
    if (gw > (uint64_t)ctx->n_cus * 8) gw = (uint64_t)ctx->n_cus * 8;
    hipLaunchKernelGGL(k_subset_count, dim3((unsigned)gw), dim3(256), 0, ctx->stream, m->d_indptr, m->d_indices, d_rowmask,
                       d_colmap, N, d_counts);
    uint64_t ge = (N + 255) / 256;
    if (ge < 1) ge = 1;
    if (ge > 65535) ge = 65535;
    hipLaunchKernelGGL(k_mask_to_i64, dim3((unsigned)ge), dim3(256), 0, ctx->stream, d_rowmask, N, d_flag);
    SRX_HIP(ctx, hipGetLastError());
    SRX_TRY(scan_exclusive(ctx, d_counts, N, d_off, &d_total));
    int64_t total = 0;
    SRX_TRY(d2h(ctx, &total, d_total, sizeof(int64_t)));
    SRX_TRY(scan_exclusive(ctx, d_flag, N, d_newid, nullptr));
    srx_mat* c = nullptr;
    SRX_TRY(srx_matrix_alloc(ctx, n_keep_rows, n_keep_cols, (uint64_t)total, m->dtype, m->store, &c));
    c->row_offset = m->row_offset;
    if (m->store == SRX_STORE_F32)
        hipLaunchKernelGGL((k_subset_fill<float>), dim3((unsigned)gw), dim3(256), 0, ctx->stream, m->d_indptr, m->d_indices,
                           (const float*)m->d_values, d_rowmask, d_colmap, N, d_off, d_newid, c->d_indptr, c->d_indices,
                           (float*)c->d_values);
    else
        hipLaunchKernelGGL((k_subset_fill<double>), dim3((unsigned)gw), dim3(256), 0, ctx->stream, m->d_indptr, m->d_indices,
                           (const double*)m->d_values, d_rowmask, d_colmap, N, d_off, d_newid, c->d_indptr, c->d_indices,
                           (double*)c->d_values);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        srx_matrix_free(c);
        return fail(ctx, SRX_E_HIP, "subset: %s", hipGetErrorString(e));
    }
    *out = c;
    return SRX_OK;
}

// ndarray-stats 0.5.1 quantile_axis_mut(.., q, &Linear): index = q (n - 1); lower + (higher - lower) * fract(index)
static int32_t quantile_linear(srx_ctx* ctx, std::vector<double> v, double q, double* out) {
    const size_t n = v.size();
    if (n == 0) return fail(ctx, SRX_E_ARG, "Error calculating percentile: empty input");
    if (!(q >= 0.0 && q <= 1.0)) return fail(ctx, SRX_E_ARG, "Error calculating percentile: quantile %g outside [0, 1]", q);
    // the reference wraps the sums in noisy_float's n64(), which panics on NaN (mod.rs:152-158); a NaN would also break
    // the strict weak ordering nth_element relies on
    for (double x : v)
        if (x != x) return fail(ctx, SRX_E_NAN, "Error calculating percentile: NaN among the sums (n64() panics in the reference)");
    const double idx = q * (double)(n - 1);
    const size_t lo = (size_t)std::floor(idx), hi = (size_t)std::ceil(idx);
    std::nth_element(v.begin(), v.begin() + lo, v.end());
    const double lower = v[lo];
    double higher = lower;
    if (hi != lo) higher = *std::min_element(v.begin() + lo + 1, v.end());      // the next order statistic
    *out = lower + (higher - lower) * (idx - std::floor(idx));
    return SRX_OK;
}

static int32_t filter_mask(srx_ctx* ctx, uint64_t n, const std::vector<uint32_t>* counts, const std::vector<double>& sums,
                           srx_flex lower, srx_flex upper, std::vector<uint8_t>& mask) {
    if (lower.kind < SRX_FLEX_NONE || lower.kind > SRX_FLEX_RELATIVE || upper.kind < SRX_FLEX_NONE ||
        upper.kind > SRX_FLEX_RELATIVE)
        return fail(ctx, SRX_E_ARG, "bad FlexValue kind");
    // calculate_percentiles (mod.rs:148-174): f64::MIN / f64::MAX when the limit is not Relative
    double lp = -std::numeric_limits<double>::max(), up = std::numeric_limits<double>::max();
    // a quantile is a property of ALL cells / genes: on a row shard it would be the shard's own, a different threshold on
    // every rank
    if (ctx->n_ranks > 1 && (lower.kind == SRX_FLEX_RELATIVE || upper.kind == SRX_FLEX_RELATIVE))
        return fail(ctx, SRX_E_ARG, "FlexValue::Relative needs the sums of the whole matrix: filter before sharding the rows "
                                    "(or use Absolute limits) — this context holds rank %d of %d", ctx->rank, ctx->n_ranks);
    if (lower.kind == SRX_FLEX_RELATIVE) SRX_TRY(quantile_linear(ctx, sums, lower.relative, &lp));
    if (upper.kind == SRX_FLEX_RELATIVE) SRX_TRY(quantile_linear(ctx, sums, upper.relative, &up));
    mask.assign(n, 1);
    for (uint64_t i = 0; i < n; ++i) {                                  // create_filter_mask (mod.rs:33-84)
        bool ok = true;
        if (lower.kind == SRX_FLEX_ABSOLUTE) ok = ok && (*counts)[i] >= lower.absolute;
        else if (lower.kind == SRX_FLEX_RELATIVE) ok = ok && sums[i] >= lp;
        if (upper.kind == SRX_FLEX_ABSOLUTE) ok = ok && (*counts)[i] <= upper.absolute;
        else if (upper.kind == SRX_FLEX_RELATIVE) ok = ok && sums[i] <= up;
        mask[i] = ok ? 1 : 0;
    }
    return SRX_OK;
}

}  // namespace srx

using namespace srx;

extern "C" {

// masks in the STORED orientation; the result keeps the storage format of the input
static int32_t subset_stored(srx_mat* m, const uint8_t* row_mask, const uint8_t* col_mask, srx_mat** out) {
    SRX_TRY(subset_device(m, row_mask, col_mask, out));
    (*out)->csc = m->csc;
    return SRX_OK;
}

int32_t srx_subset(srx_mat* m, const uint8_t* row_mask, const uint8_t* col_mask, srx_mat** out) {
    if (!m || !out) return fail(m ? m->ctx : nullptr, SRX_E_ARG, "null argument");
    *out = nullptr;
    return m->csc ? subset_stored(m, col_mask, row_mask, out) : subset_stored(m, row_mask, col_mask, out);
}

// filter along the stored rows / the stored columns; cells are the rows of a CSR matrix and the columns of a CSC one
static int32_t filter_stored_rows(srx_mat* m, srx_flex lower, srx_flex upper, srx_mat** out, uint8_t* mask_out) {
    srx_ctx* ctx = m->ctx;
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    const bool need_count = lower.kind == SRX_FLEX_ABSOLUTE || upper.kind == SRX_FLEX_ABSOLUTE;      // mod.rs:91-92
    std::vector<uint32_t> counts;
    std::vector<double> sums(m->n_rows);
    if (need_count) {
        counts.resize(m->n_rows);
        SRX_TRY(row_number(m, counts.data()));
    }
    SRX_TRY(row_stat(m, 0, sums.data(), nullptr));
    std::vector<uint8_t> mask;
    SRX_TRY(filter_mask(ctx, m->n_rows, need_count ? &counts : nullptr, sums, lower, upper, mask));
    if (mask_out) memcpy(mask_out, mask.data(), mask.size());
    return subset_stored(m, mask.data(), nullptr, out);
}

static int32_t filter_stored_cols(srx_mat* m, srx_flex lower, srx_flex upper, srx_mat** out, uint8_t* mask_out) {
    srx_ctx* ctx = m->ctx;
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t G = m->n_cols;
    std::vector<uint32_t> counts(G);
    std::vector<double> sums(G);
    // one column pass (the cached moments): counts AND sums; eff_dir() maps the stored Column to the caller's direction
    const int32_t col = m->csc ? SRX_ROW : SRX_COLUMN;
    SRX_TRY(srx_compute_number(m, col, counts.data()));
    SRX_TRY(srx_compute_sum(m, col, sums.data()));
    std::vector<uint8_t> mask;
    SRX_TRY(filter_mask(ctx, G, &counts, sums, lower, upper, mask));
    if (mask_out) memcpy(mask_out, mask.data(), mask.size());
    return subset_stored(m, nullptr, mask.data(), out);
}

int32_t srx_filter_cells(srx_mat* m, srx_flex lower, srx_flex upper, srx_mat** out, uint8_t* mask_out) {
    if (!m || !out) return fail(m ? m->ctx : nullptr, SRX_E_ARG, "null argument");
    *out = nullptr;
    return m->csc ? filter_stored_cols(m, lower, upper, out, mask_out) : filter_stored_rows(m, lower, upper, out, mask_out);
}

int32_t srx_filter_genes(srx_mat* m, srx_flex lower, srx_flex upper, srx_mat** out, uint8_t* mask_out) {
    if (!m || !out) return fail(m ? m->ctx : nullptr, SRX_E_ARG, "null argument");
    *out = nullptr;
    return m->csc ? filter_stored_rows(m, lower, upper, out, mask_out) : filter_stored_cols(m, lower, upper, out, mask_out);
}

int32_t srx_matrix_download_pattern(srx_mat* m, uint64_t* indptr_out, uint64_t* indices_out) {
    if (!m) return fail(nullptr, SRX_E_ARG, "null matrix");
    srx_ctx* ctx = m->ctx;
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    if (indptr_out) {
        std::vector<int64_t> ip(m->n_rows + 1);
        SRX_TRY(d2h(ctx, ip.data(), m->d_indptr, ip.size() * sizeof(int64_t)));
        for (size_t i = 0; i < ip.size(); ++i) indptr_out[i] = (uint64_t)ip[i];
    }
    if (indices_out && m->nnz) {
        std::vector<int32_t> ix(m->nnz);
        SRX_TRY(d2h(ctx, ix.data(), m->d_indices, ix.size() * sizeof(int32_t)));
        for (size_t i = 0; i < ix.size(); ++i) indices_out[i] = (uint64_t)ix[i];
    }
    return SRX_OK;
}

}  // extern "C"
