// csc.hip — CSC storage (ArrayData::CscMatrix / DynCscMatrix) on the device.
//
// Reference: src/shared/statistics/helper/csc.rs:15-216 (number / sum / variance / min-max / std-dev for CSC),
// src/memory/processing/scale/mod.rs:25-57,104-139 (scale_row_csc / scale_col_csc), transform/mod.rs (log1p on the
// value array), src/shared/mod.rs:204-215,261-290 (densify of a CSC matrix for the PCA).
//
// A CSC matrix of X (cells x genes) IS the CSR matrix of X^T, and every one of the reference's CSC loops is its CSR
// loop with the direction exchanged — including the quirks: the scattered axis (Row for CSC, csc.rs:148-163) takes
// the naive E[x^2] - E[x]^2 over the non-zeros with a guard for empty lines, the major axis (Column for CSC,
// csc.rs:164-177) takes the two-pass form with NO guard (NaN for an empty column), exactly as csr.rs:158-186 does
// with the roles of Row and Column swapped.  So a CSC matrix is held as the CSR of X^T with `srx_mat::csc` set, the
// entry points exchange the direction (common.hpp: eff_dir), and no CSC-specific statistics kernel exists.
//
// The one thing that needs the other orientation is the PCA (compaction, Gram and SpMM walk cells): the device
// TRANSPOSE below builds the CSR of X from the stored CSR of X^T —
//   1. k_hist:     entries per output row (global u32 atomics; a format conversion, not a per-step pass)
//   2. scan_exclusive -> output row offsets
//   3. k_scatter:  every entry to its output row through an atomic cursor (order inside a row arbitrary)
//   4. k_row_order: one wave per output row orders the row by column with a BITMAP COUNTING SORT in LDS: the
//      columns of a canonical row are distinct, so rank(c) = popcount of the row's column bitmap below c
//      (n_cols bits + one prefix count per 32-bit word per wave); O(nnz_row + n_cols / 32) per row, no comparison
//      network, deterministic output whatever order step 3 produced.
#include "common.hpp"

#include <algorithm>

namespace srx {

int32_t scan_exclusive(srx_ctx* ctx, const int64_t* d_in, uint64_t n, int64_t* d_out, int64_t** total_dev);   // pca_form.hip

__global__ void k_hist(const int32_t* __restrict__ idx, uint64_t nnz, int64_t* __restrict__ cnt) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < nnz; i += stride) atomicAdd((unsigned long long*)&cnt[idx[i]], 1ull);
}

template <typename T>
__global__ void k_scatter(const int64_t* __restrict__ indptr, const int32_t* __restrict__ idx, const T* __restrict__ vals,
                          uint64_t n_rows, const int64_t* __restrict__ out_ptr, unsigned long long* __restrict__ cursor,
                          int32_t* __restrict__ out_idx, T* __restrict__ out_vals) {
    const uint64_t wave = global_wave_id();
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) / kWave;
    const int lane = lane_id();
    for (uint64_t r = wave; r < n_rows; r += n_waves) {
        const int64_t lo = indptr[r], hi = indptr[r + 1];
        for (int64_t p = lo + lane; p < hi; p += kWave) {
            const int32_t c = idx[p];
            const unsigned long long q = (unsigned long long)out_ptr[c] + atomicAdd(&cursor[c], 1ull);
            out_idx[q] = (int32_t)r;
            out_vals[q] = vals[p];
        }
    }
}

// One wave per row: bitmap of the row's columns + per-word prefix counts in LDS, then every entry goes to
// row_start + rank(column).  `words` = ceil(n_cols / 32); LDS per wave = 2 * words * 4 bytes.
template <typename T>
__global__ void k_row_order(const int64_t* __restrict__ ptr, const int32_t* __restrict__ in_idx, const T* __restrict__ in_vals,
                            uint64_t n_rows, int words, int32_t* __restrict__ out_idx, T* __restrict__ out_vals,
                            int* __restrict__ flag) {
    extern __shared__ uint32_t lds_u32[];
    const int waves_per_wg = blockDim.x / kWave;
    const int wave_in_wg = threadIdx.x / kWave;
    uint32_t* bits = lds_u32 + (size_t)wave_in_wg * 2 * words;
    uint32_t* pre = bits + words;
    const int lane = lane_id();
    const uint64_t wave = (uint64_t)blockIdx.x * waves_per_wg + wave_in_wg;
    const uint64_t n_waves = (uint64_t)gridDim.x * waves_per_wg;
    for (uint64_t r = wave; r < n_rows; r += n_waves) {
        const int64_t lo = ptr[r], hi = ptr[r + 1];
        if (hi - lo <= 1) {                              // nothing to order
            if (hi > lo && lane == 0) { out_idx[lo] = in_idx[lo]; out_vals[lo] = in_vals[lo]; }
            continue;
        }
        for (int w = lane; w < words; w += kWave) bits[w] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int64_t p = lo + lane; p < hi; p += kWave) {
            const int32_t c = in_idx[p];
            const uint32_t old = atomicOr(&bits[c >> 5], 1u << (c & 31));
            if (old & (1u << (c & 31))) atomicOr(flag, 2);        // duplicate entry: the input was not canonical
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // exclusive prefix of the word popcounts, 64 words at a time
        uint32_t run = 0;
        for (int w0 = 0; w0 < words; w0 += kWave) {
            const int w = w0 + lane;
            const uint32_t pc = w < words ? (uint32_t)__popc(bits[w]) : 0u;
            uint32_t inc = pc;
#pragma unroll
            for (int off = 1; off < kWave; off <<= 1) {
                const uint32_t o = __shfl_up(inc, off, kWave);
                if (lane >= off) inc += o;
            }
            if (w < words) pre[w] = run + inc - pc;
            run += __shfl(inc, kWave - 1, kWave);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int64_t p = lo + lane; p < hi; p += kWave) {
            const int32_t c = in_idx[p];
            const uint32_t rank = pre[c >> 5] + (uint32_t)__popc(bits[c >> 5] & ((1u << (c & 31)) - 1u));
            out_idx[lo + rank] = c;
            out_vals[lo + rank] = in_vals[p];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

// CSR of the transpose of `m`'s stored matrix (shape n_cols x n_rows), same value storage; `out->csc` is left clear.
int32_t transpose_device(srx_mat* m, srx_mat** out) {
    srx_ctx* ctx = m->ctx;
    *out = nullptr;
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t R = m->n_rows, Cn = m->n_cols, nnz = m->nnz;
    if (R >= (1ull << 31)) return fail(ctx, SRX_E_BOUNDS, "transpose: %llu rows exceed int32 device indices", (unsigned long long)R);
    const int words = (int)((R + 31) / 32);              // columns of the OUTPUT = rows of the stored matrix
    const size_t lds_per_wave = 2 * (size_t)words * sizeof(uint32_t);
    if (lds_per_wave > 160 * 1024)
        return fail(ctx, SRX_E_ARG, "transpose: %llu columns exceed the LDS bitmap of the row-ordering pass", (unsigned long long)R);
    srx_mat* t = nullptr;
    SRX_TRY(srx_matrix_alloc(ctx, Cn, R, nnz, m->dtype, m->store, &t));
    auto bail = [&](int32_t rc) { srx_matrix_free(t); return rc; };
    int64_t* d_cnt;
    unsigned long long* d_cur;
    int32_t* tmp_idx;
    void* tmp_val;
    int* d_flag;
    int32_t rc;
    if ((rc = scratch(ctx, "tr_cnt", (Cn ? Cn : 1) * sizeof(int64_t), (void**)&d_cnt))) return bail(rc);
    if ((rc = scratch(ctx, "tr_cur", (Cn ? Cn : 1) * sizeof(unsigned long long), (void**)&d_cur))) return bail(rc);
    if ((rc = scratch(ctx, "tr_idx", (nnz ? nnz : 1) * sizeof(int32_t), (void**)&tmp_idx))) return bail(rc);
    if ((rc = scratch(ctx, "tr_val", (nnz ? nnz : 1) * val_bytes(m), &tmp_val))) return bail(rc);
    if ((rc = scratch(ctx, "flag", 64, (void**)&d_flag))) return bail(rc);
    hipError_t e = hipMemsetAsync(d_cnt, 0, (Cn ? Cn : 1) * sizeof(int64_t), ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_cur, 0, (Cn ? Cn : 1) * sizeof(unsigned long long), ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_flag, 0, sizeof(int), ctx->stream);
    if (e != hipSuccess) return bail(fail(ctx, SRX_E_HIP, "transpose: %s", hipGetErrorString(e)));
    uint64_t g = (nnz + 255) / 256;
    g = std::min<uint64_t>(std::max<uint64_t>(g, 1), (uint64_t)ctx->n_cus * 32);
    hipLaunchKernelGGL(k_hist, dim3((unsigned)g), dim3(256), 0, ctx->stream, m->d_indices, nnz, d_cnt);
    if ((rc = scan_exclusive(ctx, d_cnt, Cn, t->d_indptr, nullptr))) return bail(rc);
    uint64_t gr = (R + 3) / 4;
    gr = std::min<uint64_t>(std::max<uint64_t>(gr, 1), (uint64_t)ctx->n_cus * 16);
    int waves = (int)std::min<size_t>(4, lds_per_wave ? (160 * 1024) / lds_per_wave : 4);
    if (waves < 1) waves = 1;
    uint64_t go = (Cn + waves - 1) / waves;
    go = std::min<uint64_t>(std::max<uint64_t>(go, 1), (uint64_t)ctx->n_cus * 16);
    const size_t lds = lds_per_wave * waves;
    if (is_f32(m)) {
        hipLaunchKernelGGL((k_scatter<float>), dim3((unsigned)gr), dim3(256), 0, ctx->stream, m->d_indptr, m->d_indices,
                           (const float*)m->d_values, R, t->d_indptr, d_cur, tmp_idx, (float*)tmp_val);
        (void)hipFuncSetAttribute((const void*)k_row_order<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_row_order<float>), dim3((unsigned)go), dim3(waves * kWave), lds, ctx->stream, t->d_indptr, tmp_idx,
                           (const float*)tmp_val, Cn, words, t->d_indices, (float*)t->d_values, d_flag);
    } else {
        hipLaunchKernelGGL((k_scatter<double>), dim3((unsigned)gr), dim3(256), 0, ctx->stream, m->d_indptr, m->d_indices,
                           (const double*)m->d_values, R, t->d_indptr, d_cur, tmp_idx, (double*)tmp_val);
        (void)hipFuncSetAttribute((const void*)k_row_order<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_row_order<double>), dim3((unsigned)go), dim3(waves * kWave), lds, ctx->stream, t->d_indptr, tmp_idx,
                           (const double*)tmp_val, Cn, words, t->d_indices, (double*)t->d_values, d_flag);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return bail(fail(ctx, SRX_E_HIP, "transpose kernels: %s", hipGetErrorString(e)));
    int flag = 0;
    if ((rc = d2h(ctx, &flag, d_flag, sizeof(int)))) return bail(rc);
    if (flag) return bail(fail(ctx, SRX_E_FORMAT, "transpose: duplicate entries (the matrix is not canonical)"));
    t->row_offset = 0;
    *out = t;
    return SRX_OK;
}

}  // namespace srx

using namespace srx;

extern "C" {

int32_t srx_matrix_upload_csc(srx_ctx* ctx, const srx_csr* h, int32_t store, srx_mat** out) {
    if (!ctx || !h || !out) return fail(ctx, SRX_E_ARG, "srx_matrix_upload_csc: null argument");
    *out = nullptr;
    if (ctx->n_ranks > 1)
        return fail(ctx, SRX_E_ARG, "CSC matrices are not sharded across ranks (shard by cells: upload CSR row ranges)");
    // the CSC arrays of X (n_rows x n_cols) are the CSR arrays of X^T (n_cols x n_rows)
    srx_csr t = *h;
    t.n_rows = h->n_cols;
    t.n_cols = h->n_rows;
    if (!t.indptr || (t.nnz && (!t.indices || !t.values))) return fail(ctx, SRX_E_ARG, "srx_matrix_upload_csc: null CSC slice");
    if (t.indptr[t.n_rows] - t.indptr[0] != t.nnz || t.indptr[0] != 0)
        return fail(ctx, SRX_E_FORMAT, "X is not a CSC matrix: col_offsets do not span nnz");
    SRX_TRY(upload_on(ctx, &t, store, ctx->stream, out));
    (*out)->csc = true;
    return SRX_OK;
}

int32_t srx_matrix_format(const srx_mat* m, int32_t* format_out) {
    if (!m || !format_out) return fail(nullptr, SRX_E_ARG, "null argument");
    *format_out = m->csc ? SRX_FORMAT_CSC : SRX_FORMAT_CSR;
    return SRX_OK;
}

int32_t srx_matrix_to_csr(srx_mat* m, srx_mat** out) {
    if (!m || !out) return fail(m ? m->ctx : nullptr, SRX_E_ARG, "null argument");
    if (!m->csc) return srx_matrix_clone(m, out);
    return transpose_device(m, out);
}

int32_t srx_matrix_to_csc(srx_mat* m, srx_mat** out) {
    if (!m || !out) return fail(m ? m->ctx : nullptr, SRX_E_ARG, "null argument");
    if (m->csc) return srx_matrix_clone(m, out);
    if (m->ctx->n_ranks > 1) return fail(m->ctx, SRX_E_ARG, "CSC matrices are not sharded across ranks");
    SRX_TRY(transpose_device(m, out));
    (*out)->csc = true;
    return SRX_OK;
}

}  // extern "C"
