/*
 * srx_oracle.c — serial CPU restatement of the reference loops (TEST INFRASTRUCTURE ONLY;
 * see srx_oracle.h).  Every function cites the reference file:line it follows.  Loops are
 * deliberately serial and in the reference's accumulation order: the reference is
 * single-threaded on this path (SURVEY.md headline fact 4).
 */
#include "srx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* dtype dispatch: mirrors match_dyn_csr_matrix! (src/shared/mod.rs:110-129); dtypes the
 * macro panics on are rejected with -1. */
#define ORC_DISPATCH(m, FN, ...)                                                        \
    switch ((m)->dtype) {                                                               \
    case ORC_I8:  return FN##_i8((m), (const int8_t*)(m)->values, __VA_ARGS__);         \
    case ORC_I16: return FN##_i16((m), (const int16_t*)(m)->values, __VA_ARGS__);       \
    case ORC_I32: return FN##_i32((m), (const int32_t*)(m)->values, __VA_ARGS__);       \
    case ORC_U8:  return FN##_u8((m), (const uint8_t*)(m)->values, __VA_ARGS__);        \
    case ORC_U16: return FN##_u16((m), (const uint16_t*)(m)->values, __VA_ARGS__);      \
    case ORC_U32: return FN##_u32((m), (const uint32_t*)(m)->values, __VA_ARGS__);      \
    case ORC_F32: return FN##_f32((m), (const float*)(m)->values, __VA_ARGS__);         \
    case ORC_F64: return FN##_f64((m), (const double*)(m)->values, __VA_ARGS__);        \
    default: return -1;                                                                 \
    }

#define ORC_FOR_ALL_TYPES(X) \
    X(int8_t, i8) X(int16_t, i16) X(int32_t, i32) X(uint8_t, u8) X(uint16_t, u16) \
    X(uint32_t, u32) X(float, f32) X(double, f64)

/* ---- number: csr.rs:16-38 (dtype-independent) ---------------------------------------- */
int orc_number(const orc_csr* m, int dir, uint32_t* out) {
    if (m->dtype < ORC_I8 || m->dtype > ORC_F64) return -1;
    if (dir == ORC_ROW) {                       /* csr.rs:21-28: windows(2) differences */
        for (uint64_t i = 0; i < m->n_rows; ++i)
            out[i] = (uint32_t)(m->indptr[i + 1] - m->indptr[i]);
    } else {                                    /* csr.rs:29-36: histogram of col_indices */
        memset(out, 0, sizeof(uint32_t) * m->n_cols);
        for (uint64_t p = 0; p < m->nnz; ++p) out[m->indices[p]] += 1;
    }
    return 0;
}

/* ---- sum: csr.rs:81-102 --------------------------------------------------------------- */
#define DEF_SUM(T, S)                                                                   \
    static int sum_##S(const orc_csr* m, const T* v, int dir, double* out) {            \
        if (dir == ORC_ROW) {                   /* csr.rs:87-93: sequential per-row sum */\
            for (uint64_t i = 0; i < m->n_rows; ++i) {                                  \
                double s = 0.0;                                                         \
                for (uint64_t p = m->indptr[i]; p < m->indptr[i + 1]; ++p)              \
                    s += (double)v[p];                                                  \
                out[i] = s;                                                             \
            }                                                                           \
        } else {                                /* csr.rs:94-100: scatter-add in storage order */\
            for (uint64_t j = 0; j < m->n_cols; ++j) out[j] = 0.0;                      \
            for (uint64_t p = 0; p < m->nnz; ++p) out[m->indices[p]] += (double)v[p];   \
        }                                                                               \
        return 0;                                                                       \
    }
ORC_FOR_ALL_TYPES(DEF_SUM)
int orc_sum(const orc_csr* m, int dir, double* out) { ORC_DISPATCH(m, sum, dir, out) }

/* ---- variance: csr.rs:149-188 --------------------------------------------------------- */
#define DEF_VAR(T, S)                                                                   \
    static int var_##S(const orc_csr* m, const T* v, int dir, double* out) {            \
        uint64_t n = dir == ORC_ROW ? m->n_rows : m->n_cols;                            \
        double* sum = (double*)malloc(sizeof(double) * (n ? n : 1));                    \
        uint32_t* cnt = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));              \
        sum_##S(m, v, dir, sum);                /* csr.rs:154 */                         \
        orc_number(m, dir, cnt);                /* csr.rs:155 */                         \
        if (dir == ORC_ROW) {                   /* csr.rs:158-171: two-pass, NaN if empty */\
            for (uint64_t i = 0; i < m->n_rows; ++i) {                                  \
                double mean = sum[i] / (double)cnt[i];                                  \
                double acc = 0.0;                                                       \
                for (uint64_t p = m->indptr[i]; p < m->indptr[i + 1]; ++p) {            \
                    double d = (double)v[p] - mean;                                     \
                    acc += d * d;                                                       \
                }                                                                       \
                out[i] = acc / (double)cnt[i];                                          \
            }                                                                           \
        } else {                                /* csr.rs:172-186: nz-only, naive sumsq */\
            double* sq = (double*)calloc(n ? n : 1, sizeof(double));                    \
            for (uint64_t p = 0; p < m->nnz; ++p) {                                     \
                double x = (double)v[p];                                                \
                sq[m->indices[p]] += x * x;                                             \
            }                                                                           \
            for (uint64_t j = 0; j < n; ++j) {                                          \
                out[j] = 0.0;                                                           \
                if (cnt[j] > 0) {                                                       \
                    double mean = sum[j] / (double)cnt[j];                              \
                    out[j] = sq[j] / (double)cnt[j] - mean * mean;                      \
                }                                                                       \
            }                                                                           \
            free(sq);                                                                   \
        }                                                                               \
        free(sum);                                                                      \
        free(cnt);                                                                      \
        return 0;                                                                       \
    }
ORC_FOR_ALL_TYPES(DEF_VAR)
int orc_variance(const orc_csr* m, int dir, double* out) { ORC_DISPATCH(m, var, dir, out) }

/* csr.rs:225-228 */
int orc_std_dev(const orc_csr* m, int dir, double* out) {
    int rc = orc_variance(m, dir, out);
    if (rc) return rc;
    uint64_t n = dir == ORC_ROW ? m->n_rows : m->n_cols;
    for (uint64_t i = 0; i < n; ++i) out[i] = sqrt(out[i]);
    return 0;
}

/* ---- min/max: csr.rs:194-223 (Rust f64::min/max ignore NaN operands like fmin/fmax) --- */
#define DEF_MINMAX(T, S)                                                                \
    static int minmax_##S(const orc_csr* m, const T* v, int dir, double* mn, double* mx) { \
        uint64_t n = dir == ORC_ROW ? m->n_rows : m->n_cols;                            \
        for (uint64_t i = 0; i < n; ++i) { mn[i] = INFINITY; mx[i] = -INFINITY; }       \
        if (dir == ORC_ROW) {                                                           \
            for (uint64_t i = 0; i < m->n_rows; ++i)                                    \
                for (uint64_t p = m->indptr[i]; p < m->indptr[i + 1]; ++p) {            \
                    double x = (double)v[p];                                            \
                    mn[i] = fmin(mn[i], x);                                             \
                    mx[i] = fmax(mx[i], x);                                             \
                }                                                                       \
        } else {                                                                        \
            for (uint64_t p = 0; p < m->nnz; ++p) {                                     \
                double x = (double)v[p];                                                \
                uint64_t j = m->indices[p];                                             \
                mn[j] = fmin(mn[j], x);                                                 \
                mx[j] = fmax(mx[j], x);                                                 \
            }                                                                           \
        }                                                                               \
        return 0;                                                                       \
    }
ORC_FOR_ALL_TYPES(DEF_MINMAX)
int orc_min_max(const orc_csr* m, int dir, double* mn, double* mx) {
    ORC_DISPATCH(m, minmax, dir, mn, mx)
}

/* ---- normalize_total: scale/mod.rs ----------------------------------------------------- */
#define DEF_NORM(T, S)                                                                  \
    static int norm_##S(const orc_csr* m, const T* v, double target, int dir, double* out) { \
        uint64_t n = dir == ORC_ROW ? m->n_rows : m->n_cols;                            \
        double* scale = (double*)malloc(sizeof(double) * (n ? n : 1));                  \
        sum_##S(m, v, dir, scale);              /* scale/mod.rs:8 / :92 */               \
        for (uint64_t i = 0; i < n; ++i)        /* scale/mod.rs:9-15 / :93-99 */         \
            scale[i] = (scale[i] == 0.0) ? 0.0 : target / scale[i];                     \
        if (dir == ORC_ROW) {                   /* scale/mod.rs:66-83: v *= scale[row] */\
            for (uint64_t i = 0; i < m->n_rows; ++i) {                                  \
                double s = scale[i];                                                    \
                for (uint64_t p = m->indptr[i]; p < m->indptr[i + 1]; ++p)              \
                    out[p] = (double)v[p] * s;                                          \
            }                                                                           \
        } else {                                /* scale/mod.rs:148-166: v *= scale[col] */\
            for (uint64_t p = 0; p < m->nnz; ++p)                                       \
                out[p] = (double)v[p] * scale[m->indices[p]];                           \
        }                                                                               \
        free(scale);                                                                    \
        return 0;                                                                       \
    }
ORC_FOR_ALL_TYPES(DEF_NORM)
int orc_normalize_total(const orc_csr* m, double target_sum, int dir, double* out_f64) {
    ORC_DISPATCH(m, norm, target_sum, dir, out_f64)
}

/* ---- log1p: transform/mod.rs:36-57 ------------------------------------------------------ */
int orc_log1p(const orc_csr* m, void* out) {
    switch (m->dtype) {
    case ORC_F64: {                             /* :38-42 */
        const double* v = (const double*)m->values;
        double* o = (double*)out;
        for (uint64_t p = 0; p < m->nnz; ++p) o[p] = log1p(v[p]);
        return 0;
    }
    case ORC_F32: {                             /* :43-47 stays f32 */
        const float* v = (const float*)m->values;
        float* o = (float*)out;
        for (uint64_t p = 0; p < m->nnz; ++p) o[p] = log1pf(v[p]);
        return 0;
    }
#define LOG1P_PROMOTE(T, C)                                                             \
    case C: {                                   /* :48-55 promote to f64 */              \
        const T* v = (const T*)m->values;                                               \
        double* o = (double*)out;                                                       \
        for (uint64_t p = 0; p < m->nnz; ++p) o[p] = log1p((double)v[p]);               \
        return 0;                                                                       \
    }
        LOG1P_PROMOTE(int8_t, ORC_I8)
        LOG1P_PROMOTE(int16_t, ORC_I16)
        LOG1P_PROMOTE(int32_t, ORC_I32)
        LOG1P_PROMOTE(uint8_t, ORC_U8)
        LOG1P_PROMOTE(uint16_t, ORC_U16)
        LOG1P_PROMOTE(uint32_t, ORC_U32)
    default: return -1;
    }
}

/* ---- HVG top-n: dim_red/mod.rs:135-140 --------------------------------------------------
 * Rust's slice::sort_by is a stable merge sort; comparator b.1.partial_cmp(&a.1) puts
 * larger variances first and keeps equal variances in ascending gene order. */
typedef struct { uint64_t idx; double var; } orc_iv;

static void merge_desc(orc_iv* a, orc_iv* tmp, size_t lo, size_t mid, size_t hi) {
    size_t i = lo, j = mid, k = lo;
    while (i < mid && j < hi) {
        /* take from the right run only when strictly greater: stability */
        if (a[j].var > a[i].var) tmp[k++] = a[j++]; else tmp[k++] = a[i++];
    }
    while (i < mid) tmp[k++] = a[i++];
    while (j < hi) tmp[k++] = a[j++];
    memcpy(a + lo, tmp + lo, (hi - lo) * sizeof(orc_iv));
}
static void msort_desc(orc_iv* a, orc_iv* tmp, size_t lo, size_t hi) {
    if (hi - lo < 2) return;
    size_t mid = lo + (hi - lo) / 2;
    msort_desc(a, tmp, lo, mid);
    msort_desc(a, tmp, mid, hi);
    merge_desc(a, tmp, lo, mid, hi);
}

int orc_select_hvg(const double* var, uint64_t n_genes, uint64_t n, uint64_t* idx_out,
                   uint64_t* n_out) {
    for (uint64_t j = 0; j < n_genes; ++j)
        if (var[j] != var[j]) return -2;        /* NaN: reference unwrap() panics */
    orc_iv* a = (orc_iv*)malloc(sizeof(orc_iv) * (n_genes ? n_genes : 1));
    orc_iv* t = (orc_iv*)malloc(sizeof(orc_iv) * (n_genes ? n_genes : 1));
    for (uint64_t j = 0; j < n_genes; ++j) { a[j].idx = j; a[j].var = var[j]; }
    msort_desc(a, t, 0, n_genes);
    uint64_t take = n < n_genes ? n : n_genes;  /* .take(n) */
    for (uint64_t j = 0; j < take; ++j) idx_out[j] = a[j].idx;
    *n_out = take;
    free(a);
    free(t);
    return 0;
}

/* ---- densify selected columns: shared/mod.rs:230-259 ------------------------------------ */
#define DEF_DENS(T, S)                                                                  \
    static int dens_##S(const orc_csr* m, const T* v, const uint64_t* sel, uint64_t k,  \
                        double* dense) {                                                \
        /* HashMap col -> out_col (:241); a later duplicate in sel overrides */         \
        int64_t* map = (int64_t*)malloc(sizeof(int64_t) * (m->n_cols ? m->n_cols : 1)); \
        for (uint64_t j = 0; j < m->n_cols; ++j) map[j] = -1;                           \
        for (uint64_t c = 0; c < k; ++c) {                                              \
            if (sel[c] >= m->n_cols) { free(map); return -3; } /* utils bounds check */ \
            map[sel[c]] = (int64_t)c;                                                   \
        }                                                                               \
        memset(dense, 0, sizeof(double) * m->n_rows * k);                               \
        for (uint64_t i = 0; i < m->n_rows; ++i)                                        \
            for (uint64_t p = m->indptr[i]; p < m->indptr[i + 1]; ++p) {                \
                int64_t c = map[m->indices[p]];                                         \
                if (c >= 0) dense[i * k + (uint64_t)c] = (double)v[p];                  \
            }                                                                           \
        free(map);                                                                      \
        return 0;                                                                       \
    }
ORC_FOR_ALL_TYPES(DEF_DENS)
int orc_densify_selected(const orc_csr* m, const uint64_t* sel, uint64_t k, double* dense) {
    ORC_DISPATCH(m, dens, sel, k, dense)
}

/* ---- per-gene (cnt, sum, sumsq): the three column passes of csr.rs:29-36,94-100,175-178 - */
#define DEF_MOM(T, S)                                                                   \
    static int mom_##S(const orc_csr* m, const T* v, uint64_t* cnt, double* sum,        \
                       double* sq) {                                                    \
        for (uint64_t j = 0; j < m->n_cols; ++j) { cnt[j] = 0; sum[j] = 0.0; sq[j] = 0.0; } \
        for (uint64_t p = 0; p < m->nnz; ++p) {                                         \
            double x = (double)v[p];                                                    \
            uint64_t j = m->indices[p];                                                 \
            cnt[j] += 1;                                                                \
            sum[j] += x;                                                                \
            sq[j] += x * x;                                                             \
        }                                                                               \
        return 0;                                                                       \
    }
ORC_FOR_ALL_TYPES(DEF_MOM)
int orc_gene_moments(const orc_csr* m, uint64_t* cnt, double* sum, double* sumsq) {
    ORC_DISPATCH(m, mom, cnt, sum, sumsq)
}
