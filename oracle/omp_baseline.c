/*
 * omp_baseline.c — threaded CPU baseline for bench.py's `cpu_baseline_threaded` leg (TEST / MEASUREMENT INFRASTRUCTURE
 * ONLY, like everything under oracle/; never loaded by the product).
 *
 * SURVEY.md 8(d), variant (ii): the reference's path with every loop that parallelises trivially spread over OpenMP
 * threads — a stand-in for "what SingleRust would do with Rayon on all loops" — and the PCA taken through the k x k
 * covariance of the selected genes + a symmetric eigen-solve (done by the caller with LAPACK) instead of the reference's
 * full SVD of the densified N x k matrix: algorithmically much cheaper than the reference, i.e. a baseline that favours
 * the CPU.  Arithmetic follows the cited loops: scale = (sum == 0) ? 0 : target / sum, v * scale, ln_1p in f64
 * (scale/mod.rs:9-15,66-83, transform/mod.rs:38-42); per-gene (nnz, sum, sumsq) and the nz-only variance
 * (csr.rs:149-188); stable descending selection (dim_red/mod.rs:135-140).
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double var; uint64_t idx; } gv_t;
static int cmp_desc(const void* a, const void* b) {
    const gv_t *x = a, *y = b;
    if (x->var > y->var) return -1;
    if (x->var < y->var) return 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);     /* stable: ties keep the ascending index */
}

/* values_f32 (raw counts) -> out_f64 (normalised + log1p); hvg_out[k] in rank order; cov[k*k] = Z^T Z of the selected
 * genes (Z = (X - mean) / sd, ddof 0) in ascending-gene order `order_out[k]`; mean / sd in that order.
 * seconds[0..3] = normalise+log1p, moments+selection, Gram, (unused).  Returns 0, or -1 when out of memory. */
int orc_omp_pipeline(uint64_t n_rows, uint64_t n_cols, const uint64_t* indptr, const uint64_t* indices,
                     const float* values_f32, double target, uint64_t n_hvg, int n_threads, double* out_f64,
                     uint64_t* hvg_out, uint64_t* order_out, double* cov, double* mean, double* sd, double* seconds) {
    if (n_threads < 1) n_threads = 1;
    omp_set_num_threads(n_threads);
    const uint64_t k = n_hvg < n_cols ? n_hvg : n_cols;
    double t0 = omp_get_wtime();
#pragma omp parallel for schedule(dynamic, 256)
    for (uint64_t r = 0; r < n_rows; ++r) {
        double s = 0.0;
        for (uint64_t p = indptr[r]; p < indptr[r + 1]; ++p) s += (double)values_f32[p];
        const double scale = s == 0.0 ? 0.0 : target / s;
        for (uint64_t p = indptr[r]; p < indptr[r + 1]; ++p) out_f64[p] = log1p((double)values_f32[p] * scale);
    }
    double t1 = omp_get_wtime();
    seconds[0] = t1 - t0;
    /* per-gene moments: per-thread partials, then a reduction */
    double* part = calloc((size_t)n_threads * n_cols * 3, sizeof(double));
    if (!part) return -1;
#pragma omp parallel
    {
        double* mine = part + (size_t)omp_get_thread_num() * n_cols * 3;
#pragma omp for schedule(static)
        for (uint64_t r = 0; r < n_rows; ++r)
            for (uint64_t p = indptr[r]; p < indptr[r + 1]; ++p) {
                const uint64_t j = indices[p];
                const double x = out_f64[p];
                mine[j] += 1.0;
                mine[n_cols + j] += x;
                mine[2 * n_cols + j] += x * x;
            }
    }
    gv_t* gv = malloc(n_cols * sizeof *gv);
    double* gsum = malloc(n_cols * 2 * sizeof *gsum);
    if (!gv || !gsum) return -1;
#pragma omp parallel for
    for (uint64_t j = 0; j < n_cols; ++j) {
        double c = 0, s = 0, q = 0;
        for (int t = 0; t < n_threads; ++t) {
            const double* pt = part + (size_t)t * n_cols * 3;
            c += pt[j]; s += pt[n_cols + j]; q += pt[2 * n_cols + j];
        }
        gsum[j] = s;
        gsum[n_cols + j] = q;
        gv[j].idx = j;
        gv[j].var = c > 0 ? q / c - (s / c) * (s / c) : 0.0;
    }
    free(part);
    qsort(gv, n_cols, sizeof *gv, cmp_desc);
    int32_t* remap = malloc(n_cols * sizeof *remap);
    if (!remap) return -1;
    for (uint64_t j = 0; j < n_cols; ++j) remap[j] = -1;
    for (uint64_t i = 0; i < k; ++i) hvg_out[i] = gv[i].idx;
    /* ascending-gene order of the selection */
    uint8_t* sel = calloc(n_cols, 1);
    for (uint64_t i = 0; i < k; ++i) sel[gv[i].idx] = 1;
    uint64_t slot = 0;
    for (uint64_t j = 0; j < n_cols; ++j)
        if (sel[j]) {
            remap[j] = (int32_t)slot;
            order_out[slot] = j;
            const double mu = gsum[j] / (double)n_rows;
            double var = gsum[n_cols + j] / (double)n_rows - mu * mu;
            if (var < 0) var = 0;
            mean[slot] = mu;
            sd[slot] = sqrt(var) > 0 ? sqrt(var) : 1.0;
            ++slot;
        }
    free(sel); free(gv);
    double t2 = omp_get_wtime();
    seconds[1] = t2 - t1;
    /* Gram of the selected columns: per-thread k x k accumulators (upper triangle), reduced, then standardised */
    double* gpart = calloc((size_t)n_threads * k * k, sizeof(double));
    if (!gpart) return -1;
#pragma omp parallel
    {
        double* g = gpart + (size_t)omp_get_thread_num() * k * k;
        int32_t cj[4096];
        double cv[4096];
#pragma omp for schedule(dynamic, 64)
        for (uint64_t r = 0; r < n_rows; ++r) {
            int m = 0;
            for (uint64_t p = indptr[r]; p < indptr[r + 1] && m < 4096; ++p) {
                const int32_t c = remap[indices[p]];
                if (c >= 0) { cj[m] = c; cv[m] = out_f64[p]; ++m; }
            }
            for (int a = 0; a < m; ++a) {
                double* row = g + (size_t)cj[a] * k;
                const double va = cv[a];
                for (int b = a; b < m; ++b) row[cj[b]] += va * cv[b];
            }
        }
    }
#pragma omp parallel for schedule(static)
    for (uint64_t e = 0; e < k * k; ++e) {
        double s = 0;
        for (int t = 0; t < n_threads; ++t) s += gpart[(size_t)t * k * k + e];
        cov[e] = s;
    }
    free(gpart); free(remap); free(gsum);
    const double nd = (double)n_rows;
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < k; ++i)
        for (uint64_t j = i; j < k; ++j) {
            const double c = (cov[i * k + j] - nd * mean[i] * mean[j]) / (sd[i] * sd[j]);
            cov[i * k + j] = c;
            cov[j * k + i] = c;
        }
    seconds[2] = omp_get_wtime() - t2;
    seconds[3] = 0.0;
    return 0;
}

/* Standardised covariance Z^T Z of a GIVEN feature selection (sel[k], any order: slot i = feature sel[i]) of the f64 matrix
 * `values` — the independent k x k check of a PCA at sizes where the oracle's exact SVD of the densified N x k matrix does not
 * fit a test (tests/test_fullsize_gpu.py at configs[2]: 1.3M x 28k).  mean / sd over ALL rows, ddof 0 (pca/mod.rs:87-94);
 * cov[i][j] = (sum_r x_ri x_rj - N mean_i mean_j) / (sd_i sd_j), both triangles.  Returns 0, -1 out of memory, -2 when a
 * feature is selected twice or out of range. */
int orc_omp_cov_selected(uint64_t n_rows, uint64_t n_cols, const uint64_t* indptr, const uint64_t* indices,
                         const double* values, const uint64_t* sel, uint64_t k, int n_threads, double* cov, double* mean,
                         double* sd) {
    if (n_threads < 1) n_threads = 1;
    omp_set_num_threads(n_threads);
    int32_t* remap = malloc(n_cols * sizeof *remap);
    double* gpart = calloc((size_t)n_threads * k * k, sizeof(double));
    double* spart = calloc((size_t)n_threads * k, sizeof(double));
    if (!remap || !gpart || !spart) { free(remap); free(gpart); free(spart); return -1; }
    for (uint64_t j = 0; j < n_cols; ++j) remap[j] = -1;
    for (uint64_t i = 0; i < k; ++i) {
        if (sel[i] >= n_cols || remap[sel[i]] >= 0) { free(remap); free(gpart); free(spart); return -2; }
        remap[sel[i]] = (int32_t)i;
    }
#pragma omp parallel
    {
        double* g = gpart + (size_t)omp_get_thread_num() * k * k;
        double* s = spart + (size_t)omp_get_thread_num() * k;
        int32_t* cj = malloc(k * sizeof *cj);
        double* cv = malloc(k * sizeof *cv);
#pragma omp for schedule(dynamic, 64)
        for (uint64_t r = 0; r < n_rows; ++r) {
            uint64_t m = 0;
            for (uint64_t p = indptr[r]; p < indptr[r + 1]; ++p) {
                const int32_t c = remap[indices[p]];
                if (c >= 0) { cj[m] = c; cv[m] = values[p]; ++m; }
            }
            for (uint64_t a = 0; a < m; ++a) {
                double* row = g + (size_t)cj[a] * k;
                const double va = cv[a];
                s[cj[a]] += va;
                for (uint64_t b = 0; b < m; ++b) row[cj[b]] += va * cv[b];
            }
        }
        free(cj); free(cv);
    }
    const double nd = (double)n_rows;
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < k; ++i) {
        double s = 0, q = 0;
        for (int t = 0; t < n_threads; ++t) { s += spart[(size_t)t * k + i]; q += gpart[(size_t)t * k * k + i * k + i]; }
        mean[i] = s / nd;
        double var = q / nd - mean[i] * mean[i];
        sd[i] = var > 0 ? sqrt(var) : 1.0;
    }
#pragma omp parallel for schedule(static)
    for (uint64_t e = 0; e < k * k; ++e) {
        double s = 0;
        for (int t = 0; t < n_threads; ++t) s += gpart[(size_t)t * k * k + e];
        const uint64_t i = e / k, j = e % k;
        cov[e] = (s - nd * (mean[i] * mean[j])) / (sd[i] * sd[j]);      /* (mean_i mean_j) first: symmetric to the last bit */
    }
    free(remap); free(gpart); free(spart);
    return 0;
}

/* scores = Z V for every row (pca/mod.rs:156-185) from the transformed f64 values: slot(c) = remap of the selected features
 * (`order[k]`: slot i = feature order[i]), pv[k * n_pc] = V scaled by 1 / sd (row i = slot i), cvec[n_pc] = (mean / sd) V.
 * OpenMP over rows; the last leg of the threaded baseline at full size (scipy's sparse product is single-threaded). */
int orc_omp_scores(uint64_t n_rows, uint64_t n_cols, const uint64_t* indptr, const uint64_t* indices, const double* values,
                   const uint64_t* order, uint64_t k, const double* pv, const double* cvec, uint64_t n_pc, int n_threads,
                   double* scores) {
    if (n_threads < 1) n_threads = 1;
    omp_set_num_threads(n_threads);
    int32_t* remap = malloc(n_cols * sizeof *remap);
    if (!remap) return -1;
    for (uint64_t j = 0; j < n_cols; ++j) remap[j] = -1;
    for (uint64_t i = 0; i < k; ++i) remap[order[i]] = (int32_t)i;
#pragma omp parallel for schedule(dynamic, 256)
    for (uint64_t r = 0; r < n_rows; ++r) {
        double* out = scores + r * n_pc;
        for (uint64_t c = 0; c < n_pc; ++c) out[c] = -cvec[c];
        for (uint64_t p = indptr[r]; p < indptr[r + 1]; ++p) {
            const int32_t s = remap[indices[p]];
            if (s < 0) continue;
            const double v = values[p];
            const double* row = pv + (size_t)s * n_pc;
            for (uint64_t c = 0; c < n_pc; ++c) out[c] += v * row[c];
        }
    }
    free(remap);
    return 0;
}
