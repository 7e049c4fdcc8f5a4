/*
 * srx_synth.h — deterministic synthetic count matrices for tests and bench.py
 * (SURVEY.md §8d "Synthetic inputs"), generated DIRECTLY IN HBM (no H2D), with a bit-identical
 * host generator so the CPU oracle can be fed the same matrix.
 *
 * Not part of the reference's API (its own generator is an unseeded random COO,
 * src/memory/processing/mod.rs:343-376); lives in libsrx_hip.so as measurement plumbing.
 *
 * Model (counter-based: every entry is a pure function of (seed, global row, slot), so any
 * rank can generate any row range):
 *   - row nnz r_i = clamp(round(d*G*exp(sigma*z_i - sigma^2/2)), 0, r_max), z_i ~ N(0,1):
 *     log-normal library-size spread; rows with hash % 10000 == 0 are forced EMPTY
 *     (exercises sum == 0 -> scale 0, scale/mod.rs:10-11);
 *   - columns: the gene axis is cut into r_i integer strata, one uniform draw per stratum
 *     => sorted, unique, canonical CSR;
 *   - values: UMI-like integers 1 + Geometric(1/2) (mean 2), exact in f32;
 *   - planted structure, because the PCA of structureless noise is ill-posed (flat spectrum):
 *     each cell has one of n_types cell types (geometric size decay); type t owns the marker
 *     genes [t*M, (t+1)*M); a cell draws its own markers expr_boost x more often (the strata
 *     are laid over a virtual axis on which those genes are replicated) and with
 *     value_boost x the count.  The standardised HVG matrix then has ~n_types-1 eigenvalues
 *     well above the noise bulk.
 */
#ifndef SRX_SYNTH_H
#define SRX_SYNTH_H

#include "srx.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SRX_SYNTH_MAX_TYPES 128

typedef struct srx_synth_params {
    uint64_t seed;
    uint64_t n_rows_global;
    uint64_t n_cols;
    double density;          /* mean nnz fraction d                                        */
    double lib_sigma;        /* log-normal sigma of row nnz (0.3)                          */
    double type_decay;       /* type t has weight decay^t (0.98)                           */
    uint32_t n_types;        /* 1 = no planted structure                                   */
    uint32_t marker_genes;   /* M; n_types * M <= n_cols                                   */
    uint32_t expr_boost;     /* B >= 1                                                     */
    uint32_t value_boost;    /* VB >= 1                                                    */
    uint32_t skew;           /* 0: strata of equal width (uniform gene densities); 1: stratum s of a
                                row with r entries starts at (Gv - r B) (s/r)^2 + s B — the quadratic map of
                                SURVEY.md 8(d): gene density ~ 1 / sqrt(gene index), the first genes are
                                present in almost every cell (per-gene atomic contention, Gram owners
                                of very different weight)                                        */
    uint32_t reserved_;
} srx_synth_params;

/* Fill the defaults used by bench.py for a given shape. */
void srx_synth_defaults(srx_synth_params* p, uint64_t seed, uint64_t n_rows_global,
                        uint64_t n_cols, double density);

/* Host: row offsets of rows [row_begin, row_end) rebased to 0 (row_end-row_begin+1 entries). */
int32_t srx_synth_indptr(const srx_synth_params* p, uint64_t row_begin, uint64_t row_end,
                         uint64_t* indptr_out);
/* Host reference generator (u64 indices + f32 values, the reference layout with F32 values). */
int32_t srx_synth_fill_host(const srx_synth_params* p, uint64_t row_begin, uint64_t row_end,
                            const uint64_t* indptr, uint64_t* indices_out, float* values_out);
/* Device generator: allocates an srx_mat holding rows [row_begin, row_end) and fills it in
 * HBM.  dtype = logical dtype of the values (SRX_F32 typical), store = srx_store.  The handle leaves
 * in the state srx_matrix_upload leaves one in: with the 16-bit index mirror (n_cols <= 65536), the
 * gene-tile cuts and (n_cols <= 38000) the per-gene non-zero counts. */
int32_t srx_synth_generate(srx_ctx* ctx, const srx_synth_params* p, uint64_t row_begin,
                           uint64_t row_end, int32_t dtype, int32_t store, srx_mat** out);

#ifdef __cplusplus
}
#endif
#endif
