/*
 * srx_oracle.h — CPU restatement of the SingleRust hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle: plain serial C that follows the reference's loops line by
 * line (same accumulation order, same guards).  It is NOT part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Parity status: the reference (Rust) cannot be built in this environment and ships no
 * golden vectors.  The normalise/stat functions are pinned by the reference's only
 * numeric test (src/memory/processing/mod.rs:419-481: row/column sums == target ±1e-6)
 * and by the hand-derived 4x5 known-answer test of SURVEY.md §8(c).  The PCA arithmetic
 * lives in the un-vendored crate single_algebra 0.1.0-alpha.3: PARITY UNPINNED there; the
 * oracle (oracle/pca_oracle.py) restates the in-tree dead predecessor
 * src/shared/processing/pca/mod.rs:74-215.
 *
 * Layout = the reference's nalgebra_sparse::CsrMatrix<T>: usize (u64) row_offsets /
 * col_indices, typed values.  dtype codes follow match_dyn_csr_matrix!
 * (src/shared/mod.rs:110-129).
 */
#ifndef SRX_ORACLE_H
#define SRX_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_I8 = 0, ORC_I16 = 1, ORC_I32 = 2, ORC_U8 = 3, ORC_U16 = 4, ORC_U32 = 5,
       ORC_F32 = 6, ORC_F64 = 7 };
enum { ORC_ROW = 0, ORC_COLUMN = 1 };          /* src/shared/mod.rs:39-42 */

typedef struct {
    uint64_t n_rows, n_cols, nnz;
    const uint64_t* indptr;   /* n_rows + 1 */
    const uint64_t* indices;  /* nnz, sorted + unique per row */
    void* values;             /* nnz elements of dtype */
    int32_t dtype;
} orc_csr;

/* csr.rs:16-38 */
int orc_number(const orc_csr* m, int dir, uint32_t* out);
/* csr.rs:81-102 */
int orc_sum(const orc_csr* m, int dir, double* out);
/* csr.rs:149-188 */
int orc_variance(const orc_csr* m, int dir, double* out);
/* csr.rs:225-228 */
int orc_std_dev(const orc_csr* m, int dir, double* out);
/* csr.rs:194-223 */
int orc_min_max(const orc_csr* m, int dir, double* mn, double* mx);
/* scale/mod.rs:7-23,59-89 (Row) and :91-107,141-173 (Column): out_f64[nnz] receives the
 * scaled values (the reference turns X into DynCsrMatrix::F64). out may alias values when
 * dtype == F64. */
int orc_normalize_total(const orc_csr* m, double target_sum, int dir, double* out_f64);
/* transform/mod.rs:36-57: F64 -> f64 ln_1p, F32 -> f32 ln_1p (stays f32), others promote
 * to f64.  `out` has element type f32 when dtype==F32, else f64; may alias values when
 * the dtype is F32/F64. */
int orc_log1p(const orc_csr* m, void* out);
/* dim_red/mod.rs:135-140: stable descending sort of (idx,var), first n indices in rank
 * order.  Returns -2 when a NaN is met (the reference's partial_cmp().unwrap() panics). */
int orc_select_hvg(const double* var, uint64_t n_genes, uint64_t n, uint64_t* idx_out,
                   uint64_t* n_out);
/* shared/mod.rs:230-259: dense row-major n_rows x k, column c = gene sel[c]. */
int orc_densify_selected(const orc_csr* m, const uint64_t* sel, uint64_t k, double* dense);
/* superset helper used by tests: per-gene (cnt, sum, sumsq) in storage order. */
int orc_gene_moments(const orc_csr* m, uint64_t* cnt, double* sum, double* sumsq);

#ifdef __cplusplus
}
#endif
#endif
