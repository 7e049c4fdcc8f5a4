/*
 * srx.h — C ABI of libsrx_hip.so: the MI355X (gfx950) implementation of SingleRust's
 * sparse count-matrix hot path
 *
 *     normalize_total(1e4, Row) -> log1p -> per-gene moments / HVG(n) -> n_pc-component PCA
 *
 * This header IS the drop-in boundary.  The reference (SingleRust/SingleRust @ 2024_10_08,
 * pure Rust) has no FFI of its own; each entry point below replaces the body of one
 * reference function and is what a Rust shim in that function would bind (see
 * INTEGRATION.md for the `extern "C"` block and the shim).  Paths cite the reference as
 * <file>:<lines> relative to the reference repository root.
 *
 * Conventions
 *   - plain C types only; every function returns an srx_status (0 = ok, <0 = error) and
 *     never throws or aborts across the boundary; srx_last_error() gives the message the
 *     shim turns into anyhow!(..).
 *   - the caller owns every host buffer; the library never keeps a host pointer after a
 *     call returns.  An srx_mat owns device memory only.
 *   - one srx_ctx per process and GPU (one process per GPU); a ctx is not thread-safe —
 *     the reference holds the IMArrayElement RwLock for the whole op, the shim does too.
 *   - there is NO CPU fallback: without a usable HIP device every compute call fails with
 *     SRX_E_HIP.
 */
#ifndef SRX_H
#define SRX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRX_ABI_VERSION 6      /* 2: srx_matrix_reserve_results, kernel classes 7-9, srx_synth_params.skew; 3: kernel class 10;
                                  4: srx_comm_info, srx_prof_get_aux; 5: srx_comm_overlap_info; 6: srx_gram_mode_info, srx_gram_exchange_ranges */

typedef struct srx_ctx srx_ctx;   /* one GPU + stream + (optional) RCCL communicator      */
typedef struct srx_mat srx_mat;   /* device-resident CSR (the `X` of an IMAnnData)        */

typedef enum srx_status {
    SRX_OK = 0,
    SRX_E_ARG = -1,      /* null pointer / inconsistent sizes                             */
    SRX_E_DTYPE = -2,    /* dtype outside match_dyn_csr_matrix! (src/shared/mod.rs:110-129) */
    SRX_E_FORMAT = -3,   /* "X is not a CSR matrix" (scale/mod.rs:87, transform/mod.rs:58) */
    SRX_E_BOUNDS = -4,   /* column index >= n_cols, selection index out of range          */
    SRX_E_HIP = -5,      /* HIP runtime error / no device                                 */
    SRX_E_RCCL = -6,
    SRX_E_OOM = -7,
    SRX_E_NAN = -8,      /* NaN variance in HVG ranking: partial_cmp().unwrap() panic,
                            dim_red/mod.rs:138                                            */
    SRX_E_SHAPE = -9,    /* pca_inplace needs k >= 2 and N >= 5 (dim_red/mod.rs:38-41)    */
    SRX_E_NOCONV = -10   /* subspace iteration hit max_iter before reaching tol           */
} srx_status;

/* Value dtypes: the set match_dyn_csr_matrix! accepts (src/shared/mod.rs:114-124).  The
 * variants it panics on (I64, U64, Usize, Bool, String) map to SRX_E_DTYPE. */
typedef enum srx_dtype {
    SRX_I8 = 0, SRX_I16 = 1, SRX_I32 = 2, SRX_U8 = 3, SRX_U16 = 4, SRX_U32 = 5,
    SRX_F32 = 6, SRX_F64 = 7
} srx_dtype;

/* src/shared/mod.rs:39-42 */
typedef enum srx_direction { SRX_ROW = 0, SRX_COLUMN = 1 } srx_direction;

/* The reference data model: nalgebra_sparse::CsrMatrix<T> slices as the
 * shim obtains them under the IMArrayElement guard — row_offsets(), col_indices(),
 * values()/values_mut().  usize == uint64_t on x86-64. */
typedef struct srx_csr {
    uint64_t n_rows, n_cols, nnz;
    const uint64_t* indptr;    /* n_rows + 1                                              */
    const uint64_t* indices;   /* nnz; sorted and unique within a row (canonical CSR)     */
    void* values;              /* nnz elements of `dtype`                                  */
    int32_t dtype;             /* srx_dtype                                               */
} srx_csr;

/* Device value storage chosen at upload. AUTO: I8/I16/U8/U16/F32 -> f32 (exact), I32/U32/
 * F64 -> f64 — and the storage then FOLLOWS the reference's variant: where the reference turns X
 * into DynCsrMatrix::F64 (normalize_total on anything, log1p on a non-F32 matrix:
 * scale/mod.rs:74-83, transform/mod.rs:48-55) an AUTO handle held in f32 is widened to f64, under
 * the separate calls and under srx_pipeline alike; srx_matrix_copy_values on an AUTO destination
 * follows the source's storage.  An explicit F32 keeps f32 throughout: half the HBM traffic of
 * every pass, still within the 1e-5 bound; F64 reproduces the reference's f64 arithmetic to ~1e-15. */
typedef enum srx_store { SRX_STORE_AUTO = 0, SRX_STORE_F32 = 1, SRX_STORE_F64 = 2 } srx_store;

/* ---- context ---------------------------------------------------------------------------- */
int32_t srx_abi_version(void);
int32_t srx_device_count(int32_t* n_out);
int32_t srx_ctx_create(int32_t device_id, srx_ctx** out);
void    srx_ctx_destroy(srx_ctx* ctx);
int32_t srx_ctx_synchronize(srx_ctx* ctx);
/* Message of the last failing call on this ctx (or, with ctx == NULL, on this thread). */
const char* srx_last_error(const srx_ctx* ctx);

/* ---- row sharding over the GPUs of one node (new; the reference is single-process) ------
 * One process per GPU.  Rank 0 obtains an id, the host broadcasts the 128 bytes by any
 * means (torch.distributed / MPI / a file), every rank calls srx_comm_init.  Afterwards
 * every per-gene reduction (moments, Z^T Y blocks, Gram matrices, the global cell count)
 * is summed across ranks with ONE ncclAllReduce(f64, sum) each over xGMI; per-cell results
 * stay local to the rank that owns the rows.  Entry points whose answer would be the shard's own
 * rather than the matrix's refuse a sharded context with SRX_E_ARG: srx_filter_cells /
 * srx_filter_genes with a FlexValue::Relative limit (a quantile of all cells' / genes' sums) and
 * srx_compute_min_max(Column); CSC handles are single-rank. */
#define SRX_UNIQUE_ID_BYTES 128
int32_t srx_comm_unique_id(void* id_out_128);
int32_t srx_comm_init(srx_ctx* ctx, int32_t n_ranks, int32_t rank, const void* id_128);
/* Alternative to srx_comm_init for hosts that already own a transport (MPI, sockets): every
 * cross-rank sum of the path (3G+1 doubles of moments, the packed Gram tiles or the k x 64 block,
 * 64 x 64 blocks) is handed to `fn` as a host buffer to be summed over all ranks IN PLACE
 * (return 0 on success).  Collective calls must be made by all ranks in the same order. */
typedef int32_t (*srx_host_allreduce_fn)(void* user, double* buf, uint64_t count);
int32_t srx_comm_init_host(srx_ctx* ctx, int32_t n_ranks, int32_t rank, srx_host_allreduce_fn fn, void* user);
int32_t srx_comm_destroy(srx_ctx* ctx);
/* What the context's cross-rank sums go through: *kind_out = 0 (no communicator), 1 (RCCL), 2 (host transport);
 * *rccl_version_out = ncclGetVersion() (0 when RCCL is not loaded); *ranks_seen_out = a 1.0 summed over the
 * communicator with the same all-reduce the path uses (a COLLECTIVE call when kind != 0: every rank makes it) —
 * equal to n_ranks when every rank really takes part.  Any out pointer may be NULL. */
int32_t srx_comm_info(srx_ctx* ctx, int32_t* kind_out, int32_t* n_ranks_out, int32_t* rccl_version_out,
                      int32_t* ranks_seen_out);
/* How the Gram solver's one exchange has been run on this context so far: *split_exchanges_out = the number of Gram
 * matrices whose packed triangle went round the ranks in three pieces, two of them UNDER the second half of the stripe
 * kernel (the arrangement of a multi-rank RCCL context); *cu_masked_out = 1 when that second half ran on a stream whose
 * CU mask keeps it off the CUs left to the collective (0: the mask could not be set, or no split exchange yet).
 * Introspection for tests and the bench line; either pointer may be NULL. */
int32_t srx_comm_overlap_info(srx_ctx* ctx, int32_t* split_exchanges_out, int32_t* cu_masked_out);
/* Which accumulation mode the LAST sparse Gram launch of this context ran in (the kernel decides on the device, from value
 * statistics of the compacted matrix — summed over the ranks first when the rows are sharded, so that every rank takes the
 * same mode): *mode_out = 0 (no launch yet), 1 (f64 LDS atomics), 2 (fixed-point products, 64-bit integer LDS atomics:
 * non-negative values whose binary exponents lie within 6 of the largest's).  Drains the stream.  Introspection for tests. */
int32_t srx_gram_mode_info(srx_ctx* ctx, int32_t* mode_out);
/* The Gram solver's one exchange when the rows are sharded over RCCL ranks: the packed upper triangle of X_sel^T X_sel
 * (k (k + 1) / 2 doubles, row-major) is summed in THREE all-reduces — offsets_out[0] .. [1] and [2] .. [3] first (under the
 * second half of the stripe kernel), [1] .. [2] after it; offsets_out[3] = k (k + 1) / 2.  The ranges depend on k alone (every
 * rank, with or without rows, issues the same three calls).  Pure host helper, no GPU needed. */
int32_t srx_gram_exchange_ranges(uint64_t k, uint64_t* offsets_out);
/* Contiguous nnz-balanced row ranges: cut[r]..cut[r+1] is rank r's rows (cut has
 * n_ranks+1 entries).  Pure host helper, no GPU needed. */
int32_t srx_partition_rows(const uint64_t* indptr, uint64_t n_rows, int32_t n_ranks,
                           uint64_t* cut_out);

/* ---- matrix ------------------------------------------------------------------------------ */
/* Narrowing upload of a reference-layout CSR (u64 -> i32 column indices on device, i64 row
 * offsets).  Validates sortedness/bounds on device (SRX_E_FORMAT / SRX_E_BOUNDS).
 * The indices are narrowed on the HOST side of the link by the transfer workers (to 16 bits
 * when n_cols <= 65536: 2 of their 8 bytes cross PCIe); the host buffers may be pageable or
 * pinned (hipHostMalloc / hipHostRegister: values of the storage type are then copied by one
 * DMA straight out of them).  The call returns when every host buffer has been read. */
int32_t srx_matrix_upload(srx_ctx* ctx, const srx_csr* host, int32_t store, srx_mat** out);
/* CSC storage (ArrayData::CscMatrix / DynCscMatrix): `host` describes X (n_rows cells x n_cols
 * genes) with indptr = col_offsets[n_cols+1], indices = row_indices[nnz] (sorted per column).
 * Replaces the CSC arms of the reference: src/shared/statistics/helper/csc.rs:15-216,
 * scale_row_csc / scale_col_csc (src/memory/processing/scale/mod.rs:25-57,104-139), log1p on a
 * CSC matrix, convert_to_array_f64_csc[_selected] (src/shared/mod.rs:204-215,261-290) for the PCA.
 * Every entry point that takes an srx_mat accepts the handle and follows the reference's CSC
 * arithmetic (e.g. compute_variance(Column) is the two-pass form with NaN for an empty gene,
 * csc.rs:164-177); srx_matrix_info reports the shape of X; srx_matrix_download_pattern / _values
 * return the CSC arrays.  srx_pca / srx_pipeline transpose on the device first.  Not available
 * on a CSC handle: srx_gene_moments, srx_spmm, multi-rank contexts (shard by cells = CSR rows). */
int32_t srx_matrix_upload_csc(srx_ctx* ctx, const srx_csr* host, int32_t store, srx_mat** out);
enum { SRX_FORMAT_CSR = 0, SRX_FORMAT_CSC = 1 };
int32_t srx_matrix_format(const srx_mat* m, int32_t* format_out);
/* Storage conversion on the device (a copy; the input is left as it is). */
int32_t srx_matrix_to_csr(srx_mat* m, srx_mat** out);
int32_t srx_matrix_to_csc(srx_mat* m, srx_mat** out);
/* Uninitialised device CSR for producers that fill HBM directly (synthetic generator, a
 * host that already holds device buffers). `dtype` is the logical dtype the values are
 * deemed to have (what DynCsrMatrix variant X is). */
int32_t srx_matrix_alloc(srx_ctx* ctx, uint64_t n_rows, uint64_t n_cols, uint64_t nnz,
                         int32_t dtype, int32_t store, srx_mat** out);
/* Device pointers of an srx_mat: indptr = int64[n_rows+1], indices = int32[nnz],
 * values = float[nnz] or double[nnz] (see srx_matrix_info). */
int32_t srx_matrix_device_ptrs(srx_mat* m, void** indptr, void** indices, void** values);
typedef struct srx_mat_info {
    uint64_t n_rows, n_cols, nnz;  /* shape of X (cells x genes), CSR or CSC                  */
    int32_t dtype;        /* current LOGICAL dtype (F64 after normalize_total, ...)       */
    int32_t store;        /* SRX_STORE_F32 or SRX_STORE_F64                               */
    uint64_t row_offset;  /* global index of local row 0 (sharded runs)                   */
    uint64_t n_rows_global;
} srx_mat_info;
int32_t srx_matrix_info(const srx_mat* m, srx_mat_info* out);
/* Tell a shard where it sits in the global matrix (used for deterministic seeds only). */
int32_t srx_matrix_set_shard(srx_mat* m, uint64_t row_offset);
/* D2H of the current values, converted to SRX_F32 or SRX_F64 (what the shim copies back
 * into values_mut(), or into the fresh Vec<f64> when the DynCsrMatrix variant changes). */
int32_t srx_matrix_download_values(srx_mat* m, void* values_out, int32_t dtype_out);
/* Build the pattern-only index structures of the device layout now (the gene-tile cuts of every
 * row used by the per-gene passes) instead of lazily at first use; clones inherit them.  They
 * depend on indptr/indices only, never on the values. */
int32_t srx_matrix_prepare(srx_mat* m);
/* Allocate the handle's result block now (N x n_components f64 scores + the small results of a PCA over
 * n_selected features) instead of inside the first srx_pca / srx_pipeline on it: a pipeline step on a prepared
 * and reserved matrix makes no device allocation.  Clones inherit the reservation (each gets its own block). */
int32_t srx_matrix_reserve_results(srx_mat* m, uint64_t n_selected, int32_t n_components);
/* Deep copy on device (normalize_total / log1p_transform, the copying forms,
 * processing/mod.rs:314-322,329-332, deep_clone X). */
int32_t srx_matrix_clone(srx_mat* m, srx_mat** out);
/* Overwrite dst's values/dtype from src (same sparsity pattern); bench uses it to restore
 * raw counts between steps. */
int32_t srx_matrix_copy_values(srx_mat* dst, const srx_mat* src);
void    srx_matrix_free(srx_mat* m);

/* ---- statistics: memory::statistics::* (src/memory/statistics/mod.rs:10-46) ------------- */
/* compute_number -> csr.rs:16-38.  out: n_rows (Row) or n_cols (Column) u32. */
int32_t srx_compute_number(srx_mat* m, int32_t direction, uint32_t* out);
/* compute_sum -> csr.rs:81-102. f64 accumulate; exact for integer data. */
int32_t srx_compute_sum(srx_mat* m, int32_t direction, double* out);
/* compute_variance -> csr.rs:149-188. Column: variance over the NON-ZERO entries,
 * sumsq/cnt - mean^2, 0.0 when cnt == 0.  Row: two-pass, NaN for an empty row. */
int32_t srx_compute_variance(srx_mat* m, int32_t direction, double* out);
/* compute_std_dev -> csr.rs:225-228. */
int32_t srx_compute_std_dev(srx_mat* m, int32_t direction, double* out);
/* compute_min_max -> csr.rs:194-223 (+inf/-inf for empty). */
int32_t srx_compute_min_max(srx_mat* m, int32_t direction, double* min_out, double* max_out);
/* compute_qc_variables (statistics/mod.rs:48-72): the eight vectors of StatisticsContainer
 * (structs/mod.rs:1-10) from ONE pass over the rows (number, sum, variance per cell; NaN variance
 * for an empty cell, csr.rs:161) and ONE pass over the columns (the cached per-gene moments) — the
 * reference makes ~14 serial passes.  Per-cell outputs have n_rows entries, per-gene outputs n_cols;
 * any pointer may be NULL. */
int32_t srx_compute_qc_variables(srx_mat* m, uint32_t* num_per_cell, uint32_t* num_per_gene, double* expr_per_gene,
                                 double* expr_per_cell, double* variance_per_gene, double* variance_per_cell,
                                 double* std_dev_per_cell, double* std_dev_per_gene);
/* Superset used internally: per-gene (nnz_j, sum x, sum x^2) from ONE pass. Any output may
 * be NULL. */
int32_t srx_gene_moments(srx_mat* m, uint64_t* cnt, double* sum, double* sumsq);

/* ---- QC filters: memory::processing::filter_cells / filter_genes (processing/mod.rs:86-146, :245-299) -------
 * FlexValue (src/shared/mod.rs:62-66): Absolute(u32) thresholds the nnz COUNT of the cell / gene, Relative(p)
 * the linear-interpolated p-quantile of the SUMS (calculate_percentiles, mod.rs:148-174), None = no limit; the
 * nine (lower, upper) combinations of create_filter_mask (mod.rs:33-84).  `*out` receives a NEW matrix holding
 * the kept rows / columns in their original order (the reference's `subset`; the in-place forms swap it in);
 * `mask_out` (n_rows resp. n_cols bytes, may be NULL) receives the boolean mask. */
enum { SRX_FLEX_NONE = 0, SRX_FLEX_ABSOLUTE = 1, SRX_FLEX_RELATIVE = 2 };
typedef struct srx_flex {
    int32_t kind;      /* SRX_FLEX_* */
    uint32_t absolute; /* FlexValue::Absolute(u32) */
    double relative;   /* FlexValue::Relative(f64), a quantile in [0, 1] */
} srx_flex;
int32_t srx_filter_cells(srx_mat* m, srx_flex lower, srx_flex upper, srx_mat** out, uint8_t* mask_out);
int32_t srx_filter_genes(srx_mat* m, srx_flex lower, srx_flex upper, srx_mat** out, uint8_t* mask_out);
/* anndata `subset` on X: boolean masks over rows / columns (NULL = keep all), order preserved. */
int32_t srx_subset(srx_mat* m, const uint8_t* row_mask, const uint8_t* col_mask, srx_mat** out);
/* Sparsity pattern back to the host in the reference's layout (u64 offsets / indices); either may be NULL. */
int32_t srx_matrix_download_pattern(srx_mat* m, uint64_t* indptr_out, uint64_t* indices_out);

/* ---- normalise / log1p: memory::processing::* ------------------------------------------- */
/* normalize_total_inplace (processing/mod.rs:303-312 -> scale/mod.rs:7-23,59-89;
 * Column: :91-107,141-173).  scale = (sum == 0) ? 0 : target/sum; v *= scale.  The logical
 * dtype becomes F64 whatever it was (scale/mod.rs:74-83). */
int32_t srx_normalize_total_inplace(srx_mat* m, double target_sum, int32_t direction);
/* log1p_transform_inplace (processing/mod.rs:324-326 -> transform/mod.rs:36-57). Logical
 * F32 stays F32 (f32::ln_1p), everything else becomes F64. */
int32_t srx_log1p_inplace(srx_mat* m);
/* Fused fast path of the two calls above with Direction::Row: one read + one write of the
 * values (8 B/nnz at f32 storage).  row_sums_out (host, n_rows f64) may be NULL. */
int32_t srx_normalize_log1p_inplace(srx_mat* m, double target_sum, double* row_sums_out);

/* ---- feature selection: dim_red::select_features, arm HighlyVariable(n)
 * (dim_red/mod.rs:135-140): stable descending sort of the nz-only gene variances, first n
 * gene indices IN RANK ORDER.  idx_out has room for min(n, n_cols) entries. */
int32_t srx_select_hvg(srx_mat* m, uint64_t n, uint64_t* idx_out, uint64_t* n_out);

/* ---- PCA: dim_red::pca_inplace (dim_red/mod.rs:24-94); arithmetic spec
 * src/shared/processing/pca/mod.rs:74-215.  Computed by randomized block subspace
 * iteration on the implicit standardised matrix Z = (X[:, sel] - 1 mu^T) D^-1 with a
 * CSR x dense-panel SpMM (forward) and its transpose, or — Gram solver — with X_sel^T X_sel
 * accumulated once from the sparse rows; never densifies X.  Between Rayleigh–Ritz steps the
 * block is advanced by Chebyshev filters of C = Z^T Z (plain powers during the warm-up); the
 * whole iteration runs on the device (SRX_NO_GRAPH=1 / SRX_NO_CHEB=1 / SRX_PCA_TRACE=1 in the
 * environment: no hipGraph replay / plain sweeps instead of filters / residual trace on stderr).
 * Hard spectra (a few strong components over a flat bulk) are handled by bounding the filter
 * degree, by deflation rounds of <= 32 components when one round stalls, and by a last-resort
 * mode (CholeskyQR3 after every application of C; SRX_PCA_ROBUST=1 forces it): SRX_E_NOCONV is
 * left for a zero matrix, NaN input or an exhausted max_iter.  n_components beyond the structural
 * rank bound min(k, N - centred) is SRX_E_ARG (k <= 64 excepted: the exact eigen-solver takes any
 * rank); components beyond the NUMERICAL rank of the selected columns (a handful of cells, most
 * selected columns empty) come back with explained variance 0 and a zero vector (k > 64) or an
 * arbitrary unit vector of the null space (k <= 64) — the reference returns whatever its SVD
 * makes of a zero singular value. */
typedef enum srx_pca_solver {
    SRX_SOLVER_AUTO = 0,
    SRX_SOLVER_GRAM = 1,   /* explicit sparse Gram X_sel^T X_sel once, dense k x k iteration  */
    SRX_SOLVER_SPMM = 2    /* matrix-free: forward + transposed SpMM every iteration         */
} srx_pca_solver;

typedef struct srx_pca_opts {
    int32_t n_components;  /* < 0 = None -> 2        (dim_red/mod.rs:52); any value up to k
                              with the Gram solver (beyond 56 the solve runs in deflation
                              rounds of 48), up to 56 with the SpMM solver                */
    int32_t center;        /* < 0 = None -> true     (:55)                                */
    int32_t scale;         /* < 0 = None -> true     (:56)                                */
    int32_t n_threads;     /* accepted, ignored on GPU (:61)                              */
    int32_t block;         /* panel width l; 0 -> default (64)                            */
    int32_t max_iter;      /* bound on the sweeps (one sweep = power applications of C:
                              3 in the Gram solver, 1 in the SpMM solver) after the
                              warm-up; 0 -> default 200 (Gram) / 600 (SpMM)               */
    int32_t solver;        /* srx_pca_solver; 0 = auto (Gram when k <= 16384)             */
    double  tol;           /* relative Ritz-residual tolerance; 0 -> default (1e-7 with
                              f32 storage, 1e-9 with f64 storage)                         */
    uint64_t seed;         /* start panel seed (the reference has no randomness here)     */
} srx_pca_opts;

typedef struct srx_pca_info {
    uint64_t n_cells_global;
    uint32_t k;            /* selected features                                           */
    uint32_t n_pc;
    uint32_t block;
    uint32_t n_iter;       /* sweep equivalents executed (applications of C / power)      */
    double residual;       /* final max relative Ritz residual over the n_pc pairs        */
    uint64_t nnz_selected; /* non-zeros of the HVG-compacted CSR actually walked (local)  */
    uint32_t solver;       /* srx_pca_solver actually used                                */
    uint32_t reserved_;
} srx_pca_info;

/* sel: k feature indices in selection order (NULL = FeatureSelection::None, all genes).
 * Host outputs, each may be NULL:
 *   scores      n_rows x n_pc  row-major  (obsm["X_pca"], dim_red/mod.rs:105-106)
 *   components  k x n_pc       row-major  (V[:, :n_pc], pca/mod.rs:144)
 *   evr         n_pc           explained_variance_ratio (pca/mod.rs:145-149)
 *   mean, std   k              column mean / std (ddof 0) over ALL cells (pca/mod.rs:87-91)
 * Sign convention (the reference has none): each component is flipped so that its
 * largest-|.| entry is positive.  A zero-variance selected column gives NaN in the
 * reference (pca/mod.rs:108); here its std is treated as 1 (documented deviation). */
int32_t srx_pca(srx_mat* m, const uint64_t* sel, uint64_t k, const srx_pca_opts* opts,
                double* scores, double* components, double* evr, double* mean, double* std_,
                srx_pca_info* info);
/* varm["PCA_loadings"] layout (dim_red/mod.rs:108-118): n_vars x n_pc, row sel[i] <-
 * loadings row i = components[i,:] * std[i] (pca/mod.rs:204-215), other rows 0. */
int32_t srx_pca_loadings(const double* components, const double* std_, const uint64_t* sel,
                         uint64_t k, uint64_t n_pc, uint64_t n_vars, double* out);

/* The raw operators the PCA is built from, for a caller-supplied 64-column panel (k x 64
 * row-major): y_out (n_rows x 64) = X[:, sel] * panel, t_out (k x 64) = X[:, sel]^T * y, and
 * gram_out (k x k) = X[:, sel]^T X[:, sel] (panel not needed).  `sel` must be strictly
 * ascending; any output may be NULL. */
int32_t srx_spmm(srx_mat* m, const uint64_t* sel, uint64_t k, const double* panel, double* y_out,
                 double* t_out, double* gram_out);

/* ---- fused pipeline: normalize_total_inplace(target, Row) -> log1p_transform_inplace ->
 * pca_inplace(n_pc, center, scale, .., HighlyVariable(n_hvg)) with every intermediate
 * resident in HBM.  Results stay on the device until fetched. */
typedef struct srx_pipeline_result {
    srx_pca_info pca;
    double ms_normalize, ms_moments, ms_select, ms_compact, ms_pca; /* hipEvent stage times */
} srx_pipeline_result;
int32_t srx_pipeline(srx_mat* m, double target_sum, uint64_t n_hvg, const srx_pca_opts* opts,
                     srx_pipeline_result* res);
/* Fetch what the last srx_pca / srx_pipeline on `m` left in HBM (any pointer may be NULL;
 * hvg_idx has room for `k` entries). */
int32_t srx_result_fetch(srx_mat* m, double* scores, double* components, double* evr,
                         double* mean, double* std_, uint64_t* hvg_idx);

/* ---- backed (out-of-core) mode: the matrix is visited as consecutive ROW TILES ---------------
 * Replaces src/backed/statistics/mod.rs:5-45 (compute_number / compute_sum with
 * ComputationMode::Chunked(size); chunk loops src/shared/statistics/mod.rs:17-41,59-83,
 * csr.rs:48-74,112-143) and carries the whole path for matrices larger than HBM: two sweeps over
 * the row tiles, only the per-gene moments, the k x k Gram tiles and the HVG-compacted rows stay
 * on the device.  Unlike the reference's Direction::Row chunk loop (csr.rs:126 drops the chunk's
 * row offset), per-row outputs of a tile are written by the caller at the tile's global offset.
 *
 * A tile is an srx_csr whose `indptr` may be a WINDOW of the matrix's row_offsets (n_rows+1
 * entries, indptr[0] != 0 allowed) with `indices` / `values` pointing at the tile's first entry.
 * The upload of a tile overlaps the kernels of the previous one (own stream + staging buffers).
 * Multi-GPU: every rank runs its own session over its own row range; srx_backed_select and
 * srx_backed_solve are collective (one all-reduce each).
 *
 *   sweep 1  srx_backed_stats_tile  per tile: [normalize_total(Row)] [log1p] -> (cnt,sum,sumsq) +=
 *            srx_backed_moments     column statistics so far (compute_number / compute_sum, Column)
 *   select   srx_backed_select      FeatureSelection::HighlyVariable(n) | explicit list | all
 *   sweep 2  srx_backed_gram_tile   per tile, same transform: compaction + Gram tiles +=
 *   solve    srx_backed_solve       PCA (Gram solver) + scores of every tile; srx_backed_fetch  */
typedef struct srx_backed srx_backed;
enum { SRX_BACKED_NORMALIZE = 1, SRX_BACKED_LOG1P = 2 };   /* `transform` bits                    */
int32_t srx_backed_create(srx_ctx* ctx, uint64_t n_cols, int32_t store, srx_backed** out);
void    srx_backed_destroy(srx_backed* b);
/* row_number_out / row_sum_out (NULL or tile->n_rows entries): compute_number / compute_sum in
 * Direction::Row of the tile's RAW values. */
int32_t srx_backed_stats_tile(srx_backed* b, const srx_csr* tile, double target_sum, int32_t transform,
                              uint32_t* row_number_out, double* row_sum_out);
/* per-gene count / sum / sum of squares of the (transformed) values over all tiles so far, summed
 * over ranks; any pointer may be NULL. */
int32_t srx_backed_moments(srx_backed* b, uint64_t* cnt, double* sum, double* sumsq, uint64_t* n_rows_global);
/* n_hvg > 0: HighlyVariable(n_hvg); else `sel` (n_sel indices) or, with sel == NULL, all features
 * (at most 8192 selected).  opts as for srx_pca (solver 2 is refused).  sel_out (room for
 * min(n_hvg, n_cols) or n_sel entries) receives the selection in the reference's order. */
int32_t srx_backed_select(srx_backed* b, uint64_t n_hvg, const uint64_t* sel, uint64_t n_sel,
                          const srx_pca_opts* opts, uint64_t* sel_out, uint64_t* n_out);
int32_t srx_backed_gram_tile(srx_backed* b, const srx_csr* tile, double target_sum, int32_t transform);
int32_t srx_backed_solve(srx_backed* b, srx_pca_info* info);
/* as srx_result_fetch; scores = (rows of this rank, in tile order) x n_pc */
int32_t srx_backed_fetch(srx_backed* b, double* scores, double* components, double* evr, double* mean,
                         double* std_, uint64_t* sel);

/* ---- measurement hooks --------------------------------------------------------------------
 * When enabled, every launch of a kernel class is bracketed by hipEvents on the ctx stream.
 * srx_prof_get returns accumulated device time, launch count and the ALGORITHMIC bytes
 *  the launches moved. */
typedef enum srx_kernel_class {
    SRX_K_NORMALIZE = 0,   /* fused row-sum + scale + log1p                               */
    SRX_K_MOMENTS = 1,     /* gene-tiled (cnt, sum, sumsq)                                */
    SRX_K_COMPACT = 2,     /* HVG compaction                                              */
    SRX_K_SPMM_FWD = 3,    /* Y = X_sel P - 1 c^T                                         */
    SRX_K_SPMM_T = 4,      /* T = X_sel^T Y                                               */
    SRX_K_GRAM = 5,        /* G = X_sel^T X_sel (sparse outer products into LDS tiles)     */
    SRX_K_DENSE = 6,       /* dense k x k times k x 64 products of the Gram solver        */
    SRX_K_ROWSUM = 7,      /* per-cell sums of the raw values (first pass of srx_pipeline)  */
    SRX_K_ITERATE = 8,     /* the k x 64 subspace iteration as a whole (graph replays + the
                              launches between them); contains the SRX_K_DENSE launches     */
    SRX_K_SELECT = 9,      /* device-side HighlyVariable(n): variances, ranks, selection    */
    SRX_K_BUCKET = 10,     /* the Gram kernel's owner records: counts, their read-back, bucket pass;
                              SRX_K_GRAM is the stripe kernel alone                         */
    SRX_K_COUNT_ = 11
} srx_kernel_class;
int32_t srx_prof_enable(srx_ctx* ctx, uint32_t class_mask);
int32_t srx_prof_reset(srx_ctx* ctx);
int32_t srx_prof_get(srx_ctx* ctx, int32_t kernel_class, double* total_ms, uint64_t* launches,
                     double* algorithmic_bytes);
/* Bytes of AUXILIARY structures the launches of a class read or wrote besides their algorithmic bytes (the Gram
 * kernel's owner records and block offsets): this implementation's own, never part of a roofline figure. */
int32_t srx_prof_get_aux(srx_ctx* ctx, int32_t kernel_class, double* aux_bytes);

#ifdef __cplusplus
}
#endif
#endif /* SRX_H */
