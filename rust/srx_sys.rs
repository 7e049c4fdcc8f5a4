// rust/srx_sys.rs — GENERATED from include/srx.h by scripts/gen_rust_bindings.py; do not edit.
// `extern "C"` declarations of libsrx_hip.so for the shim of INTEGRATION.md (src/gpu/ffi.rs in a SingleRust
// checkout, behind a cargo feature).  Not compile-checked here: the build image has no rustc.  The hand-written
// part of the binding (the `upload` helper and the replaced function bodies) is rust/shim.rs.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_void};

pub const SRX_ABI_VERSION: i32 = 6;
pub const SRX_UNIQUE_ID_BYTES: usize = 128;
pub const SRX_OK: i32 = 0;
pub const SRX_E_ARG: i32 = -1;
pub const SRX_E_DTYPE: i32 = -2;
pub const SRX_E_FORMAT: i32 = -3;
pub const SRX_E_BOUNDS: i32 = -4;
pub const SRX_E_HIP: i32 = -5;
pub const SRX_E_RCCL: i32 = -6;
pub const SRX_E_OOM: i32 = -7;
pub const SRX_E_NAN: i32 = -8;
pub const SRX_E_SHAPE: i32 = -9;
pub const SRX_E_NOCONV: i32 = -10;
pub const SRX_I8: i32 = 0;
pub const SRX_I16: i32 = 1;
pub const SRX_I32: i32 = 2;
pub const SRX_U8: i32 = 3;
pub const SRX_U16: i32 = 4;
pub const SRX_U32: i32 = 5;
pub const SRX_F32: i32 = 6;
pub const SRX_F64: i32 = 7;
pub const SRX_ROW: i32 = 0;
pub const SRX_COLUMN: i32 = 1;
pub const SRX_STORE_AUTO: i32 = 0;
pub const SRX_STORE_F32: i32 = 1;
pub const SRX_STORE_F64: i32 = 2;
pub const SRX_FORMAT_CSR: i32 = 0;
pub const SRX_FORMAT_CSC: i32 = 1;
pub const SRX_FLEX_NONE: i32 = 0;
pub const SRX_FLEX_ABSOLUTE: i32 = 1;
pub const SRX_FLEX_RELATIVE: i32 = 2;
pub const SRX_SOLVER_AUTO: i32 = 0;
pub const SRX_SOLVER_GRAM: i32 = 1;
pub const SRX_SOLVER_SPMM: i32 = 2;
pub const SRX_BACKED_NORMALIZE: i32 = 1;
pub const SRX_BACKED_LOG1P: i32 = 2;
pub const SRX_K_NORMALIZE: i32 = 0;
pub const SRX_K_MOMENTS: i32 = 1;
pub const SRX_K_COMPACT: i32 = 2;
pub const SRX_K_SPMM_FWD: i32 = 3;
pub const SRX_K_SPMM_T: i32 = 4;
pub const SRX_K_GRAM: i32 = 5;
pub const SRX_K_DENSE: i32 = 6;
pub const SRX_K_ROWSUM: i32 = 7;
pub const SRX_K_ITERATE: i32 = 8;
pub const SRX_K_SELECT: i32 = 9;
pub const SRX_K_BUCKET: i32 = 10;
pub const SRX_K_COUNT_: i32 = 11;

#[repr(C)] pub struct SrxCtx { _private: [u8; 0] }      // opaque `srx_ctx`
#[repr(C)] pub struct SrxMat { _private: [u8; 0] }      // opaque `srx_mat`
#[repr(C)] pub struct SrxBacked { _private: [u8; 0] }      // opaque `srx_backed`

#[repr(C)]
#[derive(Clone, Copy)]
pub struct SrxCsr {      // `srx_csr`
    pub n_rows: u64,
    pub n_cols: u64,
    pub nnz: u64,
    pub indptr: *const u64,
    pub indices: *const u64,
    pub values: *mut c_void,
    pub dtype: i32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct SrxMatInfo {      // `srx_mat_info`
    pub n_rows: u64,
    pub n_cols: u64,
    pub nnz: u64,
    pub dtype: i32,
    pub store: i32,
    pub row_offset: u64,
    pub n_rows_global: u64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct SrxFlex {      // `srx_flex`
    pub kind: i32,
    pub absolute: u32,
    pub relative: f64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct SrxPcaOpts {      // `srx_pca_opts`
    pub n_components: i32,
    pub center: i32,
    pub scale: i32,
    pub n_threads: i32,
    pub block: i32,
    pub max_iter: i32,
    pub solver: i32,
    pub tol: f64,
    pub seed: u64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct SrxPcaInfo {      // `srx_pca_info`
    pub n_cells_global: u64,
    pub k: u32,
    pub n_pc: u32,
    pub block: u32,
    pub n_iter: u32,
    pub residual: f64,
    pub nnz_selected: u64,
    pub solver: u32,
    pub reserved: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct SrxPipelineResult {      // `srx_pipeline_result`
    pub pca: SrxPcaInfo,
    pub ms_normalize: f64,
    pub ms_moments: f64,
    pub ms_select: f64,
    pub ms_compact: f64,
    pub ms_pca: f64,
}

pub type SrxHostAllreduceFn = extern "C" fn(user: *mut c_void, buf: *mut f64, count: u64) -> i32;      // `srx_host_allreduce_fn`

#[link(name = "srx_hip")]
extern "C" {
    pub fn srx_abi_version() -> i32;
    pub fn srx_device_count(n_out: *mut i32) -> i32;
    pub fn srx_ctx_create(device_id: i32, out: *mut *mut SrxCtx) -> i32;
    pub fn srx_ctx_destroy(ctx: *mut SrxCtx);
    pub fn srx_ctx_synchronize(ctx: *mut SrxCtx) -> i32;
    pub fn srx_last_error(ctx: *const SrxCtx) -> *const c_char;
    pub fn srx_comm_unique_id(id_out_128: *mut c_void) -> i32;
    pub fn srx_comm_init(ctx: *mut SrxCtx, n_ranks: i32, rank: i32, id_128: *const c_void) -> i32;
    pub fn srx_comm_init_host(ctx: *mut SrxCtx, n_ranks: i32, rank: i32, fn_: SrxHostAllreduceFn, user: *mut c_void) -> i32;
    pub fn srx_comm_destroy(ctx: *mut SrxCtx) -> i32;
    pub fn srx_comm_info(ctx: *mut SrxCtx, kind_out: *mut i32, n_ranks_out: *mut i32, rccl_version_out: *mut i32,
                         ranks_seen_out: *mut i32) -> i32;
    pub fn srx_comm_overlap_info(ctx: *mut SrxCtx, split_exchanges_out: *mut i32, cu_masked_out: *mut i32) -> i32;
    pub fn srx_gram_mode_info(ctx: *mut SrxCtx, mode_out: *mut i32) -> i32;
    pub fn srx_gram_exchange_ranges(k: u64, offsets_out: *mut u64) -> i32;
    pub fn srx_partition_rows(indptr: *const u64, n_rows: u64, n_ranks: i32, cut_out: *mut u64) -> i32;
    pub fn srx_matrix_upload(ctx: *mut SrxCtx, host: *const SrxCsr, store: i32, out: *mut *mut SrxMat) -> i32;
    pub fn srx_matrix_upload_csc(ctx: *mut SrxCtx, host: *const SrxCsr, store: i32, out: *mut *mut SrxMat) -> i32;
    pub fn srx_matrix_format(m: *const SrxMat, format_out: *mut i32) -> i32;
    pub fn srx_matrix_to_csr(m: *mut SrxMat, out: *mut *mut SrxMat) -> i32;
    pub fn srx_matrix_to_csc(m: *mut SrxMat, out: *mut *mut SrxMat) -> i32;
    pub fn srx_matrix_alloc(ctx: *mut SrxCtx, n_rows: u64, n_cols: u64, nnz: u64, dtype: i32, store: i32,
                            out: *mut *mut SrxMat) -> i32;
    pub fn srx_matrix_device_ptrs(m: *mut SrxMat, indptr: *mut *mut c_void, indices: *mut *mut c_void,
                                  values: *mut *mut c_void) -> i32;
    pub fn srx_matrix_info(m: *const SrxMat, out: *mut SrxMatInfo) -> i32;
    pub fn srx_matrix_set_shard(m: *mut SrxMat, row_offset: u64) -> i32;
    pub fn srx_matrix_download_values(m: *mut SrxMat, values_out: *mut c_void, dtype_out: i32) -> i32;
    pub fn srx_matrix_prepare(m: *mut SrxMat) -> i32;
    pub fn srx_matrix_reserve_results(m: *mut SrxMat, n_selected: u64, n_components: i32) -> i32;
    pub fn srx_matrix_clone(m: *mut SrxMat, out: *mut *mut SrxMat) -> i32;
    pub fn srx_matrix_copy_values(dst: *mut SrxMat, src: *const SrxMat) -> i32;
    pub fn srx_matrix_free(m: *mut SrxMat);
    pub fn srx_compute_number(m: *mut SrxMat, direction: i32, out: *mut u32) -> i32;
    pub fn srx_compute_sum(m: *mut SrxMat, direction: i32, out: *mut f64) -> i32;
    pub fn srx_compute_variance(m: *mut SrxMat, direction: i32, out: *mut f64) -> i32;
    pub fn srx_compute_std_dev(m: *mut SrxMat, direction: i32, out: *mut f64) -> i32;
    pub fn srx_compute_min_max(m: *mut SrxMat, direction: i32, min_out: *mut f64, max_out: *mut f64) -> i32;
    pub fn srx_compute_qc_variables(m: *mut SrxMat, num_per_cell: *mut u32, num_per_gene: *mut u32,
                                    expr_per_gene: *mut f64, expr_per_cell: *mut f64, variance_per_gene: *mut f64,
                                    variance_per_cell: *mut f64, std_dev_per_cell: *mut f64,
                                    std_dev_per_gene: *mut f64) -> i32;
    pub fn srx_gene_moments(m: *mut SrxMat, cnt: *mut u64, sum: *mut f64, sumsq: *mut f64) -> i32;
    pub fn srx_filter_cells(m: *mut SrxMat, lower: SrxFlex, upper: SrxFlex, out: *mut *mut SrxMat, mask_out: *mut u8) -> i32;
    pub fn srx_filter_genes(m: *mut SrxMat, lower: SrxFlex, upper: SrxFlex, out: *mut *mut SrxMat, mask_out: *mut u8) -> i32;
    pub fn srx_subset(m: *mut SrxMat, row_mask: *const u8, col_mask: *const u8, out: *mut *mut SrxMat) -> i32;
    pub fn srx_matrix_download_pattern(m: *mut SrxMat, indptr_out: *mut u64, indices_out: *mut u64) -> i32;
    pub fn srx_normalize_total_inplace(m: *mut SrxMat, target_sum: f64, direction: i32) -> i32;
    pub fn srx_log1p_inplace(m: *mut SrxMat) -> i32;
    pub fn srx_normalize_log1p_inplace(m: *mut SrxMat, target_sum: f64, row_sums_out: *mut f64) -> i32;
    pub fn srx_select_hvg(m: *mut SrxMat, n: u64, idx_out: *mut u64, n_out: *mut u64) -> i32;
    pub fn srx_pca(m: *mut SrxMat, sel: *const u64, k: u64, opts: *const SrxPcaOpts, scores: *mut f64,
                   components: *mut f64, evr: *mut f64, mean: *mut f64, std: *mut f64, info: *mut SrxPcaInfo) -> i32;
    pub fn srx_pca_loadings(components: *const f64, std: *const f64, sel: *const u64, k: u64, n_pc: u64,
                            n_vars: u64, out: *mut f64) -> i32;
    pub fn srx_spmm(m: *mut SrxMat, sel: *const u64, k: u64, panel: *const f64, y_out: *mut f64, t_out: *mut f64,
                    gram_out: *mut f64) -> i32;
    pub fn srx_pipeline(m: *mut SrxMat, target_sum: f64, n_hvg: u64, opts: *const SrxPcaOpts,
                        res: *mut SrxPipelineResult) -> i32;
    pub fn srx_result_fetch(m: *mut SrxMat, scores: *mut f64, components: *mut f64, evr: *mut f64, mean: *mut f64,
                            std: *mut f64, hvg_idx: *mut u64) -> i32;
    pub fn srx_backed_create(ctx: *mut SrxCtx, n_cols: u64, store: i32, out: *mut *mut SrxBacked) -> i32;
    pub fn srx_backed_destroy(b: *mut SrxBacked);
    pub fn srx_backed_stats_tile(b: *mut SrxBacked, tile: *const SrxCsr, target_sum: f64, transform: i32,
                                 row_number_out: *mut u32, row_sum_out: *mut f64) -> i32;
    pub fn srx_backed_moments(b: *mut SrxBacked, cnt: *mut u64, sum: *mut f64, sumsq: *mut f64,
                              n_rows_global: *mut u64) -> i32;
    pub fn srx_backed_select(b: *mut SrxBacked, n_hvg: u64, sel: *const u64, n_sel: u64, opts: *const SrxPcaOpts,
                             sel_out: *mut u64, n_out: *mut u64) -> i32;
    pub fn srx_backed_gram_tile(b: *mut SrxBacked, tile: *const SrxCsr, target_sum: f64, transform: i32) -> i32;
    pub fn srx_backed_solve(b: *mut SrxBacked, info: *mut SrxPcaInfo) -> i32;
    pub fn srx_backed_fetch(b: *mut SrxBacked, scores: *mut f64, components: *mut f64, evr: *mut f64,
                            mean: *mut f64, std: *mut f64, sel: *mut u64) -> i32;
    pub fn srx_prof_enable(ctx: *mut SrxCtx, class_mask: u32) -> i32;
    pub fn srx_prof_reset(ctx: *mut SrxCtx) -> i32;
    pub fn srx_prof_get(ctx: *mut SrxCtx, kernel_class: i32, total_ms: *mut f64, launches: *mut u64,
                        algorithmic_bytes: *mut f64) -> i32;
    pub fn srx_prof_get_aux(ctx: *mut SrxCtx, kernel_class: i32, aux_bytes: *mut f64) -> i32;
}
