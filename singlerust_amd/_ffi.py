"""ctypes binding of libsrx_hip.so (include/srx.h, include/srx_synth.h).

This is plumbing: it loads the in-tree HIP library and declares its C ABI.  There is no
CPU fallback — if the library is missing, or no GPU is visible when a compute call is made,
the call raises (``SrxError``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsrx_hip.so")

# srx_status
OK, E_ARG, E_DTYPE, E_FORMAT, E_BOUNDS, E_HIP, E_RCCL, E_OOM, E_NAN, E_SHAPE, E_NOCONV = (
    0, -1, -2, -3, -4, -5, -6, -7, -8, -9, -10)
STATUS_NAMES = {0: "SRX_OK", -1: "SRX_E_ARG", -2: "SRX_E_DTYPE", -3: "SRX_E_FORMAT",
                -4: "SRX_E_BOUNDS", -5: "SRX_E_HIP", -6: "SRX_E_RCCL", -7: "SRX_E_OOM",
                -8: "SRX_E_NAN", -9: "SRX_E_SHAPE", -10: "SRX_E_NOCONV"}
# srx_dtype
I8, I16, I32, U8, U16, U32, F32, F64 = range(8)
ROW, COLUMN = 0, 1
STORE_AUTO, STORE_F32, STORE_F64 = 0, 1, 2
K_NORMALIZE, K_MOMENTS, K_COMPACT, K_SPMM_FWD, K_SPMM_T, K_GRAM, K_DENSE, K_ROWSUM, K_ITERATE, K_SELECT, K_BUCKET = range(11)
SOLVER_AUTO, SOLVER_GRAM, SOLVER_SPMM = 0, 1, 2
UNIQUE_ID_BYTES = 128


class SrxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{STATUS_NAMES.get(code, code)}: {msg}")
        self.code = code
        self.msg = msg


class Csr(C.Structure):
    _fields_ = [("n_rows", C.c_uint64), ("n_cols", C.c_uint64), ("nnz", C.c_uint64),
                ("indptr", C.c_void_p), ("indices", C.c_void_p), ("values", C.c_void_p),
                ("dtype", C.c_int32)]


class MatInfo(C.Structure):
    _fields_ = [("n_rows", C.c_uint64), ("n_cols", C.c_uint64), ("nnz", C.c_uint64),
                ("dtype", C.c_int32), ("store", C.c_int32), ("row_offset", C.c_uint64),
                ("n_rows_global", C.c_uint64)]


class PcaOpts(C.Structure):
    _fields_ = [("n_components", C.c_int32), ("center", C.c_int32), ("scale", C.c_int32),
                ("n_threads", C.c_int32), ("block", C.c_int32), ("max_iter", C.c_int32),
                ("solver", C.c_int32), ("tol", C.c_double), ("seed", C.c_uint64)]


class PcaInfo(C.Structure):
    _fields_ = [("n_cells_global", C.c_uint64), ("k", C.c_uint32), ("n_pc", C.c_uint32),
                ("block", C.c_uint32), ("n_iter", C.c_uint32), ("residual", C.c_double),
                ("nnz_selected", C.c_uint64), ("solver", C.c_uint32), ("reserved_", C.c_uint32)]


class PipelineResult(C.Structure):
    _fields_ = [("pca", PcaInfo), ("ms_normalize", C.c_double), ("ms_moments", C.c_double),
                ("ms_select", C.c_double), ("ms_compact", C.c_double), ("ms_pca", C.c_double)]


class SynthParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_rows_global", C.c_uint64), ("n_cols", C.c_uint64),
                ("density", C.c_double), ("lib_sigma", C.c_double), ("type_decay", C.c_double),
                ("n_types", C.c_uint32), ("marker_genes", C.c_uint32), ("expr_boost", C.c_uint32),
                ("value_boost", C.c_uint32), ("skew", C.c_uint32), ("reserved_", C.c_uint32)]


class Flex(C.Structure):            # srx_flex: FlexValue (src/shared/mod.rs:62-66)
    _fields_ = [("kind", C.c_int32), ("absolute", C.c_uint32), ("relative", C.c_double)]


HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_double), C.c_uint64)
FLEX_NONE, FLEX_ABSOLUTE, FLEX_RELATIVE = 0, 1, 2
BACKED_NORMALIZE, BACKED_LOG1P = 1, 2
P = C.c_void_p
_SIGS = {
    # name: (restype, argtypes)
    "srx_abi_version": (C.c_int32, []),
    "srx_device_count": (C.c_int32, [C.POINTER(C.c_int32)]),
    "srx_ctx_create": (C.c_int32, [C.c_int32, C.POINTER(P)]),
    "srx_ctx_destroy": (None, [P]),
    "srx_ctx_synchronize": (C.c_int32, [P]),
    "srx_last_error": (C.c_char_p, [P]),
    "srx_comm_unique_id": (C.c_int32, [P]),
    "srx_comm_init": (C.c_int32, [P, C.c_int32, C.c_int32, P]),
    "srx_comm_init_host": (C.c_int32, [P, C.c_int32, C.c_int32, C.c_void_p, P]),
    "srx_comm_destroy": (C.c_int32, [P]),
    "srx_comm_info": (C.c_int32, [P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "srx_comm_overlap_info": (C.c_int32, [P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "srx_gram_mode_info": (C.c_int32, [P, C.POINTER(C.c_int32)]),
    "srx_gram_exchange_ranges": (C.c_int32, [C.c_uint64, P]),
    "srx_partition_rows": (C.c_int32, [P, C.c_uint64, C.c_int32, P]),
    "srx_matrix_upload": (C.c_int32, [P, C.POINTER(Csr), C.c_int32, C.POINTER(P)]),
    "srx_matrix_upload_csc": (C.c_int32, [P, C.POINTER(Csr), C.c_int32, C.POINTER(P)]),
    "srx_matrix_format": (C.c_int32, [P, C.POINTER(C.c_int32)]),
    "srx_matrix_to_csr": (C.c_int32, [P, C.POINTER(P)]),
    "srx_matrix_to_csc": (C.c_int32, [P, C.POINTER(P)]),
    "srx_matrix_alloc": (C.c_int32, [P, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32, C.c_int32, C.POINTER(P)]),
    "srx_matrix_device_ptrs": (C.c_int32, [P, C.POINTER(P), C.POINTER(P), C.POINTER(P)]),
    "srx_matrix_info": (C.c_int32, [P, C.POINTER(MatInfo)]),
    "srx_matrix_set_shard": (C.c_int32, [P, C.c_uint64]),
    "srx_matrix_download_values": (C.c_int32, [P, P, C.c_int32]),
    "srx_matrix_prepare": (C.c_int32, [P]),
    "srx_matrix_reserve_results": (C.c_int32, [P, C.c_uint64, C.c_int32]),
    "srx_matrix_clone": (C.c_int32, [P, C.POINTER(P)]),
    "srx_matrix_copy_values": (C.c_int32, [P, P]),
    "srx_matrix_free": (None, [P]),
    "srx_compute_number": (C.c_int32, [P, C.c_int32, P]),
    "srx_compute_sum": (C.c_int32, [P, C.c_int32, P]),
    "srx_compute_variance": (C.c_int32, [P, C.c_int32, P]),
    "srx_compute_std_dev": (C.c_int32, [P, C.c_int32, P]),
    "srx_compute_min_max": (C.c_int32, [P, C.c_int32, P, P]),
    "srx_compute_qc_variables": (C.c_int32, [P, P, P, P, P, P, P, P, P]),
    "srx_filter_cells": (C.c_int32, [P, Flex, Flex, C.POINTER(C.c_void_p), P]),
    "srx_filter_genes": (C.c_int32, [P, Flex, Flex, C.POINTER(C.c_void_p), P]),
    "srx_subset": (C.c_int32, [P, P, P, C.POINTER(C.c_void_p)]),
    "srx_matrix_download_pattern": (C.c_int32, [P, P, P]),
    "srx_gene_moments": (C.c_int32, [P, P, P, P]),
    "srx_normalize_total_inplace": (C.c_int32, [P, C.c_double, C.c_int32]),
    "srx_log1p_inplace": (C.c_int32, [P]),
    "srx_normalize_log1p_inplace": (C.c_int32, [P, C.c_double, P]),
    "srx_select_hvg": (C.c_int32, [P, C.c_uint64, P, C.POINTER(C.c_uint64)]),
    "srx_pca": (C.c_int32, [P, P, C.c_uint64, C.POINTER(PcaOpts), P, P, P, P, P, C.POINTER(PcaInfo)]),
    "srx_pca_loadings": (C.c_int32, [P, P, P, C.c_uint64, C.c_uint64, C.c_uint64, P]),
    "srx_spmm": (C.c_int32, [P, P, C.c_uint64, P, P, P, P]),
    "srx_pipeline": (C.c_int32, [P, C.c_double, C.c_uint64, C.POINTER(PcaOpts), C.POINTER(PipelineResult)]),
    "srx_result_fetch": (C.c_int32, [P, P, P, P, P, P, P]),
    "srx_backed_create": (C.c_int32, [P, C.c_uint64, C.c_int32, C.POINTER(P)]),
    "srx_backed_destroy": (None, [P]),
    "srx_backed_stats_tile": (C.c_int32, [P, C.POINTER(Csr), C.c_double, C.c_int32, P, P]),
    "srx_backed_moments": (C.c_int32, [P, P, P, P, C.POINTER(C.c_uint64)]),
    "srx_backed_select": (C.c_int32, [P, C.c_uint64, P, C.c_uint64, C.POINTER(PcaOpts), P, C.POINTER(C.c_uint64)]),
    "srx_backed_gram_tile": (C.c_int32, [P, C.POINTER(Csr), C.c_double, C.c_int32]),
    "srx_backed_solve": (C.c_int32, [P, C.POINTER(PcaInfo)]),
    "srx_backed_fetch": (C.c_int32, [P, P, P, P, P, P, P]),
    "srx_prof_enable": (C.c_int32, [P, C.c_uint32]),
    "srx_prof_reset": (C.c_int32, [P]),
    "srx_prof_get": (C.c_int32, [P, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    "srx_prof_get_aux": (C.c_int32, [P, C.c_int32, C.POINTER(C.c_double)]),
    # srx_synth.h
    "srx_synth_defaults": (None, [C.POINTER(SynthParams), C.c_uint64, C.c_uint64, C.c_uint64, C.c_double]),
    "srx_synth_indptr": (C.c_int32, [C.POINTER(SynthParams), C.c_uint64, C.c_uint64, P]),
    "srx_synth_fill_host": (C.c_int32, [C.POINTER(SynthParams), C.c_uint64, C.c_uint64, P, P, P]),
    "srx_synth_generate": (C.c_int32, [P, C.POINTER(SynthParams), C.c_uint64, C.c_uint64, C.c_int32, C.c_int32, C.POINTER(P)]),
}
EXPORTS = tuple(_SIGS)

_lib = None


def lib() -> C.CDLL:
    """Load libsrx_hip.so (built in-tree by singlerust_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SrxError(E_HIP, f"{LIB_PATH} is missing: run `python -m singlerust_amd.build` "
                                  "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int, ctx=None) -> None:
    if rc != OK:
        msg = lib().srx_last_error(ctx)
        raise SrxError(rc, msg.decode() if msg else "")


def ptr(a):
    """void* of a numpy array (or None)."""
    return None if a is None else C.c_void_p(a.ctypes.data)
