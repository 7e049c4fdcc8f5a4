"""CPU parity oracle for the SingleRust hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package; the product (``singlerust_amd``) never does.

* ``srx_oracle.c`` — serial C restatement of the reference loops
  (src/shared/statistics/helper/csr.rs, src/memory/processing/scale/mod.rs,
  src/memory/processing/transform/mod.rs, src/memory/processing/dim_red/mod.rs:135-140,
  src/shared/mod.rs:230-259), loaded here through ctypes.
* ``pca_oracle.py`` — numpy restatement of src/shared/processing/pca/mod.rs:74-215
  (PARITY UNPINNED: the live PCA is in the un-vendored crate single_algebra 0.1.0-alpha.3
  and the reference's tests assert nothing numeric about it).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liborc.so")

I8, I16, I32, U8, U16, U32, F32, F64 = range(8)
ROW, COLUMN = 0, 1

NP_DTYPES = {
    I8: np.int8, I16: np.int16, I32: np.int32, U8: np.uint8, U16: np.uint16,
    U32: np.uint32, F32: np.float32, F64: np.float64,
}
DTYPE_CODE = {np.dtype(v): k for k, v in NP_DTYPES.items()}


_OMP_PATH = os.path.join(_HERE, "liborc_omp.so")


def build(force: bool = False) -> str:
    """Compile liborc.so (and the threaded bench baseline liborc_omp.so) with gcc (recipe: oracle/Makefile)."""
    for target, srcname in (("liborc.so", "srx_oracle.c"), ("liborc_omp.so", "omp_baseline.c")):
        path, src = os.path.join(_HERE, target), os.path.join(_HERE, srcname)
        if force or not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-B", target], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def omp_pipeline(m, target_sum: float, n_hvg: int, n_threads: int):
    """Threaded CPU baseline (omp_baseline.c; SURVEY.md 8(d) variant (ii)): normalise + log1p, per-gene moments,
    HighlyVariable(n_hvg), k x k standardised covariance of the selection.  Returns (values_f64, hvg (rank order),
    order (ascending genes), cov, mean, sd, seconds[3])."""
    build()
    lo = ctypes.CDLL(_OMP_PATH)
    k = min(n_hvg, m.n_cols)
    vals = np.ascontiguousarray(m.values, dtype=np.float32)
    out = np.empty(len(vals), np.float64)
    hv, order = np.zeros(k, np.uint64), np.zeros(k, np.uint64)
    cov, mean, sd, secs = np.zeros((k, k)), np.zeros(k), np.zeros(k), np.zeros(4)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lo.orc_omp_pipeline.restype = ctypes.c_int
    lo.orc_omp_pipeline.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_double, ctypes.c_uint64, ctypes.c_int] + [ctypes.c_void_p] * 7
    ip = np.ascontiguousarray(m.indptr, dtype=np.uint64)
    ix = np.ascontiguousarray(m.indices, dtype=np.uint64)
    rc = lo.orc_omp_pipeline(m.n_rows, m.n_cols, vp(ip), vp(ix), vp(vals), target_sum, n_hvg, n_threads, vp(out), vp(hv),
                             vp(order), vp(cov), vp(mean), vp(sd), vp(secs))
    if rc != 0:
        raise MemoryError("orc_omp_pipeline: out of memory")
    return out, hv, order, cov, mean, sd, secs


def omp_cov_selected(m, sel, n_threads: int):
    """Standardised covariance Z^T Z (k x k, f64) of the selection `sel` (slot i = feature sel[i]) of the f64 matrix `m`, mean and
    sd (ddof 0) per slot — omp_baseline.c::orc_omp_cov_selected: the independent eigen-reference of a PCA at sizes where the
    exact SVD of the densified N x k matrix (pca_oracle) does not fit a test."""
    build()
    lo = ctypes.CDLL(_OMP_PATH)
    sel = np.ascontiguousarray(sel, dtype=np.uint64)
    k = len(sel)
    vals = np.ascontiguousarray(m.values, dtype=np.float64)
    ip = np.ascontiguousarray(m.indptr, dtype=np.uint64)
    ix = np.ascontiguousarray(m.indices, dtype=np.uint64)
    cov, mean, sd = np.zeros((k, k)), np.zeros(k), np.zeros(k)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lo.orc_omp_cov_selected.restype = ctypes.c_int
    lo.orc_omp_cov_selected.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p]
    rc = lo.orc_omp_cov_selected(m.n_rows, m.n_cols, vp(ip), vp(ix), vp(vals), vp(sel), k, n_threads, vp(cov), vp(mean), vp(sd))
    if rc == -1:
        raise MemoryError("orc_omp_cov_selected: out of memory")
    if rc != 0:
        raise ValueError("orc_omp_cov_selected: a feature is selected twice or out of range")
    return cov, mean, sd


def omp_scores(m, values_f64, order, pv, cvec, n_threads: int):
    """scores = Z V over all rows on `n_threads` OpenMP threads (omp_baseline.c::orc_omp_scores): `order` = the selected
    features (slot i = feature order[i]), pv = V / sd[:, None] (k x n_pc), cvec = (mean / sd) @ V."""
    build()
    lo = ctypes.CDLL(_OMP_PATH)
    order = np.ascontiguousarray(order, dtype=np.uint64)
    pv = np.ascontiguousarray(pv, dtype=np.float64)
    cvec = np.ascontiguousarray(cvec, dtype=np.float64)
    k, n_pc = pv.shape
    ip = np.ascontiguousarray(m.indptr, dtype=np.uint64)
    ix = np.ascontiguousarray(m.indices, dtype=np.uint64)
    vals = np.ascontiguousarray(values_f64, dtype=np.float64)
    scores = np.empty((m.n_rows, n_pc))
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lo.orc_omp_scores.restype = ctypes.c_int
    lo.orc_omp_scores.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                  ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                                  ctypes.c_int, ctypes.c_void_p]
    if lo.orc_omp_scores(m.n_rows, m.n_cols, vp(ip), vp(ix), vp(vals), vp(order), k, vp(pv), vp(cvec), n_pc, n_threads,
                         vp(scores)) != 0:
        raise MemoryError("orc_omp_scores: out of memory")
    return scores


class _Csr(ctypes.Structure):
    _fields_ = [
        ("n_rows", ctypes.c_uint64), ("n_cols", ctypes.c_uint64), ("nnz", ctypes.c_uint64),
        ("indptr", ctypes.c_void_p), ("indices", ctypes.c_void_p), ("values", ctypes.c_void_p),
        ("dtype", ctypes.c_int32),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


class Csr:
    """Reference-layout CSR: u64 indptr / indices, typed values (numpy arrays)."""

    def __init__(self, n_rows, n_cols, indptr, indices, values):
        self.n_rows, self.n_cols = int(n_rows), int(n_cols)
        self.indptr = np.ascontiguousarray(indptr, dtype=np.uint64)
        self.indices = np.ascontiguousarray(indices, dtype=np.uint64)
        self.values = np.ascontiguousarray(values)
        assert self.values.dtype in DTYPE_CODE, f"unsupported dtype {self.values.dtype}"
        assert self.indptr.shape == (self.n_rows + 1,)
        assert self.indices.shape == self.values.shape
        self.nnz = int(self.values.shape[0])

    @property
    def dtype_code(self):
        return DTYPE_CODE[self.values.dtype]

    def c(self):
        return _Csr(self.n_rows, self.n_cols, self.nnz, self.indptr.ctypes.data,
                    self.indices.ctypes.data, self.values.ctypes.data, self.dtype_code)

    def with_values(self, values):
        return Csr(self.n_rows, self.n_cols, self.indptr, self.indices, values)

    def _n(self, direction):
        return self.n_rows if direction == ROW else self.n_cols


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed rc={rc}")


def compute_number(m: Csr, direction: int) -> np.ndarray:
    out = np.zeros(m._n(direction), dtype=np.uint32)
    c = m.c()
    _check(lib().orc_number(ctypes.byref(c), direction, _p(out)), "number")
    return out


def compute_sum(m: Csr, direction: int) -> np.ndarray:
    out = np.zeros(m._n(direction), dtype=np.float64)
    c = m.c()
    _check(lib().orc_sum(ctypes.byref(c), direction, _p(out)), "sum")
    return out


def compute_variance(m: Csr, direction: int) -> np.ndarray:
    out = np.zeros(m._n(direction), dtype=np.float64)
    c = m.c()
    _check(lib().orc_variance(ctypes.byref(c), direction, _p(out)), "variance")
    return out


def compute_std_dev(m: Csr, direction: int) -> np.ndarray:
    out = np.zeros(m._n(direction), dtype=np.float64)
    c = m.c()
    _check(lib().orc_std_dev(ctypes.byref(c), direction, _p(out)), "std_dev")
    return out


def compute_min_max(m: Csr, direction: int):
    mn = np.zeros(m._n(direction), dtype=np.float64)
    mx = np.zeros(m._n(direction), dtype=np.float64)
    c = m.c()
    _check(lib().orc_min_max(ctypes.byref(c), direction, _p(mn), _p(mx)), "min_max")
    return mn, mx


def gene_moments(m: Csr):
    cnt = np.zeros(m.n_cols, dtype=np.uint64)
    s = np.zeros(m.n_cols, dtype=np.float64)
    sq = np.zeros(m.n_cols, dtype=np.float64)
    c = m.c()
    _check(lib().orc_gene_moments(ctypes.byref(c), _p(cnt), _p(s), _p(sq)), "gene_moments")
    return cnt, s, sq


def normalize_total(m: Csr, target_sum: float, direction: int) -> Csr:
    """Returns a new F64 Csr (the reference turns X into DynCsrMatrix::F64)."""
    out = np.zeros(m.nnz, dtype=np.float64)
    c = m.c()
    _check(lib().orc_normalize_total(ctypes.byref(c), ctypes.c_double(target_sum), direction,
                                     _p(out)), "normalize_total")
    return m.with_values(out)


def log1p_transform(m: Csr) -> Csr:
    """F32 stays F32, everything else becomes F64 (transform/mod.rs:36-57)."""
    out = np.zeros(m.nnz, dtype=np.float32 if m.values.dtype == np.float32 else np.float64)
    c = m.c()
    _check(lib().orc_log1p(ctypes.byref(c), _p(out)), "log1p")
    return m.with_values(out)


def select_hvg(variances: np.ndarray, n: int) -> np.ndarray:
    v = np.ascontiguousarray(variances, dtype=np.float64)
    out = np.zeros(min(int(n), v.shape[0]), dtype=np.uint64)
    n_out = ctypes.c_uint64(0)
    rc = lib().orc_select_hvg(_p(v), ctypes.c_uint64(v.shape[0]), ctypes.c_uint64(int(n)),
                              _p(out) if out.size else None, ctypes.byref(n_out))
    if rc == -2:
        raise ValueError("NaN variance: reference partial_cmp().unwrap() panics")
    _check(rc, "select_hvg")
    return out[: n_out.value]


def densify_selected(m: Csr, sel) -> np.ndarray:
    sel = np.ascontiguousarray(sel, dtype=np.uint64)
    dense = np.zeros((m.n_rows, sel.shape[0]), dtype=np.float64)
    c = m.c()
    _check(lib().orc_densify_selected(ctypes.byref(c), _p(sel), ctypes.c_uint64(sel.shape[0]),
                                      _p(dense)), "densify_selected")
    return dense
