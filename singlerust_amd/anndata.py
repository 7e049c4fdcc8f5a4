"""Host-side mirror of the reference's data model for the hot path.

``IMAnnData`` here is the minimum of ``anndata_memory::IMAnnData`` the path touches: ``X``
(an AnnData-shaped CSR, device-resident in HBM as an ``srx_mat``), ``obs_names`` /
``var_names``, ``obsm`` and ``varm``.  Enums follow src/shared/mod.rs:17-60.
"""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass

import numpy as np

from . import _ffi as F


class Direction(enum.IntEnum):          # src/shared/mod.rs:39-42
    Row = 0
    Column = 1

    ROW = 0
    COLUMN = 1

    def is_row(self) -> bool:           # src/shared/mod.rs:53-59
        return self == Direction.Row


class FeatureSelection:                 # src/shared/mod.rs:17-23
    @dataclass(frozen=True)
    class HighlyVariableCol:
        col: str

    @dataclass(frozen=True)
    class HighlyVariable:
        n: int

    @dataclass(frozen=True)
    class Randomized:
        n: int

    @dataclass(frozen=True)
    class VarianceThreshold:
        threshold: float

    @dataclass(frozen=True)
    class NoSelection:
        pass

    None_ = NoSelection()


class FlexValue:                        # src/shared/mod.rs:62-66
    @dataclass(frozen=True)
    class Absolute:
        value: int

    @dataclass(frozen=True)
    class Relative:
        value: float

    @dataclass(frozen=True)
    class NoLimit:
        pass

    None_ = NoLimit()

    @staticmethod
    def to_c(v) -> "F.Flex":
        if v is None or isinstance(v, FlexValue.NoLimit):
            return F.Flex(F.FLEX_NONE, 0, 0.0)
        if isinstance(v, FlexValue.Absolute):
            return F.Flex(F.FLEX_ABSOLUTE, int(v.value), 0.0)
        if isinstance(v, FlexValue.Relative):
            return F.Flex(F.FLEX_RELATIVE, 0, float(v.value))
        raise TypeError(f"not a FlexValue: {v!r}")


NP_OF_DTYPE = {F.I8: np.int8, F.I16: np.int16, F.I32: np.int32, F.U8: np.uint8, F.U16: np.uint16,
               F.U32: np.uint32, F.F32: np.float32, F.F64: np.float64}
DTYPE_OF_NP = {np.dtype(v): k for k, v in NP_OF_DTYPE.items()}


class Context:
    """One GPU + stream (+ RCCL communicator): an ``srx_ctx``."""

    _default = None

    def __init__(self, device_id: int = 0):
        self._h = C.c_void_p()
        F.check(F.lib().srx_ctx_create(int(device_id), C.byref(self._h)))
        self.device_id = int(device_id)
        self.n_ranks, self.rank = 1, 0

    @classmethod
    def default(cls) -> "Context":
        if cls._default is None:
            cls._default = Context(0)
        return cls._default

    @property
    def handle(self):
        return self._h

    def synchronize(self) -> None:
        F.check(F.lib().srx_ctx_synchronize(self._h), self._h)

    def comm_init(self, n_ranks: int, rank: int, unique_id: bytes) -> None:
        buf = (C.c_char * F.UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        F.check(F.lib().srx_comm_init(self._h, n_ranks, rank, buf), self._h)
        self.n_ranks, self.rank = n_ranks, rank

    def comm_init_host(self, n_ranks: int, rank: int, allreduce) -> None:
        """srx_comm_init_host: ``allreduce(np.ndarray[f64])`` sums the array over all ranks in place."""
        def _cb(_user, buf, count):
            try:
                allreduce(np.ctypeslib.as_array(buf, shape=(int(count),)))
                return 0
            except Exception:            # never unwind through the C frames
                return 1
        self._host_cb = F.HOST_ALLREDUCE_FN(_cb)      # kept alive with the context
        F.check(F.lib().srx_comm_init_host(self._h, n_ranks, rank, C.cast(self._host_cb, C.c_void_p), None), self._h)
        self.n_ranks, self.rank = n_ranks, rank

    @staticmethod
    def comm_unique_id() -> bytes:
        buf = (C.c_char * F.UNIQUE_ID_BYTES)()
        F.check(F.lib().srx_comm_unique_id(buf))
        return bytes(buf)

    def prof_enable(self, mask: int) -> None:
        F.check(F.lib().srx_prof_enable(self._h, mask), self._h)

    def prof_reset(self) -> None:
        F.check(F.lib().srx_prof_reset(self._h), self._h)

    def prof_get(self, cls_: int):
        ms, n, b = C.c_double(), C.c_uint64(), C.c_double()
        F.check(F.lib().srx_prof_get(self._h, cls_, C.byref(ms), C.byref(n), C.byref(b)), self._h)
        return ms.value, n.value, b.value

    def prof_get_aux(self, cls_: int) -> float:
        b = C.c_double()
        F.check(F.lib().srx_prof_get_aux(self._h, cls_, C.byref(b)), self._h)
        return b.value

    def close(self) -> None:
        if self._h:
            F.lib().srx_ctx_destroy(self._h)
            self._h = C.c_void_p()


class DeviceCsr:
    """Owning wrapper of an ``srx_mat`` (device-resident X)."""

    def __init__(self, ctx: Context, handle):
        self.ctx = ctx
        self._h = handle

    @classmethod
    def upload(cls, ctx: Context, n_rows, n_cols, indptr, indices, values, store=F.STORE_AUTO, csc=False):
        """CSR arrays of X, or with ``csc`` its CSC arrays (col_offsets, row_indices, values)."""
        indptr = np.ascontiguousarray(indptr, dtype=np.uint64)
        indices = np.ascontiguousarray(indices, dtype=np.uint64)
        values = np.ascontiguousarray(values)
        if values.dtype not in DTYPE_OF_NP:
            # match_dyn_csr_matrix! panics on I64/U64/Usize/Bool/String (src/shared/mod.rs:117-126)
            raise F.SrxError(F.E_DTYPE, f"{values.dtype} CSR matrices are not supported for this operation")
        d = F.Csr(int(n_rows), int(n_cols), int(values.shape[0]), indptr.ctypes.data,
                  indices.ctypes.data, values.ctypes.data, DTYPE_OF_NP[values.dtype])
        h = C.c_void_p()
        up = F.lib().srx_matrix_upload_csc if csc else F.lib().srx_matrix_upload
        F.check(up(ctx.handle, C.byref(d), int(store), C.byref(h)), ctx.handle)
        return cls(ctx, h)

    def is_csc(self) -> bool:
        f = C.c_int32(0)
        F.check(F.lib().srx_matrix_format(self._h, C.byref(f)), self.ctx.handle)
        return f.value == 1

    def to_csr(self) -> "DeviceCsr":
        h = C.c_void_p()
        F.check(F.lib().srx_matrix_to_csr(self._h, C.byref(h)), self.ctx.handle)
        return DeviceCsr(self.ctx, h)

    def to_csc(self) -> "DeviceCsr":
        h = C.c_void_p()
        F.check(F.lib().srx_matrix_to_csc(self._h, C.byref(h)), self.ctx.handle)
        return DeviceCsr(self.ctx, h)

    @property
    def handle(self):
        return self._h

    def info(self) -> F.MatInfo:
        i = F.MatInfo()
        F.check(F.lib().srx_matrix_info(self._h, C.byref(i)), self.ctx.handle)
        return i

    def values(self, dtype=None) -> np.ndarray:
        """Current values in the logical dtype (F32 stays f32, everything computed is f64)."""
        i = self.info()
        if dtype is None:
            dtype = np.float32 if i.dtype == F.F32 else np.float64
        dtype = np.dtype(dtype)
        out = np.empty(i.nnz, dtype=dtype)
        F.check(F.lib().srx_matrix_download_values(self._h, F.ptr(out), F.F32 if dtype == np.float32 else F.F64),
                self.ctx.handle)
        return out

    def prepare(self) -> None:
        """Build the pattern-only index structures of the device layout now (clones inherit them)."""
        F.check(F.lib().srx_matrix_prepare(self._h), self.ctx.handle)

    def reserve_results(self, n_selected: int, n_components: int) -> None:
        """Allocate the result block (scores + small PCA results) ahead of the first solve; clones inherit it."""
        F.check(F.lib().srx_matrix_reserve_results(self._h, n_selected, n_components), self.ctx.handle)

    def clone(self) -> "DeviceCsr":
        h = C.c_void_p()
        F.check(F.lib().srx_matrix_clone(self._h, C.byref(h)), self.ctx.handle)
        return DeviceCsr(self.ctx, h)

    def copy_values_from(self, other: "DeviceCsr") -> None:
        F.check(F.lib().srx_matrix_copy_values(self._h, other._h), self.ctx.handle)

    def free(self) -> None:
        if self._h:
            # a matrix must not outlive its context (srx_matrix_free synchronises the context's stream): once the
            # context is closed the device memory has gone with it and only the handle is dropped
            if getattr(self.ctx, "_h", None):
                F.lib().srx_matrix_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class IMAnnData:
    """anndata_memory::IMAnnData, reduced to what the hot path reads and writes."""

    def __init__(self, x: DeviceCsr, indptr, indices, obs_names, var_names):
        self._x = x
        self.indptr = indptr
        self.indices = indices
        self.obs_names = list(obs_names)
        self.var_names = list(var_names)
        self.obsm: dict[str, np.ndarray] = {}
        self.varm: dict[str, np.ndarray] = {}
        self.var: dict[str, np.ndarray] = {}
        self.obs: dict[str, np.ndarray] = {}
        self.uns: dict[str, object] = {}

    @classmethod
    def new_basic(cls, x, obs_names=None, var_names=None, ctx: Context | None = None, store=F.STORE_AUTO, csc=False):
        """IMAnnData::new_basic(x, obs_names, var_names) (src/memory/processing/mod.rs:381).

        ``x``: ``(n_rows, n_cols, indptr, indices, values)`` (CSR arrays; with ``csc=True`` the CSC arrays
        col_offsets / row_indices / values) or a scipy.sparse CSR / CSC matrix (ArrayData::CsrMatrix / CscMatrix)."""
        ctx = ctx or Context.default()
        if hasattr(x, "indptr") and hasattr(x, "tocsr"):
            fmt = getattr(x, "format", "csr")
            if fmt not in ("csr", "csc"):
                raise F.SrxError(F.E_FORMAT, "X is neither a CSC nor a CSR matrix")
            csc = fmt == "csc"
            n_rows, n_cols = x.shape
            indptr, indices, values = x.indptr, x.indices, x.data
        else:
            n_rows, n_cols, indptr, indices, values = x
        indptr = np.ascontiguousarray(indptr, dtype=np.uint64)
        indices = np.ascontiguousarray(indices, dtype=np.uint64)
        dev = DeviceCsr.upload(ctx, n_rows, n_cols, indptr, indices, values, store, csc=csc)
        obs_names = obs_names if obs_names is not None else [f"obs{i}" for i in range(n_rows)]
        var_names = var_names if var_names is not None else [f"var{i}" for i in range(n_cols)]
        return cls(dev, indptr, indices, obs_names, var_names)

    def x(self) -> DeviceCsr:
        return self._x

    def n_obs(self) -> int:
        return int(self._x.info().n_rows)

    def n_vars(self) -> int:
        return int(self._x.info().n_cols)

    def x_dtype(self) -> np.dtype:
        return np.dtype(NP_OF_DTYPE[self._x.info().dtype])

    def x_values(self, dtype=None) -> np.ndarray:
        return self._x.values(dtype)

    @classmethod
    def _from_device(cls, dev: DeviceCsr, obs_names, var_names) -> "IMAnnData":
        """Wrap a matrix the library produced (filter / subset): the host pattern copy is downloaded."""
        i = dev.info()
        indptr = np.zeros((i.n_cols if dev.is_csc() else i.n_rows) + 1, dtype=np.uint64)
        indices = np.zeros(i.nnz, dtype=np.uint64)
        F.check(F.lib().srx_matrix_download_pattern(dev.handle, F.ptr(indptr), F.ptr(indices)), dev.ctx.handle)
        return cls(dev, indptr, indices, obs_names, var_names)

    def _adopt(self, other: "IMAnnData") -> None:
        """subset_inplace: this object takes over `other`'s X and names; per-row / per-column annotations that no
        longer match are dropped."""
        self._x.free()
        self._x = other._x
        self.indptr, self.indices = other.indptr, other.indices
        self.obs_names, self.var_names = other.obs_names, other.var_names
        self.obsm, self.varm, self.obs, self.var = {}, {}, {}, {}

    def deep_clone(self) -> "IMAnnData":
        c = IMAnnData(self._x.clone(), self.indptr, self.indices, self.obs_names, self.var_names)
        c.obsm = {k: v.copy() for k, v in self.obsm.items()}
        c.varm = {k: v.copy() for k, v in self.varm.items()}
        c.var = {k: v.copy() for k, v in self.var.items()}
        c.obs = {k: v.copy() for k, v in self.obs.items()}
        return c
