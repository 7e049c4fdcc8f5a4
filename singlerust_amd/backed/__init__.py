"""``single_rust::backed`` (src/backed/mod.rs) over libsrx_hip: the matrix stays on disk / in host memory and is
visited as consecutive row chunks.

The reference reads ``AnnData<H5>`` (HDF5 — not in this image, and out of scope per SURVEY.md §8(f)3); the
store here is the same three arrays an h5ad ``X`` group holds (``indptr``, ``indices``, ``data``) as flat binary
files next to a small ``meta.json``, memory-mapped.  ``BackedAnnData.x().iter(chunk_size)`` yields
``(chunk, start, end)`` like ``ArrayElemOp::iter``.

* ``statistics.compute_number / compute_sum(adata, direction, mode)`` — src/backed/statistics/mod.rs:5-45.
* ``processing.pca_pipeline`` — the whole hot path out-of-core (two sweeps over the chunks); the reference's
  ``backed::processing`` is empty, this is the same machinery carried through (include/srx.h, "backed mode").
"""
from __future__ import annotations

import ctypes as C
import json
import os
from dataclasses import dataclass

import numpy as np

from .. import _ffi as F
from ..anndata import Context, DTYPE_OF_NP, Direction

__all__ = ["BackedAnnData", "BackedCsr", "ComputationMode", "BackedSession", "statistics", "processing"]


class ComputationMode:                       # src/shared/mod.rs:25-28
    @dataclass(frozen=True)
    class Chunked:
        size: int

    @dataclass(frozen=True)
    class Whole:
        pass


class CsrChunk:
    """Rows [start, end) of a backed CSR: a WINDOW of the row offsets plus the chunk's slice of indices / values
    (nothing is copied or rebased on the host — the device rebases, include/srx.h)."""

    def __init__(self, indptr, indices, values, n_cols: int):
        self.indptr, self.indices, self.values, self.n_cols = indptr, indices, values, int(n_cols)
        self.n_rows = int(indptr.shape[0] - 1)
        self.nnz = int(indptr[-1] - indptr[0])

    def c_struct(self):
        """srx_csr + the arrays that must stay alive while it is in use."""
        ip = np.ascontiguousarray(self.indptr, dtype=np.uint64)
        ix = np.ascontiguousarray(self.indices, dtype=np.uint64)
        vv = np.ascontiguousarray(self.values)
        if vv.dtype not in DTYPE_OF_NP:
            raise F.SrxError(F.E_DTYPE, f"{vv.dtype} CSR matrices are not supported for this operation")
        d = F.Csr(self.n_rows, self.n_cols, self.nnz, ip.ctypes.data, ix.ctypes.data, vv.ctypes.data, DTYPE_OF_NP[vv.dtype])
        return d, (ip, ix, vv)


class BackedCsr:
    """Flat on-disk CSR: ``meta.json`` + ``indptr.bin`` (u64, n_rows+1) + ``indices.bin`` (u64) + ``data.bin``."""

    def __init__(self, indptr, indices, values, n_cols: int):
        self.indptr, self.indices, self.values = indptr, indices, values
        self.n_rows = int(indptr.shape[0] - 1)
        self.n_cols = int(n_cols)
        self.nnz = int(values.shape[0])

    @staticmethod
    def write(path: str, indptr, indices, values, n_cols: int) -> None:
        os.makedirs(path, exist_ok=True)
        values = np.ascontiguousarray(values)
        if values.dtype not in DTYPE_OF_NP:
            raise F.SrxError(F.E_DTYPE, f"{values.dtype} CSR matrices are not supported for this operation")
        np.ascontiguousarray(indptr, dtype=np.uint64).tofile(os.path.join(path, "indptr.bin"))
        np.ascontiguousarray(indices, dtype=np.uint64).tofile(os.path.join(path, "indices.bin"))
        values.tofile(os.path.join(path, "data.bin"))
        meta = {"format": "srx-flat-csr-1", "n_rows": int(len(indptr) - 1), "n_cols": int(n_cols),
                "nnz": int(values.shape[0]), "dtype": values.dtype.name}
        with open(os.path.join(path, "meta.json"), "w") as f:
            json.dump(meta, f)

    @classmethod
    def open(cls, path: str) -> "BackedCsr":
        with open(os.path.join(path, "meta.json")) as f:
            meta = json.load(f)
        if meta.get("format") != "srx-flat-csr-1":
            raise F.SrxError(F.E_FORMAT, f"{path}: not a flat CSR store")
        n, nnz = int(meta["n_rows"]), int(meta["nnz"])

        def mm(name, dtype, count):
            if count == 0:
                return np.zeros(0, dtype=dtype)
            return np.memmap(os.path.join(path, name), dtype=dtype, mode="r", shape=(count,))

        indptr = mm("indptr.bin", np.uint64, n + 1)
        if int(indptr[-1]) != nnz or int(indptr[0]) != 0:
            raise F.SrxError(F.E_FORMAT, "X is not a CSR matrix: row_offsets do not span nnz")
        return cls(indptr, mm("indices.bin", np.uint64, nnz), mm("data.bin", np.dtype(meta["dtype"]), nnz), meta["n_cols"])

    def iter(self, chunk_size: int):
        """``ArrayElemOp::iter(chunk_size)``: (chunk, start, end) over consecutive row ranges."""
        if chunk_size <= 0:
            raise ValueError("chunk_size must be positive")
        for start in range(0, self.n_rows, chunk_size):
            end = min(start + chunk_size, self.n_rows)
            lo, hi = int(self.indptr[start]), int(self.indptr[end])
            yield CsrChunk(self.indptr[start:end + 1], self.indices[lo:hi], self.values[lo:hi], self.n_cols), start, end


class BackedAnnData:
    """The part of ``AnnData<B>`` the backed path touches: ``x()``, ``n_obs()``, ``n_vars()``."""

    def __init__(self, x: BackedCsr, ctx: Context | None = None):
        self._x = x
        self.ctx = ctx or Context.default()

    @classmethod
    def open(cls, path: str, ctx: Context | None = None) -> "BackedAnnData":
        return cls(BackedCsr.open(path), ctx)

    def x(self) -> BackedCsr:
        return self._x

    def n_obs(self) -> int:
        return self._x.n_rows

    def n_vars(self) -> int:
        return self._x.n_cols


class BackedSession:
    """Owning wrapper of an ``srx_backed`` (include/srx.h)."""

    def __init__(self, ctx: Context, n_cols: int, store=F.STORE_AUTO):
        self.ctx = ctx
        self.n_cols = int(n_cols)
        h = C.c_void_p()
        F.check(F.lib().srx_backed_create(ctx.handle, self.n_cols, int(store), C.byref(h)), ctx.handle)
        self._h = h

    def _chk(self, rc):
        F.check(rc, self.ctx.handle)

    def stats_tile(self, chunk: CsrChunk, target_sum=0.0, transform=0, row_number=None, row_sum=None) -> None:
        d, keep = chunk.c_struct()
        self._chk(F.lib().srx_backed_stats_tile(self._h, C.byref(d), float(target_sum), int(transform), F.ptr(row_number),
                                               F.ptr(row_sum)))
        del keep

    def moments(self):
        cnt = np.zeros(self.n_cols, dtype=np.uint64)
        s = np.zeros(self.n_cols, dtype=np.float64)
        q = np.zeros(self.n_cols, dtype=np.float64)
        n = C.c_uint64(0)
        self._chk(F.lib().srx_backed_moments(self._h, F.ptr(cnt), F.ptr(s), F.ptr(q), C.byref(n)))
        return cnt, s, q, int(n.value)

    def select(self, n_hvg=0, sel=None, opts: F.PcaOpts | None = None) -> np.ndarray:
        cap = min(int(n_hvg), self.n_cols) if n_hvg else (len(sel) if sel is not None else self.n_cols)
        out = np.zeros(max(cap, 1), dtype=np.uint64)
        n_out = C.c_uint64(0)
        selp = None if sel is None else np.ascontiguousarray(sel, dtype=np.uint64)
        self._chk(F.lib().srx_backed_select(self._h, int(n_hvg), F.ptr(selp), 0 if selp is None else len(selp),
                                           None if opts is None else C.byref(opts), F.ptr(out), C.byref(n_out)))
        return out[:int(n_out.value)]

    def gram_tile(self, chunk: CsrChunk, target_sum=0.0, transform=0) -> None:
        d, keep = chunk.c_struct()
        self._chk(F.lib().srx_backed_gram_tile(self._h, C.byref(d), float(target_sum), int(transform)))
        del keep

    def solve(self) -> F.PcaInfo:
        info = F.PcaInfo()
        rc = F.lib().srx_backed_solve(self._h, C.byref(info))
        self._chk(rc)
        return info

    def fetch(self, n_rows: int, info: F.PcaInfo):
        k, n_pc = int(info.k), int(info.n_pc)
        scores = np.zeros((int(n_rows), n_pc), dtype=np.float64)
        comps = np.zeros((k, n_pc), dtype=np.float64)
        evr = np.zeros(n_pc, dtype=np.float64)
        mean = np.zeros(k, dtype=np.float64)
        std = np.zeros(k, dtype=np.float64)
        sel = np.zeros(k, dtype=np.uint64)
        self._chk(F.lib().srx_backed_fetch(self._h, F.ptr(scores), F.ptr(comps), F.ptr(evr), F.ptr(mean), F.ptr(std), F.ptr(sel)))
        return scores, comps, evr, mean, std, sel

    def close(self) -> None:
        if self._h:
            F.lib().srx_backed_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


from . import processing, statistics  # noqa: E402
