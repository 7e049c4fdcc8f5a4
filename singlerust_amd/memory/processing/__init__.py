"""``single_rust::memory::processing`` (src/memory/processing/mod.rs:303-332) over libsrx_hip."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ... import _ffi as F
from ...anndata import DeviceCsr, Direction, FlexValue, IMAnnData
from . import dim_red


def normalize_total_inplace(adata: IMAnnData, target_sum: float, direction: Direction) -> None:
    """processing/mod.rs:303-312 -> scale/mod.rs:7-23,59-89 (Row) / :91-107,141-173 (Column)."""
    F.check(F.lib().srx_normalize_total_inplace(adata.x().handle, float(target_sum), int(direction)),
            adata.x().ctx.handle)


def normalize_total(adata: IMAnnData, target_sum: float, direction: Direction) -> IMAnnData:
    """processing/mod.rs:314-322: deep_clone, then the in-place form."""
    new = adata.deep_clone()
    normalize_total_inplace(new, target_sum, direction)
    return new


def log1p_transform_inplace(adata: IMAnnData) -> None:
    """processing/mod.rs:324-326 -> transform/mod.rs:8-62."""
    F.check(F.lib().srx_log1p_inplace(adata.x().handle), adata.x().ctx.handle)


def log1p_transform(adata: IMAnnData) -> IMAnnData:
    """processing/mod.rs:328-332."""
    new = adata.deep_clone()
    log1p_transform_inplace(new)
    return new


def normalize_log1p_inplace(adata: IMAnnData, target_sum: float):
    """Fused fast path (not in the reference): normalize_total_inplace(.., Row) followed by
    log1p_transform_inplace in one pass over the values.  Returns the raw row sums."""
    import numpy as np
    sums = np.zeros(adata.n_obs(), dtype=np.float64)
    F.check(F.lib().srx_normalize_log1p_inplace(adata.x().handle, float(target_sum), F.ptr(sums)),
            adata.x().ctx.handle)
    return sums




def _filter(adata: IMAnnData, lower, upper, genes: bool) -> IMAnnData:
    fn = F.lib().srx_filter_genes if genes else F.lib().srx_filter_cells
    n = adata.n_vars() if genes else adata.n_obs()
    mask = np.zeros(n, dtype=np.uint8)
    h = C.c_void_p()
    F.check(fn(adata.x().handle, FlexValue.to_c(lower), FlexValue.to_c(upper), C.byref(h), F.ptr(mask)),
            adata.x().ctx.handle)
    keep = mask.astype(bool)
    obs = adata.obs_names if genes else [nm for nm, k in zip(adata.obs_names, keep) if k]
    var = [nm for nm, k in zip(adata.var_names, keep) if k] if genes else adata.var_names
    out = IMAnnData._from_device(DeviceCsr(adata.x().ctx, h), obs, var)
    out.uns["filter_mask"] = keep
    return out


def filter_cells(adata: IMAnnData, lower_lim, upper_lim) -> IMAnnData:
    """processing/mod.rs:118-146: cells kept by the nnz-count (Absolute) / sum-quantile (Relative) limits."""
    return _filter(adata, lower_lim, upper_lim, genes=False)


def filter_cells_inplace(adata: IMAnnData, lower_lim, upper_lim) -> None:
    """processing/mod.rs:86-116."""
    adata._adopt(_filter(adata, lower_lim, upper_lim, genes=False))


def filter_genes(adata: IMAnnData, lower_lim, upper_lim) -> IMAnnData:
    """processing/mod.rs:271-299."""
    return _filter(adata, lower_lim, upper_lim, genes=True)


def filter_genes_inplace(adata: IMAnnData, lower_lim, upper_lim) -> None:
    """processing/mod.rs:245-269."""
    adata._adopt(_filter(adata, lower_lim, upper_lim, genes=True))


__all__ = ["normalize_total_inplace", "normalize_total", "log1p_transform_inplace", "log1p_transform",
           "normalize_log1p_inplace", "filter_cells", "filter_cells_inplace", "filter_genes", "filter_genes_inplace",
           "dim_red"]
