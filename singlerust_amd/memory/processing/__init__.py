"""``single_rust::memory::processing`` (src/memory/processing/mod.rs:303-332) over libsrx_hip."""
from __future__ import annotations

from ... import _ffi as F
from ...anndata import Direction, IMAnnData
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


__all__ = ["normalize_total_inplace", "normalize_total", "log1p_transform_inplace", "log1p_transform",
           "normalize_log1p_inplace", "dim_red"]
