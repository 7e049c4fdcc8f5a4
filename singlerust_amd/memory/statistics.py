"""``single_rust::memory::statistics`` (src/memory/statistics/mod.rs:10-72) over libsrx_hip."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from .. import _ffi as F
from ..anndata import Direction, IMAnnData


def _n(adata: IMAnnData, direction) -> int:
    i = adata.x().info()
    return int(i.n_rows if int(direction) == F.ROW else i.n_cols)


def compute_number(adata: IMAnnData, direction: Direction) -> np.ndarray:
    """statistics/mod.rs:10-15 -> csr.rs:16-38; Vec<u32>."""
    out = np.zeros(_n(adata, direction), dtype=np.uint32)
    F.check(F.lib().srx_compute_number(adata.x().handle, int(direction), F.ptr(out)), adata.x().ctx.handle)
    return out


def compute_sum(adata: IMAnnData, direction: Direction) -> np.ndarray:
    """statistics/mod.rs:17-22 -> csr.rs:81-102; Vec<f64>."""
    out = np.zeros(_n(adata, direction), dtype=np.float64)
    F.check(F.lib().srx_compute_sum(adata.x().handle, int(direction), F.ptr(out)), adata.x().ctx.handle)
    return out


def compute_variance(adata: IMAnnData, direction: Direction) -> np.ndarray:
    """statistics/mod.rs:24-29 -> csr.rs:149-188 (Column: variance over the non-zeros only)."""
    out = np.zeros(_n(adata, direction), dtype=np.float64)
    F.check(F.lib().srx_compute_variance(adata.x().handle, int(direction), F.ptr(out)), adata.x().ctx.handle)
    return out


def compute_std_dev(adata: IMAnnData, direction: Direction) -> np.ndarray:
    """statistics/mod.rs:41-46 -> csr.rs:225-228."""
    out = np.zeros(_n(adata, direction), dtype=np.float64)
    F.check(F.lib().srx_compute_std_dev(adata.x().handle, int(direction), F.ptr(out)), adata.x().ctx.handle)
    return out


def compute_min_max(adata: IMAnnData, direction: Direction):
    """statistics/mod.rs:31-39 -> csr.rs:194-223; (min, max)."""
    n = _n(adata, direction)
    mn, mx = np.zeros(n, dtype=np.float64), np.zeros(n, dtype=np.float64)
    F.check(F.lib().srx_compute_min_max(adata.x().handle, int(direction), F.ptr(mn), F.ptr(mx)),
            adata.x().ctx.handle)
    return mn, mx


def gene_moments(adata: IMAnnData):
    """Superset (not in the reference): per-gene (nnz, sum, sumsq) from one pass."""
    g = _n(adata, Direction.Column)
    cnt = np.zeros(g, dtype=np.uint64)
    s, sq = np.zeros(g, dtype=np.float64), np.zeros(g, dtype=np.float64)
    F.check(F.lib().srx_gene_moments(adata.x().handle, F.ptr(cnt), F.ptr(s), F.ptr(sq)), adata.x().ctx.handle)
    return cnt, s, sq


@dataclass
class StatisticsContainer:          # src/memory/statistics/structs/mod.rs:1-10
    num_per_cell: np.ndarray
    num_per_gene: np.ndarray
    expr_per_gene: np.ndarray
    expr_per_cell: np.ndarray
    variance_per_gene: np.ndarray
    variance_per_cell: np.ndarray
    std_dev_per_cell: np.ndarray
    std_dev_per_gene: np.ndarray


def compute_qc_variables(adata: IMAnnData) -> StatisticsContainer:
    """statistics/mod.rs:48-72: one row pass + one column pass on the device (srx_compute_qc_variables)."""
    n, g = adata.n_obs(), adata.n_vars()
    out = StatisticsContainer(
        num_per_cell=np.zeros(n, np.uint32), num_per_gene=np.zeros(g, np.uint32),
        expr_per_gene=np.zeros(g), expr_per_cell=np.zeros(n),
        variance_per_gene=np.zeros(g), variance_per_cell=np.zeros(n),
        std_dev_per_cell=np.zeros(n), std_dev_per_gene=np.zeros(g))
    F.check(F.lib().srx_compute_qc_variables(
        adata.x().handle, F.ptr(out.num_per_cell), F.ptr(out.num_per_gene), F.ptr(out.expr_per_gene),
        F.ptr(out.expr_per_cell), F.ptr(out.variance_per_gene), F.ptr(out.variance_per_cell),
        F.ptr(out.std_dev_per_cell), F.ptr(out.std_dev_per_gene)), adata.x().ctx.handle)
    return out


def qc_vars_inplace(adata: IMAnnData) -> None:
    """statistics/mod.rs:74-103: the eight vectors as obs / var columns (same column names)."""
    d = compute_qc_variables(adata)
    adata.obs["num_genes_per_cell"] = d.num_per_cell
    adata.obs["sum_expr_per_cell"] = d.expr_per_cell
    adata.obs["var_expr_per_cell"] = d.variance_per_cell
    adata.obs["std_dev_per_cell"] = d.std_dev_per_cell
    adata.var["num_cells_per_gene"] = d.num_per_gene
    adata.var["sum_expr_per_gene"] = d.expr_per_gene
    adata.var["var_expr_per_gene"] = d.variance_per_gene
    adata.var["std_dev_per_gene"] = d.std_dev_per_gene
