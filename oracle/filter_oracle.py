"""CPU restatement of the reference's QC filters (TEST INFRASTRUCTURE ONLY — never imported by the product).

Follows, line by line:
  * src/memory/processing/mod.rs:16-31   calculate_cell_stats  (compute_number Row if an Absolute limit is
                                          present, compute_sum Row always)
  * :33-84                                create_filter_mask    (the nine FlexValue combinations)
  * :148-174                              calculate_percentiles (quantiles of the SUMS; f64::MIN / f64::MAX when a
                                          limit is not Relative)
  * :176-243                              the gene-side twins (compute_number / compute_sum Column)
  * :86-146, :245-299                     filter_* = mask -> indices (src/shared/processing/mod.rs:11-50) -> subset
The quantile itself lives in a third-party crate that is NOT under /root/reference: `ndarray-stats = "0.5.1"`
(Cargo.toml:35), `quantile_axis_mut(Axis(0), n64(p), &interpolate::Linear)`.  Its published algorithm, restated:
index = p * (len - 1); lower / higher = the order statistics at floor / ceil of the index;
result = lower + (higher - lower) * fract(index).  Parity of the quantile is pinned on that description only.
The subset keeps rows / columns in their original order (anndata `SelectInfoElem::Index` of ascending indices).
"""
from __future__ import annotations

import numpy as np

from . import COLUMN, ROW, Csr, compute_number, compute_sum

F64_MIN = -np.finfo(np.float64).max      # Rust f64::MIN
F64_MAX = np.finfo(np.float64).max


def absolute(v):
    return ("abs", int(v))


def relative(p):
    return ("rel", float(p))


NONE = None


def _quantile_linear(values: np.ndarray, q: float) -> float:
    s = np.sort(np.asarray(values, dtype=np.float64))
    n = s.shape[0]
    if n == 0:
        raise ValueError("Error calculating percentile: empty input")
    idx = q * (n - 1)
    lo, hi = int(np.floor(idx)), int(np.ceil(idx))
    frac = idx - np.floor(idx)
    return float(s[lo] + (s[hi] - s[lo]) * frac)


def calculate_percentiles(values, lower, upper):                       # mod.rs:148-174
    lp = _quantile_linear(values, lower[1]) if lower is not None and lower[0] == "rel" else F64_MIN
    up = _quantile_linear(values, upper[1]) if upper is not None and upper[0] == "rel" else F64_MAX
    return lp, up


def create_filter_mask(n, counts, sums, lower, upper, lp, up) -> np.ndarray:      # mod.rs:33-84 / :192-243
    mask = np.ones(n, dtype=bool)
    if lower is not None:
        mask &= (counts >= np.uint32(lower[1])) if lower[0] == "abs" else (sums >= lp)
    if upper is not None:
        mask &= (counts <= np.uint32(upper[1])) if upper[0] == "abs" else (sums <= up)
    return mask


def subset(m: Csr, row_mask=None, col_mask=None) -> Csr:
    indptr = m.indptr.astype(np.int64)
    rows = np.arange(m.n_rows) if row_mask is None else np.flatnonzero(row_mask)
    if col_mask is None:
        newcol = np.arange(m.n_cols, dtype=np.int64)
        n_cols = m.n_cols
    else:
        newcol = np.full(m.n_cols, -1, dtype=np.int64)
        kept = np.flatnonzero(col_mask)
        newcol[kept] = np.arange(kept.shape[0])
        n_cols = int(kept.shape[0])
    out_ptr, out_idx, out_val = [0], [], []
    for r in rows:
        lo, hi = indptr[r], indptr[r + 1]
        c = newcol[m.indices[lo:hi].astype(np.int64)]
        keep = c >= 0
        out_idx.append(c[keep])
        out_val.append(m.values[lo:hi][keep])
        out_ptr.append(out_ptr[-1] + int(keep.sum()))
    idx = np.concatenate(out_idx) if out_idx else np.zeros(0, np.int64)
    val = np.concatenate(out_val) if out_val else np.zeros(0, m.values.dtype)
    return Csr(len(rows), n_cols, np.asarray(out_ptr, np.uint64), idx.astype(np.uint64), val.astype(m.values.dtype))


def _need_count(lower, upper):
    return (lower is not None and lower[0] == "abs") or (upper is not None and upper[0] == "abs")


def filter_cells(m: Csr, lower, upper):                               # mod.rs:86-146
    counts = compute_number(m, ROW) if _need_count(lower, upper) else None
    sums = compute_sum(m, ROW)
    lp, up = calculate_percentiles(sums, lower, upper)
    mask = create_filter_mask(m.n_rows, counts, sums, lower, upper, lp, up)
    return subset(m, row_mask=mask), mask


def filter_genes(m: Csr, lower, upper):                               # mod.rs:245-299
    counts = compute_number(m, COLUMN) if _need_count(lower, upper) else None
    sums = compute_sum(m, COLUMN)
    lp, up = calculate_percentiles(sums, lower, upper)
    mask = create_filter_mask(m.n_cols, counts, sums, lower, upper, lp, up)
    return subset(m, col_mask=mask), mask
