"""CPU restatement of the reference's CSC arms (TEST INFRASTRUCTURE ONLY — never imported by the product).

Follows, loop by loop:
  * src/shared/statistics/helper/csc.rs:15-35    number_whole_helper  (Row: histogram of row_indices; Column: offsets diff)
  * csc.rs:71-95                                  sum_whole_helper     (Row: scatter-add in storage order; Column: per-column
                                                                       sequential sum)
  * csc.rs:132-180                                variance_whole_helper (Row: E[x^2]-E[x]^2 over the non-zeros, 0 for an empty
                                                                       row; Column: two-pass sum((v-mean)^2)/count with NO guard:
                                                                       0/0 = NaN for an empty column)
  * csc.rs:186-212                                min_max_whole_helper
  * csc.rs:214-216                                std_dev_whole
  * src/memory/processing/scale/mod.rs:7-57,91-139  scale_row / scale_col on a CSC matrix (result is always F64)
  * src/memory/processing/transform/mod.rs:8-62   log1p on the value array (F32 stays F32, everything else -> f64)
  * src/shared/mod.rs:261-290                     convert_to_array_f64_csc_selected (dense N x k, column c = gene selected[c])
Sequential accumulation orders are kept (np.add.at applies its updates in index order; the per-column sums are plain
left-to-right loops), so integer-valued inputs compare bit-exactly and floats to the last rounding.
"""
from __future__ import annotations

import numpy as np

from . import ROW


class Csc:
    """Reference-layout CSC of X (n_rows x n_cols): col_offsets (n_cols+1), row_indices, values."""

    def __init__(self, n_rows, n_cols, col_offsets, row_indices, values):
        self.n_rows, self.n_cols = int(n_rows), int(n_cols)
        self.col_offsets = np.ascontiguousarray(col_offsets, dtype=np.uint64)
        self.row_indices = np.ascontiguousarray(row_indices, dtype=np.uint64)
        self.values = np.ascontiguousarray(values)
        assert self.col_offsets.shape == (self.n_cols + 1,) and self.row_indices.shape == self.values.shape
        self.nnz = int(self.values.shape[0])

    @classmethod
    def from_scipy(cls, x):
        x = x.tocsc()
        x.sort_indices()
        return cls(x.shape[0], x.shape[1], x.indptr, x.indices, x.data)

    def with_values(self, values):
        return Csc(self.n_rows, self.n_cols, self.col_offsets, self.row_indices, values)

    def _cols(self):
        off = self.col_offsets.astype(np.int64)
        v = self.values.astype(np.float64)
        for c in range(self.n_cols):
            yield c, v[off[c]:off[c + 1]]


def compute_number(m: Csc, direction: int) -> np.ndarray:
    if direction == ROW:                                            # csc.rs:20-26
        out = np.zeros(m.n_rows, dtype=np.uint32)
        np.add.at(out, m.row_indices.astype(np.int64), np.uint32(1))
        return out
    return np.diff(m.col_offsets.astype(np.int64)).astype(np.uint32)  # csc.rs:27-33


def compute_sum(m: Csc, direction: int) -> np.ndarray:
    if direction == ROW:                                            # csc.rs:78-84
        out = np.zeros(m.n_rows, dtype=np.float64)
        np.add.at(out, m.row_indices.astype(np.int64), m.values.astype(np.float64))
        return out
    out = np.zeros(m.n_cols, dtype=np.float64)                      # csc.rs:85-91
    for c, col in m._cols():
        s = 0.0
        for v in col:
            s += v
        out[c] = s
    return out


def compute_variance(m: Csc, direction: int) -> np.ndarray:
    s = compute_sum(m, direction)                                   # csc.rs:137-138
    cnt = compute_number(m, direction).astype(np.float64)
    if direction == ROW:                                            # csc.rs:141-156
        sq = np.zeros(m.n_rows, dtype=np.float64)
        v = m.values.astype(np.float64)
        np.add.at(sq, m.row_indices.astype(np.int64), v * v)
        out = np.zeros(m.n_rows, dtype=np.float64)
        nz = cnt > 0
        mean = s[nz] / cnt[nz]
        out[nz] = sq[nz] / cnt[nz] - mean * mean
        return out
    out = np.zeros(m.n_cols, dtype=np.float64)                      # csc.rs:157-170: no guard
    with np.errstate(invalid="ignore", divide="ignore"):
        for c, col in m._cols():
            mean = s[c] / cnt[c]
            acc = 0.0
            for v in col:
                d = v - mean
                acc += d * d
            out[c] = np.float64(acc) / cnt[c]
    return out


def compute_std_dev(m: Csc, direction: int) -> np.ndarray:
    return np.sqrt(compute_variance(m, direction))                  # csc.rs:214-216


def compute_min_max(m: Csc, direction: int):
    v = m.values.astype(np.float64)
    if direction == ROW:                                            # csc.rs:193-201
        mn = np.full(m.n_rows, np.inf)
        mx = np.full(m.n_rows, -np.inf)
        r = m.row_indices.astype(np.int64)
        np.minimum.at(mn, r, v)
        np.maximum.at(mx, r, v)
        return mn, mx
    mn = np.full(m.n_cols, np.inf)                                  # csc.rs:202-212
    mx = np.full(m.n_cols, -np.inf)
    for c, col in m._cols():
        if col.size:
            mn[c], mx[c] = col.min(), col.max()
    return mn, mx


def normalize_total(m: Csc, target_sum: float, direction: int) -> Csc:
    """scale_row / scale_col on a CSC matrix (scale/mod.rs:7-57,91-139): X becomes F64."""
    sums = compute_sum(m, direction)
    scale = np.where(sums == 0.0, 0.0, target_sum / np.where(sums == 0.0, 1.0, sums))
    v = m.values.astype(np.float64)
    if direction == ROW:
        v = v * scale[m.row_indices.astype(np.int64)]               # :34-38
    else:
        per_entry_col = np.repeat(np.arange(m.n_cols), np.diff(m.col_offsets.astype(np.int64)))
        v = v * scale[per_entry_col]                                # :113-117
    return m.with_values(v)


def log1p_transform(m: Csc) -> Csc:
    if m.values.dtype == np.float32:
        return m.with_values(np.log1p(m.values))
    return m.with_values(np.log1p(m.values.astype(np.float64)))


def densify_selected(m: Csc, sel) -> np.ndarray:
    """convert_to_array_f64_csc_selected (src/shared/mod.rs:261-290)."""
    out = np.zeros((m.n_rows, len(sel)), dtype=np.float64)
    off = m.col_offsets.astype(np.int64)
    for new_col, col in enumerate(np.asarray(sel, dtype=np.int64)):
        if col < m.n_cols:
            lo, hi = off[col], off[col + 1]
            out[m.row_indices[lo:hi].astype(np.int64), new_col] = m.values[lo:hi].astype(np.float64)
    return out
