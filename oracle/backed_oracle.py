"""CPU restatement of the reference's chunked statistics (TEST INFRASTRUCTURE ONLY — never imported by the product).

Follows:
  * src/backed/statistics/mod.rs:5-45            compute_number / compute_sum with ComputationMode::Chunked(size)
  * src/shared/statistics/mod.rs:17-41, 59-83    number::chunked / sum::chunked: a zeroed Vec of length n_obs (Row) or
                                                  n_vars (Column), then one helper call per (chunk, start, end) of
                                                  x.iter(chunk_size) — start / end are dropped
  * src/shared/statistics/helper/csr.rs:48-74    number_chunk_helper: Row `reference[i] += row_len(i)` with i the row
                                                  index INSIDE the chunk; Column `reference[col] += 1`
  * csr.rs:112-143                               sum_chunk_helper: same indexing, f64::from(value) summed
The chunk iterator is anndata's (`ArrayElemOp::iter`, crate anndata 0.4.2, absent from /root/reference): it yields
consecutive row ranges of `chunk_size` rows (the last one shorter) as CSR matrices with rebased row offsets.

`as_written=True` reproduces the Row loops literally: every chunk adds onto entries [0, chunk_rows) of the output, so
only the first `chunk_size` entries are ever non-zero and they hold sums over all chunks.  That contradicts the
function's own contract (a length-n_obs vector of per-row values; ComputationMode::Whole gives exactly that), the
reference has no test for it, and SURVEY.md §8(f)3 records it as a defect; the product writes each chunk at its own
rows, i.e. `as_written=False`, which equals the Whole result.  Column direction has no such issue.
"""
from __future__ import annotations

import numpy as np

from . import ROW, Csr


def iter_chunks(m: Csr, chunk_size: int):
    indptr = np.asarray(m.indptr, dtype=np.int64)
    for start in range(0, m.n_rows, chunk_size):
        end = min(start + chunk_size, m.n_rows)
        lo, hi = int(indptr[start]), int(indptr[end])
        yield Csr(end - start, m.n_cols, indptr[start:end + 1] - lo, np.asarray(m.indices)[lo:hi], np.asarray(m.values)[lo:hi]), start, end


def number_chunked(m: Csr, chunk_size: int, direction: int, as_written: bool = False) -> np.ndarray:
    out = np.zeros(m.n_rows if direction == ROW else m.n_cols, dtype=np.uint32)
    for chunk, start, _ in iter_chunks(m, chunk_size):
        ip = np.asarray(chunk.indptr, dtype=np.int64)
        if direction == ROW:
            base = 0 if as_written else start
            for i in range(chunk.n_rows):                       # csr.rs:57-62
                if base + i < out.shape[0]:
                    out[base + i] += np.uint32(ip[i + 1] - ip[i])
        else:
            for c in np.asarray(chunk.indices, dtype=np.int64):  # csr.rs:66-70
                if c < out.shape[0]:
                    out[c] += 1
    return out


def sum_chunked(m: Csr, chunk_size: int, direction: int, as_written: bool = False) -> np.ndarray:
    out = np.zeros(m.n_rows if direction == ROW else m.n_cols, dtype=np.float64)
    for chunk, start, _ in iter_chunks(m, chunk_size):
        ip = np.asarray(chunk.indptr, dtype=np.int64)
        vals = np.asarray(chunk.values).astype(np.float64)
        if direction == ROW:
            base = 0 if as_written else start
            for i in range(chunk.n_rows):                       # csr.rs:124-131: row sum, then +=
                if base + i < out.shape[0]:
                    s = 0.0
                    for v in vals[ip[i]:ip[i + 1]]:
                        s += v
                    out[base + i] += s
        else:
            idx = np.asarray(chunk.indices, dtype=np.int64)
            for c, v in zip(idx, vals):                          # csr.rs:134-140
                if c < out.shape[0]:
                    out[c] += v
    return out
