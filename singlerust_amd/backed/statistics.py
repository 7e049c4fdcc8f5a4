"""``single_rust::backed::statistics`` (src/backed/statistics/mod.rs:5-45).

``ComputationMode::Chunked(size)`` walks ``adata.x().iter(size)`` (src/shared/statistics/mod.rs:17-41,59-83); each
chunk is uploaded while the previous one is still being reduced on the GPU.  Direction::Column accumulates over
the chunks; Direction::Row writes each chunk at ITS rows — the reference's chunk helpers index the output by the
chunk-local row (csr.rs:57-62,126-131), which folds every chunk onto the first ``size`` entries; that is a defect,
not a contract, and is not reproduced (the test suite keeps both forms side by side).
``ComputationMode::Whole`` loads the matrix and takes the resident path.
"""
from __future__ import annotations

import numpy as np

from .. import _ffi as F
from ..anndata import Direction, IMAnnData
from . import BackedAnnData, BackedSession, ComputationMode


def _whole(adata: BackedAnnData) -> IMAnnData:
    x = adata.x()
    import scipy.sparse as sp
    m = sp.csr_matrix((np.asarray(x.values), np.asarray(x.indices).astype(np.int64), np.asarray(x.indptr).astype(np.int64)),
                      shape=(x.n_rows, x.n_cols))
    return IMAnnData.new_basic(m, ctx=adata.ctx)


def _chunked(adata: BackedAnnData, size: int, direction, want_sum: bool):
    x = adata.x()
    row = int(direction) == F.ROW
    sess = BackedSession(adata.ctx, x.n_cols)
    try:
        num = np.zeros(x.n_rows, dtype=np.uint32) if row and not want_sum else None
        sums = np.zeros(x.n_rows, dtype=np.float64) if row and want_sum else None
        for chunk, start, end in x.iter(size):
            sess.stats_tile(chunk, row_number=None if num is None else num[start:end],
                            row_sum=None if sums is None else sums[start:end])
        if row:
            return sums if want_sum else num
        cnt, s, _, _ = sess.moments()
        return s if want_sum else cnt.astype(np.uint32)        # the reference counts in u32
    finally:
        sess.close()


def compute_number(adata: BackedAnnData, direction: Direction, mode) -> np.ndarray:
    """src/backed/statistics/mod.rs:5-24; Vec<u32>."""
    if isinstance(mode, ComputationMode.Chunked):
        return _chunked(adata, mode.size, direction, want_sum=False)
    from ..memory import statistics as mem
    return mem.compute_number(_whole(adata), direction)


def compute_sum(adata: BackedAnnData, direction: Direction, mode) -> np.ndarray:
    """src/backed/statistics/mod.rs:26-45; Vec<f64>."""
    if isinstance(mode, ComputationMode.Chunked):
        return _chunked(adata, mode.size, direction, want_sum=True)
    from ..memory import statistics as mem
    return mem.compute_sum(_whole(adata), direction)
