"""The hot path out-of-core: normalize_total(Row) -> log1p -> HVG -> PCA over a backed matrix in two sweeps.

Not in the reference (``src/backed/processing/mod.rs`` is empty); SURVEY.md §8(f)3 asks for the chunked machinery
"generalised to the whole pipeline".  Per sweep every chunk is uploaded once; what stays in HBM is the per-gene
moments, the k x k Gram tiles and the HVG-compacted rows (include/srx.h, "backed mode").
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .. import _ffi as F
from . import BackedAnnData, BackedSession


@dataclass
class BackedPcaResult:
    x_pca: np.ndarray               # n_obs x n_pc (rows in store order)
    components: np.ndarray          # k x n_pc, rows in selection order
    explained_variance_ratio: np.ndarray
    mean: np.ndarray
    std: np.ndarray
    selected: np.ndarray            # what select_features returns (variance-rank order for HighlyVariable)
    info: F.PcaInfo
    row_sums: np.ndarray            # compute_sum(Row) of the raw counts, a by-product of sweep 1


def pca_pipeline(adata: BackedAnnData, chunk_size: int, target_sum: float = 1e4, n_hvg: int = 2000,
                 n_components: int = 50, center: bool = True, scale: bool = True, store=F.STORE_AUTO,
                 seed: int = 0, tol: float = 0.0, max_iter: int = 0, selected=None) -> BackedPcaResult:
    x = adata.x()
    tf = F.BACKED_NORMALIZE | F.BACKED_LOG1P
    sess = BackedSession(adata.ctx, x.n_cols, store)
    try:
        row_sums = np.zeros(x.n_rows, dtype=np.float64)
        for chunk, start, end in x.iter(chunk_size):
            sess.stats_tile(chunk, target_sum, tf, row_sum=row_sums[start:end])
        opts = F.PcaOpts(int(n_components), int(center), int(scale), 0, 0, int(max_iter), 0, float(tol), int(seed))
        sess.select(n_hvg if selected is None else 0, selected, opts)
        for chunk, _, _ in x.iter(chunk_size):
            sess.gram_tile(chunk, target_sum, tf)
        info = sess.solve()
        scores, comps, evr, mean, std, sel = sess.fetch(x.n_rows, info)
        return BackedPcaResult(scores, comps, evr, mean, std, sel, info, row_sums)
    finally:
        sess.close()
