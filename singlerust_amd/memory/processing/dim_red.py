"""``single_rust::memory::processing::dim_red`` (src/memory/processing/dim_red/mod.rs)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ... import _ffi as F
from ...anndata import FeatureSelection, IMAnnData
from .. import statistics


def select_features(adata: IMAnnData, feature_selection) -> np.ndarray:
    """dim_red/mod.rs:123-156: feature indices in selection order (Vec<usize>)."""
    fs = feature_selection
    if isinstance(fs, FeatureSelection.HighlyVariable):            # :135-140
        g = adata.n_vars()
        out = np.zeros(min(int(fs.n), g), dtype=np.uint64)
        n_out = C.c_uint64(0)
        F.check(F.lib().srx_select_hvg(adata.x().handle, int(fs.n), F.ptr(out) if out.size else None,
                                       C.byref(n_out)), adata.x().ctx.handle)
        return out[: n_out.value]
    if isinstance(fs, FeatureSelection.HighlyVariableCol):         # :125-134
        if fs.col not in adata.var:
            raise KeyError(f"Error accessing column '{fs.col}'")
        mask = np.asarray(adata.var[fs.col])
        if mask.dtype != np.bool_:
            raise TypeError(f"Column '{fs.col}' is not boolean")
        return np.nonzero(mask)[0].astype(np.uint64)
    if isinstance(fs, FeatureSelection.Randomized):                # :141-146 (thread_rng: unseeded)
        idx = np.random.default_rng().permutation(adata.n_vars()).astype(np.uint64)
        return idx[: int(fs.n)]
    if isinstance(fs, FeatureSelection.VarianceThreshold):         # :147-153
        from ...anndata import Direction
        var = statistics.compute_variance(adata, Direction.Column)
        return np.nonzero(var > fs.threshold)[0].astype(np.uint64)
    if isinstance(fs, FeatureSelection.NoSelection) or fs is None:  # :154
        return np.arange(adata.n_vars(), dtype=np.uint64)
    raise TypeError(f"unknown FeatureSelection {fs!r}")


def pca_inplace(adata: IMAnnData, n_components=None, center=None, scale=None, n_threads=None,
                feature_selection=FeatureSelection.None_, svd_mode=None, *, block=0, max_iter=0, tol=0.0,
                seed=0, store_loadings=False, solver=0) -> F.PcaInfo:
    """dim_red/mod.rs:24-94.  Stores obsm["X_pca"] (n_obs x n_pc f64), the only output the
    reference keeps (:105-106); with store_loadings also varm["PCA_loadings"] in the layout
    of :108-118.  ``svd_mode`` (FaerSVD / LapackSVD marker) is accepted and ignored: the GPU
    path is a randomized subspace iteration on CSR x dense-panel SpMMs."""
    sel = select_features(adata, feature_selection)
    k = int(sel.shape[0])
    n_pc = min(2 if n_components is None else int(n_components), k)             # :52
    opts = F.PcaOpts(-1 if n_components is None else int(n_components),
                     -1 if center is None else int(bool(center)),
                     -1 if scale is None else int(bool(scale)),
                     -1 if n_threads is None else int(n_threads),
                     int(block), int(max_iter), int(solver), float(tol), int(seed))
    n = adata.n_obs()
    scores = np.zeros((n, n_pc), dtype=np.float64)
    comps = np.zeros((k, n_pc), dtype=np.float64)
    evr = np.zeros(n_pc, dtype=np.float64)
    mean = np.zeros(k, dtype=np.float64)
    std = np.zeros(k, dtype=np.float64)
    info = F.PcaInfo()
    F.check(F.lib().srx_pca(adata.x().handle, F.ptr(sel), k, C.byref(opts), F.ptr(scores), F.ptr(comps),
                            F.ptr(evr), F.ptr(mean), F.ptr(std), C.byref(info)), adata.x().ctx.handle)
    adata.obsm["X_pca"] = scores                                                # :105-106
    adata.uns["pca"] = {"components": comps, "explained_variance_ratio": evr, "mean": mean, "std": std,
                        "selected_features": sel, "n_iter": info.n_iter, "residual": info.residual,
                        "solver": info.solver}
    if store_loadings:                                                          # :108-118
        full = np.zeros((adata.n_vars(), n_pc), dtype=np.float64)
        F.check(F.lib().srx_pca_loadings(F.ptr(comps), F.ptr(std), F.ptr(sel), k, n_pc, adata.n_vars(),
                                         F.ptr(full)))
        adata.varm["PCA_loadings"] = full
    return info
