"""numpy restatement of the reference PCA — TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the live arithmetic is in the un-vendored crate
``single_algebra = "0.1.0-alpha.3"`` (Cargo.toml:42; call sites
src/memory/processing/dim_red/mod.rs:53-57,66,69,78) and the reference's tests assert
nothing numeric about PCA.  This file follows the only in-tree statement of the maths, the
dead predecessor ``src/shared/processing/pca/mod.rs``:

* fit        :74-154  mean_axis(0); std_axis(0, ddof=0); centre; scale; SVD;
                      eigenvalues = s^2/(nrows-1); total = sum(all); ratio; V[:, :n]
* transform  :156-185 centre, scale, X . components
* loadings   :204-215 components^T * std_dev (broadcast over rows)

and the orchestration of ``pca_inplace`` (dim_red/mod.rs:24-94): select features
(:123-156), densify the selected columns in selection order (shared/mod.rs:230-259),
n_components = min(n.unwrap_or(2), k) (:52), center/scale default true (:55-56).
"""
from __future__ import annotations

import numpy as np

from . import COLUMN, Csr, compute_variance, densify_selected, select_hvg


class Pca:
    def __init__(self, n_components=None, center=True, scale=False):
        self.n_components = n_components
        self.center = center
        self.scale = scale
        self.components = None
        self.mean = None
        self.std_dev = None
        self.explained_variance_ratio = None
        self.total_variance = None
        self.eigenvalues = None

    def fit(self, data: np.ndarray) -> None:
        nrows, ncols = data.shape
        n_components = self.n_components if self.n_components is not None else min(nrows, ncols)
        centered = np.array(data, dtype=np.float64, copy=True)          # :83
        if self.center or self.scale:                                   # :85
            mean = data.mean(axis=0)                                    # :87
            std = data.std(axis=0, ddof=0) if self.scale else np.ones(ncols)  # :90-94
            if self.center:
                centered -= mean                                        # :100-102
            if self.scale:
                centered /= std                                         # :107-109 (0/0 -> NaN)
            self.mean, self.std_dev = mean, std
        else:
            self.mean, self.std_dev = np.zeros(ncols), np.ones(ncols)   # :117-118
        _, s, vt = np.linalg.svd(centered, full_matrices=False)        # :124
        eigenvalues = s * s / (nrows - 1)                               # :131
        total = eigenvalues.sum()                                       # :132
        ratio = eigenvalues / total                                     # :133
        self.components = vt.T[:, :n_components].copy()                 # :144  V[:, :n]
        self.explained_variance_ratio = ratio[:n_components].copy()     # :145-149
        self.total_variance = total
        self.eigenvalues = eigenvalues

    def transform(self, data: np.ndarray) -> np.ndarray:
        centered = np.array(data, dtype=np.float64, copy=True)          # :169
        if self.center:
            centered -= self.mean                                       # :171-175
        if self.scale:
            centered /= self.std_dev                                    # :177-181
        return centered @ self.components                               # :184

    def compute_loadings(self) -> np.ndarray:
        return self.components.T * self.std_dev[None, :]                # :204-215


def select_features_hvg(m: Csr, n: int) -> np.ndarray:
    """FeatureSelection::HighlyVariable(n) (dim_red/mod.rs:135-140)."""
    return select_hvg(compute_variance(m, COLUMN), n)


def pca_inplace(m: Csr, n_components=None, center=None, scale=None, selected=None):
    """Returns (X_pca [N x n_pc], components [k x n_pc], explained_variance_ratio, mean, std).

    `selected` is the feature index list in selection order (None = FeatureSelection::None,
    all genes).  Mirrors dim_red/mod.rs:24-94; obsm["X_pca"] is the only output the
    reference stores (:105-106), the rest is returned for parity checks.
    """
    if selected is None:
        selected = np.arange(m.n_cols, dtype=np.uint64)                 # :154
    dense = densify_selected(m, selected)                               # :34
    k = len(selected)
    n_pc = min(2 if n_components is None else int(n_components), k)     # :52
    pca = Pca(n_components=n_pc,
              center=True if center is None else bool(center),          # :55
              scale=True if scale is None else bool(scale))             # :56
    pca.fit(dense)                                                      # :66
    scores = pca.transform(dense)                                       # :69
    return scores, pca.components, pca.explained_variance_ratio, pca.mean, pca.std_dev
