"""Generates tests/golden/*.npz — small input/expected-output vectors for the hot path.

The reference (Rust, un-buildable here: no rustc/cargo, un-vendored deps) ships no golden
vectors; its only numeric pin is the row/column-sum property of
src/memory/processing/mod.rs:419-481.  The expected outputs below are therefore computed by
INDEPENDENT maths (numpy / scipy.sparse, not the oracle and not the GPU library), following
the reference's definitions:
  number   csr.rs:16-38     sum      csr.rs:81-102    variance csr.rs:149-188 (nz-only, ddof 0)
  normalise scale/mod.rs:7-23,59-89   log1p transform/mod.rs:36-57   HVG dim_red/mod.rs:135-140
  PCA      src/shared/processing/pca/mod.rs:74-215 via sklearn StandardScaler + PCA(full)
plus the hand-derived 4x5 known-answer test of SURVEY.md §8(c).
Run:  python tests/golden/make_golden.py
"""
import os

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))


def expected(m: sp.csr_matrix, target=1e4, n_hvg=None, n_pc=None):
    m = m.astype(np.float64)
    N, G = m.shape
    out = {}
    out["number_row"] = np.diff(m.indptr).astype(np.uint32)
    out["number_col"] = np.bincount(m.indices, minlength=G).astype(np.uint32)
    out["sum_row"] = np.asarray(m.sum(axis=1)).ravel()
    out["sum_col"] = np.asarray(m.sum(axis=0)).ravel()
    cnt = out["number_col"].astype(np.float64)
    sq = np.asarray(m.multiply(m).sum(axis=0)).ravel()
    with np.errstate(invalid="ignore", divide="ignore"):
        var_col = np.where(cnt > 0, sq / cnt - (out["sum_col"] / cnt) ** 2, 0.0)
    out["var_col"] = var_col
    # normalise rows then log1p
    s = out["sum_row"]
    scale = np.where(s == 0, 0.0, target / np.where(s == 0, 1.0, s))
    norm = sp.diags(scale) @ m
    norm = sp.csr_matrix(norm)
    norm.sort_indices()
    # keep the pattern of m (explicit zeros from scale 0 cannot appear: empty rows have no entries)
    assert np.array_equal(norm.indptr, m.indptr) and np.array_equal(norm.indices, m.indices)
    out["norm_values"] = norm.data.copy()
    lg = norm.copy()
    lg.data = np.log1p(lg.data)
    out["log_values"] = lg.data.copy()
    cntl = np.bincount(lg.indices, minlength=G).astype(np.float64)
    sl = np.asarray(lg.sum(axis=0)).ravel()
    ql = np.asarray(lg.multiply(lg).sum(axis=0)).ravel()
    with np.errstate(invalid="ignore", divide="ignore"):
        out["log_var_col"] = np.where(cntl > 0, ql / cntl - (sl / cntl) ** 2, 0.0)
    if n_hvg:
        order = np.argsort(-out["log_var_col"], kind="stable")[:n_hvg]
        out["hvg"] = order.astype(np.uint64)
        if n_pc:
            from sklearn.decomposition import PCA
            from sklearn.preprocessing import StandardScaler
            dense = np.asarray(lg[:, order].todense())
            z = StandardScaler(with_mean=True, with_std=True).fit_transform(dense)   # ddof 0
            p = PCA(n_components=n_pc, svd_solver="full").fit(z)
            out["pca_scores"] = p.transform(z)
            out["pca_components"] = p.components_.T.copy()
            out["pca_evr"] = p.explained_variance_ratio_.copy()
    return out


def save(name, m, **kw):
    e = expected(m, **kw)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), n_rows=m.shape[0], n_cols=m.shape[1],
                        indptr=m.indptr.astype(np.uint64), indices=m.indices.astype(np.uint64),
                        values=m.data, **e)
    print(name, m.shape, m.nnz)


def planted(seed, n, g, n_types, block, density, rng_vals=True):
    """Small matrix with a clear low-rank structure so the top PCs are well separated."""
    rng = np.random.default_rng(seed)
    t = rng.integers(0, n_types, n)
    prob = np.full((n, g), density)
    for i in range(n_types):
        prob[np.ix_(t == i, np.arange(i * block, (i + 1) * block))] = 0.6 + 0.03 * i
    mask = rng.random((n, g)) < prob
    vals = 1 + rng.geometric(0.5, size=(n, g))
    boost = np.zeros((n, g))
    for i in range(n_types):
        boost[np.ix_(t == i, np.arange(i * block, (i + 1) * block))] = 3 + i
    dense = mask * (vals + boost * mask)
    dense[rng.integers(0, n, 3), :] = 0          # a few empty cells
    return sp.csr_matrix(dense.astype(np.float64))


if __name__ == "__main__":
    rng = np.random.default_rng(42)
    # 1) the reference's own test shape: 1000 x 100, ~10 % fill, Uniform(0,50) f64, duplicates summed
    r = rng.integers(0, 1000, 10000)
    c = rng.integers(0, 100, 10000)
    m = sp.coo_matrix((rng.uniform(0, 50, 10000), (r, c)), shape=(1000, 100)).tocsr()
    m.sum_duplicates(); m.sort_indices()
    save("ref_shape_1000x100", m, n_hvg=20)
    # 2) small integer counts with empty rows / empty genes / single-nnz rows
    d = (rng.random((64, 40)) < 0.15) * rng.integers(1, 9, (64, 40))
    d[5, :] = 0; d[17, :] = 0; d[:, 7] = 0; d[:, 39] = 0; d[9, :] = 0; d[9, 3] = 4
    save("counts_64x40", sp.csr_matrix(d.astype(np.float64)), n_hvg=10)
    # 3) planted-structure matrix for PCA parity
    save("planted_600x240", planted(3, 600, 240, 6, 12, 0.05), n_hvg=120, n_pc=5)
