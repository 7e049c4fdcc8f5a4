"""Shared test helpers: seeded restatements of the reference's test-data generator."""
import numpy as np

import oracle


def create_large_test_data(nrows, ncols, sparsity, seed=0, dtype=np.float64, lo=0.0, hi=50.0):
    """src/memory/processing/mod.rs:343-376: nrows*ncols/sparsity random COO pushes with
    Uniform(0,50) values, duplicates summed on COO->CSR — seeded here (the reference uses an
    unseeded thread_rng).  Returns a reference-layout oracle.Csr."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    nnz = int(round(nrows * ncols * (1.0 / sparsity)))
    r = rng.integers(0, nrows, nnz)
    c = rng.integers(0, ncols, nnz)
    if np.issubdtype(np.dtype(dtype), np.integer):
        v = rng.integers(1, 6, nnz).astype(np.float64)   # small counts; sums of duplicates stay in range
    else:
        v = rng.uniform(lo, hi, nnz)
    m = sp.coo_matrix((v, (r, c)), shape=(nrows, ncols)).tocsr()
    m.sum_duplicates()
    m.sort_indices()
    return oracle.Csr(nrows, ncols, m.indptr, m.indices, m.data.astype(dtype))


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.maximum(np.abs(b), 1e-300)
    return float(np.max(np.abs(a - b) / den)) if a.size else 0.0
