"""Differential fuzz of the statistics / normalise / log1p entry points (CSR and CSC handles) against the C oracle
(development helper; the oracle is the checker only)."""
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle
from oracle import COLUMN, ROW
import singlerust_amd as sr
from singlerust_amd.memory import processing, statistics as st

ctx = sr.Context.default()
rng = np.random.default_rng(777)
bad = 0
dts = [np.float32, np.float64, np.uint8, np.int16, np.uint16, np.int32, np.uint32]
for it in range(200):
    n, g = int(rng.integers(1, 300)), int(rng.choice([1, 7, 64, 500, 9000, 70000]))
    dens = float(rng.choice([0.0, 0.002, 0.05, 0.5]))
    if n * g * dens > 3e6:
        dens = 3e6 / (n * g)
    dt = dts[int(rng.integers(len(dts)))]
    x = sp.random(n, g, density=dens, random_state=int(rng.integers(1 << 30)), format="csr",
                  data_rvs=lambda s: rng.integers(1, 100, s).astype(np.float64), dtype=np.float64).astype(dt)
    x.sort_indices()
    m = oracle.Csr(n, g, x.indptr, x.indices, x.data)
    exact = np.issubdtype(dt, np.integer)
    for fmt in ("csr", "csc"):
        try:
            a = sr.IMAnnData.new_basic(x if fmt == "csr" else x.tocsc(), ctx=ctx)
            for d, od in ((sr.Direction.Row, ROW), (sr.Direction.Column, COLUMN)):
                ok = np.array_equal(st.compute_number(a, d), oracle.compute_number(m, od))
                s1, s0 = st.compute_sum(a, d), oracle.compute_sum(m, od)
                ok &= np.array_equal(s1, s0) if exact else np.allclose(s1, s0, rtol=1e-6 if dt == np.float32 else 1e-13)
                mn1, mx1 = st.compute_min_max(a, d); mn0, mx0 = oracle.compute_min_max(m, od)
                ok &= np.array_equal(mn1, mn0) and np.array_equal(mx1, mx0)
                if fmt == "csr":                                   # (the CSC variance formulas differ by direction: test_csc_gpu)
                    v1, v0 = st.compute_variance(a, d), oracle.compute_variance(m, od)
                    ok &= np.array_equal(np.isnan(v1), np.isnan(v0)) and np.allclose(v1[~np.isnan(v0)], v0[~np.isnan(v0)], rtol=1e-6, atol=1e-9)
                if not ok:
                    bad += 1
                    print("MISMATCH stats", fmt, n, g, dens, dt.__name__, d)
            for d, od in ((sr.Direction.Row, ROW), (sr.Direction.Column, COLUMN)):
                b = a.deep_clone()
                processing.normalize_total_inplace(b, 1e4, d)
                processing.log1p_transform_inplace(b)
                want = oracle.log1p_transform(oracle.normalize_total(m, 1e4, od)).values.astype(np.float64)
                got = b.x_values(np.float64)
                if fmt == "csc":
                    got = sp.csc_matrix((got, x.tocsc().indices, x.tocsc().indptr), shape=(n, g)).tocsr()
                    got.sort_indices(); got = got.data
                tol = 1e-6 if b.x().info().store == 1 else 4e-16
                if got.shape != want.shape or (want.size and np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-300)) > tol):
                    bad += 1
                    print("MISMATCH normalise", fmt, n, g, dens, dt.__name__, d)
        except Exception as e:
            bad += 1
            print("exception", fmt, n, g, dens, dt.__name__, repr(e)[:120])
print("stats fuzz: 400 matrices,", bad, "problems")
