"""Differential fuzz of filter_cells / filter_genes against oracle/filter_oracle.py (development helper; the oracle is used
as the checker only)."""
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle
from oracle import filter_oracle as fo
import singlerust_amd as sr
from singlerust_amd import _ffi as F
from singlerust_amd.memory import processing

ctx = sr.Context.default()
rng = np.random.default_rng(12345)
bad = 0


def flex(kind, v):
    if kind == 0:
        return sr.FlexValue.NoLimit(), fo.NONE
    if kind == 1:
        return sr.FlexValue.Absolute(int(v)), fo.absolute(int(v))
    return sr.FlexValue.Relative(float(v)), fo.relative(float(v))


for it in range(300):
    n, g = int(rng.integers(1, 60)), int(rng.integers(1, 40))
    dens = float(rng.choice([0.0, 0.05, 0.3, 0.9]))
    x = sp.random(n, g, density=dens, random_state=int(rng.integers(1 << 30)), format="csr",
                  data_rvs=lambda s: rng.integers(1, 6, s).astype(np.float32), dtype=np.float32)
    x.sort_indices()
    m = oracle.Csr(n, g, x.indptr, x.indices, x.data)
    lk, uk = int(rng.integers(0, 3)), int(rng.integers(0, 3))
    lv = rng.integers(0, 12) if lk == 1 else rng.choice([0.0, 0.1, 0.25, 0.5, 1.0])
    uv = rng.integers(0, 30) if uk == 1 else rng.choice([0.0, 0.5, 0.75, 0.9, 1.0])
    (lo, olo), (hi, ohi) = flex(lk, lv), flex(uk, uv)
    for which in ("cells", "genes"):
        try:
            a = sr.IMAnnData.new_basic(x, ctx=ctx)
            got = processing.filter_cells(a, lo, hi) if which == "cells" else processing.filter_genes(a, lo, hi)
            want, _ = fo.filter_cells(m, olo, ohi) if which == "cells" else fo.filter_genes(m, olo, ohi)
            ok = (got.n_obs(), got.n_vars()) == (want.n_rows, want.n_cols) and np.array_equal(got.x_values(np.float64), want.values.astype(np.float64))
            if ok and want.n_rows:
                ok = np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)
        except Exception as e:
            ok = False
            print("exception", which, n, g, dens, lk, lv, uk, uv, repr(e)[:100])
        if not ok:
            bad += 1
            print("MISMATCH", which, n, g, dens, lk, lv, uk, uv)
print("filter fuzz: 600 cases,", bad, "mismatches")
