"""Differential fuzz of the kernel-level entry srx_spmm (forward product, transposed product, sparse Gram) against scipy over
random shapes in ONE process (scratch buffers are reused across shapes: stale contents must not matter).  Integer-valued
matrices: the Gram matrix must match exactly.  Development helper; scipy is the checker."""
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, "."); sys.path.insert(0, "tests")
import singlerust_amd as sr
from singlerust_amd import _ffi as F

ctx = sr.Context.default()
lib = F.lib()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for it in range(N):
    n = int(rng.choice([1, 7, 100, 513, 1025, 5000, 30000, 70000]))
    g = int(rng.choice([70, 200, 900, 3000, 6000]))
    dens = float(rng.choice([0.005, 0.03, 0.15, 0.4]))
    if n * g * dens > 6e7:
        dens = 6e7 / (n * g)
    k = int(rng.integers(65, min(g, 3000) + 1))
    store = int(rng.choice([1, 2]))
    x = sp.random(n, g, density=dens, random_state=int(rng.integers(1 << 30)), format="csr",
                  data_rvs=lambda s: rng.integers(1, 9, s).astype(np.float64), dtype=np.float64)
    x.sort_indices()
    a = sr.IMAnnData.new_basic(x, ctx=ctx, store=store)
    sel = np.sort(rng.choice(g, k, replace=False)).astype(np.uint64)
    P = rng.standard_normal((k, 64))
    y, t, gram = np.zeros((n, 64)), np.zeros((k, 64)), np.zeros((k, k))
    rc = lib.srx_spmm(a.x().handle, F.ptr(sel), k, F.ptr(P), F.ptr(y), F.ptr(t), F.ptr(gram))
    tag = f"n={n} g={g} dens={dens:.4f} k={k} store={store} nnz={x.nnz}"
    if rc != 0:
        print("FAIL", tag, rc, (lib.srx_last_error(ctx.handle) or b"").decode()[:100]); bad += 1
        continue
    A = x[:, sel.astype(np.int64)]
    Pu = P.astype(np.float32).astype(np.float64) if store == 1 else P
    want_y = A @ Pu
    scale = max(1.0, np.abs(want_y).max())
    ok = np.isfinite(y).all() and np.abs(y - want_y).max() / scale < (2e-6 if store == 1 else 1e-12)
    want_g = (A.T @ A).toarray()
    ok_g = np.array_equal(gram, want_g)
    want_t = A.T @ y
    ok_t = np.abs(t - want_t).max() <= 1e-12 * max(1.0, np.abs(want_t).max())
    if not (ok and ok_g and ok_t):
        bad += 1
        print("MISMATCH", tag, "fwd", ok, "gram", ok_g, "t", ok_t)
print(f"gram / spmm fuzz: {N} cases, {bad} problems")
