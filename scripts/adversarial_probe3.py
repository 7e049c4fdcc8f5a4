"""Option corner cases of pca_inplace: all components, extreme tolerances and budgets, value dtypes — development helper."""
import sys
import traceback

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle
from oracle import pca_oracle, ROW
import singlerust_amd as sr
from singlerust_amd import _ffi as F
from singlerust_amd.memory import processing
from singlerust_amd.memory.processing import dim_red

ctx = sr.Context.default()
rng = np.random.default_rng(0)
base = sp.random(4000, 600, density=0.08, random_state=1, data_rvs=lambda s: rng.integers(1, 20, s).astype(np.float64), dtype=np.float64, format="csr")
base.sort_indices()


def col_err(got, ref):
    w = 0.0
    for c in range(ref.shape[1]):
        s = 1.0 if np.dot(got[:, c], ref[:, c]) >= 0 else -1.0
        e = np.linalg.norm(got[:, c] - s * ref[:, c]) / np.linalg.norm(ref[:, c])
        w = max(w, e) if np.isfinite(e) else float("inf")
    return w


def case(name, x, hvg, npc, store=2, **kw):
    try:
        a = sr.IMAnnData.new_basic(x, ctx=ctx, store=store)
        processing.normalize_total_inplace(a, 1e4, sr.Direction.Row)
        processing.log1p_transform_inplace(a)
        info = dim_red.pca_inplace(a, npc, None, None, None, sr.FeatureSelection.HighlyVariable(hvg), None, **kw)
        m = oracle.Csr(x.shape[0], x.shape[1], x.indptr, x.indices, x.data)
        lg = oracle.log1p_transform(oracle.normalize_total(m, 1e4, ROW))
        want, wc, wevr, *_ = pca_oracle.pca_inplace(lg, npc, None, None, a.uns["pca"]["selected_features"])
        evr = a.uns["pca"]["explained_variance_ratio"]
        print(f"{name}: ok iters={info.n_iter} resid={info.residual:.1e} evr_err={np.abs(evr / wevr[:len(evr)] - 1).max():.1e} "
              f"scores_err(first 5)={col_err(a.obsm['X_pca'][:, :5], want[:, :5]):.1e}", flush=True)
    except F.SrxError as e:
        print(f"{name}: SrxError {str(e)[:120]}", flush=True)
    except Exception:
        print(f"{name}: EXCEPTION {traceback.format_exc().splitlines()[-1][:160]}", flush=True)


case("k=200 npc=200 (all)", base, 200, 200)
case("k=100 npc=99", base, 100, 99)
case("k=64 npc=64", base, 64, 64)
case("k=66 npc=66", base, 66, 66)
case("tol 1e-14", base, 300, 10, tol=1e-14)
case("tol 1e-2", base, 300, 10, tol=1e-2)
case("max_iter 1", base, 300, 10, max_iter=1)
case("f32 store", base.astype(np.float32), 300, 20, store=1)
for dt in (np.int8, np.int16, np.int32, np.uint8, np.uint16, np.uint32):
    case(f"dtype {np.dtype(dt).name}", base.astype(dt), 300, 10, store=0)
case("solver 2 k=300", base, 300, 20, solver=2)
case("solver 2 npc 56", base, 300, 56, solver=2)
