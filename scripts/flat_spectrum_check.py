import sys, numpy as np, scipy.sparse as sp
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import singlerust_amd as sr
from singlerust_amd.memory import processing
from singlerust_amd.memory.processing import dim_red
rng = np.random.default_rng(0)
x = sp.random(5000, 800, density=0.05, random_state=1, format="csr", data_rvs=lambda s: rng.integers(1, 20, s).astype(np.float32)).astype(np.float32)
x.sort_indices()
ctx = sr.Context(0)
a = sr.IMAnnData.new_basic((5000, 800, x.indptr, x.indices, x.data), ctx=ctx, store=1)
processing.normalize_total_inplace(a, 1e4, sr.Direction.Row); processing.log1p_transform_inplace(a)
for tol, mi in ((0, 0), (1e-3, 0), (1e-9, 30)):
    try:
        info = dim_red.pca_inplace(a, 10, None, None, None, sr.FeatureSelection.HighlyVariable(300), None, tol=tol, max_iter=mi)
        print("tol", tol, "-> n_iter", info.n_iter, "residual", info.residual)
    except sr.SrxError as e:
        print("tol", tol, "-> error", e.code, str(e)[:120])
