"""Development helper: small-k pipeline cases looking for non-finite scores."""
import ctypes as C
import sys
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import singlerust_amd as sr
from singlerust_amd import _ffi as F
ctx = sr.Context.default(); lib = F.lib()
bad = 0
for seed in range(40):
    rng = np.random.default_rng(seed)
    n, g, dens, hvg, npc = 4000, 130, 0.1, 10, 5
    x = sp.random(n, g, density=dens, random_state=seed, format="csr", data_rvs=lambda s: rng.integers(1, 30, s).astype(np.float64), dtype=np.float64)
    x.sort_indices()
    a = sr.IMAnnData.new_basic(x, ctx=ctx, store=1)
    opts = F.PcaOpts(npc, 0, 1, -1, 0, 0, 0, 0.0, seed)
    res = F.PipelineResult()
    rc = lib.srx_pipeline(a.x().handle, 1e4, hvg, C.byref(opts), C.byref(res))
    if rc: print("rc", rc); continue
    kk = int(res.pca.k)
    scores, comps = np.zeros((n, npc)), np.zeros((kk, npc))
    F.check(lib.srx_result_fetch(a.x().handle, F.ptr(scores), F.ptr(comps), None, None, None, None), ctx.handle)
    if not np.isfinite(scores).all():
        bad += 1
        r, c = np.where(~np.isfinite(scores))
        print("seed", seed, "non-finite scores:", len(r), "rows", np.unique(r)[:10], "cols", np.unique(c), "comps finite", np.isfinite(comps).all())
print("bad", bad)
