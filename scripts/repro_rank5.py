"""Reproduction of fuzz_pca.py seed 83 case 13: 5 cells x 130 genes, 13 non-zeros, HVG 65 (6 non-empty columns), 5 components,
no centring / scaling — exact rank 5 = block width, k = 65 > 64."""
import ctypes as C
import sys
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, ".")
import singlerust_amd as sr
from singlerust_amd import _ffi as F
rng = np.random.default_rng(83)
for it in range(14):
    n = int(rng.choice([5, 6, 9, 33, 64, 65, 200, 1500, 4000])); g = int(rng.choice([2, 3, 17, 64, 130, 700, 2500]))
    dens = float(rng.choice([0.02, 0.1, 0.5])); store = int(rng.choice([1, 2]))
    cen, sc = int(rng.integers(0, 2)), int(rng.integers(0, 2)); hvg = int(rng.choice([2, 10, 64, 65, 300, 5000]))
    x = sp.random(n, g, density=dens, random_state=int(rng.integers(1 << 30)), format="csr",
                  data_rvs=lambda s: rng.integers(1, 30, s).astype(np.float64), dtype=np.float64)
    x.sort_indices()
    k = min(hvg, g); rank = min(k, n - (1 if cen else 0))
    npc = int(min(rng.choice([1, 2, 5, 30, 56, 57, 100]), rank))
ctx = sr.Context.default()
lib = F.lib()
for npc_try in (5, 4):
    a = sr.IMAnnData.new_basic(x, ctx=ctx, store=store)
    opts = F.PcaOpts(npc_try, cen, sc, -1, 0, 0, 0, 0.0, 13)
    res = F.PipelineResult()
    rc = lib.srx_pipeline(a.x().handle, 1e4, hvg, C.byref(opts), C.byref(res))
    print("n", n, "g", g, "hvg", hvg, "npc", npc_try, "cen", cen, "sc", sc, "store", store, "->", rc,
          (lib.srx_last_error(ctx.handle) or b"").decode()[:100], "resid", res.pca.residual, flush=True)
