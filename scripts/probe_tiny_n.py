"""Probe: very few cells (N <= 9) with k = 65 > 64 selected features — the block is as wide as the rank bound."""
import ctypes as C
import sys
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, ".")
import singlerust_amd as sr
from singlerust_amd import _ffi as F
ctx = sr.Context.default()
lib = F.lib()
rng = np.random.default_rng(7)
fails = 0
for n in (5, 6, 9):
    for dens in (0.02, 0.1, 0.5):
        for cen in (0, 1):
            for sc in (0, 1):
                x = sp.random(n, 130, density=dens, random_state=int(rng.integers(1 << 30)), format="csr",
                              data_rvs=lambda s: rng.integers(1, 30, s).astype(np.float64), dtype=np.float64)
                x.sort_indices()
                rank = n - cen
                for npc in (rank, max(1, rank - 2)):
                    a = sr.IMAnnData.new_basic(x, ctx=ctx, store=2)
                    opts = F.PcaOpts(npc, cen, sc, -1, 0, 0, 0, 0.0, 1)
                    res = F.PipelineResult()
                    rc = lib.srx_pipeline(a.x().handle, 1e4, 65, C.byref(opts), C.byref(res))
                    if rc != 0:
                        fails += 1
                        print(f"n={n} dens={dens} cen={cen} sc={sc} npc={npc} nnz={x.nnz} rows={np.diff(x.indptr)} -> {rc}",
                              (lib.srx_last_error(ctx.handle) or b"").decode()[:60], flush=True)
print("tiny-N probe:", fails, "failures of 72")
