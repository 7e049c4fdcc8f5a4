import ctypes as C, sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import singlerust_amd as sr
from singlerust_amd import _ffi as F
ctx = sr.Context.default(); lib = F.lib()
rng = np.random.default_rng(2024)
# replay the fuzz sequence up to the failing cases
cases = []
for it in range(160):
    n = int(rng.choice([5, 6, 9, 33, 64, 65, 200, 1500, 4000])); g = int(rng.choice([2, 3, 17, 64, 130, 700, 2500]))
    dens = float(rng.choice([0.02, 0.1, 0.5])); store = int(rng.choice([1, 2])); cen, sc = int(rng.integers(0, 2)), int(rng.integers(0, 2))
    hvg = int(rng.choice([2, 10, 64, 65, 300, 5000]))
    x = sp.random(n, g, density=dens, random_state=int(rng.integers(1 << 30)), format="csr", data_rvs=lambda s: rng.integers(1, 30, s).astype(np.float64), dtype=np.float64)
    x.sort_indices(); k = min(hvg, g); rank = min(k, n - (1 if cen else 0))
    npc = int(min(rng.choice([1, 2, 5, 30, 56, 57, 100]), rank))
    if npc < 1: continue
    if (n, g, dens, hvg) in ((33, 17, 0.02, 10), (64, 2500, 0.02, 64), (64, 64, 0.02, 64)):
        a = sr.IMAnnData.new_basic(x, ctx=ctx, store=store)
        opts = F.PcaOpts(npc, cen, sc, -1, 0, 0, 0, 0.0, it); res = F.PipelineResult()
        print('CASE', n, g, dens, store, cen, sc, hvg, npc, 'nnz', x.nnz, 'nonempty rows', int((np.diff(x.indptr) > 0).sum()), flush=True)
        rc = lib.srx_pipeline(a.x().handle, 1e4, hvg, C.byref(opts), C.byref(res))
        print('  rc', rc, (lib.srx_last_error(ctx.handle) or b'').decode()[:100], flush=True)
