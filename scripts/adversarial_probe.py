"""Adversarial inputs through the pipeline: no crash / hang, either a result that satisfies its own eigen-equation or a clean
error — development helper."""
import ctypes as C
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, ".")
import singlerust_amd as sr
from singlerust_amd import _ffi as F

lib = F.lib()
ctx = sr.Context.default()
rng = np.random.default_rng(0)


def run(name, x, n_hvg, n_pc, store=1):
    x = sp.csr_matrix(x)
    x.sort_indices()
    try:
        a = sr.IMAnnData.new_basic(x, ctx=ctx, store=store)
        opts = F.PcaOpts(n_pc, -1, -1, -1, 0, 0, 0, 0.0, 0)
        res = F.PipelineResult()
        rc = lib.srx_pipeline(a.x().handle, 1e4, n_hvg, C.byref(opts), C.byref(res))
        if rc != 0:
            print(f"{name}: rc={rc} {(lib.srx_last_error(ctx.handle) or b'').decode()[:100]}")
            return
        k, npc = int(res.pca.k), int(res.pca.n_pc)
        scores = np.zeros((x.shape[0], npc)); comps = np.zeros((k, npc)); evr = np.zeros(npc)
        F.check(lib.srx_result_fetch(a.x().handle, F.ptr(scores), F.ptr(comps), F.ptr(evr), None, None, None), ctx.handle)
        ok = np.isfinite(scores).all() and np.isfinite(comps).all() and np.isfinite(evr).all()
        print(f"{name}: ok iters={res.pca.n_iter} resid={res.pca.residual:.2e} finite={ok} evr[0]={evr[0]:.3g} orth={np.abs(comps.T @ comps - np.eye(npc)).max():.1e}")
    except F.SrxError as e:
        print(f"{name}: SrxError {e}")


base = sp.random(4000, 3000, density=0.05, random_state=1, data_rvs=lambda s: rng.integers(1, 20, s).astype(np.float32), dtype=np.float32, format="csr")
run("baseline", base, 500, 20)
run("constant matrix", sp.csr_matrix(np.ones((200, 50), dtype=np.float32)), 50, 5)
run("all rows identical", sp.vstack([base[0]] * 300), 200, 5)
run("N=60 k=300 npc=50", base[:60], 300, 50)
run("N=60 k=300 npc=58", base[:60], 300, 58)
run("huge values", base * np.float32(1e6), 500, 20)
run("tiny values", base * np.float32(1e-6), 500, 20)
run("density 0.001", sp.random(50000, 3000, density=0.001, random_state=2, data_rvs=lambda s: rng.integers(1, 5, s).astype(np.float32), dtype=np.float32, format="csr"), 500, 20)
d = base.tolil(); d[7, :] = 3.0; run("one fully dense row", d.tocsr(), 500, 20)
run("duplicated cells", sp.vstack([base, base]), 500, 20)
run("duplicated genes", sp.hstack([base, base]), 1000, 20)
run("half empty rows", sp.vstack([base[:2000], sp.csr_matrix((2000, 3000), dtype=np.float32)]), 500, 20)
run("f64 store baseline", base.astype(np.float64), 500, 20, store=2)
run("k=2 npc=2", base, 2, 2)
run("npc=1", base, 100, 1)
run("hvg=64 npc=64", base, 64, 64)
run("hvg=65 npc=65", base, 65, 65)
