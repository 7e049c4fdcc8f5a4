"""Differential fuzz of the fused pipeline / pca_inplace against the exact-SVD oracle over random shapes and options
(development helper; the oracle is the checker only)."""
import ctypes as C
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle
from oracle import ROW
import singlerust_amd as sr
from singlerust_amd import _ffi as F

ctx = sr.Context.default()
lib = F.lib()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 2024)
bad = skipped = 0
for it in range(160):
    n = int(rng.choice([5, 6, 9, 33, 64, 65, 200, 1500, 4000]))
    g = int(rng.choice([2, 3, 17, 64, 130, 700, 2500]))
    dens = float(rng.choice([0.02, 0.1, 0.5]))
    store = int(rng.choice([1, 2]))
    cen, sc = int(rng.integers(0, 2)), int(rng.integers(0, 2))
    hvg = int(rng.choice([2, 10, 64, 65, 300, 5000]))
    x = sp.random(n, g, density=dens, random_state=int(rng.integers(1 << 30)), format="csr",
                  data_rvs=lambda s: rng.integers(1, 30, s).astype(np.float64), dtype=np.float64)
    x.sort_indices()
    k = min(hvg, g)
    rank = min(k, n - (1 if cen else 0))
    npc = int(min(rng.choice([1, 2, 5, 30, 56, 57, 100]), rank))
    if npc < 1:
        continue
    a = sr.IMAnnData.new_basic(x, ctx=ctx, store=store)
    opts = F.PcaOpts(npc, cen, sc, -1, 0, 0, 0, 0.0, it)
    res = F.PipelineResult()
    rc = lib.srx_pipeline(a.x().handle, 1e4, hvg, C.byref(opts), C.byref(res))
    tag = f"n={n} g={g} dens={dens} store={store} cen={cen} sc={sc} hvg={hvg} npc={npc}"
    if rc != 0:
        msg = (lib.srx_last_error(ctx.handle) or b"").decode()
        # legitimate refusals: too few features / cells, zero matrix
        if rc == F.E_SHAPE or "block lost rank" in msg and x.nnz == 0:
            skipped += 1
            continue
        m = oracle.Csr(n, g, x.indptr, x.indices, x.data)
        lg = oracle.log1p_transform(oracle.normalize_total(m, 1e4, ROW))
        if "block lost rank" in msg:
            # more components asked for than the selected columns have numerical rank, with k > 64 (k <= 64 goes to the
            # exact eigen-solver and takes any rank): documented refusal (include/srx.h, srx_pca)
            sel = oracle.select_hvg(oracle.compute_variance(lg, 1), hvg)
            zz = oracle.densify_selected(lg, sel)
            zz = zz - (zz.mean(axis=0) if cen else 0.0)
            if len(sel) > 64 and np.linalg.matrix_rank(zz) < npc:
                skipped += 1
                continue
        print("FAIL", tag, rc, msg[:90]); bad += 1
        continue
    kk = int(res.pca.k)
    scores, comps, evr, hv = np.zeros((n, npc)), np.zeros((kk, npc)), np.zeros(npc), np.zeros(kk, np.uint64)
    F.check(lib.srx_result_fetch(a.x().handle, F.ptr(scores), F.ptr(comps), F.ptr(evr), None, None, F.ptr(hv)), ctx.handle)
    m = oracle.Csr(n, g, x.indptr, x.indices, x.data)
    lg = oracle.log1p_transform(oracle.normalize_total(m, 1e4, ROW))
    dense = oracle.densify_selected(lg, hv)
    sd = dense.std(axis=0)
    z = dense - (dense.mean(axis=0) if cen else 0.0)
    if sc:
        z = z / np.where(sd > 0, sd, 1.0)
    s = np.linalg.svd(z, compute_uv=False)
    th = (s * s)[:npc]
    tot = (s * s).sum()
    tol = 2e-5 if store == 1 else 1e-7
    ok = np.isfinite(scores).all() and np.isfinite(comps).all()
    # components whose eigenvalue is resolved by the storage precision: f32 values carry ~1e-7 theta_1 of input rounding
    # (f64: pairs below 1e-5 theta_1 are judged on that scale by the solver, and behind a deflation round they inherit
    #  ~tol * theta_1 / theta_i of the locked vectors' error)
    sig = th > (1e-2 if store == 1 else 1e-5) * th[0]
    if ok and tot > 0:
        ok &= np.allclose(evr[sig], th[sig] / tot, rtol=10 * tol)
        cv = z.T @ (z @ comps)
        ok &= bool((np.linalg.norm(cv[:, sig] - comps[:, sig] * th[sig], axis=0) / th[sig]).max() < 10 * tol) if sig.any() else True
        ok &= np.abs(scores - z @ comps).max() <= 10 * tol * max(1.0, np.abs(scores).max())
    if not ok:
        bad += 1
        cv = z.T @ (z @ comps)
        print("MISMATCH", tag, "resid", res.pca.residual, "| evr err", np.abs(evr[sig] / (th[sig] / tot) - 1).max() if sig.any() else None,
              "| eig resid", (np.linalg.norm(cv[:, sig] - comps[:, sig] * th[sig], axis=0) / th[sig]).max() if sig.any() else None,
              "| scores", np.abs(scores - z @ comps).max(), "| n_sig", int(sig.sum()), "th ratio", th[sig][-1] / th[0])
print(f"pca fuzz: 160 cases, {skipped} refused as out of contract, {bad} problems")
