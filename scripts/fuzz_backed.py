"""Backed sessions and CSC handles against the resident CSR pipeline over random shapes, chunk sizes and options
(development helper)."""
import ctypes as C
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, ".")
import singlerust_amd as sr
from singlerust_amd import _ffi as F, backed

ctx = sr.Context.default()
lib = F.lib()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 99)
bad = 0


def resident(x, hvg, npc, store):
    a = sr.IMAnnData.new_basic(x, ctx=ctx, store=store)
    opts = F.PcaOpts(npc, -1, -1, -1, 0, 0, 0, 0.0, 0)
    res = F.PipelineResult()
    rc = lib.srx_pipeline(a.x().handle, 1e4, hvg, C.byref(opts), C.byref(res))
    if rc:
        return rc, None
    k = int(res.pca.k)
    scores, comps, evr, hv = np.zeros((x.shape[0], npc)), np.zeros((k, npc)), np.zeros(npc), np.zeros(k, np.uint64)
    F.check(lib.srx_result_fetch(a.x().handle, F.ptr(scores), F.ptr(comps), F.ptr(evr), None, None, F.ptr(hv)), ctx.handle)
    return 0, (scores, comps, evr, hv)


def close(a, b, tol):
    w = 0.0
    for c in range(b.shape[1]):
        s = 1.0 if np.dot(a[:, c], b[:, c]) >= 0 else -1.0
        nb = np.linalg.norm(b[:, c])
        w = max(w, np.linalg.norm(a[:, c] - s * b[:, c]) / (nb if nb > 0 else 1.0))
    return w < tol


for it in range(60):
    n = int(rng.choice([30, 200, 1500, 6000])); g = int(rng.choice([40, 300, 2000])); dens = float(rng.choice([0.03, 0.2]))
    store = int(rng.choice([1, 2])); hvg = int(rng.choice([20, 100, 1000])); npc = int(rng.choice([2, 10, 40]))
    x = sp.random(n, g, density=dens, random_state=int(rng.integers(1 << 30)), format="csr",
                  data_rvs=lambda s: rng.integers(1, 30, s).astype(np.float32), dtype=np.float32)
    x.sort_indices()
    npc = min(npc, min(hvg, g), n - 1)
    rc, ref = resident(x, hvg, npc, store)
    tag = f"n={n} g={g} dens={dens} store={store} hvg={hvg} npc={npc}"
    if rc:
        print("resident refused", tag, rc); continue
    tol = 1e-4 if store == 1 else 1e-6
    gaps_ok = np.ones(npc, bool)
    ev = ref[2]
    gaps_ok[:-1] &= (ev[:-1] - ev[1:]) > 1e-3 * ev[:-1]
    gaps_ok[1:] &= (ev[:-1] - ev[1:]) > 1e-3 * ev[:-1]
    gaps_ok &= ev > 1e-8 * ev[0]                 # vectors of numerically zero eigenvalues are arbitrary
    # backed session
    chunk = int(rng.choice([1, 7, 100, 1000, 10 ** 6])) if n <= 200 else int(rng.choice([100, 777, 5000]))
    bx = backed.BackedCsr(x.indptr.astype(np.uint64), x.indices.astype(np.uint64), x.data, g)
    try:
        r = backed.processing.pca_pipeline(backed.BackedAnnData(bx, ctx), chunk, 1e4, hvg, npc, store=store)
        ok = np.array_equal(r.selected, ref[3]) and np.allclose(r.explained_variance_ratio, ref[2], rtol=1e-5, atol=1e-12)
        if gaps_ok.any():
            ok &= close(r.x_pca[:, gaps_ok], ref[0][:, gaps_ok], tol)
    except Exception as e:
        ok = False
        print("backed exception", tag, chunk, repr(e)[:100])
    if not ok:
        bad += 1; print("MISMATCH backed", tag, "chunk", chunk)
    # CSC handle (its HVG ranking uses the CSC variance formula: compare eigenvalues only when the selections agree)
    rc2, c = resident(x.tocsc(), hvg, npc, store)
    if rc2:
        if rc2 != F.E_NAN:                      # an empty gene makes the CSC variance NaN, as in the reference
            bad += 1; print("CSC failed", tag, rc2)
    elif set(c[3].tolist()) == set(ref[3].tolist()):
        if not np.allclose(c[2], ref[2], rtol=1e-5, atol=1e-12):
            bad += 1; print("MISMATCH csc", tag)
print("backed / CSC fuzz: 60 cases,", bad, "problems")
