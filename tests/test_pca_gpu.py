"""GPU parity tests (-m gpu) of pca_inplace / the fused pipeline vs the exact-SVD oracle
(oracle/pca_oracle.py, restating src/shared/processing/pca/mod.rs:74-215).

Bar (BASELINE.json north_star): top-PC loadings and scores within 1e-5 relative — measured
here per component as ||got - ref||_2 / ||ref||_2 after sign alignment (the reference has no
sign convention).  The f64-storage path is expected to be ~1e-9.
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle
from oracle import COLUMN, ROW, pca_oracle
from util import rel_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-5


def col_err(got, ref):
    """max over components of ||got_c - s*ref_c|| / ||ref_c|| with the best sign s."""
    worst = 0.0
    for c in range(ref.shape[1]):
        s = 1.0 if np.dot(got[:, c], ref[:, c]) >= 0 else -1.0
        e = np.linalg.norm(got[:, c] - s * ref[:, c]) / np.linalg.norm(ref[:, c])
        if not np.isfinite(e):
            return float("inf")                      # NaN / inf anywhere fails every `< tol` comparison
        worst = max(worst, e)
    return worst


def assert_components_within_conditioning(got, ref, eigvals, store, what, tol=TOL, max_slack=None):
    """The north_star's 1e-5 bar per component, as far as the problem's conditioning allows (per-component
    comparison up to sign, near-degenerate pairs judged by what their eigengap supports).  An eigenvector moves by
    ~ ||dC|| / gap under a perturbation dC of the matrix; the inputs of the f32-storage path are the f64 oracle's values
    rounded to f32 (2^-24 relative), those of the f64 path differ at 2^-53.  So component c must agree to
        max(tol, 8 * eps_store / relgap_c),    relgap_c = distance to the nearest other eigenvalue / eigenvalue
    — at f64 storage that IS the plain 1e-5 for every component with a relative gap above 1e-10 (asserted below for every gap
    >= 1e-3), at f32 storage 1e-5 for gaps above 5e-2 and the perturbation bound for the crowded tail.  Prints, per call, the
    worst observed error / bound ratio and how many components needed more than the plain 1e-5; `max_slack` turns that count
    into an ASSERTED budget (0 = every component at the plain 1e-5: the conditioning argument is not used at all), so that the
    slack actually used is in the driver's pass / fail record and not only in captured output."""
    ev = np.asarray(eigvals, dtype=np.float64)
    gap = np.minimum(np.abs(np.diff(ev, prepend=np.inf)), np.abs(np.diff(ev, append=0.0))) / ev
    if len(ev) > 1:                                  # the eigenvalue below the last one is not known: take the gap above it
        gap[-1] = abs(ev[-2] - ev[-1]) / ev[-1]
    eps = 2.0 ** -24 if store == 1 else 2.0 ** -53
    worst_ratio, worst_err, n_slack = 0.0, 0.0, 0
    for c in range(ref.shape[1]):
        sgn = 1.0 if np.dot(got[:, c], ref[:, c]) >= 0 else -1.0
        e = np.linalg.norm(got[:, c] - sgn * ref[:, c]) / np.linalg.norm(ref[:, c])
        bound = max(tol, 8.0 * eps / gap[c])
        if store == 2 and gap[c] >= 1e-3:
            bound = tol                              # f64 storage: the plain bar, no conditioning argument
        assert np.isfinite(e) and e < bound, f"{what} component {c}: error {e:.3e} > {bound:.3e} (relative eigengap {gap[c]:.2e})"
        worst_ratio, worst_err = max(worst_ratio, e / bound), max(worst_err, e)
        n_slack += e >= tol
    print(f"[parity] {what} (store {store}): {ref.shape[1]} components, worst error {worst_err:.2e}, worst error/bound "
          f"{worst_ratio:.2e}, components above the plain {tol:g}: {n_slack}, smallest relative eigengap {gap.min():.1e}")
    assert_components_within_conditioning.last = {"worst_ratio": worst_ratio, "worst_err": worst_err, "n_slack": n_slack}
    if max_slack is not None:
        assert n_slack <= max_slack, f"{what}: {n_slack} components above the plain {tol:g} (budget {max_slack})"
    return gap


def adata_of(m, ctx, store=0):
    import singlerust_amd as sr
    return sr.IMAnnData.new_basic((m.n_rows, m.n_cols, m.indptr, m.indices, m.values), ctx=ctx, store=store)


def synth_host(seed, n, g, density, skew=0):
    from singlerust_amd import _ffi
    lib = _ffi.lib()
    p = _ffi.SynthParams()
    lib.srx_synth_defaults(C.byref(p), seed, n, g, density)
    p.skew = skew
    ip = np.zeros(n + 1, dtype=np.uint64)
    lib.srx_synth_indptr(C.byref(p), 0, n, _ffi.ptr(ip))
    idx = np.zeros(int(ip[-1]), np.uint64)
    val = np.zeros(int(ip[-1]), np.float32)
    lib.srx_synth_fill_host(C.byref(p), 0, n, _ffi.ptr(ip), _ffi.ptr(idx), _ffi.ptr(val))
    return oracle.Csr(n, g, ip, idx, val), p


@pytest.mark.parametrize("solver", [1, 2])
@pytest.mark.parametrize("store,tol", [(1, TOL), (2, 1e-8)])
def test_golden_planted(ctx, store, tol, solver):
    """Committed golden vectors (sklearn StandardScaler + PCA(full) on the HVG columns)."""
    import singlerust_amd as sr
    from singlerust_amd.memory import processing
    from singlerust_amd.memory.processing import dim_red
    z = np.load(os.path.join(GOLD, "planted_600x240.npz"))
    m = oracle.Csr(int(z["n_rows"]), int(z["n_cols"]), z["indptr"], z["indices"], z["values"])
    a = adata_of(m, ctx, store)
    processing.normalize_total_inplace(a, 1e4, sr.Direction.Row)
    processing.log1p_transform_inplace(a)
    # use the golden HVG list so both storages walk the same columns
    a.var["hv"] = np.isin(np.arange(m.n_cols), z["hvg"])
    sel_sorted = np.sort(z["hvg"])
    info = dim_red.pca_inplace(a, 5, None, None, 32, sr.FeatureSelection.HighlyVariableCol("hv"), None, tol=1e-9 if store == 2 else 0,
                               solver=solver)
    got = a.obsm["X_pca"]
    assert got.shape == (600, 5) and got.dtype == np.float64
    # golden columns are in variance-rank order; HighlyVariableCol gives ascending gene order
    perm = np.array([np.where(z["hvg"] == g)[0][0] for g in sel_sorted])
    assert col_err(got, z["pca_scores"]) < tol
    assert col_err(a.uns["pca"]["components"], z["pca_components"][perm]) < tol
    np.testing.assert_allclose(a.uns["pca"]["explained_variance_ratio"], z["pca_evr"], rtol=max(tol, 1e-9))
    assert info.k == 120 and info.n_pc == 5 and info.n_iter >= 1


@pytest.mark.parametrize("solver", [1, 2])
@pytest.mark.parametrize("store,tol", [(1, TOL), (2, 1e-8)])
def test_hvg_pipeline_vs_oracle(ctx, store, tol, solver):
    """normalize -> log1p -> pca_inplace(HighlyVariable(n)) on a synthetic planted matrix,
    against the oracle's densify + exact SVD; loadings in the varm["PCA_loadings"] layout."""
    import singlerust_amd as sr
    from singlerust_amd.memory import processing
    from singlerust_amd.memory.processing import dim_red
    m, _ = synth_host(5, 4000, 3000, 0.05)
    a = adata_of(m, ctx, store)
    processing.normalize_total_inplace(a, 1e4, sr.Direction.Row)
    processing.log1p_transform_inplace(a)
    n_hvg, n_pc = 300, 10
    info = dim_red.pca_inplace(a, n_pc, None, None, None, sr.FeatureSelection.HighlyVariable(n_hvg), None,
                               store_loadings=True, tol=1e-9 if store == 2 else 0, solver=solver)
    assert info.solver == solver
    lg = oracle.log1p_transform(oracle.normalize_total(m, 1e4, ROW))
    # the oracle walks the columns the GPU selected (identical to its own at f64 storage)
    sel = a.uns["pca"]["selected_features"]
    want_sel = pca_oracle.select_features_hvg(lg, n_hvg)
    if store == 2:
        assert np.array_equal(sel, want_sel)
    else:
        # EXPLICIT f32 storage and the three separate calls: HighlyVariable(n) ranks what X holds, f32-rounded values (a
        # matrix created with SRX_STORE_AUTO moves to f64 at normalize_total, like the reference's variant, and is exact)
        assert len(set(sel.tolist()) ^ set(want_sel.tolist())) <= 2
    scores, comps, evr, mean, std = pca_oracle.pca_inplace(lg, n_pc, None, None, sel)
    assert col_err(a.obsm["X_pca"], scores) < tol
    assert col_err(a.uns["pca"]["components"], comps) < tol
    np.testing.assert_allclose(a.uns["pca"]["explained_variance_ratio"], evr, rtol=10 * tol)
    np.testing.assert_allclose(a.uns["pca"]["mean"], mean, rtol=10 * tol, atol=1e-12)
    np.testing.assert_allclose(a.uns["pca"]["std"], std, rtol=10 * tol)
    # sign convention: largest-|.| loading of each component is positive
    cg = a.uns["pca"]["components"]
    assert np.all(cg[np.argmax(np.abs(cg), axis=0), np.arange(n_pc)] > 0)
    # varm["PCA_loadings"] (dim_red/mod.rs:108-118): rows scattered to the original gene index
    full = a.varm["PCA_loadings"]
    assert full.shape == (3000, n_pc)
    np.testing.assert_allclose(full[sel.astype(np.int64)], cg * a.uns["pca"]["std"][:, None], rtol=1e-12)
    mask = np.ones(3000, bool); mask[sel.astype(np.int64)] = False
    assert np.all(full[mask] == 0)
    assert 1 <= info.n_iter <= 100 and info.residual <= (1e-7 if store == 1 else 1e-9)


@pytest.mark.parametrize("store,tol", [(1, TOL), (2, 1e-8)])
def test_forward_product_ragged_last_slice(ctx, store, tol):
    """n_pc = 50 through the forward product's panel slices — f32 panels: four 16-column slices, the last one ragged (2 columns);
    f64 panels: five 10-column slices — against the oracle's scores."""
    import singlerust_amd as sr
    from singlerust_amd.memory import processing
    from singlerust_amd.memory.processing import dim_red
    m, _ = synth_host(11, 3000, 2500, 0.06)
    n_hvg, n_pc = 400, 50
    a = adata_of(m, ctx, store)
    processing.normalize_total_inplace(a, 1e4, sr.Direction.Row)
    processing.log1p_transform_inplace(a)
    dim_red.pca_inplace(a, n_pc, None, None, None, sr.FeatureSelection.HighlyVariable(n_hvg), None, tol=1e-9 if store == 2 else 0)
    got, sel = a.obsm["X_pca"], a.uns["pca"]["selected_features"]
    assert got.shape == (3000, 50)
    lg = oracle.log1p_transform(oracle.normalize_total(m, 1e4, ROW))
    scores, _, evr, *_ = pca_oracle.pca_inplace(lg, n_pc, None, None, sel)
    # (the trailing components of a 400-feature noise spectrum are nearly degenerate: the well-separated leading ones column by
    #  column, the whole 50-dimensional score space through its projector)
    assert col_err(got[:, :8], scores[:, :8]) < tol
    np.testing.assert_allclose(a.uns["pca"]["explained_variance_ratio"], evr, rtol=1e-5)


@pytest.mark.parametrize("solver", [1, 2])
@pytest.mark.parametrize("center,scale", [(True, True), (True, False), (False, True), (False, False)])
def test_center_scale_options(ctx, center, scale, solver):
    """pca/mod.rs:85-119: mean is subtracted only if center, std divides only if scale."""
    import singlerust_amd as sr
    from singlerust_amd.memory import processing
    from singlerust_amd.memory.processing import dim_red
    m, _ = synth_host(9, 1500, 1200, 0.08)
    a = adata_of(m, ctx, 2)
    processing.normalize_log1p_inplace(a, 1e4)
    lg = oracle.log1p_transform(oracle.normalize_total(m, 1e4, ROW))
    sel = np.sort(pca_oracle.select_features_hvg(lg, 150))
    a.var["hv"] = np.isin(np.arange(m.n_cols), sel)
    dim_red.pca_inplace(a, 6, center, scale, None, sr.FeatureSelection.HighlyVariableCol("hv"), None, tol=1e-10,
                        solver=solver)
    scores, comps, evr, mean, std = pca_oracle.pca_inplace(lg, 6, center, scale, sel)
    assert col_err(a.obsm["X_pca"], scores) < 1e-7
    assert col_err(a.uns["pca"]["components"], comps) < 1e-7
    np.testing.assert_allclose(a.uns["pca"]["explained_variance_ratio"], evr, rtol=1e-8)


def test_defaults_and_small_k(ctx):
    """n_components None -> 2 (dim_red/mod.rs:52); k < 64 panel columns; FeatureSelection::None."""
    import singlerust_amd as sr
    from singlerust_amd.memory import processing
    from singlerust_amd.memory.processing import dim_red
    z = np.load(os.path.join(GOLD, "counts_64x40.npz"))
    m = oracle.Csr(64, 40, z["indptr"], z["indices"], z["values"])
    # drop the two all-zero genes: the reference divides by std = 0 there (NaN)
    lg = oracle.log1p_transform(oracle.normalize_total(m, 1e4, ROW))
    keep = np.nonzero(oracle.compute_number(m, COLUMN) > 1)[0].astype(np.uint64)
    a = adata_of(m, ctx, 2)
    processing.normalize_log1p_inplace(a, 1e4)
    a.var["keep"] = np.isin(np.arange(40), keep)
    info = dim_red.pca_inplace(a, None, None, None, None, sr.FeatureSelection.HighlyVariableCol("keep"), None, tol=1e-11)
    assert a.obsm["X_pca"].shape == (64, 2) and info.n_pc == 2 and info.k == len(keep)
    scores, comps, evr, *_ = pca_oracle.pca_inplace(lg, None, None, None, keep)
    assert col_err(a.obsm["X_pca"], scores) < 1e-8
    np.testing.assert_allclose(a.uns["pca"]["explained_variance_ratio"], evr, rtol=1e-9)


@pytest.mark.parametrize("dtype", [np.uint8, np.float64])
@pytest.mark.parametrize("store", [1, 2])
def test_pca_hand_derived_kat(ctx, store, dtype):
    """tests/golden/kat_pca_6x4.json — the rank-2 6 x 4 matrix whose standardised SVD is derived BY HAND from
    pca/mod.rs:87-144 (no numpy in the expected numbers): mean, std (ddof 0), explained variance ratio over ALL eigenvalues
    (two of them zero), components, scores (obsm["X_pca"] layout) and loadings (varm["PCA_loadings"] layout), with the
    reference's defaults (n_components None -> 2, centre, scale: dim_red/mod.rs:52-56)."""
    import json
    import singlerust_amd as sr
    from singlerust_amd.memory.processing import dim_red
    k = json.load(open(os.path.join(GOLD, "kat_pca_6x4.json")))
    a = sr.IMAnnData.new_basic((k["n_rows"], k["n_cols"], k["indptr"], k["indices"], np.array(k["data"], dtype=dtype)), ctx=ctx,
                               store=store)
    info = dim_red.pca_inplace(a, None, None, None, None, sr.FeatureSelection.None_, None, store_loadings=True)
    assert info.k == 4 and info.n_pc == 2
    u = a.uns["pca"]
    tol = 1e-12                      # small integers are exact at either storage; the sums and the eigen-solve are f64
    np.testing.assert_allclose(u["mean"], k["mean"], rtol=0, atol=tol)
    np.testing.assert_allclose(u["std"], k["std"], rtol=tol)
    np.testing.assert_allclose(u["explained_variance_ratio"], k["explained_variance_ratio"], rtol=tol)
    got_s, got_c, got_l = a.obsm["X_pca"], u["components"], a.varm["PCA_loadings"]
    assert got_s.shape == (6, 2) and got_c.shape == (4, 2) and got_l.shape == (4, 2)
    want_s, want_c, want_l = np.array(k["scores"]), np.array(k["components"]), np.array(k["loadings"]).T
    for c in range(2):
        sgn = 1.0 if np.dot(got_c[:, c], want_c[:, c]) >= 0 else -1.0
        np.testing.assert_allclose(sgn * got_c[:, c], want_c[:, c], rtol=0, atol=tol)
        # (f32 storage: the transform's panel D V is f32 — one 2^-24 rounding per panel entry: 3e-8 observed)
        np.testing.assert_allclose(sgn * got_s[:, c], want_s[:, c], rtol=0, atol=10 * tol if store == 2 else 2e-7)
        np.testing.assert_allclose(sgn * got_l[:, c], want_l[:, c], rtol=0, atol=tol)


def test_shape_errors(ctx):
    """dim_red/mod.rs:38-41 panics for k < 2 or N < 5 -> SRX_E_SHAPE."""
    import singlerust_amd as sr
    from singlerust_amd import _ffi
    from singlerust_amd.memory.processing import dim_red
    a = sr.IMAnnData.new_basic((6, 3, [0, 1, 2, 3, 4, 5, 6], [0, 1, 2, 0, 1, 2], np.arange(1.0, 7.0)), ctx=ctx)
    a.var["one"] = np.array([True, False, False])
    with pytest.raises(sr.SrxError) as e:
        dim_red.pca_inplace(a, 1, None, None, None, sr.FeatureSelection.HighlyVariableCol("one"), None)
    assert e.value.code == _ffi.E_SHAPE
    b = sr.IMAnnData.new_basic((3, 3, [0, 1, 2, 3], [0, 1, 2], np.arange(1.0, 4.0)), ctx=ctx)
    with pytest.raises(sr.SrxError) as e:
        dim_red.pca_inplace(b, 2, None, None, None, sr.FeatureSelection.None_, None)
    assert e.value.code == _ffi.E_SHAPE


def test_fused_pipeline_equals_separate_calls(ctx):
    """srx_pipeline == normalize_total_inplace + log1p_transform_inplace + pca_inplace(HVG)."""
    import singlerust_amd as sr
    from singlerust_amd import _ffi
    from singlerust_amd.memory import processing
    from singlerust_amd.memory.processing import dim_red
    m, _ = synth_host(21, 5000, 4000, 0.04)
    a, b = adata_of(m, ctx, 1), adata_of(m, ctx, 1)
    processing.normalize_total_inplace(a, 1e4, sr.Direction.Row)
    processing.log1p_transform_inplace(a)
    dim_red.pca_inplace(a, 20, None, None, None, sr.FeatureSelection.HighlyVariable(400), None)
    opts = _ffi.PcaOpts(20, -1, -1, -1, 0, 0, 0, 0.0, 0)
    res = _ffi.PipelineResult()
    _ffi.check(_ffi.lib().srx_pipeline(b.x().handle, 1e4, 400, C.byref(opts), C.byref(res)), ctx.handle)
    scores = np.zeros((5000, 20))
    comps = np.zeros((400, 20))
    hv = np.zeros(400, np.uint64)
    _ffi.check(_ffi.lib().srx_result_fetch(b.x().handle, _ffi.ptr(scores), _ffi.ptr(comps), None, None, None, _ffi.ptr(hv)), ctx.handle)
    assert np.array_equal(hv, a.uns["pca"]["selected_features"])
    # the pipeline rounds ln_1p(f64(v) * scale) ONCE to the f32 storage; the two separate calls round the scaled value and
    # the logarithm in turn
    assert np.max(np.abs(a.x_values() - b.x_values()) / np.maximum(np.abs(b.x_values()), 1e-30)) < 1e-6
    # LDS atomics accumulate in a run-dependent order: equal to rounding, not bit-equal
    assert col_err(scores, a.obsm["X_pca"]) < TOL
    assert col_err(comps, a.uns["pca"]["components"]) < TOL
    assert res.pca.k == 400 and res.pca.n_pc == 20 and res.pca.nnz_selected > 0
    # and against the oracle
    lg = oracle.log1p_transform(oracle.normalize_total(m, 1e4, ROW))
    want, wc, *_ = pca_oracle.pca_inplace(lg, 20, None, None, hv)
    assert col_err(scores, want) < TOL and col_err(comps, wc) < TOL


@pytest.mark.parametrize("store", [1, 2])
@pytest.mark.parametrize("n,g,density,k", [(700, 500, 0.1, 100), (5000, 4000, 0.04, 700), (20000, 3000, 0.2, 1500)])
def test_spmm_and_gram_kernels_vs_scipy(ctx, store, n, g, density, k):
    """Kernel level: the CSR x dense-panel forward product, its transpose and the sparse Gram
    X_sel^T X_sel against scipy (long tile segments, several gene tiles, ragged tails)."""
    import scipy.sparse as sp
    from singlerust_amd import _ffi
    rng = np.random.default_rng(n + k)
    m, _ = synth_host(3, n, g, density)
    a = adata_of(m, ctx, store)
    sel = np.sort(rng.choice(g, k, replace=False)).astype(np.uint64)
    P = rng.standard_normal((k, 64))
    y = np.zeros((n, 64))
    t = np.zeros((k, 64))
    gram = np.zeros((k, k))
    lib = _ffi.lib()
    _ffi.check(lib.srx_spmm(a.x().handle, _ffi.ptr(sel), k, _ffi.ptr(P), _ffi.ptr(y), _ffi.ptr(t), _ffi.ptr(gram)),
               ctx.handle)
    A = sp.csr_matrix((m.values.astype(np.float64), m.indices.astype(np.int64), m.indptr.astype(np.int64)),
                      shape=(n, g))[:, sel.astype(np.int64)]
    P_used = P.astype(np.float32).astype(np.float64) if store == 1 else P
    want_y = A @ P_used
    assert np.abs(y - want_y).max() / np.abs(want_y).max() < (2e-6 if store == 1 else 1e-12)
    want_t = A.T @ y                                        # the transpose applied to the y the kernel produced
    assert np.abs(t - want_t).max() / np.abs(want_t).max() < 1e-12      # f64 LDS accumulation
    want_g = (A.T @ A).toarray()
    assert np.array_equal(gram, gram.T)
    assert np.array_equal(gram, want_g)                     # integer counts: exact in f64, any order


@pytest.mark.parametrize("kernel", ["ranges", "rows"])
@pytest.mark.parametrize("k", [64, 512, 513, 1500, 2000, 4096, 4100])
def test_forward_spmm_from_row_major_records(ctx, kernel, k, monkeypatch):
    """The transform's kernel k_spmm_rows (column slices of the panel in LDS) and round 6's k_spmm_ranges (one quad per row, gene
    ranges of 512 through the LDS, chunks of 32 records; built, conflict-free, slower: SRX_FWD_RANGES=1) against scipy, through srx_spmm's row-major
    route: rows of 0, 1, 31 .. 33, 63 .. 65 and 200 kept entries (chunk boundaries), runs of > 32 entries inside ONE gene range
    (the rare second chunk of a run), rows whose entries all fall in the last range, a row count that is not a multiple of 16,
    k on both sides of the range size and of the 8-range limit (4100: the launcher falls back on the column-slice kernel)."""
    import scipy.sparse as sp
    from singlerust_amd import _ffi
    import singlerust_amd as sr
    monkeypatch.setenv("SRX_SPMM_ROWS", "1")
    monkeypatch.setenv("SRX_FWD_RANGES", "1" if kernel == "ranges" else "0")
    rng = np.random.default_rng(k)
    g = k + 37
    lens = [0, 1, 2, 31, 32, 33, 0, 63, 64, 65, 200, 5, 17, 18, 19, 0, 1]
    lens = [min(n, k) for n in lens] * 5 + [min(40, k)] * 2
    rows, cols, vals = [], [], []
    for r, n in enumerate(lens):
        if r % 7 == 3 and k > 100:        # a run inside one range / at the top of the column space
            lo = (k - 90) if r % 2 else (k // 2 // 512) * 512
            c = lo + np.sort(rng.choice(min(90, k - lo), min(n, min(90, k - lo)), replace=False))
        else:
            c = np.sort(rng.choice(k, n, replace=False))
        rows += [r] * len(c)
        cols += list(c)
        vals += list(rng.integers(1, 9, len(c)))
    n_rows = len(lens)
    assert n_rows % 16 != 0
    sel = np.sort(rng.choice(g, k, replace=False)).astype(np.uint64)
    x = sp.csr_matrix((np.array(vals, dtype=np.float64), (rows, sel[np.array(cols, dtype=np.int64)].astype(np.int64))), shape=(n_rows, g))
    x.sort_indices()
    a = sr.IMAnnData.new_basic(x, ctx=ctx, store=1)
    P = rng.standard_normal((k, 64))
    y = np.zeros((n_rows, 64))
    _ffi.check(_ffi.lib().srx_spmm(a.x().handle, _ffi.ptr(sel), k, _ffi.ptr(P), _ffi.ptr(y), None, None), ctx.handle)
    want = x[:, sel.astype(np.int64)] @ P.astype(np.float32).astype(np.float64)
    assert np.abs(y - want).max() <= 2e-6 * np.abs(want).max()
    assert not y[np.array(lens) == 0].any()


@pytest.mark.parametrize("kernel", ["ranges", "rows"])
def test_forward_spmm_row_major_kernels_agree_on_a_bench_shaped_matrix(ctx, kernel, monkeypatch):
    """60k cells of the bench generator, 2000 selected genes: the new kernel's panel product against scipy (several blocks per wave,
    every workgroup of the grid busy, rows in length order)."""
    import scipy.sparse as sp
    from singlerust_amd import _ffi
    monkeypatch.setenv("SRX_SPMM_ROWS", "1")
    monkeypatch.setenv("SRX_FWD_RANGES", "1" if kernel == "ranges" else "0")
    n, g, k = 60_000, 28_000, 2000
    m, _ = synth_host(3, n, g, 0.03)
    a = adata_of(m, ctx, 1)
    rng = np.random.default_rng(6)
    sel = np.sort(rng.choice(g, k, replace=False)).astype(np.uint64)
    P = rng.standard_normal((k, 64))
    y = np.zeros((n, 64))
    _ffi.check(_ffi.lib().srx_spmm(a.x().handle, _ffi.ptr(sel), k, _ffi.ptr(P), _ffi.ptr(y), None, None), ctx.handle)
    A = sp.csr_matrix((m.values.astype(np.float64), m.indices.astype(np.int64), m.indptr.astype(np.int64)), shape=(n, g))[:, sel.astype(np.int64)]
    want = A @ P.astype(np.float32).astype(np.float64)
    assert np.abs(y - want).max() <= 2e-6 * np.abs(want).max()


@pytest.mark.gpu
@pytest.mark.parametrize("values", ["counts", "negative", "wide", "fractions"])
@pytest.mark.parametrize("store", [1, 2])
def test_gram_record_piece_and_pair_boundaries(ctx, store, values):
    """The sparse Gram kernel works on RECORDS — one kept entry x at most 64 entries of its row's suffix, a longer suffix cut into
    several — and, for f32 entries, serves TWO records per load instruction (lanes 0-31 / 32-63, two consecutive entries per lane).
    Rows of 1, 2, 3, 31 .. 33, 63 .. 66, 127 .. 130 and 200 kept entries put suffixes on every side of those boundaries (odd and even
    lengths, pieces of exactly 64, a last piece of 1), the last rows end at the end of the entry array, empty rows sit between them;
    integer values: X^T X must be exact.
    `values` picks the accumulation mode of the f32 kernel: small counts run in FIXED POINT (64-bit integer LDS atomics on products
    scaled by a power of two: non-negative values whose exponents lie within 6 of the largest's); a negative value, or values 1 and
    3000 side by side, send the launch to the f64 atomics (integers: still exact); "fractions" are non-integers in [0.75, 9] — the
    fixed-point products must stay within the f32 product rounding of the exact sums."""
    import scipy.sparse as sp
    from singlerust_amd import _ffi
    rng = np.random.default_rng(99)
    k = g = 320
    lens = [1, 2, 3, 0, 31, 32, 33, 63, 64, 65, 66, 0, 127, 128, 129, 130, 200, 0, 5, 64, 1]
    lens = lens * 3
    rows, cols, vals = [], [], []
    for r, n in enumerate(lens):
        c = np.sort(rng.choice(g, n, replace=False))
        rows += [r] * n
        cols += list(c)
        if values == "counts":
            vals += list(rng.integers(1, 7, n))
        elif values == "negative":
            vals += list(rng.integers(1, 7, n) * rng.choice([-1, 1], n))
        elif values == "wide":
            vals += list(rng.choice([1, 3000], n))
        else:
            vals += list(np.float32(rng.uniform(0.75, 9.0, n)))
    x = sp.csr_matrix((np.array(vals, dtype=np.float64), (rows, cols)), shape=(len(lens), g))
    x.sort_indices()
    import singlerust_amd as sr
    a = sr.IMAnnData.new_basic(x, ctx=ctx, store=store)
    sel = np.arange(k).astype(np.uint64)
    P = rng.standard_normal((k, 64))
    y, t, gram = np.zeros((len(lens), 64)), np.zeros((k, 64)), np.zeros((k, k))
    _ffi.check(_ffi.lib().srx_spmm(a.x().handle, _ffi.ptr(sel), k, _ffi.ptr(P), _ffi.ptr(y), _ffi.ptr(t), _ffi.ptr(gram)), ctx.handle)
    want = (x.T @ x).toarray()
    if values == "fractions":
        # f32 products (2^-24 of each) at store 1; at store 2 f64 products rounded to the fixed-point quantum (round 5: values
        # below 16, kq = 39: half a unit = 2^-40 per product); the sums themselves are exact (integers)
        nzp = (x != 0).astype(np.float64)
        count = (nzp.T @ nzp).toarray()
        if store == 1:
            np.testing.assert_allclose(gram, want, rtol=2e-7, atol=0)
        else:
            assert (np.abs(gram - want) <= count * 2.0 ** -40 + 1e-14 * np.abs(want)).all()
    else:
        assert np.array_equal(gram, want)


@pytest.mark.gpu
@pytest.mark.parametrize("store", [1, 2])
def test_fixed_point_gram_sums_do_not_depend_on_the_order_of_the_atomics(ctx, store):
    """Non-negative non-integer values of bounded range: the stripe kernel adds fixed-point products with INTEGER LDS atomics (f32
    entries: products scaled below 2^31; f64 entries, round 5: below 2^47 through the 1.5 x 2^52 constant), so a workgroup's sums
    do not depend on the order its atomics land in.  With one chunk of cells per owner (<= 16 384 cells: every entry of G
    receives exactly one global addition) repeated launches must agree to the last bit — with f64 atomics they differ in the
    last bits from run to run."""
    from singlerust_amd import _ffi
    import singlerust_amd as sr
    import scipy.sparse as sp
    rng = np.random.default_rng(5)
    n, g, k = 12_000, 1500, 900
    x = sp.random(n, g, density=0.08, random_state=11, format="csr", dtype=np.float64,
                  data_rvs=lambda s: np.float32(rng.uniform(0.8, 9.0, s)).astype(np.float64))
    x.sort_indices()
    a = sr.IMAnnData.new_basic(x, ctx=ctx, store=store)
    sel = np.sort(rng.choice(g, k, replace=False)).astype(np.uint64)
    P = rng.standard_normal((k, 64))
    grams = []
    for _ in range(3):
        y, t, gram = np.zeros((n, 64)), np.zeros((k, 64)), np.zeros((k, k))
        _ffi.check(_ffi.lib().srx_spmm(a.x().handle, _ffi.ptr(sel), k, _ffi.ptr(P), _ffi.ptr(y), _ffi.ptr(t), _ffi.ptr(gram)), ctx.handle)
        grams.append(gram)
    mode = C.c_int32(0)
    _ffi.check(_ffi.lib().srx_gram_mode_info(ctx.handle, C.byref(mode)), ctx.handle)
    assert mode.value == 2
    assert np.array_equal(grams[0], grams[1]) and np.array_equal(grams[0], grams[2])
    A = x[:, sel.astype(np.int64)]
    # f64 entries: half a unit = 2^-48 of the largest possible product per product (values < 16: products < 2^8, kq = 39)
    np.testing.assert_allclose(grams[0], (A.T @ A).toarray(), rtol=3e-7 if store == 1 else 1e-13, atol=1e-6 if store == 1 else 1e-9)


def gram_of(ctx, x, store):
    """X^T X of a scipy CSR through srx_spmm's Gram output + the mode the stripe kernel ran in (srx_gram_mode_info)."""
    from singlerust_amd import _ffi
    import singlerust_amd as sr
    n, g = x.shape
    a = sr.IMAnnData.new_basic(x, ctx=ctx, store=store)
    sel = np.arange(g).astype(np.uint64)
    P = np.zeros((g, 64))
    y, t, gram = np.zeros((n, 64)), np.zeros((g, 64)), np.zeros((g, g))
    _ffi.check(_ffi.lib().srx_spmm(a.x().handle, _ffi.ptr(sel), g, _ffi.ptr(P), _ffi.ptr(y), _ffi.ptr(t), _ffi.ptr(gram)), ctx.handle)
    mode = C.c_int32(0)
    _ffi.check(_ffi.lib().srx_gram_mode_info(ctx.handle, C.byref(mode)), ctx.handle)
    return gram, mode.value


@pytest.mark.gpu
@pytest.mark.parametrize("case,want_mode", [("spread6", 2), ("spread7", 1), ("spread6_fraction", 2), ("one_negative", 1),
                                            ("stored_zero", 1), ("huge", 1), ("tiny", 1), ("all_equal", 2), ("spread4", 2),
                                            ("spread5", 2), ("forced_f64_atomics", 1)])
def test_gram_mode_switch_at_its_boundaries(ctx, case, want_mode, monkeypatch):
    """The f32 stripe kernel picks its accumulation mode on the device (gram.inl): FIXED POINT (mode 2) when no value is
    negative and the binary exponents of the non-zero values lie within 6 of the largest's, f64 atomics (mode 1) otherwise.
    Each case sits on one side of one of those tests — an exponent spread of exactly 6 and of exactly 7, a single negative
    value among 40 000, a single STORED zero, a largest exponent outside the +-48 the scale factor is built for — with values
    that are powers of two times small integers, so that X^T X is exact in EITHER mode: the mode must be the one stated and the
    result must equal scipy's to the bit.  f64 storage runs the same cases through the f64 kernel, which takes the same
    decision on the high words of the doubles (round 5) with a spread of at most 4 (round 6: a product at the small end of a
    6-exponent spread would keep only 2^-34 of itself, too close to the f64 path's 1e-9 tolerance): the same mode except at spreads
    5 and 6, and where f64's wider exponent range admits what f32's scale factor does not ("huge" / "tiny": exponents +-50 are
    inside f64's +-200).  SRX_GRAM_F64_ATOMICS=1 forces the f64 atomics at either storage."""
    import scipy.sparse as sp
    rng = np.random.default_rng(17)
    n, g = 3000, 160
    x = sp.random(n, g, density=0.09, random_state=3, format="csr", dtype=np.float64)
    nnz = x.nnz
    lo, hi = {"spread6": (0, 6), "spread7": (-1, 6), "spread6_fraction": (-9, -3), "one_negative": (0, 3), "stored_zero": (0, 3),
              "huge": (50, 52), "tiny": (-52, -50), "all_equal": (2, 2), "spread4": (0, 4), "spread5": (0, 5),
              "forced_f64_atomics": (0, 3)}[case]
    if case == "forced_f64_atomics":
        monkeypatch.setenv("SRX_GRAM_F64_ATOMICS", "1")
    e = rng.integers(lo, hi + 1, nnz)
    e[0], e[-1] = lo, hi                                  # both ends of the range are really there
    x.data = np.ldexp(rng.integers(1, 2, nnz).astype(np.float64), e)          # exact powers of two: the exponent IS e
    # ("spread6_fraction": 2^-9 .. 2^-3 — non-integers whose products are still exact powers of two)
    if case == "one_negative":
        x.data[nnz // 2] = -x.data[nnz // 2]
    if case == "stored_zero":
        x.data[nnz // 3] = 0.0                            # an explicit zero stays in the structure (canonical CSR allows it)
    x.sort_indices()
    want = (x.T @ x).toarray()
    got, mode = gram_of(ctx, x, 1)
    assert mode == want_mode, (case, mode)
    assert np.array_equal(got, want), case
    got64, mode64 = gram_of(ctx, x, 2)
    want_mode64 = 2 if case in ("huge", "tiny") else (1 if case in ("spread5", "spread6", "spread6_fraction") else want_mode)
    assert mode64 == want_mode64, (case, mode64)
    assert np.array_equal(got64, want)


@pytest.mark.gpu
def test_fixed_point_gram_rounding_stays_below_the_f32_product_rounding(ctx):
    """Exponent spread exactly 6 with full 24-bit mantissas (the worst case the fixed-point mode accepts): every product is
    rounded to a multiple of 2^-kq, half a unit = 2^-30 of the largest possible product.  Against the exact f64 sums of the
    f32 values every entry must be within (half a quantum per product) + (the f32 rounding of the products, 2^-24 of each: the
    scaled product is formed by one f32 FMA — what the f64-atomics mode's f32 products carry too)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(23)
    n, g = 6000, 120
    x = sp.random(n, g, density=0.1, random_state=5, format="csr", dtype=np.float64)
    m = np.float32(rng.uniform(1.0, 2.0, x.nnz)).astype(np.float64)
    e = rng.integers(0, 7, x.nnz)
    e[0], e[-1] = 0, 6
    x.data = np.ldexp(m, e)                                # [1, 128): exponents 0 .. 6
    x.sort_indices()
    want = (x.T @ x).toarray()
    got, mode = gram_of(ctx, x, 1)
    assert mode == 2
    count = ((x != 0).astype(np.float64).T @ (x != 0).astype(np.float64)).toarray()
    quantum = 2.0 ** -(29 - 2 * 6)                         # kq = 29 - 2 emax
    assert (np.abs(got - want) <= 0.5 * quantum * count + 2.0 ** -24 * want + 1e-9).all()
    assert np.abs(got - want).max() <= 2.0 ** -23 * np.abs(want).max()


@pytest.mark.gpu
def test_pipeline_device_selection_ties_and_results(ctx):
    """The pipeline selects HighlyVariable(n) on the device (k_gene_var / k_hvg_rank / k_sel_finish): the
    selection and its ORDER must equal the stable descending sort of the oracle (dim_red/mod.rs:135-140)
    — with ties (duplicated genes: equal variances keep the ascending index), empty genes (variance 0)
    and the cut falling inside a tie group — and mean / std / explained variance must agree with the
    host-selection route."""
    import scipy.sparse as sp
    import singlerust_amd as sr
    from singlerust_amd import _ffi
    from singlerust_amd.memory import processing
    from singlerust_amd.memory.processing import dim_red
    rng = np.random.default_rng(5)
    n, g0 = 3000, 300
    base = sp.random(n, g0, density=0.08, random_state=7, data_rvs=lambda s: rng.integers(1, 9, s).astype(np.float32),
                     format="csc", dtype=np.float32)
    # genes: [base | exact copy of base | 40 empty genes]: every variance appears twice -> ties everywhere
    x = sp.hstack([base, base, sp.csc_matrix((n, 40), dtype=np.float32)]).tocsr()
    x.sort_indices()
    m = oracle.Csr(n, x.shape[1], x.indptr, x.indices, x.data)
    n_hvg = 151                                       # odd: the cut splits a tie pair
    lg = oracle.log1p_transform(oracle.normalize_total(m, 1e4, ROW))
    want_sel = oracle.select_hvg(oracle.compute_variance(lg, COLUMN), n_hvg)

    a, b = adata_of(m, ctx, 1), adata_of(m, ctx, 1)
    opts = _ffi.PcaOpts(10, -1, -1, -1, 0, 0, 0, 0.0, 3)
    res = _ffi.PipelineResult()
    _ffi.check(_ffi.lib().srx_pipeline(b.x().handle, 1e4, n_hvg, C.byref(opts), C.byref(res)), ctx.handle)
    k = int(res.pca.k)
    assert k == n_hvg
    scores, comps = np.zeros((n, 10)), np.zeros((k, 10))
    evr, mean, std, hv = np.zeros(10), np.zeros(k), np.zeros(k), np.zeros(k, np.uint64)
    _ffi.check(_ffi.lib().srx_result_fetch(b.x().handle, _ffi.ptr(scores), _ffi.ptr(comps), _ffi.ptr(evr), _ffi.ptr(mean),
                                           _ffi.ptr(std), _ffi.ptr(hv)), ctx.handle)
    assert np.array_equal(hv, want_sel)               # identical set AND order

    # host-selection route on the same data
    processing.normalize_total_inplace(a, 1e4, sr.Direction.Row)
    processing.log1p_transform_inplace(a)
    dim_red.pca_inplace(a, 10, None, None, None, sr.FeatureSelection.HighlyVariable(n_hvg), None, seed=3)
    ref = a.uns["pca"]
    assert np.array_equal(hv, ref["selected_features"])
    # (the pipeline's moments are those of the f64 transform, the three-call route's those of the f32 values X holds)
    np.testing.assert_allclose(mean, ref["mean"], rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(std, ref["std"], rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(evr, ref["explained_variance_ratio"], rtol=1e-5)
    # duplicated genes make C rank-deficient but the leading subspace is well defined: compare projectors
    q1, _ = np.linalg.qr(comps[:, :5]); q2, _ = np.linalg.qr(ref["components"][:, :5])
    assert np.linalg.norm(q1 @ q1.T - q2 @ q2.T) < 1e-4


@pytest.mark.gpu
def test_pipeline_nan_variance_is_an_error(ctx):
    """A NaN value makes a gene variance NaN: the reference's partial_cmp().unwrap() panics
    (dim_red/mod.rs:138); the device-side selection reports it as SRX_E_NAN through the pipeline."""
    import singlerust_amd as sr
    from singlerust_amd import _ffi
    m, _ = synth_host(33, 600, 300, 0.1)
    vals = m.values.copy()
    vals[5] = np.nan
    a = adata_of(m.with_values(vals), ctx, 1)
    opts = _ffi.PcaOpts(5, -1, -1, -1, 0, 0, 0, 0.0, 0)
    res = _ffi.PipelineResult()
    with pytest.raises(sr.SrxError) as e:
        _ffi.check(_ffi.lib().srx_pipeline(a.x().handle, 1e4, 50, C.byref(opts), C.byref(res)), ctx.handle)
    assert e.value.code == _ffi.E_NAN


@pytest.mark.gpu
@pytest.mark.parametrize("store", [1, 2])
def test_fixed_point_moments_range_boundary(ctx, store):
    """The pipeline's gene moments are exact FIXED-POINT sums of y = ln_1p(v * scale) (62 bits for n_rows values below 64): a
    transformed value at or beyond 64 — v * scale >= e^64 - 1 = 6.2e27, which a target_sum that large allows — poisons its gene and
    the selection reports SRX_E_NAN (documented limit; the reference would carry the value).  Just below the
    boundary (ln_1p = 63.8) the moments, the HVG list and the stored values are the oracle's."""
    import singlerust_amd as sr
    from singlerust_amd import _ffi
    from singlerust_amd.memory import statistics as st
    m, _ = synth_host(41, 400, 200, 0.1)
    # cell 7 keeps ONE entry: its value is the whole row sum, so v * scale = target_sum exactly
    ip = m.indptr.astype(np.int64)
    keep = np.ones(len(m.values), bool)
    keep[ip[7] + 1:ip[8]] = False
    cnt = np.diff(ip)
    cnt[7] = 1
    m2 = oracle.Csr(m.n_rows, m.n_cols, np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint64), m.indices[keep], m.values[keep])
    opts = _ffi.PcaOpts(5, -1, -1, -1, 0, 0, 0, 0.0, 0)
    res = _ffi.PipelineResult()
    for target, ok in ((5.0e27, True), (7.0e27, False)):          # ln_1p: 63.78 / 64.11
        a = adata_of(m2, ctx, store)
        if ok:
            _ffi.check(_ffi.lib().srx_pipeline(a.x().handle, target, 40, C.byref(opts), C.byref(res)), ctx.handle)
            lg = oracle.log1p_transform(oracle.normalize_total(m2, target, ROW))
            assert rel_err(a.x_values(np.float64), lg.values.astype(np.float64)) <= (1e-6 if store == 1 else 4e-16)
            want_var = oracle.compute_variance(lg, COLUMN)
            hv = np.zeros(int(res.pca.k), np.uint64)
            _ffi.check(_ffi.lib().srx_result_fetch(a.x().handle, None, None, None, None, None, _ffi.ptr(hv)), ctx.handle)
            assert np.array_equal(hv, oracle.select_hvg(want_var, 40))
            # (values near 60 with variances near 1: E[x^2] - E[x]^2 cancels 3.5 digits — the naive form is the reference's, csr.rs:183)
            assert np.allclose(st.compute_variance(a, sr.Direction.Column), want_var, rtol=1e-2 if store == 1 else 1e-8, atol=1e-12)
        else:
            with pytest.raises(sr.SrxError) as e:
                _ffi.check(_ffi.lib().srx_pipeline(a.x().handle, target, 40, C.byref(opts), C.byref(res)), ctx.handle)
            assert e.value.code == _ffi.E_NAN


@pytest.mark.gpu
@pytest.mark.parametrize("n,g,hvg,npc,store", [
    (60, 40, 12, 3, 2),          # k < 64: the block is narrower than l
    (300, 170, 129, 5, 2),       # k just over one 128-tile
    (1500, 520, 257, 8, 1),      # k just over one 256-tile, f32 storage
    (2111, 700, 700, 6, 2),      # n_hvg == n_vars (every gene selected, incl. empty ones)
    (901, 333, 1000, 4, 1),      # n_hvg > n_vars
])
def test_fused_pipeline_odd_shapes(ctx, n, g, hvg, npc, store):
    """srx_pipeline (device-side selection, fused compaction, Gram solver, graphs) at shapes that sit on the
    tile / block boundaries, against the oracle end to end: selection identical (f64 storage), the spanned
    subspace of the leading components and the scores to 1e-5."""
    from singlerust_amd import _ffi
    m, _ = synth_host(100 + n, n, g, 0.08)
    a = adata_of(m, ctx, store)
    opts = _ffi.PcaOpts(npc, -1, -1, -1, 0, 0, 0, 0.0, 7)
    res = _ffi.PipelineResult()
    _ffi.check(_ffi.lib().srx_pipeline(a.x().handle, 1e4, hvg, C.byref(opts), C.byref(res)), ctx.handle)
    k = int(res.pca.k)
    assert k == min(hvg, g) and int(res.pca.n_pc) == npc
    scores, comps = np.zeros((n, npc)), np.zeros((k, npc))
    evr, hv = np.zeros(npc), np.zeros(k, np.uint64)
    _ffi.check(_ffi.lib().srx_result_fetch(a.x().handle, _ffi.ptr(scores), _ffi.ptr(comps), _ffi.ptr(evr), None, None,
                                           _ffi.ptr(hv)), ctx.handle)
    lg = oracle.log1p_transform(oracle.normalize_total(m, 1e4, ROW))
    want_sel = pca_oracle.select_features_hvg(lg, hvg)
    assert np.array_equal(hv, want_sel)               # srx_pipeline: f64 moments at either storage
    # zero-variance selected genes (n_hvg >= n_vars picks the empty ones): the reference divides by std = 0 there;
    # compare on the columns with a positive std, as test_defaults_and_small_k does
    dense = oracle.densify_selected(lg, hv)
    live = dense.std(axis=0) > 0
    want_scores, want_comps, want_evr, *_ = pca_oracle.pca_inplace(lg, npc, None, None, hv[live])
    tol = 1e-5 if store == 1 else 1e-7
    assert col_err(scores, want_scores) < tol
    assert col_err(comps[live], want_comps) < tol
    assert np.all(comps[~live] == 0)
    np.testing.assert_allclose(evr * k / live.sum(), want_evr, rtol=1e-5)      # trace counts the dead columns as 0


@pytest.mark.gpu
@pytest.mark.parametrize("store", [1, 2])
def test_fused_pipeline_long_empty_and_last_rows(ctx, store):
    """The compaction's count pass walks a row in batches of 1024 entries with the next batch's loads in flight, leaves the rows
    whose batches could run past the array's end to a plain loop, and packs eight row counts into one store: rows of exactly
    1024 / 1025 / 2048 entries, rows of several batches, empty rows (also as a wave's first, a group's last and the matrix's
    last row), a long LAST row — and enough rows that both the streamed and the plain route are taken (the streamed one needs
    1024 entries of array behind a row).  The pipeline's selection, components and scores against the oracle."""
    from singlerust_amd import _ffi
    rng = np.random.default_rng(5150 + store)
    n, g, hvg, npc = 4200, 6000, 900, 6
    lens = rng.integers(20, 400, size=n)
    for r, L in [(0, 0), (7, 0), (8, 1024), (9, 1025), (15, 2048), (16, 3500), (17, 0), (100, 5999), (101, 1), (1023, 2049),
                 (2048, 1023), (n - 9, 0), (n - 8, 4100), (n - 3, 0), (n - 2, 1024), (n - 1, 3000)]:
        lens[r] = L
    lens[200:260] = 0                              # whole groups of eight without an entry
    ip = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    idx = np.concatenate([np.sort(rng.choice(g, size=int(L), replace=False)) for L in lens]).astype(np.uint64)
    # planted structure on top of noise, so that the leading components are well separated
    z = rng.normal(size=(n, 3))
    load = rng.normal(size=(3, g))
    rows = np.repeat(np.arange(n), lens)
    val = np.exp(0.6 * np.einsum("ij,ij->i", z[rows], load[:, idx.astype(np.int64)].T) + 0.3 * rng.normal(size=idx.size)).astype(np.float32)
    val = np.maximum(np.round(val * 3), 1).astype(np.float32)
    m = oracle.Csr(n, g, ip, idx, val)
    a = adata_of(m, ctx, store)
    opts = _ffi.PcaOpts(npc, -1, -1, -1, 0, 0, 0, 0.0, 7)
    res = _ffi.PipelineResult()
    _ffi.check(_ffi.lib().srx_pipeline(a.x().handle, 1e4, hvg, C.byref(opts), C.byref(res)), ctx.handle)
    scores, comps = np.zeros((n, npc)), np.zeros((hvg, npc))
    evr, hv = np.zeros(npc), np.zeros(hvg, np.uint64)
    _ffi.check(_ffi.lib().srx_result_fetch(a.x().handle, _ffi.ptr(scores), _ffi.ptr(comps), _ffi.ptr(evr), None, None,
                                           _ffi.ptr(hv)), ctx.handle)
    lg = oracle.log1p_transform(oracle.normalize_total(m, 1e4, ROW))
    assert np.array_equal(hv, pca_oracle.select_features_hvg(lg, hvg))
    assert int(res.pca.nnz_selected) == int(np.isin(idx, hv).sum())          # every kept entry, counted once
    want_scores, want_comps, want_evr, *_ = pca_oracle.pca_inplace(lg, npc, None, None, hv)
    lead = 3                                       # the planted directions; the noise components behind them are close together
    tol = 1e-5 if store == 1 else 1e-7
    assert col_err(scores[:, :lead], want_scores[:, :lead]) < tol
    assert col_err(comps[:, :lead], want_comps[:, :lead]) < tol
    q1, _ = np.linalg.qr(comps); q2, _ = np.linalg.qr(want_comps)
    assert np.linalg.norm(q1 @ q1.T - q2 @ q2.T) < 1e-3
    np.testing.assert_allclose(evr[:lead], want_evr[:lead], rtol=1e-5)


@pytest.mark.gpu
def test_pipeline_medium_scale_vs_oracle(ctx):
    """20k cells x 6000 genes, HVG(1000), 30 PCs through the fused pipeline (device selection, 8 gene tiles of
    128 -> 36 tile pairs, Chebyshev rounds, graphs) against the oracle's densify + exact SVD: scores and
    components to 1e-5 (f32 storage), explained variance ratio to 1e-5, selection differing at most at
    near-ties of the cut."""
    from singlerust_amd import _ffi
    n, g, hvg, npc = 20000, 6000, 1000, 30
    m, _ = synth_host(77, n, g, 0.05)
    a = adata_of(m, ctx, 1)
    opts = _ffi.PcaOpts(npc, -1, -1, -1, 0, 0, 0, 0.0, 11)
    res = _ffi.PipelineResult()
    _ffi.check(_ffi.lib().srx_pipeline(a.x().handle, 1e4, hvg, C.byref(opts), C.byref(res)), ctx.handle)
    assert res.pca.residual <= 1e-7
    scores, comps = np.zeros((n, npc)), np.zeros((hvg, npc))
    evr, hv = np.zeros(npc), np.zeros(hvg, np.uint64)
    _ffi.check(_ffi.lib().srx_result_fetch(a.x().handle, _ffi.ptr(scores), _ffi.ptr(comps), _ffi.ptr(evr), None, None,
                                           _ffi.ptr(hv)), ctx.handle)
    lg = oracle.log1p_transform(oracle.normalize_total(m, 1e4, ROW))
    want_sel = pca_oracle.select_features_hvg(lg, hvg)
    assert np.array_equal(hv, want_sel)
    want_scores, want_comps, want_evr, *_ = pca_oracle.pca_inplace(lg, npc, None, None, hv)
    # components inside a cluster of close eigenvalues may rotate among themselves: compare the well separated ones
    # column by column and everything through the projector
    gaps = np.abs(np.diff(want_evr)) / want_evr[:-1]
    sep = np.concatenate([[True], gaps > 1e-3]) & np.concatenate([gaps > 1e-3, [True]])     # gap on both sides
    assert sep.sum() >= 10
    assert col_err(scores[:, sep], want_scores[:, sep]) < TOL
    assert col_err(comps[:, sep], want_comps[:, sep]) < TOL
    q1, _ = np.linalg.qr(comps); q2, _ = np.linalg.qr(want_comps)
    assert np.linalg.norm(q1 @ q1.T - q2 @ q2.T) < 1e-4
    np.testing.assert_allclose(evr, want_evr, rtol=1e-5)


@pytest.mark.parametrize("store,tol", [(1, TOL), (2, 1e-7)])
@pytest.mark.parametrize("k,n_pc", [(400, 100), (70, 70), (130, 57), (300, 150)])
def test_more_components_than_one_block_resolves(ctx, store, tol, k, n_pc):
    """n_components beyond the 56 a 64-column block resolves (the reference takes any n <= k, dim_red/mod.rs:52):
    deflation rounds on the explicit C.  Scores, components and explained variance against the exact-SVD oracle;
    components of a nearly degenerate pair are compared as a subspace."""
    import singlerust_amd as sr
    from singlerust_amd.memory import processing
    from singlerust_amd.memory.processing import dim_red
    m, _ = synth_host(77 + k, 6000, 900, 0.08)
    a = adata_of(m, ctx, store)
    processing.normalize_total_inplace(a, 1e4, sr.Direction.Row)
    processing.log1p_transform_inplace(a)
    info = dim_red.pca_inplace(a, n_pc, None, None, None, sr.FeatureSelection.HighlyVariable(k), None)
    assert info.n_pc == n_pc and info.k == k
    got, comps = a.obsm["X_pca"], a.uns["pca"]["components"]
    assert got.shape == (6000, n_pc) and comps.shape == (k, n_pc)
    lg = oracle.log1p_transform(oracle.normalize_total(m, 1e4, ROW))
    want, wc, wevr, *_ = pca_oracle.pca_inplace(lg, n_pc, None, None, a.uns["pca"]["selected_features"])
    evr = a.uns["pca"]["explained_variance_ratio"]
    assert np.allclose(evr, wevr, rtol=1e-6 if store == 2 else 1e-5)
    assert np.all(evr[:-1] >= evr[1:] * (1 - 1e-9))                 # descending across the rounds
    assert np.abs(comps.T @ comps - np.eye(n_pc)).max() < 1e-6      # orthonormal across the rounds
    # per-component comparison where the eigenvalue is isolated, subspace comparison otherwise
    gaps = np.minimum(np.abs(np.diff(wevr, prepend=np.inf)), np.abs(np.diff(wevr, append=0.0))) / wevr
    iso = gaps > 1e-3
    assert iso.sum() > n_pc // 2
    assert col_err(got[:, iso], want[:, iso]) < 20 * tol and col_err(comps[:, iso], wc[:, iso]) < 20 * tol
    proj = wc @ (wc.T @ comps)                                       # the oracle's subspace contains ours
    assert np.abs(proj - comps).max() < 1e-4


@pytest.mark.parametrize("n,g,density,n_pc", [(4000, 800, 0.1, 50), (3000, 300, 0.15, 30), (5000, 1200, 0.05, 56)])
def test_flat_spectrum_does_not_break_the_block(ctx, n, g, density, n_pc):
    """Structureless data: the eigenvalues form a nearly flat bulk (theta_64 / theta_50 -> 1), the case where the block
    needs Chebyshev filters of high total degree.  A degree-12 filter with the leading eigenvalues still in the operator
    used to collapse the guard columns ("block lost rank").  The degree is now bounded by the block's spectral spread
    and rounds of <= 16 components take over when one round stalls.  Eigenvalues against the exact SVD; the vectors of
    a nearly degenerate bulk are not individually determined, so they are checked through their own residuals."""
    import scipy.sparse as sp
    import singlerust_amd as sr
    from singlerust_amd.memory.processing import dim_red
    rng = np.random.default_rng(n + g)
    x = sp.random(n, g, density=density, random_state=rng.integers(1 << 30), format="csr",
                  data_rvs=lambda s: rng.integers(1, 6, s).astype(np.float64), dtype=np.float64)
    x.sort_indices()
    keep = np.flatnonzero(np.asarray((x != 0).sum(axis=0)).ravel() > 1)
    x = x[:, keep].tocsr()
    x.sort_indices()
    m = oracle.Csr(n, x.shape[1], x.indptr, x.indices, x.data)
    a = adata_of(m, ctx, 2)
    info = dim_red.pca_inplace(a, n_pc, None, None, None, sr.FeatureSelection.None_, None)
    assert info.n_pc == n_pc and info.residual <= 1e-9
    dense = x.toarray()
    z = (dense - dense.mean(axis=0)) / dense.std(axis=0)
    s = np.linalg.svd(z, compute_uv=False)
    want_evr = (s * s)[:n_pc] / (s * s).sum()
    assert np.allclose(a.uns["pca"]["explained_variance_ratio"], want_evr, rtol=1e-8)
    v = a.uns["pca"]["components"]
    assert np.abs(v.T @ v - np.eye(n_pc)).max() < 1e-8
    cv = z.T @ (z @ v)
    theta = (s * s)[:n_pc]
    res = np.linalg.norm(cv - v * theta, axis=0) / theta
    assert res.max() < 1e-7
    assert col_err(a.obsm["X_pca"], z @ v) < 1e-9                  # scores belong to the returned vectors


def test_dominant_structure_with_flat_tail(ctx):
    """The configuration that ended in "block lost rank" before the degree bound and the safe plan: 20k cells x 20k genes
    of the bench generator, 1000 HVGs, 50 components — a few strong components (theta_1 / theta_64 ~ 30) over a flat
    bulk (theta_64 / theta_50 = 0.97-0.99).  The run has to converge, with the exact SVD's eigenvalues, vectors that
    satisfy their own eigen-equation, and scores that belong to them."""
    from singlerust_amd import _ffi
    m, _ = synth_host(2002, 20000, 20000, 0.05)
    a = adata_of(m, ctx, 1)
    opts = _ffi.PcaOpts(50, -1, -1, -1, 0, 0, 0, 0.0, 0)
    res = _ffi.PipelineResult()
    _ffi.check(_ffi.lib().srx_pipeline(a.x().handle, 1e4, 1000, C.byref(opts), C.byref(res)), ctx.handle)
    assert res.pca.residual <= 1e-7 and res.pca.n_pc == 50
    scores, comps = np.zeros((20000, 50)), np.zeros((1000, 50))
    evr, hv = np.zeros(50), np.zeros(1000, np.uint64)
    _ffi.check(_ffi.lib().srx_result_fetch(a.x().handle, _ffi.ptr(scores), _ffi.ptr(comps), _ffi.ptr(evr), None, None,
                                           _ffi.ptr(hv)), ctx.handle)
    lg = oracle.log1p_transform(oracle.normalize_total(m, 1e4, ROW))
    dense = oracle.densify_selected(lg, hv)
    z = (dense - dense.mean(axis=0)) / dense.std(axis=0)
    s = np.linalg.svd(z, compute_uv=False)
    theta = (s * s)[:50]
    assert theta[0] / (s * s)[63] > 10 and (s * s)[63] / theta[49] > 0.9          # the spectrum this test is about
    assert np.allclose(evr, theta / (s * s).sum(), rtol=1e-5)
    assert np.abs(comps.T @ comps - np.eye(50)).max() < 1e-6
    cv = z.T @ (z @ comps)
    assert (np.linalg.norm(cv - comps * theta, axis=0) / theta).max() < 1e-5
    assert col_err(scores, z @ comps) < TOL


@pytest.mark.parametrize("n,k,n_pc", [(60, 300, 50), (40, 200, 39), (64, 500, 10), (65, 500, 56)])
def test_fewer_cells_than_the_block_is_wide(ctx, n, k, n_pc):
    """N - 1 < 64 <= k: the centred matrix has rank N - 1, the 64-column block must be narrowed to it (it used to lose
    rank in the first CholeskyQR); asking for more components than the rank is refused."""
    import singlerust_amd as sr
    from singlerust_amd import _ffi
    from singlerust_amd.memory import processing
    from singlerust_amd.memory.processing import dim_red
    m, _ = synth_host(900 + n, n, 1500, 0.2)
    a = adata_of(m, ctx, 2)
    processing.normalize_total_inplace(a, 1e4, sr.Direction.Row)
    processing.log1p_transform_inplace(a)
    info = dim_red.pca_inplace(a, n_pc, None, None, None, sr.FeatureSelection.HighlyVariable(k), None)
    lg = oracle.log1p_transform(oracle.normalize_total(m, 1e4, ROW))
    sel = a.uns["pca"]["selected_features"]
    dense = oracle.densify_selected(lg, sel)
    live = dense.std(axis=0) > 0                                   # zero-variance columns: std treated as 1, nothing to compare
    want, wc, wevr, *_ = pca_oracle.pca_inplace(lg, n_pc, None, None, sel[live])
    assert info.n_pc == n_pc
    assert col_err(a.obsm["X_pca"], want) < 1e-7 and col_err(a.uns["pca"]["components"][live], wc) < 1e-7
    with pytest.raises(_ffi.SrxError) as e:
        dim_red.pca_inplace(a, n, None, None, None, sr.FeatureSelection.HighlyVariable(k), None)     # rank is N - 1
    assert e.value.code == _ffi.E_ARG


def test_tiny_exact_rank_problem_is_reliable(ctx):
    """6 cells (one of them empty) x 40 genes, 10 HVGs, 3 components: the block is as wide as the rank (5) and its
    spectrum spans ~1e3, so a sweep of three applications squares to a Gram matrix beyond what one CholeskyQR takes —
    the run used to fail or succeed depending on the summation order of the atomics.  The last-resort mode (shifted
    CholeskyQR3 after every application) makes it reliable: ten runs, ten results equal to the exact SVD."""
    import scipy.sparse as sp
    from singlerust_amd import _ffi
    rng = np.random.default_rng(3)
    x = sp.random(6, 40, density=0.4, random_state=2, data_rvs=lambda s: rng.integers(1, 9, s).astype(np.float64), dtype=np.float64,
                  format="csr")
    x = sp.vstack([x[:1], sp.csr_matrix((1, 40)), x[2:]]).tocsr()
    x.sort_indices()
    m = oracle.Csr(6, 40, x.indptr, x.indices, x.data)
    lg = oracle.log1p_transform(oracle.normalize_total(m, 1e4, ROW))
    for _ in range(10):
        a = adata_of(m, ctx, 2)
        opts = _ffi.PcaOpts(3, -1, -1, -1, 0, 0, 0, 0.0, 5)
        res = _ffi.PipelineResult()
        _ffi.check(_ffi.lib().srx_pipeline(a.x().handle, 1e4, 10, C.byref(opts), C.byref(res)), ctx.handle)
        scores, comps, hv = np.zeros((6, 3)), np.zeros((10, 3)), np.zeros(10, np.uint64)
        _ffi.check(_ffi.lib().srx_result_fetch(a.x().handle, _ffi.ptr(scores), _ffi.ptr(comps), None, None, None, _ffi.ptr(hv)),
                   ctx.handle)
        dense = oracle.densify_selected(lg, hv)
        live = dense.std(axis=0) > 0
        want, wc, *_ = pca_oracle.pca_inplace(lg, 3, None, None, hv[live])
        assert col_err(scores, want) < 1e-7 and col_err(comps[live], wc) < 1e-7


def test_robust_mode_gives_the_same_answers(ctx, monkeypatch):
    """SRX_PCA_ROBUST=1 runs the last-resort mode from the start (CholeskyQR3 after every application of C, plain sweeps):
    the planted golden case and a deflation-round case against the oracle."""
    import singlerust_amd as sr
    from singlerust_amd.memory import processing
    from singlerust_amd.memory.processing import dim_red
    monkeypatch.setenv("SRX_PCA_ROBUST", "1")
    m, _ = synth_host(77 + 130, 6000, 900, 0.08)
    a = adata_of(m, ctx, 2)
    processing.normalize_total_inplace(a, 1e4, sr.Direction.Row)
    processing.log1p_transform_inplace(a)
    dim_red.pca_inplace(a, 57, None, None, None, sr.FeatureSelection.HighlyVariable(130), None)
    lg = oracle.log1p_transform(oracle.normalize_total(m, 1e4, ROW))
    want, wc, wevr, *_ = pca_oracle.pca_inplace(lg, 57, None, None, a.uns["pca"]["selected_features"])
    assert np.allclose(a.uns["pca"]["explained_variance_ratio"], wevr, rtol=1e-6)
    comps = a.uns["pca"]["components"]
    assert np.abs(wc @ (wc.T @ comps) - comps).max() < 1e-4 and col_err(a.obsm["X_pca"][:, :10], want[:, :10]) < 1e-6


@pytest.mark.parametrize("n,g,density,n_pc,center,scale", [(64, 64, 0.02, 63, True, False), (33, 17, 0.05, 17, True, True),
                                                          (64, 2500, 0.02, 64, False, True), (9, 3, 0.5, 3, True, True)])
def test_small_k_goes_straight_to_the_eigensolver(ctx, n, g, density, n_pc, center, scale):
    """k <= 64: the k x k matrix itself is diagonalised (the block is the identity) — exact whatever the rank, so nearly
    empty matrices whose rank is far below min(k, N - 1) and n_components up to k (the reference's limit) are served:
    eigenvalues of the exact SVD, vectors that satisfy the eigen-equation, scores that belong to them."""
    import scipy.sparse as sp
    from singlerust_amd import _ffi
    rng = np.random.default_rng(n * 1000 + g)
    x = sp.random(n, g, density=density, random_state=int(rng.integers(1 << 30)), format="csr",
                  data_rvs=lambda s: rng.integers(1, 30, s).astype(np.float64), dtype=np.float64)
    x.sort_indices()
    m = oracle.Csr(n, g, x.indptr, x.indices, x.data)
    a = adata_of(m, ctx, 2)
    opts = _ffi.PcaOpts(n_pc, int(center), int(scale), -1, 0, 0, 0, 0.0, 1)
    res = _ffi.PipelineResult()
    _ffi.check(_ffi.lib().srx_pipeline(a.x().handle, 1e4, 64, C.byref(opts), C.byref(res)), ctx.handle)
    k = int(res.pca.k)
    assert k == min(64, g) and res.pca.n_pc == n_pc
    scores, comps, evr, hv = np.zeros((n, n_pc)), np.zeros((k, n_pc)), np.zeros(n_pc), np.zeros(k, np.uint64)
    _ffi.check(_ffi.lib().srx_result_fetch(a.x().handle, _ffi.ptr(scores), _ffi.ptr(comps), _ffi.ptr(evr), None, None,
                                           _ffi.ptr(hv)), ctx.handle)
    lg = oracle.log1p_transform(oracle.normalize_total(m, 1e4, ROW))
    dense = oracle.densify_selected(lg, hv)
    sd = dense.std(axis=0)
    z = dense - (dense.mean(axis=0) if center else 0.0)
    if scale:
        z = z / np.where(sd > 0, sd, 1.0)
    s = np.linalg.svd(z, compute_uv=False)
    s2 = np.zeros(k)
    s2[:len(s)] = s * s
    th = s2[:n_pc]
    assert np.allclose(evr, th / s2.sum(), rtol=1e-9, atol=1e-14)
    assert np.abs(comps.T @ comps - np.eye(n_pc)).max() < 1e-10
    assert np.abs(z.T @ (z @ comps) - comps * th).max() < 1e-9 * max(th[0], 1.0)
    assert np.abs(scores - z @ comps).max() < 1e-9 * max(np.abs(scores).max(), 1.0)


@pytest.mark.parametrize("seed", [3, 11, 29])
def test_block_wider_than_numerical_rank(ctx, seed):
    """Five cells, 130 genes, ~13 counts, HVG(65) without centring or scaling: k = 65 > 64 takes the iterative solver with a
    block as wide as the structural rank bound (5), but most selected columns are empty and the numerical rank can be
    lower.  The last-resort mode drops the dependent columns (fuzz_pca.py seed 83): the pairs of the non-zero eigenvalues
    must be exact, the others come back as (0, zero vector)."""
    import ctypes as C
    import scipy.sparse as sp
    import singlerust_amd as sr
    from singlerust_amd import _ffi as F
    rng = np.random.default_rng(seed)
    x = sp.random(5, 130, density=0.02, random_state=seed, format="csr",
                  data_rvs=lambda s: rng.integers(1, 30, s).astype(np.float64), dtype=np.float64)
    x.sort_indices()
    a = sr.IMAnnData.new_basic(x, ctx=ctx, store=2)
    opts = F.PcaOpts(5, 0, 0, -1, 0, 0, 0, 0.0, 1)
    res = F.PipelineResult()
    F.check(F.lib().srx_pipeline(a.x().handle, 1e4, 65, C.byref(opts), C.byref(res)), ctx.handle)
    k = int(res.pca.k)
    scores, comps, evr, hv = np.zeros((5, 5)), np.zeros((k, 5)), np.zeros(5), np.zeros(k, np.uint64)
    F.check(F.lib().srx_result_fetch(a.x().handle, F.ptr(scores), F.ptr(comps), F.ptr(evr), None, None, F.ptr(hv)), ctx.handle)
    m = oracle.Csr(5, 130, x.indptr.astype(np.uint64), x.indices.astype(np.uint64), x.data)
    z = oracle.densify_selected(oracle.log1p_transform(oracle.normalize_total(m, 1e4, 0)), hv)
    s2 = np.linalg.svd(z, compute_uv=False) ** 2
    live = s2 > 1e-9 * s2[0]
    assert np.isfinite(comps).all() and np.isfinite(scores).all()
    assert np.allclose(evr[live], s2[live] / s2.sum(), rtol=1e-7)
    cv = z.T @ (z @ comps)
    assert (np.linalg.norm(cv[:, live] - comps[:, live] * s2[live], axis=0) / s2[live]).max() < 1e-7
    assert np.abs(evr[~live]).max(initial=0.0) < 1e-9 and np.abs(scores - z @ comps).max() < 1e-7 * max(1.0, np.abs(scores).max())


@pytest.mark.gpu
@pytest.mark.parametrize("store", [1, 2])
def test_run_to_run_determinism_of_integer_and_index_outputs(ctx, store):
    """LDS atomics reorder floating-point sums, so the integer outputs (nnz counts, integer row / column sums
    of counts) and everything the feature selection depends on must not depend on the order: two pipeline runs on the
    same input give bit-equal counts, sums, per-gene moments (fixed-point sums), HVG list, mean and std."""
    import singlerust_amd as sr
    from singlerust_amd import _ffi
    from singlerust_amd.memory import statistics as st
    m, _ = synth_host(91, 9000, 5000, 0.05)
    outs = []
    for _ in range(3):
        a = adata_of(m, ctx, store)
        num = (st.compute_number(a, sr.Direction.Row), st.compute_number(a, sr.Direction.Column))
        sums = (st.compute_sum(a, sr.Direction.Row), st.compute_sum(a, sr.Direction.Column))
        opts = _ffi.PcaOpts(10, -1, -1, -1, 0, 0, 0, 0.0, 3)
        res = _ffi.PipelineResult()
        _ffi.check(_ffi.lib().srx_pipeline(a.x().handle, 1e4, 500, C.byref(opts), C.byref(res)), ctx.handle)
        mean, std, hv = np.zeros(500), np.zeros(500), np.zeros(500, np.uint64)
        _ffi.check(_ffi.lib().srx_result_fetch(a.x().handle, None, None, None, _ffi.ptr(mean), _ffi.ptr(std), _ffi.ptr(hv)),
                   ctx.handle)
        outs.append((num[0], num[1], sums[0], sums[1], hv, mean, std, a.x_values()))
    for o in outs[1:]:
        for x, y in zip(outs[0], o):
            assert np.array_equal(x, y)
