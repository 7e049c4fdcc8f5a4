"""GPU parity tests (-m gpu) of CSC storage (src/shared/statistics/helper/csc.rs, scale_*_csc, PCA from a CSC matrix)
against oracle/csc_oracle.py, plus the device transpose."""
import ctypes as C

import numpy as np
import pytest

import oracle
from oracle import COLUMN, ROW, csc_oracle, pca_oracle
from test_pca_gpu import TOL, col_err, synth_host

pytestmark = pytest.mark.gpu


def random_csc(n, g, density, seed, dtype, empty_rows=(), empty_cols=()):
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    if np.issubdtype(np.dtype(dtype), np.integer):
        rvs = lambda s: rng.integers(1, 9, s).astype(np.float64)
    else:
        rvs = lambda s: rng.uniform(0.0, 50.0, s)
    x = sp.random(n, g, density=density, random_state=seed, format="lil", data_rvs=rvs, dtype=np.float64)
    for r in empty_rows:
        x[r, :] = 0
    for c in empty_cols:
        x[:, c] = 0
    x = x.tocsc()
    x.eliminate_zeros()
    x.sort_indices()
    return csc_oracle.Csc(n, g, x.indptr, x.indices, x.data.astype(dtype))


def adata_csc(m, ctx, store=0):
    import singlerust_amd as sr
    return sr.IMAnnData.new_basic((m.n_rows, m.n_cols, m.col_offsets, m.row_indices, m.values), ctx=ctx, store=store, csc=True)


@pytest.mark.parametrize("dtype", [np.float64, np.float32, np.uint16, np.int32])
def test_csc_statistics_match_oracle(ctx, dtype):
    """compute_number / sum / variance / std_dev / min_max, both directions, on a CSC matrix with empty rows and an
    empty column: counts bit-exact, integer sums exact, Column variance of the empty gene is NaN (csc.rs:164-177 has
    no guard), Row variance of an empty cell is 0 (csc.rs:151)."""
    import singlerust_amd as sr
    from singlerust_amd.memory import statistics as st
    m = random_csc(700, 90, 0.1, 3, dtype, empty_rows=(0, 17, 699), empty_cols=(5,))
    a = adata_csc(m, ctx)
    assert a.x().is_csc() and (a.n_obs(), a.n_vars()) == (700, 90)
    exact = np.issubdtype(np.dtype(dtype), np.integer)
    for d, od in ((sr.Direction.Row, ROW), (sr.Direction.Column, COLUMN)):
        assert np.array_equal(st.compute_number(a, d), csc_oracle.compute_number(m, od))
        got, want = st.compute_sum(a, d), csc_oracle.compute_sum(m, od)
        assert np.array_equal(got, want) if exact else np.allclose(got, want, rtol=1e-13 if dtype == np.float64 else 1e-6, atol=0)
        gv, wv = st.compute_variance(a, d), csc_oracle.compute_variance(m, od)
        assert np.array_equal(np.isnan(gv), np.isnan(wv))
        ok = ~np.isnan(wv)
        assert np.allclose(gv[ok], wv[ok], rtol=1e-9, atol=1e-9)
        gs = st.compute_std_dev(a, d)
        assert np.allclose(gs[ok], np.sqrt(np.maximum(wv[ok], 0)), rtol=1e-6, atol=1e-6)
        gmn, gmx = st.compute_min_max(a, d)
        wmn, wmx = csc_oracle.compute_min_max(m, od)
        assert np.array_equal(gmn, wmn) and np.array_equal(gmx, wmx)
    vcol = st.compute_variance(a, sr.Direction.Column)
    assert np.isnan(vcol[5])
    assert st.compute_variance(a, sr.Direction.Row)[17] == 0.0


def test_csc_equals_csr_of_the_same_matrix(ctx):
    """The same X uploaded as CSR and as CSC: statistics that the reference defines identically for both storages
    agree; normalize_total + log1p leave the same matrix; download returns the CSC arrays."""
    import scipy.sparse as sp
    import singlerust_amd as sr
    from singlerust_amd.memory import processing, statistics as st
    m = random_csc(500, 120, 0.08, 9, np.float32)
    x = sp.csc_matrix((m.values, m.row_indices.astype(np.int64), m.col_offsets.astype(np.int64)), shape=(500, 120))
    a_csc = sr.IMAnnData.new_basic(x, ctx=ctx)                      # scipy CSC -> ArrayData::CscMatrix
    a_csr = sr.IMAnnData.new_basic(x.tocsr(), ctx=ctx)
    assert a_csc.x().is_csc() and not a_csr.x().is_csc()
    for d in (sr.Direction.Row, sr.Direction.Column):
        assert np.array_equal(st.compute_number(a_csc, d), st.compute_number(a_csr, d))
        assert np.allclose(st.compute_sum(a_csc, d), st.compute_sum(a_csr, d), rtol=1e-12)
    for direction in (sr.Direction.Row, sr.Direction.Column):
        b_csc, b_csr = a_csc.deep_clone(), a_csr.deep_clone()
        processing.normalize_total_inplace(b_csc, 1e4, direction)
        processing.normalize_total_inplace(b_csr, 1e4, direction)
        processing.log1p_transform_inplace(b_csc)
        processing.log1p_transform_inplace(b_csr)
        want = csc_oracle.log1p_transform(csc_oracle.normalize_total(m, 1e4, ROW if direction == sr.Direction.Row else COLUMN))
        got = b_csc.x_values(np.float64)
        assert np.allclose(got, want.values, rtol=1e-6, atol=0)
        back = sp.csc_matrix((got, m.row_indices.astype(np.int64), m.col_offsets.astype(np.int64)), shape=(500, 120)).tocsr()
        back.sort_indices()
        assert np.allclose(back.data, b_csr.x_values(np.float64), rtol=1e-6, atol=0)


@pytest.mark.parametrize("store", [1, 2])
def test_transpose_roundtrip(ctx, store):
    """srx_matrix_to_csr / to_csc: the device transpose (histogram, scan, scatter, bitmap counting sort per row) is
    bit-exact against scipy, including empty rows / columns and rows longer than one wave."""
    import scipy.sparse as sp
    import singlerust_amd as sr
    from singlerust_amd import _ffi
    m, _ = synth_host(31, 3000, 5000, 0.06)
    x = sp.csr_matrix((m.values, m.indices.astype(np.int64), m.indptr.astype(np.int64)), shape=(3000, 5000))
    a = sr.IMAnnData.new_basic(x, ctx=ctx, store=store)
    t = a.x().to_csc()
    assert t.is_csc()
    i = t.info()
    assert (i.n_rows, i.n_cols, i.nnz) == (3000, 5000, x.nnz)
    ip = np.zeros(5001, np.uint64)
    ix = np.zeros(x.nnz, np.uint64)
    _ffi.check(_ffi.lib().srx_matrix_download_pattern(t.handle, _ffi.ptr(ip), _ffi.ptr(ix)), ctx.handle)
    want = x.tocsc()
    want.sort_indices()
    assert np.array_equal(ip, want.indptr) and np.array_equal(ix, want.indices)
    assert np.array_equal(t.values(np.float64), want.data.astype(np.float64))
    back = t.to_csr()
    assert not back.is_csc()
    ip2 = np.zeros(3001, np.uint64)
    ix2 = np.zeros(x.nnz, np.uint64)
    _ffi.check(_ffi.lib().srx_matrix_download_pattern(back.handle, _ffi.ptr(ip2), _ffi.ptr(ix2)), ctx.handle)
    assert np.array_equal(ip2, x.indptr) and np.array_equal(ix2, x.indices)
    assert np.array_equal(back.values(np.float64), x.data.astype(np.float64))


def test_csc_pipeline_and_pca_match_oracle(ctx):
    """normalize_total(Row) -> log1p -> HighlyVariable -> pca_inplace on a CSC matrix: the HVG list follows the CSC
    variance (two-pass form), scores / components within 1e-5 of the exact-SVD oracle on the densified CSC matrix."""
    import scipy.sparse as sp
    import singlerust_amd as sr
    from singlerust_amd import _ffi
    from singlerust_amd.memory import processing
    from singlerust_amd.memory.processing import dim_red
    mr, _ = synth_host(41, 4000, 1500, 0.08)
    x = sp.csr_matrix((mr.values, mr.indices.astype(np.int64), mr.indptr.astype(np.int64)), shape=(4000, 1500))
    keep = np.flatnonzero(np.diff(x.tocsc().indptr) > 0)           # an empty gene makes the CSC variance NaN (tested below)
    x = x[:, keep].tocsc()
    x.sort_indices()
    m = csc_oracle.Csc.from_scipy(x)
    n, g = x.shape
    lg = csc_oracle.log1p_transform(csc_oracle.normalize_total(m, 1e4, ROW))
    want_sel = oracle.select_hvg(csc_oracle.compute_variance(lg, COLUMN), 300)
    dense = csc_oracle.densify_selected(lg, want_sel)
    p = pca_oracle.Pca(15, True, True)
    p.fit(dense)
    want_scores = p.transform(dense)

    a = adata_csc(m, ctx, 1)
    processing.normalize_total_inplace(a, 1e4, sr.Direction.Row)
    processing.log1p_transform_inplace(a)
    sel = dim_red.select_features(a, sr.FeatureSelection.HighlyVariable(300))
    assert np.array_equal(sel, want_sel)
    dim_red.pca_inplace(a, 15, None, None, None, sr.FeatureSelection.HighlyVariable(300), None)
    assert a.obsm["X_pca"].shape == (n, 15)
    assert col_err(a.obsm["X_pca"], want_scores) < TOL
    assert col_err(a.uns["pca"]["components"], p.components) < TOL

    # the fused pipeline on a CSC handle
    b = adata_csc(m, ctx, 1)
    opts = _ffi.PcaOpts(15, -1, -1, -1, 0, 0, 0, 0.0, 0)
    res = _ffi.PipelineResult()
    _ffi.check(_ffi.lib().srx_pipeline(b.x().handle, 1e4, 300, C.byref(opts), C.byref(res)), ctx.handle)
    scores = np.zeros((n, 15))
    hv = np.zeros(300, np.uint64)
    _ffi.check(_ffi.lib().srx_result_fetch(b.x().handle, _ffi.ptr(scores), None, None, None, None, _ffi.ptr(hv)), ctx.handle)
    assert np.array_equal(hv, want_sel) and col_err(scores, want_scores) < TOL

    # an empty gene: compute_variance(Column) is NaN on CSC and HighlyVariable panics in the reference (partial_cmp unwrap)
    x2 = sp.hstack([x[:, :50], sp.csc_matrix((n, 1)), x[:, 50:100]]).tocsc()
    c = adata_csc(csc_oracle.Csc.from_scipy(x2), ctx, 1)
    with pytest.raises(_ffi.SrxError) as e:
        dim_red.select_features(c, sr.FeatureSelection.HighlyVariable(10))
    assert e.value.code == _ffi.E_NAN


def test_csc_filters(ctx):
    """filter_cells / filter_genes on a CSC matrix == the same filters on the CSR matrix of the same X."""
    import scipy.sparse as sp
    import singlerust_amd as sr
    from singlerust_amd.memory import processing
    m = random_csc(800, 100, 0.1, 13, np.float32)
    x = sp.csc_matrix((m.values, m.row_indices.astype(np.int64), m.col_offsets.astype(np.int64)), shape=(800, 100))
    a_csc = sr.IMAnnData.new_basic(x, ctx=ctx)
    a_csr = sr.IMAnnData.new_basic(x.tocsr(), ctx=ctx)
    for lo, hi in ((sr.FlexValue.Absolute(5), sr.FlexValue.Absolute(15)), (sr.FlexValue.Relative(0.1), sr.FlexValue.Relative(0.9))):
        f1, f2 = processing.filter_cells(a_csc, lo, hi), processing.filter_cells(a_csr, lo, hi)
        assert f1.x().is_csc() and (f1.n_obs(), f1.n_vars()) == (f2.n_obs(), f2.n_vars()) and f1.obs_names == f2.obs_names
        assert f1.n_obs() < 800
    for lo, hi in ((sr.FlexValue.Absolute(80), sr.FlexValue.Absolute(100)), (sr.FlexValue.Relative(0.1), sr.FlexValue.NoLimit())):
        g1, g2 = processing.filter_genes(a_csc, lo, hi), processing.filter_genes(a_csr, lo, hi)
        assert (g1.n_obs(), g1.n_vars()) == (g2.n_obs(), g2.n_vars()) and g1.var_names == g2.var_names
        assert g1.n_vars() < 100
        back = g1.x().to_csr()
        assert np.array_equal(np.sort(back.values(np.float64)), np.sort(g2.x_values(np.float64)))
