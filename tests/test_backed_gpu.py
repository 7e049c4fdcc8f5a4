"""GPU parity tests (-m gpu) of backed (out-of-core) mode: chunked statistics (src/backed/statistics/mod.rs:5-45)
and the two-sweep pipeline over row tiles, against the oracle and against the resident path."""
import ctypes as C

import numpy as np
import pytest

import oracle
from oracle import COLUMN, ROW, backed_oracle, pca_oracle
from test_pca_gpu import TOL, adata_of, col_err, synth_host
from util import create_large_test_data

pytestmark = pytest.mark.gpu


def store_of(m, tmp_path, name="x"):
    from singlerust_amd import backed
    p = str(tmp_path / name)
    backed.BackedCsr.write(p, m.indptr, m.indices, m.values, m.n_cols)
    return p


@pytest.mark.parametrize("dtype", [np.float64, np.float32, np.uint16, np.int32])
@pytest.mark.parametrize("chunk", [1, 37, 300, 1000, 5000])
def test_chunked_number_and_sum_match_oracle(ctx, tmp_path, dtype, chunk):
    """compute_number / compute_sum, ComputationMode::Chunked(size), both directions: bit-exact counts; sums exact for
    integer values, 1e-13 for floats (summation order); equal to ComputationMode::Whole."""
    import singlerust_amd as sr
    from singlerust_amd import backed
    m = create_large_test_data(1000, 100, 10.0, seed=11, dtype=dtype)
    ad = backed.BackedAnnData.open(store_of(m, tmp_path), ctx)
    mode = backed.ComputationMode.Chunked(chunk)
    exact = np.issubdtype(np.dtype(dtype), np.integer)
    for d, od in ((sr.Direction.Row, ROW), (sr.Direction.Column, COLUMN)):
        got_n = backed.statistics.compute_number(ad, d, mode)
        want_n = backed_oracle.number_chunked(m, chunk, od)
        assert got_n.dtype == np.uint32 and np.array_equal(got_n, want_n)
        assert np.array_equal(want_n, oracle.compute_number(m, od))          # chunked == whole
        got_s = backed.statistics.compute_sum(ad, d, mode)
        want_s = backed_oracle.sum_chunked(m, chunk, od) if chunk >= 37 else oracle.compute_sum(m, od)
        if exact:
            assert np.array_equal(got_s, want_s)
        else:
            assert np.allclose(got_s, want_s, rtol=1e-13 if dtype == np.float64 else 1e-6, atol=0)
    whole = backed.ComputationMode.Whole()
    assert np.array_equal(backed.statistics.compute_number(ad, sr.Direction.Column, whole), oracle.compute_number(m, COLUMN))
    assert np.array_equal(backed.statistics.compute_number(ad, sr.Direction.Row, whole), oracle.compute_number(m, ROW))


def test_chunked_edge_cases(ctx, tmp_path):
    """Empty rows at chunk borders, an all-empty chunk, a chunk larger than the matrix."""
    import scipy.sparse as sp
    import singlerust_amd as sr
    from singlerust_amd import backed
    rng = np.random.default_rng(3)
    x = sp.random(600, 50, density=0.1, random_state=5, format="csr", data_rvs=lambda s: rng.integers(1, 9, s).astype(np.float32),
                  dtype=np.float32).tolil()
    x[100:200, :] = 0                        # chunk 100..200 is empty with chunk_size 100
    x = x.tocsr()
    x.eliminate_zeros()
    x.sort_indices()
    m = oracle.Csr(600, 50, x.indptr, x.indices, x.data)
    ad = backed.BackedAnnData.open(store_of(m, tmp_path), ctx)
    for chunk in (100, 7, 10000):
        mode = backed.ComputationMode.Chunked(chunk)
        for d, od in ((sr.Direction.Row, ROW), (sr.Direction.Column, COLUMN)):
            assert np.array_equal(backed.statistics.compute_number(ad, d, mode), oracle.compute_number(m, od))
            assert np.array_equal(backed.statistics.compute_sum(ad, d, mode), oracle.compute_sum(m, od))


def run_resident(m, ctx, n_hvg, n_pc, store, seed=0):
    from singlerust_amd import _ffi
    a = adata_of(m, ctx, store)
    opts = _ffi.PcaOpts(n_pc, -1, -1, -1, 0, 0, 0, 0.0, seed)
    res = _ffi.PipelineResult()
    _ffi.check(_ffi.lib().srx_pipeline(a.x().handle, 1e4, n_hvg, C.byref(opts), C.byref(res)), ctx.handle)
    k = int(res.pca.k)
    scores, comps = np.zeros((m.n_rows, n_pc)), np.zeros((k, n_pc))
    evr, mean, std, hv = np.zeros(n_pc), np.zeros(k), np.zeros(k), np.zeros(k, np.uint64)
    _ffi.check(_ffi.lib().srx_result_fetch(a.x().handle, _ffi.ptr(scores), _ffi.ptr(comps), _ffi.ptr(evr), _ffi.ptr(mean),
                                           _ffi.ptr(std), _ffi.ptr(hv)), ctx.handle)
    return scores, comps, evr, mean, std, hv


@pytest.mark.parametrize("store,chunk", [(1, 700), (1, 5000), (2, 1300), (1, 64)])
def test_backed_pipeline_matches_resident_and_oracle(ctx, tmp_path, store, chunk):
    """normalize_total -> log1p -> HVG(400) -> 20-PC PCA over row tiles == the resident pipeline == the oracle:
    identical HVG list (set and order), scores / components within 1e-5."""
    from singlerust_amd import backed
    m, _ = synth_host(21, 5000, 4000, 0.04)
    if chunk == 64:
        m, _ = synth_host(22, 900, 700, 0.08)
    n_hvg, n_pc = (400, 20) if chunk != 64 else (120, 10)
    ad = backed.BackedAnnData.open(store_of(m, tmp_path), ctx)
    r = backed.processing.pca_pipeline(ad, chunk, 1e4, n_hvg, n_pc, store=store)
    scores, comps, evr, mean, std, hv = run_resident(m, ctx, n_hvg, n_pc, store)
    assert np.array_equal(r.selected, hv)
    assert col_err(r.x_pca, scores) < TOL and col_err(r.components, comps) < TOL
    assert np.allclose(r.explained_variance_ratio, evr, rtol=1e-6)
    assert np.allclose(r.mean, mean, rtol=1e-9, atol=1e-12) and np.allclose(r.std, std, rtol=1e-9, atol=1e-12)
    assert np.array_equal(r.row_sums, oracle.compute_sum(m, ROW))             # integer counts: exact
    assert int(r.info.n_cells_global) == m.n_rows and int(r.info.k) == n_hvg
    lg = oracle.log1p_transform(oracle.normalize_total(m, 1e4, ROW))
    want_sel = oracle.select_hvg(oracle.compute_variance(lg, COLUMN), n_hvg)
    assert np.array_equal(r.selected, want_sel)
    want, wc, *_ = pca_oracle.pca_inplace(lg, n_pc, None, None, r.selected)
    assert col_err(r.x_pca, want) < TOL and col_err(r.components, wc) < TOL


def test_backed_pipeline_explicit_selection_and_errors(ctx, tmp_path):
    """Explicit feature list (host-selection route), FeatureSelection::None on a narrow matrix, and the session's
    ordering errors."""
    from singlerust_amd import _ffi, backed
    m, _ = synth_host(5, 2000, 300, 0.1)
    ad = backed.BackedAnnData.open(store_of(m, tmp_path), ctx)
    rng = np.random.default_rng(1)
    sel = rng.permutation(300)[:90].astype(np.uint64)
    r = backed.processing.pca_pipeline(ad, 333, 1e4, 0, 8, selected=sel)
    r80 = backed.processing.pca_pipeline(ad, 500, 1e4, 0, 80, selected=np.arange(300, dtype=np.uint64))   # two deflation rounds over row tiles
    assert np.array_equal(r.selected, sel)
    lg = oracle.log1p_transform(oracle.normalize_total(m, 1e4, ROW))
    want, wc, *_ = pca_oracle.pca_inplace(lg, 8, None, None, sel)
    assert col_err(r.x_pca, want) < TOL and col_err(r.components, wc) < TOL
    want80, wc80, *_ = pca_oracle.pca_inplace(lg, 80, None, None, np.arange(300))
    assert r80.x_pca.shape == (2000, 80) and np.abs(wc80 @ (wc80.T @ r80.components) - r80.components).max() < 1e-4
    assert col_err(r80.x_pca[:, :20], want80[:, :20]) < TOL
    # FeatureSelection::None
    r2 = backed.processing.pca_pipeline(ad, 1000, 1e4, 0, 5)
    want2, wc2, *_ = pca_oracle.pca_inplace(lg, 5, None, None, np.arange(300))
    assert col_err(r2.x_pca, want2) < TOL and col_err(r2.components, wc2) < TOL
    # ordering errors
    s = backed.BackedSession(ctx, 300)
    chunk = next(ad.x().iter(500))[0]
    with pytest.raises(_ffi.SrxError):
        s.gram_tile(chunk, 1e4, 3)                      # before select
    with pytest.raises(_ffi.SrxError):
        s.select(50)                                    # no tile yet
    s.stats_tile(chunk, 1e4, 3)
    s.select(50, opts=_ffi.PcaOpts(5, -1, -1, -1, 0, 0, 0, 0.0, 0))
    with pytest.raises(_ffi.SrxError):
        s.stats_tile(chunk, 1e4, 3)                     # sweep 1 is over
    with pytest.raises(_ffi.SrxError):
        s.solve()                                       # the Gram sweep saw no rows
    s.close()
    # wrong column count
    s = backed.BackedSession(ctx, 301)
    with pytest.raises(_ffi.SrxError):
        s.stats_tile(chunk)
    s.close()


def test_wide_selection_at_f64_storage_walks_gene_ranges(ctx, tmp_path):
    """6000 selected features at f64 storage: the forward kernel's panel slice of four f64 columns holds 5118 genes, so the
    scores come from two gene ranges, the second accumulating (round 2 refused this selection in backed mode).  Resident
    pipeline and backed session against the oracle's exact SVD."""
    from singlerust_amd import backed
    m, _ = synth_host(31, 1500, 6400, 0.03)
    n_hvg, n_pc = 6000, 6
    scores, comps, evr, mean, std, hv = run_resident(m, ctx, n_hvg, n_pc, 2)
    lg = oracle.log1p_transform(oracle.normalize_total(m, 1e4, ROW))
    want_sel = oracle.select_hvg(oracle.compute_variance(lg, COLUMN), n_hvg)
    assert np.array_equal(hv, want_sel)
    want, wc, wevr, *_ = pca_oracle.pca_inplace(lg, n_pc, None, None, hv)
    assert col_err(scores, want) < 1e-7 and col_err(comps, wc) < 1e-7
    assert np.allclose(evr, wevr, rtol=1e-8)
    ad = backed.BackedAnnData.open(store_of(m, tmp_path), ctx)
    r = backed.processing.pca_pipeline(ad, 400, 1e4, n_hvg, n_pc, store=2)
    assert np.array_equal(r.selected, hv)
    assert col_err(r.x_pca, want) < 1e-7 and col_err(r.components, wc) < 1e-7
