"""GPU parity tests (-m gpu): statistics / normalise / log1p through the C ABI vs the CPU oracle.

Bars (BASELINE.json north_star): integer nnz counts and integer row/column sums BIT-EXACT;
normalised / log1p values within 1e-5 relative (observed ~1e-7 at f32 storage, ~1e-15 at f64).
"""
import json
import os

import numpy as np
import pytest

import oracle
from oracle import COLUMN, ROW
from util import create_large_test_data, rel_err

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
ALL_DTYPES = [np.int8, np.int16, np.int32, np.uint8, np.uint16, np.uint32, np.float32, np.float64]
TOL = 1e-5        # north_star tolerance on normalised values
TOL_F64 = 1e-12   # what f64 device storage actually delivers


def adata_of(m, ctx, store=0):
    import singlerust_amd as sr
    return sr.IMAnnData.new_basic((m.n_rows, m.n_cols, m.indptr, m.indices, m.values), ctx=ctx, store=store)


@pytest.mark.parametrize("dtype", ALL_DTYPES)
def test_kat_4x5(ctx, dtype):
    import singlerust_amd as sr
    from singlerust_amd.memory import processing, statistics
    D = sr.Direction
    k = json.load(open(os.path.join(GOLD, "kat_4x5.json")))
    m = oracle.Csr(4, 5, k["indptr"], k["indices"], np.array(k["data"], dtype=dtype))
    a = adata_of(m, ctx)
    assert statistics.compute_number(a, D.Row).tolist() == k["number_row"]
    assert statistics.compute_number(a, D.Column).tolist() == k["number_col"]
    assert statistics.compute_sum(a, D.Row).tolist() == k["sum_row"]
    assert statistics.compute_sum(a, D.Column).tolist() == k["sum_col"]
    assert statistics.compute_variance(a, D.Column).tolist() == k["var_col"]
    vr = statistics.compute_variance(a, D.Row)
    assert np.isnan(vr[2]) and vr[[0, 1, 3]].tolist() == [1.0, 0.0, 2.0]
    mn, mx = statistics.compute_min_max(a, D.Column)
    assert mn.tolist() == [1, 1, 4, 1, np.inf] and mx.tolist() == [2, 3, 4, 1, -np.inf]
    processing.normalize_total_inplace(a, 1e4, D.Row)
    assert a.x_dtype() == np.float64                                   # scale/mod.rs:82
    want = np.array([float.fromhex(h) for h in k["normalized_hex"]])
    assert rel_err(a.x_values(), want) < 1e-7
    processing.log1p_transform_inplace(a)
    want = np.array([float.fromhex(h) for h in k["log1p_hex"]])
    assert rel_err(a.x_values(), want) < 2e-7
    from singlerust_amd.memory.processing import dim_red
    assert dim_red.select_features(a, sr.FeatureSelection.HighlyVariable(2)).tolist() == k["hvg2"]
    assert dim_red.select_features(a, sr.FeatureSelection.HighlyVariable(5)).tolist() == k["hvg5"]


@pytest.mark.parametrize("dtype", ALL_DTYPES)
def test_statistics_match_oracle_all_dtypes(ctx, dtype):
    import singlerust_amd as sr
    from singlerust_amd.memory import statistics
    D = sr.Direction
    m = create_large_test_data(700, 300, 12.0, seed=5, dtype=dtype)
    a = adata_of(m, ctx)
    assert np.array_equal(statistics.compute_number(a, D.Row), oracle.compute_number(m, ROW))
    assert np.array_equal(statistics.compute_number(a, D.Column), oracle.compute_number(m, COLUMN))
    exact = np.issubdtype(np.dtype(dtype), np.integer)
    for d, od in ((D.Row, ROW), (D.Column, COLUMN)):
        got, want = statistics.compute_sum(a, d), oracle.compute_sum(m, od)
        if exact:
            assert np.array_equal(got, want)                           # integer sums: bit-exact
        else:
            assert rel_err(got, want) < 1e-6
    gv, wv = statistics.compute_variance(a, D.Column), oracle.compute_variance(m, COLUMN)
    np.testing.assert_allclose(gv, wv, rtol=1e-6, atol=1e-6 * np.max(np.abs(wv)))
    gs, ws = statistics.compute_std_dev(a, D.Column), oracle.compute_std_dev(m, COLUMN)
    np.testing.assert_allclose(gs, ws, rtol=1e-4, atol=1e-3 * np.max(np.abs(ws)))
    gr, wr = statistics.compute_variance(a, D.Row), oracle.compute_variance(m, ROW)
    ok = ~np.isnan(wr)
    assert np.array_equal(np.isnan(gr), np.isnan(wr))                  # NaN exactly where the reference gives NaN
    np.testing.assert_allclose(gr[ok], wr[ok], rtol=1e-6, atol=1e-6 * np.max(np.abs(wr[ok])))
    for d, od in ((D.Row, ROW), (D.Column, COLUMN)):
        (gmn, gmx), (wmn, wmx) = statistics.compute_min_max(a, d), oracle.compute_min_max(m, od)
        assert np.array_equal(gmn, wmn) and np.array_equal(gmx, wmx)   # values are exact in storage


@pytest.mark.parametrize("store,tol", [(1, TOL), (2, TOL_F64)])
def test_reference_property_normalize_total(ctx, store, tol):
    """src/memory/processing/mod.rs:419-481 run against the GPU path: 1000 x 100, sparsity 10,
    target 1e4; row sums (Row) / column sums (Column) == target.  At f64 storage the reference's
    own ±1e-6 absolute bar holds; f32 storage meets the north_star 1e-5 relative bar."""
    import singlerust_amd as sr
    from singlerust_amd.memory import processing
    D = sr.Direction
    m = create_large_test_data(1000, 100, 10.0, seed=123)
    a = adata_of(m, ctx, store)
    n = processing.normalize_total(a, 1e4, D.Row)                      # copying form, :314-322
    assert n.x_dtype() == np.float64 and a.x_dtype() == np.float64
    assert rel_err(a.x_values(), m.values) < 1e-6                      # the source is untouched
    got = oracle.Csr(m.n_rows, m.n_cols, m.indptr, m.indices, n.x_values())
    rows = oracle.compute_sum(got, ROW)
    nonempty = oracle.compute_number(m, ROW) > 0
    bar = 1e-6 if store == 2 else 1e4 * 1e-5
    assert np.all(np.abs(rows[nonempty] - 1e4) < bar)
    assert np.all(rows[~nonempty] == 0.0)
    assert rel_err(n.x_values(), oracle.normalize_total(m, 1e4, ROW).values) < tol
    c = processing.normalize_total(a, 1e4, D.Column)
    gotc = oracle.Csr(m.n_rows, m.n_cols, m.indptr, m.indices, c.x_values())
    assert np.all(np.abs(oracle.compute_sum(gotc, COLUMN) - 1e4) < bar)
    assert rel_err(c.x_values(), oracle.normalize_total(m, 1e4, COLUMN).values) < tol


@pytest.mark.parametrize("name", ["ref_shape_1000x100", "counts_64x40", "planted_600x240"])
@pytest.mark.parametrize("store,tol", [(1, TOL), (2, TOL_F64)])
def test_against_golden(ctx, name, store, tol):
    import singlerust_amd as sr
    from singlerust_amd.memory import processing, statistics
    from singlerust_amd.memory.processing import dim_red
    D = sr.Direction
    z = np.load(os.path.join(GOLD, name + ".npz"))
    m = oracle.Csr(int(z["n_rows"]), int(z["n_cols"]), z["indptr"], z["indices"], z["values"])
    a = adata_of(m, ctx, store)
    assert np.array_equal(statistics.compute_number(a, D.Row), z["number_row"])
    assert np.array_equal(statistics.compute_number(a, D.Column), z["number_col"])
    assert rel_err(statistics.compute_sum(a, D.Row), z["sum_row"]) < tol
    assert rel_err(statistics.compute_sum(a, D.Column), z["sum_col"]) < tol
    processing.normalize_total_inplace(a, 1e4, D.Row)
    assert rel_err(a.x_values(), z["norm_values"]) < tol
    processing.log1p_transform_inplace(a)
    assert rel_err(a.x_values(), z["log_values"]) < tol
    v = statistics.compute_variance(a, D.Column)
    np.testing.assert_allclose(v, z["log_var_col"], rtol=50 * tol, atol=tol)
    if store == 2:
        assert np.array_equal(dim_red.select_features(a, sr.FeatureSelection.HighlyVariable(len(z["hvg"]))), z["hvg"])


def test_fused_normalize_log1p_equals_two_calls(ctx):
    import singlerust_amd as sr
    from singlerust_amd.memory import processing
    m = create_large_test_data(900, 2500, 4.0, seed=9, dtype=np.float32)     # rows > 1024 nnz exist? no: ~625
    a, b = adata_of(m, ctx), adata_of(m, ctx)
    sums = processing.normalize_log1p_inplace(a, 1e4)
    processing.normalize_total_inplace(b, 1e4, sr.Direction.Row)
    processing.log1p_transform_inplace(b)
    assert np.array_equal(a.x_values(), b.x_values())
    assert rel_err(sums, oracle.compute_sum(m, ROW)) < 1e-6
    want = oracle.log1p_transform(oracle.normalize_total(m, 1e4, ROW)).values
    assert rel_err(a.x_values(), want) < TOL


def test_long_rows_take_the_streaming_path(ctx):
    """Rows longer than the 1024-value register cache (kRowCache*64) re-read from memory."""
    import singlerust_amd as sr
    from singlerust_amd.memory import processing, statistics
    rng = np.random.default_rng(2)
    lens = np.array([0, 1, 63, 64, 65, 1023, 1024, 1025, 4000, 7000, 2, 0])
    G = 8000
    indptr = np.zeros(len(lens) + 1, dtype=np.uint64)
    indptr[1:] = np.cumsum(lens)
    idx = np.concatenate([np.sort(rng.choice(G, l, replace=False)) for l in lens]).astype(np.uint64)
    val = rng.integers(1, 50, int(indptr[-1])).astype(np.float32)
    m = oracle.Csr(len(lens), G, indptr, idx, val)
    a = adata_of(m, ctx)
    assert np.array_equal(statistics.compute_sum(a, sr.Direction.Row), oracle.compute_sum(m, ROW))
    assert np.array_equal(statistics.compute_sum(a, sr.Direction.Column), oracle.compute_sum(m, COLUMN))
    processing.normalize_log1p_inplace(a, 1e4)
    want = oracle.log1p_transform(oracle.normalize_total(m, 1e4, ROW)).values
    assert rel_err(a.x_values(), want) < TOL
    cnt, s, sq = statistics.gene_moments(a)
    wc, ws, wq = oracle.gene_moments(oracle.Csr(len(lens), G, indptr, idx, a.x_values()))
    assert np.array_equal(cnt, wc)
    np.testing.assert_allclose(s, ws, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(sq, wq, rtol=1e-12, atol=1e-12)


def test_gene_tiling_many_genes(ctx):
    """G = 32000 > 4 x 8000: the per-gene pass runs 4 LDS gene tiles; counts stay bit-exact."""
    import singlerust_amd as sr
    from singlerust_amd.memory import statistics
    m = create_large_test_data(3000, 32000, 40.0, seed=4, dtype=np.uint16)
    a = adata_of(m, ctx)
    assert np.array_equal(statistics.compute_number(a, sr.Direction.Column), oracle.compute_number(m, COLUMN))
    assert np.array_equal(statistics.compute_sum(a, sr.Direction.Column), oracle.compute_sum(m, COLUMN))
    gv, wv = statistics.compute_variance(a, sr.Direction.Column), oracle.compute_variance(m, COLUMN)
    np.testing.assert_allclose(gv, wv, rtol=1e-9, atol=1e-9)
    (gmn, gmx), (wmn, wmx) = statistics.compute_min_max(a, sr.Direction.Column), oracle.compute_min_max(m, COLUMN)
    assert np.array_equal(gmn, wmn) and np.array_equal(gmx, wmx)


@pytest.mark.parametrize("n_genes", [9000, 32000, 65536, 70000])
def test_tile_cuts_against_the_oracle(ctx, n_genes):
    """The index mirror and the gene-tile cuts come out of ONE walk over the indices (k_narrow16_tiles: up to 65 536 genes); beyond,
    the cuts are binary-searched and there is no mirror (k_tile_ptr).  Either way the per-gene statistics are the oracle's —
    empty rows, rows confined to the last tile and a gene count that fills the 16-bit range included."""
    import singlerust_amd as sr
    from singlerust_amd.memory import statistics
    m = create_large_test_data(2500, n_genes, 30.0, seed=n_genes, dtype=np.uint16)
    # a few empty rows and a few rows living in the last tile only
    ip, ix, vals = m.indptr.copy(), m.indices.copy(), m.values.copy()
    for r in (0, 7, 2499):
        ix[ip[r]:ip[r + 1]] = np.sort(n_genes - 1 - np.arange(ip[r + 1] - ip[r]))
    keep = np.ones(len(ix), bool)
    for r in (3, 1200):
        keep[ip[r]:ip[r + 1]] = False
    lens = np.diff(ip)
    lens[[3, 1200]] = 0
    m2 = oracle.Csr(2500, n_genes, np.concatenate([[0], np.cumsum(lens)]), ix[keep], vals[keep])
    a = adata_of(m2, ctx)
    # up to 65 536 genes the upload sends the indices as 16-bit values and widens them on the device: what the handle holds is
    # the host's pattern
    from singlerust_amd import _ffi as F
    ip_d, ix_d = np.zeros(2501, np.uint64), np.zeros(len(m2.indices), np.uint64)
    F.check(F.lib().srx_matrix_download_pattern(a.x().handle, F.ptr(ip_d), F.ptr(ix_d)), ctx.handle)
    assert np.array_equal(ip_d, m2.indptr) and np.array_equal(ix_d, m2.indices)
    assert np.array_equal(statistics.compute_number(a, sr.Direction.Column), oracle.compute_number(m2, COLUMN))
    assert np.array_equal(statistics.compute_sum(a, sr.Direction.Column), oracle.compute_sum(m2, COLUMN))
    np.testing.assert_allclose(statistics.compute_variance(a, sr.Direction.Column), oracle.compute_variance(m2, COLUMN), rtol=1e-9,
                               atol=1e-9)


def statistics_row_numbers(a):
    import singlerust_amd as sr
    from singlerust_amd.memory import statistics
    return statistics.compute_number(a, sr.Direction.Row)


def test_error_behaviour(ctx):
    """Unsupported dtype -> the macro's panic message; bad CSR -> SRX_E_FORMAT / SRX_E_BOUNDS;
    NaN variance in HVG ranking -> SRX_E_NAN (partial_cmp().unwrap(), dim_red/mod.rs:138)."""
    import singlerust_amd as sr
    from singlerust_amd import _ffi
    from singlerust_amd.memory.processing import dim_red
    with pytest.raises(sr.SrxError) as e:
        sr.IMAnnData.new_basic((2, 3, [0, 1, 2], [0, 1], np.array([1, 2], dtype=np.int64)), ctx=ctx)
    assert e.value.code == _ffi.E_DTYPE and "not supported for this operation" in str(e.value)
    with pytest.raises(sr.SrxError) as e:
        sr.IMAnnData.new_basic((2, 3, [0, 1, 2], [0, 7], np.array([1.0, 2.0])), ctx=ctx)
    assert e.value.code == _ffi.E_BOUNDS
    with pytest.raises(sr.SrxError) as e:
        sr.IMAnnData.new_basic((1, 3, [0, 2], [2, 1], np.array([1.0, 2.0])), ctx=ctx)
    assert e.value.code == _ffi.E_FORMAT
    # corrupted INTERIOR row offsets (first and last are fine: the span test passes) on both upload routes — at most 65 536
    # columns, where the device walks the rows of the 16-bit indices before the canonical-CSR validation, and beyond:
    # a clean SRX_E_FORMAT, and the context is still good afterwards
    rng = np.random.default_rng(4)
    for n_cols in (300, 70_000):
        n_rows, per = 2000, 5
        indptr = (np.arange(n_rows + 1) * per).astype(np.uint64)
        idx = (rng.integers(0, n_cols - 100, (n_rows, 1)) + np.cumsum(rng.integers(1, 11, (n_rows, per)), axis=1)).ravel().astype(np.uint64)
        vals = np.ones(n_rows * per, np.float32)
        for bad_at, bad_val in ((700, 10 ** 12), (700, 5), (1999, n_rows * per + 1), (1, 2 ** 63)):
            ip = indptr.copy()
            ip[bad_at] = bad_val
            with pytest.raises(sr.SrxError) as e:
                sr.IMAnnData.new_basic((n_rows, n_cols, ip, idx, vals), ctx=ctx)
            assert e.value.code == _ffi.E_FORMAT and "row_offsets" in str(e.value)
        ok = sr.IMAnnData.new_basic((n_rows, n_cols, indptr, idx, vals), ctx=ctx)
        assert statistics_row_numbers(ok).tolist() == [per] * n_rows
    a = sr.IMAnnData.new_basic((2, 2, [0, 1, 2], [0, 1], np.array([np.nan, 1.0])), ctx=ctx)
    with pytest.raises(sr.SrxError) as e:
        dim_red.select_features(a, sr.FeatureSelection.HighlyVariable(1))
    assert e.value.code == _ffi.E_NAN


def test_empty_matrix_and_empty_rows(ctx):
    import singlerust_amd as sr
    from singlerust_amd.memory import processing, statistics
    a = sr.IMAnnData.new_basic((3, 4, [0, 0, 0, 0], np.zeros(0, np.uint64), np.zeros(0, np.float32)), ctx=ctx)
    assert statistics.compute_sum(a, sr.Direction.Row).tolist() == [0, 0, 0]
    assert statistics.compute_number(a, sr.Direction.Column).tolist() == [0, 0, 0, 0]
    assert statistics.compute_variance(a, sr.Direction.Column).tolist() == [0, 0, 0, 0]
    processing.normalize_log1p_inplace(a, 1e4)
    assert a.x_values().size == 0


def test_synth_device_equals_host_generator(ctx):
    """The in-HBM generator and its host twin are bit-identical (so the oracle sees the same X)."""
    import ctypes as C
    from singlerust_amd import _ffi, DeviceCsr
    lib = _ffi.lib()
    p = _ffi.SynthParams()
    lib.srx_synth_defaults(C.byref(p), 77, 5000, 6000, 0.04)
    h = C.c_void_p()
    _ffi.check(lib.srx_synth_generate(ctx.handle, C.byref(p), 1000, 4000, _ffi.F32, _ffi.STORE_F32, C.byref(h)), ctx.handle)
    d = DeviceCsr(ctx, h)
    ip = np.zeros(3001, dtype=np.uint64)
    lib.srx_synth_indptr(C.byref(p), 1000, 4000, _ffi.ptr(ip))
    idx = np.zeros(int(ip[-1]), np.uint64)
    val = np.zeros(int(ip[-1]), np.float32)
    lib.srx_synth_fill_host(C.byref(p), 1000, 4000, _ffi.ptr(ip), _ffi.ptr(idx), _ffi.ptr(val))
    assert d.info().nnz == ip[-1] and d.info().row_offset == 1000
    assert np.array_equal(d.values(), val)
    import singlerust_amd as sr
    from singlerust_amd.memory import statistics
    a = sr.IMAnnData(d, ip, idx, range(3000), range(6000))
    m = oracle.Csr(3000, 6000, ip, idx, val)
    assert np.array_equal(statistics.compute_number(a, sr.Direction.Column), oracle.compute_number(m, COLUMN))
    assert np.array_equal(statistics.compute_sum(a, sr.Direction.Column), oracle.compute_sum(m, COLUMN))
    assert np.array_equal(statistics.compute_sum(a, sr.Direction.Row), oracle.compute_sum(m, ROW))


def test_log1p_f32_accuracy(ctx):
    """The f32 ln_1p of the fused pass (hardware log2 + Goldberg correction) against f64 log1p over
    the whole positive range, tiny and huge arguments included; F32 stays F32 (transform/mod.rs:43-47)."""
    import singlerust_amd as sr
    from singlerust_amd.memory import processing
    x = np.concatenate([np.logspace(-30, 30, 20001), np.linspace(0, 100, 5001), [0.0, 1e-45, 3.4e38]]).astype(np.float32)
    n = x.size
    a = sr.IMAnnData.new_basic((1, n, [0, n], np.arange(n), x), ctx=ctx)
    processing.log1p_transform_inplace(a)
    assert a.x_dtype() == np.float32
    got = a.x_values().astype(np.float64)
    want = np.log1p(x.astype(np.float64))
    ok = want > 0
    assert np.max(np.abs(got[ok] - want[ok]) / want[ok]) < 3e-7
    assert np.all(got[~ok] == 0.0)


def test_prepare_and_clone_share_the_pattern_index(ctx):
    """srx_matrix_prepare builds the pattern-only gene-tile cuts once; clones inherit them and give
    the same per-gene statistics as a matrix that builds them lazily."""
    import singlerust_amd as sr
    from singlerust_amd.memory import processing, statistics
    m = create_large_test_data(2000, 20000, 30.0, seed=8, dtype=np.float32)      # 3 gene tiles
    a = adata_of(m, ctx)
    a.x().prepare()
    b = a.deep_clone()
    processing.normalize_log1p_inplace(b, 1e4)
    c = adata_of(m, ctx)
    processing.normalize_log1p_inplace(c, 1e4)
    for d in (sr.Direction.Row, sr.Direction.Column):
        assert np.array_equal(statistics.compute_number(b, d), statistics.compute_number(c, d))
    np.testing.assert_allclose(statistics.compute_variance(b, sr.Direction.Column),
                               statistics.compute_variance(c, sr.Direction.Column), rtol=1e-12, atol=1e-14)
    assert np.array_equal(statistics.compute_sum(a, sr.Direction.Column), oracle.compute_sum(m, COLUMN) if False else statistics.compute_sum(adata_of(m, ctx), sr.Direction.Column))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.uint16])
def test_compute_qc_variables_matches_oracle(ctx, dtype):
    """compute_qc_variables (statistics/mod.rs:48-72): the eight vectors from one row pass + one column
    pass, against the oracle's eight separate reference loops — counts bit-exact, sums of integer data
    bit-exact, variances to rounding, NaN exactly for the empty cells (csr.rs:161)."""
    import scipy.sparse as sp
    import singlerust_amd as sr
    from singlerust_amd.memory import statistics
    rng = np.random.default_rng(11)
    n, g = 700, 450
    x = sp.random(n, g, density=0.05, random_state=3, format="csr",
                  data_rvs=lambda s: rng.integers(1, 40, s).astype(np.float64)).astype(dtype)
    x[17] = 0; x[300] = 0                                   # two empty cells
    x.eliminate_zeros(); x.sort_indices()
    m = oracle.Csr(n, g, x.indptr, x.indices, x.data.astype(dtype))
    a = sr.IMAnnData.new_basic((n, g, m.indptr, m.indices, m.values), ctx=ctx)
    got = statistics.compute_qc_variables(a)
    assert np.array_equal(got.num_per_cell, oracle.compute_number(m, ROW))
    assert np.array_equal(got.num_per_gene, oracle.compute_number(m, COLUMN))
    assert np.array_equal(got.expr_per_cell, oracle.compute_sum(m, ROW))        # integer data: exact
    assert np.array_equal(got.expr_per_gene, oracle.compute_sum(m, COLUMN))
    wv = oracle.compute_variance(m, ROW)
    assert np.array_equal(np.isnan(got.variance_per_cell), np.isnan(wv)) and np.isnan(wv).sum() == 2
    ok = ~np.isnan(wv)
    np.testing.assert_allclose(got.variance_per_cell[ok], wv[ok], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(got.variance_per_gene, oracle.compute_variance(m, COLUMN), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(got.std_dev_per_cell[ok], oracle.compute_std_dev(m, ROW)[ok], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(got.std_dev_per_gene, oracle.compute_std_dev(m, COLUMN), rtol=1e-12, atol=1e-12)
    statistics.qc_vars_inplace(a)
    assert set(a.obs) >= {"num_genes_per_cell", "sum_expr_per_cell", "var_expr_per_cell", "std_dev_per_cell"}
    assert set(a.var) >= {"num_cells_per_gene", "sum_expr_per_gene", "var_expr_per_gene", "std_dev_per_gene"}


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_upload_from_pinned_caller_buffers(ctx, dtype):
    """Values the caller holds in PINNED host memory (hipHostMalloc — a backed reader's tile buffers) are copied by one DMA straight
    out of them, under the workers' index narrowing (upload_on); pageable values go through the staging buffers.  Same handle
    either way: pattern, values and the per-gene sums."""
    import ctypes as C
    import singlerust_amd as sr
    from singlerust_amd import _ffi as F
    from singlerust_amd.memory import statistics
    m = create_large_test_data(40_000, 3000, 40.0, seed=77, dtype=np.uint16)          # ~3M non-zeros: above the direct-copy threshold (8 MiB)
    vals = m.values.astype(dtype)
    hip = C.CDLL("libamdhip64.so")
    hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    hip.hipHostFree.argtypes = [C.c_void_p]
    raw = C.c_void_p()
    assert hip.hipHostMalloc(C.byref(raw), vals.nbytes, 0) == 0
    try:
        ctype = C.c_float if dtype == np.float32 else C.c_double
        pinned = np.ctypeslib.as_array(C.cast(raw, C.POINTER(ctype)), shape=(len(vals),))
        pinned[:] = vals
        store = F.STORE_F32 if dtype == np.float32 else F.STORE_F64
        got = []
        for v in (pinned, vals):
            d = sr.DeviceCsr.upload(ctx, m.n_rows, m.n_cols, m.indptr.astype(np.uint64), m.indices.astype(np.uint64), v, store)
            a = sr.IMAnnData(d, None, None, [], [])
            ip_d, ix_d = np.zeros(m.n_rows + 1, np.uint64), np.zeros(len(vals), np.uint64)
            F.check(F.lib().srx_matrix_download_pattern(d.handle, F.ptr(ip_d), F.ptr(ix_d)), ctx.handle)
            assert np.array_equal(ip_d, m.indptr) and np.array_equal(ix_d, m.indices)
            got.append((d.values(np.float64), statistics.compute_sum(a, sr.Direction.Column)))
            d.free()
        assert np.array_equal(got[0][0], vals.astype(np.float64)) and np.array_equal(got[1][0], got[0][0])
        assert np.array_equal(got[0][1], got[1][1])
    finally:
        hip.hipHostFree(raw)
