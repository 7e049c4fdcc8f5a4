"""CPU tests (-m "not gpu"): pin the oracle before trusting it.

* against the hand-derived 4x5 KAT (tests/golden/kat_4x5.json) and the hand-derived PCA KAT (kat_pca_6x4.json);
* against the committed golden vectors computed with independent numpy/scipy/sklearn maths
  (tests/golden/make_golden.py);
* against the reference's own property test, restated: after normalize_total every
  (non-empty) row / column sums to target ± 1e-6 (src/memory/processing/mod.rs:419-481).
"""
import json
import os

import numpy as np
import pytest

import oracle
from oracle import COLUMN, ROW, Csr
from oracle import pca_oracle
from util import create_large_test_data

GOLD = os.path.join(os.path.dirname(__file__), "golden")
ALL_DTYPES = [np.int8, np.int16, np.int32, np.uint8, np.uint16, np.uint32, np.float32, np.float64]


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    m = Csr(int(z["n_rows"]), int(z["n_cols"]), z["indptr"], z["indices"], z["values"])
    return m, z


@pytest.mark.parametrize("dtype", ALL_DTYPES)
def test_kat_4x5(dtype):
    k = json.load(open(os.path.join(GOLD, "kat_4x5.json")))
    m = Csr(k["n_rows"], k["n_cols"], k["indptr"], k["indices"], np.array(k["data"], dtype=dtype))
    assert oracle.compute_number(m, ROW).tolist() == k["number_row"]
    assert oracle.compute_number(m, COLUMN).tolist() == k["number_col"]
    assert oracle.compute_sum(m, ROW).tolist() == k["sum_row"]
    assert oracle.compute_sum(m, COLUMN).tolist() == k["sum_col"]
    assert oracle.compute_variance(m, COLUMN).tolist() == k["var_col"]
    vr = oracle.compute_variance(m, ROW)
    assert np.isnan(vr[2]) and vr[[0, 1, 3]].tolist() == [1.0, 0.0, 2.0]      # csr.rs:161: 0/0 for the empty row
    n = oracle.normalize_total(m, 1e4, ROW)
    assert n.values.dtype == np.float64                                      # scale/mod.rs:82: becomes F64
    assert [float.hex(x) for x in n.values] == k["normalized_hex"]
    lg = oracle.log1p_transform(n)
    assert [float.hex(x) for x in lg.values] == k["log1p_hex"]
    v = oracle.compute_variance(lg, COLUMN)
    np.testing.assert_allclose(v, k["log_var_col"], rtol=0, atol=5e-9)
    assert oracle.select_hvg(v, 2).tolist() == k["hvg2"]
    assert oracle.select_hvg(v, 5).tolist() == k["hvg5"]                      # stable ties: ascending index
    mn, mx = oracle.compute_min_max(m, COLUMN)
    assert mn.tolist() == [1, 1, 4, 1, np.inf] and mx.tolist() == [2, 3, 4, 1, -np.inf]


def test_log1p_dtype_semantics():
    """transform/mod.rs:36-57: F32 stays F32 (f32::ln_1p), ints are promoted to F64."""
    idx = dict(n_rows=1, n_cols=3, indptr=[0, 3], indices=[0, 1, 2])
    f32 = oracle.log1p_transform(Csr(values=np.array([0.5, 2.0, 1e-8], np.float32), **idx))
    assert f32.values.dtype == np.float32
    np.testing.assert_allclose(f32.values, np.log1p(np.array([0.5, 2.0, 1e-8], np.float64)), rtol=2e-7)  # libm log1pf: <= 1 ulp
    u8 = oracle.log1p_transform(Csr(values=np.array([1, 2, 255], np.uint8), **idx))
    assert u8.values.dtype == np.float64
    np.testing.assert_allclose(u8.values, np.log1p(np.array([1.0, 2.0, 255.0])), rtol=5e-16)   # libm vs numpy: <= 1 ulp


@pytest.mark.parametrize("name", ["ref_shape_1000x100", "counts_64x40", "planted_600x240"])
def test_against_golden(name):
    m, z = load(name)
    assert np.array_equal(oracle.compute_number(m, ROW), z["number_row"])
    assert np.array_equal(oracle.compute_number(m, COLUMN), z["number_col"])
    np.testing.assert_allclose(oracle.compute_sum(m, ROW), z["sum_row"], rtol=1e-13)
    np.testing.assert_allclose(oracle.compute_sum(m, COLUMN), z["sum_col"], rtol=1e-13)
    np.testing.assert_allclose(oracle.compute_variance(m, COLUMN), z["var_col"], rtol=1e-9, atol=1e-9)
    n = oracle.normalize_total(m, 1e4, ROW)
    np.testing.assert_allclose(n.values, z["norm_values"], rtol=1e-13)
    lg = oracle.log1p_transform(n)
    np.testing.assert_allclose(lg.values, z["log_values"], rtol=1e-13)
    v = oracle.compute_variance(lg, COLUMN)
    np.testing.assert_allclose(v, z["log_var_col"], rtol=1e-9, atol=1e-12)
    if "hvg" in z:
        assert np.array_equal(oracle.select_hvg(v, len(z["hvg"])), z["hvg"])


def test_integer_sums_bit_exact():
    m, z = load("counts_64x40")
    for dt in (np.uint8, np.int16, np.uint32, np.float32):
        mm = m.with_values(m.values.astype(dt))
        assert np.array_equal(oracle.compute_sum(mm, ROW), z["sum_row"])
        assert np.array_equal(oracle.compute_sum(mm, COLUMN), z["sum_col"])


def test_reference_property_normalize_total():
    """src/memory/processing/mod.rs:419-481 restated (1000 x 100, sparsity 10, target 1e4)."""
    m = create_large_test_data(1000, 100, 10.0, seed=123)
    target = 1e4
    n = oracle.normalize_total(m, target, ROW)
    assert n.values.dtype == np.float64                      # `DynCsrMatrix::F64(csr)` arm, :135
    rows = oracle.compute_sum(n, ROW)
    nonempty = oracle.compute_number(m, ROW) > 0             # the reference test is flaky on empty rows
    assert np.all(np.abs(rows[nonempty] - target) < 1e-6)    # check_row_sums :451-462
    assert np.all(rows[~nonempty] == 0.0)                    # scale/mod.rs:10-11: sum == 0 -> scale 0
    c = oracle.normalize_total(m, target, COLUMN)
    cols = oracle.compute_sum(c, COLUMN)
    assert np.all(np.abs(cols - target) < 1e-6)              # check_column_sums :464-481


def test_hvg_stable_and_nan():
    v = np.array([1.0, 3.0, 3.0, 0.5, 3.0, 0.0, 0.0])
    assert oracle.select_hvg(v, 4).tolist() == [1, 2, 4, 0]
    assert oracle.select_hvg(v, 100).tolist() == [1, 2, 4, 0, 3, 5, 6]       # take(n) clamps
    with pytest.raises(ValueError):
        oracle.select_hvg(np.array([1.0, np.nan]), 1)                        # unwrap() panic


def test_densify_selected_order():
    """shared/mod.rs:230-259: column c of the dense matrix is gene sel[c] (selection order)."""
    m, _ = load("counts_64x40")
    sel = np.array([12, 3, 30], dtype=np.uint64)
    d = oracle.densify_selected(m, sel)
    import scipy.sparse as sp
    full = sp.csr_matrix((m.values, m.indices.astype(np.int64), m.indptr.astype(np.int64)),
                         shape=(m.n_rows, m.n_cols)).toarray()
    assert np.array_equal(d, full[:, [12, 3, 30]])


def test_pca_oracle_vs_sklearn():
    """The numpy restatement of pca/mod.rs:74-215 equals StandardScaler(ddof 0) + full-SVD PCA up
    to the sign of each component; eigenvalue normalisation is s^2/(n-1) over ALL components."""
    m, z = load("planted_600x240")
    lg = oracle.log1p_transform(oracle.normalize_total(m, 1e4, ROW))
    sel = pca_oracle.select_features_hvg(lg, 120)
    assert np.array_equal(sel, z["hvg"])
    scores, comps, evr, mean, std = pca_oracle.pca_inplace(lg, 5, None, None, sel)
    for c in range(5):
        s = np.sign(np.dot(comps[:, c], z["pca_components"][:, c]))
        np.testing.assert_allclose(s * comps[:, c], z["pca_components"][:, c], atol=1e-9)
        np.testing.assert_allclose(s * scores[:, c], z["pca_scores"][:, c], atol=1e-8)
    np.testing.assert_allclose(evr, z["pca_evr"], rtol=1e-10)
    # defaults: n_components None -> 2 (dim_red/mod.rs:52)
    s2, c2, *_ = pca_oracle.pca_inplace(lg, None, None, None, sel)
    assert s2.shape == (600, 2) and c2.shape == (120, 2)


def _kat_pca():
    k = json.load(open(os.path.join(GOLD, "kat_pca_6x4.json")))
    return k, {n: np.array(k[n], dtype=np.float64) for n in ("mean", "std", "explained_variance_ratio", "components", "scores",
                                                               "loadings", "eigenvalues")}


@pytest.mark.parametrize("dtype", [np.uint8, np.float32, np.float64])
def test_pca_oracle_against_the_hand_derived_kat(dtype):
    """tests/golden/kat_pca_6x4.json: a rank-2 6 x 4 matrix whose standardised SVD follows by hand from
    pca/mod.rs:87-144 — the first PCA vector that does not come out of numpy.  Pins the oracle's mean / std (ddof 0),
    eigenvalue normalisation s^2/(n-1), ratio over ALL eigenvalues, component / score / loading layout."""
    k, w = _kat_pca()
    m = Csr(k["n_rows"], k["n_cols"], k["indptr"], k["indices"], np.array(k["data"], dtype=dtype))
    assert np.array_equal(oracle.densify_selected(m, np.arange(4, dtype=np.uint64)), np.array(k["dense"], dtype=np.float64))
    pca = pca_oracle.Pca(n_components=k["n_components"], center=True, scale=True)
    pca.fit(np.array(k["dense"], dtype=np.float64))
    np.testing.assert_allclose(pca.mean, w["mean"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(pca.std_dev, w["std"], rtol=1e-15)
    np.testing.assert_allclose(pca.eigenvalues, w["eigenvalues"], atol=1e-14)
    assert abs(pca.total_variance - k["total_variance"]) < 1e-14
    scores, comps, evr, mean, std = pca_oracle.pca_inplace(m, None, None, None, None)     # defaults: 2 components, centre, scale
    np.testing.assert_allclose(evr, w["explained_variance_ratio"], rtol=1e-14)
    for c in range(2):
        s = np.sign(np.dot(comps[:, c], w["components"][:, c]))
        np.testing.assert_allclose(s * comps[:, c], w["components"][:, c], atol=1e-14)
        np.testing.assert_allclose(s * scores[:, c], w["scores"][:, c], atol=1e-14)
        np.testing.assert_allclose(s * pca.compute_loadings()[c], w["loadings"][c], atol=1e-14)


def test_omp_cov_selected_is_the_k_by_k_form_of_the_pca_oracle():
    """oracle/omp_baseline.c::orc_omp_cov_selected (the full-size c3 eigen-reference of tests/test_fullsize_gpu.py) against the
    exact-SVD oracle: eigh of Z^T Z gives the SVD's V and s^2, mean / sd per slot in selection order."""
    m, z = load("planted_600x240")
    lg = oracle.log1p_transform(oracle.normalize_total(m, 1e4, ROW))
    sel = pca_oracle.select_features_hvg(lg, 120)
    cov, mean, sd = oracle.omp_cov_selected(lg, sel, 3)
    scores, comps, evr, wmean, wstd = pca_oracle.pca_inplace(lg, 5, None, None, sel)
    np.testing.assert_allclose(mean, wmean, rtol=1e-13)
    np.testing.assert_allclose(sd, wstd, rtol=1e-13)
    assert np.array_equal(cov, cov.T)
    w, v = np.linalg.eigh(cov)
    order = np.argsort(w)[::-1][:5]
    np.testing.assert_allclose(w[order] / np.trace(cov), evr, rtol=1e-10)
    for c in range(5):
        s = np.sign(np.dot(v[:, order[c]], comps[:, c]))
        np.testing.assert_allclose(s * v[:, order[c]], comps[:, c], atol=1e-9)
    with pytest.raises(ValueError):
        oracle.omp_cov_selected(lg, np.array([3, 3], np.uint64), 2)


def test_filter_oracle_kat_4x5():
    """Hand-derived from the cited loops (processing/mod.rs:33-84, 148-174) on the 4x5 KAT matrix:
    row counts [2,1,0,3], row sums [4,2,0,6]; column counts [2,2,1,1,0], column sums [3,4,4,1,0]."""
    from oracle import filter_oracle as fo
    m = oracle.Csr(4, 5, [0, 2, 3, 3, 6], [1, 3, 0, 0, 1, 2], np.array([3, 1, 2, 1, 1, 4], dtype=np.float64))
    out, mask = fo.filter_cells(m, fo.absolute(2), fo.NONE)                 # n_genes >= 2
    assert mask.tolist() == [True, False, False, True]
    assert out.indptr.tolist() == [0, 2, 5] and out.indices.tolist() == [1, 3, 0, 1, 2]
    # Relative limits act on the SUMS: sorted [0,2,4,6]; q(0.5) = 2 + (4-2)*0.5 = 3; q(1.0) = 6
    out, mask = fo.filter_cells(m, fo.relative(0.5), fo.relative(1.0))
    assert mask.tolist() == [True, False, False, True]
    # mixed: count >= 1 and sum <= q(0.5) = 3  -> only row 1 (count 1, sum 2)
    out, mask = fo.filter_cells(m, fo.absolute(1), fo.relative(0.5))
    assert mask.tolist() == [False, True, False, False] and out.values.tolist() == [2.0]
    # genes: sums sorted [0,1,3,4,4]: q(0.25) = 1, q(0.9): idx 3.6 -> 4 + (4-4)*0.6 = 4
    out, mask = fo.filter_genes(m, fo.relative(0.25), fo.relative(0.9))
    assert mask.tolist() == [True, True, True, True, False] and out.n_cols == 4
    out, mask = fo.filter_genes(m, fo.absolute(2), fo.absolute(2))
    assert mask.tolist() == [True, True, False, False, False]
    assert out.indptr.tolist() == [0, 1, 2, 2, 4] and out.indices.tolist() == [1, 0, 0, 1] and out.values.tolist() == [3, 2, 1, 1]
    # (None, None) keeps everything; not-Relative limits use f64::MIN / f64::MAX
    assert fo.calculate_percentiles(np.array([1.0, 2.0]), fo.NONE, fo.absolute(3)) == (fo.F64_MIN, fo.F64_MAX)
    out, mask = fo.filter_cells(m, fo.NONE, fo.NONE)
    assert mask.all() and out.values.tolist() == m.values.tolist()


def test_backed_oracle_chunked_equals_whole_and_flat_store_roundtrip(tmp_path):
    """oracle/backed_oracle.py: chunked number / sum == the whole-matrix helpers for every chunk size (Column
    always; Row with per-chunk offsets), the as-written Row loop folds all chunks onto the first `size` entries;
    the flat store (singlerust_amd.backed.BackedCsr) round-trips and iterates like ArrayElemOp::iter."""
    from oracle import backed_oracle
    from singlerust_amd.backed import BackedCsr
    from util import create_large_test_data
    m = create_large_test_data(300, 40, 10.0, seed=1, dtype=np.uint16)
    for ch in (1, 37, 100, 500):
        for d in (ROW, COLUMN):
            assert np.array_equal(backed_oracle.number_chunked(m, ch, d), oracle.compute_number(m, d))
            assert np.array_equal(backed_oracle.sum_chunked(m, ch, d), oracle.compute_sum(m, d))
    whole = oracle.compute_number(m, ROW)
    aw = backed_oracle.number_chunked(m, 100, ROW, as_written=True)
    assert np.array_equal(aw[:100], whole[:100] + whole[100:200] + whole[200:300]) and not aw[100:].any()
    p = str(tmp_path / "store")
    BackedCsr.write(p, m.indptr, m.indices, m.values, m.n_cols)
    b = BackedCsr.open(p)
    assert (b.n_rows, b.n_cols, b.nnz) == (300, 40, len(m.values)) and b.values.dtype == np.uint16
    seen = 0
    for chunk, start, end in b.iter(64):
        assert chunk.n_rows == end - start and start == seen
        lo, hi = int(m.indptr[start]), int(m.indptr[end])
        assert chunk.nnz == hi - lo and np.array_equal(chunk.indices, np.asarray(m.indices[lo:hi], dtype=np.uint64))
        assert int(chunk.indptr[0]) == lo                  # a window of the row offsets, not rebased
        seen = end
    assert seen == 300


def test_csc_oracle_is_the_csr_oracle_of_the_transpose():
    """oracle/csc_oracle.py restates helper/csc.rs loop by loop; the product holds a CSC matrix as the CSR of X^T and
    exchanges the direction.  That identity is pinned here on the two restatements: bit-identical number / sum /
    variance / min-max (same accumulation orders), including NaN for an empty column (csc.rs:164-177 == csr.rs:158-171
    on the transpose) and 0 for an empty row."""
    import scipy.sparse as sp
    from oracle import csc_oracle
    rng = np.random.default_rng(0)
    x = sp.random(200, 60, density=0.1, random_state=1, format="lil", data_rvs=lambda s: rng.uniform(0, 50, s))
    x[:, 7] = 0
    x[11, :] = 0
    x = x.tocsc()
    x.eliminate_zeros()
    x.sort_indices()
    m = csc_oracle.Csc.from_scipy(x)
    t = Csr(60, 200, x.indptr, x.indices, x.data)                  # the CSR arrays of X^T are the CSC arrays of X
    for d, fd in ((ROW, COLUMN), (COLUMN, ROW)):
        assert np.array_equal(csc_oracle.compute_number(m, d), oracle.compute_number(t, fd))
        assert np.array_equal(csc_oracle.compute_sum(m, d), oracle.compute_sum(t, fd))
        assert np.array_equal(csc_oracle.compute_variance(m, d), oracle.compute_variance(t, fd), equal_nan=True)
        a, b = csc_oracle.compute_min_max(m, d), oracle.compute_min_max(t, fd)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.isnan(csc_oracle.compute_variance(m, COLUMN)[7]) and csc_oracle.compute_variance(m, ROW)[11] == 0.0
    # normalize_total on CSC == on the CSR of the same X (the reference's test_normalize_total bar: sums == target)
    n = csc_oracle.normalize_total(m, 1e4, ROW)
    s = csc_oracle.compute_sum(n, ROW)
    assert np.all(np.abs(s[s > 0] - 1e4) < 1e-6)
    dense = csc_oracle.densify_selected(m, [3, 1, 59])
    assert np.array_equal(dense, x.toarray()[:, [3, 1, 59]])


def test_oracle_under_address_and_ub_sanitizers():
    """SURVEY.md 5 (race detection / sanitizers): the C restatement is raw-pointer code and it is the checker — its
    entry points are walked on the KAT, on empty rows / genes and on random matrices of every dtype under ASan + UBSan."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    p = subprocess.run(["make", "-C", root, "sanitize"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "oracle sanitize walk: ok" in p.stdout
    os.remove(os.path.join(root, "sanitize_driver"))
