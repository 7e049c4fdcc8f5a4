"""GPU parity tests (-m gpu) of the QC filters — SURVEY.md §8(f) rank 1: filter_cells / filter_genes
(src/memory/processing/mod.rs:86-146, :245-299) against the numpy restatement oracle/filter_oracle.py, and
the reference's own tests test_filter_cells / test_filter_genes (:385-417) restated."""
import numpy as np
import pytest

import oracle
from oracle import filter_oracle as fo
from util import create_large_test_data

pytestmark = pytest.mark.gpu


def _adata(m, ctx, store=0):
    import singlerust_amd as sr
    return sr.IMAnnData.new_basic((m.n_rows, m.n_cols, m.indptr, m.indices, m.values), ctx=ctx, store=store)


def _flex(sr, v):
    if v is None:
        return sr.FlexValue.None_
    return sr.FlexValue.Absolute(v[1]) if v[0] == "abs" else sr.FlexValue.Relative(v[1])


def _same_csr(a, want):
    assert a.n_obs() == want.n_rows and a.n_vars() == want.n_cols
    assert np.array_equal(a.indptr, want.indptr) and np.array_equal(a.indices, want.indices)
    assert np.array_equal(a.x_values(np.float64), want.values.astype(np.float64))       # a gather: bit-exact


COMBOS = [(fo.absolute(5), fo.absolute(15)), (fo.relative(0.1), fo.relative(0.9)), (fo.absolute(6), fo.relative(0.8)),
          (fo.relative(0.2), fo.absolute(14)), (fo.absolute(8), None), (None, fo.absolute(12)),
          (fo.relative(0.35), None), (None, fo.relative(0.65)), (None, None)]


@pytest.mark.parametrize("lower,upper", COMBOS)
def test_filter_cells_all_flexvalue_combinations(ctx, lower, upper):
    import singlerust_amd as sr
    from singlerust_amd.memory import processing
    m = create_large_test_data(1000, 100, 10.0, seed=4)
    want, wmask = fo.filter_cells(m, lower, upper)
    got = processing.filter_cells(_adata(m, ctx), _flex(sr, lower), _flex(sr, upper))
    assert np.array_equal(got.uns["filter_mask"], wmask)
    _same_csr(got, want)
    assert got.obs_names == [f"obs{i}" for i in np.flatnonzero(wmask)]


@pytest.mark.parametrize("lower,upper", [(fo.absolute(50), fo.absolute(100)), (fo.relative(0.1), fo.relative(0.9)),
                                          (fo.absolute(90), fo.relative(0.7)), (fo.relative(0.3), None), (None, None)])
def test_filter_genes_combinations(ctx, lower, upper):
    import singlerust_amd as sr
    from singlerust_amd.memory import processing
    m = create_large_test_data(1000, 100, 10.0, seed=5)
    want, wmask = fo.filter_genes(m, lower, upper)
    got = processing.filter_genes(_adata(m, ctx), _flex(sr, lower), _flex(sr, upper))
    assert np.array_equal(got.uns["filter_mask"], wmask)
    _same_csr(got, want)
    assert got.var_names == [f"var{i}" for i in np.flatnonzero(wmask)]


def test_reference_test_filter_cells_and_genes(ctx):
    """mod.rs:385-417: on the random 1000 x 100 matrix both the Absolute(5, 15) / Relative(0.1, 0.9) cell filters and
    the Absolute(50, 100) / Relative(0.1, 0.9) gene filters drop something."""
    import singlerust_amd as sr
    from singlerust_amd.memory import processing
    FV = sr.FlexValue
    m = create_large_test_data(1000, 100, 10.0, seed=6)
    a = _adata(m, ctx)
    assert processing.filter_cells(a, FV.Absolute(5), FV.Absolute(15)).n_obs() < a.n_obs()
    assert processing.filter_cells(a, FV.Relative(0.1), FV.Relative(0.9)).n_obs() < a.n_obs()
    assert processing.filter_genes(a, FV.Absolute(50), FV.Absolute(100)).n_vars() < a.n_vars()
    assert processing.filter_genes(a, FV.Relative(0.1), FV.Relative(0.9)).n_vars() < a.n_vars()


def test_filters_inplace_then_the_hot_path(ctx):
    """filter_cells_inplace -> filter_genes_inplace -> normalize_total -> log1p on the filtered matrix equals the
    oracle on the oracle-filtered matrix (the filters sit right before the hot path)."""
    import singlerust_amd as sr
    from singlerust_amd.memory import processing, statistics
    FV = sr.FlexValue
    m = create_large_test_data(800, 120, 8.0, seed=9)
    w1, _ = fo.filter_cells(m, fo.absolute(8), fo.relative(0.95))
    w2, _ = fo.filter_genes(w1, fo.absolute(60), None)
    a = _adata(m, ctx, store=2)
    processing.filter_cells_inplace(a, FV.Absolute(8), FV.Relative(0.95))
    processing.filter_genes_inplace(a, FV.Absolute(60), FV.None_)
    _same_csr(a, w2)
    processing.normalize_total_inplace(a, 1e4, sr.Direction.Row)
    processing.log1p_transform_inplace(a)
    want = oracle.log1p_transform(oracle.normalize_total(w2, 1e4, oracle.ROW))
    np.testing.assert_allclose(a.x_values(), want.values, rtol=5e-16, atol=0)
    assert np.array_equal(statistics.compute_number(a, sr.Direction.Column), oracle.compute_number(w2, oracle.COLUMN))


def test_filter_everything_and_bad_quantile(ctx):
    import singlerust_amd as sr
    from singlerust_amd import _ffi
    from singlerust_amd.memory import processing
    FV = sr.FlexValue
    m = create_large_test_data(200, 50, 10.0, seed=2)
    got = processing.filter_cells(_adata(m, ctx), FV.Absolute(10_000), FV.None_)        # nothing survives
    assert got.n_obs() == 0 and got.n_vars() == 50 and got.x().info().nnz == 0
    with pytest.raises(sr.SrxError) as e:
        processing.filter_cells(_adata(m, ctx), FV.Relative(1.5), FV.None_)
    assert e.value.code == _ffi.E_ARG
