"""Adversarial inputs through filters, statistics, backed sessions and CSC handles — development helper."""
import sys
import traceback

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, ".")
import singlerust_amd as sr
from singlerust_amd import _ffi as F, backed
from singlerust_amd.memory import processing, statistics as st

ctx = sr.Context.default()
rng = np.random.default_rng(0)
base = sp.random(500, 80, density=0.1, random_state=1, data_rvs=lambda s: rng.integers(1, 20, s).astype(np.float32), dtype=np.float32, format="csr")


def attempt(name, fn):
    try:
        print(f"{name}: {fn()}", flush=True)
    except F.SrxError as e:
        print(f"{name}: SrxError {str(e)[:110]}", flush=True)
    except Exception:
        print(f"{name}: EXCEPTION {traceback.format_exc().splitlines()[-1][:150]}", flush=True)


def ad(x, **kw):
    x = sp.csr_matrix(x) if not sp.issparse(x) or x.format not in ("csr", "csc") else x
    x.sort_indices()
    return sr.IMAnnData.new_basic(x, ctx=ctx, **kw)


attempt("filter everything out (cells)", lambda: processing.filter_cells(ad(base), sr.FlexValue.Absolute(10**6), sr.FlexValue.NoLimit()).n_obs())
attempt("filter everything out (genes)", lambda: processing.filter_genes(ad(base), sr.FlexValue.Absolute(10**6), sr.FlexValue.NoLimit()).n_vars())


def empty_then_stats():
    f = processing.filter_cells(ad(base), sr.FlexValue.Absolute(10**6), sr.FlexValue.NoLimit())
    return (st.compute_sum(f, sr.Direction.Column).sum(), st.compute_number(f, sr.Direction.Row).shape)


attempt("stats on a 0-row matrix", empty_then_stats)
attempt("normalize on a 0-row matrix", lambda: processing.normalize_total_inplace(processing.filter_cells(ad(base), sr.FlexValue.Absolute(10**6), sr.FlexValue.NoLimit()), 1e4, sr.Direction.Row))
attempt("1 x 1 matrix stats", lambda: st.compute_variance(ad(sp.csr_matrix(np.array([[3.0]], dtype=np.float32))), sr.Direction.Column))
attempt("single column", lambda: st.compute_sum(ad(base[:, :1]), sr.Direction.Row).sum())
attempt("relative filter on all-equal sums", lambda: processing.filter_cells(ad(sp.csr_matrix(np.ones((50, 10), dtype=np.float32))), sr.FlexValue.Relative(0.1), sr.FlexValue.Relative(0.9)).n_obs())
attempt("nan value", lambda: st.compute_sum(ad(sp.csr_matrix(np.array([[1.0, np.nan], [2.0, 0.0]], dtype=np.float32))), sr.Direction.Row))
attempt("negative values log1p", lambda: (processing.log1p_transform_inplace(a := ad(sp.csr_matrix(np.array([[-0.5, -2.0], [2.0, 0.0]], dtype=np.float32)))), a.x_values())[1])


def backed_cases():
    x = backed.BackedCsr(base.indptr.astype(np.uint64), base.indices.astype(np.uint64), base.data, 80)
    a = backed.BackedAnnData(x, ctx)
    out = []
    for chunk in (1, 499, 500, 501, 10**6):
        r = backed.processing.pca_pipeline(a, chunk, 1e4, 40, 5)
        out.append((chunk, r.x_pca.shape, float(r.info.residual)))
    return out


attempt("backed chunk sizes", backed_cases)


def backed_empty_rows():
    e = sp.vstack([sp.csr_matrix((300, 80), dtype=np.float32), base, sp.csr_matrix((300, 80), dtype=np.float32)]).tocsr()
    x = backed.BackedCsr(e.indptr.astype(np.uint64), e.indices.astype(np.uint64), e.data, 80)
    r = backed.processing.pca_pipeline(backed.BackedAnnData(x, ctx), 100, 1e4, 40, 5)
    return r.x_pca.shape, float(r.info.residual), np.abs(r.x_pca[:300]).max() < 1e9


attempt("backed with all-empty tiles", backed_empty_rows)
attempt("csc empty", lambda: st.compute_number(ad(sp.csc_matrix((30, 20), dtype=np.float32)), sr.Direction.Row).sum())
attempt("csc pipeline tiny", lambda: processing.normalize_total_inplace(ad(base.tocsc()), 1e4, sr.Direction.Row))
attempt("qc on empty", lambda: st.compute_qc_variables(ad(sp.csr_matrix((10, 5), dtype=np.float32))).num_per_cell.sum())
