"""The pipeline's alternative orders (environment switches read once per process, so each runs in its own process):
default = the moments pass stores the transformed values; SRX_WB_SIDE=1 = the in-place pass on the side stream beside the
iteration; SRX_NO_LAZY=1 = round 1's order (in-place pass first, moments of the stored values).  All three must leave the same
matrix (to the storage type's rounding), the first two the same selection, and all the same principal components."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(tmp, name, store, **env):
    out = os.path.join(tmp, name + ".npz")
    e = dict(os.environ, **env)
    e["PYTHONPATH"] = ROOT + os.pathsep + e.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "pipeline_order_worker.py"), out, str(store)], env=e,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


@pytest.mark.gpu
@pytest.mark.parametrize("store,vtol", [(1, 4e-7), (2, 1e-14)])
def test_pipeline_orders_agree(tmp_path, store, vtol):
    base = run(str(tmp_path), "default", store)
    side = run(str(tmp_path), "side", store, SRX_WB_SIDE="1")
    eager = run(str(tmp_path), "eager", store, SRX_NO_LAZY="1")
    # the device-side ranking of HighlyVariable(n): candidates above a sampled threshold (default), every gene
    # (SRX_HVG_FULL_RANK), and the fallback taken on the device when the threshold leaves too few candidates
    full = run(str(tmp_path), "full", store, SRX_HVG_FULL_RANK="1")
    miss = run(str(tmp_path), "miss", store, SRX_HVG_FORCE_MISS="1")
    np.testing.assert_array_equal(base["hv"], full["hv"])
    np.testing.assert_array_equal(base["hv"], miss["hv"])
    np.testing.assert_allclose(miss["evr"], base["evr"], rtol=1e-9)       # (the Gram sums differ in their last bits run to run)
    assert base["residual"] <= 1e-7 and side["residual"] <= 1e-7 and eager["residual"] <= 1e-7
    # the stored matrix: default and side stream store the same function of the same arguments
    np.testing.assert_array_equal(base["values"], side["values"])
    np.testing.assert_allclose(eager["values"], base["values"], rtol=vtol, atol=0)
    # the exact selection in both lazy orders; the eager one ranks what X holds (identical at f64 storage)
    np.testing.assert_array_equal(base["hv"], side["hv"])
    if store == 2:
        np.testing.assert_array_equal(base["hv"], eager["hv"])
    for other in (side,) + ((eager,) if np.array_equal(base["hv"], eager["hv"]) else ()):
        np.testing.assert_allclose(other["evr"], base["evr"], rtol=1e-6)
        sign = np.sign(np.sum(other["scores"] * base["scores"], axis=0))
        err = np.linalg.norm(other["scores"] * sign - base["scores"], axis=0) / np.linalg.norm(base["scores"], axis=0)
        assert err[:10].max() < 1e-5
