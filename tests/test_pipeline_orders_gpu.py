"""The device-side ranking of HighlyVariable(n) has three routes (environment switches read once per process, so each runs in
its own process): candidates above a sampled threshold (default), every gene ranked (SRX_HVG_FULL_RANK), and the fallback
taken ON THE DEVICE when the threshold leaves too few candidates (SRX_HVG_FORCE_MISS forces it).  All three must give the same
selection in the same order, the same stored matrix and the same principal components."""
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
@pytest.mark.parametrize("store", [1, 2])
def test_selection_routes_agree(tmp_path, store):
    base = run(str(tmp_path), "default", store)
    full = run(str(tmp_path), "full", store, SRX_HVG_FULL_RANK="1")
    miss = run(str(tmp_path), "miss", store, SRX_HVG_FORCE_MISS="1")
    assert base["residual"] <= 1e-7
    for other in (full, miss):
        np.testing.assert_array_equal(base["hv"], other["hv"])
        np.testing.assert_array_equal(base["values"], other["values"])      # the moments pass stores the same function of the same arguments
        np.testing.assert_allclose(other["evr"], base["evr"], rtol=1e-9)    # (the Gram sums differ in their last bits run to run)
        sign = np.sign(np.sum(other["scores"] * base["scores"], axis=0))
        err = np.linalg.norm(other["scores"] * sign - base["scores"], axis=0) / np.linalg.norm(base["scores"], axis=0)
        assert err[:10].max() < 1e-5
