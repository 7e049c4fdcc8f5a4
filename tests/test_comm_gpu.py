"""GPU test (-m gpu) of the RCCL path on ONE GPU: a 1-rank communicator runs the very same
ncclAllReduce(f64, sum) calls (moments, Gram matrix / k x l block) the row-sharded runs use, so
dlopen names, signatures, enum values and stream usage are exercised without a second device.
The multi-rank arithmetic itself is covered on CPU by tests/test_sharding_gloo_cpu.py."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("solver", [1, 2])
def test_single_rank_communicator_runs_the_collectives(solver):
    import singlerust_amd as sr
    from singlerust_amd import _ffi
    from test_pca_gpu import synth_host, col_err
    lib = _ffi.lib()
    m, _ = synth_host(13, 3000, 2500, 0.05)
    opts = _ffi.PcaOpts(8, -1, -1, -1, 0, 0, solver, 0.0, 7)

    def run(ctx):
        a = sr.IMAnnData.new_basic((m.n_rows, m.n_cols, m.indptr, m.indices, m.values), ctx=ctx, store=1)
        res = _ffi.PipelineResult()
        _ffi.check(lib.srx_pipeline(a.x().handle, 1e4, 200, C.byref(opts), C.byref(res)), ctx.handle)
        scores = np.zeros((3000, 8))
        comps = np.zeros((200, 8))
        hv = np.zeros(200, np.uint64)
        _ffi.check(lib.srx_result_fetch(a.x().handle, _ffi.ptr(scores), _ffi.ptr(comps), None, None, None, _ffi.ptr(hv)),
                   ctx.handle)
        return scores, comps, hv, res

    plain = sr.Context(0)
    s0, c0, h0, _ = run(plain)
    comm = sr.Context(0)
    comm.comm_init(1, 0, sr.Context.comm_unique_id())
    s1, c1, h1, r1 = run(comm)
    assert np.array_equal(h0, h1)
    assert r1.pca.n_cells_global == 3000 and r1.pca.solver == solver
    assert col_err(s1, s0) < 1e-5 and col_err(c1, c0) < 1e-5
    _ffi.check(lib.srx_comm_destroy(comm.handle), comm.handle)
    comm.close()
    plain.close()


def test_gram_exchange_overlapped_with_the_second_half():
    """The sharded-run arrangement of the Gram solver (two launches of the stripe kernel, the first half's rows of the packed
    triangle all-reduced on the communication stream under the second launch) on a 1-rank RCCL communicator: SRX_GRAM_OVERLAP=1
    forces the split a multi-rank context takes by itself."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([sys.executable, os.path.join(here, "comm_overlap_worker.py")], env=dict(os.environ, SRX_GRAM_OVERLAP="1"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0 and b"overlap ok" in p.stdout, p.stderr.decode()[-3000:]
