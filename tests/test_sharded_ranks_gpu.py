"""GPU test (-m gpu) of the ROW-SHARDED path with more than one rank: N processes, all on GPU 0 (RCCL refuses two
ranks on one device, so the sums over ranks go through srx_comm_init_host and the stdlib socket star), each
holding a contiguous nnz-balanced row range.  Every rank must arrive at the same per-gene results as the
single-rank run, and the concatenated per-cell results must equal it — for the resident pipeline (both solvers)
and for backed sessions."""
import os
import subprocess
import sys

import numpy as np
import pytest

from test_pca_gpu import TOL, col_err, synth_host

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def launch(world, mode, tmp_path):
    key = f"t{os.getpid()}_{mode}_{world}"
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "sharded_worker.py"), str(r), str(world), key, str(tmp_path), mode],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [np.load(os.path.join(tmp_path, f"rank{r}_{mode}.npz")) for r in range(world)]


@pytest.mark.parametrize("world,solver", [(2, 1), (3, 1), (2, 2)])
def test_row_sharded_ranks_match_single_rank(ctx, tmp_path, world, solver):
    import ctypes as C
    import singlerust_amd as sr
    from singlerust_amd import _ffi as F
    from singlerust_amd.memory import statistics as st
    m, _ = synth_host(17, 6000, 3000, 0.05)
    n_hvg, n_pc = 300, 12
    np.savez(os.path.join(tmp_path, "input.npz"), n_rows=m.n_rows, n_cols=m.n_cols, indptr=m.indptr, indices=m.indices,
             values=m.values, n_hvg=n_hvg, n_pc=n_pc, solver=solver, chunk=900)
    # single rank reference
    a = sr.IMAnnData.new_basic((m.n_rows, m.n_cols, m.indptr, m.indices, m.values), ctx=ctx, store=1)
    num_col, sum_col, sum_row = (st.compute_number(a, sr.Direction.Column), st.compute_sum(a, sr.Direction.Column),
                                 st.compute_sum(a, sr.Direction.Row))
    opts = F.PcaOpts(n_pc, -1, -1, -1, 0, 0, solver, 0.0, 5)
    pr = F.PipelineResult()
    F.check(F.lib().srx_pipeline(a.x().handle, 1e4, n_hvg, C.byref(opts), C.byref(pr)), ctx.handle)
    scores, comps = np.zeros((m.n_rows, n_pc)), np.zeros((n_hvg, n_pc))
    evr, mean, std, hv = np.zeros(n_pc), np.zeros(n_hvg), np.zeros(n_hvg), np.zeros(n_hvg, np.uint64)
    F.check(F.lib().srx_result_fetch(a.x().handle, F.ptr(scores), F.ptr(comps), F.ptr(evr), F.ptr(mean), F.ptr(std), F.ptr(hv)),
            ctx.handle)

    modes = ["resident"] + (["backed"] if solver == 1 else [])
    for mode in modes:
        rs = launch(world, mode, tmp_path)
        assert rs[0]["r0"] == 0 and rs[-1]["r1"] == m.n_rows and all(rs[i]["r1"] == rs[i + 1]["r0"] for i in range(world - 1))
        for r in rs:                                   # per-gene quantities: identical on every rank
            assert int(r["n_global"]) == m.n_rows
            assert np.array_equal(r["hv"], hv)
            assert np.allclose(r["mean"], mean, rtol=1e-10, atol=1e-12) and np.allclose(r["std"], std, rtol=1e-10, atol=1e-12)
            assert np.allclose(r["evr"], evr, rtol=1e-6)
            assert col_err(r["comps"], comps) < TOL
            if mode == "resident":
                assert np.array_equal(r["num_col"], num_col) and np.array_equal(r["sum_col"], sum_col)
                # compute_min_max(Column) and a Relative filter would be the shard's own answer: refused
                assert list(r["shard_local_rc"]) == [F.E_ARG, F.E_ARG]
        assert np.array_equal(np.concatenate([r["sum_row"] for r in rs]), sum_row)
        got = np.concatenate([r["scores"] for r in rs], axis=0)     # per-cell quantities: the rank's own rows
        # every rank fixes the sign of a component the same way (it is decided on the replicated k x 64 block)
        for r in rs[1:]:
            assert np.allclose(r["comps"], rs[0]["comps"], rtol=0, atol=1e-9)
        assert col_err(got, scores) < TOL


def test_a_rank_without_rows_takes_part_in_every_sum(ctx, tmp_path):
    """Three ranks, the last one EMPTY (srx_partition_rows over two ranks + an empty range): it uploads a 0-row shard, runs the
    same pipeline, contributes zeros to every sum over the ranks and arrives at the same per-gene results; its own score block is
    0 x n_pc.  (With an RCCL communicator such a rank must issue the same collectives as the others: launch_gram decides the split
    of the Gram exchange from rank-invariant data — ADVICE r3; this test covers the rest of the path on the host transport.)"""
    import ctypes as C
    import singlerust_amd as sr
    from singlerust_amd import _ffi as F
    m, _ = synth_host(23, 5000, 2500, 0.05)
    n_hvg, n_pc = 200, 8
    np.savez(os.path.join(tmp_path, "input.npz"), n_rows=m.n_rows, n_cols=m.n_cols, indptr=m.indptr, indices=m.indices,
             values=m.values, n_hvg=n_hvg, n_pc=n_pc, solver=1, chunk=900)
    a = sr.IMAnnData.new_basic((m.n_rows, m.n_cols, m.indptr, m.indices, m.values), ctx=ctx, store=1)
    opts = F.PcaOpts(n_pc, -1, -1, -1, 0, 0, 1, 0.0, 5)
    pr = F.PipelineResult()
    F.check(F.lib().srx_pipeline(a.x().handle, 1e4, n_hvg, C.byref(opts), C.byref(pr)), ctx.handle)
    scores, comps, evr, hv = np.zeros((m.n_rows, n_pc)), np.zeros((n_hvg, n_pc)), np.zeros(n_pc), np.zeros(n_hvg, np.uint64)
    F.check(F.lib().srx_result_fetch(a.x().handle, F.ptr(scores), F.ptr(comps), F.ptr(evr), None, None, F.ptr(hv)), ctx.handle)
    rs = launch(3, "resident-empty", tmp_path)
    assert rs[2]["r0"] == rs[2]["r1"] == m.n_rows and rs[2]["scores"].shape == (0, n_pc)
    for r in rs:
        assert int(r["n_global"]) == m.n_rows and np.array_equal(r["hv"], hv)
        assert np.allclose(r["evr"], evr, rtol=1e-6) and col_err(r["comps"], comps) < TOL
    assert col_err(np.concatenate([r["scores"] for r in rs], axis=0), scores) < TOL
