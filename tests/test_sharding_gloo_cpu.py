"""CPU tests of the multi-GPU path's host logic, world_size 2 over gloo (no GPU needed).

What shards and what is exchanged is fixed by construction in pca.hip / genes.hip: rows are cut
by srx_partition_rows (nnz-balanced), every rank owns a contiguous row range, and the ONLY
cross-rank traffic is a sum all-reduce of (a) the packed per-gene moments [cnt | sum | sumsq | N]
and (b) the k x l block A^T Y (+ 1^T Y) of each subspace iteration.  These tests replay exactly
that exchange pattern with the CPU oracle standing in for the per-rank kernels and
torch.distributed(gloo).all_reduce standing in for ncclAllReduce, and check that the sharded
result equals the unsharded one and the exact-SVD oracle.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _subspace(local_rows_dense, mu, dinv, n_pc, l, allreduce, iters=400, tol=1e-11, seed=0):
    """The driver of pca.hip in numpy: W' = D (A^T (A D W - 1 c^T) - mu (1^T Y)), Rayleigh-Ritz,
    CholeskyQR; `allreduce` sums the k x l block (+ 1^T Y) across ranks."""
    A = local_rows_dense
    k = A.shape[1]
    W = np.linalg.qr(np.random.default_rng(seed).standard_normal((k, l)))[0]
    for _ in range(iters):
        P = dinv[:, None] * W
        Y = A @ P - (mu @ P)[None, :]
        T = np.concatenate([A.T @ Y, Y.sum(0)[None, :]], axis=0)
        T = allreduce(T)                                   # the one exchange per iteration
        Wp = dinv[:, None] * (T[:-1] - np.outer(mu, T[-1]))
        H = W.T @ Wp
        th, U = np.linalg.eigh((H + H.T) / 2)
        th, U = th[::-1], U[:, ::-1]
        R = Wp @ U - (W @ U) * th[None, :]
        res = np.max(np.linalg.norm(R[:, :n_pc], axis=0) / th[:n_pc])
        V = W @ U[:, :n_pc]
        if res < tol:
            break
        Rc = np.linalg.cholesky(Wp.T @ Wp).T
        W = Wp @ np.linalg.inv(Rc)
    P = dinv[:, None] * V
    return A @ P - (mu @ P)[None, :], V, th[:n_pc]


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from oracle import COLUMN, ROW, pca_oracle
    from singlerust_amd import _ffi
    from test_pca_gpu import synth_host                    # host generator only (no GPU touched)
    import ctypes as C

    n, g, n_hvg, n_pc, l = 1200, 900, 80, 6, 32
    m, _ = synth_host(31, n, g, 0.08)
    cut = np.zeros(world + 1, dtype=np.uint64)
    assert _ffi.lib().srx_partition_rows(_ffi.ptr(m.indptr), n, world, _ffi.ptr(cut)) == 0
    r0, r1 = int(cut[rank]), int(cut[rank + 1])
    a, b = int(m.indptr[r0]), int(m.indptr[r1])
    shard = oracle.Csr(r1 - r0, g, m.indptr[r0:r1 + 1] - m.indptr[r0], m.indices[a:b], m.values[a:b])

    def allreduce(x):
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()

    # per-cell stages are shard-local: no exchange
    lg = oracle.log1p_transform(oracle.normalize_total(shard, 1e4, ROW))
    # (a) one all-reduce of the packed moments [cnt | sum | sumsq | N]
    cnt, s, sq = oracle.gene_moments(lg)
    packed = allreduce(np.concatenate([cnt.astype(np.float64), s, sq, [float(shard.n_rows)]]))
    gc, gs, gq, gn = packed[:g], packed[g:2 * g], packed[2 * g:3 * g], packed[3 * g]
    with np.errstate(invalid="ignore", divide="ignore"):
        var = np.where(gc > 0, gq / gc - (gs / gc) ** 2, 0.0)
    sel = oracle.select_hvg(var, n_hvg)                    # identical on every rank
    sel_sorted = np.sort(sel)
    mu = gs[sel_sorted.astype(np.int64)] / gn
    sd = np.sqrt(gq[sel_sorted.astype(np.int64)] / gn - mu ** 2)
    dense_local = oracle.densify_selected(lg, sel_sorted)
    # (b) subspace iteration with the k x l all-reduce
    scores, V, th = _subspace(dense_local, mu, 1.0 / sd, n_pc, l, allreduce)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), scores=scores, V=V, th=th, sel=sel, r0=r0, r1=r1,
             var=var, n=gn)
    dist.barrier()
    dist.destroy_process_group()


def test_partition_and_sharded_pipeline_equal_unsharded(tmp_path):
    import torch.multiprocessing as mp
    import oracle
    from oracle import COLUMN, ROW, pca_oracle
    from test_pca_gpu import synth_host, col_err
    world, port = 2, 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    m, _ = synth_host(31, 1200, 900, 0.08)
    lg = oracle.log1p_transform(oracle.normalize_total(m, 1e4, ROW))
    # global moments / HVG are identical on both ranks and equal the unsharded oracle's
    var = oracle.compute_variance(lg, COLUMN)
    for p in parts:
        np.testing.assert_allclose(p["var"], var, rtol=1e-10, atol=1e-12)
        assert p["n"] == 1200
        assert np.array_equal(p["sel"], parts[0]["sel"])
    assert np.array_equal(parts[0]["sel"], oracle.select_hvg(var, 80))
    assert parts[0]["r0"] == 0 and parts[0]["r1"] == parts[1]["r0"] and parts[1]["r1"] == 1200
    # scores stay row-sharded; stacked they equal the exact-SVD oracle (up to sign)
    sel_sorted = np.sort(parts[0]["sel"])
    want_scores, want_comps, *_ = pca_oracle.pca_inplace(lg, 6, None, None, sel_sorted)
    got = np.vstack([p["scores"] for p in parts])
    assert col_err(got, want_scores) < 1e-8
    for p in parts:
        assert col_err(p["V"], want_comps) < 1e-8
        np.testing.assert_allclose(p["V"], parts[0]["V"], atol=1e-12)      # replicated k-side state


def test_partition_rows_edge_cases():
    from singlerust_amd import _ffi
    lib = _ffi.lib()
    indptr = np.array([0, 0, 0, 10, 10, 20], dtype=np.uint64)      # empty rows at the front
    cut = np.zeros(4, dtype=np.uint64)
    assert lib.srx_partition_rows(_ffi.ptr(indptr), 5, 3, _ffi.ptr(cut)) == 0
    assert cut[0] == 0 and cut[-1] == 5 and np.all(np.diff(cut.astype(np.int64)) >= 0)
    empty = np.zeros(1, dtype=np.uint64)
    cut2 = np.zeros(3, dtype=np.uint64)
    assert lib.srx_partition_rows(_ffi.ptr(empty), 0, 2, _ffi.ptr(cut2)) == 0
    assert cut2.tolist() == [0, 0, 0]


def _star_worker(rank, world, key, q):
    from singlerust_amd.rendezvous import StarGroup
    g = StarGroup(rank, world, key=key, timeout=60)
    data = g.broadcast_bytes(bytes(range(128)) if rank == 0 else None, 128)
    g.barrier()
    m = g.allreduce_max(float(rank) * 1.5)
    g.barrier()
    g.close()
    q.put((rank, data == bytes(range(128)), m))


def test_star_rendezvous_three_ranks():
    """bench.py's handshake (RCCL id broadcast, barrier, max over ranks) without torch."""
    import multiprocessing as mp
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    key = f"test_{os.getpid()}"
    ps = [ctxm.Process(target=_star_worker, args=(r, 3, key, q)) for r in range(3)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=60) for _ in ps)
    for p in ps:
        p.join(30)
        assert p.exitcode == 0
    assert [o[0] for o in out] == [0, 1, 2] and all(o[1] for o in out) and all(o[2] == 3.0 for o in out)
