"""CPU tests of the multi-GPU path's host logic, world_size 2 over gloo (no GPU needed).

What shards and what is exchanged is fixed by construction in pca.hip / genes.hip: rows are cut
by srx_partition_rows (nnz-balanced), every rank owns a contiguous row range, and the ONLY
cross-rank traffic is a sum all-reduce of (a) the packed per-gene moments [cnt | sum | sumsq | N]
and (b) the k x l block A^T Y (+ 1^T Y) of each subspace iteration (matrix-free solver) or (b') the packed upper
triangle of X_sel^T X_sel in three ranges + the 513 statistics words in front of it (Gram solver, the default).  These tests replay exactly
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


def _gram_worker(rank, world, port, out_dir):
    """The DEFAULT (Gram) solver's exchange pattern, as launch_gram / run_pca issue it (pca_form.hip, pca_solve.hip):
      1. packed gene moments, one all-reduce;
      2. (f32 entries) the value statistics that pick the stripe kernel's accumulation mode, as one-hot exponent
         histograms + a negative count, one all-reduce of 513 doubles (k_gstat_onehot / k_gstat_decode);
      3. the packed upper triangle of X_sel^T X_sel in THREE all-reduces over the ranges srx_gram_exchange_ranges gives
         (first and last rows, then the middle) — every rank issues all three with the same counts, a rank WITHOUT rows too;
      4. nothing else: C, the k x 64 iteration and the l x l algebra are replicated.
    The oracle stands in for the kernels, gloo for RCCL; rank 1 of 3 holds NO rows."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from oracle import ROW
    from singlerust_amd import _ffi
    from test_pca_gpu import synth_host

    n, g, n_hvg, n_pc = 1500, 700, 90, 8
    m, _ = synth_host(47, n, g, 0.09)
    cut2 = np.zeros(3, dtype=np.uint64)
    assert _ffi.lib().srx_partition_rows(_ffi.ptr(m.indptr), n, 2, _ffi.ptr(cut2)) == 0
    cut = [0, int(cut2[1]), int(cut2[1]), n]                      # rank 1: an empty range between the two nnz-balanced halves
    r0, r1 = cut[rank], cut[rank + 1]
    a, b = int(m.indptr[r0]), int(m.indptr[r1])
    shard = oracle.Csr(r1 - r0, g, m.indptr[r0:r1 + 1] - m.indptr[r0], m.indices[a:b], m.values[a:b])
    calls = []

    def allreduce(x):
        calls.append(int(np.size(x)))
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()

    lg = oracle.log1p_transform(oracle.normalize_total(shard, 1e4, ROW))
    cnt, s, sq = oracle.gene_moments(lg)
    packed = allreduce(np.concatenate([cnt.astype(np.float64), s, sq, [float(shard.n_rows)]]))
    gc, gs, gq, gn = packed[:g], packed[g:2 * g], packed[2 * g:3 * g], packed[3 * g]
    with np.errstate(invalid="ignore", divide="ignore"):
        var = np.where(gc > 0, gq / gc - (gs / gc) ** 2, 0.0)
    sel = oracle.select_hvg(var, n_hvg)
    sel_sorted = np.sort(sel).astype(np.int64)
    k = len(sel_sorted)
    mu = gs[sel_sorted] / gn
    sd = np.sqrt(gq[sel_sorted] / gn - mu ** 2)
    A = oracle.densify_selected(lg, sel_sorted)                  # this rank's rows of X[:, sel] (0 x k on the empty rank)
    # 2. value statistics -> one-hot exponent bins, summed, decoded: the same mode on every rank
    bins = np.zeros(513)
    nz = np.float32(A[A != 0])
    if nz.size:
        bits = np.abs(nz).view(np.uint32)
        bins[int(bits.max() >> 23)] += 1.0
        bins[256 + int(bits.min() >> 23)] += 1.0
        if (nz < 0).any():
            bins[512] += 1.0
    bins = allreduce(bins)
    emax, emin = int(np.nonzero(bins[:256])[0].max()) - 127, int(np.nonzero(bins[256:512])[0].min()) - 127
    fixed_point = bins[512] == 0 and emax - emin <= 6 and -48 < emax < 48
    # 3. the packed triangle in the library's three ranges
    iu = np.triu_indices(k)
    G = A.T @ A
    if fixed_point:                                               # the kernel's products: scaled, rounded to integers, summed exactly
        kq = 29 - 2 * emax
        Af = np.float32(A)
        G = np.zeros((k, k))
        for row in Af:
            j = np.nonzero(row)[0]
            p = np.floor(np.float32(np.float32(row[j][:, None] * np.float32(2.0 ** kq)) * row[j][None, :] + np.float32(0.5)).astype(np.float64))
            G[np.ix_(j, j)] += p * 2.0 ** -kq
    Pk = np.ascontiguousarray(G[iu])
    offs = np.zeros(4, dtype=np.uint64)
    assert _ffi.lib().srx_gram_exchange_ranges(k, _ffi.ptr(offs)) == 0
    o = [int(v) for v in offs]
    assert o[0] == 0 and o[3] == k * (k + 1) // 2 and o[0] < o[1] <= o[2] < o[3]
    Pk[o[0]:o[1]] = allreduce(Pk[o[0]:o[1]])
    Pk[o[2]:o[3]] = allreduce(Pk[o[2]:o[3]])
    Pk[o[1]:o[2]] = allreduce(Pk[o[1]:o[2]])
    # 4. replicated: C = D (G - N mu mu^T) D, its leading eigenpairs (stand-in for the k x 64 iteration), the shard's scores
    Gf = np.zeros((k, k))
    Gf[iu] = Pk
    Gf = Gf + np.triu(Gf, 1).T
    Cm = (Gf - gn * np.outer(mu, mu)) / np.outer(sd, sd)
    w, v = np.linalg.eigh(Cm)
    order = np.argsort(w)[::-1][:n_pc]
    V = v[:, order]
    V = V * np.sign(V[np.abs(V).argmax(axis=0), np.arange(n_pc)])[None, :]
    scores = (A - mu[None, :]) / sd[None, :] @ V if A.shape[0] else np.zeros((0, n_pc))
    np.savez(os.path.join(out_dir, f"grank{rank}.npz"), scores=scores, V=V, th=w[order], sel=sel, calls=np.array(calls),
             mode=int(fixed_point), evr=w[order] / np.trace(Cm), rows=r1 - r0)
    dist.barrier()
    dist.destroy_process_group()


def test_gram_solver_exchange_in_three_ranges_with_an_empty_rank(tmp_path):
    import torch.multiprocessing as mp
    import oracle
    from oracle import COLUMN, ROW, pca_oracle
    from test_pca_gpu import synth_host, col_err
    world, port = 3, 31500 + (os.getpid() % 2000)
    mp.spawn(_gram_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"grank{r}.npz") for r in range(world)]
    assert parts[1]["rows"] == 0 and parts[0]["rows"] > 0 and parts[2]["rows"] > 0
    m, _ = synth_host(47, 1500, 700, 0.09)
    lg = oracle.log1p_transform(oracle.normalize_total(m, 1e4, ROW))
    sel = oracle.select_hvg(oracle.compute_variance(lg, COLUMN), 90)
    k = 90
    for p in parts:
        assert np.array_equal(p["sel"], sel)
        # the same five collectives with the same counts on every rank, the empty one included
        assert np.array_equal(p["calls"], parts[0]["calls"]) and len(p["calls"]) == 5
        assert p["calls"][0] == 3 * 700 + 1 and p["calls"][1] == 513 and p["calls"][2:].sum() == k * (k + 1) // 2
        assert p["mode"] == 1                                   # normalised + log1p'd counts: fixed point, on EVERY rank
        np.testing.assert_array_equal(p["V"], parts[0]["V"])    # replicated state: identical to the bit
    want_scores, want_comps, want_evr, *_ = pca_oracle.pca_inplace(lg, 8, None, None, np.sort(sel))
    got = np.vstack([p["scores"] for p in parts])
    assert got.shape == want_scores.shape
    # (the fixed-point products carry the f32 rounding of the stored values: the 1e-5 bar of the f32 storage)
    assert col_err(got, want_scores) < 1e-5 and col_err(parts[0]["V"], want_comps) < 1e-5
    np.testing.assert_allclose(parts[0]["evr"], want_evr, rtol=1e-5)


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
