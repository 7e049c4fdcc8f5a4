"""Worker of tests/test_sharded_ranks_gpu.py: one rank of an N-rank row-sharded run (all ranks on GPU 0, sums over
ranks through the host all-reduce hook and the stdlib socket star).  Writes its results as an .npz file."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    import faulthandler
    faulthandler.enable()
    rank, world, key, out_dir, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    import singlerust_amd as sr
    from singlerust_amd import _ffi as F, backed
    from singlerust_amd.rendezvous import StarGroup
    lib = F.lib()
    z = np.load(os.path.join(out_dir, "input.npz"))
    n, g = int(z["n_rows"]), int(z["n_cols"])
    indptr, indices, values = z["indptr"], z["indices"], z["values"]
    cut = np.zeros(world + 1, dtype=np.uint64)
    if mode == "resident-empty":                # the LAST rank holds no rows (more ranks than work, a filter that emptied a shard)
        F.check(lib.srx_partition_rows(F.ptr(indptr), n, world - 1, F.ptr(cut)))
        cut[world] = n
        mode = "resident"
    else:
        F.check(lib.srx_partition_rows(F.ptr(indptr), n, world, F.ptr(cut)))
    r0, r1 = int(cut[rank]), int(cut[rank + 1])
    lo, hi = int(indptr[r0]), int(indptr[r1])
    group = StarGroup(rank, world, key=key)
    ctx = sr.Context(0)
    ctx.comm_init_host(world, rank, group.allreduce_sum_f64)
    n_hvg, n_pc = int(z["n_hvg"]), int(z["n_pc"])
    opts = F.PcaOpts(n_pc, -1, -1, -1, 0, 0, int(z["solver"]), 0.0, 5)
    res = {}
    if mode == "resident":
        a = sr.IMAnnData.new_basic((r1 - r0, g, indptr[r0:r1 + 1] - indptr[r0], indices[lo:hi], values[lo:hi]), ctx=ctx, store=1)
        F.check(lib.srx_matrix_set_shard(a.x().handle, r0), ctx.handle)
        from singlerust_amd.memory import statistics as st
        res["num_col"] = st.compute_number(a, sr.Direction.Column)
        res["sum_col"] = st.compute_sum(a, sr.Direction.Column)
        res["sum_row"] = st.compute_sum(a, sr.Direction.Row)
        # entry points whose answer would be this shard's own refuse a sharded context (srx.h, "row sharding")
        shard_local = []
        mn, mx = np.zeros(g), np.zeros(g)
        shard_local.append(lib.srx_compute_min_max(a.x().handle, 1, F.ptr(mn), F.ptr(mx)))
        out_h = C.c_void_p()
        shard_local.append(lib.srx_filter_cells(a.x().handle, F.Flex(F.FLEX_RELATIVE, 0, 0.1), F.Flex(F.FLEX_NONE, 0, 0.0),
                                                C.byref(out_h), None))
        res["shard_local_rc"] = np.array(shard_local)
        pr = F.PipelineResult()
        F.check(lib.srx_pipeline(a.x().handle, 1e4, n_hvg, C.byref(opts), C.byref(pr)), ctx.handle)
        k = int(pr.pca.k)
        scores, comps = np.zeros((r1 - r0, n_pc)), np.zeros((k, n_pc))
        evr, mean, std, hv = np.zeros(n_pc), np.zeros(k), np.zeros(k), np.zeros(k, np.uint64)
        F.check(lib.srx_result_fetch(a.x().handle, F.ptr(scores), F.ptr(comps), F.ptr(evr), F.ptr(mean), F.ptr(std), F.ptr(hv)),
                ctx.handle)
        res.update(scores=scores, comps=comps, evr=evr, mean=mean, std=std, hv=hv, n_global=int(pr.pca.n_cells_global))
    else:                                       # backed: each rank streams its own row range in tiles
        x = backed.BackedCsr(indptr[r0:r1 + 1], indices, values, g)     # window of the row offsets: no rebase
        x.indices, x.values = indices, values
        ad = backed.BackedAnnData.__new__(backed.BackedAnnData)
        ad._x, ad.ctx = x, ctx

        class Win:                               # BackedCsr.iter over the rank's window of the global arrays
            n_rows, n_cols = r1 - r0, g

            def iter(self, size):
                for s in range(r0, r1, size):
                    e = min(s + size, r1)
                    a0, a1 = int(indptr[s]), int(indptr[e])
                    yield backed.CsrChunk(indptr[s:e + 1], indices[a0:a1], values[a0:a1], g), s - r0, e - r0
        ad._x = Win()
        r = backed.processing.pca_pipeline(ad, int(z["chunk"]), 1e4, n_hvg, n_pc, store=1, seed=5)
        res.update(scores=r.x_pca, comps=r.components, evr=r.explained_variance_ratio, mean=r.mean, std=r.std,
                   hv=r.selected, n_global=int(r.info.n_cells_global), sum_row=r.row_sums)
    res["r0"], res["r1"] = r0, r1
    np.savez(os.path.join(out_dir, f"rank{rank}_{sys.argv[5]}.npz"), **res)
    group.barrier()
    group.close()         # the context outlives the matrices: both go with the process


if __name__ == "__main__":
    main()
