"""Worker of tests/test_comm_gpu.py::test_gram_exchange_overlapped_with_the_second_half (own process: SRX_GRAM_OVERLAP is read
once).  A 1-rank RCCL communicator with the split Gram launch — first half of the owners, its two row ranges of the packed
triangle summed on the communication stream under the second launch, the middle range after it — against a plain context."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
sys.path.insert(0, __file__.rsplit("/", 1)[0])


def main():
    import singlerust_amd as sr
    from singlerust_amd import _ffi
    from test_pca_gpu import col_err, synth_host
    lib = _ffi.lib()
    for n, g, k, npc in ((3000, 2500, 200, 8), (20000, 4000, 1003, 20)):      # (k = 1003: the last stripe is short)
        m, _ = synth_host(13, n, g, 0.05)
        opts = _ffi.PcaOpts(npc, -1, -1, -1, 0, 0, 1, 0.0, 7)

        def run(ctx):
            a = sr.IMAnnData.new_basic((m.n_rows, m.n_cols, m.indptr, m.indices, m.values), ctx=ctx, store=1)
            res = _ffi.PipelineResult()
            _ffi.check(lib.srx_pipeline(a.x().handle, 1e4, k, C.byref(opts), C.byref(res)), ctx.handle)
            scores, comps, evr = np.zeros((n, npc)), np.zeros((k, npc)), np.zeros(npc)
            _ffi.check(lib.srx_result_fetch(a.x().handle, _ffi.ptr(scores), _ffi.ptr(comps), _ffi.ptr(evr), None, None, None), ctx.handle)
            return scores, comps, evr
        plain = sr.Context(0)
        s0, c0, e0 = run(plain)
        comm = sr.Context(0)
        comm.comm_init(1, 0, sr.Context.comm_unique_id())
        s1, c1, e1 = run(comm)
        assert col_err(s1, s0) < 1e-6 and col_err(c1, c0) < 1e-6, (col_err(s1, s0), col_err(c1, c0))
        # the exchange really went the split way, its second half on the CU-masked stream (srx_comm_overlap_info); the plain
        # context never splits
        splits, masked = C.c_int32(-1), C.c_int32(-1)
        _ffi.check(lib.srx_comm_overlap_info(comm.handle, C.byref(splits), C.byref(masked)), comm.handle)
        assert splits.value >= 1 and masked.value == 1, (splits.value, masked.value)
        _ffi.check(lib.srx_comm_overlap_info(plain.handle, C.byref(splits), C.byref(masked)), plain.handle)
        assert splits.value == 0 and masked.value == 0
        assert np.allclose(e1, e0, rtol=1e-9)
        _ffi.check(lib.srx_comm_destroy(comm.handle), comm.handle)
        comm.close()
        plain.close()
    print("overlap ok")


if __name__ == "__main__":
    main()
