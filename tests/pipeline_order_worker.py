"""Worker of tests/test_pipeline_orders_gpu.py: one srx_pipeline call on a small synthetic matrix under whatever SRX_* order
switches the environment holds (they are read once per process), results to an .npz."""
import ctypes as C
import sys

import numpy as np


def main(out, store):
    sys.path.insert(0, __file__.rsplit("/", 2)[0])
    import singlerust_amd as sr
    from singlerust_amd import _ffi
    from tests.test_pca_gpu import adata_of, synth_host
    ctx = sr.Context()
    n, g, hvg, npc = 6000, 3000, 400, 20
    m, _ = synth_host(123, n, g, 0.06)
    a = adata_of(m, ctx, store)
    opts = _ffi.PcaOpts(npc, -1, -1, -1, 0, 0, 0, 0.0, 11)
    res = _ffi.PipelineResult()
    _ffi.check(_ffi.lib().srx_pipeline(a.x().handle, 1e4, hvg, C.byref(opts), C.byref(res)), ctx.handle)
    scores, comps = np.zeros((n, npc)), np.zeros((hvg, npc))
    evr, hv = np.zeros(npc), np.zeros(hvg, np.uint64)
    _ffi.check(_ffi.lib().srx_result_fetch(a.x().handle, _ffi.ptr(scores), _ffi.ptr(comps), _ffi.ptr(evr), None, None, _ffi.ptr(hv)),
               ctx.handle)
    np.savez(out, scores=scores, comps=comps, evr=evr, hv=hv, values=np.asarray(a.x().values(np.float64)), residual=res.pca.residual)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
