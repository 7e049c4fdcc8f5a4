"""Pipeline sweep over generator parameters (number of cell types, marker genes, library-size spread, density), HVG counts
and component counts, looking for failures — development helper."""
import ctypes as C
import itertools
import sys
import time

sys.path.insert(0, ".")
import singlerust_amd as sr
from singlerust_amd import _ffi as F

lib = F.lib()
ctx = sr.Context.default()
fails = 0
t0 = time.time()
cases = []
for n_types, markers, density, hvg, npc, store in itertools.product((1, 4, 30), (20, 300), (0.01, 0.06), (200, 1000, 3000), (10, 50), (F.STORE_F32, F.STORE_F64)):
    cases.append((30000, 12000, n_types, markers, density, hvg, npc, store, 0))
cases += [(30000, 12000, 8, 100, 0.05, 1000, 50, F.STORE_F32, 2), (30000, 12000, 8, 100, 0.05, 9000, 30, F.STORE_F32, 0),
          (30000, 12000, 8, 100, 0.05, 9000, 30, F.STORE_F32, 1), (2000, 12000, 8, 100, 0.05, 1000, 50, F.STORE_F32, 0),
          (300, 12000, 8, 100, 0.05, 500, 50, F.STORE_F32, 0), (70, 3000, 3, 50, 0.1, 200, 50, F.STORE_F64, 0)]
for (n, g, n_types, markers, density, hvg, npc, store, solver) in cases:
    p = F.SynthParams()
    lib.srx_synth_defaults(C.byref(p), 77 + n_types + markers, n, g, density)
    p.n_types, p.marker_genes = n_types, markers
    h = C.c_void_p()
    if lib.srx_synth_generate(ctx.handle, C.byref(p), 0, n, F.F32 if store == F.STORE_F32 else F.F64, store, C.byref(h)) != 0:
        print(f"(generator refuses n={n} g={g} types={n_types} markers={markers})")
        continue
    opts = F.PcaOpts(npc, -1, -1, -1, 0, 0, solver, 0.0, 0)
    res = F.PipelineResult()
    ctx.synchronize()
    t1 = time.time()
    rc = lib.srx_pipeline(h, 1e4, hvg, C.byref(opts), C.byref(res))
    ms = (time.time() - t1) * 1e3
    msg = "" if rc == 0 else (lib.srx_last_error(ctx.handle) or b"").decode()[:90]
    if rc != 0:
        fails += 1
    if rc != 0 or res.pca.n_iter > 150:
        print(f"n={n} types={n_types} markers={markers} dens={density} hvg={hvg} npc={npc} store={store} solver={solver}: rc={rc} "
              f"iters={res.pca.n_iter} resid={res.pca.residual:.2e} {ms:.1f} ms {msg}", flush=True)
    lib.srx_matrix_free(h)
print(f"{len(cases)} cases, {fails} failures, {time.time() - t0:.1f} s")
