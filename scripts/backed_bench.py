"""Backed (out-of-core) mode at scale, beside the resident pipeline on the same host matrix (development helper).

  python scripts/backed_bench.py [cells] [genes] [density] [chunk_rows]

Prints: host generation time, resident route (upload + srx_pipeline), backed route (two sweeps over row tiles,
upload of a tile overlapping the kernels of the previous one), H2D-inclusive cells/s of both, and parity of the two.
"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import singlerust_amd as sr                      # noqa: E402
from singlerust_amd import _ffi as F, backed    # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
g = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000
dens = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 25_000
lib = F.lib()
ctx = sr.Context.default()
p = F.SynthParams()
lib.srx_synth_defaults(C.byref(p), 2002, n, g, dens)
t0 = time.time()
ip = np.zeros(n + 1, dtype=np.uint64)
lib.srx_synth_indptr(C.byref(p), 0, n, F.ptr(ip))
idx = np.zeros(int(ip[-1]), np.uint64)
val = np.zeros(int(ip[-1]), np.float32)
lib.srx_synth_fill_host(C.byref(p), 0, n, F.ptr(ip), F.ptr(idx), F.ptr(val))
nnz = int(ip[-1])
print(f"host matrix {n} x {g}, nnz {nnz:.3e} ({(idx.nbytes + val.nbytes) / 1e9:.2f} GB reference layout u64+f32) "
      f"generated in {time.time() - t0:.1f} s", flush=True)

opts = F.PcaOpts(50, -1, -1, -1, 0, 0, 0, 0.0, 0)


def resident():
    t = time.time()
    a = sr.IMAnnData.new_basic((n, g, ip, idx, val), ctx=ctx, store=1)
    ctx.synchronize()
    t_up = time.time() - t
    res = F.PipelineResult()
    t = time.time()
    F.check(lib.srx_pipeline(a.x().handle, 1e4, 2000, C.byref(opts), C.byref(res)), ctx.handle)
    scores = np.zeros((n, 50))
    hv = np.zeros(2000, np.uint64)
    F.check(lib.srx_result_fetch(a.x().handle, F.ptr(scores), None, None, None, None, F.ptr(hv)), ctx.handle)
    t_pipe = time.time() - t
    a.x().free()
    return t_up, t_pipe, scores, hv


for rep in range(2):
    t_up, t_pipe, s_res, hv_res = resident()
    print(f"resident: upload {t_up:.3f} s ({(idx.nbytes + val.nbytes) / 1e9 / t_up:.1f} GB/s of host data), pipeline+fetch "
          f"{t_pipe * 1e3:.1f} ms -> {n / (t_up + t_pipe):.3e} cells/s H2D-inclusive", flush=True)

x = backed.BackedCsr(ip, idx, val, g)
ad = backed.BackedAnnData(x, ctx)
for rep in range(2):
    t = time.time()
    r = backed.processing.pca_pipeline(ad, chunk, 1e4, 2000, 50, store=1)
    dt = time.time() - t
    print(f"backed  : {(n + chunk - 1) // chunk} tiles of {chunk} rows, two sweeps {dt:.3f} s "
          f"({2 * (idx.nbytes + val.nbytes) / 1e9 / dt:.1f} GB/s of host data) -> {n / dt:.3e} cells/s", flush=True)


def col_err(got, ref):
    w = 0.0
    for c in range(ref.shape[1]):
        s = 1.0 if np.dot(got[:, c], ref[:, c]) >= 0 else -1.0
        e = np.linalg.norm(got[:, c] - s * ref[:, c]) / np.linalg.norm(ref[:, c])
        w = max(w, e) if np.isfinite(e) else float("inf")
    return w


print("HVG lists identical:", bool(np.array_equal(r.selected, hv_res)), " scores rel err vs resident:", col_err(r.x_pca, s_res))
print("score column norms (resident / backed):", np.linalg.norm(s_res, axis=0)[:3], np.linalg.norm(r.x_pca, axis=0)[:3],
      " max |diff|:", float(np.max(np.abs(np.abs(r.x_pca) - np.abs(s_res)))))
