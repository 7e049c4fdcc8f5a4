"""Upload-rate probe: the c3 matrix in the reference layout (u64 offsets / indices, f32 values) in pageable (or pinned) host memory, timed through
srx_matrix_upload under the current SRX_UP_WORKERS.  Usage: python scripts/upload_probe.py [n_cells] [n_genes] [pinned]"""
import concurrent.futures as cf
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import singlerust_amd as sr
from singlerust_amd import _ffi as F

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_300_000
g = int(sys.argv[2]) if len(sys.argv) > 2 else 28_000
pin = len(sys.argv) > 3 and sys.argv[3] == "pinned"
lib = F.lib()
ctx = sr.Context(0)
p = F.SynthParams()
lib.srx_synth_defaults(C.byref(p), 1234, n, g, 0.03)
ip = np.zeros(n + 1, dtype=np.uint64)
lib.srx_synth_indptr(C.byref(p), 0, n, F.ptr(ip))
nnz = int(ip[-1])
if pin:
    hip = C.CDLL("libamdhip64.so")
    hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    hi_, hv_ = C.c_void_p(), C.c_void_p()
    assert hip.hipHostMalloc(C.byref(hi_), nnz * 8, 0) == 0 and hip.hipHostMalloc(C.byref(hv_), nnz * 4, 0) == 0
    idx = np.ctypeslib.as_array(C.cast(hi_, C.POINTER(C.c_uint64)), shape=(nnz,))
    val = np.ctypeslib.as_array(C.cast(hv_, C.POINTER(C.c_float)), shape=(nnz,))
else:
    idx, val = np.zeros(nnz, np.uint64), np.zeros(nnz, np.float32)


def fill(r0):
    r1 = min(n, r0 + 20_000)
    e0, e1 = int(ip[r0]), int(ip[r1])
    sub = (ip[r0:r1 + 1] - ip[r0]).astype(np.uint64)
    lib.srx_synth_fill_host(C.byref(p), r0, r1, F.ptr(sub), F.ptr(idx[e0:e1]), F.ptr(val[e0:e1]))


with cf.ThreadPoolExecutor(max_workers=64) as ex:
    list(ex.map(fill, range(0, n, 20_000)))
host_bytes = ip.nbytes + idx.nbytes + val.nbytes
ts = []
for _ in range(4):
    t0 = time.perf_counter()
    m = sr.DeviceCsr.upload(ctx, n, g, ip, idx, val, F.STORE_F32)
    ctx.synchronize()
    ts.append(time.perf_counter() - t0)
    m.free()
print(f"{'pinned' if pin else 'pageable'}, workers {os.environ.get('SRX_UP_WORKERS', 'default')}: upload s {[round(t, 4) for t in ts]}  best {host_bytes / min(ts) / 1e9:.1f} GB/s of host bytes "
      f"({host_bytes / 1e9:.1f} GB)")
