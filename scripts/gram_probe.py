"""Class time of the Gram kernel inside srx_pipeline at c3 (knock-out builds: later stages may fail — ignored)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import singlerust_amd as sr
from singlerust_amd import _ffi as F
lib = F.lib(); ctx = sr.Context(0)
p = F.SynthParams(); lib.srx_synth_defaults(C.byref(p), 3003, 1_300_000, 28_000, 0.03)
h = C.c_void_p()
F.check(lib.srx_synth_generate(ctx.handle, C.byref(p), 0, 1_300_000, F.F32, F.STORE_F32, C.byref(h)), ctx.handle)
pristine = sr.DeviceCsr(ctx, h); pristine.prepare()
opts = F.PcaOpts(50, -1, -1, -1, 0, 0, 0, 0.0, 12345); res = F.PipelineResult()
ctx.prof_enable((1 << F.K_GRAM) | (1 << F.K_BUCKET))
for it in range(6):
    m = pristine.clone()
    if it == 2: ctx.prof_reset()
    rc = lib.srx_pipeline(m.handle, 1e4, 2000, C.byref(opts), C.byref(res)); ctx.synchronize(); m.free()
ms, cnt, b = ctx.prof_get(F.K_GRAM)
print(f"gram {ms / max(cnt, 1):.3f} ms ({cnt} launches, last rc {rc})")
