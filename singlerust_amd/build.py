"""Build libsrx_hip.so (and the bench/test synthetic generator) for gfx950 with hipcc.

In-tree, explicit hipcc: objects go to singlerust_amd/csrc/build/, the shared library to
singlerust_amd/lib/.  hipcc cross-compiles gfx950 code objects without a GPU.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsrx_hip.so")

SOURCES = ["ctx.hip", "rows.hip", "genes.hip", "comm.hip", "pca_form.hip", "pca_solve.hip", "pca.hip", "synth.hip", "filter.hip", "csc.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
         "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libsrx_hip.so cannot be built")


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h", ".inl"))]
    headers += [os.path.join(HERE, "..", "include", f) for f in os.listdir(os.path.join(HERE, "..", "include"))]

    def compile_one(src: str) -> str:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            cmd = [hipcc, *FLAGS, *os.environ.get("SRX_EXTRA_FLAGS", "").split(), "-c", s, "-o", o]   # (experiments: -D switches)
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        return o

    with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
