"""C++ host mirror (singlerust_amd/host/single_rust.hpp): the reference's own test_normalize_total
(src/memory/processing/mod.rs:421-481) restated in C++ against the mirror, compiled with g++ and
linked against the C-ABI library only."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "singlerust_amd", "lib")


def _compile(tmp_path):
    exe = str(tmp_path / "host_mirror_test")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror",
           os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp"), "-o", exe,
           "-L" + LIBDIR, "-lsrx_hip", "-Wl,-rpath," + LIBDIR]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return exe


def test_host_mirror_compiles_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = _compile(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2
    assert "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_host_mirror_reference_tests(tmp_path):
    exe = _compile(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout
