"""CPU test of bench.py's own launcher: `python bench.py --gpus 2` without a launcher environment starts two ranks; with no
GPU here each rank fails at context creation (there is no CPU fallback) and the launcher must report that as a non-zero exit
with nothing on stdout."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_propagates_rank_failure():
    import ctypes as C
    from singlerust_amd import _ffi
    n = C.c_int32(0)
    if _ffi.lib().srx_device_count(C.byref(n)) == 0 and n.value > 0:
        import pytest
        pytest.skip("a GPU is visible here")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--lean"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT, timeout=120)
    assert p.returncode != 0
    assert not p.stdout.decode().strip()
    err = p.stderr.decode()
    assert "no CPU fallback" in err and "[bench launcher] rank" in err
