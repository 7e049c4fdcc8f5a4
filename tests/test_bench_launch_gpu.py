"""GPU test (-m gpu) of bench.py's N > 1 launch path: two ranks with the environment torch.distributed.run gives them
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), both on GPU 0 (SRX_BENCH_DEVICE), the sums over ranks
through the host all-reduce hook (SRX_BENCH_COLLECTIVE=host — RCCL refuses two ranks on one device).  Rank 0 must
print exactly one JSON line describing the whole job."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_line():
    world, cells = 2, 60000
    env = dict(os.environ, WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(20000 + os.getpid() % 20000),
               SRX_BENCH_DEVICE="0", SRX_BENCH_COLLECTIVE="host", TORCHELASTIC_RUN_ID=f"t{os.getpid()}")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--cells", str(cells), "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline"]
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              cwd=ROOT) for r in range(world)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e.decode()[-3000:]
    lines0 = [ln for ln in outs[0][0].decode().splitlines() if ln.strip()]
    assert len(lines0) == 1, lines0                        # one JSON line, nothing else on rank 0's stdout
    assert not outs[1][0].decode().strip()                 # and nothing at all on the other rank's
    assert len(lines0[0]) < 8000                           # a record the driver parses, not a document
    d = json.loads(lines0[0])
    # N > 1 defaults to STRONG scaling (BASELINE.json configs[3]: the same cells row-sharded over the GPUs) and carries a
    # `weak` block measured after it
    assert d["n_gpus"] == world and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert d["unit"] == "cells/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    c = d["config"]
    assert c["cells_global"] == cells and c["parallelism"].startswith(f"row-shard x{world}") and c["collective"] == "host-star"
    # whole-job value: all ranks' cells over the max-over-ranks time of the timed steps
    assert abs(d["value"] - cells / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    w = d["weak"]
    assert w["scaling"] == "weak" and w["cells_global"] == world * cells
    assert abs(w["value"] - world * cells / (w["ms_per_step"] * 1e-3)) <= 1e-6 * w["value"]
    assert d["unattributed_ms_per_step"] is not None
    assert c["pca_residual"] < 1e-6 and d["roofline"]["frac"] > 0 and d["roofline"]["bound"] == "hbm"


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_self_launch_two_ranks():
    """`python bench.py --gpus 2` typed as it stands (no launcher environment): bench.py starts its own two ranks and still
    prints exactly one JSON line; the line says what the sums went through and how many ranks the communicator saw."""
    cells = 60000
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cells", str(cells), "--steps", "2", "--warmup", "1",
           "--lean"]
    p = subprocess.run(cmd, env=_clean_env(SRX_BENCH_DEVICE="0", SRX_BENCH_COLLECTIVE="host"), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, cwd=ROOT, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    assert len(lines[0]) < 8000
    d = json.loads(lines[0])
    c = d["config"]
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and c["cells_global"] == cells
    assert c["collective"] == "host-star" and c["kind"] == "host" and c["n_ranks_seen"] == 2 and c["launcher"] == "self"
    assert c["pca_residual"] < 1e-6


def test_host_fallback_is_refused_unless_asked_for():
    """Two ranks on ONE device: RCCL cannot come up.  Without SRX_BENCH_COLLECTIVE=host the launch must fail (non-zero exit,
    no JSON line) instead of quietly producing a scaling number over the host star."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cells", "20000", "--steps", "1", "--warmup", "0", "--lean"]
    p = subprocess.run(cmd, env=_clean_env(SRX_BENCH_DEVICE="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT,
                       timeout=600)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.decode().splitlines() if ln.strip().startswith("{")]
    assert "SRX_BENCH_COLLECTIVE=host" in p.stderr.decode()


def test_backed_run_across_two_ranks():
    """configs[4]'s form on more than one rank: `bench.py --gpus 2 --backed` — every rank streams its own row range of the
    host matrix through its own backed session, the moments and the Gram triangle are summed over the ranks — against the
    single-rank backed run of the same matrix (same selection, same residual)."""
    cells = 60000
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c2", "--cells", str(cells), "--backed", "--tile-rows", "8000",
            "--steps", "1"]
    env = _clean_env(SRX_BENCH_DEVICE="0", SRX_BENCH_COLLECTIVE="host")
    p1 = subprocess.run(base + ["--gpus", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT, timeout=600)
    assert p1.returncode == 0, p1.stderr.decode()[-3000:]
    p2 = subprocess.run(base + ["--gpus", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT, timeout=600)
    assert p2.returncode == 0, p2.stderr.decode()[-3000:]
    d1 = json.loads([ln for ln in p1.stdout.decode().splitlines() if ln.strip()][-1])
    lines2 = [ln for ln in p2.stdout.decode().splitlines() if ln.strip()]
    assert len(lines2) == 1 and len(lines2[0]) < 8000
    d2 = json.loads(lines2[0])
    assert d2["n_gpus"] == 2 and d2["config"]["cells_global"] == cells and d2["config"]["n_ranks_seen"] == 2
    r1, r2 = d1["runs"][0], d2["runs"][0]
    assert r2["cells_global_seen"] == cells == r1["cells_global_seen"]
    assert r1["residual"] < 1e-6 and r2["residual"] < 1e-6
    assert abs(d2["value"] - cells / (d2["ms_per_step"] * 1e-3)) <= 1e-6 * d2["value"]


def test_default_command_prints_a_bounded_record():
    """The driver's own command shape (one GPU, no --lean, every side block and the CPU baselines on — at a reduced cell count
    so that the test stays short): ONE stdout line under 8000 bytes with roofline and cpu_baseline in it; the full document in
    bench_full.json."""
    cells = 60000
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--cells", str(cells), "--steps", "3", "--warmup", "1",
           "--cpu-sample-cells", "2000"]
    p = subprocess.run(cmd, env=_clean_env(SRX_BENCH_NO_C5="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT, timeout=1200)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    assert len(lines[0]) < 8000, len(lines[0])
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "roofline_next",
              "roofline_spmm", "cpu_baseline", "gpu_over_cpu"):
        assert k in d, k
    assert d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    assert "note" not in lines[0]
    with open(os.path.join(ROOT, "bench_full.json")) as fh:
        full = json.loads(fh.read())
    assert full["value"] == d["value"] and "kernels" in full and "strong_scaling_budget" in full
