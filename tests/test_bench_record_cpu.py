"""CPU test of bench.py's stdout record: whatever the full document holds, the one line the driver parses stays under
bench.MAX_LINE_BYTES (a 20.7 KB line came back `parsed: null`), carries the contract's fields and no prose blocks."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _worst_case():
    long = "x" * 5000
    prof = {name: {"launches": 20, "avg_ms": 1.234567890123, "alg_bytes_per_launch": 1.2345678901e9, "aux_bytes_per_launch": 1e9,
                   "GBps": 1234.56789, "frac_of_peak": 0.123456789, "hbm_traffic_per_launch": 9.87654321e9, "note": long}
            for name in bench.KERNEL_SYMBOL}
    roof = {"bound": "hbm", "kernel": long, "achieved": 1234.56789, "peak": 8000.0, "unit": "GB/s", "frac": 0.15432, "traffic": 1.6e10,
            "launches": 20, "avg_ms": 2.4609123, "alg_bytes_per_launch": 7.7428e8, "aux_bytes_per_launch": 1e9,
            "other_bounds": {"a": {"note": long}}, "note": long, "stage": {"name": long, "classes": {k: 1.0 for k in prof}}}
    big = {"note": long, "runs": [{"total_s": 1.0, "note": long}] * 10, "config": {"workload": long}}
    return {
        "metric": "cells/sec end-to-end normalise->HVG->50-PC PCA; SpMM achieved HBM GB/s vs peak", "value": 134030123.456789,
        "unit": "cells/s", "n_gpus": 8, "steps": 20, "warmup": 5, "ms_per_step": 9.69912345678, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": long, "cells_global": 1300000, "genes": 28000, "nnz_rank0": 1092174992, "hvg": 2000, "n_pc": 50,
                   "panel_width": 64, "parallelism": "row-shard x8 (nnz-balanced)", "collective": "rccl", "kind": "rccl",
                   "rccl_version": 22204, "n_ranks": 8, "n_ranks_seen": 8, "launcher": "torch.distributed.run / external",
                   "gram_exchange_split_launches": 3, "gram_exchange_cu_masked": True, "nnz_hvg_compacted_rank0": 93484474,
                   "subspace_iterations": [6] * 200, "pca_residual": 1.0454e-08, "pca_solver": "gram", "hvg_selection": long,
                   "cold_step_ms": 11.939, "prepare_ms": 2.3235, "f64_storage_ms_per_step": 16.091,
                   "incl_h2d_cells_per_s": 7793200.0, "incl_h2d_pageable_cells_per_s": 7046200.0,
                   "skewed_genes_ms_per_step": 17.761, "hard_spectrum_ms_per_step": 4.3061, "gram_formation_ms_per_step": 4.6015,
                   "iterate_ms_per_step": 0.83206, "predicted_speedup_8_gpus": 3.8606, "shard_step_ms_at_8_ranks": 2.3414,
                   "scaling_note": long},
        "roofline": roof, "roofline_next": dict(roof), "roofline_spmm": dict(roof), "kernels": prof,
        "kernel_ms_per_step": {k: 1.23456789 for k in prof}, "unattributed_ms_per_step": 0.123456789,
        "class_timers": big, "step_roofline": {"frac_of_peak": 0.29441, "note": long}, "stage_ms_per_step": {"pca": 5.0},
        "weak": {"scaling": "weak", "cells_global": 10400000, "steps": 5, "ms_per_step": 10.1, "value": 1e9, "unit": "cells/s",
                 "subspace_iterations": [6] * 50},
        "f64_storage": big, "cold_step": {"value": 108880000.0, "ms": 11.9, "note": long}, "roofline_spmm_iter": big,
        "skewed_genes": big, "hard_spectrum": big, "strong_scaling_budget": big, "incl_h2d": big, "incl_h2d_pageable": big,
        "cpu_baseline": {"value": 136310.0, "unit": "cells/s", "cores": 128, "kind": "port", "cpu_model": "AMD EPYC 9575F 64-Core Processor",
                         "seconds": 9.537, "sample": long, "host_cores": 256},
        "cpu_baseline_reference_faithful": big, "cpu_baseline_c1_serial": big, "c5_backed": big, "gpu_over_cpu": 983.3,
    }


def test_record_is_bounded_and_complete():
    full = _worst_case()
    rec = bench.compact_record(full)
    line = json.dumps(rec)
    assert len(line) <= bench.MAX_LINE_BYTES <= 6000, len(line)
    assert "\n" not in line
    back = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "roofline_next", "roofline_spmm", "cpu_baseline", "gpu_over_cpu"):
        assert k in back, k
    # the contract's numbers are not rounded
    assert back["value"] == full["value"] and back["ms_per_step"] == full["ms_per_step"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in back["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in back["cpu_baseline"], k
    assert len(back["cpu_baseline"]["sample"]) <= 160
    assert isinstance(back["config"]["workload"], str) and "model" not in back["config"]
    # no prose blocks, no nested documents
    assert "note" not in json.dumps(back)
    for v in back["config"].values():
        assert not isinstance(v, (dict, list))


def test_record_of_a_lean_run_has_no_empty_blocks():
    full = {k: v for k, v in _worst_case().items() if k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                                           "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                                                           "roofline", "kernel_ms_per_step")}
    rec = bench.compact_record(full)
    assert "cpu_baseline" not in rec and "weak" not in rec and "roofline" in rec
    assert len(json.dumps(rec)) <= bench.MAX_LINE_BYTES
