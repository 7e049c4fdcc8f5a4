#!/usr/bin/env python3
"""bench.py — cells/s of the hot path  normalize_total(1e4, Row) -> log1p -> gene moments ->
HVG(2000) -> 50-PC PCA  on synthetic CSR resident in HBM, plus the SpMM's achieved HBM GB/s.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one srx_pipeline() call over one fresh copy of the raw count matrix (the pipeline
normalises X in place, so every step gets its own copy, cloned before the timed region).
Workload at N=1: BASELINE.json configs[2], the configuration the north_star targets are
quoted on (1.3M x 28k, ~3 % nnz — it fits one GPU); N>1: weak scaling, every rank owns a
1.3M-cell row shard of an (N x 1.3M)-cell matrix, gene moments and the k x l blocks are
all-reduced over RCCL.  One JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (cells per GPU, genes, density, seed)       SURVEY.md §8 table
    "c2": (100_000, 20_000, 0.05, 2002),
    "c3": (1_300_000, 28_000, 0.03, 3003),
    # config 5's total size on ONE GPU (48 GB of CSR in HBM: no out-of-core tiling needed on 288 GB); a
    # large-offset (nnz > 2^32) correctness / scale check, not the headline
    "c5": (10_000_000, 30_000, 0.02, 5005),
}
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


KERNEL_SYMBOL = {"normalize_log1p": "k_row_pass<float,NORM,LOG>", "gene_moments": "k_gene_moments<float>",
                 "hvg_compact": "k_tcount + k_tfill (+ scans)", "spmm_fwd": "k_spmm_fwd (CSR x 64-col panel)",
                 "spmm_t": "k_spmm_t", "gram_sparse": "k_gram_sparse<float> (+ k_gram_reduce)",
                 "dense_apply": "k_dense_apply"}
ROOF_NOTE = {
    "gram_sparse": "algorithmic bytes = HVG-compacted matrix (8-byte records + tile row pointers) read once + G "
                   "written once. Not an HBM-bound kernel: one f64 LDS atomic per product (3.4e9 per launch at c3) "
                   "plus two staged LDS reads and ~16 VALU instructions per 64-lane pass; the LDS pipe and VALU issue "
                   "are each 50-65 % busy (profiles/r01_pmc_gram_v3.md), random-address f64 LDS atomics alone would "
                   "take 2.2 ms at the 2.5 lanes/clk/CU measured by bench_micro/lds_atomic_banks.hip",
    "spmm_fwd": "algorithmic bytes per SURVEY.md 8(d): nnz_w*(4+4) + (n_t*N+1)*8 + k*64*4 + the output, which for this "
                "launch (the transform) is the N x n_pc f64 score matrix written by the SpMM itself",
}


def load_traffic(config):
    """HBM bytes per pipeline step from the committed PMC passes (profiles/make_traffic.py), per bench kernel class."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"r01_traffic_{config}.json")
    try:
        with open(path) as fh:
            t = json.load(fh)["kernels"]
    except (OSError, ValueError, KeyError):
        return {}
    cls = {"normalize_log1p": ("k_row_pass",), "gene_moments": ("k_gene_moments",), "spmm_fwd": ("k_spmm_fwd",),
           "spmm_t": ("k_spmm_t",), "gram_sparse": ("k_gram_sparse", "k_gram_reduce"),
           "dense_apply": ("k_dense_apply",),
           "hvg_compact": ("k_tcount", "k_tfill", "k_scan_block_sums", "k_scan_serial", "k_scan_apply", "k_seglen")}
    out = {}
    for name, subs in cls.items():
        tot, hit = 0.0, False
        for sym, d in t.items():
            if any(sym.startswith(s_) or (" " + s_) in sym or ("srx::" + s_) in sym for s_ in subs):
                tot += d["hbm_bytes"] * d["launches_per_step"]      # bytes per pipeline step
                hit = True
        if hit:
            out[name] = tot
    return out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--cells", type=int, default=0, help="override cells per GPU")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="weak (default): the config's cell count PER GPU; strong: the config's cell count in total, "
                         "row-sharded over the GPUs (BASELINE.json configs[3] read literally)")
    ap.add_argument("--hvg", type=int, default=2000)
    ap.add_argument("--npc", type=int, default=50)
    ap.add_argument("--target-sum", type=float, default=1e4)
    ap.add_argument("--solver", type=int, default=0, help="0 auto, 1 explicit Gram, 2 matrix-free SpMM iteration")
    ap.add_argument("--storage", default="f32", choices=("f32", "f64"),
                    help="value storage in HBM: f32 (default; exact for count data, meets the 1e-5 bar) or f64")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-cells", type=int, default=24000)
    ap.add_argument("--max-copies-gb", type=float, default=180.0)
    return ap.parse_args()


def cpu_baseline(F, params, genes, n_cells, hvg, npc, target):
    """The oracle (a port of the reference's serial loops + exact-SVD PCA) timed on this box's
    host cores over a bounded sample of the SAME synthetic workload (first n_cells rows)."""
    import numpy as np
    import oracle
    from oracle import pca_oracle
    lib = F.lib()
    ip = np.zeros(n_cells + 1, dtype=np.uint64)
    lib.srx_synth_indptr(C.byref(params), 0, n_cells, F.ptr(ip))
    idx = np.zeros(int(ip[-1]), np.uint64)
    val = np.zeros(int(ip[-1]), np.float32)
    lib.srx_synth_fill_host(C.byref(params), 0, n_cells, F.ptr(ip), F.ptr(idx), F.ptr(val))
    m = oracle.Csr(n_cells, genes, ip, idx, val)
    oracle.lib()
    t0 = time.perf_counter()
    n = oracle.normalize_total(m, target, oracle.ROW)
    lg = oracle.log1p_transform(n)
    t1 = time.perf_counter()
    sel = pca_oracle.select_features_hvg(lg, hvg)
    t2 = time.perf_counter()
    pca_oracle.pca_inplace(lg, npc, None, None, sel)
    t3 = time.perf_counter()
    total = t3 - t0
    return {
        "value": n_cells / total, "unit": "cells/s", "cores": os.cpu_count(), "kind": "port",
        "sample": f"first {n_cells} cells of the same synthetic matrix ({int(ip[-1])} nnz): serial C restatement of "
                  f"normalize_total+log1p ({t1 - t0:.2f}s) and nz-variance HVG({hvg}) ({t2 - t1:.2f}s) on 1 core, "
                  f"densify + full-SVD PCA via numpy/LAPACK on all cores ({t3 - t2:.2f}s); "
                  "exact-SVD cost is linear in cells at fixed k, so cells/s carries to the full size",
        "seconds": total,
    }


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N with N > 1 must be launched through torch.distributed.run (one rank per GPU)")
        a.gpus = world
    import numpy as np
    import singlerust_amd as sr
    from singlerust_amd import _ffi as F
    lib = F.lib()

    # One process per GPU.  The handshake (128-byte RCCL id, barriers, max over ranks) uses a
    # stdlib socket star, not torch.distributed: the torch wheel bundles its own HIP runtime and RCCL,
    # and once it is imported into this process ncclCommInitRank of the system RCCL that
    # libsrx_hip.so uses fails; nothing on the data path needs torch.  Device synchronisation goes
    # through the library (hipStreamSynchronize on the stream every kernel of the path runs on).
    # SRX_BENCH_DEVICE: development override (several ranks on one GPU to exercise the N > 1 path on a 1-GPU box)
    ctx = sr.Context(int(os.environ.get("SRX_BENCH_DEVICE", local_rank)))
    from singlerust_amd.rendezvous import StarGroup
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ     # under torch.distributed.run
    group = StarGroup(rank, world)
    dist = group if (world > 1 or launched) else None
    json_fd = None
    if dist is not None:
        # RCCL prints a version banner on stdout at init: keep stdout clean for the one JSON line
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        collective, comm_err = "rccl", ""
        try:
            # every rank goes through the same sequence whatever fails where (a rank that raised early
            # would leave the others waiting in the star): id (zeros = rank 0 could not make one) ->
            # init -> agreement on the outcome
            uid = None
            if rank == 0:
                try:
                    uid = sr.Context.comm_unique_id()
                except Exception as e:          # noqa: BLE001
                    uid, comm_err = bytes(F.UNIQUE_ID_BYTES), str(e)
            uid = group.broadcast_bytes(uid, F.UNIQUE_ID_BYTES)
            ok = uid != bytes(F.UNIQUE_ID_BYTES) and os.environ.get("SRX_BENCH_COLLECTIVE", "rccl") == "rccl"
            if ok:
                try:
                    ctx.comm_init(world, rank, uid)
                except Exception as e:          # noqa: BLE001
                    ok, comm_err = False, str(e)
            if group.allreduce_max(0.0 if ok else 1.0) > 0.0:
                # RCCL could not be brought up on some rank (or SRX_BENCH_COLLECTIVE=host): the sums over ranks
                # go through the host transport hook of the C-ABI (srx_comm_init_host) over the rendezvous
                # sockets instead.  Same arithmetic, slower exchange; the JSON line says which one ran.
                F.check(lib.srx_comm_destroy(ctx.handle), ctx.handle)
                ctx.comm_init_host(world, rank, group.allreduce_sum_f64)
                collective = "host-star"
                print(f"[bench rank {rank}] RCCL unavailable ({comm_err or 'see other ranks'}): host all-reduce", file=sys.stderr)
        finally:
            # fd 1 stays on stderr for the rest of a multi-rank run (RCCL may warn on stdout at any collective); the one
            # JSON line goes to the saved descriptor at the end
            json_fd = saved

    cells, genes, density, seed = CONFIGS[a.config]
    if a.cells:
        cells = a.cells
    if a.scaling == "strong":
        cells = (cells + world - 1) // world
    n_global = cells * world
    params = F.SynthParams()
    lib.srx_synth_defaults(C.byref(params), seed, n_global, genes, density)
    row0, row1 = rank * cells, (rank + 1) * cells

    t_gen = time.perf_counter()
    h = C.c_void_p()
    f64 = a.storage == "f64"
    F.check(lib.srx_synth_generate(ctx.handle, C.byref(params), row0, row1, F.F64 if f64 else F.F32,
                                   F.STORE_F64 if f64 else F.STORE_F32, C.byref(h)), ctx.handle)
    pristine = sr.DeviceCsr(ctx, h)
    pristine.prepare()          # pattern-only row/gene-tile cuts: part of the resident layout, like indptr
    pristine.reserve_results(a.hvg, a.npc)      # output block (scores + small results): clones get their own, before the clock
    info = pristine.info()
    nnz = int(info.nnz)
    t_gen = time.perf_counter() - t_gen
    bytes_per_copy = nnz * (12 if f64 else 8) + (cells + 1) * 8
    n_steps_total = a.warmup + a.steps
    max_copies = max(1, int(a.max_copies_gb * 1e9 // bytes_per_copy) - 1)
    n_copies = min(n_steps_total, max_copies)
    copies = [pristine.clone() for _ in range(n_copies)]
    ctx.synchronize()

    opts = F.PcaOpts(a.npc, -1, -1, -1, 0, 0, a.solver, 0.0, 12345)
    res = F.PipelineResult()

    def step(mat):
        F.check(lib.srx_pipeline(mat.handle, a.target_sum, a.hvg, C.byref(opts), C.byref(res)), ctx.handle)

    def sync_all():
        ctx.synchronize()               # every kernel of the path runs on this context's stream
        if dist is not None:
            dist.barrier()

    # warmup (untimed)
    used = 0
    for _ in range(a.warmup):
        step(copies[used]); used += 1
    # timed: chunks of fresh copies; restoring copies from the pristine matrix is outside the clock
    prof_mask = sum(1 << c for c in (F.K_NORMALIZE, F.K_MOMENTS, F.K_COMPACT, F.K_SPMM_FWD, F.K_SPMM_T, F.K_GRAM, F.K_DENSE, F.K_ROWSUM))
    ctx.prof_enable(prof_mask)
    ctx.prof_reset()
    elapsed = 0.0
    done = 0
    stage = {"normalize": 0.0, "moments": 0.0, "select": 0.0, "pca": 0.0}
    iters = []
    while done < a.steps:
        if used >= n_copies:
            for c in copies:
                c.copy_values_from(pristine)
            used = 0
        chunk = min(a.steps - done, n_copies - used)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(chunk):
            ts = time.perf_counter()
            step(copies[used]); used += 1
            if os.environ.get("SRX_BENCH_TRACE"):
                print(f"[bench] step {done} copy {used - 1}: {(time.perf_counter() - ts) * 1e3:.2f} ms host, pca stage {res.ms_pca:.2f} ms",
                      file=sys.stderr)
            stage["normalize"] += res.ms_normalize; stage["moments"] += res.ms_moments
            stage["select"] += res.ms_select; stage["pca"] += res.ms_pca
            iters.append(int(res.pca.n_iter))
        sync_all()
        elapsed += time.perf_counter() - t0
        done += chunk
    if dist is not None:
        elapsed = dist.allreduce_max(elapsed)

    prof = {}
    names = {F.K_NORMALIZE: "normalize_log1p", F.K_MOMENTS: "gene_moments", F.K_COMPACT: "hvg_compact",
             F.K_SPMM_FWD: "spmm_fwd", F.K_SPMM_T: "spmm_t", F.K_GRAM: "gram_sparse", F.K_DENSE: "dense_apply",
             F.K_ROWSUM: "row_sums"}
    for cls_, name in names.items():
        ms, n, b = ctx.prof_get(cls_)
        if n:
            prof[name] = {"launches": n, "avg_ms": ms / n, "alg_bytes_per_launch": b / n,
                          "GBps": (b / n) / (ms / n * 1e-3) / 1e9 if ms > 0 else None,
                          "frac_of_peak": (b / n) / (ms / n * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else None}
    ctx.prof_enable(0)

    if rank == 0:
        value = n_global * a.steps / elapsed
        fwd = prof.get("spmm_fwd", {})
        traffic = load_traffic(a.config)
        for name, d in prof.items():
            per_step = traffic.get(name)
            d["hbm_traffic_per_launch"] = per_step / (d["launches"] / a.steps) if per_step else None
        dom_name = max(prof, key=lambda k: prof[k]["avg_ms"] * prof[k]["launches"]) if prof else None
        dom = prof.get(dom_name, {})

        def roof(name, d, kernel, note):
            return {"bound": "hbm", "kernel": kernel, "achieved": d.get("GBps"), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": d.get("frac_of_peak"), "traffic": d.get("hbm_traffic_per_launch"),
                    "launches": d.get("launches"), "avg_ms": d.get("avg_ms"),
                    "alg_bytes_per_launch": d.get("alg_bytes_per_launch"), "note": note}

        out = {
            "metric": "cells/sec end-to-end normalise->HVG->50-PC PCA; SpMM achieved HBM GB/s vs peak",
            "value": value, "unit": "cells/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": a.scaling,
            "vs_baseline": None, "dtype": a.storage, "data": "synthetic",
            "config": {
                "workload": f"{a.config}: {cells} cells/GPU x {genes} genes, density {density}, seed {seed}; "
                            f"normalize_total(1e4,Row)+log1p+HVG({a.hvg})+{a.npc}-PC PCA; values {a.storage} / indices i32 in HBM",
                "cells_global": n_global, "genes": genes, "nnz_per_gpu": nnz, "hvg": a.hvg, "n_pc": a.npc,
                "panel_width": 64, "parallelism": f"row-shard x{world}",
                **({"collective": collective} if dist is not None else {}),
                "nnz_hvg_compacted_per_gpu": int(res.pca.nnz_selected),
                "subspace_iterations": iters, "pca_residual": float(res.pca.residual),
                "pca_solver": {1: "gram", 2: "spmm"}.get(int(res.pca.solver), "?"),
                # the ~80 small launches of the subspace iteration are replayed from captured hipGraphs (three
                # segments per solve); their kernels are therefore not bracketed by per-class events
                "pca_iteration_hip_graphs": int(res.pca.solver) == 1 and not os.environ.get("SRX_NO_GRAPH"),
            },
            # the kernel with the largest share of the step (live HIP-event timing on the ctx stream)
            "roofline": roof(dom_name, dom, KERNEL_SYMBOL.get(dom_name, dom_name), ROOF_NOTE.get(dom_name, "")),
            # BASELINE.json's second metric: the CSR x 64-column-panel SpMM against the HBM peak
            "roofline_spmm": roof("spmm_fwd", fwd, KERNEL_SYMBOL["spmm_fwd"], ROOF_NOTE["spmm_fwd"]),
            "kernels": prof,
            "stage_ms_per_step": {k: v / a.steps for k, v in stage.items()},
            "setup": {"generate_s": t_gen, "copies": n_copies},
        }
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(F, params, genes, min(a.cpu_sample_cells, cells), a.hvg, a.npc,
                                                   a.target_sum)
                out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
            except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU line
                out["cpu_baseline"] = {"value": None, "unit": "cells/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {e!r}"}
        if json_fd is not None:
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(out) + "\n").encode())
        else:
            print(json.dumps(out), flush=True)

    for c in copies:
        c.free()
    pristine.free()
    if dist is not None:
        dist.barrier()
        dist.close()
    ctx.close()


if __name__ == "__main__":
    main()
