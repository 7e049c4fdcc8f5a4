#!/usr/bin/env python3
"""bench.py — cells/s of the hot path  normalize_total(1e4, Row) -> log1p -> gene moments ->
HVG(2000) -> 50-PC PCA  on synthetic CSR resident in HBM, plus the SpMM's achieved HBM GB/s.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one srx_pipeline() call over one fresh copy of the raw count matrix (the pipeline normalises X in place,
so every step gets its own copy, cloned — with its pattern-only index structures and its reserved result block —
before the timed region: a step makes no device allocation).

N = 1: BASELINE.json configs[2], the configuration the north_star targets are quoted on (1.3M x 28k, ~3 % nnz; it
fits one GPU).  N > 1: configs[3] read literally — the SAME 1.3M cells row-sharded over the N GPUs (strong scaling,
nnz-balanced shards, RCCL all-reduce of the gene moments and of the packed Gram triangle); a `weak` block (1.3M cells
per GPU) is measured after it with fewer steps.

One JSON line on rank 0.  Besides the contract's fields it carries (N = 1 only, each a short bounded run after the
timed region): the f64-storage pipeline (`f64_storage`), the matrix-free solver's SpMM kernels (`roofline_spmm_iter`),
a skewed-gene matrix (`skewed_genes`), a hard spectrum (`hard_spectrum`), the rate including the upload of a host CSR
and the download of the scores (`incl_h2d`), the pipeline on a fresh un-prepared handle (`cold_step`), configs[4] streamed
from pinned host memory through the backed session (`c5_backed`), and three CPU baselines timed on this box (`cpu_baseline`:
the OpenMP restatement over the WHOLE matrix on every physical core; `cpu_baseline_reference_faithful`: serial loops + full SVD on a sample;
`cpu_baseline_c1_serial`: BASELINE.md's C1 whole on one thread).  `python bench.py --gpus N` with N > 1 and no launcher
environment starts its own N ranks (one per GPU).
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
    # name: (cells, genes, density, seed)       SURVEY.md §8 table
    "c2": (100_000, 20_000, 0.05, 2002),
    "c3": (1_300_000, 28_000, 0.03, 3003),
    # config 5's total size on ONE GPU (48 GB of CSR in HBM: no out-of-core tiling needed on 288 GB); `--backed`
    # streams it as row tiles from pinned host memory instead (configs[4] as specified)
    "c5": (10_000_000, 30_000, 0.02, 5005),
}
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable

KERNEL_SYMBOL = {"normalize_log1p": "k_row_apply<T> / k_normalize<T> (the separate in-place calls; the pipeline's moments pass stores the values itself)",
                 "row_sums": "k_row_sum<float>", "gene_moments": "k_gene_moments<float,u16,XF,WB> (f64 moments of the transform + the in-place store)",
                 "select": "k_gene_var + k_hvg_rank + k_hvg_take + k_sel_finish",
                 "hvg_compact": "k_rowcount + k_tfill (+ scan)", "spmm_fwd": "k_spmm_rows (row-major records x 64-col panel)",
                 "spmm_t": "k_spmm_t", "gram_sparse": "k_gram_stripes<float>",
                 "gram_bucket": "k_rec_count + k_rec_scan + k_bucket (owner records of the Gram kernel)",
                 "iterate": "k x 64 subspace iteration (hipGraph replays)", "dense_apply": "k_dense_apply"}
# what each roofline object's bytes are and what the kernel is really bound by: DESIGN.md section 7 (the prose lives there, not in the line)
ROOF_NOTE = {
    "gram_sparse": "alg. bytes = compacted matrix + row pointers read once, packed G written once; bound by its gather loads and the "
                   "L2s' fabric side, not HBM: DESIGN.md sections 3b, 7; profiles/r06_pmc_gram.md",
    "gene_moments": "alg. bytes = 16-bit index + f32 value read, transformed value stored back (nnz * 10) + pointers, cuts, row sums; "
                    "latency-bound 0.5 ms above its read + store floor: DESIGN.md section 7; profiles/r06_pmc_gram.md",
    "spmm_fwd": "alg. bytes = compacted matrix + pointers / row order + panel + the N x n_pc f64 scores; bound by its record reads and "
                "stores: DESIGN.md section 3c; profiles/r06_pmc_spmm.md",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--cells", type=int, default=0, help="override the config's cell count")
    ap.add_argument("--scaling", default="auto", choices=("auto", "weak", "strong"),
                    help="N > 1: strong (default; the config's cells in total, row-sharded = BASELINE.json configs[3]) "
                         "or weak (the config's cells PER GPU)")
    ap.add_argument("--hvg", type=int, default=2000)
    ap.add_argument("--npc", type=int, default=50)
    ap.add_argument("--target-sum", type=float, default=1e4)
    ap.add_argument("--solver", type=int, default=0, help="0 auto, 1 explicit Gram, 2 matrix-free SpMM iteration")
    ap.add_argument("--storage", default="f32", choices=("f32", "f64"),
                    help="value storage in HBM of the headline run: f32 (default; exact for count data, HighlyVariable(n) "
                         "ranked on f64 moments) or f64")
    ap.add_argument("--skew", type=int, default=0, help="1: the quadratic gene-density map of the generator (srx_synth.h)")
    ap.add_argument("--backed", action="store_true",
                    help="out-of-core form (configs[4]): the matrix stays in pinned host memory and is streamed as row tiles "
                         "through the srx_backed_* session, two sweeps; reports H2D-bound cells/s")
    ap.add_argument("--tile-rows", type=int, default=250_000, help="--backed: cells per streamed tile")
    ap.add_argument("--lean", action="store_true", help="only the headline measurement (no extra blocks, no CPU baselines)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-cells", type=int, default=24000, help="reference-faithful CPU baseline: cells of the sample")
    ap.add_argument("--host-sample-cells", type=int, default=0,
                    help="cells of the host-side legs (incl_h2d blocks, threaded CPU baseline); 0 = the whole configuration")
    ap.add_argument("--max-copies-gb", type=float, default=180.0)
    return ap.parse_args()


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def load_traffic(config):
    """HBM bytes per pipeline step from the committed PMC passes (profiles/make_traffic.py), per bench kernel class."""
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_traffic_{config}.json")
        try:
            with open(path) as fh:
                t = json.load(fh)["kernels"]
        except (OSError, ValueError, KeyError):
            continue
        cls = {"normalize_log1p": ("k_row_pass", "k_row_apply"), "row_sums": ("k_row_sum",), "gene_moments": ("k_gene_moments",),
               "spmm_fwd": ("k_spmm_fwd", "k_spmm_rows"), "spmm_t": ("k_spmm_t",),
               "gram_sparse": ("k_gram_stripes", "k_gram_sparse", "k_gram_reduce"),
               "gram_bucket": ("k_bucket", "k_rec_count", "k_rec_scan"),
               "dense_apply": ("k_dense_apply",),
               "hvg_compact": ("k_tcount", "k_rowcount", "k_tfill", "k_scan_block_sums", "k_scan_serial", "k_scan_apply", "k_seglen")}
        # the moments pass runs exactly once per pipeline step: its launches_per_step calibrates the file's step count (the r05
        # tables were made with a wrong steps argument: 7 launches recorded as 1.75 per step)
        once = [d["launches_per_step"] for sym, d in t.items() if "k_gene_moments" in sym and d.get("launches_per_step")]
        norm = max(once) if once else 1.0
        out = {}
        for name, subs in cls.items():
            tot, hit = 0.0, False
            for sym, d in t.items():
                if any(sym.startswith(s_) or (" " + s_) in sym or ("srx::" + s_) in sym for s_ in subs):
                    tot += d["hbm_bytes"] * d["launches_per_step"] / norm      # bytes per pipeline step
                    hit = True
            if hit:
                out[name] = tot
        return out, rnd
    return {}, None


class Bench:
    """One context, its rendezvous group, and the workload runner."""

    def __init__(self, a):
        import singlerust_amd as sr
        from singlerust_amd import _ffi as F
        self.a, self.sr, self.F = a, sr, F
        self.lib = F.lib()
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        # One process per GPU.  The handshake (128-byte RCCL id, barriers, max over ranks) uses a stdlib socket star,
        # not torch.distributed: the torch wheel bundles its own HIP runtime and RCCL, and once it is imported into
        # this process ncclCommInitRank of the system RCCL that libsrx_hip.so uses fails; nothing on the data path
        # needs torch.  SRX_BENCH_DEVICE: development override (several ranks on one GPU)
        self.ctx = sr.Context(int(os.environ.get("SRX_BENCH_DEVICE", self.local_rank)))
        from singlerust_amd.rendezvous import StarGroup
        launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ     # under torch.distributed.run
        self.group = StarGroup(self.rank, self.world)
        self.dist = self.group if (self.world > 1 or launched) else None
        self.json_fd = None
        self.collective = None
        self.comm_info = None
        if self.dist is not None:
            self._comm_init()

    def _comm_init(self):
        F, sr, group, ctx, rank, world = self.F, self.sr, self.group, self.ctx, self.rank, self.world
        # RCCL prints a version banner on stdout at init: keep stdout clean for the one JSON line
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        collective, comm_err = "rccl", ""
        try:
            # every rank goes through the same sequence whatever fails where (a rank that raised early would leave the
            # others waiting in the star): id (zeros = rank 0 could not make one) -> init -> agreement on the outcome
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
                # RCCL could not be brought up on some rank (or SRX_BENCH_COLLECTIVE=host): the sums over ranks go
                # through the host transport hook of the C-ABI (srx_comm_init_host) over the rendezvous sockets.
                # Same arithmetic, slower exchange; the JSON line says which one ran.
                if os.environ.get("SRX_BENCH_COLLECTIVE", "rccl") != "host":
                    # a scaling curve over the host star would not be a measurement of RCCL over xGMI: refuse
                    print(f"[bench rank {rank}] RCCL could not be brought up ({comm_err or 'see other ranks'}); "
                          "SRX_BENCH_COLLECTIVE=host selects the host transport explicitly", file=sys.stderr)
                    sys.exit(3)
                F.check(self.lib.srx_comm_destroy(ctx.handle), ctx.handle)
                ctx.comm_init_host(world, rank, group.allreduce_sum_f64)
                collective = "host-star"
        finally:
            # fd 1 stays on stderr for the rest of a multi-rank run (RCCL may warn on stdout at any collective); the one
            # JSON line goes to the saved descriptor at the end
            self.json_fd = saved
        self.collective = collective
        # what the sums really go through: kind, RCCL version, and a 1.0 summed over the communicator by the path's own
        # all-reduce (a collective call: every rank makes it) — must equal the world size
        kind, nr, ver, seen = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        F.check(self.lib.srx_comm_info(ctx.handle, C.byref(kind), C.byref(nr), C.byref(ver), C.byref(seen)), ctx.handle)
        self.comm_info = {"collective": collective, "kind": {0: "none", 1: "rccl", 2: "host"}[kind.value],
                          "rccl_version": ver.value, "n_ranks": nr.value, "n_ranks_seen": seen.value,
                          "launcher": os.environ.get("SRX_BENCH_LAUNCHER", "torch.distributed.run / external")}
        if seen.value != world or (collective == "rccl") != (kind.value == 1):
            print(f"[bench rank {rank}] communicator check failed: {self.comm_info}", file=sys.stderr)
            sys.exit(3)

    def overlap_info(self):
        """How the Gram solver's exchange ran on this rank (srx_comm_overlap_info): split in three pieces with two of them under
        the second half of the stripe kernel, and whether that half ran on the CU-masked stream."""
        n, masked = C.c_int32(0), C.c_int32(0)
        try:
            self.F.check(self.lib.srx_comm_overlap_info(self.ctx.handle, C.byref(n), C.byref(masked)), self.ctx.handle)
        except Exception:       # noqa: BLE001
            return {}
        return {"gram_exchange_split_launches": n.value, "gram_exchange_cu_masked": bool(masked.value)}

    def sync_all(self):
        self.ctx.synchronize()               # every kernel of the path runs on this context's streams (the pipeline joins them)
        if self.dist is not None:
            self.dist.barrier()

    def params(self, config, n_global, skew=0):
        _, genes, density, seed = CONFIGS[config]
        p = self.F.SynthParams()
        self.lib.srx_synth_defaults(C.byref(p), seed, n_global, genes, density)
        p.skew = skew
        return p

    def shard(self, p, n_global):
        """This rank's row range: nnz-balanced contiguous cut of the global matrix (srx_partition_rows on the generator's
        row offsets — 8 bytes per cell on the host, no values)."""
        import numpy as np
        if self.world == 1:
            return 0, n_global
        ip = np.zeros(n_global + 1, dtype=np.uint64)
        self.lib.srx_synth_indptr(C.byref(p), 0, n_global, self.F.ptr(ip))
        cut = np.zeros(self.world + 1, dtype=np.uint64)
        self.F.check(self.lib.srx_partition_rows(self.F.ptr(ip), n_global, self.world, self.F.ptr(cut)))
        return int(cut[self.rank]), int(cut[self.rank + 1])

    def run(self, config, n_global, row0, row1, storage, steps, warmup, solver=0, hvg=None, skew=0, tolerate_noconv=False, headline=False,
            per_step=False):
        """K timed pipeline steps on rows [row0, row1) of the synthetic matrix; returns the measurements of this rank
        (times already maxed over ranks)."""
        a, F, sr, lib, ctx = self.a, self.F, self.sr, self.lib, self.ctx
        hvg = hvg or a.hvg
        p = self.params(config, n_global, skew)
        f64 = storage == "f64"
        t_gen = time.perf_counter()
        h = C.c_void_p()
        F.check(lib.srx_synth_generate(ctx.handle, C.byref(p), row0, row1, F.F64 if f64 else F.F32,
                                       F.STORE_F64 if f64 else F.STORE_F32, C.byref(h)), ctx.handle)
        pristine = sr.DeviceCsr(ctx, h)
        pristine.prepare()          # pattern-only structures (gene-tile cuts, 16-bit index mirror, per-gene counts): part of the resident layout
        pristine.reserve_results(hvg, a.npc)      # output block (scores + small results): clones get their own, before the clock
        info = pristine.info()
        nnz = int(info.nnz)
        ctx.synchronize()
        t_gen = time.perf_counter() - t_gen
        bytes_per_copy = nnz * (12 if f64 else 8) + (row1 - row0 + 1) * 8 + (row1 - row0) * a.npc * 8
        n_total = warmup + steps
        max_copies = max(1, int(a.max_copies_gb * 1e9 // max(bytes_per_copy, 1)) - 1)
        n_copies = min(n_total, max_copies)
        copies = [pristine.clone() for _ in range(n_copies)]
        ctx.synchronize()
        opts = F.PcaOpts(a.npc, -1, -1, -1, 0, 0, solver, 0.0, 12345)
        res = F.PipelineResult()
        failures = []

        def step(mat):
            rc = lib.srx_pipeline(mat.handle, a.target_sum, hvg, C.byref(opts), C.byref(res))
            if rc == F.E_NOCONV and tolerate_noconv:
                failures.append(rc)
                return
            F.check(rc, ctx.handle)

        used = 0
        for _ in range(warmup):
            step(copies[used % n_copies]); used += 1
            if used % n_copies == 0 and used < warmup:
                for c in copies:
                    c.copy_values_from(pristine)
        if used >= n_copies:
            for c in copies:
                c.copy_values_from(pristine)
            used = 0
        classes = {F.K_NORMALIZE: "normalize_log1p", F.K_ROWSUM: "row_sums", F.K_MOMENTS: "gene_moments", F.K_SELECT: "select",
                   F.K_COMPACT: "hvg_compact", F.K_GRAM: "gram_sparse", F.K_BUCKET: "gram_bucket", F.K_ITERATE: "iterate",
                   F.K_DENSE: "dense_apply",
                   F.K_SPMM_FWD: "spmm_fwd", F.K_SPMM_T: "spmm_t"}
        stage_classes = (F.K_COMPACT, F.K_BUCKET, F.K_GRAM)      # the roofline object's stage (Gram formation)

        def timed_pass(n_steps, mask):
            """n_steps pipeline steps between two synchronisations, the class timers of `mask` on: seconds, per-class records,
            the pipeline's own stage clocks, iteration counts."""
            nonlocal used
            ctx.prof_enable(mask)
            ctx.prof_reset()
            elapsed_, done = 0.0, 0
            stage_ = {"normalize": 0.0, "moments": 0.0, "select": 0.0, "pca": 0.0}
            iters_ = []
            while done < n_steps:
                if used >= n_copies:            # restoring copies from the pristine matrix is outside the clock
                    for c in copies:
                        c.copy_values_from(pristine)
                    used = 0
                chunk = min(n_steps - done, n_copies - used)
                self.sync_all()
                t0 = time.perf_counter()
                for _ in range(chunk):
                    ts = time.perf_counter()
                    step(copies[used]); used += 1
                    if per_step:                # every step between two synchronisations of its own: min / median over the steps
                        self.sync_all()
                        step_ms.append((time.perf_counter() - ts) * 1e3)
                    if os.environ.get("SRX_BENCH_TRACE"):
                        print(f"[bench] step {done} copy {used - 1}: {(time.perf_counter() - ts) * 1e3:.2f} ms host, "
                              f"pca stage {res.ms_pca:.2f} ms", file=sys.stderr)
                    stage_["normalize"] += res.ms_normalize; stage_["moments"] += res.ms_moments
                    stage_["select"] += res.ms_select; stage_["pca"] += res.ms_pca
                    iters_.append(int(res.pca.n_iter))
                self.sync_all()
                elapsed_ += time.perf_counter() - t0
                done += chunk
            if self.dist is not None:
                elapsed_ = self.dist.allreduce_max(elapsed_)
            prof_ = {}
            for cls_, name in classes.items():
                if not (mask >> cls_) & 1:
                    continue
                ms, n, b = ctx.prof_get(cls_)
                if n:
                    aux = ctx.prof_get_aux(cls_)
                    prof_[name] = {"launches": n, "avg_ms": ms / n, "alg_bytes_per_launch": b / n,
                                   **({"aux_bytes_per_launch": aux / n} if aux else {}),
                                   "GBps": (b / n) / (ms / n * 1e-3) / 1e9 if ms > 0 else None,
                                   "frac_of_peak": (b / n) / (ms / n * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else None}
            return elapsed_, prof_, stage_, iters_

        all_mask = sum(1 << c for c in classes)
        step_ms = []
        breakdown = None
        if headline:
            # The K timed steps carry the class timers of the roofline object's stage only (compaction, owner records, stripe
            # kernel: six event records per step); a timer is two event records around its launches, and with all eight classes
            # on the records themselves cost 0.05-0.08 ms of a 9.7 ms step (measured: 9.74-9.81 against 9.66-9.74 with none).
            # The other classes are timed in a SEPARATE pass of a few steps behind the clock, reported as such.
            elapsed, prof, stage, iters = timed_pass(steps, sum(1 << c for c in stage_classes))
            bd_steps = max(3, min(steps, 8))
            bd_elapsed, bd_prof, _, _ = timed_pass(bd_steps, all_mask)
            for name, rec in bd_prof.items():
                if name not in prof:
                    # (downstream divides a class's launches by the K timed steps: the record is scaled to K steps' worth)
                    prof[name] = {**rec, "launches": rec["launches"] * steps / bd_steps, "timed_in": "breakdown pass"}
            breakdown = {"steps": bd_steps, "ms_per_step": bd_elapsed / bd_steps * 1e3,
                         "kernel_ms_per_step": {name: rec["avg_ms"] * rec["launches"] / bd_steps for name, rec in bd_prof.items()},
                         "note": "all class timers on, behind the timed region: the source of every class time outside the Gram formation stage"}
        else:
            elapsed, prof, stage, iters = timed_pass(steps, all_mask)
        ctx.prof_enable(0)
        out = {"elapsed": elapsed, "ms_per_step": elapsed / steps * 1e3, "nnz": nnz, "prof": prof,
               "stage_ms_per_step": {k_: v / steps for k_, v in stage.items()}, "iters": iters,
               "residual": float(res.pca.residual), "nnz_selected": int(res.pca.nnz_selected),
               "solver": {1: "gram", 2: "spmm"}.get(int(res.pca.solver), "?"), "generate_s": t_gen, "copies": n_copies,
               "noconv_steps": len(failures), "genes": int(info.n_cols), "breakdown": breakdown, "step_ms": sorted(step_ms)}
        for c in copies:
            c.free()
        pristine.free()
        return out

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.close()
        self.ctx.close()


def attributed(prof, steps):
    """ms per step inside the bracketed kernel classes that are NOT overlapped (the in-place write-back runs on the side
    stream beside the iteration: reported, not added; dense_apply sits inside `iterate`)."""
    per = {k_: v["avg_ms"] * v["launches"] / steps for k_, v in prof.items()}
    serial = sum(per.get(k_, 0.0) for k_ in ("row_sums", "gene_moments", "select", "hvg_compact", "gram_bucket", "gram_sparse", "iterate", "spmm_fwd", "spmm_t"))
    return per, serial


LDS_PEAK_GBS = 128 * 256 * 2.4          # 128 B / clk / CU x 256 CUs x 2.4 GHz = 78 643 GB/s (MI355X_MICROARCH.md)
LDS_F64_ATOMIC_LANES_PER_S = 2.5 * 256 * 2.4e9     # random-address ds_add_f64: 2.5 lanes / clk / CU (bench_micro/lds_atomic_banks.hip)
LDS_U64_ATOMIC_LANES_PER_S = 64 / 13.6 * 256 * 2.4e9   # random-address ds_add_u64: 13.6 clk per 64 lanes (same microbenchmark)
GATHER_LOADS_PER_S = 1e8 / 3.04e-3                 # L2-resident wave loads of the Gram kernel's shape (bench_micro/l2_gather.hip)


def roof(d, kernel, note, other=None):
    """The contract's roofline object (HBM: algorithmic bytes / live hipEvent time).  `other`: what ELSE bounds the kernel —
    the resource its inner loop actually runs on, measured the same way (work per launch / the same launch time)."""
    return {"bound": "hbm", "kernel": kernel, "achieved": d.get("GBps"), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": d.get("frac_of_peak"), "traffic": d.get("hbm_traffic_per_launch"), "launches": d.get("launches"),
            "avg_ms": d.get("avg_ms"), "alg_bytes_per_launch": d.get("alg_bytes_per_launch"),
            **({"aux_bytes_per_launch": d["aux_bytes_per_launch"]} if d.get("aux_bytes_per_launch") else {}),
            **({"other_bounds": other} if other else {}), "note": note}


def other_bounds(name, d, nnz_sel, n_cells, sigma=0.3):
    """Secondary rooflines of the two kernels whose inner loop does not run on HBM (DESIGN.md sections 3b / 3c)."""
    ms = d.get("avg_ms")
    if not ms or not nnz_sel or not n_cells:
        return None
    t = ms * 1e-3
    if name == "spmm_fwd":
        # every kept entry reads its gene's 64 panel columns from LDS (4 slices x 16 columns x 4 B), plus ~12 % of padded slots
        lds_bytes = nnz_sel * 64 * 4.0
        return {"lds_read": {"achieved": lds_bytes / t / 1e9, "peak": LDS_PEAK_GBS, "unit": "GB/s",
                             "frac": lds_bytes / t / 1e9 / LDS_PEAK_GBS,
                             "note": "panel reads: kept entries x 64 columns x 4 B out of the LDS panel slices (padded record "
                                     "slots not counted); the LDS cannot hold the 2000 x 50 panel (400 KB), so ANY gather "
                                     "formulation moves these bytes: 0.30 ms at the LDS peak against 0.16 ms of HBM time"}}
    if name == "gram_sparse":
        import math
        m = nnz_sel / n_cells
        products = n_cells * (m * m * math.exp(sigma * sigma) + m) / 2.0       # sum over cells of m_i (m_i + 1) / 2, log-normal m_i
        loads = nnz_sel * 1.07 / 2.0
        return {"lds_atomics": {"achieved": products / t, "peak": LDS_U64_ATOMIC_LANES_PER_S, "unit": "lane-atomics/s",
                                "frac": products / t / LDS_U64_ATOMIC_LANES_PER_S, "products_per_launch_estimate": products,
                                "note": "one 64-bit LDS atomic lane per scalar product — INTEGER atomics (the kernel's fixed-point "
                                        "mode: non-negative f32 values of bounded range, which the bench's are); peak = the measured "
                                        "random-address ds_add_u64 rate (ds_add_f64: 2.5 lanes per clock and CU, 0.53 of it)"},
                "valu": {"note": "round 4: SQ_ACTIVE_INST_VALU = 100 % of the launch at 28.6 instructions per load; round 5 (assembly "
                                 "core): 21 VALU + 18 scalar instructions per load = 2.1 / 1.8 ms-equivalents of issue in a 2.66 ms launch "
                                 "(profiles/r05_pmc_gram.md)"},
                "gather_loads": {"achieved": loads / t, "peak": GATHER_LOADS_PER_S, "unit": "load instructions/s",
                                 "frac": loads / t / GATHER_LOADS_PER_S, "loads_per_launch_estimate": loads,
                                 "note": "one load instruction per TWO owner records (records = kept entries x 1.07: a suffix longer "
                                         "than 64 entries is several); peak = the L2-resident rate of bench_micro/l2_gather.hip, "
                                         "1e8 wave loads in 3.04 ms on 256 CUs with 16 waves x 8 loads in flight (5.0 ms out of the "
                                         "Infinity Cache, 6.3 out of HBM: 59 % of this kernel's requests hit L2)"}}
    return None


def cold_step(B, config, n_global, storage, reps=3):
    """What a caller pays who runs the path ONCE per dataset: srx_pipeline on a FRESH handle — no srx_matrix_prepare, no
    srx_matrix_reserve_results before the clock; the matrix itself resident in HBM.  The first call builds the pattern-only
    structures the handle does not carry yet — none since round 6 for up to 38 000 genes: the 16-bit index mirror, the gene-tile cuts and
    the per-gene counts are made in one walk where the indices are written, by srx_matrix_upload and srx_synth_generate alike — and
    allocates the result block inside the step.  Run after
    the timed region, so the context is warm (scratch buffers, captured graphs, code objects): the cost measured is the
    matrix's own.  `prepare_ms` / `reserve_ms`: the two set-up calls timed on their own on another fresh handle."""
    a, F, sr, lib, ctx = B.a, B.F, B.sr, B.lib, B.ctx
    p = B.params(config, n_global)
    f64 = storage == "f64"
    opts = F.PcaOpts(a.npc, -1, -1, -1, 0, 0, 0, 0.0, 12345)
    res = F.PipelineResult()

    def fresh():
        h = C.c_void_p()
        F.check(lib.srx_synth_generate(ctx.handle, C.byref(p), 0, n_global, F.F64 if f64 else F.F32,
                                       F.STORE_F64 if f64 else F.STORE_F32, C.byref(h)), ctx.handle)
        m = sr.DeviceCsr(ctx, h)
        ctx.synchronize()
        return m
    runs = []
    for _ in range(reps):
        m = fresh()
        t0 = time.perf_counter()
        F.check(lib.srx_pipeline(m.handle, a.target_sum, a.hvg, C.byref(opts), C.byref(res)), ctx.handle)
        ctx.synchronize()
        runs.append((time.perf_counter() - t0) * 1e3)
        m.free()
    m = fresh()
    t0 = time.perf_counter()
    m.prepare()
    ctx.synchronize()
    t1 = time.perf_counter()
    m.reserve_results(a.hvg, a.npc)
    ctx.synchronize()
    t2 = time.perf_counter()
    m.free()
    best = min(runs)
    return {"ms": best, "value": n_global / (best * 1e-3), "unit": "cells/s", "runs_ms": runs,
            "prepare_ms": (t1 - t0) * 1e3, "reserve_results_ms": (t2 - t1) * 1e3,
            "note": "srx_pipeline on a fresh, un-prepared, un-reserved handle (the matrix resident in HBM, the context warm): "
                    "the device-resident cost of running the path once per dataset; the headline `value` is the steady state of "
                    "repeated steps on prepared clones (pattern-only structures and result block amortised)"}


class HostMatrix:
    """The first `n_cells` rows of the synthetic matrix in the REFERENCE layout on the host (u64 offsets / indices, f32 values),
    filled by a thread pool; the two big arrays in PINNED memory (hipHostMalloc) when `pinned` — what a caller gets who allocates
    X for the device — or in plain numpy arrays (pageable: what a Rust Vec is)."""

    def __init__(self, B, config, n_cells, pinned=True):
        import concurrent.futures as cf
        import numpy as np
        F, lib = B.F, B.lib
        cells, self.genes, _, _ = CONFIGS[config]
        p = B.params(config, B.a.cells or cells)
        self.n = n_cells
        t0 = time.perf_counter()
        self.ip = np.zeros(n_cells + 1, dtype=np.uint64)
        lib.srx_synth_indptr(C.byref(p), 0, n_cells, F.ptr(self.ip))
        nnz = int(self.ip[-1])
        self._hip, self._raw = None, []
        self.pinned = False
        if pinned:
            hip = C.CDLL("libamdhip64.so")
            hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
            hip.hipHostFree.argtypes = [C.c_void_p]
            hidx, hval = C.c_void_p(), C.c_void_p()
            if hip.hipHostMalloc(C.byref(hidx), max(nnz, 1) * 8, 0) == 0:
                if hip.hipHostMalloc(C.byref(hval), max(nnz, 1) * 4, 0) == 0:
                    self._hip, self._raw, self.pinned = hip, [hidx, hval], True
                    self.idx = np.ctypeslib.as_array(C.cast(hidx, C.POINTER(C.c_uint64)), shape=(nnz,))
                    self.val = np.ctypeslib.as_array(C.cast(hval, C.POINTER(C.c_float)), shape=(nnz,))
                else:
                    hip.hipHostFree(hidx)
        if not self.pinned:
            self.idx, self.val = np.zeros(nnz, np.uint64), np.zeros(nnz, np.float32)
        tile = 20_000
        ip, idx, val = self.ip, self.idx, self.val

        def fill(r0):
            r1 = min(n_cells, r0 + tile)
            e0, e1 = int(ip[r0]), int(ip[r1])
            sub = (ip[r0:r1 + 1] - ip[r0]).astype(np.uint64)
            lib.srx_synth_fill_host(C.byref(p), r0, r1, F.ptr(sub), F.ptr(idx[e0:e1]), F.ptr(val[e0:e1]))
        with cf.ThreadPoolExecutor(max_workers=max(1, min(usable_cores(), 64))) as ex:
            list(ex.map(fill, range(0, n_cells, tile)))
        self.generate_s = time.perf_counter() - t0
        self.host_bytes = self.ip.nbytes + self.idx.nbytes + self.val.nbytes

    def head(self, n):
        """(ip, idx, val) of the first n cells: views, no copy."""
        n = min(n, self.n)
        e = int(self.ip[n])
        return self.ip[:n + 1], self.idx[:e], self.val[:e]

    def free(self):
        self.idx = self.val = None
        if self._hip is not None:
            for h in self._raw:
                self._hip.hipHostFree(h)
            self._hip, self._raw = None, []


def incl_h2d(B, H, what):
    """SURVEY.md 8(d)'s second form of the metric: cells/s INCLUDING the H2D of the CSR and the D2H of the results — the rate
    a caller sees who hands over HOST buffers for every pipeline: upload of the reference-layout CSR (u64 indices narrowed on
    the host side of the link by the H2D workers), the pipeline on the fresh handle, download of the f64 scores."""
    import numpy as np
    a, F, sr, lib, ctx = B.a, B.F, B.sr, B.lib, B.ctx
    n, ip, idx, val = H.n, H.ip, H.idx, H.val
    opts = F.PcaOpts(a.npc, -1, -1, -1, 0, 0, 0, 0.0, 12345)
    res = F.PipelineResult()
    scores = np.zeros((n, a.npc))
    best = None
    for _ in range(2 if n > 500_000 else 3):
        t0 = time.perf_counter()
        m = sr.DeviceCsr.upload(ctx, n, H.genes, ip, idx, val, F.STORE_F32)
        ctx.synchronize()
        t1 = time.perf_counter()
        F.check(lib.srx_pipeline(m.handle, a.target_sum, a.hvg, C.byref(opts), C.byref(res)), ctx.handle)
        ctx.synchronize()
        t2 = time.perf_counter()
        F.check(lib.srx_result_fetch(m.handle, F.ptr(scores), None, None, None, None, None), ctx.handle)
        t3 = time.perf_counter()
        m.free()
        cur = (t3 - t0, t1 - t0, t2 - t1, t3 - t2)
        if best is None or cur[0] < best[0]:
            best = cur
    return {"cells": n, "host_bytes": H.host_bytes, "host_memory": "pinned (hipHostMalloc)" if H.pinned else "pageable",
            "upload_s": best[1], "pipeline_s": best[2], "d2h_scores_s": best[3], "total_s": best[0],
            "value": n / best[0], "unit": "cells/s", "upload_GBps_of_host_bytes": H.host_bytes / best[1] / 1e9,
            "d2h_GBps": scores.nbytes / best[3] / 1e9,
            "note": f"{what}: srx_matrix_upload of the reference-layout host CSR (u64 offsets / indices, f32 values: "
                    f"{H.host_bytes / 1e9:.1f} GB; the H2D workers narrow the indices to 16 bits on the host side of the link: 6 of a non-zero's 12 bytes cross it) + srx_pipeline "
                    "(first call on the handle: it also counts the non-zeros per gene and allocates the result block) + "
                    "srx_result_fetch of the f64 scores; best of the runs.  The upload dominates — the drop-in uploads once and "
                    "runs the whole path on the handle; a matrix larger than HBM goes through the backed session (--backed)"}


def _blas_limit(n):
    """Context manager limiting the BLAS / OpenMP pools numpy and scipy call into (threadpoolctl; a no-op without it)."""
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=n)
    except Exception:       # noqa: BLE001
        import contextlib
        return contextlib.nullcontext()


def physical_cores():
    """Physical cores this process may run on (unique (socket, core) pairs of /proc/cpuinfo among the affinity set)."""
    try:
        allowed = os.sched_getaffinity(0)
        seen, cpu, phys, core = set(), None, None, None
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("processor"):
                    cpu = int(line.split(":")[1])
                elif line.startswith("physical id"):
                    phys = int(line.split(":")[1])
                elif line.startswith("core id"):
                    core = int(line.split(":")[1])
                elif not line.strip():
                    if cpu in allowed and phys is not None and core is not None:
                        seen.add((phys, core))
                    cpu = phys = core = None
        return len(seen) or usable_cores()
    except Exception:       # noqa: BLE001
        return usable_cores()


def cpu_baselines(B, H):
    """Three CPU legs on this box's host cores (oracle/ = the checker, here as the thing timed):
    cpu_baseline                     — the OpenMP restatement over the WHOLE host matrix `H` (the metric's configuration at full
                                       size by default) on every physical core (SURVEY.md 8(d) ii);
    cpu_baseline_reference_faithful  — the reference's own algorithm (serial loops, densify, full SVD) on a 24k-cell sample,
                                       BLAS pinned to <= 64 threads, extrapolated linearly in cells (said so);
    cpu_baseline_c1_serial           — BASELINE.md section 3's C1 (2.7k x 32k, 7 %) whole, one thread."""
    import numpy as np
    import oracle
    from oracle import pca_oracle
    a = B.a
    oracle.lib()
    out = {}
    model, cores, phys = cpu_model(), usable_cores(), physical_cores()
    env_threads = int(os.environ.get("SRX_BENCH_CPU_THREADS", "0"))
    threads = max(1, env_threads or min(cores, phys))
    ip, idx, val, genes = H.ip, H.idx, H.val, H.genes
    # (ii) threaded: OpenMP over every loop + k x k covariance and a symmetric eigen-solve instead of the full SVD
    n2 = H.n
    m2 = oracle.Csr(n2, genes, ip, idx, val)
    t0 = time.perf_counter()
    vals, hv, order, cov, mean, sd, secs = oracle.omp_pipeline(m2, a.target_sum, a.hvg, threads)
    t1 = time.perf_counter()
    import scipy.linalg as sla
    k = cov.shape[0]
    with _blas_limit(min(threads, 64)):
        w, v = sla.eigh(cov, subset_by_index=[max(0, k - a.npc), k - 1])
    t2 = time.perf_counter()
    # scores = Z V over every cell (OpenMP, omp_baseline.c::orc_omp_scores)
    oracle.omp_scores(m2, vals, order, v / sd[:, None], (mean / sd) @ v, threads)
    t3 = time.perf_counter()
    del vals
    out["cpu_baseline"] = {
        "value": n2 / (t3 - t0), "unit": "cells/s", "cores": threads, "kind": "port", "cpu_model": model, "host_cores": cores,
        "host_physical_cores": phys,
        "sample": f"{'the WHOLE matrix' if n2 >= (B.a.cells or CONFIGS[B.a.config][0]) else 'the first ' + str(n2) + ' cells'} "
                  f"({n2} cells, {len(val)} nnz), timed whole: oracle/omp_baseline.c on {threads} OpenMP threads — normalise + log1p "
                  f"{secs[0]:.2f} s, moments + HVG {secs[1]:.2f} s, k x k Gram {secs[2]:.2f} s — then LAPACK eigh of the {k} x {k} "
                  f"covariance (top {a.npc}, <= {min(threads, 64)} BLAS threads) {t2 - t1:.2f} s and the scores of every cell (OpenMP) "
                  f"{t3 - t2:.2f} s; SURVEY.md 8(d) variant (ii): a restatement, algorithmically cheaper than the reference's full "
                  "SVD, i.e. a baseline that favours the CPU",
        "seconds": t3 - t0}
    ip, idx, val = H.head(a.cpu_sample_cells)
    # (i) reference-faithful: the serial loops + densify + full-SVD PCA (numpy / LAPACK) on a bounded sample
    n1 = min(a.cpu_sample_cells, len(ip) - 1)
    e1 = int(ip[n1])
    m1 = oracle.Csr(n1, genes, ip[:n1 + 1], idx[:e1], val[:e1])
    threads = min(threads, 64)
    with _blas_limit(threads):
        t0 = time.perf_counter()
        lg = oracle.log1p_transform(oracle.normalize_total(m1, a.target_sum, oracle.ROW))
        t1 = time.perf_counter()
        sel = pca_oracle.select_features_hvg(lg, a.hvg)
        t2 = time.perf_counter()
        pca_oracle.pca_inplace(lg, a.npc, None, None, sel)
        t3 = time.perf_counter()
    out["cpu_baseline_reference_faithful"] = {
        "value": n1 / (t3 - t0), "unit": "cells/s", "cores": threads, "kind": "port", "cpu_model": model,
        "sample": f"first {n1} cells ({e1} nnz): serial C restatement of normalize_total + log1p ({t1 - t0:.2f} s) and nz-variance "
                  f"HVG({a.hvg}) ({t2 - t1:.2f} s) on 1 thread — the reference's loops are serial —, densify + full-SVD PCA via "
                  f"numpy / LAPACK pinned to {threads} BLAS threads ({t3 - t2:.2f} s).  EXTRAPOLATED: the exact SVD is linear in "
                  "cells at fixed k, so the rate carries to the full size; not the headline baseline",
        "seconds": t3 - t0}
    # 2.7k x 32k at 7 %, the reference's own CPU-runnable case, whole, on ONE thread
    c1_cells, c1_genes, c1_density, c1_seed = 2700, 32000, 0.07, 1001
    p1 = B.F.SynthParams()
    B.lib.srx_synth_defaults(C.byref(p1), c1_seed, c1_cells, c1_genes, c1_density)
    ip1 = np.zeros(c1_cells + 1, dtype=np.uint64)
    B.lib.srx_synth_indptr(C.byref(p1), 0, c1_cells, B.F.ptr(ip1))
    idx1 = np.zeros(int(ip1[-1]), np.uint64)
    val1 = np.zeros(int(ip1[-1]), np.float32)
    B.lib.srx_synth_fill_host(C.byref(p1), 0, c1_cells, B.F.ptr(ip1), B.F.ptr(idx1), B.F.ptr(val1))
    mc = oracle.Csr(c1_cells, c1_genes, ip1, idx1, val1)
    with _blas_limit(1):
        t0 = time.perf_counter()
        lg = oracle.log1p_transform(oracle.normalize_total(mc, a.target_sum, oracle.ROW))
        t1 = time.perf_counter()
        sel = pca_oracle.select_features_hvg(lg, a.hvg)
        t2 = time.perf_counter()
        pca_oracle.pca_inplace(lg, a.npc, None, None, sel)
        t3 = time.perf_counter()
    out["cpu_baseline_c1_serial"] = {
        "value": c1_cells / (t3 - t0), "unit": "cells/s", "cores": 1, "kind": "port", "cpu_model": model,
        "sample": f"BASELINE.json configs[0] / BASELINE.md section 3 C1 whole: {c1_cells} x {c1_genes}, {int(ip1[-1])} nnz, seed "
                  f"{c1_seed}; reference-faithful serial restatement on 1 thread: normalise + log1p {t1 - t0:.3f} s, HVG({a.hvg}) "
                  f"{t2 - t1:.3f} s, densify + full-SVD PCA (1 BLAS thread) {t3 - t2:.2f} s",
        "seconds": t3 - t0}
    return out


def backed_run(B, config=None, cells_override=0, n_runs=None):
    """configs[4] as specified: the matrix lives in pinned host memory and goes through the backed session as row tiles (two
    sweeps: statistics, then compaction + Gram), H2D overlapped with the kernels of the previous tile.  N ranks: every rank
    holds and streams ITS row range of the same matrix (nnz-balanced cut; 72 GB / N of host CSR each) through its own session;
    the per-gene moments and the packed Gram triangle are summed over the ranks inside srx_backed_select / srx_backed_solve.
    Returns the JSON object of the run on rank 0, None elsewhere (`--backed` prints it; the default line carries it as the
    `c5_backed` block)."""
    import numpy as np
    a, F, lib, ctx = B.a, B.F, B.lib, B.ctx
    config = config or a.config
    cells, genes, density, seed = CONFIGS[config]
    if cells_override:
        cells = cells_override
    p = B.params(config, cells)
    row0, row1 = B.shard(p, cells)
    rows = row1 - row0
    tile = a.tile_rows
    t_gen = time.perf_counter()
    ip = np.zeros(rows + 1, dtype=np.uint64)
    lib.srx_synth_indptr(C.byref(p), row0, row1, F.ptr(ip))            # offsets of this rank's rows, from 0
    nnz = int(ip[-1])
    # pinned host memory for the two big arrays (the session's H2D workers copy straight out of it)
    hidx, hval = C.c_void_p(), C.c_void_p()
    hip = C.CDLL("libamdhip64.so")
    hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    pinned = hip.hipHostMalloc(C.byref(hidx), max(nnz, 1) * 8, 0) == 0 and hip.hipHostMalloc(C.byref(hval), max(nnz, 1) * 4, 0) == 0
    if pinned:
        idx = np.ctypeslib.as_array(C.cast(hidx, C.POINTER(C.c_uint64)), shape=(nnz,))
        val = np.ctypeslib.as_array(C.cast(hval, C.POINTER(C.c_float)), shape=(nnz,))
    else:
        idx, val = np.zeros(nnz, np.uint64), np.zeros(nnz, np.float32)
    # fill tile by tile on a thread pool (the host generator is serial per call; ctypes releases the GIL)
    import concurrent.futures as cf

    def fill(r0):                                                  # r0: local row
        r1 = min(rows, r0 + tile)
        e0, e1 = int(ip[r0]), int(ip[r1])
        sub = (ip[r0:r1 + 1] - ip[r0]).astype(np.uint64)
        lib.srx_synth_fill_host(C.byref(p), row0 + r0, row0 + r1, F.ptr(sub), F.ptr(idx[e0:e1]), F.ptr(val[e0:e1]))
    with cf.ThreadPoolExecutor(max_workers=max(1, min(usable_cores() // max(B.world, 1), 64))) as ex:
        list(ex.map(fill, range(0, rows, tile)))
    t_gen = time.perf_counter() - t_gen
    host_bytes = ip.nbytes + idx.nbytes + val.nbytes
    opts = F.PcaOpts(a.npc, -1, -1, -1, 0, 0, 1, 0.0, 12345)
    xf = F.BACKED_NORMALIZE | F.BACKED_LOG1P

    def tiles():
        for r0 in range(0, rows, tile):
            r1 = min(rows, r0 + tile)
            e0 = int(ip[r0])
            yield F.Csr(r1 - r0, genes, int(ip[r1]) - e0, ip[r0:].ctypes.data, idx[e0:].ctypes.data, val[e0:].ctypes.data, F.F32)

    def tmax(x):                                                   # the slowest rank's time
        return B.dist.allreduce_max(x) if B.dist is not None else x

    runs = []
    for _ in range(max(1, n_runs if n_runs is not None else a.steps)):
        h = C.c_void_p()
        F.check(lib.srx_backed_create(ctx.handle, genes, F.STORE_F32, C.byref(h)), ctx.handle)
        B.sync_all()
        t0 = time.perf_counter()
        for t in tiles():
            F.check(lib.srx_backed_stats_tile(h, C.byref(t), a.target_sum, xf, None, None), ctx.handle)
        t1 = time.perf_counter()
        n_out = C.c_uint64()
        sel = np.zeros(a.hvg, np.uint64)
        F.check(lib.srx_backed_select(h, a.hvg, None, 0, C.byref(opts), F.ptr(sel), C.byref(n_out)), ctx.handle)
        t2 = time.perf_counter()
        for t in tiles():
            F.check(lib.srx_backed_gram_tile(h, C.byref(t), a.target_sum, xf), ctx.handle)
        t3 = time.perf_counter()
        info = F.PcaInfo()
        F.check(lib.srx_backed_solve(h, C.byref(info)), ctx.handle)
        B.sync_all()
        t4 = time.perf_counter()
        lib.srx_backed_destroy(h)
        runs.append({"total_s": tmax(t4 - t0), "sweep1_s": tmax(t1 - t0), "select_s": tmax(t2 - t1), "sweep2_s": tmax(t3 - t2),
                     "solve_s": tmax(t4 - t3), "residual": float(info.residual), "iterations": int(info.n_iter),
                     "cells_global_seen": int(info.n_cells_global)})
    best = min(runs, key=lambda r: r["total_s"])
    out = {
        "metric": "cells/sec end-to-end normalise->HVG->50-PC PCA; SpMM achieved HBM GB/s vs peak",
        "value": cells / best["total_s"], "unit": "cells/s", "n_gpus": B.world, "steps": len(runs), "warmup": 0,
        "ms_per_step": best["total_s"] * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{config} OUT OF CORE: {cells} cells x {genes} genes, density {density}, seed {seed}; the CSR "
                               f"(rank 0: {host_bytes / 1e9:.1f} GB: u64 offsets / indices + f32 values) stays in "
                               f"{'pinned' if pinned else 'pageable'} host memory and is streamed as {tile}-cell tiles "
                               "through srx_backed_* (sweep 1: statistics; select; sweep 2: compaction + Gram; solve)"
                               + (f"; rows {row0}..{row1} on rank 0 of {B.world}, moments and Gram triangle all-reduced" if B.world > 1 else ""),
                   "cells_global": cells, "genes": genes, "nnz_rank0": nnz, "hvg": a.hvg, "n_pc": a.npc, "tile_rows": tile,
                   "parallelism": f"row-shard x{B.world} (nnz-balanced), one backed session per rank",
                   **(B.comm_info if B.dist is not None and B.comm_info else {})},
        "h2d": {"host_bytes_per_sweep_rank0": host_bytes, "GBps_sweep1_rank0": host_bytes / best["sweep1_s"] / 1e9,
                "GBps_sweep2_rank0": host_bytes / best["sweep2_s"] / 1e9,
                "note": "both sweeps cross PCIe (u64 indices narrowed to 16 bits on the host side of the link: 6 of a non-zero's 12 bytes cross it); upload-bound: the "
                        "kernels of a tile run under the upload of the next one"},
        "runs": runs, "setup": {"host_generate_s": t_gen},
    }
    del idx, val
    if pinned:
        hip.hipHostFree.argtypes = [C.c_void_p]
        hip.hipHostFree(hidx); hip.hipHostFree(hval)
    return out if B.rank == 0 else None


MAX_LINE_BYTES = 6000         # the stdout record's bound (asserted; tests/test_bench_record_cpu.py)


def _r(x, sig=5):
    """Round a float to `sig` significant digits (the record is a record, not a dump of doubles)."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float(f"{x:.{sig}g}")


def _pick(d, keys):
    d = d or {}
    return {k_: _r(d.get(k_)) for k_ in keys}


def compact_record(full):
    """The one stdout line: the contract's fields + roofline (dominant kernel) + roofline_next + roofline_spmm + cpu_baseline
    + a handful of scalars.  No prose beyond `config.workload`, the kernel names and a <= 160-character CPU sample.  Everything
    else of `full` lives in bench_full.json."""
    cfg = full.get("config") or {}
    keep_cfg = ("workload", "cells_global", "genes", "nnz_rank0", "hvg", "n_pc", "panel_width", "parallelism", "collective",
                "kind", "rccl_version", "n_ranks", "n_ranks_seen", "launcher", "gram_exchange_split_launches",
                "gram_exchange_cu_masked", "nnz_hvg_compacted_rank0", "pca_residual", "pca_solver", "cold_step_ms", "prepare_ms",
                "f64_storage_ms_per_step", "incl_h2d_cells_per_s", "incl_h2d_pageable_cells_per_s", "skewed_genes_ms_per_step",
                "skewed_genes_ms_min", "skewed_genes_ms_median",
                "hard_spectrum_ms_per_step", "gram_formation_ms_per_step", "iterate_ms_per_step", "predicted_speedup_8_gpus",
                "shard_step_ms_at_8_ranks", "tile_rows")
    config = {k_: _r(cfg[k_]) for k_ in keep_cfg if cfg.get(k_) is not None}
    config["workload"] = str(config.get("workload", ""))[:400]
    its = cfg.get("subspace_iterations")
    if its:
        config["subspace_iterations"] = int(max(its))
    # value / ms_per_step at full precision (value == cells / ms_per_step must hold to the last digit)
    rec = {k_: full.get(k_) for k_ in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                       "scaling", "vs_baseline", "dtype", "data")}
    rec["config"] = config
    roof_keys = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_ms", "alg_bytes_per_launch")
    if full.get("roofline"):
        rec["roofline"] = _pick(full["roofline"], roof_keys)
        rec["roofline"]["kernel"] = str(rec["roofline"].get("kernel"))[:80]
    if full.get("roofline_next"):
        rec["roofline_next"] = _pick(full["roofline_next"], ("kernel", "frac", "avg_ms", "traffic"))
        rec["roofline_next"]["kernel"] = str(rec["roofline_next"].get("kernel"))[:80]
    if full.get("roofline_spmm"):
        rec["roofline_spmm"] = _pick(full["roofline_spmm"], ("kernel", "frac", "achieved", "avg_ms", "traffic", "alg_bytes_per_launch"))
        rec["roofline_spmm"]["kernel"] = str(rec["roofline_spmm"].get("kernel"))[:80]
    if full.get("kernel_ms_per_step"):
        rec["kernel_ms_per_step"] = {k_: _r(v, 4) for k_, v in full["kernel_ms_per_step"].items()}
    if full.get("step_roofline"):
        rec["step_frac_of_hbm_peak"] = _r(full["step_roofline"].get("frac_of_peak"))
    cb = full.get("cpu_baseline")
    if cb:
        rec["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "cpu_model", "seconds"))
        rec["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:160]
    for k_ in ("gpu_over_cpu",):
        if full.get(k_) is not None:
            rec[k_] = _r(full[k_])
    cold = full.get("cold_step") or {}
    if cold.get("value"):
        rec["cold_cells_per_s"] = _r(cold["value"])
    if full.get("unattributed_ms_per_step") is not None:
        rec["unattributed_ms_per_step"] = _r(full["unattributed_ms_per_step"], 4)
    if full.get("weak"):
        rec["weak"] = {k_: full["weak"].get(k_) for k_ in ("scaling", "cells_global", "steps", "ms_per_step", "value", "unit")}
    if full.get("runs"):         # --backed: the best run's clocks
        best = min(full["runs"], key=lambda r: r["total_s"])
        rec["runs"] = [{k_: _r(v) for k_, v in best.items()}]
    if full.get("h2d"):          # --backed
        rec["h2d"] = _pick(full["h2d"], ("host_bytes_per_sweep_rank0", "GBps_sweep1_rank0", "GBps_sweep2_rank0"))
    rec["full"] = "bench_full.json (+ stderr)"
    # never lose the line to its own bound: optional members go first (cannot happen with the fields above; a guard, not a path)
    for drop in ("kernel_ms_per_step", "runs", "h2d", "weak", "roofline_next", "roofline_spmm"):
        if len(json.dumps(rec)) <= MAX_LINE_BYTES:
            break
        rec.pop(drop, None)
    return rec


def self_launch(n):
    """`python bench.py --gpus N` typed as it stands (no launcher): this process becomes the launcher — N ranks of this very
    command, one per GPU, with the environment torch.distributed.run would give them (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT); rank 0 inherits stdout (the one JSON line), the others' stdout goes to stderr.  Any rank
    failing fails the launch (non-zero exit) and takes the others down (by PID)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               TORCHELASTIC_RUN_ID=f"self{os.getpid()}", SRX_BENCH_LAUNCHER="self")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.abspath(__file__), *sys.argv[1:]]
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                      stdout=None if r == 0 else sys.stderr.fileno()))
    rc = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"[bench launcher] rank {r} exited with {code}: stopping the other ranks", file=sys.stderr)
                for o in pending:
                    procs[o].terminate()
        if pending:
            time.sleep(0.05)
    sys.exit(rc)


def main():
    a = parse()
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != a.gpus:
        if world_env == 1 and a.gpus > 1 and "RANK" not in os.environ:
            self_launch(a.gpus)
        a.gpus = world_env
    B = Bench(a)
    rank, world = B.rank, B.world
    if a.backed:
        out = backed_run(B, cells_override=a.cells)
        if rank == 0:
            print("[bench full] " + json.dumps(out), file=sys.stderr, flush=True)
            line = json.dumps(compact_record(out))
            if B.json_fd is not None:
                sys.stdout.flush()
                os.write(B.json_fd, (line + "\n").encode())
            else:
                print(line, flush=True)
        B.close()
        return
    cells, genes, density, seed = CONFIGS[a.config]
    if a.cells:
        cells = a.cells
    scaling = a.scaling if a.scaling != "auto" else "strong"      # (N = 1: the label of the N > 1 runs of the same command)
    n_global = cells if scaling == "strong" or world == 1 else cells * world
    p = B.params(a.config, n_global, a.skew)
    row0, row1 = B.shard(p, n_global)
    main_ = B.run(a.config, n_global, row0, row1, a.storage, a.steps, a.warmup, solver=a.solver, skew=a.skew, headline=True)
    weak = None
    if world > 1 and scaling == "strong" and not a.lean:
        # the other reading of the scaling question: per-GPU work fixed (the config's cells on EVERY GPU)
        ng = cells * world
        pw = B.params(a.config, ng, a.skew)
        w0, w1 = B.shard(pw, ng)
        wk = min(a.steps, 5)
        w = B.run(a.config, ng, w0, w1, a.storage, wk, min(a.warmup, 2), solver=a.solver, skew=a.skew)
        weak = {"scaling": "weak", "cells_global": ng, "steps": wk, "ms_per_step": w["ms_per_step"],
                "value": ng * wk / w["elapsed"], "unit": "cells/s", "subspace_iterations": w["iters"]}

    extra = {}
    if rank == 0 and world == 1 and not a.lean:
        def attempt(name, fn):
            try:
                extra[name] = fn()
            except Exception as e:      # an extra block is never a reason to lose the headline line
                extra[name] = {"failed": repr(e)}
        k5 = min(a.steps, 5)
        other = "f64" if a.storage == "f32" else "f32"

        def f_other():
            r = B.run(a.config, n_global, 0, n_global, other, k5, 1, solver=a.solver)
            per, serial = attributed(r["prof"], k5)
            return {"storage": other, "steps": k5, "ms_per_step": r["ms_per_step"], "value": n_global * k5 / r["elapsed"],
                    "unit": "cells/s", "pca_residual": r["residual"], "subspace_iterations": r["iters"],
                    "kernel_ms_per_step": per,
                    "note": "the same pipeline with f64 value storage (12 B per non-zero, 16-byte compacted records): the "
                            "reference's f64 arithmetic to ~1e-15; HighlyVariable(n) is the reference's at either storage"}
        attempt("f64_storage" if other == "f64" else "f32_storage", f_other)
        attempt("cold_step", lambda: cold_step(B, a.config, n_global, a.storage))

        def f_iter():
            r = B.run(a.config, n_global, 0, n_global, a.storage, 2, 1, solver=2)
            fw, tr = r["prof"].get("spmm_fwd", {}), r["prof"].get("spmm_t", {})
            return {"steps": 2, "ms_per_step": r["ms_per_step"], "subspace_iterations": r["iters"], "pca_residual": r["residual"],
                    "spmm_fwd": roof(fw, KERNEL_SYMBOL["spmm_fwd"], "f64 panels (the matrix-free iteration runs its products in f64)"),
                    "spmm_t": roof(tr, KERNEL_SYMBOL["spmm_t"], "N m 64 f64 LDS lane-atomics per launch; without them the launch takes the same time (profiles/history.md): ~8 instructions per non-zero for 64 multiply-adds"),
                    "note": "--solver 2: the matrix-free subspace iteration, one forward and one transposed SpMM per application of "
                            "C — the kernels BASELINE.json's second metric means; the default Gram solver runs the forward "
                            "SpMM once per solve (roofline_spmm)"}
        attempt("roofline_spmm_iter", f_iter)

        def f_skew():
            # (3 steps after 1 warm-up read 17.8 on one box and 19.6 on another — 10 steps, each between its own
            #  synchronisations, minimum and median reported)
            r = B.run(a.config, n_global, 0, n_global, a.storage, 10, 2, solver=a.solver, skew=1 - a.skew, tolerate_noconv=True, per_step=True)
            per, _ = attributed(r["prof"], 10)
            sm = r["step_ms"]
            return {"skew": 1 - a.skew, "steps": 10, "ms_per_step": r["ms_per_step"], "kernel_ms_per_step": per,
                    "step_ms_min": sm[0] if sm else None, "step_ms_median": sm[len(sm) // 2] if sm else None,
                    "nnz_hvg_compacted": r["nnz_selected"], "noconv_steps": r["noconv_steps"],
                    "note": "srx_synth skew = 1: gene density ~ 1/sqrt(gene index) (7x between the first and the last genes): "
                            "contention on the per-gene LDS accumulators of the moments pass, owners of very different weight "
                            "in the Gram kernel (stripes are cut by row count, not by measured work)"}
        attempt("skewed_genes", f_skew)

        def f_hard():
            c2 = CONFIGS["c2"]
            r = B.run("c2", c2[0], 0, c2[0], a.storage, 3, 1, hvg=1000, tolerate_noconv=True)
            return {"workload": "c2 (100k x 20k, 5 %), HighlyVariable(1000): theta_64 / theta_50 = 0.97-0.99 (flat tail)",
                    "steps": 3, "ms_per_step": r["ms_per_step"], "subspace_iterations": r["iters"], "pca_residual": r["residual"],
                    "noconv_steps": r["noconv_steps"]}
        attempt("hard_spectrum", f_hard)

        def f_budget():
            # The shard one rank holds at P = 2 / 4 / 8 ranks (the first n / P cells), timed here on ONE GPU without communication:
            # the measured input of the strong-scaling prediction — the sharded kernels shrink with 1 / P, the
            # replicated part (selection, the k x 64 iteration, launch gaps) does not.  NOT a multi-GPU measurement.
            out = {"note": "per-rank compute of the shard a rank holds at P ranks, on one GPU, no communication; predicted_speedup adds "
                           "exchange_ms (moments 0.05 + the exposed half of the 16 MB Gram ring at ~77 GB/s per direction + one host "
                           "read) and divides the 1-GPU step by the sum: a budget, not a measurement"}
            for P in (2, 4, 8):
                n_p = n_global // P
                r = B.run(a.config, n_global, 0, n_p, a.storage, 5, 2, solver=a.solver)
                per, _ = attributed(r["prof"], 5)
                exch = 0.05 + (P - 1) / P * 16.0e6 * 0.5 / 77e9 * 1e3 + 0.03
                out[str(P)] = {"cells": n_p, "ms_per_step": r["ms_per_step"], "iterate_ms": per.get("iterate"),
                               "exchange_ms": exch, "predicted_speedup": main_["ms_per_step"] / (r["ms_per_step"] + exch)}
            return out
        attempt("strong_scaling_budget", f_budget)

        if not a.no_cpu_baseline:
            # the host-side legs at the metric's configuration in FULL (--host-sample-cells 0, the default): the whole matrix
            # in the reference layout on the host — H2D-inclusive rate from pinned and from pageable caller buffers, the
            # threaded CPU baseline over every cell
            H = None
            try:
                ns = min(a.host_sample_cells or n_global, n_global)
                H = HostMatrix(B, a.config, ns, pinned=True)
                full = "the whole matrix" if ns == n_global else f"the first {ns} cells"
                attempt("incl_h2d", lambda: incl_h2d(B, H, f"{full}, caller buffers in pinned host memory"))
                if H.pinned and not os.environ.get("SRX_BENCH_NO_PAGEABLE"):
                    def f_pageable():
                        import numpy as np
                        Hp = HostMatrix.__new__(HostMatrix)
                        Hp.n, Hp.genes, Hp.ip, Hp.pinned, Hp.host_bytes = H.n, H.genes, H.ip, False, H.host_bytes
                        Hp.idx, Hp.val = np.array(H.idx), np.array(H.val)          # plain (pageable) copies: a Rust Vec
                        r = incl_h2d(B, Hp, f"{full}, caller buffers in PAGEABLE host memory")
                        del Hp
                        return r
                    attempt("incl_h2d_pageable", f_pageable)
                try:
                    extra.update(cpu_baselines(B, H))
                except Exception as e:      # the baseline is a reported number, never a reason to lose the GPU line
                    extra["cpu_baseline"] = {"value": None, "unit": "cells/s", "cores": usable_cores(), "kind": "port",
                                             "sample": f"failed: {e!r}"}
            except Exception as e:
                extra["incl_h2d"] = {"failed": repr(e)}
            finally:
                if H is not None:
                    H.free()

        if a.config == "c3" and not a.cells and not os.environ.get("SRX_BENCH_NO_C5"):
            def f_c5():
                r = backed_run(B, config="c5", n_runs=1)
                return {k_: r[k_] for k_ in ("value", "unit", "ms_per_step", "config", "h2d", "runs", "setup")}
            attempt("c5_backed", f_c5)

    if rank == 0:
        prof = main_["prof"]
        value = n_global * a.steps / main_["elapsed"]
        traffic, traffic_round = load_traffic(a.config)
        for name, d in prof.items():
            per_step = traffic.get(name)
            d["hbm_traffic_per_launch"] = per_step / (d["launches"] / a.steps) if per_step else None
        per, serial = attributed(prof, a.steps)
        # the step as a whole against the HBM peak: algorithmic bytes of every serial class (dense_apply sits inside iterate)
        serial_cls = [k_ for k_ in prof if k_ not in ("dense_apply", "normalize_log1p")]
        step_alg = sum(prof[k_]["alg_bytes_per_launch"] * prof[k_]["launches"] / a.steps for k_ in serial_cls)
        step_aux = sum(prof[k_].get("aux_bytes_per_launch", 0.0) * prof[k_]["launches"] / a.steps for k_ in serial_cls)
        step_traffic = sum(traffic[k_] for k_ in serial_cls if traffic.get(k_)) if traffic else None
        # `roofline` names the dominant kernel of the largest STAGE of the step, `roofline_next` that of the second largest —
        # stage sums, not single classes: the Gram kernel and the moments pass are within 1 % of each other as classes and the
        # choice used to flip from run to run; as stages (Gram formation = compaction + owner records + stripe
        # kernel against normalise + moments = row sums + moments pass) they are 1.5 ms apart.
        STAGES = {"gram_formation": ("hvg_compact", "gram_bucket", "gram_sparse"), "normalise_moments": ("row_sums", "gene_moments"),
                  "transform": ("spmm_fwd", "spmm_t")}
        stage_ms = {sn: sum(per.get(c, 0.0) for c in cls) for sn, cls in STAGES.items()}
        order = sorted(stage_ms, key=stage_ms.get, reverse=True)

        def stage_block(sn):
            cls = [c for c in STAGES[sn] if c in prof]
            alg = sum(prof[c]["alg_bytes_per_launch"] * prof[c]["launches"] / a.steps for c in cls)
            aux = sum(prof[c].get("aux_bytes_per_launch", 0.0) * prof[c]["launches"] / a.steps for c in cls)
            trf = sum(traffic[c] for c in cls if traffic.get(c)) if traffic and all(traffic.get(c) for c in cls) else None
            t = stage_ms[sn] * 1e-3
            return {"name": sn, "classes": {c: per.get(c) for c in cls}, "ms_per_step": stage_ms[sn], "alg_bytes_per_step": alg,
                    "aux_bytes_per_step": aux, "achieved_GBps": alg / t / 1e9 if t else None,
                    "frac": alg / t / 1e9 / HBM_PEAK_GBS if t else None, "hbm_traffic_per_step": trf,
                    "traffic_over_alg": trf / alg if trf and alg else None}

        def dominant(sn):
            cls = {c: per.get(c, 0.0) for c in STAGES[sn] if c in prof}
            return max(cls, key=cls.get) if cls else None
        dom_name = dominant(order[0]) if order else None
        dom = prof.get(dom_name, {})
        dom2_name = dominant(order[1]) if len(order) > 1 else None
        gram_stage = stage_block("gram_formation")
        out = {
            "metric": "cells/sec end-to-end normalise->HVG->50-PC PCA; SpMM achieved HBM GB/s vs peak",
            "value": value, "unit": "cells/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": main_["ms_per_step"], "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": a.storage, "data": "synthetic",
            "config": {
                "workload": f"{a.config}: {n_global} cells x {genes} genes, density {density}, seed {seed}"
                            f"{', skewed gene densities' if a.skew else ''}; normalize_total(1e4,Row)+log1p+HVG({a.hvg})+"
                            f"{a.npc}-PC PCA; values {a.storage} / indices i32 in HBM; rows {row0}..{row1} on rank 0",
                "cells_global": n_global, "genes": genes, "nnz_rank0": main_["nnz"], "hvg": a.hvg, "n_pc": a.npc,
                "panel_width": 64, "parallelism": f"row-shard x{world} (nnz-balanced)",
                **(B.comm_info if B.dist is not None else {}),
                **(B.overlap_info() if B.dist is not None else {}),
                "nnz_hvg_compacted_rank0": main_["nnz_selected"],
                "subspace_iterations": main_["iters"], "pca_residual": main_["residual"], "pca_solver": main_["solver"],
                "hvg_selection": "f64 moments of ln_1p(v * scale) from the raw matrix: identical to the reference's at f32 storage",
                # the ~80 small launches of the subspace iteration are replayed from captured hipGraphs (three
                # segments per solve); they are bracketed as ONE class (`iterate`)
                "pca_iteration_hip_graphs": main_["solver"] == "gram" and not os.environ.get("SRX_NO_GRAPH"),
                # the side measurements of this run that belong beside `value` (full blocks further down the line)
                "cold_step_ms": (extra.get("cold_step") or {}).get("ms"),
                "prepare_ms": (extra.get("cold_step") or {}).get("prepare_ms"),
                "f64_storage_ms_per_step": (extra.get("f64_storage") or {}).get("ms_per_step"),
                "incl_h2d_cells_per_s": (extra.get("incl_h2d") or {}).get("value"),
                "incl_h2d_cells": (extra.get("incl_h2d") or {}).get("cells"),
                "incl_h2d_pageable_cells_per_s": (extra.get("incl_h2d_pageable") or {}).get("value"),
                "cpu_baseline_cells_per_s": (extra.get("cpu_baseline") or {}).get("value"),
                "cpu_baseline_threads": (extra.get("cpu_baseline") or {}).get("cores"),
                "cold_cells_per_s": (extra.get("cold_step") or {}).get("value"),
                "roofline_spmm_frac": (prof.get("spmm_fwd") or {}).get("frac_of_peak"),
                "roofline_spmm_ms": (prof.get("spmm_fwd") or {}).get("avg_ms"),
                "skewed_genes_ms_per_step": (extra.get("skewed_genes") or {}).get("ms_per_step"),
                "skewed_genes_ms_min": (extra.get("skewed_genes") or {}).get("step_ms_min"),
                "skewed_genes_ms_median": (extra.get("skewed_genes") or {}).get("step_ms_median"),
                "hard_spectrum_ms_per_step": (extra.get("hard_spectrum") or {}).get("ms_per_step"),
                "gram_formation_ms_per_step": gram_stage["ms_per_step"],
                "gram_formation_traffic_over_alg": gram_stage["traffic_over_alg"],
                "iterate_ms_per_step": per.get("iterate"),
                "predicted_speedup_8_gpus": ((extra.get("strong_scaling_budget") or {}).get("8") or {}).get("predicted_speedup"),
                "shard_step_ms_at_8_ranks": ((extra.get("strong_scaling_budget") or {}).get("8") or {}).get("ms_per_step"),
                "scaling_note": ("one GPU: `scaling` carries the label the same command gives at N > 1 (strong: BASELINE.json "
                                 "configs[3], the same cells row-sharded), so that the per-N lines of a scaling run read alike"
                                 if world == 1 else None),
            },
            # the kernel class with the largest share of the step (live HIP-event timing on the stream it runs on)
            "roofline": {**roof(dom, KERNEL_SYMBOL.get(dom_name, dom_name), ROOF_NOTE.get(dom_name, ""),
                                other_bounds(dom_name, dom, main_["nnz_selected"], row1 - row0)),
                         "stage": stage_block(order[0]) if order else None},
            # the dominant kernel of the second largest stage
            "roofline_next": ({**roof(prof.get(dom2_name, {}), KERNEL_SYMBOL.get(dom2_name, dom2_name), ROOF_NOTE.get(dom2_name, ""),
                                      other_bounds(dom2_name, prof.get(dom2_name, {}), main_["nnz_selected"], row1 - row0)),
                               "stage": stage_block(order[1])} if dom2_name else None),
            # BASELINE.json's second metric: the CSR x 64-column-panel SpMM against the HBM peak
            "roofline_spmm": roof(prof.get("spmm_fwd", {}), KERNEL_SYMBOL["spmm_fwd"], ROOF_NOTE["spmm_fwd"],
                                  other_bounds("spmm_fwd", prof.get("spmm_fwd", {}), main_["nnz_selected"], row1 - row0)),
            "kernels": prof,
            "kernel_ms_per_step": per,
            # step time outside every bracketed class (launch gaps, host waits, scans, small copies) — of the breakdown pass, whose
            # step carries all the class timers (its own step time minus its own class times); the timed steps carry three
            "unattributed_ms_per_step": ((main_["breakdown"]["ms_per_step"] - sum(v for k_, v in main_["breakdown"]["kernel_ms_per_step"].items()
                                                                                  if k_ not in ("dense_apply", "normalize_log1p")))
                                         if main_.get("breakdown") else main_["ms_per_step"] - serial),
            "class_timers": ({"timed_region": ["hvg_compact", "gram_bucket", "gram_sparse"], "breakdown_pass": main_["breakdown"]}
                             if main_.get("breakdown") else None),
            "step_roofline": {"step_alg_bytes": step_alg, "step_aux_bytes": step_aux, "step_hbm_traffic": step_traffic,
                              "achieved_GBps": step_alg / (main_["ms_per_step"] * 1e-3) / 1e9,
                              "frac_of_peak": step_alg / (main_["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "traffic_GBps": step_traffic / (main_["ms_per_step"] * 1e-3) / 1e9 if step_traffic else None,
                              "note": "algorithmic bytes of every kernel class of a step (minimum data each must move, SURVEY.md "
                                      "8(d)) over the whole step time; step_hbm_traffic = PMC counter traffic of the same classes "
                                      "from the committed profile"},
            "stage_ms_per_step": main_["stage_ms_per_step"],
            "traffic_source": f"profiles/{traffic_round}_traffic_{a.config}.json" if traffic_round else None,
            "setup": {"generate_s": main_["generate_s"], "copies": main_["copies"]},
        }
        if weak:
            out["weak"] = weak
        out.update(extra)
        if "cpu_baseline" in out and out["cpu_baseline"].get("value"):
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        # The full document (every note, per-kernel record, side workload) goes to bench_full.json + stderr; stdout gets
        # the RECORD: one line of <= 6 KB the driver parses (a 20.7 KB line came back `parsed: null`).
        full_text = json.dumps(out)
        for path in (os.path.join(ROOT, "bench_full.json"), os.path.join(ROOT, "gpurun_out", "bench_full.json")):
            try:
                if os.path.isdir(os.path.dirname(path)):
                    with open(path, "w") as fh:
                        fh.write(full_text + "\n")
            except OSError:
                pass
        print("[bench full] " + full_text, file=sys.stderr, flush=True)
        line = json.dumps(compact_record(out))
        assert len(line) <= MAX_LINE_BYTES, len(line)
        if B.json_fd is not None:
            sys.stdout.flush()
            os.write(B.json_fd, (line + "\n").encode())
        else:
            print(line, flush=True)
    B.close()


if __name__ == "__main__":
    main()
