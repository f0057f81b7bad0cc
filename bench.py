#!/usr/bin/env python3
"""Headline benchmark: grid-points/s of the complete_analysis sweep (BASELINE.json `metric`).

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path: ONE complete_analysis sweep of the BASELINE configs[1]
workload -- README hyperbolic model, args [m, phi0, L] = [1, 1, 1], extent (-1, 1, -1, 1),
8192 x 8192 field grid, six f64 per point (48 B) written to a device-resident (N0, N1, 6) array.
Multi-GPU (weak scaling, BASELINE configs[4] style): the outer *parameter* axis is sharded, every
rank sweeps the full grid for its own parameter row (L differs per rank); results stay on the
rank's GPU (no data-path collective: the rows are independent; the only collectives are the timing
barrier and the MAX over ranks).

Timed region: inputs (parameters, code object) resident on the device, result left in HBM.
PyTorch provides the device buffer, the stream and torch.distributed only; every launch goes
through the C ABI (libinflx_hip.so).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
BYTES_PER_POINT = 48  # 6 x f64 written, 0 read


def host_threads() -> int:
    """CPU threads this process may actually use: the cgroup CPU quota if one is set (a GPU box hands
    each GPU a share of the host), else the scheduler affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def recorded_traffic(kernel: str, model: str, n: int):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/), or None when
    no measurement of this exact workload is on record."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None
    if rec.get("workload", "").startswith(f"{model} {n}x{n} ") and kernel in rec:
        return rec[kernel]["traffic_bytes"]
    return None


def cpu_baseline(model_name: str, args, extent, budget_s: float = 12.0):
    """Oracle (C restatement of the reference's rayon sweep: five indirect calls per point into the
    gcc -O3 model object + ops::complete_analysis) on all host cores, bounded sample."""
    import numpy as np

    import oracle
    from inflatox_amd import example_models, workloads

    spec = example_models.get(model_name)
    src, _ = oracle.emit_c_source(workloads.model_for(model_name), **spec.compiler_kwargs)
    om = oracle.OracleModel(oracle.compile_c_model(src))
    cores = host_threads()
    n = 1024
    t0 = time.perf_counter()
    om.grid_sweep(oracle.OP.COMPLETE, args, extent, n, n, threads=cores)  # warm-up + calibration
    t_cal = time.perf_counter() - t0
    # choose the sample so that one pass takes about budget/3 seconds
    scale = max(1.0, (budget_s / 3.0) / max(t_cal, 1e-3))
    n = int(min(8192, 1024 * np.sqrt(scale)) // 64 * 64)
    best = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        om.grid_sweep(oracle.OP.COMPLETE, args, extent, n, n, threads=cores)
        best = min(best, time.perf_counter() - t0)
    return {
        "value": n * n / best,
        "unit": "grid-points/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{model_name} {n}x{n} grid over the same extent (per-point cost does not depend on grid size), best of 3 passes, "
        f"{cores} threads ({os.cpu_count()} logical CPUs on the host), gcc -O3 -march=native model object + C restatement of the Rust sweep",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="hyperbolic")
    ap.add_argument("--grid", dest="n", type=int, default=8192, help="grid points per axis")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    opt = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != opt.gpus:
        raise SystemExit(f"--gpus {opt.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {opt.gpus}")
    # INFLX_BENCH_FORCE_DIST=1 exercises the process-group path (init, barrier, MAX) with a single rank
    distributed = world > 1 or os.environ.get("INFLX_BENCH_FORCE_DIST") == "1"
    # Rehearsal knob for a one-GPU box: INFLX_BENCH_REHEARSE=1 lets several ranks share the visible GPUs
    # (rank -> device local_rank % count) and synchronise over gloo, because RCCL refuses two ranks on one
    # device.  Everything else -- sharding plan, per-rank sweeps, barrier, MAX over ranks -- is the real path.
    rehearse = os.environ.get("INFLX_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank %= max(1, torch.cuda.device_count())
    backend = "gloo" if rehearse else "nccl"
    comm_device = "cpu" if rehearse else f"cuda:{local_rank}"
    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL announces itself on stdout ("Librccl path : ..."); stdout carries the one JSON line only,
        # so point fd 1 at stderr while the process group comes up
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if rehearse:
                dist.init_process_group(backend)
            else:
                dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    from inflatox_amd import _native, workloads
    from inflatox_amd.distributed import plan_shard

    spec, art = workloads.artifact_for(opt.model)
    lib = _native.InflatoxDevLib(art.shared_object_path, device=local_rank)

    N0 = N1 = opt.n
    # outer parameter axis of `world` rows (the last parameter -- L for the hyperbolic model, as in
    # BASELINE configs[4] -- scaled over [1, 2)); plan_shard hands every rank its block: one row
    all_rows = np.tile(np.array(spec.args, dtype=np.float64), (world, 1))
    all_rows[:, -1] *= 1.0 + np.arange(world) / world
    plan = plan_shard(world, N0, world, rank)
    assert plan.axis == "param" and plan.p_count == 1 and plan.row_count == N0
    args = all_rows[plan.p_begin]
    out = torch.empty((N0, N1, 6), dtype=torch.float64, device=f"cuda:{local_rank}")
    # a stream of our own: torch's default stream has the NULL handle, which the C ABI reads as "the model's
    # own stream"; with an explicit one the HIP events below are recorded on the stream the kernels run on
    launch_stream = torch.cuda.Stream(device=f"cuda:{local_rank}")
    stream = launch_stream.cuda_stream
    nbytes = out.numel() * 8

    def step():
        lib.sweep_device(_native.OP_COMPLETE, args, out.data_ptr(), nbytes, spec.extent, N0, N1, stream=stream)

    for _ in range(opt.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP events on the launch stream bracket the same region (torch's current stream is the stream the
    # sweeps are enqueued on): device time per step, next to the wall-clock figure the value is computed from
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(launch_stream)
    for _ in range(opt.steps):
        step()
    ev1.record(launch_stream)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    step_ms_events = ev0.elapsed_time(ev1) / opt.steps
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant kernel, measured live with HIP events on the launch stream.  For a model whose values do
    # not depend on x[1] a sweep is two launches (per-row evaluation, ~4 % of the time, then the store
    # stream); the store stream is the dominant kernel and the one the roofline prices.
    row_path = lib.stage_info["out_mask"] & 2 == 0
    ms_kernel = lib.sweep_device_timed(_native.OP_COMPLETE, args, out.data_ptr(), nbytes, spec.extent, N0, N1, stream=stream, repeats=max(5, opt.steps), dominant_only=row_path)
    ms_sweep = lib.sweep_device_timed(_native.OP_COMPLETE, args, out.data_ptr(), nbytes, spec.extent, N0, N1, stream=stream, repeats=max(5, opt.steps))
    points = N0 * N1
    achieved = BYTES_PER_POINT * points / (ms_kernel * 1e-3) / 1e9

    # outside the timed region: the statistics path of a sharded sweep -- every rank reduces its own block
    # on the device (summary-only sweep), three six-element all-reduces combine the ranks (RCCL when N > 1)
    stats_info = None
    try:
        t0 = time.perf_counter()
        local = lib.sweep_stats(args, spec.extent, N0, N1)
        if distributed:
            from inflatox_amd.distributed import all_reduce_summary

            local = all_reduce_summary(local, device=comm_device)
        stats_info = {
            "ms": (time.perf_counter() - t0) * 1e3,
            "nanmax": [None if not np.isfinite(v) else float(v) for v in local["max"]],
            "non_nan": [int(v) for v in local["count"]],
        }
    except Exception as exc:  # noqa: BLE001 -- never let the optional extra break the benchmark line
        stats_info = {"error": str(exc)[:200]}

    if rank == 0:
        kernel = "inflx_sweep_rowstream6" if row_path else "inflx_sweep_tile_complete"
        line = {
            "metric": "grid-points/sec on complete_analysis sweep; achieved HBM GB/s vs peak",
            "value": world * points * opt.steps / elapsed,
            "unit": "grid-points/s",
            "n_gpus": world,
            "steps": opt.steps,
            "warmup": opt.warmup,
            "ms_per_step": elapsed / opt.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{opt.model} model, {N0}x{N1} field grid, args {spec.args.tolist()}, extent {list(spec.extent)}, complete_analysis (6 f64/point, AoS), device-resident result",
                "parameter_rows_per_gpu": 1,
                "parallelism": (f"parameter-axis x{world}" if world > 1 else "single GPU") + (" [REHEARSAL: ranks share GPUs, gloo]" if rehearse else ""),
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": recorded_traffic(kernel, opt.model, opt.n),
                "kernel": kernel,
                "kernel_ms": ms_kernel,
                "sweep_ms": ms_sweep,
                "timed_region_ms_per_step_hip_events": step_ms_events,
                "kernels_per_step": ["inflx_sweep_rowvals_complete", "inflx_sweep_rowstream6"] if row_path else ["inflx_sweep_tile_complete"],
                "algorithmic_bytes_per_launch": BYTES_PER_POINT * points,
            },
        }
        line["summary_sweep"] = stats_info
        if world == 1 and not opt.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(opt.model, spec.args, spec.extent)
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
