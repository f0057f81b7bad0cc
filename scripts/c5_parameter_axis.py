#!/usr/bin/env python3
"""BASELINE configs[4]: the hyperbolic 8192x8192 sweep over a parameter axis of P rows, sharded over the GPUs of
one node (one process per GPU; `python -m torch.distributed.run --nproc-per-node N scripts/c5_parameter_axis.py`).

Every rank takes its contiguous block of parameter rows (inflatox_amd.distributed.plan_shard), sweeps it with ONE
call into its own HBM (64 rows = 206 GB per GPU for P = 512 on 8 GPUs) and reduces it to the summary on the device;
the ranks then combine the summaries with three six-element all-reduces (RCCL).  The arrays themselves never leave
the GPU that computed them: gathering 1.65 TB is neither possible nor needed (Multi-GPU).

  --rows P        total parameter rows (default 64 x world size)
  --grid N        grid points per axis (default 8192)
Single process (no torchrun): world size 1, i.e. the per-GPU share of the 8-GPU job.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from inflatox_amd import _native  # noqa: E402
import workloads  # noqa: E402
from inflatox_amd.distributed import all_reduce_summary, plan_shard  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=None)
    ap.add_argument("--grid", type=int, default=8192)
    opt = ap.parse_args()
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
    # INFLX_BENCH_REHEARSE=1 (as in bench.py): ranks share the visible GPUs and talk over gloo -- a one-GPU rehearsal
    rehearse = os.environ.get("INFLX_BENCH_REHEARSE") == "1"
    if rehearse:
        local %= max(1, torch.cuda.device_count())
    comm = "cpu" if rehearse else f"cuda:{local}"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # before the first HIP call of the process
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist

        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    P = opt.rows or 64 * world
    n = opt.grid
    spec, art = workloads.artifact_for("hyperbolic")
    lib = _native.InflatoxDevLib(art.shared_object_path, device=local)
    rows = np.tile(np.array(spec.args, dtype=np.float64), (P, 1))
    rows[:, -1] = np.linspace(0.2, 2.0, P)  # L, as SURVEY 8(d) proposes for C5
    plan = plan_shard(P, n, world, rank)
    mine = rows[plan.p_begin : plan.p_begin + plan.p_count]
    out = torch.empty((plan.p_count, plan.row_count, n, 6), dtype=torch.float64, device=f"cuda:{local}")
    stream = torch.cuda.Stream(device=f"cuda:{local}")
    # warm-up: one parameter row of this rank's own block (the same row range as the timed call: a rank of a
    # row-sharded plan owns fewer than n rows, and the buffer is sized for its share)
    lib.sweep_device(_native.OP_COMPLETE, mine[:1], out.data_ptr(), out[:1].numel() * 8, spec.extent, n, n, row_begin=plan.row_begin, row_count=plan.row_count, stream=stream.cuda_stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    lib.sweep_device(_native.OP_COMPLETE, mine, out.data_ptr(), out.numel() * 8, spec.extent, n, n, row_begin=plan.row_begin, row_count=plan.row_count, stream=stream.cuda_stream)
    torch.cuda.synchronize()
    t_sweep = time.perf_counter() - t0
    t0 = time.perf_counter()
    local_summary = lib.sweep_stats(mine, spec.extent, n, n, row_begin=plan.row_begin, row_count=plan.row_count)
    total = all_reduce_summary(local_summary, device=comm) if world > 1 else local_summary
    t_stats = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([t_sweep], dtype=torch.float64, device=comm)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t_sweep = float(t.item())
    points = P * n * n if plan.axis == "param" else P * n * n
    if rank == 0:
        print(json.dumps({
            "workload": f"hyperbolic {n}x{n} x {P} parameter rows over {world} GPU(s), {plan.p_count} row(s) on rank 0 ({out.numel() * 8 / 1e9:.1f} GB resident)",
            "sweep_s": t_sweep, "points_per_s": points / t_sweep, "achieved_GBps_per_gpu": 48 * plan.p_count * plan.row_count * n / t_sweep / 1e9,
            "summary_s": t_stats, "non_nan": [int(v) for v in total["count"]], "nanmax": [None if not np.isfinite(v) else float(v) for v in total["max"]],
        }))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
