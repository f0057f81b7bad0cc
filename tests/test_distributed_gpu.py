"""The RCCL code path of the sharded sweep on a one-GPU box: ONE rank, backend "nccl", device tensors.

``ShardedSweep.run(gather=True, force_collective=True)`` sends the rank's block through ``all_gather_into_tensor`` and
``all_reduce_summary`` sends the device-reduced summary through its three all-reduces -- the very calls an 8-GPU job
makes (inflatox_amd/distributed.py), here with a world of one, so that RCCL itself has run them at least once before the
driver's multi-GPU bench does.  The worker runs in a process of its own (a process group lives and dies with it)."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker():
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import socket

    import numpy as np
    import torch
    import torch.distributed as dist

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    torch.cuda.set_device(0)
    real_stdout = os.dup(1)
    os.dup2(2, 1)  # RCCL announces itself on stdout
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dist.barrier()

    import workloads
    from inflatox_amd import _native
    from inflatox_amd.distributed import HipCompute, ShardedSweep, all_reduce_summary, numpy_summary

    spec, art = workloads.artifact_for("hyperbolic")
    lib = _native.InflatoxDevLib(art.shared_object_path, device=0)
    n0, n1, P = 96, 128, 3
    rows = np.tile(np.asarray(spec.args, dtype=np.float64), (P, 1))
    rows[:, -1] = [0.5, 1.0, 1.7]
    sweep = ShardedSweep(HipCompute(lib, spec.extent, n0, n1), rank=0, world=1)
    plan, local = sweep.run(rows, n0)  # the shortcut of a one-rank world: no collective
    plan2, full = sweep.run(rows, n0, gather=True, force_collective=True)  # through RCCL's all-gather
    torch.cuda.synchronize()
    assert full.is_cuda and tuple(full.shape) == (P, n0, n1, 6)
    same = bool(torch.equal(torch.nan_to_num(full, nan=-7.0), torch.nan_to_num(local, nan=-7.0)))
    host = lib.sweep_host(_native.OP_COMPLETE, rows, np.array(spec.extent).reshape(2, 2), n0, n1)
    same_host = bool(np.array_equal(full.cpu().numpy(), host, equal_nan=True))
    # the summary: reduced on the device by the sweep kernels, combined by three RCCL all-reduces on device tensors
    local_summary = lib.sweep_stats(rows, spec.extent, n0, n1)
    combined = all_reduce_summary(local_summary, device="cuda:0")
    want = numpy_summary(host)
    summary_ok = all(np.array_equal(np.asarray(combined[k]), np.asarray(want[k])) for k in ("min", "max", "count"))
    result = {"backend": dist.get_backend(), "world": dist.get_world_size(), "plan": [plan.axis, plan.p_begin, plan.p_count], "gathered_equals_local": same,
              "gathered_equals_host_sweep": same_host, "summary_equals_numpy": summary_ok}  # fmt: skip
    dist.destroy_process_group()
    os.dup2(real_stdout, 1)
    print(json.dumps(result), flush=True)


def test_one_rank_nccl_all_gather_and_all_reduce():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    proc = subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-3000:]
    rec = json.loads([ln for ln in proc.stdout.splitlines() if ln.strip()][-1])
    assert rec["backend"] == "nccl" and rec["world"] == 1
    assert rec["plan"] == ["param", 0, 3]
    assert rec["gathered_equals_local"] and rec["gathered_equals_host_sweep"] and rec["summary_equals_numpy"], rec


if __name__ == "__main__" and sys.argv[1:] == ["worker"]:
    _worker()
