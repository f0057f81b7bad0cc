"""bench.py host logic that needs no GPU: `python bench.py --gpus N` without a launcher around it re-launches itself
under torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1) before anything has touched the GPU."""

import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_invocation_with_several_gpus_launches_torchrun():
    env = dict(os.environ, INFLX_BENCH_DRY_LAUNCH="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HSA_ENABLE_IPC_MODE_LEGACY"):
        env.pop(k, None)
    proc = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--steps", "7", "--warmup", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert proc.returncode == 0, proc.stderr[-2000:]
    rec = json.loads(proc.stdout.strip().splitlines()[-1])
    cmd = rec["launch"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-7:] == [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "7", "--warmup", "2"]
    assert rec["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # the parent must not have imported torch (let alone initialised HIP) to get here: nothing but the launch line is printed
    assert len(proc.stdout.strip().splitlines()) == 1


def test_under_a_launcher_the_world_size_must_match():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    proc = subprocess.run([sys.executable, "bench.py", "--gpus", "4"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert proc.returncode != 0 and "WORLD_SIZE=2" in (proc.stderr + proc.stdout)


def test_rows_per_gpu_are_chosen_from_the_free_hbm_up_front():
    """bench.choose_rows_per_gpu: BASELINE configs[4]'s 64 rows per GPU (206 GB) on a fresh MI355X, as many rows as fit beside the
    headroom on a device that has less free, never fewer than one; the figure is computed before anything is allocated."""
    sys.path.insert(0, ROOT)
    import bench

    row = 8192 * 8192 * 48
    assert bench.choose_rows_per_gpu(64, 287 * 10**9, row) == 64  # a fresh MI355X: configs[4] as written
    assert bench.choose_rows_per_gpu(64, 288 * 10**9 // 2, row) == (144 * 10**9 - bench.HBM_HEADROOM_BYTES) // row == 39  # half the HBM taken by someone else
    assert bench.choose_rows_per_gpu(64, 20 * 10**9, row) == 1 and bench.choose_rows_per_gpu(64, 0, row) == 1
    assert bench.choose_rows_per_gpu(1, 287 * 10**9, row) == 1  # N = 1: the one row of configs[1]
    assert bench.choose_rows_per_gpu(64, 64 * row + bench.HBM_HEADROOM_BYTES, row) == 64 and bench.choose_rows_per_gpu(64, 64 * row + bench.HBM_HEADROOM_BYTES - 1, row) == 63
