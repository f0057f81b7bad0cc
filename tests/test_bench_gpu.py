"""bench.py contract on a GPU box: the one-line JSON of the N=1 run and a two-rank rehearsal of the
N>1 path (ranks share the GPU and synchronise over gloo -- RCCL refuses two ranks per device)."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline")


def _line(proc):
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must carry exactly one line, got {len(lines)}"
    return json.loads(lines[0])


def test_single_gpu_line():
    proc = subprocess.run(
        [sys.executable, "bench.py", "--grid", "2048", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=900
    )
    line = _line(proc)
    for key in REQUIRED:
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 5 and line["warmup"] == 2
    assert line["unit"] == "grid-points/s" and line["dtype"] == "f64" and line["scaling"] == "weak"
    assert line["value"] > 1e9  # north_star's floor, on a grid 16x smaller than the headline one
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    assert "workload" in line["config"] and "model" not in line["config"]
    assert roof["per_rank"][0]["rank"] == 0 and abs(roof["per_rank"][0]["kernel_ms"] - roof["kernel_ms"]) < 1e-9
    assert line["rccl_ranks"] == 0 and line["ranks"] == 1
    # the roofline prices the dominant kernel as it runs inside the steps; the isolated relaunch is reported beside it
    assert 0.5 * roof["kernel_ms_isolated"] < roof["kernel_ms"] < 2.0 * roof["kernel_ms_isolated"] and roof["kernel_ms"] <= 1.05 * roof["timed_region_ms_per_step_hip_events"]
    assert line["config"]["parameter_rows_per_gpu"] == 1 and line["config"]["baseline_config"] == "configs[1]"
    assert line["config"]["untimed_sweeps_before_warmup"] == 64  # clock settling, disclosed on the line
    # BASELINE configs[2] and [3] and the PCIe-inclusive front-end call ride on the same line (never part of `value`)
    sec = {rec["workload"].split(",")[0]: rec for rec in line["secondary"]}
    assert all("error" not in rec for rec in line["secondary"]), line["secondary"]
    assert sec["D5-brane model"]["points_per_s"] > 1e9 and sec["EGNO supergravity model"]["points_per_s"] > 1e9
    assert "x 32 parameter rows" in sec["D5-brane model"]["workload"]
    # ... and, compact, inside `roofline` -- the object the driver's record keeps whole
    cfg = roof["configs"]
    assert set(cfg) == {"configs[2]", "configs[3]", "doc 4096x4096"}
    assert abs(cfg["configs[2]"]["ms"] - sec["D5-brane model"]["ms"]) < 1e-12 and abs(cfg["configs[3]"]["points_per_s"] - sec["EGNO supergravity model"]["points_per_s"]) < 1e-3
    assert all(0 < c["hbm_frac"] < 1 and len(c["code_object"]) == 20 and c["profile_guided"]["ms"] > 0 for c in cfg.values())
    assert len(json.dumps(roof)) < 7000  # stays inside the driver's record
    assert len(roof["next_rows"]) == len(line["next_rows"])
    assert line["end_to_end"]["points_per_s"] > 1e8
    # the single-quantity sweeps, an on-trajectory call and the raw-values planes are measured on the same line
    rows = line["next_rows"]
    assert all("error" not in rec for rec in rows), rows
    assert {rec["row"] for rec in rows} == {"f1", "f2", "f3"} and all(rec["points_per_s"] > 0 for rec in rows)
    assert {rec["path"] for rec in rows if rec["row"] == "f1"} == {"row_stream", "tile"}


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("form", ["plain", "torchrun"])
def test_two_rank_rehearsal(form):
    """N > 1 both ways the driver may start it: `python bench.py --gpus 2` (the script launches its ranks itself,
    before anything has touched the GPU) and under torch.distributed.run."""
    env = dict(os.environ, INFLX_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    rows = 3 if form == "plain" else None  # default at N > 1: 64 rows per GPU (BASELINE configs[4]); 2048^2 x 64 = 12.9 GB per rank
    tail = ["bench.py", "--gpus", "2", "--grid", "2048", "--steps", "5", "--warmup", "2"] + (["--rows-per-gpu", str(rows)] if rows else [])
    rows = rows or 64
    if form == "plain":
        cmd = [sys.executable, *tail]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), *tail]  # fmt: skip
    proc = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    line = _line(proc)
    assert line["ranks"] == 2 and line["comm_backend"] == "gloo" and line["rccl_ranks"] == 0
    assert [r["rank"] for r in line["roofline"]["per_rank"]] == [0, 1] and all(r["kernel_ms"] > 0 for r in line["roofline"]["per_rank"])
    assert "secondary" not in line
    assert line["n_gpus"] == 2 and "cpu_baseline" not in line
    assert "REHEARSAL" in line["config"]["parallelism"]
    # every rank's rows are counted: value = ranks * rows per rank * points * steps / max-over-ranks time
    assert abs(line["value"] - 2 * rows * 2048 * 2048 * 5 / (line["ms_per_step"] * 5e-3)) / line["value"] < 1e-9
    assert line["config"]["parameter_rows_per_gpu"] == rows and line["config"]["parameter_rows_total"] == 2 * rows
    assert line["config"]["untimed_sweeps_before_warmup"] == (66 if rows == 3 else 64)  # whole calls: ceil(64 / rows) * rows
    assert "configs[4]" in line["config"]["baseline_config"] and "linspace(0.2, 2.0, 512)" in line["config"]["workload"]
    # the summary of all parameter rows was combined across ranks (epsilon_V is finite everywhere)
    assert line["summary_sweep"]["non_nan"][1] == 2 * rows * 2048 * 2048


def test_two_rank_rehearsal_sizes_the_block_to_the_free_hbm():
    """The first 8-GPU run must succeed unattended: the resident block is sized UP FRONT from the HBM that is free on the ranks'
    devices (bench.choose_rows_per_gpu), on every rank alike, and the line says which size ran.  Forced here with a 5 GB cap on the
    free figure (2048^2 x 6 x 8 B = 0.2 GB per row: 64 requested, 24 fit)."""
    env = dict(os.environ, INFLX_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", INFLX_BENCH_MAX_BLOCK_GB="5")
    env.pop("WORLD_SIZE", None)
    proc = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--grid", "2048", "--steps", "3", "--warmup", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    line = _line(proc)
    cfg = line["config"]
    assert cfg["rows_per_gpu_requested"] == 64 and cfg["parameter_rows_per_gpu"] == 24 and cfg["parameter_rows_total"] == 48
    assert cfg["hbm_free_gb_at_start"] > 0
    assert "64 rows per GPU requested" in proc.stderr and "sweeping 24 rows per GPU" in proc.stderr
    assert abs(line["value"] - 2 * 24 * 2048 * 2048 * 3 / (line["ms_per_step"] * 3e-3)) / line["value"] < 1e-9
    assert len(line["devices"]) == 2 and all("cuda:0" in d for d in line["devices"])
    assert line["summary_sweep"]["non_nan"][1] == 2 * 24 * 2048 * 2048


def test_one_rank_through_rccl():
    """The process-group path of bench.py on the real backend: INFLX_BENCH_FORCE_DIST=1 brings up a one-rank `nccl` group (RCCL), so
    that the barrier, the MIN all-reduce that agrees on the block size, the MAX over ranks, the gather of the per-rank kernel times
    and of the device names, and the summary all-reduces have all run on device tensors through RCCL before any multi-GPU job."""
    env = dict(os.environ, INFLX_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    proc = subprocess.run([sys.executable, "bench.py", "--grid", "2048", "--steps", "3", "--warmup", "1", "--rows-per-gpu", "2", "--no-extras", "--no-cpu-baseline"],
                          cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    line = _line(proc)
    assert line["comm_backend"] == "nccl" and line["rccl_ranks"] == 1 and line["ranks"] == 1 and line["n_gpus"] == 1
    assert line["config"]["parameter_rows_per_gpu"] == 2 and line["config"]["rows_per_gpu_requested"] == 2
    assert len(line["devices"]) == 1 and line["devices"][0].startswith("cuda:0 ")
    assert line["summary_sweep"]["non_nan"][1] == 2 * 2048 * 2048
    assert abs(line["value"] - 2 * 2048 * 2048 * 3 / (line["ms_per_step"] * 3e-3)) / line["value"] < 1e-9
