#!/usr/bin/env python3
"""Headline benchmark: grid-points/s of the complete_analysis sweep (BASELINE.json `metric`).

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Both forms work for N > 1: started plainly (no WORLD_SIZE in the environment) the script launches
`torch.distributed.run` itself -- before anything has touched the GPU -- and passes the one JSON line through.

A "step" is one pass of the hot path.  N = 1: ONE complete_analysis sweep of the BASELINE configs[1]
workload -- README hyperbolic model, args [m, phi0, L] = [1, 1, 1], extent (-1, 1, -1, 1),
8192 x 8192 field grid, six f64 per point (48 B) written to a device-resident (N0, N1, 6) array.
N > 1 (weak scaling): BASELINE configs[4] -- the same grid x an outer parameter axis of 512 rows,
L in linspace(0.2, 2.0, 512), sharded over the GPUs with plan_shard: 64 rows per GPU (206 GB of results
resident per GPU), so that N = 8 sweeps exactly configs[4] and a smaller N its first 64 N rows; one step =
ONE call that sweeps the rank's whole block.  Results stay on the rank's GPU (no data-path collective: the
rows are independent; the only collectives are the timing barrier, the MAX over ranks and the gather of the
per-rank kernel times for the report).  --rows-per-gpu overrides the block size (1 = one row per rank).

Timed region: inputs (parameters, code object) resident on the device, result left in HBM.
PyTorch provides the device buffer, the stream and torch.distributed only; every launch goes
through the C ABI (libinflx_hip.so).

After the timed region (N = 1 only, never part of `value`): `secondary` = BASELINE configs[2] (D5 4096^2 x 32
parameter rows in one call) and configs[3] (EGNO 4096^2) timed with HIP events, `end_to_end` = the front-end call
GeneralisedAL.complete_analysis(8192^2) -> six numpy arrays (PCIe-inclusive), `cpu_baseline` = the oracle on the
host cores.
"""

from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
FP64_LANE_RATE = 256 * 4 * 16 * 2.4e9  # FP64 VALU lane-instructions/s at full rate (78.6 TFLOP/s = 2 flop x this)
BYTES_PER_POINT = 48  # 6 x f64 written, 0 read
# parameter-row sweeps (8192^2 each, ~0.45 ms) run untimed before the warm-up steps so that the clocks have settled: 64 = 29 ms
SETTLE_ROW_SWEEPS = 64
PROFILE_ROUNDS = ("06", "05", "04", "03", "02", "01")  # profiles/rNN_traffic.json, rNN_valu.json, rNN_isa_mix.json: newest first
SIMDS = 256 * 4


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


def code_object_id(art) -> str:
    """What identifies the kernels a profile was taken of: the content tag of the artefact in the in-tree cache
    (SHA-256 over the generated model header, the kernel sources and the compiler flags -- the name the code object
    is cached under).  Deterministic, unlike the bytes of the code object, which embed the build path."""
    return os.path.splitext(os.path.basename(art.header_path))[0]


def recorded(kind: str, key: str, code_id: str):
    """The record `key` of the newest profiles/rNN_<kind>.json whose stamp (`code_objects[key]`) is the code
    object that is loaded now; None when the kernels have changed since the counters were collected (a stale
    number is worse than none) or nothing is on record."""
    for rnd in PROFILE_ROUNDS:
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", f"r{rnd}_{kind}.json")))
        except (OSError, ValueError):
            continue
        if rec.get("code_objects", {}).get(key) == code_id and key in rec:
            return rec[key], f"profiles/r{rnd}_{kind}.json"
    return None, None


def cpu_baseline(model_name: str, args, extent, budget_s: float = 12.0):
    """Oracle (C restatement of the reference's rayon sweep: five indirect calls per point into the
    -O3 model object + ops::complete_analysis) on all host cores, bounded sample.  The model object is built by clang when
    the image has one -- the reference compiles its C with `zig cc`, which is clang (compiler.py:299-310,575-584) --,
    else by gcc; the compiler is named in `sample`."""
    import numpy as np

    import oracle
    import workloads
    from workloads import example_models

    spec = example_models.get(model_name)
    src, _ = oracle.emit_c_source(workloads.model_for(model_name), **spec.compiler_kwargs)
    compilers = oracle.reference_compilers()
    cc = "clang" if "clang" in compilers else "gcc"
    om = oracle.OracleModel(oracle.compile_c_model(src, cc=cc))
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
    # BASELINE configs[0] as written: the 256 x 256 grid of the reference's own CPU-runnable case, all threads and one thread
    # (at 65 536 points the thread start-up is a visible share of a pass: best of 20)
    def at_256(threads):
        t_best = float("inf")
        for _ in range(20):
            t0 = time.perf_counter()
            om.grid_sweep(oracle.OP.COMPLETE, args, extent, 256, 256, threads=threads)
            t_best = min(t_best, time.perf_counter() - t0)
        return {"ms": t_best * 1e3, "points_per_s": 256 * 256 / t_best, "threads": threads}

    return {
        "value": n * n / best,
        "unit": "grid-points/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{model_name} {n}x{n} grid over the same extent (per-point cost does not depend on grid size), best of 3 passes, "
        f"{cores} threads ({os.cpu_count()} logical CPUs on the host), {cc} -O3 -march=native model object (reference flag list) + C restatement of the Rust sweep",
        "model_object_compiler": compilers[cc],
        "configs0_256x256": {"what": "BASELINE configs[0]: the same model on its 256x256 grid, best of 20 passes", "all_threads": at_256(cores), "one_thread": at_256(1)},
    }


HBM_HEADROOM_BYTES = 16 << 30  # left free beside the resident result block: stage / row tables, RCCL buffers, the allocator's slack


def choose_rows_per_gpu(requested: int, free_bytes: int, row_bytes: int, headroom: int = HBM_HEADROOM_BYTES) -> int:
    """Parameter rows of the resident result block of one rank, decided UP FRONT from the HBM that is free on its device
    (torch.cuda.mem_get_info) instead of asking for the full block and halving on failure: the requested number (64 = BASELINE
    configs[4] on 8 GPUs, 206 GB) if it fits beside `headroom`, else as many rows as do -- at least one.  Every rank computes its
    own figure and the ranks agree on the smallest (weak scaling: the same work per GPU) before anything is allocated."""
    fit = max(0, int(free_bytes) - int(headroom)) // max(1, int(row_bytes))
    return int(max(1, min(int(requested), fit)))


def self_launch(opt) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start torch.distributed.run as a child
    (this process has not imported torch, let alone touched a GPU) and hand its exit code on."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={opt.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]  # fmt: skip
    if os.environ.get("INFLX_BENCH_DRY_LAUNCH") == "1":  # tests: show the launch instead of performing it
        print(json.dumps({"launch": cmd, "HSA_ENABLE_IPC_MODE_LEGACY": env["HSA_ENABLE_IPC_MODE_LEGACY"]}))
        return 0
    return subprocess.run(cmd, env=env).returncode


def issue_weighted(roof, key, code_id, pps):
    """Add the issue-weighted VALU roofline of profiles/rNN_isa_mix.json (scripts/profile_isa_mix.sh) to a tile kernel's
    roofline record: every VALU instruction class of the kernel's DYNAMIC mix (SQ_INSTS_VALU_* counters) priced with its
    measured issue cost (quarter-rate v_rcp / v_rsq / v_sqrt_f64 = 3.1 fma slots, 32-bit ALU 0.7, profiles/r02_valu_rates.txt)
    gives the SIMD cycles one grid point costs; `frac_issue_weighted` = points/s x that / (1024 SIMDs x shader clock), with the
    clock the chip actually held in the profiled dispatch (GRBM_GUI_ACTIVE / 8 / duration) -- and, as
    `frac_issue_weighted_at_2.4GHz`, with the nominal clock.  Only for the code object the counters were collected on."""
    rec, src = recorded("isa_mix", key, code_id)
    if not rec or "weighted_cycles_per_point" not in rec:
        return roof
    roof = dict(roof or {"bound": "valu"})
    cyc, clk = rec["weighted_cycles_per_point"], rec.get("clock_GHz")
    roof["weighted_cycles_per_point"] = cyc  # (the dynamic mix behind it: `dynamic_mix_per_point` of the source file)
    roof["frac_issue_weighted_at_2.4GHz"] = pps * cyc / (SIMDS * 2.4e9)
    if clk:
        roof["shader_clock_GHz_under_this_kernel"] = clk
        roof["peak_points_per_s_issue_weighted"] = SIMDS * clk * 1e9 / cyc
        roof["frac_issue_weighted"] = pps * cyc / (SIMDS * clk * 1e9)
        roof["frac_issue_weighted_in_the_profiled_dispatch"] = rec.get("frac_issue_weighted")
    roof["isa_mix_source"] = src
    return roof


def secondary_workloads(_native, workloads, torch, np, device, stream, only=None, builds=("default", "tuned")):
    """BASELINE configs[2] and [3] on this GPU, kernel time by HIP events on the launch stream (no part of `value`).
    These are FP64-VALU-bound: the roofline that prices them is the VALU issue rate, from
    the SQ instruction counters on record for exactly this code object; the HBM fraction is given beside it."""
    out = []
    cases = [
        ("d5", 4096, 32, 3, "D5-brane model, 4096x4096 field grid x 32 parameter rows (a1 in linspace(2.5e-4, 1e-3, 32)) in ONE call, 25.8 GB device-resident"),
        ("egno", 4096, 1, 30, "EGNO supergravity model, 4096x4096 field grid"),
        ("doc", 4096, 1, 30, "documentation model (reference tests/test_doc.py), 4096x4096 field grid"),
    ]
    for name, n, P, repeats, text in cases:
        if only is not None and name != only:  # scripts/secondary_probe.py: one workload per profiler run
            continue
        try:
            spec, art = workloads.artifact_for(name)
            lib = _native.InflatoxDevLib(art.shared_object_path, device=device)
            rows = np.tile(np.asarray(spec.args, dtype=np.float64), (P, 1))
            if name == "d5" and P > 1:
                rows[:, 6] = np.linspace(2.5e-4, 1e-3, P)  # a1
            buf = torch.empty((P, n, n, 6), dtype=torch.float64, device=f"cuda:{device}")
            # best of three batches of back-to-back launches: a single short batch sits inside the clock governor's
            # transient and reads 10-15 % slow
            if "default" in builds:
                # one untimed batch first: after the host-side work between workloads (artefact lookup, allocation) the first ~25 launches
                # run inside the clock governor's transient (the warm-up of these side measurements, like --warmup for the steps)
                lib.sweep_device_timed(_native.OP_COMPLETE, rows, buf.data_ptr(), buf.numel() * 8, spec.extent, n, n, stream=stream, repeats=repeats)
                ms = min(lib.sweep_device_timed(_native.OP_COMPLETE, rows, buf.data_ptr(), buf.numel() * 8, spec.extent, n, n, stream=stream, repeats=repeats) for _ in range(3))
                # ... and what ONE call costs a user whose handle is idle (one call = one sweep, reference consistency_conditions.py:290-300):
                # HIP events around stage tables + tile kernel, the device drained before every call (INFLX_TIME_SINGLE_CALL)
                ms_single = min(lib.sweep_device_timed(_native.OP_COMPLETE, rows, buf.data_ptr(), buf.numel() * 8, spec.extent, n, n, stream=stream, repeats=max(3, repeats // 2), single_call=True) for _ in range(3))
            else:  # scripts/secondary_probe.py NAME:tuned -- only the profile-guided build under the profiler
                ms = ms_single = float("nan")
            pps = P * n * n / (ms * 1e-3)
            cid = code_object_id(art)
            valu, src = recorded("valu", name, cid)
            rec = {
                "workload": text,
                "kernel": "inflx_sweep_tile_complete",
                "build": "default: the reference's arithmetic (Compiler(...) as the reference's tests call it; tan_shortcut off: eta is OCML's tan of OCML's atan)",
                "ms": ms,
                "timing": f"HIP events around {repeats} back-to-back launches, best of 3 such batches",
                "single_call_ms": ms_single,
                "single_call_timing": f"one sweep from an idle handle, HIP events around stage tables + tile kernel, mean of {max(3, repeats // 2)} calls, best of 3 such batches",
                "points_per_s": pps,
                "hbm_frac": BYTES_PER_POINT * pps / 1e9 / HBM_PEAK_GBPS,
                "code_object": cid,
                "roofline": None,
            }
            if valu:
                ipp = valu["valu_insts_per_point"]
                rec["roofline"] = {
                    "bound": "valu",
                    "achieved": pps * ipp / 1e12,
                    "peak": FP64_LANE_RATE / 1e12,
                    "unit": "T lane-instr/s (FP64 VALU, full-rate issue)",
                    "frac": pps * ipp / FP64_LANE_RATE,
                    "valu_insts_per_point": ipp,
                    "source": src,
                }
            rec["roofline"] = issue_weighted(rec["roofline"], name, cid, pps)
            # the profile-guided build of the same workload (Compiler(regroup="auto", sample=(args, extent)): the model values
            # that a host measurement on the workload's own parameter values and field range clears are re-associated;
            # same parity criteria, tests/test_tuned_gpu.py) -- reported beside the default build, never instead of it
            try:
                if "tuned" not in builds:
                    raise LookupError("not requested")
                _, art_t = workloads.artifact_for(name, tuned=True)
                lib_t = _native.InflatoxDevLib(art_t.shared_object_path, device=device)
                lib_t.sweep_device_timed(_native.OP_COMPLETE, rows, buf.data_ptr(), buf.numel() * 8, spec.extent, n, n, stream=stream, repeats=repeats)  # untimed batch, as above
                ms_t = min(lib_t.sweep_device_timed(_native.OP_COMPLETE, rows, buf.data_ptr(), buf.numel() * 8, spec.extent, n, n, stream=stream, repeats=repeats) for _ in range(3))
                ms_t_single = min(lib_t.sweep_device_timed(_native.OP_COMPLETE, rows, buf.data_ptr(), buf.numel() * 8, spec.extent, n, n, stream=stream, repeats=max(3, repeats // 2), single_call=True) for _ in range(3))
                rec["profile_guided"] = {
                    "single_call_ms": ms_t_single,
                    "build": 'Compiler(regroup="auto", sample=(args, extent)): measured re-association, tan(atan t) -> t for t <= 16',
                    "regrouped_values": art_t.stage_info.get("regrouped"),
                    "ms": ms_t,
                    "points_per_s": P * n * n / (ms_t * 1e-3),
                    "hbm_frac": BYTES_PER_POINT * P * n * n / (ms_t * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    "code_object": code_object_id(art_t),
                }
                rec["profile_guided"]["roofline"] = issue_weighted(None, name + ":tuned", code_object_id(art_t), rec["profile_guided"]["points_per_s"])
                del lib_t
            except LookupError:
                pass
            except Exception as exc:  # noqa: BLE001
                rec["profile_guided"] = {"error": str(exc)[:300]}
            out.append(rec)
            del buf, lib
            torch.cuda.empty_cache()
        except Exception as exc:  # noqa: BLE001 -- an extra must never break the benchmark line
            out.append({"workload": text, "error": str(exc)[:300]})
    return out


def configs_block(secondary):
    """BASELINE configs[2] / configs[3] (and the doc model) in the compact form that rides INSIDE `roofline` -- the part of the
    line the driver's record keeps: device time, grid points per second, fraction of the HBM roofline (48 B/point), of the
    nominal FP64 VALU issue rate (2.4 GHz, one slot per instruction) and of the issue-weighted VALU roofline at the clock the
    chip holds; default build (the reference's arithmetic) and, beside it, the profile-guided build."""
    names = {"D5-brane model": "configs[2]", "EGNO supergravity model": "configs[3]", "documentation model (reference tests/test_doc.py)": "doc 4096x4096"}
    out = {}
    for rec in secondary:
        key = names.get(rec.get("workload", "").split(",")[0], rec.get("workload", "?")[:40])
        if "error" in rec:
            out[key] = {"error": rec["error"][:120]}
            continue
        roof = rec.get("roofline") or {}
        cell = {"workload": rec["workload"].split(" in ONE call")[0], "kernel": rec["kernel"], "ms": rec["ms"], "points_per_s": rec["points_per_s"], "hbm_frac": rec["hbm_frac"],
                "valu_frac_nominal": roof.get("frac"), "frac_issue_weighted": roof.get("frac_issue_weighted"), "code_object": rec["code_object"]}
        pg = rec.get("profile_guided")
        if pg and "ms" in pg:
            cell["profile_guided"] = {"ms": pg["ms"], "points_per_s": pg["points_per_s"], "hbm_frac": pg["hbm_frac"],
                                      "frac_issue_weighted": (pg.get("roofline") or {}).get("frac_issue_weighted"), "code_object": pg["code_object"]}
        out[key] = cell
    return out


def flat_configs(secondary):
    """The same figures as FLAT SCALARS for `roofline` (the driver's record keeps scalar keys of `roofline` only): prefix c2_ =
    BASELINE configs[2] (D5 4096^2 x 32 in one call), c3_ = configs[3] (EGNO 4096^2), doc_ = the doc model 4096^2.
      *_ms                     device time per sweep, sweeps enqueued back to back (throughput of a scan: tables of sweep n+1 under sweep n)
      *_single_call_ms         ONE sweep from an idle handle: HIP events around stage tables + tile kernel
      *_points_per_s, *_hbm_frac (48 B/point / 8 TB/s), *_valu_frac_nominal (VALU instr/point x points/s / 39.3e12),
      *_frac_issue_weighted    measured issue costs at the measured clock (self-defined ceiling, not the roofline of SURVEY 8d)
      *_tuned_*                the profile-guided build (Compiler(regroup="auto")) of the same workload"""
    prefix = {"D5-brane model": "c2", "EGNO supergravity model": "c3", "documentation model (reference tests/test_doc.py)": "doc"}
    out = {}
    for rec in secondary:
        pre = prefix.get(rec.get("workload", "").split(",")[0])
        if pre is None or "error" in rec:
            continue
        roof = rec.get("roofline") or {}
        out[f"{pre}_ms"] = rec["ms"]
        out[f"{pre}_single_call_ms"] = rec.get("single_call_ms")
        out[f"{pre}_points_per_s"] = rec["points_per_s"]
        out[f"{pre}_hbm_frac"] = rec["hbm_frac"]
        out[f"{pre}_valu_frac_nominal"] = roof.get("frac")
        out[f"{pre}_frac_issue_weighted"] = roof.get("frac_issue_weighted")
        pg = rec.get("profile_guided")
        if pg and "ms" in pg:
            out[f"{pre}_tuned_ms"] = pg["ms"]
            out[f"{pre}_tuned_single_call_ms"] = pg.get("single_call_ms")
            out[f"{pre}_tuned_points_per_s"] = pg["points_per_s"]
            out[f"{pre}_tuned_hbm_frac"] = pg["hbm_frac"]
    return out


def tile_path_on_configs1(_native, lib, torch, spec, n, device, stream):
    """BASELINE configs[1] evaluated PER GRID POINT: the hyperbolic 8192^2 sweep forced through inflx_sweep_tile_complete
    (INFLX_SWEEP_FORCE_TILE, an argument of inflx_sweep_device_timed_ex) -- one lane per grid point, every point evaluated, what
    the reference's loop does (src/anguelova.rs:526-539) -- instead of the row-broadcast store stream the headline `value` is
    measured on.  Same bytes written, same values (tests/test_parity_gpu.py: bit for bit at 8192^2)."""
    buf = torch.empty((n, n, 6), dtype=torch.float64, device=f"cuda:{device}")
    kw = dict(stream=stream, force_tile=True)
    lib.sweep_device_timed(_native.OP_COMPLETE, spec.args, buf.data_ptr(), buf.numel() * 8, spec.extent, n, n, repeats=20, **kw)  # untimed batch
    ms = min(lib.sweep_device_timed(_native.OP_COMPLETE, spec.args, buf.data_ptr(), buf.numel() * 8, spec.extent, n, n, repeats=20, **kw) for _ in range(3))
    single = min(lib.sweep_device_timed(_native.OP_COMPLETE, spec.args, buf.data_ptr(), buf.numel() * 8, spec.extent, n, n, repeats=10, single_call=True, **kw) for _ in range(3))
    plan = lib.sweep_plan(_native.OP_COMPLETE, 1, n, n, force_tile=True)
    del buf
    pps = n * n / (ms * 1e-3)
    return {"c1_tile_path_ms": ms, "c1_tile_path_single_call_ms": single, "c1_tile_path_points_per_s": pps, "c1_tile_path_hbm_frac": BYTES_PER_POINT * pps / 1e9 / HBM_PEAK_GBPS,
            "c1_tile_path_kernel": "inflx_sweep_tile_complete" if plan["path"] == "tile" else plan["path"]}


def next_rows_block(rows):
    """SURVEY section 8(f) rows in compact form, inside `roofline` as well: workload, path, device time, HBM fraction."""
    out = []
    for rec in rows:
        if "error" in rec:
            out.append({"row": rec.get("row"), "error": rec["error"][:120]})
        else:
            out.append({"row": rec["row"], "workload": rec["workload"].split(" (")[0][:70], "path": rec.get("path"), "ms": rec["ms"], "points_per_s": rec["points_per_s"], "hbm_frac": rec.get("hbm_frac")})
    return out


def next_rows(_native, workloads, torch, np, device, stream):
    """SURVEY.md section 8(f), the rows either side of the headline path, each timed on this GPU (never part of `value`):
    f1 the single-quantity sweeps (8 B/point: row-broadcast planes stream for the hyperbolic model, tile kernels for the doc
    model), f2 an on-trajectory call (host points in, host results out: latency, not bandwidth), f3 the raw-values sweep
    behind calc_V_array / calc_H_array (5 SoA planes).  Kernel-side figures by HIP events on the launch stream
    (sweep_device_timed, best of 3 batches of 20), host-side calls by wall clock (best of 5)."""
    import time

    out = []

    def device_sweep(row, model, op, n, layout, what, width):
        try:
            spec, art = workloads.artifact_for(model)
            lib = _native.InflatoxDevLib(art.shared_object_path, device=device)
            buf = torch.empty((n * n * width,), dtype=torch.float64, device=f"cuda:{device}")
            ms = min(lib.sweep_device_timed(op, spec.args, buf.data_ptr(), buf.numel() * 8, spec.extent, n, n, layout=layout, stream=stream, repeats=20) for _ in range(3))
            pps = n * n / (ms * 1e-3)
            plan = lib.sweep_plan(op, 1, n, n, layout=layout)
            out.append({"row": row, "workload": f"{model} {n}x{n}, {what}", "path": plan.get("path"), "ms": ms, "points_per_s": pps,
                        "bytes_per_point": 8 * width, "GBps": 8 * width * pps / 1e9, "hbm_frac": 8 * width * pps / 1e9 / HBM_PEAK_GBPS,
                        "timing": "HIP events around 20 back-to-back sweeps, best of 3 batches", "code_object": code_object_id(art)})
            del buf
        except Exception as exc:  # noqa: BLE001 -- a side figure must never cost the benchmark line
            out.append({"row": row, "workload": f"{model} {n}x{n}, {what}", "error": str(exc)[:300]})

    for op, name in ((_native.OP_CONSISTENCY, "consistency_only"), (_native.OP_EPSILON_V, "epsilon_v_only"), (_native.OP_RAPIDTURN, "consistency_rapidturn_only")):
        device_sweep("f1", "hyperbolic", op, 8192, _native.LAYOUT_AOS, f"{name} (src/anguelova.rs:138-163), one f64 per point", 1)
    device_sweep("f1", "doc", _native.OP_CONSISTENCY, 4096, _native.LAYOUT_AOS, "consistency_only, one f64 per point", 1)
    device_sweep("f1", "doc", _native.OP_EPSILON_V, 4096, _native.LAYOUT_AOS, "epsilon_v_only, one f64 per point", 1)
    device_sweep("f3", "doc", _native.OP_RAW, 4096, _native.LAYOUT_SOA, "V, v00, v10, v11, |dV|^2 as five planes (what calc_V_array reads)", 5)
    device_sweep("f3", "doc", _native.OP_HESSE, 4096, _native.LAYOUT_SOA, "v00, v01, v10, v11 as four planes (calc_H_array: the reference's own v01)", 4)
    torch.cuda.empty_cache()
    try:  # f2: the on-trajectory call of the reference's trajectory fixtures' size, and a long one
        spec, art = workloads.artifact_for("doc")
        lib = _native.InflatoxDevLib(art.shared_object_path, device=device)
        rng = np.random.default_rng(7)
        x0a, x0b, x1a, x1b = spec.extent
        for npts in (500, 1_000_000):
            pts = np.column_stack([rng.uniform(x0a, x0b, npts), rng.uniform(x1a, x1b, npts)])
            lib.sweep_on_trajectory(_native.OP_COMPLETE, spec.args, pts)
            best = float("inf")
            for _ in range(5):
                t0 = time.perf_counter()
                lib.sweep_on_trajectory(_native.OP_COMPLETE, spec.args, pts)
                best = min(best, time.perf_counter() - t0)
            out.append({"row": "f2", "workload": f"doc, complete_analysis_ot on {npts} points (host points in, host results out)", "ms": best * 1e3, "points_per_s": npts / best,
                        "timing": "wall clock of the call, best of 5"})
    except Exception as exc:  # noqa: BLE001
        out.append({"row": "f2", "error": str(exc)[:300]})
    return out


def end_to_end(workloads, np, model: str, n: int, device: int):
    """The front-end call a user of the reference makes, GeneralisedAL.complete_analysis -> six numpy arrays on the host
    (device sweep + PCIe copy), and the two opt-in forms that avoid the copy.  `cold` = the first call of the process:
    the result array is a fresh mapping whose pages do not exist yet, as with the reference's per-call np.zeros; `warm` =
    best of 3 further calls, whose result memory is recycled page-resident memory from the result pool
    (inflatox_amd/_result_pool.py) because the previous result was dropped."""
    import torch

    from inflatox_amd.consistency_conditions import GeneralisedAL

    spec, art = workloads.artifact_for(model)
    al = GeneralisedAL(art, device=device)

    def timed(fn, repeats):
        best = float("inf")
        for _ in range(repeats):
            t0 = time.perf_counter()
            res = fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
            del res
        return best

    def rec(seconds, what):
        return {"what": what, "ms": seconds * 1e3, "points_per_s": n * n / seconds, "GBps": BYTES_PER_POINT * n * n / seconds / 1e9}

    cold = timed(lambda: al.complete_analysis(spec.args, *spec.extent, n, n, progress=False), 1)
    warm = timed(lambda: al.complete_analysis(spec.args, *spec.extent, n, n, progress=False), 3)
    lean = timed(lambda: al.complete_analysis(spec.args, *spec.extent, n, n, progress=False, broadcast_views=True), 5)
    dev = timed(lambda: al.complete_analysis_device(spec.args, *spec.extent, n, n), 5)
    small = timed(lambda: al.complete_analysis(spec.args, *spec.extent, 256, 256, progress=False), 20)
    return {
        "workload": f"GeneralisedAL.complete_analysis, {model} {n}x{n} -> six arrays of the caller (never part of `value`)",
        # the reference-compatible default, first call / repeated calls
        "ms": warm * 1e3,
        "points_per_s": n * n / warm,
        "GBps": BYTES_PER_POINT * n * n / warm / 1e9,
        "default_cold": rec(cold, "first call: fresh host pages; the model ignores x1, so one evaluated column crosses PCIe and host threads write the caller's writable (N0, N1, 6) array from it (csrc/inflx_hip.cpp sweep_host_broadcast)"),
        "default_warm": rec(warm, "best of 3 repeated calls: result memory recycled by the result pool, same host-side broadcast fill"),
        "broadcast_views": rec(lean, "opt-in broadcast_views=True, best of 5: one evaluated line copied, six read-only stride-0 views (only where the model ignores one field)"),
        "device_resident": rec(dev, "complete_analysis_device, best of 5 incl. synchronisation: six torch views of a device tensor (DLPack / __cuda_array_interface__), nothing crosses PCIe"),
        # BASELINE configs[0] as written: the 256 x 256 grid through the same front-end call (3.1 MB: below the host-fill threshold,
        # the whole result crosses PCIe); cpu_baseline.configs0_256x256 is the CPU port on the same grid
        "configs0_256x256": {"what": "the default call on BASELINE configs[0]'s 256x256 grid, best of 20 (launch + copy latency, not throughput)",
                             "ms": small * 1e3, "points_per_s": 256 * 256 / small},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="hyperbolic")
    ap.add_argument("--grid", dest="n", type=int, default=8192, help="grid points per axis")
    ap.add_argument("--rows-per-gpu", type=int, default=None, help="parameter rows per GPU and step (default: 1 at N = 1, 64 at N > 1 = BASELINE configs[4] on 8 GPUs)")
    ap.add_argument("--no-settle", dest="settle", action="store_false", help="skip the untimed clock-settling sweeps before the warm-up steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary workloads and the end-to-end call")
    opt = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and opt.gpus > 1:
        raise SystemExit(self_launch(opt))
    # must be in place before the first HIP call of this process (RCCL's IPC handles, see the Environment notes)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

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

        if world == 1 and "RANK" not in os.environ:  # INFLX_BENCH_FORCE_DIST=1 without a launcher: a one-rank rendezvous of our own
            import socket

            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                port = sock.getsockname()[1]
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
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

    from inflatox_amd import _native

    import workloads
    from inflatox_amd.distributed import plan_shard

    spec, art = workloads.artifact_for(opt.model)
    lib = _native.InflatoxDevLib(art.shared_object_path, device=local_rank)

    N0 = N1 = opt.n
    # The outer parameter axis.  N = 1: the one row of BASELINE configs[1].  N > 1: BASELINE configs[4], 512 rows with
    # L (the model's last parameter) in linspace(0.2, 2.0, 512), 64 rows per GPU: 8 GPUs sweep exactly that axis, fewer
    # GPUs its first 64 N rows (weak scaling: the work per GPU is fixed).  plan_shard hands every rank its block.
    rows_per_gpu = opt.rows_per_gpu if opt.rows_per_gpu else (1 if world == 1 else 64)
    rows_requested = rows_per_gpu
    if distributed:
        assert dist.get_world_size() == opt.gpus == world, (dist.get_world_size(), opt.gpus, world)
    # The result block stays resident: rows_per_gpu x 3.2 GB (206 GB at the default 64 = 72 % of a fresh MI355X's HBM).  How many rows
    # fit is decided up front from the HBM that is free on the rank's device (choose_rows_per_gpu) -- another tenant on the GPU, a
    # smaller part -- and agreed on by all ranks (the smallest figure: weak scaling, the same work per GPU) BEFORE anything is
    # allocated; the line says what ran (`config.parameter_rows_per_gpu`, `config.rows_per_gpu_requested`, `config.hbm_free_gb_at_start`).
    # INFLX_BENCH_MAX_BLOCK_GB caps the free figure artificially (the rehearsal tests shrink the block with it).  Should the
    # allocation fail all the same (fragmentation), the block is halved -- on every rank -- until it fits.
    cap_gb = float(os.environ.get("INFLX_BENCH_MAX_BLOCK_GB", "0") or 0)
    row_bytes = N0 * N1 * 6 * 8
    free_bytes = int(torch.cuda.mem_get_info(local_rank)[0])
    if cap_gb:
        free_bytes = min(free_bytes, int(cap_gb * 1e9) + HBM_HEADROOM_BYTES)
    rows_per_gpu = choose_rows_per_gpu(rows_requested, free_bytes, row_bytes)
    if distributed:
        agreed = torch.tensor([rows_per_gpu], dtype=torch.int64, device=comm_device)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        rows_per_gpu = int(agreed.item())
    if rows_per_gpu != rows_requested:
        print(f"[bench rank {rank}] {rows_requested} rows per GPU requested, {free_bytes / 1e9:.0f} GB free on device {local_rank}: sweeping {rows_per_gpu} rows per GPU", file=sys.stderr, flush=True)
    out = None
    while True:
        try:
            out = torch.empty((rows_per_gpu, N0, N1, 6), dtype=torch.float64, device=f"cuda:{local_rank}")
            ok = 1
        except (torch.OutOfMemoryError, RuntimeError) as exc:
            if rows_per_gpu == 1 and not distributed:
                raise
            print(f"[bench rank {rank}] {rows_per_gpu} rows per GPU could not be allocated ({str(exc).splitlines()[0][:120]})", file=sys.stderr, flush=True)
            ok = 0
            torch.cuda.empty_cache()
        if distributed:  # every rank must hold the same block: halve everywhere if any rank failed
            flag = torch.tensor([ok], dtype=torch.int64, device=comm_device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        if ok:
            break
        if rows_per_gpu == 1:
            raise SystemExit("not even one parameter row fits on every rank's device")
        out = None
        torch.cuda.empty_cache()
        rows_per_gpu //= 2
    total_rows = rows_per_gpu * world
    all_rows = np.tile(np.array(spec.args, dtype=np.float64), (total_rows, 1))
    if total_rows > 1:
        all_rows[:, -1] = 0.2 + (2.0 - 0.2) * np.arange(total_rows) / 511.0  # == np.linspace(0.2, 2.0, 512)[:total_rows]
    plan = plan_shard(total_rows, N0, world, rank)
    assert plan.axis == "param" and plan.p_count == rows_per_gpu and plan.row_count == N0
    args = all_rows[plan.p_begin : plan.p_begin + plan.p_count]
    device_name = torch.cuda.get_device_name(local_rank)
    print(f"[bench rank {rank}/{world}] device {local_rank}: {device_name}, {torch.cuda.mem_get_info(local_rank)[0] / 1e9:.0f} GB free after allocating {out.numel() * 8 / 1e9:.1f} GB", file=sys.stderr, flush=True)
    # a stream of our own: torch's default stream has the NULL handle, which the C ABI reads as "the model's
    # own stream"; with an explicit one the HIP events below are recorded on the stream the kernels run on
    launch_stream = torch.cuda.Stream(device=f"cuda:{local_rank}")
    stream = launch_stream.cuda_stream
    nbytes = out.numel() * 8

    def step():
        lib.sweep_device(_native.OP_COMPLETE, args, out.data_ptr(), nbytes, spec.extent, N0, N1, stream=stream)

    # Before the W warm-up steps: untimed sweeps of the same workload until the shader / memory clocks have settled.  The
    # first ~25 launches (~11 ms) of a store stream after idle run up to 8 % slow while the clock governor finds its level
    # (kernel trace in profiles/r03_experiments.txt section 15: 437 -> 471 -> 433 us); with W = 3 the timed region would
    # sit entirely inside that transient (0.478 ms/step instead of 0.447).  Same call, same arguments, results discarded
    # like any warm-up step; the timed region below is still exactly K steps of the full workload.
    settle_steps = max(1, -(-SETTLE_ROW_SWEEPS // rows_per_gpu)) if opt.settle else 0
    for _ in range(settle_steps):
        step()
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
    # the dominant kernel on its own: the store stream of ONE parameter row of this rank's block (its launch is the
    # same for every row; the timing-only mode needs the rows to fit one table batch)
    one_row = args[:1]
    ms_kernel_isolated = lib.sweep_device_timed(_native.OP_COMPLETE, one_row, out.data_ptr(), nbytes, spec.extent, N0, N1, stream=stream, repeats=max(5, opt.steps), dominant_only=row_path)
    # ... and where the roofline prices it: INSIDE the step, exactly as the timed region enqueues it (all rows of the block, the
    # per-row evaluation overlapping on the side stream, the cross-stream waits in place), from HIP event pairs recorded on the
    # launch stream around every launch of the dominant kernel; per parameter row
    ms_kernel = lib.sweep_device_timed(_native.OP_COMPLETE, args, out.data_ptr(), nbytes, spec.extent, N0, N1, stream=stream, repeats=max(2, min(opt.steps, 40 // rows_per_gpu)), in_pipeline=True) / rows_per_gpu
    # ... and the whole step (all rows of the block, every launch of the call), per parameter row
    ms_sweep = lib.sweep_device_timed(_native.OP_COMPLETE, args, out.data_ptr(), nbytes, spec.extent, N0, N1, stream=stream, repeats=max(2, min(opt.steps, 40 // rows_per_gpu))) / rows_per_gpu
    # ... and ONE call from an idle handle (per-row evaluation + store stream between two events, the device drained before each call)
    ms_single = lib.sweep_device_timed(_native.OP_COMPLETE, args, out.data_ptr(), nbytes, spec.extent, N0, N1, stream=stream, repeats=max(2, min(opt.steps, 40 // rows_per_gpu)), single_call=True) / rows_per_gpu
    points = N0 * N1
    achieved = BYTES_PER_POINT * points / (ms_kernel * 1e-3) / 1e9

    # every rank's own dominant-kernel time and roofline fraction, for the report of an N > 1 run
    mine = torch.tensor([ms_kernel, ms_sweep, step_ms_events], dtype=torch.float64, device=comm_device)
    if distributed:
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
    else:
        gathered = [mine]
    per_rank = []
    for r, g in enumerate(gathered):
        k_ms, s_ms, e_ms = (float(v) for v in g.cpu())
        gbps = BYTES_PER_POINT * points / (k_ms * 1e-3) / 1e9
        per_rank.append({"rank": r, "kernel_ms": k_ms, "sweep_ms": s_ms, "step_ms_hip_events": e_ms, "achieved": gbps, "frac": gbps / HBM_PEAK_GBPS})

    if distributed:
        device_names = [None] * world
        dist.all_gather_object(device_names, f"cuda:{local_rank} {device_name}")
    else:
        device_names = [f"cuda:{local_rank} {device_name}"]

    # outside the timed region: the statistics path of a sharded sweep -- every rank reduces its own block
    # on the device (summary-only sweep), three six-element all-reduces combine the ranks (RCCL when N > 1)
    stats_info = None
    try:
        t0 = time.perf_counter()
        local = lib.sweep_stats(args, spec.extent, N0, N1)  # all rows of this rank's block
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
        cid = code_object_id(art)
        traffic_rec, traffic_src = recorded("traffic", kernel, cid) if (opt.model, opt.n) == ("hyperbolic", 8192) else (None, None)
        line = {
            "metric": "grid-points/sec on complete_analysis sweep; achieved HBM GB/s vs peak",
            "value": world * rows_per_gpu * points * opt.steps / elapsed,
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
                "workload": (
                    f"{opt.model} model, {N0}x{N1} field grid, args {spec.args.tolist()}, extent {list(spec.extent)}, complete_analysis (6 f64/point, AoS), device-resident result"
                    if total_rows == 1
                    else f"{opt.model} model, {N0}x{N1} field grid x {total_rows} parameter rows (rows 0..{total_rows - 1} of BASELINE configs[4]'s axis L = linspace(0.2, 2.0, 512)), "
                    f"{rows_per_gpu} rows per GPU swept by ONE call per step, extent {list(spec.extent)}, complete_analysis (6 f64/point, AoS), "
                    f"{rows_per_gpu * N0 * N1 * BYTES_PER_POINT / 1e9:.1f} GB of results resident per GPU"
                ),
                "untimed_sweeps_before_warmup": settle_steps * rows_per_gpu,  # clock settling, see the comment at the warm-up loop
                "parameter_rows_per_gpu": rows_per_gpu,
                "rows_per_gpu_requested": rows_requested,  # larger than parameter_rows_per_gpu: the block was sized to the HBM free on the ranks' devices
                "hbm_free_gb_at_start": round(free_bytes / 1e9, 1),  # rank 0's device, before the result block was allocated
                "parameter_rows_total": total_rows,
                "baseline_config": "configs[1]" if total_rows == 1 else ("configs[4]" if (total_rows, N0) == (512, 8192) else f"configs[4] axis, first {total_rows} of 512 rows"),
                "parallelism": (f"parameter-axis x{world} (plan_shard, no data-path collective)" if world > 1 else "single GPU") + (" [REHEARSAL: ranks share GPUs, gloo]" if rehearse else ""),
                # what "parity" means for the numbers on this line (tests/tolerance.py, profiles/r05_parity_stats.json)
                "parity": "this workload's model (README hyperbolic): six arrays within the literal 1e-10 of the reference's C as gcc AND as clang build it, NaN/Inf exact (measured 4e-16); "
                "roofline.configs models (D5, EGNO, doc): 1e-10 + a multiple of the reference's own measured rounding error -- its gcc and clang builds differ from each other by up to 4e-5 there",
            },
            "ranks": world,
            "comm_backend": (dist.get_backend() if distributed else None),
            "rccl_ranks": (dist.get_world_size() if distributed and dist.get_backend() == "nccl" else 0),
            "devices": device_names,
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                # HBM bytes per launch from the PMC passes on record for exactly this code object, else null
                "traffic": traffic_rec["traffic_bytes"] if traffic_rec else None,
                "traffic_source": traffic_src,
                "code_object": cid,
                "kernel": kernel,
                "kernel_ms": ms_kernel,
                "kernel_timing": "HIP event pairs on the launch stream around every launch of the dominant kernel inside full steps (per parameter row); kernel_ms_isolated: the same kernel relaunched back to back on its own",
                "kernel_ms_isolated": ms_kernel_isolated,
                "sweep_ms": ms_sweep,  # whole step / parameter rows per step
                "single_call_ms": ms_single,  # one call from an idle handle / parameter rows per call
                "call_GBps": BYTES_PER_POINT * points / (ms_sweep * 1e-3) / 1e9,
                "timed_region_ms_per_step_hip_events": step_ms_events,
                "kernels_per_step": ["inflx_sweep_rowvals_complete", "inflx_sweep_rowstream6"] if row_path else ["inflx_sweep_tile_complete"],
                "algorithmic_bytes_per_launch": BYTES_PER_POINT * points,
                "per_rank": per_rank,
            },
        }
        line["summary_sweep"] = stats_info
    if distributed:
        dist.barrier()
    if rank == 0:
        if world == 1 and not opt.no_extras:
            del out
            torch.cuda.empty_cache()
            line["secondary"] = secondary_workloads(_native, workloads, torch, np, local_rank, stream)
            line["next_rows"] = next_rows(_native, workloads, torch, np, local_rank, stream)
            # the same figures in compact form inside `roofline`, the object the driver's record keeps whole
            line["roofline"]["configs"] = configs_block(line["secondary"])
            line["roofline"]["next_rows"] = next_rows_block(line["next_rows"])
            # ... and as flat scalars, which is what the driver's parser keeps of `roofline` (round 5: the nested blocks were dropped)
            line["roofline"].update(flat_configs(line["secondary"]))
            if (opt.model, opt.n) == ("hyperbolic", 8192):
                try:
                    line["roofline"].update(tile_path_on_configs1(_native, lib, torch, spec, opt.n, local_rank, stream))
                except Exception as exc:  # noqa: BLE001 -- a side figure must never cost the benchmark line
                    line["roofline"]["c1_tile_path_error"] = str(exc)[:200]
            line["roofline"]["host_threads_budget"] = _native.host_threads()["budget"]
            try:
                line["end_to_end"] = end_to_end(workloads, np, opt.model, opt.n, local_rank)
            except Exception as exc:  # noqa: BLE001
                line["end_to_end"] = {"error": str(exc)[:300]}
        if world == 1 and not opt.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(opt.model, spec.args, spec.extent)
            except Exception as exc:  # noqa: BLE001 -- the reported baseline must never cost the benchmark line
                line["cpu_baseline"] = {"error": str(exc)[:300]}
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
