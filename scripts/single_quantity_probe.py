#!/usr/bin/env python3
"""SURVEY section 8(f) kernels on their own: device time of the single-quantity sweeps, the raw-values planes, the Hesse planes and
an on-trajectory call, for given models -- the workload behind profiles/rNN_next_rows_* (run under `rocprofv3 --kernel-trace --stats`
by scripts/profile_next_rows.sh) and the A/B of the quick-division spelling of consistency_only.

usage: single_quantity_probe.py [MODEL[:ieee] ...]     default: doc doc:ieee egno egno:ieee hyperbolic
  :ieee = the kernels built with -DINFLX_EXPERIMENT_IEEE_EPILOGUE=1 (the compiler's divisions in the hot loop as well)
INFLX_EXPERIMENT_COMPILE_ONLY=1 (CPU container): build every variant into the in-tree cache, which travels to the GPU box."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from inflatox_amd import _native  # noqa: E402
import workloads  # noqa: E402
from workloads import example_models  # noqa: E402
from inflatox_amd.compiler import Compiler  # noqa: E402

compile_only = os.environ.get("INFLX_EXPERIMENT_COMPILE_ONLY") == "1"
cases = sys.argv[1:] or ["doc", "doc:ieee", "egno", "egno:ieee", "hyperbolic"]
rounds = int(os.environ.get("INFLX_EXPERIMENT_ROUNDS", "3"))
OPS = [("consistency_only", _native.OP_CONSISTENCY, 1, _native.LAYOUT_AOS), ("consistency_rapidturn_only", _native.OP_RAPIDTURN, 1, _native.LAYOUT_AOS),
       ("epsilon_v_only", _native.OP_EPSILON_V, 1, _native.LAYOUT_AOS), ("raw planes", _native.OP_RAW, 5, _native.LAYOUT_SOA), ("hesse planes", _native.OP_HESSE, 4, _native.LAYOUT_SOA)]
if not compile_only:
    import torch

    stream = torch.cuda.current_stream().cuda_stream
best = {}
ids = {}
for case in cases * (1 if compile_only else rounds):
    name, _, fl = case.partition(":")
    spec = example_models.get(name)
    flags = list(Compiler.default_hipcc_flags) + (["-DINFLX_EXPERIMENT_IEEE_EPILOGUE=1"] if fl == "ieee" else [])
    art = Compiler(workloads.model_for(name), silent=True, compiler_flags=flags, **spec.compiler_kwargs).compile()
    ids[case] = os.path.splitext(os.path.basename(art.header_path))[0]
    if compile_only:
        print("compiled", case, flush=True)
        continue
    n = 8192 if name == "hyperbolic" else 4096
    lib = _native.InflatoxDevLib(art.shared_object_path)
    for label, op, width, layout in OPS:
        buf = torch.empty((n * n * width,), dtype=torch.float64, device="cuda:0")
        ms = min(lib.sweep_device_timed(op, spec.args, buf.data_ptr(), buf.numel() * 8, spec.extent, n, n, layout=layout, stream=stream, repeats=20) for _ in range(3))
        key = (case, label)
        best[key] = min(best.get(key, float("inf")), ms)
        del buf
    if name == "doc" and fl == "":
        rng = np.random.default_rng(7)
        x0a, x0b, x1a, x1b = spec.extent
        pts = np.column_stack([rng.uniform(x0a, x0b, 1_000_000), rng.uniform(x1a, x1b, 1_000_000)])
        lib.sweep_on_trajectory(_native.OP_COMPLETE, spec.args, pts)
    del lib
    torch.cuda.empty_cache()
if not compile_only:
    out = {}
    for (case, label), ms in best.items():
        n = 8192 if case.startswith("hyperbolic") else 4096
        width = {lbl: w for lbl, _, w, _ in OPS}[label]
        out.setdefault(case, {"code_object": ids[case], "grid": f"{n}x{n}"})[label] = {"ms": round(ms, 5), "points_per_s": n * n / (ms * 1e-3), "hbm_frac": 8 * width * n * n / (ms * 1e-3) / 8e12}
    print(json.dumps(out, indent=1))
