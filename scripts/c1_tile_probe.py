#!/usr/bin/env python3
"""(GPU box, for `rocprofv3 --kernel-trace --stats`) BASELINE configs[1] -- hyperbolic 8192^2 -- evaluated PER GRID POINT: the sweep forced
through inflx_sweep_tile_complete (INFLX_SWEEP_FORCE_TILE), 20 sweeps back to back x 3 and 10 single calls x 3, HIP-event times as one
JSON line; the kernel trace of the run is profiles/rNN_c1_tile_path_kernel_stats.csv."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import workloads
from inflatox_amd import _native

n = 8192
spec, art = workloads.artifact_for("hyperbolic")
lib = _native.InflatoxDevLib(art.shared_object_path)
buf = torch.empty((n, n, 6), dtype=torch.float64, device="cuda")
stream = torch.cuda.Stream()
kw = dict(stream=stream.cuda_stream, force_tile=True)
t = lambda **k: lib.sweep_device_timed(_native.OP_COMPLETE, spec.args, buf.data_ptr(), buf.numel() * 8, spec.extent, n, n, **kw, **k)  # noqa: E731
t(repeats=20)
back = min(t(repeats=20) for _ in range(3))
single = min(t(repeats=10, single_call=True) for _ in range(3))
print(json.dumps({"workload": "hyperbolic 8192x8192 complete_analysis, every grid point through inflx_sweep_tile_complete (INFLX_SWEEP_FORCE_TILE)",
                  "back_to_back_ms": back, "single_call_ms": single, "points_per_s": n * n / (back * 1e-3), "hbm_frac": 48 * n * n / (back * 1e-3) / 8e12,
                  "plan": lib.sweep_plan(_native.OP_COMPLETE, 1, n, n, force_tile=True)}))
