#!/usr/bin/env python3
"""Where should a tile sweep's stage tables be evaluated?  Back-to-back and single-call device time per sweep for the
tile workloads, under the policy the environment selects (INFLX_EXPERIMENT_TABLES = same | side | unset = adaptive,
INFLX_EXPERIMENT_SIDE_PRIORITY = 1: side stream at the device's highest priority).  One JSON line.
  python scripts/tables_policy_probe.py [label]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import workloads
from inflatox_amd import _native

label = sys.argv[1] if len(sys.argv) > 1 else "default"
stream = torch.cuda.Stream()
out = {"label": label, "env": {k: v for k, v in os.environ.items() if k.startswith("INFLX_EXPERIMENT")}}
for name, n, P, reps in (("doc", 4096, 1, 30), ("egno", 4096, 1, 30), ("egno", 2048, 1, 30), ("d5", 4096, 1, 30), ("d5", 4096, 32, 3), ("angular", 4096, 1, 30), ("doc", 1000, 1, 50)):
    spec, art = workloads.artifact_for(name)
    lib = _native.InflatoxDevLib(art.shared_object_path)
    rows = np.tile(np.asarray(spec.args, dtype=np.float64), (P, 1))
    if name == "d5" and P > 1:
        rows[:, 6] = np.linspace(2.5e-4, 1e-3, P)
    buf = torch.empty((P, n, n, 6), dtype=torch.float64, device="cuda")
    kw = dict(stream=stream.cuda_stream)
    t = lambda **k: lib.sweep_device_timed(_native.OP_COMPLETE, rows, buf.data_ptr(), buf.numel() * 8, spec.extent, n, n, **kw, **k)  # noqa: E731
    t(repeats=reps)
    back = min(t(repeats=reps) for _ in range(3))
    inside = min(t(repeats=reps, in_pipeline=True) for _ in range(3))
    single = min(t(repeats=max(3, reps // 2), single_call=True) for _ in range(3))
    out[f"{name}_{n}_x{P}"] = {"back_to_back_ms": round(back, 5), "tile_kernel_in_pipeline_ms": round(inside, 5), "single_call_ms": round(single, 5)}
    del buf, lib
    torch.cuda.empty_cache()
print(json.dumps(out))
