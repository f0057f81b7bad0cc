#!/usr/bin/env python3
"""Load generator of scripts/power_probe.sh: one workload swept back to back for SECONDS, device time per sweep printed
for every batch (with wall-clock stamps, to line up with the rocm-smi samples).  usage: power_load.py MODEL N P SECONDS"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import workloads
from inflatox_amd import _native

name, n, P, secs = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
variant = name.split(":")[1] if ":" in name else ""
name = name.split(":")[0]
if variant in ("inline", "hoist", "nohoist"):  # Compiler(hoist_reciprocals=...) builds (scripts/hoist_experiment.py compiles them into the cache)
    from inflatox_amd.compiler import Compiler
    from workloads import example_models

    spec = example_models.get(name)
    art = Compiler(workloads.model_for(name), silent=True, hoist_reciprocals={"inline": "inline", "hoist": True, "nohoist": False}[variant], **spec.compiler_kwargs).compile()
else:
    spec, art = workloads.artifact_for(name, tuned=True) if variant == "tuned" else workloads.artifact_for(name)
lib = _native.InflatoxDevLib(art.shared_object_path)
rows = np.tile(np.asarray(spec.args, dtype=np.float64), (P, 1))
if name == "d5" and P > 1:
    rows[:, 6] = np.linspace(2.5e-4, 1e-3, P)
buf = torch.empty((P, n, n, 6), dtype=torch.float64, device="cuda")
stream = torch.cuda.Stream()
label = name + (":" + variant if variant else "")
time.sleep(1.0)  # idle samples first
t_end = time.time() + secs
reps = max(2, int(100 / P))
while time.time() < t_end:
    ms = lib.sweep_device_timed(_native.OP_COMPLETE, rows, buf.data_ptr(), buf.numel() * 8, spec.extent, n, n, stream=stream.cuda_stream, repeats=reps)
    print(f"load t={time.time():.3f} {label} {n}x{n}x{P}: {ms:.4f} ms/sweep", flush=True)
time.sleep(0.5)
