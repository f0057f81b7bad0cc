#!/usr/bin/env python3
"""Experiment (GPU box): row-broadcast kernel vs chunk size / store flavour (grid via INFLX_ROW_STREAM_GRID)."""
import itertools
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from inflatox_amd import _native  # noqa: E402
import workloads  # noqa: E402
from workloads import example_models  # noqa: E402
from inflatox_amd.compiler import Compiler  # noqa: E402

n = 8192
stream = torch.cuda.current_stream().cuda_stream
out = torch.empty((n, n, 6), dtype=torch.float64, device="cuda:0")
spec = example_models.get("hyperbolic")


def torch_ms(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for label, fn in (("torch zero_ (memset)", lambda: out.zero_()), ("torch fill_(1.0)", lambda: out.fill_(1.0))):
    ms = torch_ms(fn)
    print(f"baseline {label}: {ms:.4f} ms {48 * n * n / ms / 1e6:7.1f} GB/s", flush=True)
art = Compiler(workloads.model_for("hyperbolic"), silent=True).compile()
lib = _native.InflatoxDevLib(art.shared_object_path)
for dom in (False, True):
    ms = sorted(lib.sweep_device_timed(_native.OP_COMPLETE, spec.args, out.data_ptr(), out.numel() * 8, spec.extent, n, n, stream=stream, repeats=20, dominant_only=dom) for _ in range(5))
    print(f"row path {'store stream only' if dom else 'rowvals + stream'}: min {ms[0]:.4f} med {ms[2]:.4f} ms  {48 * n * n / ms[2] / 1e6:7.1f} GB/s", flush=True)
