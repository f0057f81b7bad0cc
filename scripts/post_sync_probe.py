#!/usr/bin/env python3
"""(GPU box) What do the first sweeps behind a synchronisation cost in a WARM process?  bench.py's timed region starts right
behind the mandatory torch.cuda.synchronize(): 64 settling sweeps and 5 warm-up steps, synchronise, then 20 steps with a HIP event
between every two of them -- repeated, with idle gaps of different lengths before the timed steps.  One line per repetition."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import workloads  # noqa: E402
from inflatox_amd import _native  # noqa: E402

spec, art = workloads.artifact_for("hyperbolic")
lib = _native.InflatoxDevLib(art.shared_object_path)
n = 8192
out = torch.empty((n, n, 6), dtype=torch.float64, device="cuda:0")
st = torch.cuda.Stream()
steps = 20
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]


def step():
    lib.sweep_device(_native.OP_COMPLETE, spec.args, out.data_ptr(), out.numel() * 8, spec.extent, n, n, stream=st.cuda_stream)


for gap_ms in (0.0, 0.0, 0.0, 1.0, 10.0, 100.0, 0.0):
    for _ in range(64 + 5):
        step()
    torch.cuda.synchronize()
    if gap_ms:
        time.sleep(gap_ms * 1e-3)
    t0 = time.perf_counter()
    ev[0].record(st)
    for k in range(steps):
        step()
        ev[k + 1].record(st)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    ms = np.array([ev[k].elapsed_time(ev[k + 1]) for k in range(steps)])
    print(f"idle gap {gap_ms:6.1f} ms: wall/step {wall / steps:.4f}  events/step {ms.mean():.4f}  first five {np.round(ms[:5], 3)}  rest mean {ms[5:].mean():.4f}", flush=True)
