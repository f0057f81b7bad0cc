#!/usr/bin/env python3
"""How long does the first-launches transient of the store stream last?  Per-step device time (HIP events on the launch
stream) of the first 400 back-to-back 8192^2 hyperbolic sweeps of a fresh process, printed as block averages."""
import os
import sys

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
steps = 400
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
torch.cuda.synchronize()
ev[0].record(st)
for k in range(steps):
    lib.sweep_device(_native.OP_COMPLETE, spec.args, out.data_ptr(), out.numel() * 8, spec.extent, n, n, stream=st.cuda_stream)
    ev[k + 1].record(st)
torch.cuda.synchronize()
ms = np.array([ev[k].elapsed_time(ev[k + 1]) for k in range(steps)])
print("first 10 steps:", np.round(ms[:10], 3))
for a in range(0, steps, 20):
    print(f"steps {a:3d}-{a + 19:3d}: mean {ms[a:a + 20].mean():.4f} ms  min {ms[a:a + 20].min():.4f}  max {ms[a:a + 20].max():.4f}")
