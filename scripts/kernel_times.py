#!/usr/bin/env python3
"""Kernel-only timings (HIP events) of the complete_analysis sweep for the example models."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from inflatox_amd import _native  # noqa: E402
import workloads  # noqa: E402

cases = [("hyperbolic", 8192, 1), ("hyperbolic", 1024, 1), ("doc", 4096, 1), ("angular", 4096, 1), ("egno", 4096, 1), ("d5", 4096, 1), ("d5", 2048, 4)]
if len(sys.argv) > 1:
    cases = [(a.split(":")[0], int(a.split(":")[1]), int(a.split(":")[2]) if a.count(":") > 1 else 1) for a in sys.argv[1:]]
stream = torch.cuda.current_stream().cuda_stream
for name, n, P in cases:
    spec, art = workloads.artifact_for(name)
    lib = _native.InflatoxDevLib(art.shared_object_path)
    out = torch.empty((P, n, n, 6), dtype=torch.float64, device="cuda:0")
    args = np.tile(spec.args, (P, 1)) * (1.0 + 0.01 * np.arange(P))[:, None]
    for layout, lname in ((_native.LAYOUT_AOS, "aos"), (_native.LAYOUT_SOA, "soa")) if P * n * n * 48 < 40e9 else ((_native.LAYOUT_AOS, "aos"),):
        # best of three runs of 30 back-to-back launches: ten launches sit inside the clock governor's
        # transient (a trace shows 600 -> 690 -> 645 us for the same kernel), thirty reach the steady state
        big = P * n * n * 48 >= 40e9
        ms = min(lib.sweep_device_timed(_native.OP_COMPLETE, args, out.data_ptr(), out.numel() * 8, spec.extent, n, n, layout=layout, stream=stream, repeats=3 if big else 30) for _ in range(1 if big else 3))
        pts = P * n * n
        print(f"{name:10s} {n}x{n} P={P} {lname}: {ms:8.3f} ms  {pts / ms / 1e6:9.2f} Gpts/s  {48 * pts / ms / 1e6:8.1f} GB/s  info={lib.stage_info}", flush=True)
    del out
