#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) rate of GeneralisedAL.complete_analysis: Python call -> six numpy arrays."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import workloads  # noqa: E402
from inflatox_amd.consistency_conditions import GeneralisedAL  # noqa: E402

for name, n in (("hyperbolic", 4096), ("hyperbolic", 8192), ("d5", 4096)):
    spec, art = workloads.artifact_for(name)
    al = GeneralisedAL(art)
    al.complete_analysis(spec.args, *spec.extent, 256, 256, progress=False)  # warm-up (module load, buffers)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        res = al.complete_analysis(spec.args, *spec.extent, n, n, progress=False)
        best = min(best, time.perf_counter() - t0)
    print(f"{name} {n}x{n}: {best * 1e3:8.1f} ms end-to-end  {n * n / best / 1e9:6.3f} Gpts/s  {48 * n * n / best / 1e9:6.2f} GB/s (of which np.zeros + D2H)", flush=True)
    del res
