#!/usr/bin/env python3
"""Per-call latency of the drop-in front-end on small workloads (the reference's default 1000x1000 grid,
BASELINE configs[0] = 256x256, a 500-point trajectory): where fixed costs, not bandwidth, decide."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import workloads  # noqa: E402
from inflatox_amd.consistency_conditions import GeneralisedAL  # noqa: E402

for name in sys.argv[1:] or ["hyperbolic", "doc", "d5"]:
    spec, art = workloads.artifact_for(name)
    al = GeneralisedAL(art)
    traj = np.column_stack([np.linspace(spec.extent[0], spec.extent[1], 500, endpoint=False), np.linspace(spec.extent[2], spec.extent[3], 500, endpoint=False)])
    for label, fn in (
        ("complete_analysis 256x256", lambda: al.complete_analysis(spec.args, *spec.extent, 256, 256, progress=False)),
        ("complete_analysis 1000x1000", lambda: al.complete_analysis(spec.args, *spec.extent, 1000, 1000, progress=False)),
        ("consistency 1000x1000", lambda: al.consistency(spec.args, *spec.extent, 1000, 1000, progress=False)),
        ("complete_analysis_ot 500 points", lambda: al.complete_analysis_ot(spec.args, traj, progress=False)),
        ("calc_V (one point)", lambda: al.calc_V(np.array([traj[3, 0], traj[3, 1]]), spec.args)),
    ):
        fn()
        reps = 30
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        dt = (time.perf_counter() - t0) / reps
        print(f"{name:10s} {label:32s} {dt * 1e6:9.1f} us/call", flush=True)
