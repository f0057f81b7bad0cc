#!/usr/bin/env python3
"""Latency of SMALL default calls (the reference's own sizes: 256 x 256 of BASELINE configs[0], the default 1000 x 1000):
GeneralisedAL.complete_analysis (front end), the C entry point under it (inflx_complete_analysis on a preallocated array), and
the device-resident sweep + synchronisation, per model.  usage: small_call_probe.py [--sizes 64,256,...] [model ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import workloads  # noqa: E402
from inflatox_amd.consistency_conditions import GeneralisedAL, _start_stop  # noqa: E402


def best_of(fn, repeats=30):
    repeats = repeats if sizes[-1] <= 1000 else 5
    best = float("inf")
    for _ in range(repeats):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best * 1e6


argv = sys.argv[1:]
sizes = (64, 256, 512, 1000)
if argv[:1] == ["--sizes"]:
    sizes, argv = tuple(int(v) for v in argv[1].split(",")), argv[2:]
for name in argv or ["hyperbolic", "doc", "egno"]:
    spec, art = workloads.artifact_for(name)
    al = GeneralisedAL(art)
    ss = _start_stop(*spec.extent)
    args = np.ascontiguousarray(spec.args, dtype=np.float64)
    for n in sizes:
        al.complete_analysis(spec.args, *spec.extent, n, n, progress=False)
        front = best_of(lambda: al.complete_analysis(spec.args, *spec.extent, n, n, progress=False))
        out = np.zeros((n, n, 6))
        c_call = best_of(lambda: al.dylib.complete_analysis(args, out, ss, False, 0))
        dev = best_of(lambda: (al.complete_analysis_device(spec.args, *spec.extent, n, n), torch.cuda.synchronize()))
        print(f"{name:11s} {n:5d}^2  front-end {front:8.1f} us   C call {c_call:8.1f} us   device-resident + sync {dev:8.1f} us   "
              f"({n * n / front:8.1f} Mpts/s front-end)", flush=True)
