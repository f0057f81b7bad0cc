#!/usr/bin/env python3
"""Time given code objects of ONE model on its benchmark call (A/B of hand-made kernel variants; interleaved rounds).
usage: time_artefacts.py MODEL N P ROUNDS ARTEFACT [ARTEFACT ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from inflatox_amd import _native  # noqa: E402
import workloads  # noqa: E402

name, n, P, rounds = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
paths = sys.argv[5:]
spec = workloads.example_models.get(name)
rows = np.tile(np.asarray(spec.args, dtype=np.float64), (P, 1))
if name == "d5" and P > 1:
    rows[:, 6] = np.linspace(2.5e-4, 1e-3, P)
buf = torch.empty((P, n, n, 6), dtype=torch.float64, device="cuda:0")
stream = torch.cuda.current_stream().cuda_stream
libs = [_native.InflatoxDevLib(p) for p in paths]
best = [float("inf")] * len(libs)
ref = None
for lib, p in zip(libs, paths):  # results must agree bit for bit between variants of the same arithmetic
    lib.sweep_device(_native.OP_COMPLETE, rows[:1], buf.data_ptr(), buf.numel() * 8, spec.extent, 512, 512, stream=stream)
    torch.cuda.synchronize()
    got = buf.view(-1)[: 512 * 512 * 6].clone()
    if ref is None:
        ref = got
    else:
        print(os.path.basename(p), "equals first bit for bit:", bool(torch.equal(torch.nan_to_num(got, nan=-7.0), torch.nan_to_num(ref, nan=-7.0))))
reps = max(3, int(60 / P))
for r in range(rounds):
    for k, lib in enumerate(libs):
        ms = lib.sweep_device_timed(_native.OP_COMPLETE, rows, buf.data_ptr(), buf.numel() * 8, spec.extent, n, n, stream=stream, repeats=reps)
        best[k] = min(best[k], ms)
for p, ms in zip(paths, best):
    print(f"{os.path.basename(p):32s} {ms:9.3f} ms  {P * n * n / ms / 1e6:8.2f} Gpts/s", flush=True)
