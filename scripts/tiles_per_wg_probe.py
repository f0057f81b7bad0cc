#!/usr/bin/env python3
"""Vertically consecutive tiles per workgroup of the tile kernels (InflxSweepArgs::reserved0, experiment knob
INFLX_EXPERIMENT_TILES_PER_WG of csrc/inflx_hip.cpp): device time of the complete_analysis sweep per setting, interleaved rounds,
and a bit-for-bit comparison of every setting's result with the one-tile-per-workgroup result.
usage: tiles_per_wg_probe.py [model[:P] ...]      default: d5:32 egno:32 doc:16 d5:1 egno:1 doc:1"""
import os
import sys

import numpy as np
import torch

os.environ["INFLX_EXPERIMENT_TILES_PER_WG"] = "1"  # arms the knob (the library looks for it once, at its first tile launch)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import workloads  # noqa: E402
from inflatox_amd import _native  # noqa: E402

SETTINGS = (1, 2, 4, 8, 16, 128)
n = int(os.environ.get("INFLX_EXPERIMENT_N", "4096"))
for case in sys.argv[1:] or ["d5:32", "egno:32", "doc:16", "d5:1", "egno:1", "doc:1"]:
    name, _, p = case.partition(":")
    P = int(p or 1)
    spec, art = workloads.artifact_for(name)
    lib = _native.InflatoxDevLib(art.shared_object_path)
    rows = np.tile(np.asarray(spec.args, dtype=np.float64), (P, 1))
    if name == "d5" and P > 1:
        rows[:, 6] = np.linspace(2.5e-4, 1e-3, P)
    buf = torch.empty((P, n, n, 6), dtype=torch.float64, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    # correctness first: every setting writes the same bits (ragged grid too)
    ref = None
    for t in SETTINGS:
        os.environ["INFLX_EXPERIMENT_TILES_PER_WG"] = str(t)
        small = torch.full((min(P, 2), 1000, 777, 6), -3.0, dtype=torch.float64, device="cuda:0")
        lib.sweep_device(_native.OP_COMPLETE, rows[: small.shape[0]], small.data_ptr(), small.numel() * 8, spec.extent, 1000, 777, stream=stream)
        torch.cuda.synchronize()
        if ref is None:
            ref = small.clone()
        else:
            same = (small == ref) | (torch.isnan(small) & torch.isnan(ref))
            assert bool(same.all()), (case, t)
        del small
    best = {t: float("inf") for t in SETTINGS}
    for _ in range(3):
        for t in SETTINGS:
            os.environ["INFLX_EXPERIMENT_TILES_PER_WG"] = str(t)
            ms = lib.sweep_device_timed(_native.OP_COMPLETE, rows, buf.data_ptr(), buf.numel() * 8, spec.extent, n, n, stream=stream, repeats=max(4, int(40 / P)))
            best[t] = min(best[t], ms)
    os.environ["INFLX_EXPERIMENT_TILES_PER_WG"] = "1"
    print(f"{name:6s} {n}^2 x {P}: " + "   ".join(f"{t:3d} tiles/wg {best[t]:8.4f} ms" for t in SETTINGS) + "   (results bit-identical)", flush=True)
    del buf, lib
    torch.cuda.empty_cache()
