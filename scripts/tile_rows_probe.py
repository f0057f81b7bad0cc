#!/usr/bin/env python3
"""Tile height as a launch argument (InflxSweepArgs::tile_rows): device time of the complete_analysis sweep per height, interleaved
rounds, for launches of a few workgroup rounds (experiment knob INFLX_EXPERIMENT_TILE_ROWS of csrc/inflx_hip.cpp).
usage: tile_rows_probe.py [--small] [model ...]"""
import os
import sys

import numpy as np
import torch

os.environ["INFLX_EXPERIMENT_TILE_ROWS"] = "0"  # arms the knob (the library looks for it once, at its first tile launch); 0 = the library's own choice

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import workloads  # noqa: E402
from inflatox_amd import _native  # noqa: E402

argv = sys.argv[1:]
small = argv[:1] == ["--small"]
argv = argv[1:] if small else argv
HEIGHTS = (1, 2, 4, 8, 16, 32) if small else (8, 12, 16, 24, 32)
CASES = ((128, 1), (256, 1), (512, 1), (724, 1), (1000, 1), (256, 8)) if small else ((1448, 1), (2048, 1), (4096, 1), (4096, 4))
for name in argv or ["doc", "egno", "d5"]:
    spec, art = workloads.artifact_for(name)
    lib = _native.InflatoxDevLib(art.shared_object_path)
    for n, P in CASES:
        rows = np.tile(np.asarray(spec.args, dtype=np.float64), (P, 1))
        buf = torch.empty((P, n, n, 6), dtype=torch.float64, device="cuda:0")
        stream = torch.cuda.current_stream().cuda_stream
        best = {h: float("inf") for h in HEIGHTS + ("rule",)}
        for _ in range(4):
            for h in HEIGHTS + ("rule",):
                if h == "rule":  # the height launch_tiles chooses by itself
                    os.environ["INFLX_EXPERIMENT_TILE_ROWS"] = "0"
                else:
                    os.environ["INFLX_EXPERIMENT_TILE_ROWS"] = str(h)
                ms = lib.sweep_device_timed(_native.OP_COMPLETE, rows, buf.data_ptr(), buf.numel() * 8, spec.extent, n, n, stream=stream, repeats=max(5, int(40 / P)) * (4 if small else 1))
                best[h] = min(best[h], ms)
        os.environ["INFLX_EXPERIMENT_TILE_ROWS"] = "0"
        print(f"{name:6s} {n}^2 x {P}: " + "   ".join(f"{h:2d} rows {best[h]:7.4f} ms" for h in HEIGHTS) + f"   | launch_tiles' own choice {best['rule']:7.4f} ms", flush=True)
        del buf
