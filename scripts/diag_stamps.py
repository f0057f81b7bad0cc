#!/usr/bin/env python3
"""Read the s_memtime stamps of a diagnostic build of the tile kernel (built by hand, see profiles/r04_experiments.txt section 7):
per-wave cycle sums of the row loop's segments, delivered in the `count` slots of the summary.  usage: diag_stamps.py MODEL ARTEFACT N P"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from inflatox_amd import _native  # noqa: E402
import workloads  # noqa: E402

name, path, n, P = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
spec = workloads.example_models.get(name)
rows = np.tile(np.asarray(spec.args, dtype=np.float64), (P, 1))
if name == "d5" and P > 1:
    rows[:, 6] = np.linspace(2.5e-4, 1e-3, P)
lib = _native.InflatoxDevLib(path)
buf = torch.empty((P, n, n, 6), dtype=torch.float64, device="cuda:0")
for _ in range(3):
    s = lib.sweep_stats(rows, spec.extent, n, n, d_out_ptr=buf.data_ptr(), d_out_bytes=buf.numel() * 8)
total, point, epi, emit, redo, waves = (int(v) for v in s["count"])
rows_per_wave = 32
print(f"{name}: waves {waves}; per wave and grid row (cycles of s_memtime): loop {total / waves / rows_per_wave:8.0f}  point stage {point / waves / rows_per_wave:8.0f}  "
      f"epilogue {epi / waves / rows_per_wave:8.0f}  emit {emit / waves / rows_per_wave:8.0f}  |  redo loop per wave {redo / waves:10.0f}  (loop total per wave {total / waves:10.0f})")
print(f"   raw sums per wave: {[int(v) // max(waves, 1) for v in s['count']]}")
print(f"   shares of the loop: point {point / total:.3f} epilogue {epi / total:.3f} emit {emit / total:.3f} rest {(total - point - epi - emit) / total:.3f}; redo / loop {redo / total:.3f}")
