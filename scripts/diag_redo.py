#!/usr/bin/env python3
"""Distribution of the IEEE redo rows over wavefronts (diagnostic build, profiles/r04_experiments.txt section 9).  usage: diag_redo.py MODEL ARTEFACT N P"""
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
s = lib.sweep_stats(rows, spec.extent, n, n, d_out_ptr=buf.data_ptr(), d_out_bytes=buf.numel() * 8)
redo, all_rows, some, gave_up, beyond_tile0, waves = (int(v) for v in s["count"])
print(f"{name} x {P}: {waves} wavefronts; redo rows {redo} = {redo / (waves * 32):.4f} of all wavefront-rows; wavefronts that redo every row {all_rows} ({all_rows / waves:.4f}), "
      f">= 30 rows {gave_up} ({gave_up / waves:.4f}), at least one {some} ({some / waves:.4f}), at least one outside the first row tile {beyond_tile0}")
