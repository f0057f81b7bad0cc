#!/usr/bin/env python3
"""Wave-slot occupancy of a stamped diagnostic build of the tile kernel: sum of wave lifetimes against the span of the dispatch.
usage: diag_slots.py MODEL ARTEFACT N P"""
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
loop, life, prologue, first_c, last, waves = (int(v) for v in s["count"])
first = (1 << 62) - first_c
span = last - first
slots = 1024 * 3
print(f"{name} x {P}: {waves} waves; dispatch span {span} cycles; per wave: lifetime {life / waves:.0f}, of which prologue {prologue / waves:.0f}, row loop {loop / waves:.0f}, "
      f"rest (redo loop, exit) {(life - prologue - loop) / waves:.0f}")
print(f"   wave-slot occupancy = sum of lifetimes / (span x {slots} slots) = {life / (span * slots):.3f};  row-loop share of all slot time = {loop / (span * slots):.3f}")
