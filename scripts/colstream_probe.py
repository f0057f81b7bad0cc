#!/usr/bin/env python3
"""(GPU box) Time of the column-broadcast path -- inflx_sweep_colvals_* + inflx_sweep_colstream -- on a synthetic model
none of whose values depends on x[0] (the mirror image of the README's hyperbolic model; tests/test_models_extra.py holds
the same definition), N x N grid, AoS and planes.   usage: colstream_probe.py [N]"""
import os
import sys

import numpy as np
import sympy as sp
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from inflatox_amd import _native  # noqa: E402
from inflatox_amd.compiler import Compiler  # noqa: E402
from inflatox_amd.symbolic import InflationModelBuilder  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
x, y = sp.symbols("x y", real=True)
a, b = sp.symbols("a b", real=True)
model = InflationModelBuilder.new([x, y], [[1 + y**2, 0], [0, 1]], a * (y - b) ** 2 / 2, model_name="column_only", silent=True, init_sympy_printing=False, simplify=False, assertions=False).build()
art = Compiler(model, silent=True).compile()
lib = _native.InflatoxDevLib(art.shared_object_path)
args, ext = np.array([1.3, 0.4]), (-1.0, 1.0, -2.0, 1.5)
stream = torch.cuda.current_stream().cuda_stream
out = torch.empty((n, n, 6), dtype=torch.float64, device="cuda:0")
plan = lib.sweep_plan(_native.OP_COMPLETE, 1, n, n)
for layout, name in ((_native.LAYOUT_AOS, "AoS"), (_native.LAYOUT_SOA, "planes")):
    whole = min(lib.sweep_device_timed(_native.OP_COMPLETE, args, out.data_ptr(), out.numel() * 8, ext, n, n, layout=layout, stream=stream, repeats=30) for _ in range(3))
    kernel = min(lib.sweep_device_timed(_native.OP_COMPLETE, args, out.data_ptr(), out.numel() * 8, ext, n, n, layout=layout, stream=stream, repeats=30, dominant_only=True) for _ in range(3))
    gb = 48.0 * n * n / 1e9
    print(f"{name:6s} {n}x{n}: sweep {whole:.4f} ms = {gb / whole * 1e3:.0f} GB/s ... copy stream alone {kernel:.4f} ms = {gb / kernel * 1e3:.0f} GB/s   path={plan.get('path')}", flush=True)
