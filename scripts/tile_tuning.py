#!/usr/bin/env python3
"""Experiment (GPU box): tile-kernel time vs register cap (INFLX_MIN_WAVES), U placement, tile height."""
import itertools
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from inflatox_amd import _native  # noqa: E402
import workloads  # noqa: E402
from workloads import example_models  # noqa: E402
from inflatox_amd.compiler import Compiler  # noqa: E402

models = sys.argv[1:] or ["doc", "angular", "egno", "d5"]
n = 4096
stream = torch.cuda.current_stream().cuda_stream
out = torch.empty((n, n, 6), dtype=torch.float64, device="cuda:0")
for name in models:
    spec = example_models.get(name)
    for waves, ulds, rows in itertools.product((1, 2, 3), (0, 1), (16, 32, 64)):
        flags = Compiler.default_hipcc_flags + [f"-DINFLX_MIN_WAVES={waves}", f"-DINFLX_U_IN_LDS={ulds}", f"-DINFLX_TILE_ROWS={rows}"]
        try:
            art = Compiler(workloads.model_for(name), silent=True, compiler_flags=flags, **spec.compiler_kwargs).compile()
        except Exception as exc:  # noqa: BLE001
            print(name, waves, ulds, rows, "compile failed", str(exc)[:80])
            continue
        lib = _native.InflatoxDevLib(art.shared_object_path)
        ms = min(lib.sweep_device_timed(_native.OP_COMPLETE, spec.args, out.data_ptr(), out.numel() * 8, spec.extent, n, n, stream=stream, repeats=10) for _ in range(3))
        print(f"{name:8s} min_waves={waves} u_in_lds={ulds} tile_rows={rows}: {ms:7.3f} ms  {n * n / ms / 1e6:7.2f} Gpts/s", flush=True)
