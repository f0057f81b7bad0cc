#!/usr/bin/env python3
"""Experiment (GPU box): tile-kernel time of the opt-in hoisted-reciprocal division against the default,
at 2 and 1 wavefronts per SIMD (256 / 512 registers per lane)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from inflatox_amd import _native, example_models, workloads  # noqa: E402
from inflatox_amd.compiler import Compiler  # noqa: E402

n = 4096
stream = torch.cuda.current_stream().cuda_stream
out = torch.empty((n, n, 6), dtype=torch.float64, device="cuda:0")
for name in sys.argv[1:] or ["d5", "egno"]:
    spec = example_models.get(name)
    for hoist in (False, True):
        for waves in (2, 1):
            kw = dict(spec.compiler_kwargs)
            flags = Compiler.default_hipcc_flags + [f"-DINFLX_MIN_WAVES={waves}"]
            art = Compiler(workloads.model_for(name), silent=True, compiler_flags=flags, hoist_reciprocals=hoist, **kw).compile()
            lib = _native.InflatoxDevLib(art.shared_object_path)
            ms = min(lib.sweep_device_timed(_native.OP_COMPLETE, spec.args, out.data_ptr(), out.numel() * 8, spec.extent, n, n, stream=stream, repeats=30) for _ in range(3))
            print(f"{name:6s} hoist={hoist!s:5s} waves/SIMD={waves}: {ms:7.3f} ms  {n * n / ms / 1e6:7.2f} Gpts/s", flush=True)
