#!/usr/bin/env python3
"""Experiment (GPU box): tile-kernel time under the AMDGPU back-end's alternative scheduling strategies."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from inflatox_amd import _native  # noqa: E402
import workloads  # noqa: E402
from workloads import example_models  # noqa: E402
from inflatox_amd.compiler import Compiler  # noqa: E402

n = 4096
stream = torch.cuda.Stream()
out = torch.empty((n, n, 6), dtype=torch.float64, device="cuda:0")
for name in sys.argv[1:] or ["d5", "egno", "doc"]:
    spec = example_models.get(name)
    for strategy in ("default", "max-ilp", "max-memory-clause", "iterative-ilp", "iterative-minreg", "iterative-maxocc"):
        flags = list(Compiler.default_hipcc_flags) + ([] if strategy == "default" else ["-mllvm", f"-amdgpu-sched-strategy={strategy}"])
        try:
            art = Compiler(workloads.model_for(name), silent=True, compiler_flags=flags, **spec.compiler_kwargs).compile()
        except Exception as exc:  # noqa: BLE001
            print(f"{name:6s} {strategy:18s} compile failed: {str(exc)[:60]}", flush=True)
            continue
        lib = _native.InflatoxDevLib(art.shared_object_path)
        ms = min(lib.sweep_device_timed(_native.OP_COMPLETE, spec.args, out.data_ptr(), out.numel() * 8, spec.extent, n, n, stream=stream.cuda_stream, repeats=30) for _ in range(3))
        print(f"{name:6s} {strategy:18s} {ms:7.3f} ms  {n * n / ms / 1e6:7.2f} Gpts/s", flush=True)
