#!/usr/bin/env python3
"""Experiment (GPU box): effect of the device FP-contraction mode on the distance to the reference."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import tolerance as tol  # noqa: E402
from inflatox_amd import _native  # noqa: E402
import workloads  # noqa: E402
from inflatox_amd.compiler import Compiler  # noqa: E402

for name in sys.argv[1:] or ["angular", "egno", "d5"]:
    spec, _ = workloads.artifact_for(name)
    om, _ = tol._models(name)
    x0a, x0b, x1a, x1b = spec.extent
    n0, n1, ext = 45, 333, (x0a + 0.013 * (x0b - x0a), x0b, x1a + 0.007 * (x1b - x1a), x1b)
    pts = oracle.grid_points(ext, n0, n1)
    env, flaky = tol.reference_error(name, spec.args, pts)
    env = env.reshape(n0, n1, 5)
    ref_raw = om.grid_sweep(oracle.OP.RAW, spec.args, ext, n0, n1)
    for mode in ("fast", "on", "off"):
        flags = Compiler.default_hipcc_flags + [f"-ffp-contract={mode}"]
        art = Compiler(workloads.model_for(name), silent=True, compiler_flags=flags, **spec.compiler_kwargs).compile()
        lib = _native.InflatoxDevLib(art.shared_object_path)
        got = lib.sweep_host(_native.OP_RAW, spec.args, ext, n0, n1)
        with np.errstate(all="ignore"):
            ratio = np.abs(got - ref_raw) / env
            rel = np.abs(got - ref_raw) / np.abs(ref_raw)
        ok = np.isfinite(ratio)
        print(f"{name:8s} contract={mode:4s}: |gpu-ref|/E  max {ratio[ok].max():8.2f}  p99.9 {np.quantile(ratio[ok], 0.999):7.2f}  median {np.median(ratio[ok]):6.3f}   maxrel {np.nanmax(rel):.2e}", flush=True)
