#!/usr/bin/env python3
"""Experiment (GPU box): speed and parity of Compiler(regroup=True) (re-associated products/sums)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import tolerance as tol  # noqa: E402
from inflatox_amd import _native  # noqa: E402
import workloads  # noqa: E402
from workloads import example_models  # noqa: E402
from inflatox_amd.compiler import Compiler  # noqa: E402

n = 4096
out = torch.empty((n, n, 6), dtype=torch.float64, device="cuda:0")
stream = torch.cuda.current_stream().cuda_stream
for name in sys.argv[1:] or ["doc", "angular", "egno", "d5"]:
    spec = example_models.get(name)
    om, _ = tol._models(name)
    x0a, x0b, x1a, x1b = spec.extent
    n0, n1, ext = 45, 333, (x0a + 0.013 * (x0b - x0a), x0b, x1a + 0.007 * (x1b - x1a), x1b)
    pts = oracle.grid_points(ext, n0, n1)
    env, flaky = tol.reference_error(name, spec.args, pts)
    env, flaky = env.reshape(n0, n1, 5), flaky.reshape(n0, n1, 5)
    ref_raw = om.grid_sweep(oracle.OP.RAW, spec.args, ext, n0, n1)
    for regroup in (False, True):
        kw = dict(spec.compiler_kwargs)
        c = Compiler(workloads.model_for(name), silent=True, regroup=regroup, **kw)
        art = c.compile()
        lib = _native.InflatoxDevLib(art.shared_object_path)
        ms = min(lib.sweep_device_timed(_native.OP_COMPLETE, spec.args, out.data_ptr(), out.numel() * 8, spec.extent, n, n, stream=stream, repeats=10) for _ in range(3))
        got = lib.sweep_host(_native.OP_RAW, spec.args, ext, n0, n1)
        nan_mismatch = int((np.isnan(got) != np.isnan(ref_raw)).sum())
        firm_mismatch = int(((np.isnan(got) != np.isnan(ref_raw)) & ~flaky).sum())
        with np.errstate(all="ignore"):
            ratio = np.abs(got - ref_raw) / tol.allowance_raw(ref_raw, env)
        ok = np.isfinite(ratio)
        print(f"{name:8s} regroup={regroup!s:5s}: {ms:7.3f} ms {n * n / ms / 1e6:7.2f} Gpts/s  statements {c.stage_info['statements']}  NaN mismatches {nan_mismatch} (outside flaky points {firm_mismatch})  max |gpu-ref|/allowance {ratio[ok].max():.2f}", flush=True)
