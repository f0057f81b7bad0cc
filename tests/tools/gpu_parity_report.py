#!/usr/bin/env python3
"""Diagnostic (GPU box): HIP sweep vs oracle, plain relative error and the ratio to the allowance of
tests/tolerance.py (<= 1 passes)."""
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

for name in sys.argv[1:] or ["hyperbolic", "doc", "angular", "egno", "d5"]:
    spec, art = workloads.artifact_for(name)
    lib = _native.InflatoxDevLib(art.shared_object_path)
    om, _ = tol._models(name)
    x0a, x0b, x1a, x1b = spec.extent
    for n0, n1, ext in ((64, 48, spec.extent), (45, 333, (x0a + 0.013 * (x0b - x0a), x0b, x1a + 0.007 * (x1b - x1a), x1b))):
        pts = oracle.grid_points(ext, n0, n1)
        env, flaky = tol.reference_error(name, spec.args, pts)
        env, flaky = env.reshape(n0, n1, 5), flaky.reshape(n0, n1, 5)
        ref_raw = om.grid_sweep(oracle.OP.RAW, spec.args, ext, n0, n1)
        ref_out = om.grid_sweep(oracle.OP.COMPLETE, spec.args, ext, n0, n1)
        got_raw = lib.sweep_host(_native.OP_RAW, spec.args, ext, n0, n1)
        got_out = lib.sweep_host(_native.OP_COMPLETE, spec.args, ext, n0, n1)
        for key, got, ref, allowed, fl in (
            ("raw", got_raw, ref_raw, tol.allowance_raw(ref_raw, env, name), flaky),
            ("out", got_out, ref_out, tol.allowance_derived(ref_raw, env, tol.epilogue, name), flaky.any(axis=-1, keepdims=True)),
        ):
            try:
                tol.check(got, ref, allowed, fl, f"{name}/{key}", model=name)
                verdict = "PASS"
            except AssertionError as exc:
                verdict = f"FAIL {exc}"
                with np.errstate(all="ignore"):
                    rr = np.where(np.isfinite(allowed) & np.isfinite(ref) & np.isfinite(got), np.abs(got - ref) / allowed, 0)
                w = np.unravel_index(np.argmax(rr), rr.shape)
                i, j = w[0], w[1]
                print("   worst", w, "pt", pts.reshape(n0, n1, 2)[i, j], "x1/pi", pts.reshape(n0, n1, 2)[i, j, 1] / 3.14159265359)
                print("   ref raw", ref_raw[i, j], "\n   gpu raw", got_raw[i, j], "\n   env    ", env[i, j])
                tt = oracle.raw_long_double(tol._models(name)[1], spec.args, pts.reshape(n0, n1, 2)[i, j][None])
                print("   ld  raw", tt[0])
            print(f"{name:10s} {n0}x{n1} {key}: {verdict} (flaky points {int(fl.sum())})")
            nan_ok = np.array_equal(np.isnan(got), np.isnan(ref))
            inf_ok = np.array_equal(np.isinf(got), np.isinf(ref))
            fin = np.isfinite(ref) & np.isfinite(got)
            with np.errstate(all="ignore"):
                rel = np.abs(got - ref) / np.abs(ref)
                ratio = np.abs(got - ref) / allowed
            okfin = fin & np.isfinite(allowed)
            per = ["%.1e/%.2f" % (np.nanmax(np.where(fin[..., k], rel[..., k], 0)), np.nanmax(np.where(okfin[..., k], ratio[..., k], 0))) for k in range(ref.shape[-1])]
            print(f"{name:10s} {n0}x{n1} {key}: nan {nan_ok} inf {inf_ok} loose {int((~np.isfinite(allowed) & fin).sum())} maxrel/ratio per-k {per}", flush=True)
