#!/usr/bin/env python3
"""Diagnostic (GPU box): error statistics of the HIP sweep against the golden vectors."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from inflatox_amd import _native, workloads  # noqa: E402

for name in sys.argv[1:] or ["hyperbolic", "doc", "angular", "egno", "d5"]:
    spec, art = workloads.artifact_for(name)
    lib = _native.InflatoxDevLib(art.shared_object_path)
    g = np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz"))
    for tag in ("g16", "g64"):
        n0, n1 = (int(v) for v in g[f"{tag}_shape"])
        for op, key in ((_native.OP_RAW, "raw"), (_native.OP_COMPLETE, "out")):
            got = lib.sweep_host(op, g["args"], g[f"{tag}_extent"], n0, n1)
            ref = g[f"{tag}_{key}"]
            fin = np.isfinite(ref) & np.isfinite(got)
            rel = np.abs(got[fin] - ref[fin]) / np.maximum(np.abs(ref[fin]), 1e-300)
            per = []
            for k in range(ref.shape[-1]):
                f = fin[..., k]
                r = np.abs(got[..., k][f] - ref[..., k][f]) / np.maximum(np.abs(ref[..., k][f]), 1e-300)
                per.append("%.1e" % (r.max() if r.size else 0))
            print(
                f"{name:10s} {tag} {key}: nan {np.array_equal(np.isnan(got), np.isnan(ref))} inf {np.array_equal(np.isinf(got), np.isinf(ref))} "
                f"max {rel.max():.2e} p99 {np.quantile(rel, 0.99):.2e} median {np.median(rel):.2e} per-k {per}",
                flush=True,
            )
