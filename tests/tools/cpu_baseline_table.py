#!/usr/bin/env python3
"""CPU restatement of the reference path (oracle) timed on the host for the BASELINE.md table."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from bench import host_threads  # noqa: E402
import workloads  # noqa: E402
from workloads import example_models  # noqa: E402

threads = host_threads()
print(f"host threads used: {threads} (os.cpu_count() = {os.cpu_count()})")
for name, n in (("hyperbolic", 256), ("hyperbolic", 2048), ("doc", 1024), ("angular", 1024), ("egno", 1024), ("d5", 1024)):
    spec = example_models.get(name)
    src, _ = oracle.emit_c_source(workloads.model_for(name), **spec.compiler_kwargs)
    om = oracle.OracleModel(oracle.compile_c_model(src))
    for t in (threads, 1):
        nn = n if t > 1 else min(n, 512)
        om.grid_sweep(oracle.OP.COMPLETE, spec.args, spec.extent, nn, nn, threads=t)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            om.grid_sweep(oracle.OP.COMPLETE, spec.args, spec.extent, nn, nn, threads=t)
            best = min(best, time.perf_counter() - t0)
        print(f"{name:10s} {nn}x{nn} threads={t:3d}: {nn * nn / best / 1e6:9.2f} Mpts/s", flush=True)
