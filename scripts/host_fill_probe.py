#!/usr/bin/env python3
"""Host-side broadcast fill (csrc/inflx_hip.cpp: sweep_host_broadcast) against the device-to-host copy for the default
GeneralisedAL.complete_analysis call of a model that ignores one field.  Each setting runs in a process of its own (the
knobs are read once): usage: host_fill_probe.py [N]   -> one line per (INFLX_HOST_FILL, INFLX_HOST_FILL_THREADS)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, time, numpy as np
sys.path.insert(0, %r)
import workloads
from inflatox_amd.consistency_conditions import GeneralisedAL
n = int(sys.argv[1])
spec, art = workloads.artifact_for("hyperbolic")
al = GeneralisedAL(art)
al.complete_analysis(spec.args, *spec.extent, 256, 256, progress=False)
t0 = time.perf_counter(); res = al.complete_analysis(spec.args, *spec.extent, n, n, progress=False); cold = time.perf_counter() - t0
chk = float(np.nansum(res[1][::511, ::509])); del res
best = 1e9
for _ in range(5):
    t0 = time.perf_counter(); res = al.complete_analysis(spec.args, *spec.extent, n, n, progress=False); best = min(best, time.perf_counter() - t0); del res
print(f"cold {cold*1e3:7.1f} ms  warm {best*1e3:7.1f} ms = {48*n*n/best/1e9:6.1f} GB/s  checksum {chk:.6f}")
""" % ROOT

n = sys.argv[1] if len(sys.argv) > 1 else "8192"
for fill, threads in (("0", ""), ("1", "4"), ("1", "8"), ("1", "16"), ("1", "32"), ("1", "64")):
    env = dict(os.environ, INFLX_HOST_FILL=fill)
    if threads:
        env["INFLX_HOST_FILL_THREADS"] = threads
    out = subprocess.run([sys.executable, "-c", CHILD, n], env=env, capture_output=True, text=True)
    print(f"INFLX_HOST_FILL={fill} threads={threads or '-':>3}: {out.stdout.strip() or out.stderr[-400:]}", flush=True)
