#!/usr/bin/env python3
"""Where a large default call (result above the result pool's 512 MiB per-array bound) spends its time: the call itself with the
result kept, dropping the result (munmap), and how much of the result sits in transparent huge pages.
usage: big_result_probe.py [model] [n]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import workloads  # noqa: E402
from inflatox_amd.consistency_conditions import GeneralisedAL, _start_stop  # noqa: E402


def huge_kb():
    with open("/proc/self/smaps_rollup") as fh:
        for line in fh:
            if line.startswith("AnonHugePages"):
                return int(line.split()[1])
    return -1


name = sys.argv[1] if len(sys.argv) > 1 else "doc"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
for knob in ("enabled", "defrag", "shmem_enabled"):
    try:
        print(f"transparent_hugepage/{knob}: {open('/sys/kernel/mm/transparent_hugepage/' + knob).read().strip()}")
    except OSError as exc:
        print(knob, exc)
spec, art = workloads.artifact_for(name)
al = GeneralisedAL(art)
al.complete_analysis(spec.args, *spec.extent, 256, 256, progress=False)
ss = _start_stop(*spec.extent)
args = np.ascontiguousarray(spec.args, dtype=np.float64)
for rep in range(4):
    h0 = huge_kb()
    t0 = time.perf_counter()
    res = al.complete_analysis(spec.args, *spec.extent, n, n, progress=False)
    t1 = time.perf_counter()
    h1 = huge_kb()
    del res
    t2 = time.perf_counter()
    print(f"front-end call {1e3 * (t1 - t0):7.2f} ms (result kept)   drop {1e3 * (t2 - t1):7.2f} ms   AnonHugePages +{(h1 - h0) / 1024:.0f} MiB of {48 * n * n / 2**20:.0f}", flush=True)
for label, make in (("np.zeros", lambda: np.zeros((n, n, 6))), ("np.empty", lambda: np.empty((n, n, 6)))):
    for rep in range(2):
        t0 = time.perf_counter()
        out = make()
        t1 = time.perf_counter()
        al.dylib.complete_analysis(args, out, ss, False, 0)
        t2 = time.perf_counter()
        al.dylib.complete_analysis(args, out, ss, False, 0)
        t3 = time.perf_counter()
        h = huge_kb()
        del out
        t4 = time.perf_counter()
        print(f"{label}: alloc {1e3 * (t1 - t0):6.2f} ms   C call into fresh pages {1e3 * (t2 - t1):7.2f} ms   again (resident) {1e3 * (t3 - t2):7.2f} ms   drop {1e3 * (t4 - t3):6.2f} ms   AnonHugePages {h / 1024:.0f} MiB", flush=True)
