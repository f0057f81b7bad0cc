#!/usr/bin/env python3
"""Host time to ENQUEUE one device-resident sweep against the device time it takes: how far the host runs ahead of the GPU in a loop of
sweeps (why the path has no use for hipGraph replay)."""
import sys, time, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import workloads
from inflatox_amd import _native
for name, n in (("hyperbolic", 8192), ("doc", 4096)):
    spec, art = workloads.artifact_for(name)
    lib = _native.InflatoxDevLib(art.shared_object_path)
    buf = torch.empty((n, n, 6), dtype=torch.float64, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    args = np.asarray(spec.args, dtype=np.float64)
    for _ in range(20):
        lib.sweep_device(_native.OP_COMPLETE, args, buf.data_ptr(), buf.numel() * 8, spec.extent, n, n, stream=st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        lib.sweep_device(_native.OP_COMPLETE, args, buf.data_ptr(), buf.numel() * 8, spec.extent, n, n, stream=st)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name} {n}^2: host time to enqueue one sweep {1e6 * (t1 - t0) / 100:.1f} us; device time per sweep {1e6 * (t2 - t0) / 100:.1f} us")
