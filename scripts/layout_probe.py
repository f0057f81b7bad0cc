#!/usr/bin/env python3
"""One tile-kernel sweep per layout (AoS, then SoA) for a model -- run under rocprofv3 --pmc to compare
the instruction mix / stall counters of the two store paths.  usage: layout_probe.py MODEL [N]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from inflatox_amd import _native  # noqa: E402
import workloads  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "egno"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
spec, art = workloads.artifact_for(name)
lib = _native.InflatoxDevLib(art.shared_object_path)
out = torch.empty((n, n, 6), dtype=torch.float64, device="cuda:0")
stream = torch.cuda.current_stream().cuda_stream
for layout in (_native.LAYOUT_AOS, _native.LAYOUT_SOA, _native.LAYOUT_AOS, _native.LAYOUT_SOA):
    lib.sweep_device(_native.OP_COMPLETE, np.asarray(spec.args), out.data_ptr(), out.numel() * 8, spec.extent, n, n, layout=layout, stream=stream)
    torch.cuda.synchronize()
print("done")
