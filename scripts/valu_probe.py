#!/usr/bin/env python3
"""One complete_analysis sweep per example model (AoS, device-resident) -- the workload for the rocprofv3
--pmc passes whose SQ counters scripts/valu_report.py turns into profiles/r01_valu.json.
usage: valu_probe.py [MODEL:N ...]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from inflatox_amd import _native  # noqa: E402
import workloads  # noqa: E402

cases = [a.split(":") for a in sys.argv[1:]] or [["hyperbolic", "8192"], ["doc", "4096"], ["angular", "4096"], ["egno", "4096"], ["d5", "4096"]]
stream = torch.cuda.current_stream().cuda_stream
stamps = {}
for name, n in cases:
    n = int(n)
    spec, art = workloads.artifact_for(name)
    lib = _native.InflatoxDevLib(art.shared_object_path)
    stamps[name] = os.path.splitext(os.path.basename(art.header_path))[0]  # bench.py code_object_id: the cache tag
    out = torch.empty((n, n, 6), dtype=torch.float64, device="cuda:0")
    for _ in range(2):
        lib.sweep_device(_native.OP_COMPLETE, np.asarray(spec.args), out.data_ptr(), out.numel() * 8, spec.extent, n, n, stream=stream)
        torch.cuda.synchronize()
    print("swept", name, n, flush=True)
    del out, lib
if os.environ.get("INFLX_PROBE_STAMP"):
    json.dump(stamps, open(os.environ["INFLX_PROBE_STAMP"], "w"))
