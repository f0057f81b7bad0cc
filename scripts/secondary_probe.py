#!/usr/bin/env python3
"""ONE of bench.py's `secondary` workloads (d5 = D5 4096^2 x 32 parameter rows in one call, egno = EGNO 4096^2, doc = the
documentation model 4096^2), run exactly as bench.py runs it -- the workload for scripts/profile_secondary.sh, which puts
each of them under `rocprofv3 --kernel-trace --stats` on its own so that every secondary[*].ms of the benchmark line can
be recomputed from one committed CSV.  Prints the record bench.py would print for it (one JSON line).
usage: secondary_probe.py d5|egno|doc[:tuned]   (":tuned" = the profile-guided build of the workload instead of the default one)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import workloads  # noqa: E402
from inflatox_amd import _native  # noqa: E402

name, _, build = sys.argv[1].partition(":")
if build == "tuned":
    workloads.artifact_for(name, tuned=True)  # the host measurement and the compile, before the device is busy
torch.cuda.set_device(0)
stream = torch.cuda.Stream(device="cuda:0")
rec = bench.secondary_workloads(_native, workloads, torch, np, 0, stream.cuda_stream, only=name, builds=(build or "default",))
torch.cuda.synchronize()
print(json.dumps(rec[0]), flush=True)
