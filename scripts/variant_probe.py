#!/usr/bin/env python3
"""One complete_analysis sweep (4096^2 by default) per build variant of an example model -- the workload for rocprofv3
--pmc passes that compare variants of the tile kernels (scripts/variant_report.py reads the databases).
usage: variant_probe.py MODEL[:flag,flag...] ...   (flags as in scripts/hoist_experiment.py)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from inflatox_amd import _native  # noqa: E402
import workloads  # noqa: E402
from workloads import example_models  # noqa: E402
from inflatox_amd.compiler import Compiler  # noqa: E402

n = int(os.environ.get("INFLX_EXPERIMENT_N", "4096"))
stream = torch.cuda.current_stream().cuda_stream
out = torch.empty((n, n, 6), dtype=torch.float64, device="cuda:0")
for case in sys.argv[1:]:
    name, _, fl = case.partition(":")
    fl = set(fl.split(",")) - {""}
    spec = example_models.get(name)
    kw = dict(spec.compiler_kwargs)
    flags = list(Compiler.default_hipcc_flags)
    for f in fl:
        if f.startswith("D"):
            flags.append("-" + f)
    hoist = True if "hoist" in fl else (False if "nohoist" in fl else None)
    art = Compiler(workloads.model_for(name), silent=True, compiler_flags=flags, hoist_reciprocals=hoist, **kw).compile()
    lib = _native.InflatoxDevLib(art.shared_object_path)
    for _ in range(2):
        lib.sweep_device(_native.OP_COMPLETE, np.asarray(spec.args), out.data_ptr(), out.numel() * 8, spec.extent, n, n, stream=stream)
        torch.cuda.synchronize()
    print("swept", case, flush=True)
    del lib
