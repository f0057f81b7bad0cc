#!/usr/bin/env python3
"""Long dispatches of the tile kernel for the instruction-mix / shader-clock counters (scripts/profile_isa_mix.sh).

usage: isa_mix_probe.py MODEL:N:P[:tuned] ...     (default: d5:4096:32 egno:4096:32 doc:4096:64 and their tuned builds)

Each case is ONE call of P parameter rows (the BASELINE configs[2] call for D5; for the others the same row P times, so
that a dispatch lasts >= 10 ms: the effective shader clock GRBM_GUI_ACTIVE / 8 / duration is within 3 % of the in-kernel
clock only for dispatches that long, MI355X_MICROARCH.md "DVFS give-back").  The call is repeated REPEATS times back to
back; the report takes the last dispatch of each case."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from inflatox_amd import _native  # noqa: E402
import workloads  # noqa: E402

REPEATS = int(os.environ.get("INFLX_PROBE_REPEATS", "4"))
DEFAULT = ["d5:4096:32", "egno:4096:32", "doc:4096:64", "d5:4096:32:tuned", "egno:4096:32:tuned", "doc:4096:64:tuned"]
cases = [a.split(":") for a in (sys.argv[1:] or DEFAULT)]
stream = torch.cuda.current_stream().cuda_stream
stamps = []
for case in cases:
    name, n, P = case[0], int(case[1]), int(case[2])
    tuned = len(case) > 3 and case[3] == "tuned"
    spec, art = workloads.artifact_for(name, tuned=tuned)
    lib = _native.InflatoxDevLib(art.shared_object_path)
    rows = np.tile(np.asarray(spec.args, dtype=np.float64), (P, 1))
    if name == "d5" and P > 1:
        rows[:, 6] = np.linspace(2.5e-4, 1e-3, P)  # a1: BASELINE configs[2]
    out = torch.empty((P, n, n, 6), dtype=torch.float64, device="cuda:0")
    for _ in range(REPEATS):
        lib.sweep_device(_native.OP_COMPLETE, rows, out.data_ptr(), out.numel() * 8, spec.extent, n, n, stream=stream)
    torch.cuda.synchronize()
    stamps.append({"case": ":".join(case), "model": name, "n": n, "P": P, "tuned": tuned, "repeats": REPEATS,
                   "code_object": os.path.splitext(os.path.basename(art.header_path))[0], "regrouped": art.stage_info.get("regrouped")})
    print("swept", case, flush=True)
    del out, lib
    torch.cuda.empty_cache()
if os.environ.get("INFLX_PROBE_STAMP"):
    json.dump(stamps, open(os.environ["INFLX_PROBE_STAMP"], "w"))
