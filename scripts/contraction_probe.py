#!/usr/bin/env python3
"""(GPU box) Whose bits does the GPU reproduce where the reference's two builds differ?  The reference compiles its generated C
with `zig cc` = clang, which fuses a*b+c inside an expression (-ffp-contract=on); gcc -std=c17 does not.  For both emission styles
of the transpiler -- Compiler(contraction="statement") (round 1-5: a product the stager made a variable of is added already
rounded) and "expression" (that product is spelled out in the sum's statement, so that hipcc fuses what clang fuses) -- on the
reference-generated golden grids: values bit-equal to the gcc build / to the clang build, the side the GPU takes where the two
builds disagree about NaN, the largest |gpu - ref| / |ref| against each, and the device time of a 4096^2 sweep.
  python scripts/contraction_probe.py > profiles/r06_contraction.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import workloads
from inflatox_amd import _native

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
TAGS = {"doc": ("g16", "g64", "neg"), "angular": ("g16", "g64", "inner"), "egno": ("g16", "g64"), "d5": ("g16", "g64", "off")}
out = {"what": __doc__.split("\n  python")[0]}
stream = torch.cuda.Stream()
for name, tags in TAGS.items():
    g = dict(np.load(os.path.join(GOLD, f"{name}.npz")))
    rec = {}
    for style in ("statement", "expression"):
        spec, art = workloads.artifact_for(name, contraction=style)
        lib = _native.InflatoxDevLib(art.shared_object_path)
        cell = {"code_object": os.path.splitext(os.path.basename(art.header_path))[0], "stage_values": [art.stage_info[k] for k in ("nu", "nr", "nc")]}
        for key, op in (("raw", _native.OP_RAW), ("out", _native.OP_COMPLETE)):
            tot = {"values": 0, "bit_equal_gcc": 0, "bit_equal_clang": 0, "builds_differ": 0, "nan_disagreements_of_the_builds": 0, "gpu_nan_like_gcc": 0, "gpu_nan_like_clang": 0,
                   "max_rel_vs_gcc": 0.0, "max_rel_vs_clang": 0.0, "above_1e-10_vs_gcc": 0, "above_1e-10_vs_clang": 0}
            for tag in tags:
                n0, n1 = (int(v) for v in g[f"{tag}_shape"])
                got = lib.sweep_host(op, g["args"], g[f"{tag}_extent"], n0, n1)
                a, b = g[f"{tag}_{key}"], g[f"{tag}_{key}_clang"]
                tot["values"] += got.size
                same = lambda x, y: (x == y) | (np.isnan(x) & np.isnan(y))  # noqa: E731
                tot["bit_equal_gcc"] += int(same(got, a).sum())
                tot["bit_equal_clang"] += int(same(got, b).sum())
                tot["builds_differ"] += int((~same(a, b)).sum())
                nd = np.isnan(a) != np.isnan(b)
                tot["nan_disagreements_of_the_builds"] += int(nd.sum())
                tot["gpu_nan_like_gcc"] += int((np.isnan(got) == np.isnan(a))[nd].sum())
                tot["gpu_nan_like_clang"] += int((np.isnan(got) == np.isnan(b))[nd].sum())
                for ref, lab in ((a, "gcc"), (b, "clang")):
                    fin = np.isfinite(ref) & np.isfinite(got) & (ref != 0)
                    if fin.any():
                        rel = np.abs(got[fin] - ref[fin]) / np.abs(ref[fin])
                        tot[f"max_rel_vs_{lab}"] = max(tot[f"max_rel_vs_{lab}"], float(rel.max()))
                        tot[f"above_1e-10_vs_{lab}"] += int((rel > 1e-10).sum())
            cell[key] = tot
        n = 4096
        buf = torch.empty((n, n, 6), dtype=torch.float64, device="cuda")
        t = lambda: lib.sweep_device_timed(_native.OP_COMPLETE, spec.args, buf.data_ptr(), buf.numel() * 8, spec.extent, n, n, stream=stream.cuda_stream, repeats=30)  # noqa: E731
        t()
        cell["ms_4096"] = round(min(t() for _ in range(3)), 4)
        del buf
        rec[style] = cell
    out[name] = rec
    print(name, json.dumps(rec), file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
