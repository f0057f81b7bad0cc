#!/usr/bin/env python3
"""Where do the two builds of the reference (gcc, clang) disagree about NaN?  CPU only, from the committed goldens.

For every golden grid of a model: the grid points at which the NaN-ness of a model value or of one of the six outputs differs
between the gcc-built and the clang-built reference, with the point, theta / pi (D5's singular lines are theta = k pi / 2) and both
builds' values -- the "explanation" of every NaN-pattern difference the parity suite waives as flaky (tests/tolerance.py).

    python tests/tools/nan_disagreement_report.py d5 > profiles/r05_d5_nan_points.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden  # noqa: E402

import oracle  # noqa: E402

RAW = ("V", "v00", "v10", "v11", "|dV|^2")
OUT = ("consistency", "eps_V", "eps_H", "eta", "delta", "omega")
name = sys.argv[1] if len(sys.argv) > 1 else "d5"
g = golden(name)
for tag in ("g16", "g64"):
    n0, n1 = (int(v) for v in g[f"{tag}_shape"])
    pts = oracle.grid_points(g[f"{tag}_extent"], n0, n1).reshape(n0, n1, 2)
    for key, names in (("raw", RAW), ("out", OUT)):
        a, b = g[f"{tag}_{key}"], g[f"{tag}_{key}_clang"]
        diff = np.argwhere(np.isnan(a) != np.isnan(b))
        print(f"{name} {tag} {n0}x{n1} {key}: NaN-ness differs at {len(diff)} of {a.size} values")
        for i, j, k in diff:
            x0, x1 = pts[i, j]
            raw_a, raw_b = g[f"{tag}_raw"][i, j], g[f"{tag}_raw_clang"][i, j]
            print(f"   ({i:2d},{j:2d}) x0 = {x0:.6g}, x1 = {x1:.9g} = {x1 / np.pi:.6f} pi   {names[k]}: gcc {a[i, j, k]!r:>24}  clang {b[i, j, k]!r:>24}"
                  f"   [v10 gcc {raw_a[2]:.3e} clang {raw_b[2]:.3e}; v00 {raw_a[1]:.3e}]")
