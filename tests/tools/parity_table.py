#!/usr/bin/env python3
"""Condense the parity suite's own record (gpurun_out/parity_stats.json, written by tests/conftest.py at the end of a
`-m gpu` run) into the table of profiles/rNN_parity_stats.json: per model and build, the GPU against the reference's C as
gcc builds it, as clang (= the reference's `zig cc`) builds it, and the two builds of the reference against each other
-- all under the one allowance of tests/tolerance.py.

    python tests/tools/parity_table.py gpurun_out/parity_stats.json profiles/r05_parity_stats.json
"""

import json
import sys


def main(src, dst):
    recs = json.load(open(src))
    table = {}
    for r in recs:
        build = "profile-guided" if "(profile-guided build)" in r["what"] else "default"
        key = f"{r['model']} / {build}"
        row = table.setdefault(key, {})
        against = r.get("against") or "gcc"
        cell = row.setdefault(against, {"comparisons": 0, "values": 0, "worst_ratio": 0.0, "worst_at": None})
        cell["comparisons"] += 1
        cell["values"] += r["values"]
        if r.get("worst_ratio") is not None and r["worst_ratio"] > cell["worst_ratio"]:
            cell["worst_ratio"], cell["worst_at"] = r["worst_ratio"], r["what"]
        if against == "gcc-vs-clang":
            for k in ("nan_mismatch", "nan_mismatch_at_firm_points", "gpu_nan_like_gcc", "gpu_nan_like_clang", "above_1e-10"):
                cell[k] = cell.get(k, 0) + int(r.get(k, 0) or 0)
            cell["max_rel"] = max(cell.get("max_rel", 0.0), r.get("max_rel", 0.0) or 0.0)
        else:
            if r.get("excluded", 0.0) >= cell.get("max_excluded", 0.0):
                cell["max_excluded"], cell["max_excluded_at"], cell["max_excluded_cap"] = r.get("excluded", 0.0), r["what"], r.get("excluded_cap")
            cell["nan_mismatch_at_flaky_points"] = cell.get("nan_mismatch_at_flaky_points", 0) + int(r.get("nan_mismatch_at_flaky_points", 0) or 0)
            # the GPU against this build in plain relative terms (round 6): values above the literal 1e-10, the largest relative difference
            cell["above_1e-10"] = cell.get("above_1e-10", 0) + int(r.get("above_1e-10", 0) or 0)
            cell["max_rel"] = max(cell.get("max_rel", 0.0), r.get("max_rel", 0.0) or 0.0)
            if "/off/" in r["what"] or r["what"].endswith("/off"):
                cell["excluded_on_the_off_grid"] = max(cell.get("excluded_on_the_off_grid", 0.0), r.get("excluded", 0.0))
    out = {
        "what": "per model and build: GPU vs the gcc-built reference, GPU vs the clang-built reference (worst |gpu - ref| / allowance over every "
        "comparison of the -m gpu suite; <= 1 passes; NaN and Inf patterns are asserted exactly at every point whose NaN-ness the reference "
        "itself settles), and the reference's two builds against each other under the same allowance (nothing asserted: a ratio above 1 means "
        "the reference differs from itself by more than the GPU may differ from it).  above_1e-10 / max_rel: compared finite values that differ by "
        "more than the literal 1e-10 relative, and the largest relative difference -- for the GPU against each build and for the two builds "
        "against each other.  KAPPA and the exclusion caps in force: tests/tolerance.py",
        "table": table,
        "records": len(recs),
    }
    json.dump(out, open(dst, "w"), indent=1)
    for key, row in sorted(table.items()):
        print(key, {a: (round(c["worst_ratio"], 4), c["comparisons"]) for a, c in row.items()})


if __name__ == "__main__":
    main(*sys.argv[1:3])
