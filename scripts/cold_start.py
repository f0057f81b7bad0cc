#!/usr/bin/env python3
"""(CPU, no GPU needed) Cold start of every example model from an EMPTY code-object cache: seconds of hipcc behind
``Compiler.compile()`` (the core object: what stands between a user and the first ``complete_analysis``; the reference's
counterpart is one ``zig cc`` step, python/inflatox/compiler.py:568-598) and behind the first use of every other operation
(its kernel group), next to the complete artefact in one step (``kernel_groups="all"``: what every compile() cost before round 6).
usage: cold_start.py [MODEL ...] > profiles/r06_cold_start.json"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["INFLATOX_AMD_CACHE"] = tempfile.mkdtemp(prefix="inflx_cold_cache_")

import workloads  # noqa: E402
from workloads import example_models  # noqa: E402
from inflatox_amd import compiler as C  # noqa: E402

# every hipcc invocation of the package goes through subprocess.run in inflatox_amd.compiler: time those
spent = []
_run = subprocess.run


def timed_run(cmd, *a, **k):
    t0 = time.perf_counter()
    out = _run(cmd, *a, **k)
    if cmd and "hipcc" in os.path.basename(str(cmd[0])):
        spent.append(time.perf_counter() - t0)
    return out


C.subprocess.run = timed_run

report = {"cpus": os.cpu_count(), "what": "seconds of hipcc per step, empty cache; `compile` = Compiler.compile() = the core object (incl. the three-waves probe's second build where it falls back to two)"}
for name in sys.argv[1:] or ["hyperbolic", "doc", "angular", "egno", "d5"]:
    spec = example_models.get(name)
    t0 = time.perf_counter()
    model = workloads.model_for(name)
    t_symbolic = time.perf_counter() - t0
    spent.clear()
    t0 = time.perf_counter()
    art = C.Compiler(model, silent=True, **spec.compiler_kwargs).compile()
    rec = {"symbolic_stage_s": round(t_symbolic, 2), "compile_wall_s": round(time.perf_counter() - t0, 2), "compile_hipcc_s": round(sum(spent), 2), "compile_hipcc_steps": len(spent),
           "core_object_bytes": os.path.getsize(art.shared_object_path), "groups_hipcc_s": {}}
    for group in C.KERNEL_GROUPS:
        if group == "core":
            continue
        spent.clear()
        art.ensure_group(group)
        rec["groups_hipcc_s"][group] = round(sum(spent), 2)
    rec["all_groups_on_first_use_hipcc_s"] = round(rec["compile_hipcc_s"] + sum(rec["groups_hipcc_s"].values()), 2)
    spent.clear()
    full = C.Compiler(model, silent=True, kernel_groups="all", **spec.compiler_kwargs).compile()
    rec["complete_artefact_one_step_hipcc_s"] = round(sum(spent), 2)
    rec["complete_artefact_bytes"] = os.path.getsize(full.shared_object_path)
    report[name] = rec
    print(name, rec, file=sys.stderr, flush=True)
print(json.dumps(report, indent=1))
