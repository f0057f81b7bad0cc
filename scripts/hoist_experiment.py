#!/usr/bin/env python3
"""Experiment (GPU box): tile-kernel time of the hoisted-reciprocal division against the default.
usage: hoist_experiment.py MODEL[:flag,flag...] ...   flags: hoist, nohoist, inline (self-checking hoisted quotients, one point stage), regroup, share, nosqrt, tanN (tan_shortcut=N), trust (-DINFLX_DIVH_TRUST: accept every quotient),
wN (N waves/SIMD: -DINFLX_MIN_WAVES=N), inner (grid away from the first row/column), DNAME=value (any -D switch of the kernel sources, e.g.
DINFLX_HORNER_MODE=0, DINFLX_EXPERIMENT_IEEE_EPILOGUE=1, DINFLX_TAN_SHORTCUT_MAX=16)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from inflatox_amd import _native  # noqa: E402
import workloads  # noqa: E402
from workloads import example_models  # noqa: E402
from inflatox_amd.compiler import Compiler  # noqa: E402

n = int(os.environ.get("INFLX_EXPERIMENT_N", "4096"))
# INFLX_EXPERIMENT_COMPILE_ONLY=1 (CPU container): build every variant into the in-tree cache, which travels to the GPU box
compile_only = os.environ.get("INFLX_EXPERIMENT_COMPILE_ONLY") == "1"
if not compile_only:
    import torch

    stream = torch.cuda.current_stream().cuda_stream
    P = int(os.environ.get("INFLX_EXPERIMENT_P", "1"))  # parameter rows per call (BASELINE configs[2]: 32 for D5)
    out = torch.empty((P, n, n, 6), dtype=torch.float64, device="cuda:0")
cases = sys.argv[1:] or ["d5", "d5:hoist", "egno", "egno:hoist"]
rounds = int(os.environ.get("INFLX_EXPERIMENT_ROUNDS", "1"))  # interleaved repetitions (A B A B ...): box-to-box and drift noise is +-3 %
best = {}
for case in cases * rounds:
    name, _, fl = case.partition(":")
    fl = set(fl.split(",")) - {""}
    spec = example_models.get(name)
    kw = dict(spec.compiler_kwargs)
    flags = list(Compiler.default_hipcc_flags)
    for f in fl:
        if len(f) == 2 and f[0] == "w" and f[1].isdigit():
            flags.append(f"-DINFLX_MIN_WAVES={f[1]}")
    if "trust" in fl:
        flags.append("-DINFLX_DIVH_TRUST=1")
    for f in fl:
        if f.startswith("D"):
            flags.append("-" + f)
    ext = spec.extent
    if "inner" in fl:  # keep away from the first row and the first column
        x0a, x0b, x1a, x1b = ext
        ext = (x0a + 0.1 * (x0b - x0a), x0b, x1a + 0.1 * (x1b - x1a), x1b)
    hoist = True if "hoist" in fl else (False if "nohoist" in fl else ("inline" if "inline" in fl else None))  # default: the compiler's automatic choice
    if "regroup" in fl:
        kw["regroup"] = True
    if "share" in fl:
        kw["share_reciprocals"] = True
    for f in fl:
        if f.startswith("tan") and f[3:].isdigit():  # tanN: Compiler(tan_shortcut=N)
            kw["tan_shortcut"] = int(f[3:])
    if "nosqrt" in fl:  # the point stage's square roots as the compiler spells them
        kw["quick_sqrt"] = False
    art = Compiler(workloads.model_for(name), silent=True, compiler_flags=flags, hoist_reciprocals=hoist, **kw).compile()
    if compile_only:
        print("compiled", case, flush=True)
        continue
    lib = _native.InflatoxDevLib(art.shared_object_path)
    rows = np.tile(np.asarray(spec.args, dtype=np.float64), (P, 1))
    if name == "d5" and P > 1:
        rows[:, 6] = np.linspace(2.5e-4, 1e-3, P)  # a1, as in bench.py
    reps = max(2, 30 // P)
    ms = min(lib.sweep_device_timed(_native.OP_COMPLETE, rows, out.data_ptr(), out.numel() * 8, ext, n, n, stream=stream, repeats=reps) for _ in range(3))
    best[case] = min(ms, best.get(case, 1e9))
    print(f"{case:36s}: {ms:7.3f} ms  {P * n * n / ms / 1e6:7.2f} Gpts/s", flush=True)
for case, ms in best.items():
    print(f"BEST {case:36s}: {ms:7.3f} ms  {P * n * n / ms / 1e6:7.2f} Gpts/s", flush=True)
