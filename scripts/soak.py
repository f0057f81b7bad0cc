#!/usr/bin/env python3
"""Soak of the launch machinery (GPU box): random grid shapes, row ranges, parameter batches, layouts and operations for a
bounded time, every sweep compared bit for bit with the on-trajectory kernel at the same points (same stage code, no tables,
no tiles: any difference is an indexing, table-reuse, stream-ordering or parameter-slot error).  Each case runs the host-result
sweep (one stream when it is a single launch) and a BURST of back-to-back device-result sweeps with different parameters on one
stream, checked only after the whole burst has been enqueued -- the double-buffered tables and the parameter ring are reused while
earlier sweeps are still in flight.  usage: soak.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import workloads  # noqa: E402
from inflatox_amd import _native  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
OPS = [(_native.OP_COMPLETE, 6), (_native.OP_CONSISTENCY, 1), (_native.OP_RAW, 5), (_native.OP_EPSILON_V, 1), (_native.OP_RAPIDTURN, 1), (_native.OP_HESSE, 4)]
libs = {}
for name in ("hyperbolic", "doc", "d5", "egno"):
    spec, art = workloads.artifact_for(name)
    libs[name] = (spec, art, _native.InflatoxDevLib(art.shared_object_path))
stream = torch.cuda.Stream()
t_end = time.time() + budget
cases = sweeps = 0
while time.time() < t_end:
    name = ("hyperbolic", "doc", "d5", "egno")[int(rng.integers(0, 4))]
    spec, art, lib = libs[name]
    kind = int(rng.integers(0, 4))
    if kind == 0:  # small
        n0, n1 = int(rng.integers(1, 70)), int(rng.integers(1, 300))
    elif kind == 1:  # around the tile / launch thresholds
        n0, n1 = int(rng.choice([31, 32, 33, 63, 64, 65, 255, 256, 257, 511, 512])), int(rng.choice([1, 2, 255, 256, 257, 511, 512, 513, 1024]))
    elif kind == 2:  # tall
        n0, n1 = int(rng.integers(300, 5000)), int(rng.integers(1, 40))
    else:  # wide
        n0, n1 = int(rng.integers(1, 40)), int(rng.integers(300, 6000))
    P = int(rng.integers(1, 5))
    op, k = OPS[int(rng.integers(0, len(OPS)))]
    layout = _native.LAYOUT_SOA if rng.integers(0, 2) else _native.LAYOUT_AOS
    rb = int(rng.integers(0, n0))
    rc = int(rng.integers(1, n0 - rb + 1))
    x0a, x0b, x1a, x1b = spec.extent
    ss = np.array([[x0a, x0b], [x1a, x1b]])
    dx0, dx1 = (x0b - x0a) / n0, (x1b - x1a) / n1
    xs0 = np.arange(rb, rb + rc, dtype=np.float64) * dx0 + x0a
    xs1 = np.arange(n1, dtype=np.float64) * dx1 + x1a
    pts = np.stack(np.meshgrid(xs0, xs1, indexing="ij"), axis=-1).reshape(-1, 2)
    burst = int(rng.integers(1, 4))
    arg_sets = [np.stack([spec.args * (1.0 + 0.01 * (q + 7 * b)) for q in range(P)]) for b in range(burst)]

    def expected(args):
        want = np.stack([lib.sweep_on_trajectory(op, args[q], pts).reshape(rc, n1, k) for q in range(P)])
        return np.moveaxis(want, -1, 1) if layout == _native.LAYOUT_SOA else want

    wants = [expected(a) for a in arg_sets]
    what = (name, n0, n1, P, op, layout, rb, rc, burst)
    got = lib.sweep_host(op, arg_sets[0], ss, n0, n1, row_begin=rb, row_count=rc, layout=layout)
    if not np.array_equal(got.reshape(wants[0].shape), wants[0], equal_nan=True):
        print("HOST MISMATCH", what, flush=True)
        sys.exit(1)
    outs = [torch.full((wants[0].size,), -7.0, dtype=torch.float64, device="cuda:0") for _ in range(burst)]
    torch.cuda.synchronize()
    for b in range(burst):  # enqueued back to back, nothing waited for in between
        lib.sweep_device(op, arg_sets[b], outs[b].data_ptr(), outs[b].numel() * 8, ss, n0, n1, row_begin=rb, row_count=rc, layout=layout, stream=stream.cuda_stream)
    stream.synchronize()
    for b in range(burst):
        if not np.array_equal(outs[b].cpu().numpy().reshape(wants[b].shape), wants[b], equal_nan=True):
            print("DEVICE MISMATCH", what, "burst member", b, flush=True)
            sys.exit(1)
    cases += 1
    sweeps += 1 + burst
    if cases % 50 == 0:
        print(f"{cases} cases, {sweeps} sweeps, all equal", flush=True)
print(f"soak finished: {cases} cases, {sweeps} sweeps, all bit-equal to point evaluation (seed {seed})")
