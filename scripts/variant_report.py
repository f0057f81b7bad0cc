#!/usr/bin/env python3
"""VALU instructions per grid point, VALU-busy share and duration of the tile-kernel launches of scripts/variant_probe.py,
from a rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES database.
usage: variant_report.py DB N CASE [CASE ...]   (cases in the order they were given to variant_probe.py)"""
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from valu_report import dispatches  # noqa: E402

db, n, cases = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
rows = [r for r in dispatches(db) if r["kernel"].startswith("inflx_sweep_tile")]
assert len(rows) == 2 * len(cases), (len(rows), len(cases))
for k, case in enumerate(cases):
    r = rows[2 * k + 1]
    print(f"{case:64s} {r['SQ_INSTS_VALU'] * 64 / (n * n):7.1f} VALU/pt  {r['us']:7.1f} us  valu-active/wave-cycles {r['SQ_ACTIVE_INST_VALU'] / r['SQ_WAVE_CYCLES']:.3f}")
