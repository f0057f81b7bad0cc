import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import workloads
from inflatox_amd.consistency_conditions import GeneralisedAL, InflationCondition
for name in ("doc", "d5", "hyperbolic"):
    spec, art = workloads.artifact_for(name)
    al = GeneralisedAL.__new__(GeneralisedAL); InflationCondition.__init__(al, art, validate_basis=False)
    al.flag_quantum_dif(spec.args, *spec.extent, 256, 256, progress=False)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); x = al.flag_quantum_dif(spec.args, *spec.extent, 10000, 10000, progress=False); best = min(best, time.perf_counter() - t0)
    print(name, "flag_quantum_dif 10000^2: %.1f ms end-to-end, %.2e points/s, flagged %.4f" % (best*1e3, 1e8/best, x.mean()))
