import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import workloads
from inflatox_amd import _native
spec, art = workloads.artifact_for("doc")
lib = _native.InflatoxDevLib(art.shared_object_path)
rng = np.random.default_rng(7)
x0a, x0b, x1a, x1b = spec.extent
for npts in (500, 1_000_000):
    pts = np.column_stack([rng.uniform(x0a, x0b, npts), rng.uniform(x1a, x1b, npts)])
    r=lib.sweep_on_trajectory(_native.OP_COMPLETE, spec.args, pts)
    best = 1e9
    for _ in range(7):
        t0 = time.perf_counter(); r=lib.sweep_on_trajectory(_native.OP_COMPLETE, spec.args, pts); best = min(best, time.perf_counter() - t0)
    print(npts, "points:", round(best*1e3,3), "ms", float(np.nansum(r[::97])))
