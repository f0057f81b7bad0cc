import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import workloads
from inflatox_amd.consistency_conditions import GeneralisedAL
spec, art = workloads.artifact_for("hyperbolic")
al = GeneralisedAL(art)
ss = np.array([[spec.extent[0], spec.extent[1]], [spec.extent[2], spec.extent[3]]])
def t(fn, reps=20):
    fn(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e6
for n in (256, 1000, 2000, 4096):
    out = np.zeros((n, n, 6))
    a = t(lambda: al.complete_analysis(spec.args, *spec.extent, n, n, progress=False))
    b = t(lambda: al.dylib.complete_analysis(spec.args, out, ss, False, 0))
    c = t(lambda: np.zeros((n, n, 6)).fill(1.0))
    print(f"n={n}: front-end {a:9.1f} us   reused out {b:9.1f} us   np.zeros+fill {c:9.1f} us   bytes {48*n*n/1e6:.1f} MB", flush=True)
