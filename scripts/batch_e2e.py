import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from conftest import generalised_al
import workloads
spec, art = workloads.artifact_for("d5")
al = generalised_al(art)
P, n = 8, 4096
args = np.tile(spec.args, (P, 1)) * (1 + 0.01 * np.arange(P))[:, None]
al.complete_analysis_batch(args[:1], *spec.extent, 256, 256)
for mode in ("whole", "chunks"):
    os.environ["X"] = mode
    t0 = time.perf_counter()
    res = al.complete_analysis_batch(args, *spec.extent, n, n)
    dt = time.perf_counter() - t0
    print(f"d5 {n}x{n} x P={P}: {res.nbytes / 1e9:.2f} GB in {dt * 1e3:.1f} ms = {res.nbytes / dt / 1e9:.1f} GB/s, checksum {np.nansum(res[:, ::97, ::89, 1]):.6e}", flush=True)
    del res
