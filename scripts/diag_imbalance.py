#!/usr/bin/env python3
"""Imbalance between the four wavefronts of a tile workgroup, from a diagnostic build (-DINFLX_DIAG_WAVE_LIFETIME: every wavefront writes
its lifetime, in s_memtime ticks, over the first value of its first point in the tile's last row).  A workgroup's slots are held until
its slowest wavefront retires: the share of slot time lost that way is mean over workgroups of 1 - mean(lifetimes) / max(lifetimes).
usage: diag_imbalance.py MODEL [N] [P]

The diagnostic lines are NOT part of csrc/inflx_sweep_kernels.hip (they would change every code object's content tag); apply this patch
to a working copy first (`git apply`), run, and restore the file:

diff --git a/inflatox_amd/csrc/inflx_sweep_kernels.hip b/inflatox_amd/csrc/inflx_sweep_kernels.hip
index b4ba764..6116281 100644
--- a/inflatox_amd/csrc/inflx_sweep_kernels.hip
+++ b/inflatox_amd/csrc/inflx_sweep_kernels.hip
@@ -306,6 +306,9 @@ __device__ __forceinline__ void sweep_tile(const InflxSweepArgs& a) {
   const unsigned lane = tid & (kWave - 1);
   const unsigned wave = tid / kWave;
   const unsigned p = blockIdx.z;
+#ifdef INFLX_DIAG_WAVE_LIFETIME  // (diagnostic builds only, scripts/diag_imbalance.py: every wavefront leaves its lifetime in the result)
+  const uint64_t diag_t0 = __builtin_amdgcn_s_memtime();
+#endif
 
   double A[kNP];
   load_params(a.params, p, A);
@@ -495,6 +498,15 @@ __device__ __forceinline__ void sweep_tile(const InflxSweepArgs& a) {
     apply_op<OP, kTable>(mv, o, a.accuracy, kc);
     emit(o, row);
   }
+#ifdef INFLX_DIAG_WAVE_LIFETIME
+  // the wavefront's lifetime in counter ticks replaces the first value of its first point in the tile's LAST row (AoS, K = 6)
+  if constexpr (K == 6 && STORE) {
+    __builtin_amdgcn_s_waitcnt(0x0F70);  // the row's own stores have been acknowledged
+    const uint64_t diag_t1 = __builtin_amdgcn_s_memtime();
+    if (lane == 0 && wave_col0 < a.N1 && nrows > 0)
+      a.out[(((uint64_t)p * a.row_count + row0 + (uint64_t)(nrows - 1)) * a.N1 + wave_col0) * 6] = (double)(diag_t1 - diag_t0);
+  }
+#endif
   if constexpr (STATS) stat_flush(acc, a.stats);
 }
 
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import workloads  # noqa: E402
from inflatox_amd import _native  # noqa: E402
from inflatox_amd.compiler import Compiler  # noqa: E402
from workloads import example_models  # noqa: E402

name = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
P = int(sys.argv[3]) if len(sys.argv) > 3 else 8
spec = example_models.get(name)
art = Compiler(workloads.model_for(name), silent=True, compiler_flags=list(Compiler.default_hipcc_flags) + ["-DINFLX_DIAG_WAVE_LIFETIME=1"], **spec.compiler_kwargs).compile()
if os.environ.get("INFLX_EXPERIMENT_COMPILE_ONLY") == "1":
    print("compiled", name)
    sys.exit(0)
lib = _native.InflatoxDevLib(art.shared_object_path)
rows = np.tile(np.asarray(spec.args, dtype=np.float64), (P, 1))
if name == "d5" and P > 1:
    rows[:, 6] = np.linspace(2.5e-4, 1e-3, P)
buf = torch.empty((P, n, n, 6), dtype=torch.float64, device="cuda:0")
stream = torch.cuda.current_stream().cuda_stream
plan = lib.sweep_plan(_native.OP_COMPLETE, P, n, n)
th = plan["tile_rows"]
for _ in range(3):
    lib.sweep_device(_native.OP_COMPLETE, rows, buf.data_ptr(), buf.numel() * 8, spec.extent, n, n, stream=stream)
torch.cuda.synchronize()
last_rows = [min(r + th, n) - 1 for r in range(0, n, th)]
life = buf[:, last_rows][:, :, ::64, 0].cpu().numpy()  # (P, row tiles, wavefronts along the row)
w = life.shape[2] // 4 * 4
groups = life[:, :, :w].reshape(P, len(last_rows), w // 4, 4)  # the four wavefronts of a workgroup
mx, mean = groups.max(axis=-1), groups.mean(axis=-1)
print(f"{name} {n}^2 x {P}, tiles of {th} rows: wavefront lifetime mean {life.mean():.0f} ticks, sd {life.std():.0f} ({100 * life.std() / life.mean():.1f} %), min {life.min():.0f}, max {life.max():.0f}")
print(f"   slot time lost inside workgroups (1 - mean / max of the four lifetimes): {100 * (1 - mean / mx).mean():.2f} %")
first = groups[:, :, 0, :]  # the workgroups that own grid column 0
print(f"   workgroups owning column 0: lost {100 * (1 - first.mean(axis=-1) / first.max(axis=-1)).mean():.2f} %; their first wavefront {first[..., 0].mean():.0f} ticks against {first[..., 1:].mean():.0f}")
by_wave = groups.mean(axis=(0, 1, 2))
print("   mean lifetime by position in the workgroup:", " ".join(f"{v:.0f}" for v in by_wave))
per_col = life.mean(axis=(0, 1))
print(f"   by column block (64 columns): min {per_col.min():.0f}, max {per_col.max():.0f}; sd across blocks {100 * per_col.std() / per_col.mean():.1f} % (systematic differences between columns)")
per_row = life.mean(axis=(0, 2))
print(f"   by row tile: sd across tiles {100 * per_row.std() / per_row.mean():.1f} %")
