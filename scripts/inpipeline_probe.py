import sys, numpy as np, torch
sys.path.insert(0,'/root/repo')
from inflatox_amd import _native
import workloads
spec,art=workloads.artifact_for("hyperbolic")
lib=_native.InflatoxDevLib(art.shared_object_path)
n=8192
out=torch.empty((n,n,6),dtype=torch.float64,device="cuda:0")
st=torch.cuda.Stream(); s=st.cuda_stream
for _ in range(80): lib.sweep_device(_native.OP_COMPLETE, spec.args, out.data_ptr(), out.numel()*8, spec.extent, n, n, stream=s)
torch.cuda.synchronize()
for rnd in range(3):
    iso=lib.sweep_device_timed(_native.OP_COMPLETE, spec.args, out.data_ptr(), out.numel()*8, spec.extent, n, n, stream=s, repeats=20, dominant_only=True)
    pipe=lib.sweep_device_timed(_native.OP_COMPLETE, spec.args, out.data_ptr(), out.numel()*8, spec.extent, n, n, stream=s, repeats=20, in_pipeline=True)
    whole=lib.sweep_device_timed(_native.OP_COMPLETE, spec.args, out.data_ptr(), out.numel()*8, spec.extent, n, n, stream=s, repeats=20)
    print(f"isolated {iso:.4f}  in_pipeline {pipe:.4f}  whole sweep {whole:.4f} ms", flush=True)
spec,art=workloads.artifact_for("doc"); lib=_native.InflatoxDevLib(art.shared_object_path); n=4096
out=torch.empty((n,n,6),dtype=torch.float64,device="cuda:0")
for rnd in range(2):
    pipe=lib.sweep_device_timed(_native.OP_COMPLETE, spec.args, out.data_ptr(), out.numel()*8, spec.extent, n, n, stream=s, repeats=20, in_pipeline=True)
    whole=lib.sweep_device_timed(_native.OP_COMPLETE, spec.args, out.data_ptr(), out.numel()*8, spec.extent, n, n, stream=s, repeats=20)
    print(f"doc tile: in_pipeline {pipe:.4f}  whole sweep {whole:.4f} ms", flush=True)
