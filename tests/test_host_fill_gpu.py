"""Host-side broadcast fill (csrc/inflx_hip.cpp sweep_host_broadcast): a result that is constant along one grid axis is
written by host threads from the one evaluated line instead of crossing PCIe.  The bytes must be those of the device-resident
sweep -- every operation, both layouts, several parameter rows, row ranges, odd and tiny row lengths, destinations that are
not 16-byte aligned, the multi-device split -- and those of the copy path (INFLX_HOST_FILL=0)."""

import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _device_result(lib, gpu_lib, op, rows, ext, n0, n1, layout, rb, rc):
    import torch

    k = gpu_lib.OP_WIDTH[op]
    out = torch.full((len(rows) * rc * n1 * k,), -3.0, dtype=torch.float64, device="cuda:0")
    lib.sweep_device(op, rows, out.data_ptr(), out.numel() * 8, ext, n0, n1, row_begin=rb, row_count=rc, layout=layout, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("which", ["row_only", "column_only"])
def test_filled_result_equals_the_device_resident_one(which, gpu_lib):
    import workloads

    if which == "row_only":
        spec, art = workloads.artifact_for("hyperbolic")
        args, ext = np.asarray(spec.args, dtype=np.float64), spec.extent
    else:
        from test_models_extra import setup

        from inflatox_amd import Compiler

        model, args, ext, *_ = setup("column_only")
        art = Compiler(model, silent=True).compile()
    lib = gpu_lib.InflatoxDevLib(art.shared_object_path)
    assert lib.stage_info["out_mask"] == (1 if which == "row_only" else 2)
    rng = np.random.default_rng(5)
    ops = [gpu_lib.OP_COMPLETE, gpu_lib.OP_CONSISTENCY, gpu_lib.OP_RAW, gpu_lib.OP_EPSILON_V]
    # (P, n0, n1): at least 8 MiB so that the fill path is taken; odd row lengths, a single column / row, many short rows
    shapes = [(1, 700, 300), (2, 333, 257), (3, 64, 4097), (1, 20000, 9), (1, 5, 40001), (2, 1500, 130)]
    for case, (P, n0, n1) in enumerate(shapes):
        for op in ops:
            k = gpu_lib.OP_WIDTH[op]
            for layout in (gpu_lib.LAYOUT_AOS, gpu_lib.LAYOUT_SOA):
                rb = int(rng.integers(0, n0 // 3 + 1))
                rc = n0 - rb - int(rng.integers(0, n0 // 4 + 1))
                if P * rc * n1 * k * 8 < (8 << 20):
                    rb, rc = 0, n0
                if P * rc * n1 * k * 8 < (8 << 20):
                    continue
                rows = np.stack([args * (1.0 + 0.07 * q) for q in range(P)])
                got = lib.sweep_host(op, rows, ext, n0, n1, row_begin=rb, row_count=rc, layout=layout)
                want = _device_result(lib, gpu_lib, op, rows, ext, n0, n1, layout, rb, rc)
                assert np.array_equal(got.reshape(-1), want, equal_nan=True), (which, P, n0, n1, op, layout, rb, rc)
    # a destination that is only 8-byte aligned: the front-end's result arrays are page-aligned, a C caller's need not be
    P, n0, n1 = 1, 700, 301
    rows = args.reshape(1, -1)
    buf = np.zeros(n0 * n1 * 6 + 1)
    dst = buf[1:].reshape(1, n0, n1, 6)
    assert dst.ctypes.data % 16 == 8
    import ctypes as C

    rc_ = lib._lib.inflx_sweep_host(lib._h, gpu_lib.OP_COMPLETE, rows.ctypes.data_as(C.POINTER(C.c_double)), 1, rows.shape[1], dst.ctypes.data_as(C.POINTER(C.c_double)),
                                    np.asarray(ext, dtype=np.float64).ctypes.data_as(C.POINTER(C.c_double)), n0, n1, 0, n0, gpu_lib.LAYOUT_AOS)
    assert rc_ == 0 and buf[0] == 0.0
    assert np.array_equal(dst.reshape(-1), _device_result(lib, gpu_lib, gpu_lib.OP_COMPLETE, rows, ext, n0, n1, gpu_lib.LAYOUT_AOS, 0, n0), equal_nan=True)
    # the multi-device split fills every device's slab
    multi = gpu_lib.InflatoxMultiLib(art.shared_object_path, [0, 0, 0])
    for P, n0, n1 in ((1, 1001, 300), (4, 300, 257)):
        rows = np.stack([args * (1.0 + 0.07 * q) for q in range(P)])
        for layout in (gpu_lib.LAYOUT_AOS, gpu_lib.LAYOUT_SOA):
            got = multi.sweep_host(gpu_lib.OP_COMPLETE, rows, ext, n0, n1, layout=layout)
            want = _device_result(lib, gpu_lib, gpu_lib.OP_COMPLETE, rows, ext, n0, n1, layout, 0, n0)
            assert np.array_equal(got.reshape(-1), want, equal_nan=True), (which, "multi", P, n0, n1, layout)


_COPY_PATH = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
import workloads
from inflatox_amd import _native
spec, art = workloads.artifact_for("hyperbolic")
lib = _native.InflatoxDevLib(art.shared_object_path)
out = lib.sweep_host(_native.OP_COMPLETE, spec.args, spec.extent, 900, 410)
np.save(sys.argv[1], out)
"""


def test_fill_path_equals_copy_path(gpu_lib, tmp_path):
    """The same call with the fill switched off (the round-3 device-to-host copy of the whole array) gives the same bytes."""
    import workloads

    spec, art = workloads.artifact_for("hyperbolic")
    lib = gpu_lib.InflatoxDevLib(art.shared_object_path)
    filled = lib.sweep_host(gpu_lib.OP_COMPLETE, spec.args, spec.extent, 900, 410)
    env = dict(os.environ, INFLX_HOST_FILL="0")
    proc = subprocess.run([sys.executable, "-c", _COPY_PATH.format(root=ROOT), str(tmp_path / "copied.npy")], env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    assert np.array_equal(np.load(tmp_path / "copied.npy"), filled, equal_nan=True)
