"""Build and call tests/host_twin.cpp (the CPU twin of the sweep kernels) -- test infrastructure."""

import ctypes as C
import hashlib
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
_DP = C.POINTER(C.c_double)


class HostTwin:
    def __init__(self, header_text: str, contract: str = "off", cxx: str = "g++"):
        """``cxx``: the host compiler; "clang++" (the ROCm image's) understands ``contract="on"`` -- fusion within a statement, the rule
        hipcc applies to the kernels and `zig cc` to the reference's C -- which g++ reads as "off"."""
        if cxx == "clang++":
            import shutil

            cxx = shutil.which("clang++") or "/opt/rocm/lib/llvm/bin/clang++"
        # INFLX_TEST_SANITIZE=1 (manual runs under LD_PRELOAD=libasan.so): the generated stage code and csrc/inflx_ops.h under ASan + UBSan
        sanitize = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-g"] if os.environ.get("INFLX_TEST_SANITIZE") else []
        tag = hashlib.sha1((header_text + contract + cxx + " ".join(sanitize)).encode()).hexdigest()[:16]
        d = os.path.join(tempfile.gettempdir(), "inflx_host_twin")
        os.makedirs(d, exist_ok=True)
        hdr = os.path.join(d, f"{tag}.h")
        so = os.path.join(d, f"{tag}.so")
        if not os.path.exists(so):
            with open(hdr, "w") as fh:
                fh.write(header_text)
            cmd = [
                cxx, "-O2", "-std=c++17", "-fPIC", "-shared", f"-ffp-contract={contract}", *(["-mfma"] if contract != "off" else []),  # (a fused multiply-add needs the instruction) "-fno-fast-math", "-Wno-unknown-pragmas", *sanitize,
                f"-I{os.path.join(ROOT, 'inflatox_amd', 'csrc')}", f'-DINFLX_MODEL_HEADER="{hdr}"',
                os.path.join(HERE, "host_twin.cpp"), "-o", so + ".tmp",
            ]  # fmt: skip
            subprocess.run(cmd, check=True)
            os.replace(so + ".tmp", so)
        self.lib = C.CDLL(so)
        self.lib.twin_grid.argtypes = [C.c_int, _DP, _DP, C.c_size_t, C.c_size_t, _DP]
        self.lib.twin_trajectory.argtypes = [C.c_int, _DP, _DP, C.c_size_t, _DP]
        self.lib.twin_set_accuracy.argtypes = [C.c_double]
        self.lib.twin_basis.argtypes = [_DP, _DP, C.c_size_t, _DP]
        self.n_parameters = self.lib.twin_n_parameters()
        self.out_mask = self.lib.twin_out_mask()
        self.v01_is_v10 = bool(self.lib.twin_v01_is_v10())

    @staticmethod
    def _w(op):
        return {0: 6, 4: 5, 6: 4}.get(op, 1)

    def set_accuracy(self, accuracy: float):
        self.lib.twin_set_accuracy(accuracy)

    def grid(self, op, p, extent, n0, n1):
        p = np.ascontiguousarray(p, dtype=np.float64)
        ss = np.ascontiguousarray(extent, dtype=np.float64)
        k = self._w(op)
        out = np.zeros((n0, n1, k))
        self.lib.twin_grid(op, p.ctypes.data_as(_DP), ss.ctypes.data_as(_DP), n0, n1, out.ctypes.data_as(_DP))
        return out if k > 1 else out[..., 0]

    def trajectory(self, op, p, pts):
        p = np.ascontiguousarray(p, dtype=np.float64)
        pts = np.ascontiguousarray(pts, dtype=np.float64)
        k = self._w(op)
        out = np.zeros((pts.shape[0], k))
        self.lib.twin_trajectory(op, p.ctypes.data_as(_DP), pts.ctypes.data_as(_DP), pts.shape[0], out.ctypes.data_as(_DP))
        return out if k > 1 else out[..., 0]

    def basis(self, p, pts):
        p = np.ascontiguousarray(p, dtype=np.float64)
        pts = np.ascontiguousarray(pts, dtype=np.float64)
        out = np.zeros((pts.shape[0], 7))
        self.lib.twin_basis(p.ctypes.data_as(_DP), pts.ctypes.data_as(_DP), pts.shape[0], out.ctypes.data_as(_DP))
        return out
