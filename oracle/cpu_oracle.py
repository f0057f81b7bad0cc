"""ctypes bindings for ``sweep_oracle.c`` -- TEST INFRASTRUCTURE ONLY (see package docstring)."""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_sweep.so")
_lib = None


class OP:
    """Per-point operations, same numbering as ``sweep_oracle.c``."""

    COMPLETE = 0
    CONSISTENCY = 1
    RAPIDTURN = 2
    EPSILON_V = 3
    QDIF = 4
    RAW = 5


_WIDTH = {OP.COMPLETE: 6, OP.RAW: 5}


def build_sweep_library(force: bool = False) -> str:
    """Compile ``sweep_oracle.c`` (gcc, -ffp-contract=off) if it is stale; return the .so path."""
    src = os.path.join(_HERE, "sweep_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-s", "-B", "-C", _HERE, "liboracle_sweep.so"], check=True)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build_sweep_library()
        lib = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        lib.oracle_last_error.restype = C.c_char_p
        lib.oracle_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        lib.oracle_close.argtypes = [C.c_void_p]
        lib.oracle_dim.argtypes = [C.c_void_p]
        lib.oracle_dim.restype = C.c_uint32
        lib.oracle_n_par.argtypes = [C.c_void_p]
        lib.oracle_n_par.restype = C.c_uint32
        lib.oracle_name.argtypes = [C.c_void_p]
        lib.oracle_name.restype = C.c_char_p
        lib.oracle_grid_sweep.argtypes = [C.c_void_p, C.c_int, dp, C.c_size_t, C.c_void_p, dp, C.c_size_t, C.c_size_t, C.c_double, C.c_int]
        lib.oracle_trajectory_sweep.argtypes = [C.c_void_p, C.c_int, dp, C.c_size_t, dp, C.c_size_t, C.c_void_p, C.c_double, C.c_int]
        lib.oracle_potential.argtypes = [C.c_void_p, dp, dp]
        lib.oracle_potential.restype = C.c_double
        lib.oracle_hesse.argtypes = [C.c_void_p, dp, dp, dp]
        lib.oracle_raw_long_double.argtypes = [C.c_char_p, dp, C.c_size_t, dp, C.c_size_t, dp]
        _lib = lib
    return _lib


def _dptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class OracleError(Exception):
    pass


class OracleModel:
    """A per-model shared object (reference C ABI) opened by the oracle."""

    def __init__(self, so_path: str):
        lib = _load()
        h = C.c_void_p()
        rc = lib.oracle_open(so_path.encode(), C.byref(h))
        if rc != 0:
            raise OracleError(f"[{rc}] {lib.oracle_last_error().decode()}")
        self._h = h
        self.path = so_path
        self.n_fields = lib.oracle_dim(h)
        self.n_parameters = lib.oracle_n_par(h)
        self.name = lib.oracle_name(h).decode()

    def close(self):
        if getattr(self, "_h", None):
            _load().oracle_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- scalar helpers ---------------------------------------------------------------------
    def potential(self, x, p) -> float:
        x = np.ascontiguousarray(x, dtype=np.float64)
        p = np.ascontiguousarray(p, dtype=np.float64)
        return float(_load().oracle_potential(self._h, _dptr(x), _dptr(p)))

    def hesse(self, x, p) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float64)
        p = np.ascontiguousarray(p, dtype=np.float64)
        h = np.zeros(4)
        rc = _load().oracle_hesse(self._h, _dptr(x), _dptr(p), _dptr(h))
        if rc:
            raise OracleError(_load().oracle_last_error().decode())
        return h.reshape(2, 2)

    # -- sweeps -----------------------------------------------------------------------------
    def grid_sweep(self, op: int, p, extent, N0: int, N1: int, accuracy: float = 0.0, threads: int = 1) -> np.ndarray:
        """extent = (x0_start, x0_stop, x1_start, x1_stop). Returns (N0,N1[,k]) array."""
        lib = _load()
        p = np.ascontiguousarray(p, dtype=np.float64)
        ss = np.ascontiguousarray(extent, dtype=np.float64)
        if op == OP.QDIF:
            out = np.zeros((N0, N1), dtype=np.uint8)
        elif op in _WIDTH:
            out = np.zeros((N0, N1, _WIDTH[op]))
        else:
            out = np.zeros((N0, N1))
        rc = lib.oracle_grid_sweep(self._h, op, _dptr(p), p.size, out.ctypes.data_as(C.c_void_p), _dptr(ss), N0, N1, accuracy, threads)
        if rc:
            raise OracleError(f"[{rc}] {lib.oracle_last_error().decode()}")
        return out.astype(bool) if op == OP.QDIF else out

    def trajectory_sweep(self, op: int, p, traj, accuracy: float = 0.0, threads: int = 1) -> np.ndarray:
        lib = _load()
        p = np.ascontiguousarray(p, dtype=np.float64)
        traj = np.ascontiguousarray(traj, dtype=np.float64)
        n = traj.shape[0]
        if op == OP.QDIF:
            out = np.zeros((n,), dtype=np.uint8)
        elif op in _WIDTH:
            out = np.zeros((n, _WIDTH[op]))
        else:
            out = np.zeros((n,))
        rc = lib.oracle_trajectory_sweep(self._h, op, _dptr(p), p.size, _dptr(traj), n, out.ctypes.data_as(C.c_void_p), accuracy, threads)
        if rc:
            raise OracleError(f"[{rc}] {lib.oracle_last_error().decode()}")
        return out.astype(bool) if op == OP.QDIF else out

    def complete_analysis(self, p, extent, N0: int, N1: int, threads: int = 1) -> np.ndarray:
        return self.grid_sweep(OP.COMPLETE, p, extent, N0, N1, threads=threads)


def raw_long_double(so_path_ld: str, p, pts) -> np.ndarray:
    """V, v00, v10, v11, |dV|^2 at the (n,2) points, evaluated in x87 extended precision by a model
    object built from ``emit_c_source(..., long_double=True)``; rounded to float64, shape (n,5)."""
    lib = _load()
    p = np.ascontiguousarray(p, dtype=np.float64)
    pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 2)
    out = np.zeros((pts.shape[0], 5))
    rc = lib.oracle_raw_long_double(so_path_ld.encode(), _dptr(p), p.size, _dptr(pts), pts.shape[0], _dptr(out))
    if rc:
        raise OracleError(f"[{rc}] {lib.oracle_last_error().decode()}")
    return out


def grid_points(extent, N0: int, N1: int) -> np.ndarray:
    """The (N0*N1, 2) field-space points of a grid sweep, computed like src/anguelova.rs:531-533
    (index * spacing + offset, multiply then add, in float64)."""
    x0a, x0b, x1a, x1b = (float(v) for v in extent)
    dx0 = (x0b - x0a) / N0
    dx1 = (x1b - x1a) / N1
    x0 = np.arange(N0, dtype=np.float64) * dx0 + x0a
    x1 = np.arange(N1, dtype=np.float64) * dx1 + x1a
    return np.stack(np.meshgrid(x0, x1, indexing="ij"), axis=-1).reshape(-1, 2)
