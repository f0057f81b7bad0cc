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


# ---- the per-point operations on given model values ------------------------------------------------
_values_model = None


def ops_on_values(values, threads: int = 1) -> np.ndarray:
    """(n,9): op_complete_analysis [0..5], op_consistency_only [6], op_consistency_rapidturn_only [7] and
    op_epsilon_v_only [8] of ``sweep_oracle.c`` (src/anguelova.rs:99-163) for n given records
    (V, v00, v10, v11, grad_norm_squared).  The records reach the oracle's unchanged sweep code through
    ``values_model.c``, a model artefact whose functions read them from a table (x[0] = record number)."""
    global _values_model
    from .model_c import compile_c_model

    values = np.ascontiguousarray(values, dtype=np.float64)
    assert values.ndim == 2 and values.shape[1] == 5, values.shape
    if _values_model is None:
        with open(os.path.join(_HERE, "values_model.c")) as fh:
            so = compile_c_model(fh.read(), flags=["-O2", "-shared", "-fPIC", "-std=c17", "-Wall", "-Werror"])
        _values_model = (OracleModel(so), C.CDLL(so))
    om, dll = _values_model
    C.c_void_p.in_dll(dll, "inflx_values_table").value = values.ctypes.data
    n = values.shape[0]
    traj = np.zeros((n, 2))
    traj[:, 0] = np.arange(n)
    out = np.empty((n, 9))
    p = np.zeros(1)
    try:
        out[:, :6] = om.trajectory_sweep(OP.COMPLETE, p, traj, threads=threads)
        out[:, 6] = om.trajectory_sweep(OP.CONSISTENCY, p, traj, threads=threads)
        out[:, 7] = om.trajectory_sweep(OP.RAPIDTURN, p, traj, threads=threads)
        out[:, 8] = om.trajectory_sweep(OP.EPSILON_V, p, traj, threads=threads)
    finally:
        C.c_void_p.in_dll(dll, "inflx_values_table").value = None
    return out


# ---- basis validation (reference src/lib.rs:141-300) --------------------------------------------
import math


class BasisDefect(Exception):
    """LibInflxRsErr::BasisNorm / BasisOth (src/err.rs:36-37): ``kind`` is "norm" or "oth"."""

    def __init__(self, kind, vectors, value, point):
        super().__init__(f"{kind} {vectors} {value} at {point}")
        self.kind, self.vectors, self.value, self.point = kind, vectors, value, point


def basis_on_points(so_path: str, p, pts) -> np.ndarray:
    """(n,7): v.v, v.w1, w1.w1, v[0], v[1], w1[0], w1[1], obtained the way src/lib.rs:164-169 does: the C
    functions ``v``/``w1`` fill the vectors, ``inner_prod`` contracts them with the metric.  Small n only."""
    so = C.CDLL(so_path)
    dp = C.POINTER(C.c_double)
    for name in ("v", "w1"):
        getattr(so, name).argtypes = [dp, dp, dp]
        getattr(so, name).restype = None
    so.inner_prod.argtypes = [dp, dp, dp, dp]
    so.inner_prod.restype = C.c_double
    p = np.ascontiguousarray(p, dtype=np.float64)
    pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 2)
    out = np.zeros((pts.shape[0], 7))
    vec = [np.zeros(2), np.zeros(2)]
    for k in range(pts.shape[0]):
        x = np.ascontiguousarray(pts[k])
        so.v(_dptr(x), _dptr(p), _dptr(vec[0]))
        so.w1(_dptr(x), _dptr(p), _dptr(vec[1]))
        for q, (i, j) in enumerate(((0, 0), (0, 1), (1, 1))):
            out[k, q] = so.inner_prod(_dptr(x), _dptr(p), _dptr(vec[i]), _dptr(vec[j]))
        out[k, 3:5], out[k, 5:7] = vec[0], vec[1]
    return out


def _is_normal(v: float) -> bool:
    return math.isfinite(v) and abs(v) >= 2.2250738585072014e-308


def check_basis(basis: np.ndarray, pts: np.ndarray, accuracy: float) -> int:
    """The per-point tests of src/lib.rs:164-193; returns the number of points at which some inner
    product was not a normal number, raises :class:`BasisDefect` at the first violation."""
    failed = 0
    for k in range(basis.shape[0]):
        nan = False
        for q, (i, j) in enumerate(((0, 0), (0, 1), (1, 1))):
            ip = float(basis[k, q])
            if i == j:
                if not _is_normal(ip):
                    nan = True
                elif abs(ip - 1.0) >= accuracy:
                    raise BasisDefect("norm", (i,), ip, tuple(pts[k]))
            else:
                if not _is_normal(ip) and ip != 0.0:
                    nan = True
                elif abs(ip) >= accuracy:
                    raise BasisDefect("oth", (i, j), ip, tuple(pts[k]))
        failed += nan
    return failed


def domain_points(num_points, start_stop) -> list:
    """Sample points of ``validate_basis_on_domain`` (src/lib.rs:247-256), one (n,2) array per axis: the
    walk along an axis starts at that axis' STOP value, the other coordinate sits at its start value."""
    ss = np.asarray(start_stop, dtype=np.float64).reshape(-1, 2)
    out = []
    for axis, n in enumerate(num_points):
        start, stop = ss[axis]
        spacing = (stop - start) / float(n)
        pts = np.tile(ss[:, 0], (int(n), 1))
        pts[:, axis] = stop + spacing * np.arange(int(n), dtype=np.float64)
        out.append(pts)
    return out


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
