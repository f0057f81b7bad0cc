"""ctypes binding of ``libinflx_hip.so`` (C ABI: ``include/inflx_hip.h``).

This is the stand-in for the reference's PyO3 module ``libinflx_rs`` on the sweep path
(src/lib.rs:68-92).  There is deliberately no fallback: if the shared library is missing, or
no HIP device is usable, every call raises -- the product never computes on the CPU.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_PKG)
LIB_PATH = os.path.join(_PKG, "libinflx_hip.so")

OK, ERR_IO, ERR_SYMBOL, ERR_VERSION, ERR_SHAPE, ERR_DEVICE, ERR_ARG, ERR_BASIS, ERR_GSL = range(9)
SF_EDOM, SF_EDECLINED = 1, 2  # inflx_sf_bits
SF_QUIET, SF_FAIL = 0, 1  # inflx_sf_policy_t

OP_COMPLETE, OP_CONSISTENCY, OP_RAPIDTURN, OP_EPSILON_V, OP_RAW = range(5)
OP_HESSE = 6  # v00, v01, v10, v11: the projected Hesse matrix with the reference's own v01 (INFLX_SWEEP_HESSE)
OP_QDIF = 5
# kernel group (inflatox_amd.compiler.KERNEL_GROUPS) of every operation: the artefact's core object holds complete_analysis, the others
# are built and attached on first use
OP_GROUP = {OP_COMPLETE: "core", OP_CONSISTENCY: "consistency", OP_RAPIDTURN: "rapidturn", OP_EPSILON_V: "epsilon_v", OP_RAW: "raw", OP_QDIF: "qdif", OP_HESSE: "hesse"}
OP_WIDTH = {OP_COMPLETE: 6, OP_CONSISTENCY: 1, OP_RAPIDTURN: 1, OP_EPSILON_V: 1, OP_RAW: 5, OP_HESSE: 4}
LAYOUT_AOS, LAYOUT_SOA = 0, 1
SWEEP_DEFAULT, SWEEP_FORCE_TILE = 0, 1  # inflx_sweep_flags
TIME_BACK_TO_BACK, TIME_DOMINANT_ONLY, TIME_IN_PIPELINE, TIME_SINGLE_CALL = range(4)  # inflx_timing
GATHER_PEER_PUSH, GATHER_RCCL = 0, 1  # inflx_gather: the exchange step of inflx_sweep_allgather_multi_ex

_DP = C.POINTER(C.c_double)
_SIZE = C.c_size_t

# name -> (restype, argtypes); one entry per symbol declared in include/inflx_hip.h
SIGNATURES = {
    "inflx_last_error": (C.c_char_p, []),
    "inflx_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "inflx_open": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
    "inflx_close": (None, [C.c_void_p]),
    "inflx_attach": (C.c_int, [C.c_void_p, C.c_char_p]),
    "inflx_groups": (C.c_uint32, [C.c_void_p]),
    "inflx_n_fields": (C.c_uint32, [C.c_void_p]),
    "inflx_n_parameters": (C.c_uint32, [C.c_void_p]),
    "inflx_model_name": (C.c_char_p, [C.c_void_p]),
    "inflx_device_of": (C.c_int, [C.c_void_p]),
    "inflx_stage_info": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_uint32)] * 4),
    "inflx_complete_analysis": (C.c_int, [C.c_void_p, _DP, _SIZE, _DP, _DP, _SIZE, _SIZE, C.c_int, _SIZE]),
    "inflx_consistency_only": (C.c_int, [C.c_void_p, _DP, _SIZE, _DP, _DP, _SIZE, _SIZE, C.c_int, _SIZE]),
    "inflx_consistency_rapidturn_only": (C.c_int, [C.c_void_p, _DP, _SIZE, _DP, _DP, _SIZE, _SIZE, C.c_int, _SIZE]),
    "inflx_epsilon_v_only": (C.c_int, [C.c_void_p, _DP, _SIZE, _DP, _DP, _SIZE, _SIZE, C.c_int, _SIZE]),
    "inflx_flag_quantum_dif": (C.c_int, [C.c_void_p, _DP, _SIZE, C.POINTER(C.c_uint8), _DP, _SIZE, _SIZE, C.c_int, C.c_double]),
    "inflx_sweep_on_trajectory": (C.c_int, [C.c_void_p, C.c_int, _DP, _SIZE, _DP, _SIZE, _DP, C.c_int, _SIZE]),
    "inflx_sweep_host": (C.c_int, [C.c_void_p, C.c_int, _DP, _SIZE, _SIZE, _DP, _DP, _SIZE, _SIZE, _SIZE, _SIZE, C.c_int]),
    "inflx_sweep_host_planes": (C.c_int, [C.c_void_p, C.c_int, _DP, _SIZE, _SIZE, _DP, _DP, _SIZE, _SIZE, _SIZE, _SIZE, _SIZE, _SIZE]),
    "inflx_sweep_device": (
        C.c_int,
        [C.c_void_p, C.c_int, _DP, _SIZE, _SIZE, C.c_void_p, _SIZE, _DP, _SIZE, _SIZE, _SIZE, _SIZE, C.c_int, C.c_void_p],
    ),
    "inflx_sweep_device_timed": (
        C.c_int,
        [C.c_void_p, C.c_int, _DP, _SIZE, _SIZE, C.c_void_p, _SIZE, _DP, _SIZE, _SIZE, _SIZE, _SIZE, C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)],
    ),
    "inflx_sweep_device_ex": (
        C.c_int,
        [C.c_void_p, C.c_int, _DP, _SIZE, _SIZE, C.c_void_p, _SIZE, _DP, _SIZE, _SIZE, _SIZE, _SIZE, C.c_int, C.c_void_p, C.c_uint],
    ),
    "inflx_sweep_device_timed_ex": (
        C.c_int,
        [C.c_void_p, C.c_int, _DP, _SIZE, _SIZE, C.c_void_p, _SIZE, _DP, _SIZE, _SIZE, _SIZE, _SIZE, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_uint, C.POINTER(C.c_float)],
    ),
    "inflx_sweep_plan_ex": (C.c_int, [C.c_void_p, C.c_int, _SIZE, _SIZE, _SIZE, C.c_int, C.c_uint, C.POINTER(C.c_uint32)]),
    "inflx_host_threads": (C.c_int, [C.c_uint, C.POINTER(C.c_uint)]),
    "inflx_sweep_device_stats": (
        C.c_int,
        [C.c_void_p, _DP, _SIZE, _SIZE, C.c_void_p, _SIZE, _DP, _SIZE, _SIZE, _SIZE, _SIZE, C.c_void_p, C.c_void_p],
    ),
    "inflx_synchronize": (C.c_int, [C.c_void_p]),
    "inflx_sf_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint), C.c_int]),
    "inflx_sf_policy": (C.c_int, [C.c_void_p, C.c_int]),
    "inflx_uses_gsl": (C.c_int, [C.c_void_p]),
    "inflx_sweep_plan": (C.c_int, [C.c_void_p, C.c_int, _SIZE, _SIZE, _SIZE, C.c_int, C.POINTER(C.c_uint32)]),
    "inflx_basis_on_points": (C.c_int, [C.c_void_p, _DP, _SIZE, _DP, _SIZE, _DP]),
    "inflx_ops_on_values": (C.c_int, [C.c_void_p, _DP, _SIZE, _DP, C.c_int]),
    "inflx_validate_basis_at_random": (C.c_int, [C.c_void_p, C.c_uint64]),
    "inflx_validate_basis_on_domain": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), _SIZE, _DP, _SIZE, _DP, C.c_double]),
    "inflx_shard_plan": (C.c_int, [_SIZE, _SIZE, C.c_int, C.c_int, C.POINTER(_SIZE)]),
    "inflx_open_multi": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "inflx_close_multi": (None, [C.c_void_p]),
    "inflx_multi_device_count": (C.c_int, [C.c_void_p]),
    "inflx_multi_handle": (C.c_void_p, [C.c_void_p, C.c_int]),
    "inflx_sweep_host_multi": (C.c_int, [C.c_void_p, C.c_int, _DP, _SIZE, _SIZE, _DP, _DP, _SIZE, _SIZE, C.c_int, C.c_int, _SIZE]),
    "inflx_complete_analysis_multi": (C.c_int, [C.c_void_p, _DP, _SIZE, _DP, _DP, _SIZE, _SIZE, C.c_int, _SIZE]),
    "inflx_sweep_stats_multi": (C.c_int, [C.c_void_p, _DP, _SIZE, _SIZE, _DP, _SIZE, _SIZE, _SIZE, C.c_void_p]),
    "inflx_sweep_device_multi": (C.c_int, [C.c_void_p, C.c_int, _DP, _SIZE, _SIZE, C.POINTER(C.c_void_p), C.POINTER(_SIZE), _DP, _SIZE, _SIZE, C.c_int, C.POINTER(C.c_void_p)]),
    "inflx_sweep_allgather_multi": (C.c_int, [C.c_void_p, C.c_int, _DP, _SIZE, _SIZE, C.POINTER(C.c_void_p), _SIZE, _DP, _SIZE, _SIZE]),
    "inflx_sweep_allgather_multi_ex": (C.c_int, [C.c_void_p, C.c_int, _DP, _SIZE, _SIZE, C.POINTER(C.c_void_p), _SIZE, _DP, _SIZE, _SIZE, C.c_int]),
}


class Summary(C.Structure):
    """``inflx_summary``: NaN-ignoring min / max and the non-NaN count of the six outputs."""

    _fields_ = [("min", C.c_double * 6), ("max", C.c_double * 6), ("count", C.c_uint64 * 6)]

_lib = None


def build_library(force: bool = False) -> str:
    """Compile ``csrc/inflx_hip.cpp`` into ``libinflx_hip.so`` in-tree (hipcc, host code only)."""
    src = os.path.join(_PKG, "csrc", "inflx_hip.cpp")
    deps = [src, os.path.join(_PKG, "csrc", "inflx_kernel_abi.h"), os.path.join(_REPO, "include", "inflx_hip.h")]
    stale = force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(d) > os.path.getmtime(LIB_PATH) for d in deps)
    if stale:
        from .compiler import hipcc_path

        cmd = [
            hipcc_path(),
            "-O2",
            "-fPIC",
            "-shared",
            "-std=c++17",
            "-Wall",
            "-Wextra",
            f"-I{os.path.join(_REPO, 'include')}",
            f"-I{os.path.join(_PKG, 'csrc')}",
            src,
            "-o",
            LIB_PATH + ".tmp",
        ]
        subprocess.run(cmd, check=True)
        os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


def _preload_hip_runtime():
    """Make sure the process ends up with ONE HIP runtime.

    PyTorch wheels bundle their own ``libamdhip64.so`` (SONAME libamdhip64.so.7) and request it by
    the unversioned name; ``libinflx_hip.so`` requests ``libamdhip64.so.7``.  If this library were
    loaded first it would pull in /opt/rocm's copy, a later ``import torch`` would then load the
    bundled copy as a *second* runtime, and torch would find no devices.  Loading torch's copy
    first (by path, without importing torch) lets both requests resolve to the same object.
    """
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load_library():
    """Load ``libinflx_hip.so`` and declare every entry point; raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(the sweep has no CPU fallback)"
            )
        _preload_hip_runtime()
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class InflatoxShapeError(Exception):
    """Counterpart of LibInflxRsErr::Shape, which PyO3 raises as a plain Exception (err.rs:71)."""


class InflatoxBasisError(Exception):
    """Counterpart of LibInflxRsErr::BasisNorm / BasisOth (err.rs:36-37), plain Exceptions in PyO3 too."""


class InflatoxSpecialFunctionError(ArithmeticError):
    """A special function of the model was called outside its domain somewhere in the call (``INFLX_ERR_GSL``).  The reference's
    GSL error handler prints the reason and panics at that point (src/err.rs:86-103): the call does not return there either.  Here
    the result was complete when the error was raised -- NaN at the offending points -- and ``sf_errors="nan"`` returns it."""


def _raise(rc: int):
    msg = load_library().inflx_last_error().decode("utf-8", "replace")
    if rc == ERR_GSL:
        raise InflatoxSpecialFunctionError(msg)
    if rc == ERR_IO:
        raise IOError(msg)
    if rc in (ERR_SYMBOL, ERR_VERSION, ERR_DEVICE):
        raise SystemError(msg)
    if rc == ERR_SHAPE:
        raise InflatoxShapeError(msg)
    if rc == ERR_BASIS:
        raise InflatoxBasisError(msg)
    raise ValueError(msg)


def _check(rc: int):
    if rc != OK:
        _raise(rc)


def _sf_policy(mode: str) -> int:
    try:
        return {"raise": SF_FAIL, "nan": SF_QUIET}[mode]
    except (KeyError, TypeError):
        raise ValueError(f'sf_errors must be "raise" or "nan" (got {mode!r})') from None


def _f64(a, name: str) -> np.ndarray:
    """C-contiguous float64 view/copy.  (The reference panics on non-contiguous input,
    anguelova.rs:473; here it is made contiguous instead.)"""
    try:
        return np.ascontiguousarray(a, dtype=np.float64)
    except (TypeError, ValueError) as exc:
        raise ValueError(f"{name} must be convertible to a float64 array") from exc


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(_DP)


def device_count() -> int:
    n = C.c_int(0)
    rc = load_library().inflx_device_count(C.byref(n))
    return n.value if rc == OK else 0


def _ensure_group(lib, handles, art, group: str) -> None:
    """Make the kernels of ``group`` available on every handle in ``handles``: nothing to do when they are loaded; otherwise the
    artefact ``art`` the handles were opened from builds the group (one hipcc step on first use, ``CompilationArtifact.ensure_group``) and it
    is attached.  An artefact from elsewhere (a bare file) is left to ``libinflx_hip.so``, which looks for ``<artefact>.<group>``
    itself and says what is missing."""
    from .compiler import KERNEL_GROUPS

    bit = KERNEL_GROUPS[group]
    pending = [h for h in handles if not (int(lib.inflx_groups(h)) & bit)]
    if not pending or art is None:
        return
    path = art.ensure_group(group)
    if path is None:
        return
    for h in pending:
        _check(lib.inflx_attach(h, os.fsencode(path)))


class InflatoxDevLib:
    """An opened model artefact on one HIP device (counterpart of ``InflatoxPyDyLib``, lib.rs:104)."""

    def __init__(self, artefact_path: str, device: int = 0):
        lib = load_library()
        handle = C.c_void_p()
        _check(lib.inflx_open(os.fsencode(artefact_path), int(device), C.byref(handle)))
        self._h = handle
        self._lib = lib
        self.path = artefact_path
        # the artefact this file is the core object of (None: a bare file): it builds the kernel groups of the other operations on
        # first use, so the handle keeps it alive for as long as it lives itself
        from .compiler import artefact_for_path

        self._artefact = artefact_for_path(artefact_path)
        self.device = int(device)
        self.n_fields = int(lib.inflx_n_fields(handle))
        self.n_parameters = int(lib.inflx_n_parameters(handle))
        self.name = lib.inflx_model_name(handle).decode("utf-8", "replace")
        vals = [C.c_uint32(0) for _ in range(4)]
        _check(lib.inflx_stage_info(handle, *[C.byref(v) for v in vals]))
        self.stage_info = dict(zip(("n_uniform", "n_row", "n_col", "out_mask"), (v.value for v in vals)))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.inflx_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _need(self, group: str) -> None:
        """The kernel group of the operation about to run (built and attached on first use; see ``_ensure_group``)."""
        _ensure_group(self._lib, [self._h], self._artefact, group)

    # ---- special functions outside their domain (include/inflx_hip.h: inflx_sf_policy) -----------
    @property
    def uses_gsl(self) -> bool:
        """The artefact's ``USE_GSL`` global: ``Compiler(link_gsl=True)``."""
        return bool(self._lib.inflx_uses_gsl(self._h))

    def sf_status(self, clear: bool = True) -> int:
        """Bits (``SF_EDOM``, ``SF_EDECLINED``) the model's special functions have set since they were last cleared; waits for
        the handle's streams."""
        bits = C.c_uint(0)
        _check(self._lib.inflx_sf_status(self._h, C.byref(bits), int(bool(clear))))
        return int(bits.value)

    def set_sf_errors(self, mode: str) -> None:
        """``"raise"``: a call that evaluated a special function outside its domain raises :class:`InflatoxSpecialFunctionError`
        (the default for ``link_gsl=True`` artefacts, where the reference panics); ``"nan"``: NaN at the point, nothing else."""
        _check(self._lib.inflx_sf_policy(self._h, _sf_policy(mode)))

    @property
    def groups(self) -> int:
        """Bit mask of the kernel groups loaded so far (``inflx_groups``; bits: ``inflatox_amd.compiler.KERNEL_GROUPS``)."""
        return int(self._lib.inflx_groups(self._h))

    # ---- drop-ins for the #[pyfunction]s of src/anguelova.rs ---------------------------------
    def _grid(self, fn, p, out, start_stop, progress, threads, last_axis):
        p = _f64(p, "p").reshape(-1)
        ss = _f64(start_stop, "start_stop")
        if ss.shape != (2, 2):  # convert_start_stop, src/lib.rs:117-139
            raise InflatoxShapeError(f"start_stop array should have 2 rows and as many columns as there are fields (got {ss.shape})")
        if out.dtype != np.float64 or not out.flags.c_contiguous or not out.flags.writeable:
            raise ValueError("output array must be a writeable C-contiguous float64 array")
        if last_axis is not None and (out.ndim != 3 or out.shape[2] != last_axis):
            raise InflatoxShapeError(f"Output array should be 3D. Last axis must have lenght {last_axis} (got {out.shape})")
        if last_axis is None and out.ndim != 2:
            raise InflatoxShapeError(f"Output array should be 2D (got {out.shape})")
        _check(fn(self._h, _ptr(p), p.size, _ptr(out), _ptr(ss), out.shape[0], out.shape[1], int(bool(progress)), int(threads)))

    def complete_analysis(self, p, out, start_stop, progress=False, threads=0):
        """libinflx_rs.complete_analysis(lib, p, out, start_stop, progress, threads), anguelova.rs:458."""
        self._grid(self._lib.inflx_complete_analysis, p, out, start_stop, progress, threads, 6)

    def consistency_only(self, p, out, start_stop, progress=False, threads=0):
        self._need("consistency")
        self._grid(self._lib.inflx_consistency_only, p, out, start_stop, progress, threads, None)

    def consistency_rapidturn_only(self, p, out, start_stop, progress=False, threads=0):
        self._need("rapidturn")
        self._grid(self._lib.inflx_consistency_rapidturn_only, p, out, start_stop, progress, threads, None)

    def epsilon_v_only(self, p, out, start_stop, progress=False, threads=0):
        self._need("epsilon_v")
        self._grid(self._lib.inflx_epsilon_v_only, p, out, start_stop, progress, threads, None)

    def flag_quantum_dif(self, p, out, start_stop, progress=False, accuracy=1e-3):
        """libinflx_rs.flag_quantum_dif_py(lib, p, x, start_stop, progress, accuracy), anguelova.rs:574."""
        p = _f64(p, "p").reshape(-1)
        ss = _f64(start_stop, "start_stop")
        if ss.shape != (2, 2):
            raise InflatoxShapeError(f"start_stop array should have 2 rows and as many columns as there are fields (got {ss.shape})")
        if out.dtype != np.bool_ or out.ndim != 2 or not out.flags.c_contiguous or not out.flags.writeable:
            raise ValueError("output array must be a writeable C-contiguous 2-D bool array")
        self._need("qdif")
        _check(
            self._lib.inflx_flag_quantum_dif(
                self._h, _ptr(p), p.size, out.view(np.uint8).ctypes.data_as(C.POINTER(C.c_uint8)), _ptr(ss), out.shape[0], out.shape[1], int(bool(progress)), float(accuracy)
            )
        )

    def sweep_on_trajectory(self, op, p, x, progress=False, threads=0) -> np.ndarray:
        p = _f64(p, "p").reshape(-1)
        x = _f64(x, "x")
        if x.ndim != 2 or x.shape[1] != 2:
            raise InflatoxShapeError(f"trajectory array should have shape (n,2) (got {x.shape})")
        k = OP_WIDTH[op]
        self._need(OP_GROUP[op])
        from ._result_pool import result_array

        shape = (x.shape[0], k) if k > 1 else (x.shape[0],)
        # long trajectories: recycled page-resident memory (every element is written by the sweep); short ones: a plain array
        out = result_array(shape) if x.shape[0] * k * 8 >= (1 << 20) else np.zeros(shape)
        _check(self._lib.inflx_sweep_on_trajectory(self._h, op, _ptr(p), p.size, _ptr(x), x.shape[0], _ptr(out), int(bool(progress)), int(threads)))
        return out

    # ---- basis validation (src/lib.rs:141-300) --------------------------------------------------
    def basis_on_points(self, p, x) -> np.ndarray:
        """(n,7): v.v, v.w1, w1.w1, v[0], v[1], w1[0], w1[1] at the points ``x`` (n,2)."""
        p = _f64(p, "p").reshape(-1)
        x = _f64(x, "x")
        if x.ndim != 2 or x.shape[1] != 2:
            raise InflatoxShapeError(f"point array should have shape (n,2) (got {x.shape})")
        out = np.zeros((x.shape[0], 7))
        _check(self._lib.inflx_basis_on_points(self._h, _ptr(p), p.size, _ptr(x), x.shape[0], _ptr(out)))
        return out

    def ops_on_values(self, values, ieee_only: bool = False) -> np.ndarray:
        """(n,9): ``ops::complete_analysis`` [0..5], ``consistency_only`` [6], ``consistency_rapidturn_only`` [7] and
        ``epsilon_v_only`` [8] (src/anguelova.rs:99-163) of n given records (V, v00, v10, v11, |dV|^2) -- the model's own
        functions are not evaluated.  ``ieee_only``: the compiler's IEEE divisions throughout instead of the sweep
        kernels' spelling (bit-identical by construction; the tests assert it)."""
        values = _f64(values, "values")
        if values.ndim != 2 or values.shape[1] != 5:
            raise InflatoxShapeError(f"values array should have shape (n,5) (got {values.shape})")
        out = np.zeros((values.shape[0], 9))
        self._need("values")
        _check(self._lib.inflx_ops_on_values(self._h, _ptr(values), values.shape[0], _ptr(out), int(bool(ieee_only))))
        return out

    def validate_basis_at_random(self, seed: int = 0) -> None:
        _check(self._lib.inflx_validate_basis_at_random(self._h, int(seed)))

    def validate_basis_on_domain(self, num_points, p, start_stop, accuracy) -> None:
        """``validate_basis_on_domain(num_points, p, start_stop, accuracy)`` of src/lib.rs:207-212."""
        n = np.ascontiguousarray(num_points, dtype=np.uint32).reshape(-1)
        p = _f64(p, "p").reshape(-1)
        ss = _f64(start_stop, "start_stop")
        if ss.ndim != 2 or ss.shape[1] != 2 or ss.shape[0] != n.size:
            raise InflatoxShapeError(f"start_stop array should have 2 rows and as many columns as there are fields (got {ss.shape})")
        _check(
            self._lib.inflx_validate_basis_on_domain(
                self._h, n.ctypes.data_as(C.POINTER(C.c_uint32)), n.size, _ptr(p), p.size, _ptr(ss), float(accuracy)
            )
        )

    # ---- generalised sweeps -------------------------------------------------------------------
    def sweep_host(self, op, p, start_stop, N0, N1, row_begin=0, row_count=None, layout=LAYOUT_AOS) -> np.ndarray:
        """P parameter rows x rows [row_begin,row_begin+row_count) of the grid -> host ndarray."""
        p = _f64(p, "p")
        single = p.ndim == 1
        p2 = p.reshape(1, -1) if single else p
        ss = _f64(start_stop, "start_stop").reshape(-1)
        row_count = N0 - row_begin if row_count is None else row_count
        P, k = p2.shape[0], OP_WIDTH[op]
        if k == 1:
            shape = (P, row_count, N1)
        elif layout == LAYOUT_AOS:
            shape = (P, row_count, N1, k)
        else:
            shape = (P, k, row_count, N1)
        from ._result_pool import result_array

        out = result_array(shape)  # recycled page-resident memory where a dropped result of this size is at hand
        self._need(OP_GROUP[op])
        _check(self._lib.inflx_sweep_host(self._h, op, _ptr(p2), P, p2.shape[1], _ptr(out), _ptr(ss), N0, N1, row_begin, row_count, layout))
        return out[0] if single else out

    def sweep_host_planes(self, op, p, start_stop, N0, N1, first_plane, n_planes, row_begin=0, row_count=None) -> np.ndarray:
        """Planes ``[first_plane, first_plane + n_planes)`` of the planes-layout result of ``op``: (P, n_planes, rows, N1), or
        (n_planes, rows, N1) for a single parameter row.  Only those planes are copied to the host."""
        p = _f64(p, "p")
        single = p.ndim == 1
        p2 = p.reshape(1, -1) if single else p
        ss = _f64(start_stop, "start_stop").reshape(-1)
        row_count = N0 - row_begin if row_count is None else row_count
        from ._result_pool import result_array

        out = result_array((p2.shape[0], n_planes, row_count, N1))
        self._need(OP_GROUP[op])
        _check(self._lib.inflx_sweep_host_planes(self._h, op, _ptr(p2), p2.shape[0], p2.shape[1], _ptr(out), _ptr(ss), N0, N1, row_begin, row_count, first_plane, n_planes))
        return out[0] if single else out

    def sweep_device(self, op, p, d_out_ptr: int, d_out_bytes: int, start_stop, N0, N1, row_begin=0, row_count=None, layout=LAYOUT_AOS, stream: int = 0, force_tile: bool = False):
        """Enqueue a sweep whose result stays in device memory at ``d_out_ptr`` (no sync).  ``force_tile``
        (``INFLX_SWEEP_FORCE_TILE``): every grid point through the tile kernels, also for a model that ignores a grid axis."""
        p2 = _f64(p, "p")
        p2 = p2.reshape(1, -1) if p2.ndim == 1 else p2
        ss = _f64(start_stop, "start_stop").reshape(-1)
        row_count = N0 - row_begin if row_count is None else row_count
        self._need(OP_GROUP[op])
        _check(
            self._lib.inflx_sweep_device_ex(
                self._h, op, _ptr(p2), p2.shape[0], p2.shape[1], C.c_void_p(d_out_ptr), d_out_bytes, _ptr(ss), N0, N1, row_begin, row_count, layout, C.c_void_p(stream),
                SWEEP_FORCE_TILE if force_tile else SWEEP_DEFAULT,
            )
        )

    def sweep_device_timed(self, op, p, d_out_ptr, d_out_bytes, start_stop, N0, N1, row_begin=0, row_count=None, layout=LAYOUT_AOS, stream: int = 0, repeats: int = 10, dominant_only: bool = False, in_pipeline: bool = False, single_call: bool = False, force_tile: bool = False) -> float:
        """Mean duration (ms) of one sweep over ``repeats`` repetitions, HIP events on the launch stream.  Default: the sweeps
        enqueued back to back (throughput of a scan: the tables of sweep n+1 are evaluated under the kernels of sweep n);
        ``dominant_only`` times just the dominant kernel of a multi-launch sweep, relaunched on its own; ``in_pipeline``
        enqueues the full sweeps and returns the dominant kernel's time per sweep from event pairs around its launches;
        ``single_call``: ONE whole call (tables / per-row values + sweep kernel) from an idle handle, mean over ``repeats`` calls."""
        p2 = _f64(p, "p")
        p2 = p2.reshape(1, -1) if p2.ndim == 1 else p2
        ss = _f64(start_stop, "start_stop").reshape(-1)
        row_count = N0 - row_begin if row_count is None else row_count
        mode = TIME_SINGLE_CALL if single_call else (TIME_IN_PIPELINE if in_pipeline else (TIME_DOMINANT_ONLY if dominant_only else TIME_BACK_TO_BACK))
        ms = C.c_float(0.0)
        self._need(OP_GROUP[op])
        _check(
            self._lib.inflx_sweep_device_timed_ex(
                self._h, op, _ptr(p2), p2.shape[0], p2.shape[1], C.c_void_p(d_out_ptr), d_out_bytes, _ptr(ss), N0, N1, row_begin, row_count, layout, C.c_void_p(stream), repeats, mode,
                SWEEP_FORCE_TILE if force_tile else SWEEP_DEFAULT, C.byref(ms)
            )
        )
        return float(ms.value)

    def sweep_stats(self, p, start_stop, N0, N1, row_begin=0, row_count=None, d_out_ptr: int = 0, d_out_bytes: int = 0, stream: int = 0) -> dict:
        """complete_analysis sweep with the summary reduced on the device; with ``d_out_ptr == 0`` nothing
        but the summary is produced.  Returns ``{"min": (P-agnostic) array(6), "max": array(6), "count": array(6)}``."""
        p2 = _f64(p, "p")
        p2 = p2.reshape(1, -1) if p2.ndim == 1 else p2
        ss = _f64(start_stop, "start_stop").reshape(-1)
        row_count = N0 - row_begin if row_count is None else row_count
        out = Summary()
        self._need("stats")
        _check(
            self._lib.inflx_sweep_device_stats(
                self._h, _ptr(p2), p2.shape[0], p2.shape[1], C.c_void_p(d_out_ptr), d_out_bytes, _ptr(ss), N0, N1, row_begin, row_count, C.c_void_p(stream), C.byref(out)
            )
        )
        return {"min": np.array(out.min[:]), "max": np.array(out.max[:]), "count": np.array(out.count[:], dtype=np.uint64)}

    def sweep_plan(self, op, P, N1, row_count, layout=LAYOUT_AOS, force_tile: bool = False) -> dict:
        """Which kernels a sweep of this shape takes: ``{"path": "tile"|"row_stream"|"rows"|"col_stream", "batch_rows", "batches", "replicas"}``;
        for the tile path ``batch_rows`` = parameter rows per launch, ``batches`` = launches, and ``tile_rows`` = grid rows per workgroup
        tile of the first launch.  ``host_threads``: the process's host-thread budget and the helper threads a host-result call of this
        handle starts (``inflx_host_threads``)."""
        plan = (C.c_uint32 * 4)()
        _check(self._lib.inflx_sweep_plan_ex(self._h, op, P, N1, row_count, layout, SWEEP_FORCE_TILE if force_tile else SWEEP_DEFAULT, plan))
        path = ("tile", "row_stream", "rows", "col_stream")[plan[0]]
        out = {"path": path, "batch_rows": int(plan[1]), "batches": int(plan[2]), "replicas": int(plan[3])}
        if path == "tile":
            out["tile_rows"] = out.pop("replicas")
        out["host_threads"] = host_threads()
        return out

    def synchronize(self):
        _check(self._lib.inflx_synchronize(self._h))


def host_threads(devices_at_work: int = 1) -> dict:
    """``inflx_host_threads``: the process's CPU budget (affinity mask cut down to the cgroup quota) and the helper threads one device
    pipeline of a host-result call starts when ``devices_at_work`` pipelines run at once (no device needed)."""
    out = (C.c_uint * 3)()
    _check(load_library().inflx_host_threads(int(devices_at_work), out))
    return {"budget": int(out[0]), "residency": int(out[1]), "fill": int(out[2])}


def shard_plan(P: int, N0: int, world: int, rank: int) -> dict:
    """``inflx_shard_plan``: the block of the (parameter rows x grid rows) index space that part ``rank`` of ``world`` owns in a
    multi-device sweep (no device needed)."""
    plan = (_SIZE * 5)()
    _check(load_library().inflx_shard_plan(P, N0, world, rank, plan))
    return {"axis": ("param", "rows")[plan[0]], "p_begin": int(plan[1]), "p_count": int(plan[2]), "row_begin": int(plan[3]), "row_count": int(plan[4])}


class InflatoxMultiLib:
    """A model artefact opened on several HIP devices (``inflx_multi``): one call sweeps on all of them.

    ``devices``: a sequence of device indices (a device may appear more than once) or ``"all"`` / ``None`` for every visible
    device.  The counterpart of the reference's ``threads=0`` ("use the whole machine", anguelova.rs:524-540)."""

    def __init__(self, artefact_path: str, devices="all"):
        lib = load_library()
        handle = C.c_void_p()
        if devices is None or (isinstance(devices, str) and devices == "all"):
            _check(lib.inflx_open_multi(os.fsencode(artefact_path), None, 0, C.byref(handle)))
        else:
            ids = [int(d) for d in devices]
            if not ids:
                raise ValueError("devices must name at least one device")
            arr = (C.c_int * len(ids))(*ids)
            _check(lib.inflx_open_multi(os.fsencode(artefact_path), arr, len(ids), C.byref(handle)))
        self._h = handle
        self._lib = lib
        self.path = artefact_path
        from .compiler import artefact_for_path

        self._artefact = artefact_for_path(artefact_path)
        self.n_devices = int(lib.inflx_multi_device_count(handle))
        self.devices = [int(lib.inflx_device_of(lib.inflx_multi_handle(handle, k))) for k in range(self.n_devices)]
        first = lib.inflx_multi_handle(handle, 0)
        self.n_parameters = int(lib.inflx_n_parameters(first))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.inflx_close_multi(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _need(self, group: str) -> None:
        handles = [self._lib.inflx_multi_handle(self._h, k) for k in range(self.n_devices)]
        _ensure_group(self._lib, handles, self._artefact, group)

    def set_sf_errors(self, mode: str) -> None:
        """:meth:`InflatoxDevLib.set_sf_errors` for every device of the handle."""
        policy = _sf_policy(mode)
        for k in range(self.n_devices):
            _check(self._lib.inflx_sf_policy(self._lib.inflx_multi_handle(self._h, k), policy))

    def complete_analysis(self, p, out, start_stop, progress=False, threads=0):
        """``libinflx_rs.complete_analysis(lib, p, out, start_stop, progress, threads)`` (anguelova.rs:458) on the handle's
        devices: ``threads`` = 0 uses all of them, k at most k."""
        p = _f64(p, "p").reshape(-1)
        ss = _f64(start_stop, "start_stop")
        if ss.shape != (2, 2):
            raise InflatoxShapeError(f"start_stop array should have 2 rows and as many columns as there are fields (got {ss.shape})")
        if out.dtype != np.float64 or not out.flags.c_contiguous or not out.flags.writeable:
            raise ValueError("output array must be a writeable C-contiguous float64 array")
        if out.ndim != 3 or out.shape[2] != 6:
            raise InflatoxShapeError(f"Output array should be 3D. Last axis must have lenght 6 (got {out.shape})")
        _check(self._lib.inflx_complete_analysis_multi(self._h, _ptr(p), p.size, _ptr(out), _ptr(ss), out.shape[0], out.shape[1], int(bool(progress)), int(threads)))

    def sweep_host(self, op, p, start_stop, N0, N1, layout=LAYOUT_AOS, progress=False, max_devices=0, out=None) -> np.ndarray:
        """P parameter rows x the whole grid -> host ndarray, every device sweeping and copying its own block."""
        p = _f64(p, "p")
        single = p.ndim == 1
        p2 = p.reshape(1, -1) if single else p
        ss = _f64(start_stop, "start_stop").reshape(-1)
        P, k = p2.shape[0], OP_WIDTH[op]
        if k == 1:
            shape = (P, N0, N1)
        elif layout == LAYOUT_AOS:
            shape = (P, N0, N1, k)
        else:
            shape = (P, k, N0, N1)
        if out is None:
            from ._result_pool import result_array

            out = result_array(shape)
        elif out.shape != shape or out.dtype != np.float64 or not out.flags.c_contiguous or not out.flags.writeable:
            raise ValueError(f"out must be a writeable C-contiguous float64 array of shape {shape}")
        self._need(OP_GROUP[op])
        _check(self._lib.inflx_sweep_host_multi(self._h, op, _ptr(p2), P, p2.shape[1], _ptr(out), _ptr(ss), N0, N1, layout, int(bool(progress)), int(max_devices)))
        return out[0] if single else out

    def sweep_device(self, op, p, d_out_ptrs, d_out_bytes, start_stop, N0, N1, layout=LAYOUT_AOS, streams=None):
        """Device k sweeps its block (``shard_plan(P, N0, n_devices, k)``) into ``d_out_ptrs[k]`` (device memory on device k, laid
        out as the block); asynchronous on ``streams[k]`` (``None``: the handle's own stream)."""
        p2 = _f64(p, "p")
        p2 = p2.reshape(1, -1) if p2.ndim == 1 else p2
        ss = _f64(start_stop, "start_stop").reshape(-1)
        n = self.n_devices
        ptrs = (C.c_void_p * n)(*[C.c_void_p(int(v)) for v in d_out_ptrs])
        sizes = (_SIZE * n)(*[int(v) for v in d_out_bytes])
        st = None if streams is None else (C.c_void_p * n)(*[C.c_void_p(int(v)) for v in streams])
        self._need(OP_GROUP[op])
        _check(self._lib.inflx_sweep_device_multi(self._h, op, _ptr(p2), p2.shape[0], p2.shape[1], ptrs, sizes, _ptr(ss), N0, N1, layout, st))

    def sweep_allgather(self, op, p, d_full_ptrs, d_full_bytes, start_stop, N0, N1, gather="peer_push"):
        """Every ``d_full_ptrs[k]`` (device memory on device k, the whole (P, N0, N1, K) array) holds the whole result on return:
        each device sweeps its block in place; ``gather="peer_push"``: it then pushes the block to every peer over its own xGMI
        link (``hipMemcpyPeerAsync``); ``gather="rccl"``: ONE in-place ``ncclAllGather`` per contiguous image instead (equal
        blocks, one device per handle)."""
        p2 = _f64(p, "p")
        p2 = p2.reshape(1, -1) if p2.ndim == 1 else p2
        ss = _f64(start_stop, "start_stop").reshape(-1)
        ptrs = (C.c_void_p * self.n_devices)(*[C.c_void_p(int(v)) for v in d_full_ptrs])
        if gather not in ("peer_push", "rccl"):
            raise ValueError(f"unknown gather {gather!r}: choose 'peer_push' or 'rccl'")
        mode = {"peer_push": GATHER_PEER_PUSH, "rccl": GATHER_RCCL}[gather]
        self._need(OP_GROUP[op])
        _check(self._lib.inflx_sweep_allgather_multi_ex(self._h, op, _ptr(p2), p2.shape[0], p2.shape[1], ptrs, int(d_full_bytes), _ptr(ss), N0, N1, mode))

    def sweep_stats(self, p, start_stop, N0, N1, max_devices=0) -> dict:
        p2 = _f64(p, "p")
        p2 = p2.reshape(1, -1) if p2.ndim == 1 else p2
        ss = _f64(start_stop, "start_stop").reshape(-1)
        out = Summary()
        self._need("stats")
        _check(self._lib.inflx_sweep_stats_multi(self._h, _ptr(p2), p2.shape[0], p2.shape[1], _ptr(ss), N0, N1, int(max_devices), C.byref(out)))
        return {"min": np.array(out.min[:]), "max": np.array(out.max[:]), "count": np.array(out.count[:], dtype=np.uint64)}


def open_inflx_dylib(lib_path: str, check_basis: bool = True, device: int = 0) -> InflatoxDevLib:
    """Counterpart of ``libinflx_rs.open_inflx_dylib(lib_path, check_basis)`` (src/lib.rs:108-115).

    With ``check_basis`` the basis vectors of the artefact are evaluated on the device at 100 random
    points for one random parameter vector and tested for orthonormality (lib.rs:142-199); a defective
    basis raises :class:`InflatoxBasisError` (a plain ``Exception``, as in the reference).
    """
    lib = InflatoxDevLib(lib_path, device=device)
    if check_basis:
        lib.validate_basis_at_random()
    return lib
