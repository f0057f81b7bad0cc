"""Python front-end of the sweep: ``GeneralisedAL(artifact).complete_analysis(...)``.

Same class names, method names, argument order, defaults and return contracts as the
reference's front-end (python/inflatox/consistency_conditions.py:31-715); every method hands
its arrays to ``libinflx_hip.so`` (through :mod:`inflatox_amd._native`) where the reference hands
them to ``libinflx_rs``.  Extensions that the reference lacks are keyword-only (``device``) or
separate methods (``complete_analysis_batch``), so reference call sites run unchanged.
"""

from __future__ import annotations

import threading

import numpy as np

from . import _native
from ._native import InflatoxDevLib, InflatoxMultiLib, open_inflx_dylib
from ._result_pool import result_array
from .compiler import CompilationArtifact

__all__ = ["InflationCondition", "GeneralisedAL"]


def _start_stop(x0_start, x0_stop, x1_start, x1_stop) -> np.ndarray:
    # consistency_conditions.py:292-294: rows = fields, columns = (start, stop)
    return np.array([[float(x0_start), float(x0_stop)], [float(x1_start), float(x1_stop)]])


class InflationCondition:
    """Base class: owns the opened model artefact (reference consistency_conditions.py:31-50)."""

    def __init__(self, compiled_artifact: CompilationArtifact, validate_basis: bool = True, *, device: int = 0, devices=None, tuned: bool = False, sf_errors: str | None = None):
        """``sf_errors`` (extension, keyword-only): what a call does when the model evaluated a Bessel / hypergeometric function
        outside its domain somewhere on the grid.  ``"raise"``: :class:`inflatox_amd._native.InflatoxSpecialFunctionError` once the
        sweep has finished -- the reference's GSL error handler panics there (src/err.rs:86-103); ``"nan"``: the arrays are
        returned with NaN at those points.  Default (``None``): ``"raise"`` for ``Compiler(link_gsl=True)`` artefacts -- the only
        ones the reference can build such a model from --, ``"nan"`` otherwise.

        ``tuned`` (extension, keyword-only, default off): profile-guided build.  The FIRST grid sweep of this object hands
        its own parameter values and field range to ``artifact.profile_guided`` -- the call supplies everything the measurement
        needs (reference consistency_conditions.py:226-308: args, the four range values) -- and this and every later call
        run on that build (EGNO 4096^2: 0.40 -> 0.32 ms of device time).  Results then agree with the reference within the
        parity criterion instead of reproducing its arithmetic operation for operation; ``retune(args, extent)`` measures
        again for another region of parameter space.  The default (``False``) is the reference's arithmetic, unchanged.

        ``device`` (extension, keyword-only): the HIP device of this object.  ``devices`` (extension): a sequence of device
        indices or ``"all"`` -- the grid sweeps (``complete_analysis`` and the single-quantity sweeps, the batch and summary
        extensions) then run on ALL of them from one call: the outermost axis of the sweep is split into one block per
        device and every device copies its block straight into the result array (``inflx_sweep_host_multi``).  This is the
        meaning the reference gives its ``threads`` argument -- ``None`` -> 0 -> the whole machine
        (consistency_conditions.py:297, anguelova.rs:524-540) -- with GPUs as the workers: ``threads=k`` limits a call to the
        first k devices.  Default: one device, as before."""
        self.artifact = compiled_artifact
        self.multi: InflatoxMultiLib | None = None
        self._devices = devices
        if devices is not None:
            self.multi = InflatoxMultiLib(compiled_artifact.shared_object_path, devices)
            device = self.multi.devices[0]
        self.dylib: InflatoxDevLib = open_inflx_dylib(compiled_artifact.shared_object_path, validate_basis, device=device)
        self._sf_errors = sf_errors
        self._apply_sf_errors(self.dylib, self.multi)
        self._tune_pending = bool(tuned)
        self._tune_lock = threading.Lock()  # two first sweeps that arrive together measure and swap handles once
        self.tuned_on = None  # (args, extent) the profile-guided build was measured on

    def _apply_sf_errors(self, dylib, multi) -> None:
        if self._sf_errors is not None:
            dylib.set_sf_errors(self._sf_errors)
            if multi is not None:
                multi.set_sf_errors(self._sf_errors)

    def retune(self, args, extent) -> None:
        """Extension: (re)build the profile-guided code object for the parameter values ``args`` and the field range
        ``extent = (x0_start, x0_stop, x1_start, x1_stop)`` and run every later call of this object on it.  The handles of the build
        in use so far are closed (their device buffers and streams are released at once, not at some later garbage collection)."""
        sample_args = np.asarray(args, dtype=np.float64)
        sample_args = sample_args.reshape(-1, sample_args.shape[-1])[0]  # a batch is measured on its first parameter row
        extent = tuple(float(v) for v in np.asarray(extent, dtype=np.float64).reshape(-1))
        art = self.artifact.profile_guided(sample_args, extent)
        device = self.dylib.device
        new_multi = InflatoxMultiLib(art.shared_object_path, self._devices) if self._devices is not None else None
        new_dylib = open_inflx_dylib(art.shared_object_path, False, device=device)  # the basis was validated on the default build
        self._apply_sf_errors(new_dylib, new_multi)
        old_dylib, old_multi = self.dylib, self.multi
        self.dylib, self.multi, self.artifact = new_dylib, new_multi, art
        self._tune_pending = False
        self.tuned_on = (sample_args.copy(), extent)
        old_dylib.close()
        if old_multi is not None:
            old_multi.close()

    def _before_sweep(self, args, start_stop) -> None:
        """With ``tuned=True``: the first call that knows its parameter values AND a field range -- every grid sweep -- measures and
        switches builds, once (concurrent first calls wait for the one that does).  Calls made before that -- and the calls that have
        no range to measure on: ``calc_V`` / ``calc_H``, the ``*_ot`` variants, the basis validation -- run on
        the default build, i.e. in the reference's arithmetic; after the first grid sweep they run on the profile-guided build like
        everything else (same parity criterion; ``tuned_on`` says whether and on what the object has switched)."""
        if self._tune_pending:
            with self._tune_lock:
                if self._tune_pending:
                    self.retune(args, start_stop)

    # -- scalar helpers (reference :52-65,103-117); evaluated on the device through the raw op ----
    def _raw_at(self, x, args) -> np.ndarray:
        x = np.asarray(x, dtype=np.float64).reshape(1, 2)
        return self.dylib.sweep_on_trajectory(_native.OP_RAW, args, x)[0]

    def calc_V(self, x: np.ndarray, args: np.ndarray) -> float:
        """Scalar potential at field-space point ``x``."""
        return float(self._raw_at(x, args)[0])

    def calc_H(self, x: np.ndarray, args: np.ndarray) -> np.ndarray:
        """Projected Hesse matrix at ``x``: ``[[v00, v01], [v10, v11]]`` like the reference's ``hesse`` (src/lib.rs:384-420),
        every component the value of the reference's own C function of that name -- v01 included: it is evaluated on its own
        wherever its expression is not, node for node, the expression of v10 (``INFLX_OP_HESSE``)."""
        x = np.asarray(x, dtype=np.float64).reshape(1, 2)
        return self.dylib.sweep_on_trajectory(_native.OP_HESSE, args, x)[0].reshape(2, 2)

    # -- array helpers (reference :67-101,119-156); off the sweep path, served by the raw-values sweep ----
    def _raw_planes(self, args, x0_start, x0_stop, x1_start, x1_stop, N, first, count):
        """Planes [first, first + count) of the raw values (V, v00, v10, v11, |dV|^2) on the grid; only those cross PCIe."""
        n0, n1 = (int(v) for v in (N if N is not None else (8000, 8000)))
        ss = _start_stop(x0_start, x0_stop, x1_start, x1_stop)
        return self.dylib.sweep_host_planes(_native.OP_RAW, args, ss, n0, n1, first, count)

    def calc_V_array(self, args, start, stop, N=None) -> np.ndarray:
        """Potential on the grid ``start[i] + k*(stop[i]-start[i])/N[i]`` (end point excluded), shape ``N``
        (reference consistency_conditions.py:67-101, hesse_bindings.rs:68-85)."""
        return self._raw_planes(args, start[0], stop[0], start[1], stop[1], N, 0, 1)[0]

    def calc_H_array(self, args, x0_start, x0_stop, x1_start, x1_stop, N=None) -> np.ndarray:
        """Projected Hesse matrix on the grid, shape (2, 2, N0, N1) as the reference documents
        (consistency_conditions.py:119-156; its implementation passes the wrong arguments to the native
        helper and cannot run, so the documented contract is what is implemented): an ordinary writable array whose
        ``H[a, b]`` is the plane of the reference's C function ``v{a}{b}`` -- ``H[0, 1]`` the reference's own v01, in memory of
        its own (``INFLX_OP_HESSE``, four planes; 2 GB at the default 8000 x 8000 grid, as in the reference)."""
        n0, n1 = (int(v) for v in (N if N is not None else (8000, 8000)))
        ss = _start_stop(x0_start, x0_stop, x1_start, x1_stop)
        planes = self.dylib.sweep_host(_native.OP_HESSE, args, ss, n0, n1, layout=_native.LAYOUT_SOA)  # (4, N0, N1): v00, v01, v10, v11
        return planes.reshape(2, 2, n0, n1)

    def validate_basis_on_domain(self, args, start, stop, N=100, accuracy: float = 1e-3) -> None:
        """Checks that the basis {v, w1} is orthonormal (to ``accuracy``) on sample points of the domain
        and raises an exception if it is not (reference consistency_conditions.py:158-196,
        src/lib.rs:207-300).  The sample points are the reference's: each axis in turn is walked in
        ``N[axis]`` steps of ``(stop-start)/N`` beginning at its *stop* value (lib.rs:252), the other
        coordinate held at its start value.  An ``int`` N applies to both axes (the reference intends
        that, consistency_conditions.py:194, but its ``N is int`` test never fires)."""
        n_fields = self.artifact.n_fields
        start_stop = np.array([[float(a), float(b)] for (a, b) in zip(start, stop)])
        if isinstance(N, (int, np.integer)):
            N = [int(N)] * n_fields
        self.dylib.validate_basis_on_domain(N, args, start_stop, accuracy)


class GeneralisedAL(InflationCondition):
    """Generalised Anguelova-Lazaroiu consistency condition and the quantities derived from it
    (reference consistency_conditions.py:199-715)."""

    def __init__(self, compiled_artifact: CompilationArtifact, *, device: int = 0, devices=None, tuned: bool = False, sf_errors: str | None = None):
        # like the reference (consistency_conditions.py:222-224 -> :38), the constructor validates the basis
        super().__init__(compiled_artifact, device=device, devices=devices, tuned=tuned, sf_errors=sf_errors)

    # ---- the hot path ------------------------------------------------------------------------
    def complete_analysis(
        self,
        args: np.ndarray,
        x0_start: float,
        x0_stop: float,
        x1_start: float,
        x1_stop: float,
        N_x0: int = 1_000,
        N_x1: int = 1_000,
        progress: bool = True,
        threads: None | int = None,
        *,
        broadcast_views: bool = False,
    ) -> tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
        """Six (N_x0, N_x1) arrays: consistency, ε_V, ε_H, η_∥, δ, ω -- element [i, j] belongs to
        x0 = x0_start + i·(x0_stop-x0_start)/N_x0, x1 likewise (end point excluded).
        Reference: consistency_conditions.py:226-308; like there, the six arrays are strided views
        of one (N_x0, N_x1, 6) array.

        ``broadcast_views`` (extension, keyword-only, default off): for a model none of whose values depends on x1 (or on
        x0) the (N_x0, N_x1, 6) result is N_x1 (N_x0) copies of one line; with this flag only that line is evaluated and
        copied to the host, and the six arrays are READ-ONLY ``np.broadcast_to`` views of it -- same shape, same values,
        stride 0 along the constant axis.  The hyperbolic 8192 x 8192 sweep then moves 393 kB over PCIe instead of
        3.2 GB (65 ms -> < 2 ms).  Models that depend on both axes take the ordinary path whatever the flag says."""
        self._before_sweep(args, (x0_start, x0_stop, x1_start, x1_stop))
        if broadcast_views:
            line = self._constant_axis(N_x0, N_x1)
            if line is not None:
                axis, shape = line
                start_stop = _start_stop(x0_start, x0_stop, x1_start, x1_stop)
                if axis == 1:  # nothing depends on x1: one column of the grid, same x0 axis
                    small = result_array((N_x0, 1, 6))
                    start_stop[1] = [float(x1_start), float(x1_start) + (float(x1_stop) - float(x1_start)) / N_x1]
                else:  # nothing depends on x0: one row
                    small = result_array((1, N_x1, 6))
                    start_stop[0] = [float(x0_start), float(x0_start) + (float(x0_stop) - float(x0_start)) / N_x0]
                self.dylib.complete_analysis(args, small, start_stop, progress, threads if threads is not None else 0)
                return tuple(np.broadcast_to(small[:, :, k], shape) for k in range(6))
        # the reference's np.zeros((N_x0, N_x1, 6)); recycled page-resident memory when a previous result of this size
        # has been dropped (_result_pool.py) -- the sweep writes every element
        out = result_array((N_x0, N_x1, 6))
        start_stop = _start_stop(x0_start, x0_stop, x1_start, x1_stop)
        threads = threads if threads is not None else 0
        # (with `devices=...` the call runs on all of them, `threads` limiting how many: the reference's "0 = whole machine")
        (self.multi or self.dylib).complete_analysis(args, out, start_stop, progress, threads)
        return (out[:, :, 0], out[:, :, 1], out[:, :, 2], out[:, :, 3], out[:, :, 4], out[:, :, 5])

    def _constant_axis(self, N_x0: int, N_x1: int):
        """(axis along which the result is constant, full shape) for a model whose five sweep values ignore one field."""
        mask = self.dylib.stage_info["out_mask"]
        if N_x0 < 1 or N_x1 < 1:
            return None
        if mask & 2 == 0:
            return 1, (N_x0, N_x1)
        if mask & 1 == 0:
            return 0, (N_x0, N_x1)
        return None

    def complete_analysis_device(self, args, x0_start, x0_stop, x1_start, x1_stop, N_x0: int = 1_000, N_x1: int = 1_000):
        """Extension: the same sweep with a DEVICE-RESIDENT result -- six ``torch.Tensor`` views (strides as in
        :meth:`complete_analysis`) of one (N_x0, N_x1, 6) float64 tensor on this object's GPU.  Nothing crosses PCIe;
        the tensors speak DLPack (``__dlpack__``) and ``__cuda_array_interface__``, so CuPy / JAX / numba consumers
        take them without a copy.  The sweep is ordered after what torch's current stream has enqueued and torch's
        current stream is ordered after the sweep: the tensors can be used like the result of any torch operation."""
        import torch

        self._before_sweep(args, (x0_start, x0_stop, x1_start, x1_stop))
        device = torch.device("cuda", self.dylib.device)
        if getattr(self, "_torch_stream", None) is None:
            self._torch_stream = torch.cuda.Stream(device=device)
        out = torch.empty((N_x0, N_x1, 6), dtype=torch.float64, device=device)
        if out.numel():
            consumer = torch.cuda.current_stream(device)
            self._torch_stream.wait_stream(consumer)  # `out` was allocated on the consumer's stream
            self.dylib.sweep_device(_native.OP_COMPLETE, args, out.data_ptr(), out.numel() * 8, _start_stop(x0_start, x0_stop, x1_start, x1_stop), N_x0, N_x1, stream=self._torch_stream.cuda_stream)
            consumer.wait_stream(self._torch_stream)
            out.record_stream(self._torch_stream)
        return tuple(out[:, :, k] for k in range(6))

    def complete_analysis_batch(
        self,
        args: np.ndarray,
        x0_start: float,
        x0_stop: float,
        x1_start: float,
        x1_stop: float,
        N_x0: int = 1_000,
        N_x1: int = 1_000,
        layout: str = "aos",
    ) -> np.ndarray:
        """Extension: the outer parameter axes of the sweep in one call.  ``args`` is (P, n_parameters) -- or an N-D
        parameter grid (P_1, ..., P_k, n_parameters), e.g. from :meth:`parameter_grid` --; returns (P, N_x0, N_x1, 6)
        resp. (P_1, ..., P_k, N_x0, N_x1, 6) (``layout='aos'``), or (..., 6, N_x0, N_x1) (``'soa'``).  The parameter axes are
        swept as one flat outermost axis (C order); with ``devices=...`` that axis is what the devices share."""
        lay = {"aos": _native.LAYOUT_AOS, "soa": _native.LAYOUT_SOA}[layout]
        self._before_sweep(args, (x0_start, x0_stop, x1_start, x1_stop))
        ss = _start_stop(x0_start, x0_stop, x1_start, x1_stop)
        rows = np.atleast_2d(np.asarray(args, dtype=np.float64))
        lead = rows.shape[:-1]
        out = (self.multi or self.dylib).sweep_host(_native.OP_COMPLETE, rows.reshape(-1, rows.shape[-1]), ss, N_x0, N_x1, layout=lay)
        return out.reshape(lead + out.shape[1:])

    def parameter_grid(self, args, axes: dict) -> np.ndarray:
        """Extension: the N-D parameter grid of a scan.  ``args`` is the base parameter vector, ``axes`` maps parameters --
        a position in ``args``, a sympy symbol of the model or its printed name (``artifact.symbol_dictionary``) -- to
        1-D arrays of values; returns (len_1, ..., len_k, n_parameters) in the order of ``axes``, every other parameter held
        at its base value: the ``args`` of :meth:`complete_analysis_batch` / :meth:`complete_analysis_summary`."""
        base = np.asarray(args, dtype=np.float64).reshape(-1)
        if base.size != self.artifact.n_parameters:
            raise _native.InflatoxShapeError(f"expected {self.artifact.n_parameters} parameters (got {base.size})")
        slots, values = [], []
        for key, vals in axes.items():
            if isinstance(key, (int, np.integer)):
                k = int(key)
            else:
                name = key if isinstance(key, str) else self.artifact.symbol_printer._print_Symbol(key)
                target = self.artifact.symbol_dictionary.get(name, "")
                if not target.startswith("args["):
                    raise KeyError(f"{key!r} is not a parameter of this model (parameters: {[n for n, t in self.artifact.symbol_dictionary.items() if t.startswith('args[')]})")
                k = int(target[5:-1])
            if not 0 <= k < base.size or k in slots:
                raise KeyError(f"parameter position {k} out of range or named twice")
            slots.append(k)
            values.append(np.asarray(vals, dtype=np.float64).reshape(-1))
        shape = tuple(v.size for v in values)
        grid = np.empty(shape + (base.size,))
        grid[...] = base
        for d, (k, v) in enumerate(zip(slots, values)):
            grid[..., k] = v.reshape((1,) * d + (-1,) + (1,) * (len(shape) - d - 1))
        return grid

    def complete_analysis_summary(self, args, x0_start, x0_stop, x1_start, x1_stop, N_x0=1_000, N_x1=1_000) -> dict:
        """Extension: NaN-ignoring minimum, maximum and non-NaN count of the six quantities over the sweep,
        reduced on the GPU inside the sweep kernels -- what ``np.nanmin/np.nanmax`` over the six arrays of
        :meth:`complete_analysis` would give (the reference's tests do exactly that, tests/test_doc.py:58),
        without materialising or copying the arrays.  ``args`` may be (P, n_parameters) or an N-D parameter grid
        (..., n_parameters); the summary covers all of its rows."""
        self._before_sweep(args, (x0_start, x0_stop, x1_start, x1_stop))
        ss = _start_stop(x0_start, x0_stop, x1_start, x1_stop)
        rows = np.asarray(args, dtype=np.float64)
        return (self.multi or self.dylib).sweep_stats(rows.reshape(-1, rows.shape[-1]) if rows.ndim > 2 else rows, ss, N_x0, N_x1)

    # ---- single-quantity sweeps (reference :310-475) ---------------------------------------------
    def _single(self, name, args, x0_start, x0_stop, x1_start, x1_stop, N_x0, N_x1, progress, threads):
        self._before_sweep(args, (x0_start, x0_stop, x1_start, x1_stop))
        fn = getattr(self.dylib, name)
        out = result_array((N_x0, N_x1))
        start_stop = _start_stop(x0_start, x0_stop, x1_start, x1_stop)
        threads = threads if threads is not None else 0
        if self.multi is not None:
            op = {"consistency_only": _native.OP_CONSISTENCY, "epsilon_v_only": _native.OP_EPSILON_V, "consistency_rapidturn_only": _native.OP_RAPIDTURN}[fn.__name__]
            self.multi.sweep_host(op, np.asarray(args, dtype=np.float64).reshape(1, -1), start_stop, N_x0, N_x1, progress=progress, max_devices=threads, out=out.reshape(1, N_x0, N_x1))
            return out
        fn(args, out, start_stop, progress, threads)
        return out

    def consistency(self, args, x0_start, x0_stop, x1_start, x1_stop, N_x0=1_000, N_x1=1_000, progress=True, threads=None) -> np.ndarray:
        return self._single("consistency_only", args, x0_start, x0_stop, x1_start, x1_stop, N_x0, N_x1, progress, threads)

    def epsilon_v(self, args, x0_start, x0_stop, x1_start, x1_stop, N_x0=1_000, N_x1=1_000, progress=True, threads=None) -> np.ndarray:
        return self._single("epsilon_v_only", args, x0_start, x0_stop, x1_start, x1_stop, N_x0, N_x1, progress, threads)

    def consistency_rapidturn(self, args, x0_start, x0_stop, x1_start, x1_stop, N_x0=1_000, N_x1=1_000, progress=True, threads=None) -> np.ndarray:
        return self._single("consistency_rapidturn_only", args, x0_start, x0_stop, x1_start, x1_stop, N_x0, N_x1, progress, threads)

    def flag_quantum_dif(self, args, x0_start, x0_stop, x1_start, x1_stop, N_x0=10_000, N_x1=10_000, progress=True, accuracy=1e-3) -> np.ndarray:
        """Boolean (N_x0, N_x1) array: True where both components of the normalised potential gradient
        are <= ``accuracy`` (reference consistency_conditions.py:477-523, src/anguelova.rs:166-170)."""
        self._before_sweep(args, (x0_start, x0_stop, x1_start, x1_stop))
        x = result_array((N_x0, N_x1), dtype=bool)
        self.dylib.flag_quantum_dif(args, x, _start_stop(x0_start, x0_stop, x1_start, x1_stop), progress, accuracy)
        return x

    # ---- on-trajectory variants (reference :529-715) ---------------------------------------------
    def complete_analysis_ot(self, args, x, progress=True, threads=None):
        threads = threads if threads is not None else 1
        out = self.dylib.sweep_on_trajectory(_native.OP_COMPLETE, args, x, progress, threads)
        return np.split(out, 6, 1)

    def consistency_ot(self, args, x, progress=True, threads=None) -> np.ndarray:
        threads = threads if threads is not None else 1
        return self.dylib.sweep_on_trajectory(_native.OP_CONSISTENCY, args, x, progress, threads)

    def consistency_rapidturn_ot(self, args, x, progress=True, threads=None) -> np.ndarray:
        threads = threads if threads is not None else 1
        return self.dylib.sweep_on_trajectory(_native.OP_RAPIDTURN, args, x, progress, threads)

    def epsilon_v_ot(self, args, x, progress=True, threads=None) -> np.ndarray:
        threads = threads if threads is not None else 1
        return self.dylib.sweep_on_trajectory(_native.OP_EPSILON_V, args, x, progress, threads)
