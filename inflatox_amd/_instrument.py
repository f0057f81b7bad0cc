"""The transpiler's conditioning instrument: the generated stage code evaluated on the HOST, in float64 and in extended
precision, at a sample of points -- used by ``Compiler(regroup="auto", sample=...)`` to decide which model values may be
re-associated.  This is a transpile-time analysis, not a sweep path: nothing here is reachable from a sweep call, and a
host without a C++ compiler simply gets the reference's arithmetic (``choose_regroup`` returns the empty set).

Criterion (the one the parity tests apply, tests/tolerance.py, here as a design rule): a model value f_k may be regrouped
if, at every sample point,

    |f_k(regrouped) - f_k(reference form)|  <=  1e-10 |f_k|  +  C * E_k(point),

where E_k is the rounding error of the float64 evaluation of the REFERENCE form at that point, measured as |float64 -
extended precision| and maximised over the point and a few copies of it moved by a few ulps (rounding errors decorrelate
under such moves), and C = 4 -- a quarter of the smallest multiple the parity suite grants.  Where the reference's form
happens to be unusually accurate at special points of a grid (D5: v10 at theta = k pi/4, where its cancelling terms are
equal bit for bit) a regrouped form is not, and that value keeps the reference's arithmetic.
"""

from __future__ import annotations

import ctypes as C
import hashlib
import os
import shutil
import subprocess
import tempfile

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG, "csrc")
_DP = C.POINTER(C.c_double)
NAMES = ("V", "v00", "v10", "v11", "g")
RTOL, C_ERR, COPIES = 1e-10, 4.0, 6


def host_compiler() -> str | None:
    for cand in (os.environ.get("CXX"), "g++", "c++", "clang++", "/opt/rocm/lib/llvm/bin/clang++"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or shutil.which(cand)):
            return cand
    return None


def _private_build_dir() -> str:
    """A directory of this user's own (mode 0700) for the instrument's shared objects: they are dlopen'ed, so nobody
    else may be able to place files there."""
    d = os.path.join(tempfile.gettempdir(), f"inflx_instrument_{os.getuid()}")
    os.makedirs(d, mode=0o700, exist_ok=True)
    st = os.stat(d)
    if st.st_uid != os.getuid() or (st.st_mode & 0o077):
        d = tempfile.mkdtemp(prefix="inflx_instrument_")  # somebody else's, or too open: a fresh private one (no reuse)
    return d


class HostModel:
    """A generated model header compiled for the host (``long_double``: every double read as long double)."""

    def __init__(self, header_text: str, long_double: bool = False):
        cxx = host_compiler()
        if cxx is None:
            raise RuntimeError("no host C++ compiler for the conditioning instrument")
        # the shared object is a function of the header, the switch, the compiler AND the sources it includes
        key = hashlib.sha256((header_text + str(long_double) + cxx).encode())
        for dep in ("inflx_host_instrument.cpp", "inflx_ops.h", "inflx_device_math.h", "inflx_sf.h", "inflx_sf_tables.h"):
            with open(os.path.join(_CSRC, dep), "rb") as fh:
                key.update(fh.read())
        tag = key.hexdigest()[:20]
        d = _private_build_dir()
        hdr, so = os.path.join(d, f"{tag}.h"), os.path.join(d, f"{tag}.so")
        if not os.path.exists(so):
            with open(hdr + f".{os.getpid()}.tmp", "w") as fh:
                fh.write(header_text)
            os.replace(hdr + f".{os.getpid()}.tmp", hdr)
            cmd = [cxx, "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-w", f"-I{_CSRC}", f'-DINFLX_MODEL_HEADER="{hdr}"']
            if long_double:
                cmd.append("-DINFLX_INSTRUMENT_LONG_DOUBLE=1")
            cmd += [os.path.join(_CSRC, "inflx_host_instrument.cpp"), "-o", so + f".{os.getpid()}.tmp"]
            subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            os.replace(so + f".{os.getpid()}.tmp", so)
        self.lib = C.CDLL(so)
        self.lib.inflx_instrument_raw.argtypes = [_DP, _DP, C.c_size_t, _DP]
        self.lib.inflx_instrument_n_parameters.restype = C.c_uint
        self.lib.inflx_instrument_mantissa_bits.restype = C.c_uint
        self.mantissa_bits = int(self.lib.inflx_instrument_mantissa_bits())
        self.n_parameters = int(self.lib.inflx_instrument_n_parameters())

    def raw(self, args, pts) -> np.ndarray:
        args = np.ascontiguousarray(args, dtype=np.float64).reshape(-1)
        pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 2)
        if args.size != self.n_parameters:
            raise ValueError(f"sample has {args.size} parameters, the model {self.n_parameters}")
        out = np.zeros((pts.shape[0], 5))
        self.lib.inflx_instrument_raw(args.ctypes.data_as(_DP), pts.ctypes.data_as(_DP), pts.shape[0], out.ctypes.data_as(_DP))
        return out


def sample_points(extent) -> np.ndarray:
    """Two grids over the extent: the regular one a user's sweep lays down (index * spacing + start, 32 x 32: it hits the
    symmetric points of periodic models) and an offset one with odd counts."""
    x0a, x0b, x1a, x1b = (float(v) for v in extent)
    grids = []
    for n0, n1, off in ((32, 32, 0.0), (29, 31, 0.37)):
        i, j = np.meshgrid(np.arange(n0) + off, np.arange(n1) + off, indexing="ij")
        grids.append(np.column_stack([(i * ((x0b - x0a) / n0) + x0a).ravel(), (j * ((x1b - x1a) / n1) + x1a).ravel()]))
    return np.concatenate(grids)


def reference_error(exact: HostModel, exact_ld: HostModel, args, pts, seed: int = 20240):
    """(values of the reference form in float64 at pts, E): E = max over the point and COPIES few-ulp moves of it of
    |float64 - extended|, infinite where either is not finite."""
    rng = np.random.default_rng(seed)
    eps = np.finfo(np.float64).eps
    base = exact.raw(args, pts)
    env = np.zeros_like(base)
    for q in range(COPIES + 1):
        moved = pts if q == 0 else pts * (1.0 + eps * rng.integers(-8, 9, size=pts.shape))
        a = base if q == 0 else exact.raw(args, moved)
        t = exact_ld.raw(args, moved)
        with np.errstate(all="ignore"):
            e = np.abs(a - t)
        e[~(np.isfinite(a) & np.isfinite(t))] = np.inf
        env = np.maximum(env, e)
    return base, env


def choose_regroup(header_for, args, extent, log=None) -> frozenset:
    """Which of the five model values (indices 0..4 = V, v00, v10, v11, |dV|^2) pass the criterion of the module text.
    ``header_for(regroup)`` returns the generated header for a regroup argument (False, or a set of indices)."""
    say = log or (lambda *_: None)
    if host_compiler() is None:
        say("no host C++ compiler: the reference's arithmetic is kept")
        return frozenset()
    exact, exact_ld = HostModel(header_for(False)), HostModel(header_for(False), long_double=True)
    if exact_ld.mantissa_bits <= 53:
        say("long double is float64 on this host: no conditioning information, the reference's arithmetic is kept")
        return frozenset()
    pts = sample_points(extent)
    base, env = reference_error(exact, exact_ld, args, pts)
    chosen = frozenset(range(5))
    for _ in range(4):
        if not chosen:
            break
        got = HostModel(header_for(chosen)).raw(args, pts)
        keep = set()
        for k in sorted(chosen):
            with np.errstate(all="ignore"):
                allowed = RTOL * np.abs(base[:, k]) + C_ERR * env[:, k]
                both = np.isfinite(base[:, k]) & np.isfinite(got[:, k])
                ratio = np.where(both & np.isfinite(allowed), np.abs(got[:, k] - base[:, k]) / np.maximum(allowed, np.finfo(float).tiny), 0.0)
                # where the reference's own error is unbounded (singular points) nothing is compared but finiteness:
                # a regrouped form must not turn numbers into NaN (or the reverse) where the reference is robustly finite
                pattern = (np.isfinite(base[:, k]) != np.isfinite(got[:, k])) & np.isfinite(env[:, k])
            worst = float(ratio.max()) if ratio.size else 0.0
            ok = worst <= 1.0 and not pattern.any()
            say(f"regroup {NAMES[k]}: worst |regrouped - reference| / allowance {worst:.3g}, finiteness changes {int(pattern.sum())} -> {'yes' if ok else 'no'}")
            if ok:
                keep.add(k)
        if keep == set(chosen):
            return chosen
        chosen = frozenset(keep)  # the excluded values now come first in their exact form: verify the rest again
    return frozenset()
