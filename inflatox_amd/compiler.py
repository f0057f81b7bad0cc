"""Transpiler: sympy expressions -> HIP device code -> gfx950 code object.

Drop-in for the reference's ``Compiler`` / ``CompilationArtifact`` / ``CInflatoxPrinter``
(python/inflatox/compiler.py:37-120, 215-276, 279-650) with a different back-end: instead
of a C file with one ``double f(const double x[], const double args[])`` per quantity that the
native sweep calls through ``dlsym``'d pointers (five indirect calls per grid point), this
module emits ONE header of ``__device__ __forceinline__`` functions that is compiled together
with the hand-written sweep kernels (``csrc/inflx_sweep_kernels.hip``) into a per-model
gfx950 code object.  What that buys (none of it available across the reference's dylib
boundary):

  * sub-expressions shared between V, v00, v10, v11 and |dV|^2 are evaluated once;
  * *axis staging* (see staging.py): every sub-expression is classified by which grid axis it
    depends on -- nothing but parameters (U), x[0] only (R, "row"), x[1] only (C, "column"), or
    both (P, "point") -- and is evaluated once per kernel / per row / per column / per point
    respectively.  The kernels keep R values in LDS and C values in registers;
  * integer and half-integer powers become multiplication chains (the reference leaves
    ``pow(x, 4)`` to libm; OCML's generic f64 pow costs hundreds of VALU instructions).

What is kept bit-for-bit: the symbol table.  Parameters are numbered in order of first
appearance while printing in the reference's emission order (compiler.py:102-106, 474-539),
so a user's ``args`` array means the same thing here as there.
"""

from __future__ import annotations

import hashlib
import os
import re
import shutil
import subprocess
import sys
import tempfile
import weakref

import numpy as np
import sympy
from sympy.printing.c import C99CodePrinter

from .staging import HIPInflatoxPrinter, emit_stage_header  # noqa: F401  (HIPInflatoxPrinter re-exported)
from .symbolic import InflationModel
from .version import __abi_version__, __version__

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")

# Kernel groups of a model's code objects (csrc/inflx_kernel_abi.h INFLX_GROUP_*): name -> bit.  ``Compiler.compile()`` builds the
# CORE object -- everything complete_analysis needs -- in one hipcc step, like the reference's one ``zig cc`` step
# (python/inflatox/compiler.py:568-598); the group of every other operation is built when first used
# (``CompilationArtifact.ensure_group``) and loaded beside the core object (``inflx_attach``).
KERNEL_GROUPS = {"core": 1, "stats": 2, "values": 4, "consistency": 8, "rapidturn": 16, "epsilon_v": 32, "raw": 64, "qdif": 128, "hesse": 256}
ALL_GROUPS = 511
# op number (csrc/inflx_kernel_abi.h InflxOp) -> group name
GROUP_OF_OP = {0: "core", 1: "consistency", 2: "rapidturn", 3: "epsilon_v", 4: "raw", 5: "qdif", 6: "hesse"}

# artefacts of this process by the path of their core object: how _native finds the artefact that can build a missing group
_ARTEFACTS: "weakref.WeakValueDictionary[str, CompilationArtifact]" = weakref.WeakValueDictionary()


def artefact_for_path(path: str):
    """The live :class:`CompilationArtifact` whose core object is ``path`` (None: the file came from somewhere else)."""
    return _ARTEFACTS.get(os.path.abspath(path))

# The reference's generated C defines the <math.h> M_* constants itself, truncated to 12
# significant digits, whenever the header does not provide them -- and under the strict
# ``-std=c17`` it compiles with, glibc and musl do not (compiler.py:72-88, 299-310).  These
# truncated values are therefore what the reference computes with; they are reproduced by
# default (``exact_constants=False``) so that results agree with it.  Keys are sympy's macro
# names (C99CodePrinter.math_macros).
_REFERENCE_CONSTANTS = {
    "M_E": "2.71828182846",
    "M_LOG2E": "1.44269504089",
    "M_LOG10E": "0.4342944819",
    "M_LN2": "0.69314718056",
    "M_LN10": "2.30258509299",
    "M_PI": "3.14159265359",
    "M_PI_2": "1.57079632679",
    "M_PI_4": "0.78539816339",
    "M_1_PI": "0.31830988618",
    "M_2_PI": "0.63661977236",
    "M_2_SQRTPI": "1.1283791671",
    "M_SQRT2": "1.41421356237",
    "M_SQRT1_2": "0.70710678118",
}
_EXACT_CONSTANTS = {
    "M_E": "2.7182818284590452354",
    "M_LOG2E": "1.4426950408889634074",
    "M_LOG10E": "0.43429448190325182765",
    "M_LN2": "0.69314718055994530942",
    "M_LN10": "2.30258509299404568402",
    "M_PI": "3.14159265358979323846",
    "M_PI_2": "1.57079632679489661923",
    "M_PI_4": "0.78539816339744830962",
    "M_1_PI": "0.31830988618379067154",
    "M_2_PI": "0.63661977236758134308",
    "M_2_SQRTPI": "1.12837916709551257390",
    "M_SQRT2": "1.41421356237309504880",
    "M_SQRT1_2": "0.70710678118654752440",
}


class CInflatoxPrinter(C99CodePrinter):
    """C99 printer with the reference's symbol mapping (compiler.py:37-120).

    Field symbols print as ``x[i]``, their time derivatives as ``xdot[i]`` and every other
    symbol as ``args[k]`` where k counts parameters in order of first appearance.
    """

    def __init__(self, coordinate_symbols, coordinate_derivative_symbols, settings=None):
        super().__init__(settings)
        plain = super()._print_Symbol
        self.coord_dict = {plain(s): f"x[{i}]" for i, s in enumerate(coordinate_symbols)}
        self.dotcoord_dict = {plain(s): f"xdot[{i}]" for i, s in enumerate(coordinate_derivative_symbols)}
        self.param_dict = {}

    def _print_Symbol(self, expr):
        if expr.is_number:
            return expr.evalf(self._settings["precision"])
        known = self.get_symbol(expr)
        return known if known is not None else self.register_parameter(expr)

    def register_parameter(self, symbol):
        slot = f"args[{len(self.param_dict)}]"
        self.param_dict[super()._print_Symbol(symbol)] = slot
        return slot

    def get_symbol(self, symbol):
        name = super()._print_Symbol(symbol)
        if name.startswith("cse"):
            return name
        for table in (self.coord_dict, self.dotcoord_dict, self.param_dict):
            if name in table:
                return table[name]
        return None


class GSLInflatoxPrinter(CInflatoxPrinter):
    """Prints sympy's Bessel and hypergeometric functions the way the reference's GSL printer does
    (compiler.py:123-212): ``gsl_sf_bessel_J0(x[0])``, ``gsl_sf_hyperg_2F1(a, b, c, x)``, ...  Used to
    number parameters in the reference's print order and by the oracle's C emitter; device code is printed
    by :class:`inflatox_amd.staging.HIPInflatoxPrinter` (``inflx_sf_bessel_*``, csrc/inflx_sf.h)."""

    HYPERH = "gsl_sf_hyperg"
    BESSELH = "gsl_sf_bessel"
    # sympy class name -> (GSL letter, orders with a function of their own, name for other integer
    # orders, name for real orders or None)
    _FAMILIES = {
        "besselj": ("J", ("0", "1"), "Jn", "Jnu"),
        "bessely": ("Y", ("0", "1"), "Yn", "Ynu"),
        "besseli": ("I", ("0", "1"), "In", "Inu"),
        "besselk": ("K", ("0", "1"), "Kn", "Knu"),
        "jn": ("j", ("0", "1", "2"), "jl", None),
        "yn": ("y", ("0", "1", "2"), "yl", None),
    }
    _HYPER = {(2, 0): "2F0", (2, 1): "2F1", (1, 1): "1F1", (0, 1): "0F1"}

    def __init__(self, coordinate_symbols, coordinate_derivative_symbols, settings=None):
        super().__init__(coordinate_symbols, coordinate_derivative_symbols, settings)
        self.required_headers = []

    def update_preamble(self, header):
        if header not in self.required_headers:
            self.required_headers.append(header)

    def _bessel(self, expr):
        letter, own, integer_name, real_name = self._FAMILIES[expr.func.__name__]
        self.update_preamble(self.BESSELH)
        # the reference prints the argument with _print_Symbol, i.e. it only accepts a bare symbol there (anything
        # else raises inside sympy); a general argument is printed as an expression here
        arg = expr.args[1]
        nu, x = expr.args[0], (self._print_Symbol(arg) if arg.is_Symbol else self._print(arg))
        if nu.is_integer:
            n = int(float(self._print_Symbol(nu)))
            if str(n) in own:
                return f"gsl_sf_bessel_{letter}{n}({x})"
            return f"gsl_sf_bessel_{integer_name}({n}, {x})"
        if real_name is None:
            raise KeyError("No non-integer impl found.")
        # (the reference prints the order with _print_Symbol too, i.e. a bare symbol; the orders nu +- 1 that differentiation
        # produces are expressions and print as such here)
        return f"gsl_sf_bessel_{real_name}({self._print_Symbol(nu) if (nu.is_Symbol or nu.is_number) else self._print(nu)}, {x})"

    _print_besselj = _print_bessely = _print_besseli = _print_besselk = _print_jn = _print_yn = _bessel

    def _print_hyper(self, expr):
        self.update_preamble(self.HYPERH)
        ap, bq, x = expr.args[0], expr.args[1], self.doprint(expr.args[2])
        kind = self._HYPER.get((len(ap), len(bq)))
        if kind is None:
            raise Exception("Cannot compute hypergeometric functions other than 2F0, 2F1, 1F1 and 0F1")
        return f"gsl_sf_hyperg_{kind}(" + ", ".join([self.doprint(a) for a in list(ap) + list(bq)] + [x]) + ")"


# ---------------------------------------------------------------------------------------------
# artefact + compiler front-end
# ---------------------------------------------------------------------------------------------


class CompilationArtifact:
    """Output of :class:`Compiler` (reference compiler.py:215-276).

    ``shared_object_path`` points at the per-model gfx950 code object (the counterpart of the
    reference's per-model dylib); it is removed when the artefact is garbage-collected if
    ``auto_cleanup`` is set, exactly like the reference removes its dylib.
    """

    symbol_printer = C99CodePrinter()

    def __init__(self, symbol_dictionary, shared_object_path, n_fields, n_parameters, auto_cleanup=True, stage_info=None, header_path=None):
        self.symbol_dictionary = symbol_dictionary
        self.shared_object_path = shared_object_path
        self.n_fields = n_fields
        self.n_parameters = n_parameters
        self.auto_cleanup = auto_cleanup
        self.stage_info = stage_info or {}
        self.header_path = header_path
        self._recipe = None  # (model, Compiler keyword arguments): set by Compiler.compile, used by profile_guided
        self.kernel_groups = ALL_GROUPS  # groups inside shared_object_path (Compiler.compile sets what it built)
        self._build = None  # (header text, final hipcc options, content tag): what ensure_group compiles a further group from
        self._group_paths = {}
        _ARTEFACTS[os.path.abspath(shared_object_path)] = self

    def ensure_group(self, group: str) -> str | None:
        """Extension: make the kernel group ``group`` (a key of ``KERNEL_GROUPS``) of this model available and return the path of its
        code object -- ``shared_object_path + "." + group``, the place ``libinflx_hip.so`` looks for it -- or None when the core
        object carries it already.  Built by one hipcc step from the very header and options of the core object on first use (cached in
        the content-addressed cache like everything else); what :class:`inflatox_amd._native.InflatoxDevLib` calls before an operation
        other than complete_analysis."""
        bit = KERNEL_GROUPS[group]
        if self.kernel_groups & bit:
            return None
        path = self._group_paths.get(group)
        if path is None or not os.path.exists(path):
            if self._build is None:
                raise ValueError(f"this artefact does not know how it was compiled: kernel group {group!r} cannot be built (not made by Compiler.compile)")
            header_text, options, tag = self._build
            cached, log, code = _build_code_object(header_text, options, tag, bit)
            if code != 0:
                print(log.decode("utf-8", "replace"))
                raise Exception(f"hipcc compiler error while building kernel group {group!r} (see previous output)")
            path = self.shared_object_path + "." + group
            tmp = path + f".{os.getpid()}.tmp"
            shutil.copyfile(cached, tmp)
            os.replace(tmp, path)
            self._group_paths[group] = path
        return path

    def ensure_all_groups(self) -> list[str]:
        """Extension: every group beside the core object (what a C client of ``libinflx_hip.so`` wants in place before it starts)."""
        return [p for p in (self.ensure_group(g) for g in KERNEL_GROUPS) if p]

    def profile_guided(self, args, extent, silent: bool = True) -> "CompilationArtifact":
        """Extension: the profile-guided build of the same model -- ``Compiler(model, ..., regroup="auto", sample=(args,
        extent))`` with every other argument as this artefact was compiled with.  ``args`` / ``extent`` are the parameter
        values and the field range ``(x0_start, x0_stop, x1_start, x1_stop)`` of the intended sweeps: the transpiler
        MEASURES on them (on the host) which of the five model values may be re-associated within 1e-10 relative plus four times the
        reference form's own rounding error (Compiler docstring).  The measurement and the code object are cached in-tree,
        keyed by the model's generated code and the sample.  ``GeneralisedAL(artifact, tuned=True)`` calls this with the
        arguments of its first sweep."""
        if self._recipe is None:
            raise ValueError("this artefact does not know how it was compiled (not made by Compiler.compile)")
        model, kwargs = self._recipe
        kwargs = dict(kwargs, regroup="auto", sample=(args, extent), silent=silent, tan_shortcut=None)
        return Compiler(model, **kwargs).compile()

    def __del__(self):
        if getattr(self, "auto_cleanup", False):
            for path in [self.shared_object_path, *getattr(self, "_group_paths", {}).values()]:
                try:
                    os.remove(path)
                except OSError:
                    pass

    def lookup_symbol(self, symbol):
        name = self.symbol_printer._print_Symbol(symbol)
        if not isinstance(name, str):
            return None
        return self.symbol_dictionary[name]

    def print_sym_lookup_table(self):
        print("[Symbol Dictionary]")
        for old, new in self.symbol_dictionary.items():
            print(f"{old} -> {new}")


_COMMENT = re.compile(r'"(?:\\.|[^"\\])*"|(?<![0-9A-Za-z_])\'(?:\\.|[^\'\\])*\'|//[^\n]*|/\*.*?\*/', re.S)


def _code_only(source: str) -> str:
    """C/C++ source without comments and blank lines (string and character literals kept): what the content tag of a code object
    hashes, so that editing a comment in the kernel sources neither rebuilds every model nor orphans the profiles
    that are stamped with the tag."""
    text = _COMMENT.sub(lambda m: m.group(0) if m.group(0)[0] in "\"'" else " ", source)
    return "\n".join(ln.rstrip() for ln in text.splitlines() if ln.strip())


def _tile_kernel_scratch_bytes(code_object: str) -> int:
    """Scratch (spill) bytes per lane of inflx_sweep_tile_complete, from the code object's metadata note; a large value
    when it cannot be read (no llvm-readelf next to hipcc), so that the caller stays with the conservative build."""
    tool = os.path.join(os.path.dirname(os.path.realpath(hipcc_path())), "..", "lib", "llvm", "bin", "llvm-readelf")
    if not os.path.exists(tool):
        tool = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    try:
        notes = subprocess.run([tool, "--notes", code_object], capture_output=True, text=True, timeout=60).stdout
    except (OSError, subprocess.SubprocessError):
        return 1 << 30
    for block in notes.split("- .agpr_count:"):
        if re.search(r"\.name:\s+inflx_sweep_tile_complete\s", block):
            m = re.search(r"\.private_segment_fixed_size:\s+(\d+)", block)
            if m:
                return int(m.group(1))
    return 1 << 30


def _cache_dir() -> str:
    d = os.environ.get("INFLATOX_AMD_CACHE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "_jit_cache")
    os.makedirs(d, exist_ok=True)
    return d


def hipcc_path() -> str:
    cand = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(cand):
        raise FileNotFoundError("hipcc not found: the HIP back-end needs ROCm's hipcc to build the model code object")
    return cand


def _build_code_object(header_text: str, options: list[str], tag: str, groups: int):
    """One hipcc step: the kernels of ``groups`` for the model ``header_text`` under ``options`` into the content-addressed cache
    (``<tag>.hsaco`` for an object that holds the core group, ``<tag>.g<mask>.hsaco`` for the others; ``tag`` = SHA-256 over header,
    kernel sources and options).  Returns (path, compiler output, exit code)."""
    cache = _cache_dir()
    kernel_src = os.path.join(_CSRC, "inflx_sweep_kernels.hip")
    header_path = os.path.join(cache, f"{tag}.h")
    cached = os.path.join(cache, f"{tag}.hsaco" if groups == KERNEL_GROUPS["core"] else f"{tag}.g{groups}.hsaco")
    log, code = b"", 0
    if not os.path.exists(cached):
        # several ranks may compile the same model at once: every file appears atomically
        if not os.path.exists(header_path):
            tmp_hdr = header_path + f".{os.getpid()}.tmp"
            with open(tmp_hdr, "w") as fh:
                fh.write(header_text)
            os.replace(tmp_hdr, header_path)
        tmp_out = cached + f".{os.getpid()}.tmp"
        cmd = [hipcc_path(), *options, f"-DINFLX_KERNEL_GROUPS={groups}u", f'-DINFLX_MODEL_TAG="{tag}"', f"-I{_CSRC}", f'-DINFLX_MODEL_HEADER="{header_path}"', kernel_src, "-o", tmp_out]
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        log, code = proc.stdout, proc.returncode
        if code == 0:
            os.replace(tmp_out, cached)
    return cached, log, code


class Compiler:
    """Turns an :class:`InflationModel` into a gfx950 code object holding the sweep kernels.

    Same constructor as the reference (compiler.py:315-325) plus keyword extensions:

    * ``cse``/``max_cses`` are honoured exactly like the reference honours them: with ``cse=True``
      every function is passed through the same ``sympy.cse`` call first (this changes both the
      parameter numbering and the arithmetic the reference performs, so it is mirrored);
    * ``staged`` (default True): share identical sub-expressions and evaluate each in the stage of
      the grid axes it depends on (staging.py); ``False`` evaluates everything per grid point;
    * ``regroup`` (default False): additionally re-associate mixed products/sums so that row-only
      and column-only operands are combined before the per-point ones (faster, but rounding differs
      from the reference where a model cancels catastrophically).  ``True`` regroups all five model values; a
      collection of names out of ``"V", "v00", "v10", "v11", "g"`` (and ``"v"``, the basis vector of
      ``flag_quantum_dif``) regroups those only and keeps the others -- and every sub-expression they share with a
      regrouped one -- in the reference's arithmetic bit for bit.  ``"auto"`` (needs ``sample=(args, extent)``, the
      parameter values and field range of the intended sweeps) MEASURES which values qualify: the generated code is
      evaluated on the host on a sample of that range, the reference's form in float64 and in extended precision and
      the regrouped form in float64, and a value is regrouped only if the regrouped form differs from the reference's by
      less than 1e-10 relative plus four times the reference form's own rounding error at every sample point
      (inflatox_amd/_instrument.py; a quarter of the smallest allowance the parity suite grants).  EGNO, the doc and the
      angular model: all five values (EGNO 4096^2: 0.400 -> 0.321 ms); D5: V, v00 and |dV|^2 only -- its v10 is exact
      in the reference's form at theta = k pi/4 where cancelling terms are equal bit for bit, and no other form is;
    * ``exact_constants`` (default False): full-precision pi, e, ... instead of the reference's
      12-digit fallback constants;
    * ``hoist_reciprocals`` (default None = automatic): a per-point quotient whose denominator is known one
      stage earlier is formed from the correctly rounded reciprocal of that denominator with Markstein's
      multiply-FMA-FMA step instead of an 11-instruction IEEE division (csrc/inflx_device_math.h: the same
      correctly rounded quotient; rows with irregular operands are re-evaluated with IEEE divisions, so the
      stored values are the IEEE program's bit for bit).  4.8 fma-equivalents of issue time per quotient
      against 13.0, and nothing but the three operations when the numerator is a product of earlier-stage
      values (validity then follows from per-row / per-column / per-sweep range flags).  ``None`` turns it on
      when the quick point stage saves at least ``HOIST_MIN_GAIN`` instructions per point (D5: 29 of 53 divisions hoisted,
      0.592 -> 0.519 ms per 4096^2 sweep on MI355X); ``True`` / ``False`` force it.
    * ``share_reciprocals`` (default False): in the quick point stage, quotients with the same PER-POINT denominator (D5
      divides four times by ``r_44*p_5``) share one refined reciprocal (csrc/inflx_device_math.h).  Exact, and 32 fewer
      instructions per point for D5 -- but the additional comparisons and live values cost as much again (SGPR spills,
      three wavefronts per SIMD no longer fit): 13.9 -> 14.8 ms for D5 4096^2 x 32, hence off (profiles/r03_experiments.txt).
    * ``quick_sqrt`` (default ``None``): the point stage's square roots without the compiler's operand scaling and zero /
      infinity selection, behind one range test each (``inflx_sqrt_checked``; exact, like the hoisted reciprocals: a value
      outside the guard sends the grid row to the IEEE variant of the stage).  Eight instructions fewer per root -- and a
      second copy of the point stage plus the ``ok`` bookkeeping for models that would not have them otherwise: measured
      4096^2, doc 0.216 -> 0.218 ms, angular 0.256 -> 0.261, EGNO 0.403 -> 0.416 (148 -> 168 VGPRs), D5 0.446 -> 0.438.  ``None``
      therefore switches it on exactly where the quick point stage exists anyway (hoisted reciprocals on: D5).
    * ``tan_shortcut`` (default ``None``: 0 = off for every build in the reference's arithmetic, ``TUNED_TAN_SHORTCUT`` = 16 together
      with ``regroup="auto"``, the profile-guided build): the epilogue's ``tan(atan(t))``, t = |v10/v00|
      (src/anguelova.rs:128,132), is taken as ``t`` itself wherever ``t <= tan_shortcut``.  With 0 the kernels evaluate OCML's
      tan of OCML's atan, operation for operation -- the counterpart of the reference's two libm calls.  Those return
      t(1 + e) with |e| <~ (t + 1/t)*2^-53, so a shortcut bound T moves ``tan(delta)`` by at most ~(T + 1)*2^-53 relative --
      far inside the 1e-10 bar, and towards the exact value -- but the results are then no longer the composition bit for
      bit, which is why the default build does not take it; a wavefront all of whose points qualify skips the ~55
      instructions of the tangent (doc 4096^2: 0.237 -> 0.210 ms).

    * ``contraction`` (default ``"expression"``, since round 6): a product that is a direct operand of a sum is spelled out in the sum's
      own statement (``staging.Stager.sum_operand``), so that hipcc -- clang with ``-ffp-contract=on`` -- fuses the multiply-adds the
      reference's compiler (``zig cc`` = clang, same flag) fuses in the reference's one-expression C functions.  With ``"statement"``
      (rounds 1-5) a product the stager has made a (row / shared) variable of is added already rounded, which is what ``gcc -std=c17``
      does with the reference's C everywhere.  Measured on MI355X (profiles/r06_contraction.json): on the reference-generated golden
      grids D5's values bit-equal to the clang build rise from 11 716 to 12 696 of 19 040 (gcc build: 12 725 -> 11 925), values off by
      more than 1e-10 from the clang build fall from 259 to 38, and at all 10 + 25 values where the two builds disagree about NaN the
      GPU now sides with clang (before: with gcc at every one); angular 9 793 -> 12 038 of 17 240; EGNO (``cse=True``: the reference's
      own statements end the fusion) and doc unchanged; device time within +-2 %.
    * ``kernel_groups`` (default ``"core"``): which kernel groups ``compile()`` builds into the artefact's code object.  ``"core"``: what
      ``complete_analysis`` (grid and on-trajectory) and the basis validation need -- ONE hipcc step of 1-2.5 s (D5: 2.2 s), like the
      reference's one ``zig cc`` step; every other operation's group (``KERNEL_GROUPS``) is built when first used, 1.5-2.5 s each
      (``CompilationArtifact.ensure_group``).  ``"all"``: a complete artefact in one step (D5: 5.4 s), for a C client of
      ``libinflx_hip.so`` that wants a single file; or an iterable of group names (the core group is always included).

    ``link_gsl``: the reference links GSL for sympy's Bessel and hypergeometric functions
    (compiler.py:123-212).  Here nothing is linked: the Bessel functions (integer and real order; spherical ones of integer
    order) and 0F1, 1F1, 2F1, 2F0 are device functions of this package (csrc/inflx_sf.h).  As in the reference they print only with
    the flag (``GSLInflatoxPrinter``; without it sympy's printer refuses them), which also sets the artefact's ``USE_GSL`` global:
    where the reference then installs a GSL error handler that panics (compiler.py:145-149, src/dylib.rs:141-148), a call that
    evaluated one of these functions outside its domain fails with ``InflatoxSpecialFunctionError`` once its result is complete
    (``GeneralisedAL(..., sf_errors="nan")`` returns the arrays with NaN at those points instead; include/inflx_hip.h ``inflx_sf_policy``).
    """

    #: `hoist_reciprocals=None` turns the quick point stage (hoisted and shared reciprocals) on when it removes at least
    #: this many VALU instructions per grid point from the five sweep values (quick_point_gain)
    #: (D5: 224, on: 0.592 -> 0.519 ms per 4096^2 sweep in round 2.  EGNO: 88, off: with it 0.409 -> 0.422 ms in round 3 --
    #: its quick point stage needs 20 spilled registers to keep three wavefronts per SIMD, or falls back to two.)
    HOIST_MIN_GAIN = 100
    #: ... and a model below that bar with at least this many quotients by earlier-stage denominators gets them as self-checking
    #: quotients in its ONE point stage (``hoist_reciprocals="inline"``) -- provided the tile kernel then still holds three wavefronts
    #: per SIMD (decided on the built object by the criterion of MAX_SCRATCH_FOR_THREE_WAVES, cached).  EGNO: 12 quotients, 604 -> 557 VALU instructions per point
    #: (SQ counters), 0.419 -> 0.401 ms per 4096^2 sweep and 3.215 -> 3.057 ms per 8 parameter rows in one session, bit-identical
    #: (profiles/r06_experiments.txt section 7); doc has one such quotient, angular none: unchanged.
    INLINE_MIN_QUOTIENTS = 6
    #: ``tan_shortcut`` of a default build: off -- eta_parallel is OCML's tan of OCML's atan, like the reference's is libm's
    DEFAULT_TAN_SHORTCUT = 0
    #: ``tan_shortcut`` of the profile-guided build (``regroup="auto"``) unless the caller says otherwise: that build
    #: already trades the reference's bits for speed under a measured bound (profiles/r03_experiments.txt: the whole GPU
    #: parity suite is green with 16)
    TUNED_TAN_SHORTCUT = 16

    @staticmethod
    def quick_point_gain(info) -> int:
        """VALU instructions per grid point the quick point stage saves (estimate from the generated header)."""
        hoisted, pure = info.get("hoisted_quotients", 0), info.get("pure_quotients", 0)
        shared, groups = info.get("shared_quotients", 0), info.get("shared_reciprocals", 0)
        return 8 * pure + 7 * (hoisted - pure) + 7 * shared - 6 * groups

    #: three wavefronts per SIMD are kept when the tile kernel of complete_analysis needs at most this much scratch per lane
    MAX_SCRATCH_FOR_THREE_WAVES = 128
    c_prefix = "inflx_auto_"
    lib_prefix = "libinflx_auto_"

    default_hipcc_flags = [
        "--offload-arch=gfx950",
        "--genco",
        "--no-gpu-bundle-output",  # plain gfx950 ELF, loadable by hipModuleLoad and readable by llvm-readelf
        "-O3",
        "-std=c++17",
        "-fno-fast-math",
        # contraction only within a source statement, like the clang (zig cc) the reference compiles
        # with; hipcc's default `fast` also fuses across statements, which measurably moves results
        # away from the reference where a model cancels catastrophically (tests/tools/contract_experiment.py)
        "-ffp-contract=on",
        "-fno-gpu-rdc",
        "-Wall",
        "-Werror",
        "-Wno-unused-but-set-variable",
        "-Wno-unused-variable",
        # a model with very many row-stage values may not reach the requested occupancy: a remark, not an error
        "-Wno-error=pass-failed",
    ]

    def __init__(
        self,
        model: InflationModel,
        output_path: str | None = None,
        cleanup: bool = True,
        silent: bool = False,
        link_gsl: bool = False,
        cse: bool = False,
        max_cses: int = 1000,
        compiler_flags: list[str] | None = None,
        staged: bool = True,
        exact_constants: bool = False,
        regroup: bool | None = None,
        hoist_reciprocals: bool | None = None,
        tan_shortcut: float | None = None,
        share_reciprocals: bool = False,
        sample=None,
        quick_sqrt: bool | None = None,
        kernel_groups="core",
        contraction: str = "expression",
    ):
        # what CompilationArtifact.profile_guided needs to compile the same model again with a measured re-association
        self._init_kwargs = dict(output_path=None, cleanup=cleanup, silent=silent, link_gsl=link_gsl, cse=cse, max_cses=max_cses, compiler_flags=compiler_flags,
                                 staged=staged, exact_constants=exact_constants, hoist_reciprocals=hoist_reciprocals, share_reciprocals=share_reciprocals, quick_sqrt=quick_sqrt,
                                 kernel_groups=kernel_groups, contraction=contraction)
        # link_gsl: nothing is linked here -- the Bessel functions the reference takes from GSL are device
        # functions of this package (csrc/inflx_sf.h, integer orders); the flag is recorded in USE_GSL
        self.gsl = bool(link_gsl)
        if contraction not in ("statement", "expression"):
            raise ValueError('contraction must be "statement" or "expression"')
        self.contraction = contraction
        if isinstance(kernel_groups, str):
            if kernel_groups not in ("core", "all"):
                raise ValueError('kernel_groups must be "core", "all" or an iterable of group names')
            self.kernel_groups = ALL_GROUPS if kernel_groups == "all" else KERNEL_GROUPS["core"]
        else:
            unknown = [g for g in kernel_groups if g not in KERNEL_GROUPS]
            if unknown:
                raise ValueError(f"kernel_groups: unknown group(s) {unknown}; choose from {sorted(KERNEL_GROUPS)}")
            self.kernel_groups = KERNEL_GROUPS["core"]
            for g in kernel_groups:
                self.kernel_groups |= KERNEL_GROUPS[g]
        if model.dim != 2:
            raise Exception("the HIP sweep back-end supports two-field models only")
        self.symbolic_out = model
        self.output_path = output_path
        self.cleanup = cleanup
        self.silent = silent
        self.cse = cse
        self.max_cses = max_cses
        self.staged = staged
        # (no environment switches in library code: what a default Compiler() computes depends on its arguments only;
        # the experiment switches INFLX_REGROUP / INFLX_TAN_SHORTCUT live in workloads.artifact_for, test infrastructure)
        if regroup is None:
            regroup = False
        self.sample = None
        if isinstance(regroup, str) and regroup == "auto":
            if sample is None:
                raise ValueError('regroup="auto" needs sample=(args, (x0_start, x0_stop, x1_start, x1_stop)): the parameter values and the field range the decision is measured on')
            args_, extent_ = sample
            self.sample = (np.ascontiguousarray(args_, dtype=np.float64).reshape(-1), tuple(float(v) for v in extent_))
            if len(self.sample[1]) != 4:
                raise ValueError("sample extent must be (x0_start, x0_stop, x1_start, x1_stop)")
            self.regroup = "auto"  # resolved to a set of model values in _generate_hip_header
        elif isinstance(regroup, (bool, int)):
            self.regroup = bool(regroup)
        else:
            names = {"V": 0, "v00": 1, "v10": 2, "v11": 3, "g": 4, "v": 5}
            unknown = [n for n in regroup if n not in names]
            if unknown:
                raise ValueError(f"regroup: unknown model value(s) {unknown}; choose from {sorted(names)}")
            self.regroup = frozenset(names[n] for n in regroup)
        self.hoist_reciprocals = hoist_reciprocals
        self.share_reciprocals = bool(share_reciprocals)
        self.quick_sqrt = quick_sqrt  # None: wherever the quick point stage exists anyway (see the class docstring)
        if tan_shortcut is None:
            tan_shortcut = self.TUNED_TAN_SHORTCUT if self.regroup == "auto" else self.DEFAULT_TAN_SHORTCUT
        if tan_shortcut < 0 or tan_shortcut != int(tan_shortcut) or tan_shortcut >= 2**17:
            raise ValueError("tan_shortcut must be a whole number in [0, 2^17): the largest |v10/v00| for which tan(atan(t)) is taken as t")
        self.tan_shortcut = int(tan_shortcut)
        self.constants = dict(_EXACT_CONSTANTS if exact_constants else _REFERENCE_CONSTANTS)
        self.hipcc_opts = list(compiler_flags) if compiler_flags is not None else list(self.default_hipcc_flags)
        self.symbol_dict = None
        self.stage_info = None

    # -- parameter numbering, identical to the reference's emission order ----------------------
    def _cse_symbols(self):
        k = 0
        while k <= self.max_cses:
            yield sympy.symbols(f"cse{k}")
            k += 1
        raise Exception("Maximum number of common subexpressions reached!")

    def _register(self, printer, expr_or_list):
        """Print like the reference would (registering parameters as a side effect)."""
        is_list = isinstance(expr_or_list, (list, tuple))
        if self.cse:
            if is_list:
                repl, red = sympy.cse(list(expr_or_list), symbols=self._cse_symbols(), list=True)
            else:
                repl, red = sympy.cse(expr_or_list, symbols=self._cse_symbols(), order="none", list=False)
                red = [red]
            for s, d in repl:
                printer.doprint(s)
                printer.doprint(d)
            for r in red:
                printer.doprint(r)
        else:
            for e in expr_or_list if is_list else [expr_or_list]:
                printer.doprint(e)

    def _number_parameters(self):
        """Reproduce compiler.py:474-539's print order far enough to number every parameter."""
        m = self.symbolic_out
        pr = (GSLInflatoxPrinter if self.gsl else CInflatoxPrinter)(m.coordinates, m.coordinate_tangents)
        plain = C99CodePrinter()._print_Symbol
        coords = set(m.coordinates) | set(m.coordinate_tangents)

        def pending(exprs):
            want = set()
            for e in exprs:
                want |= {s for s in sympy.sympify(e).free_symbols if s not in coords}
            return {s for s in want if plain(s) not in pr.param_dict}

        dim = m.dim
        flat_metric = [sympy.sympify(m.metric[i][j]) for i in range(dim) for j in range(dim)]
        hesse = [m.hesse_cmp[a][b] for a in range(dim) for b in range(dim)]
        later = hesse + [c for vec in m.basis for c in vec] + [m.gradient_square] + list(m.eom_fields or []) + [m.eom_h, m.eom_hdot]
        later = [e for e in later if e is not None]
        # V first (compiler.py:490), then the metric inside inner_prod (:495)
        self._register(pr, sympy.sympify(m.potential))
        self._register(pr, flat_metric)
        if pending(later):
            # a parameter that appears in neither V nor the metric: fall back to printing
            # everything in the reference's order
            for a in range(dim):
                for b in range(dim):
                    self._register(pr, sympy.sympify(m.hesse_cmp[a][b]))
            for vec in m.basis:
                self._register(pr, [sympy.sympify(c) for c in vec])
            self._register(pr, sympy.sympify(m.gradient_square))
            for e in list(m.eom_fields or []) + [m.eom_h, m.eom_hdot]:
                if e is not None:
                    self._register(pr, sympy.sympify(e))
        symbol_dict = dict(pr.coord_dict)
        symbol_dict.update(pr.param_dict)
        return symbol_dict, dict(pr.param_dict)

    # -- code generation ------------------------------------------------------------------------
    def _generate_hip_header(self) -> str:
        self.symbol_dict, params = self._number_parameters()
        if not self.silent and self.cse:
            print("Converting sympy to HIP using common subexpression elimination...")
        cse = cse_vector = None
        if self.cse:
            # exactly the reference's per-function calls (compiler.py:403-404 and :425-426)
            def cse(expr):
                return sympy.cse(expr, symbols=self._cse_symbols(), order="none", list=False)

            def cse_vector(vector):
                return sympy.cse(list(vector), symbols=self._cse_symbols(), list=True)

        if isinstance(self.regroup, str):  # "auto": measured on the sample, inflatox_amd/_instrument.py
            from . import _instrument

            def header_for(regroup):
                saved, saved_info = self.regroup, self.stage_info
                self.regroup = regroup
                try:
                    return self._generate_hip_header()
                finally:
                    self.regroup, self.stage_info = saved, saved_info

            log = None if self.silent else (lambda msg: print("[regroup=auto] " + msg))
            # the measurement takes a few host compiles: its outcome is cached next to the code objects, keyed by the
            # reference form's header, the sample and the instrument's criterion
            key = hashlib.sha256()
            key.update(header_for(False).encode())
            # ... and the regrouped form that is measured against it (a change to the re-association logic of staging.py
            # or to the options that shape the point stage must not find a stale decision)
            key.update(header_for(frozenset(range(5))).encode())
            key.update(self.sample[0].tobytes() + np.asarray(self.sample[1], dtype=np.float64).tobytes())
            key.update(repr((_instrument.RTOL, _instrument.C_ERR, _instrument.COPIES)).encode())
            with open(_instrument.__file__, "rb") as fh:
                key.update(fh.read())
            decision_path = os.path.join(_cache_dir(), f"{key.hexdigest()[:20]}.regroup")
            if os.path.exists(decision_path):
                chosen = [int(t) for t in open(decision_path).read().split()]
                if log:
                    log("cached decision: " + ", ".join(_instrument.NAMES[k] for k in chosen))
            else:
                chosen = sorted(_instrument.choose_regroup(header_for, self.sample[0], self.sample[1], log=log))
                if _instrument.host_compiler() is not None:  # "no information" outcomes are not cached
                    tmp = decision_path + f".{os.getpid()}.tmp"
                    with open(tmp, "w") as fh:
                        fh.write(" ".join(str(k) for k in chosen) + "\n")
                    os.replace(tmp, decision_path)
            self.regroup = frozenset(chosen)
            self.regrouped_values = tuple(_instrument.NAMES[k] for k in sorted(chosen))
            self.symbol_dict, params = self._number_parameters()

        def emit(hoist):
            return emit_stage_header(
                self.symbolic_out,
                params,
                self.constants,
                self.symbolic_out.model_name,
                __version__,
                __abi_version__,
                staged=self.staged,
                cse=cse,
                cse_vector=cse_vector,
                regroup=self.regroup,
                hoist_reciprocals=hoist,
                share_point_reciprocals=self.share_reciprocals,
                quick_sqrt=(hoist is True or hoist == 1) if self.quick_sqrt is None else bool(self.quick_sqrt),
                contract_products=self.contraction == "expression",
            )

        if self.hoist_reciprocals is None:
            # automatic: the second copy of the point stage and the bookkeeping around it only pay when enough
            # instructions leave the per-point stage: an IEEE division is 11, a quotient by a hoisted reciprocal 3 (+1
            # comparison unless its numerator is made of earlier-stage values), one by a shared per-point reciprocal 3 + 1
            # and the 5 + 1 of the reciprocal once per group
            text, info = emit(self.staged)
            self._inline_fallback = None
            if self.quick_point_gain(info) < self.HOIST_MIN_GAIN:
                text, info = emit(False)
                if self.staged:
                    itext, iinfo = emit("inline")
                    if iinfo.get("inline_quotients", 0) >= self.INLINE_MIN_QUOTIENTS and self._inline_verdict(itext) != "no":
                        # (the plain form stays at hand: compile() falls back to it when the built kernel spills)
                        self._inline_fallback = (text, info)
                        text, info = itext, iinfo
        else:
            text, info = emit(self.hoist_reciprocals)
        value_names = ("V", "v00", "v10", "v11", "g")
        info["regrouped"] = (list(value_names) if self.regroup else []) if isinstance(self.regroup, bool) else [value_names[k] for k in sorted(self.regroup) if k < 5]
        self.stage_info = info
        return text

    def _inline_verdict_path(self, inline_header: str) -> str:
        return os.path.join(_cache_dir(), hashlib.sha256(("inline-verdict:" + " ".join(self.hipcc_opts) + inline_header).encode()).hexdigest()[:20] + ".inline")

    def _inline_verdict(self, inline_header: str):
        """"no": this model's self-checking-quotient build lost the third wavefront per SIMD (see compile()); "yes" / None: fine / not built yet."""
        path = self._inline_verdict_path(inline_header)
        return open(path).read().strip() if os.path.exists(path) else None

    def _hipcc_compile(self, header_text: str):
        kernel_src = os.path.join(_CSRC, "inflx_sweep_kernels.hip")
        deps = [kernel_src] + [os.path.join(_CSRC, f) for f in ("inflx_ops.h", "inflx_device_math.h", "inflx_kernel_abi.h")]
        # the special-function header is part of a code object only when the model calls into it (every function in it is an
        # inline device function: unreferenced ones leave no trace in the binary)
        if "inflx_sf_" in header_text:
            deps += [os.path.join(_CSRC, f) for f in ("inflx_sf.h", "inflx_sf_tables.h")]
        h = hashlib.sha256()
        h.update(header_text.encode())
        for d in deps:
            with open(d, "r", encoding="utf-8") as fh:
                h.update(_code_only(fh.read()).encode())
        opts = list(self.hipcc_opts)
        if self.gsl:
            opts.append("-DINFLX_USE_GSL=1")
        # the tile kernels keep TILE_ROWS x kNRs doubles of row-stage values in LDS (kNRs = n_row rounded up to even, the
        # stride the kernels use) beside their fixed LDS: the 12 KiB transpose buffers, the 34 polynomial coefficients of
        # the epilogue and, for models with more than 8 parameter-only values, those values.  Shrink the tile for models
        # with very many row values so that all of it stays within the 64 KiB of static LDS a kernel may declare
        info = self.stage_info or {}
        n_row = (max(1, info.get("nr", 1)) + 1) & ~1
        n_uniform = max(1, info.get("nu", 1))
        fixed_lds = 4 * 64 * 6 * 8 + 34 * 8 + (n_uniform * 8 if n_uniform > 8 else 0)
        if not any(o.startswith("-DINFLX_TILE_ROWS") for o in opts):
            rows = 32
            while rows > 1 and rows * n_row * 8 + fixed_lds > 64 * 1024:
                rows //= 2
            if rows != 32:
                opts.append(f"-DINFLX_TILE_ROWS={rows}")
        if self.tan_shortcut and not any(o.startswith("-DINFLX_TAN_SHORTCUT_MAX") for o in opts):
            opts.append(f"-DINFLX_TAN_SHORTCUT_MAX={self.tan_shortcut}")
        cache = _cache_dir()

        def build(options):
            hh = h.copy()
            hh.update(" ".join(options).encode())
            tag = hh.hexdigest()[:20]
            # what compile() builds: the core group (one hipcc step) unless the caller asked for more in the same object
            cached, log, code = _build_code_object(header_text, options, tag, self.kernel_groups)
            return cached, os.path.join(cache, f"{tag}.h"), log, code, list(options), tag

        if any(o.startswith("-DINFLX_MIN_WAVES") for o in opts):
            return build(opts)
        # Occupancy of the tile kernels.  They are bound by FP64 VALU issue, and two wavefronts per SIMD leave the pipe idle
        # ~14 % of the time (SQ counters, D5); three fill it.  Three need <= 168 vector registers: the light models are
        # below that anyway, a heavy one (D5: 194) is asked to fit (__launch_bounds__(256, 3)) and accepted when the
        # register allocator gets there with a handful of spilled values (D5: 13 registers, 3 scratch accesses per grid
        # row, 15.1 -> 14.3 ms for 4096^2 x 32); a model that would spill in earnest keeps two (the choice is cached).
        # The decision is read off the object compile() builds anyway (it holds inflx_sweep_tile_complete); the groups built later
        # follow it.
        three = opts + ["-DINFLX_MIN_WAVES=3"]
        hh = h.copy()
        hh.update(" ".join(three).encode())
        verdict_path = os.path.join(cache, f"{hh.hexdigest()[:20]}.waves")
        verdict = open(verdict_path).read().strip() if os.path.exists(verdict_path) else None
        if verdict != "2":
            result = build(three)
            if result[3] == 0 and verdict is None:
                verdict = "3" if _tile_kernel_scratch_bytes(result[0]) <= self.MAX_SCRATCH_FOR_THREE_WAVES else "2"
                tmp = verdict_path + f".{os.getpid()}.tmp"
                with open(tmp, "w") as fh:
                    fh.write(verdict + "\n")
                os.replace(tmp, verdict_path)
            if verdict == "3" or result[3] != 0:
                return result
        return build(opts + ["-DINFLX_MIN_WAVES=2"])

    def compile(self) -> CompilationArtifact:
        if not self.silent:
            print("Compiling model...")
        header = self._generate_hip_header()
        cached, header_path, log, code, options, tag = self._hipcc_compile(header)
        fallback = getattr(self, "_inline_fallback", None)
        if code == 0 and fallback is not None and self._inline_verdict(header) is None:
            # the automatic choice of self-checking quotients stands only if the tile kernel keeps three wavefronts per SIMD (the
            # quotients hold denominator AND reciprocal in registers: EGNO 138 -> 168 VGPRs and 36 B of scratch, measured faster all
            # the same); a model it would push down to two wavefronts gets the plain point stage instead
            fits = "-DINFLX_MIN_WAVES=3" in options
            tmp = self._inline_verdict_path(header) + f".{os.getpid()}.tmp"
            with open(tmp, "w") as fh:
                fh.write(("yes" if fits else "no") + "\n")
            os.replace(tmp, self._inline_verdict_path(header))
            if not fits:
                header, self.stage_info = fallback
                self._inline_fallback = None
                cached, header_path, log, code, options, tag = self._hipcc_compile(header)
        if code != 0:
            print(log.decode("utf-8", "replace"))
            print(f'Problematic source file located at: "{header_path}"')
            raise Exception("hipcc compiler error (see previous output)")
        if not self.silent and log:
            print(log.decode("utf-8", "replace"), end="")
        # like the reference, `output_path` names the generated *source* (kept unless cleanup) ...
        if self.output_path is not None and not self.cleanup:
            with open(self.output_path, "w") as fh:
                fh.write(header)
        # ... and every artefact owns its own binary in the temp dir (compiler.py:568-572), which
        # CompilationArtifact.__del__ removes; the content-addressed cache keeps the master copy.
        fd, out_path = tempfile.mkstemp(prefix=self.lib_prefix, suffix=".hsaco")
        os.close(fd)
        shutil.copyfile(cached, out_path)
        art = CompilationArtifact(
            self.symbol_dict,
            out_path,
            self.symbolic_out.dim,
            len(self.symbol_dict) - self.symbolic_out.dim,
            auto_cleanup=self.cleanup,
            stage_info=self.stage_info,
            header_path=header_path,
        )
        art._recipe = (self.symbolic_out, dict(self._init_kwargs))
        art.kernel_groups = self.kernel_groups
        art._build = (header, options, tag)
        return art
