"""Transpiler: sympy expressions -> HIP device code -> gfx950 code object.

Drop-in for the reference's ``Compiler`` / ``CompilationArtifact`` / ``CInflatoxPrinter``
(python/inflatox/compiler.py:37-120, 215-276, 279-650) with a different back-end: instead
of a C file with one ``double f(const double x[], const double args[])`` per quantity that the
native sweep calls through ``dlsym``'d pointers (five indirect calls per grid point), this
module emits ONE header of ``__device__ __forceinline__`` functions that is compiled together
with the hand-written sweep kernels (``csrc/inflx_sweep_kernels.hip``) into a per-model
gfx950 code object.  What that buys (none of it available across the reference's dylib
boundary):

  * cross-function common-subexpression elimination over V, v00, v10, v11 and |dV|^2;
  * *axis staging*: every sub-expression is classified by which grid axis it depends on --
    nothing but parameters (U), x[0] only (R, "row"), x[1] only (C, "column"), or both (P,
    "point") -- and is evaluated once per kernel / per row / per column / per point
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
import shutil
import subprocess
import sys
import tempfile
from collections import defaultdict

import sympy
from sympy.printing.c import C99CodePrinter

from .symbolic import InflationModel
from .version import __abi_version__, __version__

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")

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


class HIPInflatoxPrinter(C99CodePrinter):
    """Prints staged expressions as HIP device code.

    Symbols are resolved through an explicit name table (fields -> ``x0``/``x1``, parameters ->
    ``args[k]``, stage temporaries -> their C identifiers); an unknown symbol is an error here,
    because parameter numbering is fixed beforehand by the reference-order registration pass.
    """

    MAX_INT_POW = 64

    def __init__(self, names: dict, constants: dict):
        super().__init__()
        self.names = names
        # constants print as INFLX_<macro>, defined in the generated header
        self.math_macros = {k: "INFLX_" + v for k, v in self.math_macros.items()}
        self.constants = constants

    def _print_Symbol(self, expr):
        try:
            return self.names[expr]
        except KeyError:
            raise KeyError(f"symbol {expr!r} was not registered before printing") from None

    def _print_Integer(self, expr):
        v = int(expr)
        return str(v) if abs(v) < 2**31 else f"{v}.0"

    def _print_Pow(self, expr):
        base, exp = expr.base, expr.exp
        if exp.is_Integer:
            n = int(exp)
            if 2 <= abs(n) <= self.MAX_INT_POW:
                body = f"inflx_ipow<{abs(n)}>({self._print(base)})"
                return body if n > 0 else f"(1.0/{body})"
        elif exp.is_Rational and exp.q == 2:
            n = int(exp.p)
            if 3 <= abs(n) <= 2 * self.MAX_INT_POW:
                body = f"inflx_hpow<{abs(n)}>({self._print(base)})"
                return body if n > 0 else f"(1.0/{body})"
        return super()._print_Pow(expr)

    # functions without a C99 spelling
    def _print_coth(self, e):
        return f"inflx_coth({self._print(e.args[0])})"

    def _print_sech(self, e):
        return f"inflx_sech({self._print(e.args[0])})"

    def _print_csch(self, e):
        return f"inflx_csch({self._print(e.args[0])})"

    def _print_cot(self, e):
        return f"inflx_cot({self._print(e.args[0])})"

    def _print_sec(self, e):
        return f"inflx_sec({self._print(e.args[0])})"

    def _print_csc(self, e):
        return f"inflx_csc({self._print(e.args[0])})"


# ---------------------------------------------------------------------------------------------
# axis staging
# ---------------------------------------------------------------------------------------------

# dependence masks: bit 0 = depends on x[0] (row axis), bit 1 = depends on x[1] (column axis)
_U, _R, _C, _P = 0, 1, 2, 3
_STAGE_PREFIX = {_U: "u", _R: "r", _C: "c", _P: "p"}


class StagedProgram:
    """The five model quantities as four straight-line programs (U, R, C, P stages).

    ``defs[m]``     ordered list of (symbol, expression) evaluated in stage m;
    ``outputs``     five expressions/symbols for V, v00, v10, v11, |dV|^2 (each an atom or stage symbol);
    ``out_mask``    OR of the dependence masks of the five outputs;
    ``exports[m]``  symbols of stage m that a later stage reads (these cross LDS/registers).
    """

    def __init__(self, exprs, x0, x1):
        self.x0, self.x1 = x0, x1
        self.defs = {_U: [], _R: [], _C: [], _P: []}
        self._mask_of_symbol = {x0: _R, x1: _C}
        self._stage_of_symbol = {}
        self._memo = {_U: {}, _R: {}, _C: {}, _P: {}}
        self._mask_cache = {}
        self._counter = defaultdict(int)
        self._rename = {}  # sympy.cse temporaries -> stage symbols

        sys.setrecursionlimit(max(sys.getrecursionlimit(), 20000))
        replacements, reduced = sympy.cse(list(exprs), symbols=sympy.numbered_symbols("_inflx_cse"), order="canonical")
        for sym, definition in replacements:
            definition = definition.xreplace(self._rename)
            m = self._mask(definition)
            new_sym = self._new_symbol(m)
            self._rename[sym] = new_sym
            self.defs[m].append((new_sym, self._lower(definition, m)))
        self.outputs = []
        self.out_masks = []
        for e in reduced:
            e = e.xreplace(self._rename)
            m = self._mask(e)
            self.out_masks.append(m)
            self.outputs.append(self._as_atom(e, m))
        self.out_mask = 0
        for m in self.out_masks:
            self.out_mask |= m
        self._compute_exports()

    # -- helpers ------------------------------------------------------------------------------
    def _new_symbol(self, m):
        k = self._counter[m]
        self._counter[m] += 1
        s = sympy.Symbol(f"{_STAGE_PREFIX[m]}_{k}", real=True)
        self._stage_of_symbol[s] = m
        self._mask_of_symbol[s] = m
        return s

    def _mask(self, e):
        """Which grid axes does ``e`` depend on?"""
        if e.is_Symbol:
            return self._mask_of_symbol.get(e, _U)
        if e.is_Atom:
            return _U
        got = self._mask_cache.get(e)
        if got is None:
            got = 0
            for a in e.args:
                got |= self._mask(a)
                if got == _P:
                    break
            self._mask_cache[e] = got
        return got

    def _as_atom(self, e, m):
        """Return a symbol/number standing for ``e`` (class m), defining a stage variable if needed."""
        if e.is_Atom:
            return e
        if not e.free_symbols:
            return e  # pure number: the device compiler folds it
        hit = self._memo[m].get(e)
        if hit is not None:
            return hit
        sym = self._new_symbol(m)
        self._memo[m][e] = sym
        self.defs[m].append((sym, self._lower(e, m)))
        return sym

    def _lower(self, e, ctx):
        """Rewrite ``e`` (evaluated in stage ctx): maximal sub-trees that depend on fewer axes
        than ctx are moved to their own stage and replaced by that stage's symbol."""
        if e.is_Atom:
            return e
        m = self._mask(e)
        if m != ctx:
            return self._as_atom(e, m)
        if e.is_Add or e.is_Mul:
            groups = defaultdict(list)
            for a in e.args:
                groups[self._mask(a)].append(a)
            if len(groups) == 1:
                return e.func(*[self._lower(a, ctx) for a in e.args])
            parts = []
            for gm, items in groups.items():
                if gm == ctx:
                    parts.extend(self._lower(a, ctx) for a in items)
                else:
                    sub = e.func(*items) if len(items) > 1 else items[0]
                    parts.append(self._as_atom(sub, gm))
            return e.func(*parts)
        return e.func(*[self._lower(a, ctx) for a in e.args])

    def _compute_exports(self):
        used_by = defaultdict(set)
        for m, lst in self.defs.items():
            for _, d in lst:
                for s in d.free_symbols:
                    sm = self._stage_of_symbol.get(s)
                    if sm is not None and sm != m:
                        used_by[s].add(m)
        for o in self.outputs:
            if o.is_Symbol and o in self._stage_of_symbol:
                used_by[o].add("out")
        self.exports = {m: [s for s, _ in self.defs[m] if s in used_by] for m in (_U, _R, _C)}
        self.imports = defaultdict(list)  # stage (or "out") -> exported symbols it reads
        for m in (_U, _R, _C):
            for s in self.exports[m]:
                for user in used_by[s]:
                    self.imports[user].append(s)

    def op_count(self):
        return {m: sum(int(sympy.count_ops(d)) for _, d in lst) for m, lst in self.defs.items()}


def _emit_stage_header(model: InflationModel, param_slots: dict, constants: dict, model_name: str, staged: bool = True) -> tuple[str, dict]:
    """Return (header text, info dict) for the model."""
    x0, x1 = model.coordinates
    exprs = [
        sympy.sympify(model.potential),
        sympy.sympify(model.hesse_cmp[0][0]),
        sympy.sympify(model.hesse_cmp[1][0]),
        sympy.sympify(model.hesse_cmp[1][1]),
        sympy.sympify(model.gradient_square),
    ]
    tangents = set(model.coordinate_tangents)
    for e in exprs:
        if e.free_symbols & tangents:
            raise Exception("potential / Hesse expressions may not depend on field velocities")
    if staged:
        prog = StagedProgram(exprs, x0, x1)
    else:
        prog = _unstaged_program(exprs, x0, x1)

    plain = C99CodePrinter()._print_Symbol
    names = {x0: "x0", x1: "x1"}
    for s in set().union(*[e.free_symbols for e in exprs]) - {x0, x1}:
        names[s] = param_slots[plain(s)]
    for m, lst in prog.defs.items():
        for s, _ in lst:
            names[s] = s.name
    pr = HIPInflatoxPrinter(names, constants)

    idx = {m: {s: k for k, s in enumerate(prog.exports[m])} for m in (_U, _R, _C)}
    arr = {_U: "U", _R: "R", _C: "C"}

    def imports_for(user):
        lines = []
        for s in prog.imports.get(user, []):
            m = prog._stage_of_symbol[s]
            lines.append(f"  const double {s.name} = {arr[m]}[{idx[m][s]}];")
        return lines

    def body(m):
        lines = imports_for(m)
        for s, d in prog.defs[m]:
            lines.append(f"  const double {s.name} = {pr.doprint(d)};")
        if m in idx:
            for s, k in idx[m].items():
                lines.append(f"  {arr[m]}[{k}] = {s.name};")
        return "\n".join(lines)

    n_par = len(param_slots)
    nu, nr, nc = (len(prog.exports[m]) for m in (_U, _R, _C))
    ops = prog.op_count()
    out = []
    out.append("// Generated by inflatox_amd.Compiler -- do not edit.")
    out.append(f"// model: {model_name}; inflatox_amd v{__version__}; ABI v{__abi_version__}")
    out.append("#pragma once")
    for k, v in constants.items():
        out.append(f"#define INFLX_{k} {v}")
    out.append(f"#define INFLX_N_PARAMETERS {n_par}")
    out.append(f"#define INFLX_DIM {model.dim}")
    out.append(f'#define INFLX_MODEL_NAME "{model_name}"')
    out.append(f"#define INFLX_NU {nu}")
    out.append(f"#define INFLX_NR {nr}")
    out.append(f"#define INFLX_NC {nc}")
    out.append(f"#define INFLX_OUT_MASK {prog.out_mask}")
    out.append(f"// sympy op counts per stage: U={ops[_U]} R={ops[_R]} C={ops[_C]} P={ops[_P]}")
    out.append("")
    sig_tail = "[[maybe_unused]] const double* __restrict__ args"
    out.append("// parameter-only sub-expressions (wave-uniform)")
    out.append(f"__device__ __forceinline__ void inflx_stage_uniform({sig_tail}, [[maybe_unused]] double* __restrict__ U) {{")
    out.append(body(_U))
    out.append("}\n")
    out.append("// sub-expressions of x[0] (and parameters): once per grid row")
    out.append(
        f"__device__ __forceinline__ void inflx_stage_row([[maybe_unused]] const double x0, {sig_tail}, "
        "[[maybe_unused]] const double* __restrict__ U, [[maybe_unused]] double* __restrict__ R) {"
    )
    out.append(body(_R))
    out.append("}\n")
    out.append("// sub-expressions of x[1] (and parameters): once per grid column")
    out.append(
        f"__device__ __forceinline__ void inflx_stage_col([[maybe_unused]] const double x1, {sig_tail}, "
        "[[maybe_unused]] const double* __restrict__ U, [[maybe_unused]] double* __restrict__ C) {"
    )
    out.append(body(_C))
    out.append("}\n")
    out.append("// everything that depends on both axes, and the five model values")
    out.append(
        f"__device__ __forceinline__ void inflx_stage_point([[maybe_unused]] const double x0, [[maybe_unused]] const double x1, {sig_tail}, "
        "[[maybe_unused]] const double* __restrict__ U, [[maybe_unused]] const double* __restrict__ R, "
        "[[maybe_unused]] const double* __restrict__ C, InflxModelValues& mv) {"
    )
    lines = imports_for(_P)
    have = {s for s in prog.imports.get(_P, [])}
    for s in prog.imports.get("out", []):
        if s not in have:
            m = prog._stage_of_symbol[s]
            lines.append(f"  const double {s.name} = {arr[m]}[{idx[m][s]}];")
            have.add(s)
    for s, d in prog.defs[_P]:
        lines.append(f"  const double {s.name} = {pr.doprint(d)};")
    for field, o in zip(("V", "v00", "v10", "v11", "g"), prog.outputs):
        lines.append(f"  mv.{field} = {pr.doprint(o)};")
    out.append("\n".join(lines))
    out.append("}\n")
    info = dict(nu=nu, nr=nr, nc=nc, out_mask=prog.out_mask, out_masks=list(prog.out_masks), ops={str(k): v for k, v in ops.items()})
    return "\n".join(out), info


class _UnstagedProgram:
    pass


def _unstaged_program(exprs, x0, x1):
    """Debug/parity switch: no CSE, no staging -- the five expressions are printed as they are
    (like the reference's five separate C functions) and evaluated per grid point."""
    prog = _UnstagedProgram()
    prog.defs = {_U: [], _R: [], _C: [], _P: []}
    prog.outputs = list(exprs)
    prog.out_masks = [_P] * 5
    prog.out_mask = _P
    prog.exports = {_U: [], _R: [], _C: []}
    prog.imports = defaultdict(list)
    prog._stage_of_symbol = {}
    prog.op_count = lambda: {_U: 0, _R: 0, _C: 0, _P: sum(int(sympy.count_ops(e)) for e in exprs)}
    return prog


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

    def __del__(self):
        if getattr(self, "auto_cleanup", False):
            try:
                os.remove(self.shared_object_path)
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


def _cache_dir() -> str:
    d = os.environ.get("INFLATOX_AMD_CACHE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "_jit_cache")
    os.makedirs(d, exist_ok=True)
    return d


def hipcc_path() -> str:
    cand = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(cand):
        raise FileNotFoundError("hipcc not found: the HIP back-end needs ROCm's hipcc to build the model code object")
    return cand


class Compiler:
    """Turns an :class:`InflationModel` into a gfx950 code object holding the sweep kernels.

    Same constructor as the reference (compiler.py:315-325).  ``cse``/``max_cses`` only affect
    parameter numbering (they change the reference's print order, hence must be mirrored);
    device code is always jointly CSE'd and axis-staged unless ``staged=False``.
    """

    c_prefix = "inflx_auto_"
    lib_prefix = "libinflx_auto_"

    default_hipcc_flags = [
        "--offload-arch=gfx950",
        "--genco",
        "-O3",
        "-std=c++17",
        "-fno-fast-math",
        "-fno-gpu-rdc",
        "-Wall",
        "-Werror",
        "-Wno-unused-but-set-variable",
        "-Wno-unused-variable",
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
    ):
        if link_gsl:
            raise NotImplementedError("GSL special functions have no device implementation; link_gsl is not supported by the HIP back-end")
        if model.dim != 2:
            raise Exception("the HIP sweep back-end supports two-field models only")
        self.symbolic_out = model
        self.output_path = output_path
        self.cleanup = cleanup
        self.silent = silent
        self.cse = cse
        self.max_cses = max_cses
        self.staged = staged
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
        pr = CInflatoxPrinter(m.coordinates, m.coordinate_tangents)
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
        text, info = _emit_stage_header(self.symbolic_out, params, self.constants, self.symbolic_out.model_name, staged=self.staged)
        self.stage_info = info
        return text

    def _hipcc_compile(self, header_text: str):
        kernel_src = os.path.join(_CSRC, "inflx_sweep_kernels.hip")
        deps = [kernel_src, os.path.join(_CSRC, "inflx_ops.h"), os.path.join(_CSRC, "inflx_device_math.h"), os.path.join(_CSRC, "inflx_kernel_abi.h")]
        h = hashlib.sha256()
        h.update(header_text.encode())
        for d in deps:
            with open(d, "rb") as fh:
                h.update(fh.read())
        h.update(" ".join(self.hipcc_opts).encode())
        tag = h.hexdigest()[:20]
        cache = _cache_dir()
        cached = os.path.join(cache, f"{tag}.hsaco")
        header_path = os.path.join(cache, f"{tag}.h")
        log = b""
        code = 0
        if not os.path.exists(cached):
            with open(header_path, "w") as fh:
                fh.write(header_text)
            tmp_out = cached + f".{os.getpid()}.tmp"
            cmd = [hipcc_path(), *self.hipcc_opts, f"-I{_CSRC}", f'-DINFLX_MODEL_HEADER="{header_path}"', kernel_src, "-o", tmp_out]
            proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            log, code = proc.stdout, proc.returncode
            if code == 0:
                os.replace(tmp_out, cached)
        return cached, header_path, log, code

    def compile(self) -> CompilationArtifact:
        if not self.silent:
            print("Compiling model...")
        header = self._generate_hip_header()
        cached, header_path, log, code = self._hipcc_compile(header)
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
        return CompilationArtifact(
            self.symbol_dict,
            out_path,
            self.symbolic_out.dim,
            len(self.symbol_dict) - self.symbolic_out.dim,
            auto_cleanup=self.cleanup,
            stage_info=self.stage_info,
            header_path=header_path,
        )
