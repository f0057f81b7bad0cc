#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ from the reference's own Python stages.

Run ONLY in the authoring container (needs /root/reference; the GPU box has neither the
reference nor any need for this script -- it consumes the committed .npz/.json fixtures).

Pipeline:
  1. import the reference's ``inflatox`` Python package from /root/reference/python with three
     absent modules stubbed in ``sys.modules`` (``interruptingcow`` as a real SIGALRM timer,
     ``inflatox.version``, ``inflatox.libinflx_rs`` as names that raise when called);
  2. run the reference's ``InflationModelBuilder.new(...).build(...)`` and
     ``Compiler(...)._generate_c_file()`` on the model definitions of
     ``workloads/example_models.py`` (the models of the reference's README/tests);
  3. compile the reference-emitted C with the reference's flag list (into a temp dir), once with gcc and once with
     clang (the reference's compiler is ``zig cc`` = clang; gcc -std=c17 does not contract a*b+c, clang does);
  4. evaluate both objects through oracle/sweep_oracle.c (the C restatement of the Rust sweep) on small
     grids, adversarial points included, and store numbers only (``*_clang`` = the clang-built object):
        tests/golden/<model>.npz   args, extent, N0, N1, out (N0,N1,6), raw (N0,N1,5), v01 (N0,N1) + extras
        tests/golden/symbols.json  per-model symbol tables, N_PARAMETERS, printer strings
Nothing of the reference (source, generated C, binaries) is written into the repository.

``--add`` keeps the stored numbers (and asserts that this run reproduces them bit for bit) and adds the keys that are
missing; ``--basis`` writes basis.npz.
"""

from __future__ import annotations

import contextlib
import json
import os
import signal
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_PY = "/root/reference/python"
sys.path.insert(0, ROOT)


def _install_stubs():
    # interruptingcow.timeout: faithful ITIMER_REAL/SIGALRM context manager
    cow = types.ModuleType("interruptingcow")

    @contextlib.contextmanager
    def timeout(seconds, exception=RuntimeError):
        def handler(signum, frame):
            raise exception()

        old = signal.signal(signal.SIGALRM, handler)
        signal.setitimer(signal.ITIMER_REAL, seconds)
        try:
            yield
        finally:
            signal.setitimer(signal.ITIMER_REAL, 0)
            signal.signal(signal.SIGALRM, old)

    cow.timeout = timeout
    sys.modules["interruptingcow"] = cow

    pkg = types.ModuleType("inflatox")
    pkg.__path__ = [os.path.join(REF_PY, "inflatox")]
    sys.modules["inflatox"] = pkg
    ver = types.ModuleType("inflatox.version")
    ver.__version__ = "0.10.0"
    ver.__abi_version__ = "5.0.0"
    sys.modules["inflatox.version"] = ver
    rs = types.ModuleType("inflatox.libinflx_rs")

    def _absent(*a, **k):
        raise RuntimeError("libinflx_rs (Rust) is not built in this container")

    for nm in ("log_info", "log_warn", "open_inflx_dylib", "complete_analysis"):
        setattr(rs, nm, _absent)
    rs.log_warn = lambda msg: print(f"[ref warn] {msg}", file=sys.stderr)
    rs.log_info = lambda msg: print(f"[ref info] {msg}", file=sys.stderr)
    sys.modules["inflatox.libinflx_rs"] = rs


def load_reference():
    _install_stubs()
    import importlib

    symbolic = importlib.import_module("inflatox.symbolic")
    compiler = importlib.import_module("inflatox.compiler")
    return symbolic, compiler


def mp_truth(model, compiler_param_dict, args, ext, n0, n1, digits=50):
    """50-digit evaluation (mpmath) of V, v00, v10, v11, |dV|^2 at the float64 grid coordinates, with
    the float64 parameter values and the reference's 12-digit M_* constants (compiler.py:72-88), rounded
    to float64: the exact value of what the reference's C *means*, against which its rounding error --
    and therefore the agreement that can be demanded of any other implementation -- is measured."""
    import mpmath
    import sympy
    from sympy.printing.c import C99CodePrinter

    mpmath.mp.dps = digits
    consts = {
        # the float64 nearest to the reference's decimal literals, exactly as the C compiler reads them
        sympy.pi: sympy.Float(float("3.14159265359"), digits),
        sympy.E: sympy.Float(float("2.71828182846"), digits),
    }
    exprs = [model.potential, model.hesse_cmp[0][0], model.hesse_cmp[1][0], model.hesse_cmp[1][1], model.gradient_square]
    exprs = [sympy.sympify(e).xreplace(consts) for e in exprs]
    x0, x1 = model.coordinates
    plain = C99CodePrinter()._print_Symbol
    params = sorted((s for s in set().union(*[e.free_symbols for e in exprs]) - {x0, x1}), key=lambda s: int(compiler_param_dict[plain(s)][5:-1]))
    fn = sympy.lambdify([x0, x1, *params], exprs, modules="mpmath", cse=True)
    pvals = [mpmath.mpf(float(args[int(compiler_param_dict[plain(s)][5:-1])])) for s in params]
    dx0 = (ext[1] - ext[0]) / n0
    dx1 = (ext[3] - ext[2]) / n1
    out = np.full((n0, n1, 5), np.nan)
    for i in range(n0):
        for j in range(n1):
            a = mpmath.mpf(float(i * dx0 + ext[0]))  # the float64 coordinate the sweep uses
            b = mpmath.mpf(float(j * dx1 + ext[2]))
            try:
                vals = fn(a, b, *pvals)
                out[i, j] = [float(mpmath.re(v)) if mpmath.im(v) == 0 else np.nan for v in vals]
            except (ZeroDivisionError, ValueError, OverflowError):
                pass  # singular point: no finite truth
    return out


# grids: (tag, N0, N1, extent or None for the spec's default)
GRIDS = {
    # ("off": grids whose rows and columns MISS the model's singular lines -- hyperbolic: the row x0 = 0, where tanh(x0/L) = 0 gives
    # v11 = -inf; D5: r = 0 and theta = k pi/4 -- so that next to nothing is left out of the value comparison there, round 6)
    "hyperbolic": [("g16", 16, 16, None), ("g64", 64, 48, None), ("ragged", 7, 13, (-0.9, 1.3, 0.1, 2.0)), ("off", 24, 20, (-0.97, 1.03, -1.0, 1.0))],
    "doc": [("g16", 16, 16, None), ("g64", 64, 48, None), ("neg", 9, 11, (-1.0, 1.0, -2.0, 2.0))],
    "angular": [("g16", 16, 16, None), ("g64", 64, 48, None), ("inner", 12, 10, (-0.6, 0.6, -0.6, 0.6))],
    "egno": [("g16", 16, 16, None), ("g64", 64, 48, None)],
    "d5": [("g16", 16, 16, None), ("g64", 64, 48, None), ("off", 24, 20, (0.7, 35.3, 0.11, 12.41))],
}


def _reference_c(symbolic, compiler, name, tmp):
    """The reference's own symbolic stage and C emitter for an example model -> (model, Compiler, path of the C file)."""
    import joblib

    from workloads import example_models

    spec = example_models.get(name)
    print(f"== {name}: reference symbolic stage", flush=True)
    with joblib.parallel_backend("sequential"):
        builder = symbolic.InflationModelBuilder.new(
            spec.fields, spec.metric, spec.potential, model_name=name, init_sympy_printing=False, **spec.builder_kwargs
        )
        model = builder.build(spec.guesses)
    c_path = os.path.join(tmp, f"{name}.c")
    comp = compiler.Compiler(model, output_path=c_path, silent=True, **spec.compiler_kwargs)
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        comp._generate_c_file()
    return spec, model, comp, c_path


def _build(c_path, cc_name, cc_path):
    """The reference-emitted C built the way compiler.py:575-584 builds it, by one of the image's two C compilers."""
    import subprocess

    from oracle.model_c import REFERENCE_FLAGS

    so_path = c_path[:-2] + f".{cc_name}.so"
    subprocess.run([cc_path, "-o", so_path, c_path, *REFERENCE_FLAGS], check=True)
    return so_path


def _sweeps(om, spec, ext, n0, n1):
    """Everything a golden grid stores per compiler: the six outputs, the model values, the single quantities,
    flag_quantum_dif at three accuracies, and the reference's own v01 (hesse_bindings.rs:202-210 loads it; lib.rs:384-420
    returns it from `hesse`)."""
    from oracle import OP, grid_points

    res = {
        "out": om.grid_sweep(OP.COMPLETE, spec.args, ext, n0, n1),
        "raw": om.grid_sweep(OP.RAW, spec.args, ext, n0, n1),
        "consistency": om.grid_sweep(OP.CONSISTENCY, spec.args, ext, n0, n1),
        "rapidturn": om.grid_sweep(OP.RAPIDTURN, spec.args, ext, n0, n1),
        "epsilon_v": om.grid_sweep(OP.EPSILON_V, spec.args, ext, n0, n1),
    }
    for accuracy in (1e-3, 0.5, 0.9):  # ops::flag_quantum_diff through the reference's C function `v`
        res[f"qdif_{accuracy}"] = om.grid_sweep(OP.QDIF, spec.args, ext, n0, n1, accuracy=accuracy)
    pts = grid_points(ext, n0, n1)
    res["v01"] = np.array([om.hesse(x, spec.args)[0, 1] for x in pts]).reshape(n0, n1)
    return res


def main(models, add_only=False):
    """``add_only``: keep the stored gcc-built numbers and 50-digit values (asserting that this run reproduces the
    former bit for bit) and add what is missing -- the clang-built variants (``*_clang``) and ``*_v01``."""
    from oracle import OracleModel
    from oracle.model_c import reference_compilers

    symbolic, compiler = load_reference()
    sym_path = os.path.join(HERE, "symbols.json")
    symbols = json.load(open(sym_path)) if os.path.exists(sym_path) else {}
    tmp = tempfile.mkdtemp(prefix="inflx_golden_")
    compilers = reference_compilers()
    assert set(compilers) == {"gcc", "clang"}, compilers

    for name in models:
        spec, model, comp, c_path = _reference_c(symbolic, compiler, name, tmp)
        oms = {cc: OracleModel(_build(c_path, cc, path)) for cc, path in compilers.items()}
        om = oms["gcc"]
        symbols[name] = {
            "symbol_dictionary": comp.symbol_dict,
            "n_parameters": int(om.n_parameters),
            "n_fields": int(om.n_fields),
            "arg_names": spec.arg_names,
            "c_bytes": os.path.getsize(c_path),
        }
        assert om.n_parameters == len(spec.args), (om.n_parameters, spec.args)
        npz = os.path.join(HERE, f"{name}.npz")
        old = dict(np.load(npz)) if add_only else {}
        out = dict(old) if add_only else {"args": spec.args}
        for tag, n0, n1, ext in GRIDS[name]:
            ext = np.array(ext if ext is not None else spec.extent, dtype=np.float64)
            out[f"{tag}_extent"] = ext
            out[f"{tag}_shape"] = np.array([n0, n1])
            for cc, suffix in (("gcc", ""), ("clang", "_clang")):
                for key, arr in _sweeps(oms[cc], spec, ext, n0, n1).items():
                    full = f"{tag}_{key}{suffix}"
                    if full in old:  # the reference's stages gave the same C as when the stored numbers were made
                        assert np.array_equal(old[full], arr, equal_nan=True), f"{name}: {full} is not reproduced"
                    out[full] = arr
            if tag in ("g16", "g64") and f"{tag}_raw_mp" not in out:
                out[f"{tag}_raw_mp"] = mp_truth(model, comp.symbol_dict, spec.args, ext, n0, n1)
        if name == "doc":
            # the reference's only known-answer test on this path: tests/test_doc.py:50-51
            x = np.array([2.0, -2.0])
            out["kat_x"] = x
            for cc, suffix in (("gcc", ""), ("clang", "_clang")):
                out[f"kat_V{suffix}"] = np.array(oms[cc].potential(x, spec.args))
                out[f"kat_H{suffix}"] = oms[cc].hesse(x, spec.args)
                assert out[f"kat_V{suffix}"] == 1.9166666666666667
                assert np.allclose(out[f"kat_H{suffix}"], np.array([[0.41206897, -1.05517241], [-1.05517241, -0.07873563]]))
                full = oms[cc].grid_sweep(0, spec.args, spec.extent, 1000, 1000, threads=8)
                assert np.nanmax(full[:, :, 0]) <= 1  # tests/test_doc.py:58
                out[f"full1000_nanmax_consistency{suffix}"] = np.array(np.nanmax(full[:, :, 0]))
        np.savez_compressed(npz, **out)
        print(f"   wrote {name}.npz; symbols = {comp.symbol_dict}", flush=True)
        for m in oms.values():
            m.close()

    # printer known-answer strings (reference tests/test_compiler.py:40-53 hold the expected text)
    import sympy

    x, y, a, b, xd, yd = sympy.symbols("x y a b \\dot{{x}} \\dot{{y}}")
    pr = compiler.CInflatoxPrinter([x, y], [xd, yd])
    symbols["_printer_kat"] = {
        "x": pr._print_Symbol(x),
        "y": pr._print_Symbol(y),
        "a": pr._print_Symbol(a),
        "b": pr._print_Symbol(b),
        "xdot": pr._print_Symbol(xd),
        "ydot": pr._print_Symbol(yd),
        "x**2 + y": pr.doprint(x**2 + y),
        "x*y": pr.doprint(x * y),
        "sqrt(a)*y": pr.doprint(sympy.sqrt(a) * y),
        "sin(x)": pr.doprint(sympy.sin(x)),
    }
    json.dump(symbols, open(sym_path, "w"), indent=1, ensure_ascii=False, sort_keys=True)
    print("wrote symbols.json")


def basis_goldens(models):
    """tests/golden/basis.npz: the reference's C functions ``v``, ``w1`` and ``inner_prod`` (what
    validate_basis_* calls, src/lib.rs:141-300) at seeded points -- 64 inside the model's extent at the
    model's own parameters, and the 100 points in [-1,1)^2 with one parameter vector in [-10,10) that a
    ``validate_basis_at_random`` run would draw -- as gcc builds them and (``*_clang``) as clang does."""
    from oracle.cpu_oracle import basis_on_points
    from oracle.model_c import reference_compilers

    symbolic, compiler = load_reference()
    tmp = tempfile.mkdtemp(prefix="inflx_golden_")
    out = {}
    for name in models:
        spec, model, comp, c_path = _reference_c(symbolic, compiler, name, tmp)
        rng = np.random.default_rng(20250216 + len(name))
        x0a, x0b, x1a, x1b = spec.extent
        inside = np.stack([rng.uniform(x0a, x0b, 64), rng.uniform(x1a, x1b, 64)], axis=1)
        unit = rng.uniform(-1.0, 1.0, (100, 2))
        p_rand = rng.uniform(-10.0, 10.0, len(spec.args))
        out[f"{name}_args"] = np.asarray(spec.args, dtype=np.float64)
        out[f"{name}_inside_x"] = inside
        out[f"{name}_unit_x"] = unit
        out[f"{name}_unit_p"] = p_rand
        for cc, path in reference_compilers().items():
            so_path = _build(c_path, cc, path)
            suffix = "" if cc == "gcc" else f"_{cc}"
            out[f"{name}_inside_basis{suffix}"] = basis_on_points(so_path, spec.args, inside)
            out[f"{name}_unit_basis{suffix}"] = basis_on_points(so_path, p_rand, unit)
            out[f"{name}_unit_basis_args{suffix}"] = basis_on_points(so_path, spec.args, unit)
        print(f"   {name}: inside-extent norms {np.nanmin(out[f'{name}_inside_basis'][:, 0]):.6f}..{np.nanmax(out[f'{name}_inside_basis'][:, 0]):.6f}", flush=True)
    old_path = os.path.join(HERE, "basis.npz")
    if os.path.exists(old_path):  # the stored gcc numbers must come out again
        old = dict(np.load(old_path))
        for key, arr in old.items():
            if key in out:
                assert np.array_equal(arr, out[key], equal_nan=True), f"basis.npz: {key} is not reproduced"
    np.savez_compressed(old_path, **out)
    print("wrote basis.npz")


if __name__ == "__main__":
    names = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--basis" in sys.argv[1:]:
        basis_goldens(names or list(GRIDS))
    else:
        main(names or list(GRIDS), add_only="--add" in sys.argv[1:])
