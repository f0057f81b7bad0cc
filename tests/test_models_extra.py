"""More models through the transpiler: shapes the five reference models do not exercise.

CPU: the staged header on the host (tests/host_twin.cpp) against the oracle built from the same model;
GPU: the same through the C ABI.  The models are synthetic (not taken from the reference).
"""

import functools

import numpy as np
import pytest
import sympy as sp
import tolerance as tol
from conftest import compare
from host_twin import HostTwin

import oracle
from inflatox_amd import Compiler, InflationModelBuilder


def _build(name):
    x, y = sp.symbols("x y", real=True)
    a, b, n = sp.symbols("a b n", real=True)
    if name == "column_only":  # nothing depends on x[0]: the mirror image of the hyperbolic model
        V = a * (y - b) ** 2 / 2
        G = [[1 + y**2, 0], [0, 1]]
        args, ext = [1.3, 0.4], (-1.0, 1.0, -2.0, 1.5)
    elif name == "abs_and_sign":  # Abs / sign print as fabs / comparison chains
        V = a * x**4 / 4 + b * y**2 / 2 + x * y
        G = [[1 + sp.Abs(x), 0], [0, 2 + sp.Abs(y)]]  # Christoffel symbols differentiate |.| once -> sign
        args, ext = [0.7, 1.1], (-2.0, 2.0, -1.5, 2.5)
    elif name == "symbolic_exponent":  # generic pow() with a parameter exponent, half-integer powers
        V = a * (1 + x**2) ** n + b * (1 + y**2) ** sp.Rational(5, 2)
        G = [[1, 0], [0, (2 + sp.cos(x)) ** 2]]
        args, ext = [0.9, 0.2, 1.7], (-1.0, 2.0, -1.0, 1.0)
    elif name == "transcendental":  # exp / log / atan / sinh mix, both axes in every term
        V = a * sp.exp(-x * y / 4) + b * sp.log(2 + sp.sin(x) * sp.cos(y)) + sp.atan(x + y)
        G = [[sp.cosh(y / 3) ** 2, 0], [0, 1 + x**2]]
        args, ext = [1.5, 0.8], (-1.0, 1.0, -1.0, 1.0)
    elif name == "libm_breadth":  # erf/erfc, cbrt, asinh, softplus, 2**x, atan2 in the metric, a Piecewise branch
        V = a * sp.erf(x) + b * sp.cbrt(1 + y**2) + sp.asinh(x * y) + sp.log(1 + sp.exp(x)) + sp.Piecewise((x**2 * y, x > 0), (0, True)) + sp.erfc(y / 2) * 2 ** (x / 3)
        G = [[1 + sp.tanh(y) ** 2, 0], [0, 2 + sp.atan2(x, 1 + y**2) ** 2]]
        args, ext = [0.8, 1.2], (-1.5, 2.0, -1.0, 1.5)
    else:
        raise KeyError(name)
    model = InflationModelBuilder.new([x, y], G, V, model_name=name, silent=True, init_sympy_printing=False, simplify=False, assertions=False).build()
    return model, np.array(args), ext


MODELS = ("column_only", "abs_and_sign", "symbolic_exponent", "transcendental", "libm_breadth")


@functools.lru_cache(maxsize=None)
def setup(name):
    model, args, ext = _build(name)
    src, symdict = oracle.emit_c_source(model)
    om = oracle.OracleModel(oracle.compile_c_model(src))
    comp = Compiler(model, silent=True)
    hdr = comp._generate_hip_header()
    return model, args, ext, om, comp, hdr, symdict


@pytest.mark.parametrize("name", MODELS)
def test_host_twin_matches_oracle(name):
    model, args, ext, om, comp, hdr, symdict = setup(name)
    assert comp.symbol_dict == symdict  # same parameter numbering as the restated reference printer
    tw = HostTwin(hdr)
    n0, n1 = 37, 29
    for op_t, op_o, rtol in ((4, oracle.OP.RAW, 1e-12), (0, oracle.OP.COMPLETE, 1e-9)):
        got = tw.grid(op_t, args, ext, n0, n1)
        want = om.grid_sweep(op_o, args, ext, n0, n1)
        compare(got, want, rtol, f"{name}/{op_t}")
    # staging is exact: staged == unstaged, bit for bit
    plain = HostTwin(Compiler(model, silent=True, staged=False)._generate_hip_header())
    assert np.array_equal(tw.grid(0, args, ext, n0, n1), plain.grid(0, args, ext, n0, n1), equal_nan=True)


def test_axis_masks_of_the_synthetic_models():
    assert setup("column_only")[4].stage_info["out_mask"] == 2
    assert setup("transcendental")[4].stage_info["out_mask"] == 3


@pytest.mark.gpu
@pytest.mark.parametrize("name", MODELS)
def test_gpu_matches_oracle(name, gpu_lib):
    model, args, ext, om, comp, hdr, symdict = setup(name)
    art = Compiler(model, silent=True).compile()
    lib = gpu_lib.InflatoxDevLib(art.shared_object_path)
    for n0, n1 in ((64, 300), (5, 7)):
        got = lib.sweep_host(gpu_lib.OP_COMPLETE, args, ext, n0, n1)
        want = om.grid_sweep(oracle.OP.COMPLETE, args, ext, n0, n1)
        raw = om.grid_sweep(oracle.OP.RAW, args, ext, n0, n1)
        # the oracle model object of a synthetic model is not registered in tolerance._models: build the
        # allowance from a plain relative bound on the model values instead of a measured one
        env = 64 * tol.EPS * np.abs(raw)
        allowed = tol.allowance_derived(raw, env, tol.epilogue)
        tol.check(got, want, allowed, None, f"{name}/{n0}x{n1}")
        got_raw = lib.sweep_host(gpu_lib.OP_RAW, args, ext, n0, n1)
        compare(got_raw, raw, 1e-10, f"{name}/raw")
