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


@pytest.mark.gpu
def test_column_only_model_takes_the_column_broadcast_path(gpu_lib):
    """No model value depends on x[0]: one row image per parameter row (inflx_sweep_colvals_*) copied into every grid
    row (inflx_sweep_colstream).  Whatever the shape, row range, batch, layout and operation, element [p, i, j] is bit
    for bit what the on-trajectory kernel computes at that point; shapes whose rows are not whole 16-byte units fall
    back to the tile kernels."""
    import torch

    model, args, ext, om, comp, hdr, symdict = setup("column_only")
    art = Compiler(model, silent=True).compile()
    lib = gpu_lib.InflatoxDevLib(art.shared_object_path)
    assert lib.stage_info["out_mask"] == 2
    assert lib.sweep_plan(gpu_lib.OP_COMPLETE, 3, 300, 64)["path"] == "col_stream"
    assert lib.sweep_plan(gpu_lib.OP_RAW, 1, 301, 64)["path"] == "tile"  # 5 x 301 doubles per row: not whole 16-byte units
    assert lib.sweep_plan(gpu_lib.OP_EPSILON_V, 1, 301, 64)["path"] == "tile"
    rng = np.random.default_rng(77)
    x0a, x0b, x1a, x1b = ext
    ops = [(gpu_lib.OP_COMPLETE, 6), (gpu_lib.OP_CONSISTENCY, 1), (gpu_lib.OP_RAW, 5), (gpu_lib.OP_EPSILON_V, 1)]
    shapes = [(1, 1), (1, 258), (300, 2), (2, 2), (33, 255), (32, 256), (31, 514), (97, 64), (5, 1026), (70000, 6)]
    for case in range(30):
        n0, n1 = shapes[case] if case < len(shapes) else (int(rng.integers(1, 200)), int(rng.integers(1, 900)))
        P = int(rng.integers(1, 4))
        op, k = ops[case % len(ops)]
        layout = gpu_lib.LAYOUT_SOA if (case // 2) % 2 else gpu_lib.LAYOUT_AOS
        rb = int(rng.integers(0, n0))
        rc = int(rng.integers(1, n0 - rb + 1))
        rows = np.stack([args * (1.0 + 0.03 * q) for q in range(P)])
        ss = np.array([[x0a, x0b], [x1a, x1b]])
        dx0, dx1 = (x0b - x0a) / n0, (x1b - x1a) / n1
        xs0 = np.arange(rb, rb + rc, dtype=np.float64) * dx0 + x0a
        xs1 = np.arange(n1, dtype=np.float64) * dx1 + x1a
        # the values do not depend on x0: the trajectory needs one row only
        pts = np.column_stack([np.full(n1, xs0[0]), xs1])
        one = np.stack([lib.sweep_on_trajectory(op, rows[q], pts).reshape(n1, k) for q in range(P)])  # (P, n1, k)
        want = np.broadcast_to(one[:, None], (P, rc, n1, k))
        if layout == gpu_lib.LAYOUT_SOA:
            want = np.moveaxis(want, -1, 1)
        got = lib.sweep_host(op, rows, ss, n0, n1, row_begin=rb, row_count=rc, layout=layout)
        what = (case, n0, n1, P, op, layout, rb, rc, lib.sweep_plan(op, P, n1, rc, layout)["path"])
        assert np.array_equal(got.reshape(want.shape), want, equal_nan=True), what
    # against the oracle, and the summary of the column path (weight = number of rows)
    n0, n1 = 301, 518
    got = lib.sweep_host(gpu_lib.OP_COMPLETE, args, ext, n0, n1)
    want = om.grid_sweep(oracle.OP.COMPLETE, args, ext, n0, n1)
    compare(got, want, 1e-9, "column_only/301x518")
    from inflatox_amd.distributed import numpy_summary

    stats, ref = lib.sweep_stats(args, ext, n0, n1), numpy_summary(got)
    assert np.array_equal(stats["count"], ref["count"]) and np.array_equal(stats["min"], ref["min"]) and np.array_equal(stats["max"], ref["max"])
    # full size, device-resident: every row equals row 0; store-stream speed
    n = 8192
    out = torch.empty((n, n, 6), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    ms = lib.sweep_device_timed(gpu_lib.OP_COMPLETE, args, out.data_ptr(), out.numel() * 8, ext, n, n, repeats=20)
    torch.cuda.synchronize()
    row0 = out[:1]
    assert bool(((out == row0) | (torch.isnan(out) & torch.isnan(row0))).all())
    compare(row0[0].cpu().numpy(), om.grid_sweep(oracle.OP.COMPLETE, args, ext, 1, n)[0], 1e-9, "column_only/8192 row 0")
    print(f"column-broadcast sweep 8192^2: {ms:.3f} ms = {48 * n * n / ms / 1e9:.2f} TB/s")
    assert 48 * n * n / (ms * 1e-3) > 4.0e12  # HBM-write-bound like the row path (tile kernels: ~3e12)


# ---- the reference's own v01 (Hesse2D loads fns = [v00, v01, v10, v11], hesse_bindings.rs:202-210; `hesse` returns all four) ----
def _with_a_v01_of_its_own(name="abs_and_sign"):
    """A model whose v01 is NOT the expression tree of v10: the same function, written differently (common factors pulled out), as a
    symbolic stage leaves it when a simplification succeeds on one component and times out on the other."""
    import copy

    model, args, ext = _build(name)
    model = copy.copy(model)
    h = [list(row) for row in model.hesse_cmp]
    h[0][1] = sp.factor_terms(h[1][0])  # the same function with common factors pulled out of its sums: another tree, other roundings
    assert h[0][1] != h[1][0]
    model.hesse_cmp = h
    model.model_name = name + "_v01"
    return model, args, ext


def test_v01_is_recognised_as_v10_where_the_trees_are_equal():
    for name in MODELS:
        _, _, _, _, comp, hdr, _ = setup(name)
        assert comp.stage_info["v01_is_v10"] and "#define INFLX_V01_IS_V10 1" in hdr and "inflx_v01_point" not in hdr, name


def test_v01_of_its_own_on_the_host_matches_the_oracle():
    """INFLX_OP_HESSE = [v00, v01, v10, v11]: v01 through the generated `inflx_v01_point` against the oracle's `hesse`
    (the C function v01 of the reference's emitter, oracle/model_c.py), under the same host compiler with contraction off --
    the same expression, the same operations: equal to the last few ulps (pow chains), and NOT simply a copy of v10."""
    model, args, ext = _with_a_v01_of_its_own()
    comp = Compiler(model, silent=True)
    hdr = comp._generate_hip_header()
    assert not comp.stage_info["v01_is_v10"] and "inflx_v01_point" in hdr
    tw = HostTwin(hdr)
    assert not tw.v01_is_v10
    src, _ = oracle.emit_c_source(model)
    om = oracle.OracleModel(oracle.compile_c_model(src))
    rng = np.random.default_rng(3)
    pts = np.column_stack([rng.uniform(ext[0], ext[1], 300), rng.uniform(ext[2], ext[3], 300)])
    got = tw.trajectory(6, args, pts)
    want = np.array([om.hesse(x, args).reshape(-1) for x in pts])
    compare(got, want, 1e-11, "hesse on the host")
    assert not np.array_equal(want[:, 1], want[:, 2])  # the reference's v01 and v10 differ in their last bits here
    raw = tw.trajectory(4, args, pts)
    assert np.array_equal(got[:, [0, 2, 3]], raw[:, 1:4])  # v00, v10, v11 are the staged sweep values, bit for bit


@pytest.mark.gpu
def test_v01_of_its_own_on_the_gpu_matches_the_oracle(gpu_lib):
    from conftest import generalised_al

    model, args, ext = _with_a_v01_of_its_own()
    art = Compiler(model, silent=True).compile()
    al = generalised_al(art)
    src, _ = oracle.emit_c_source(model)
    om = oracle.OracleModel(oracle.compile_c_model(src))
    n0, n1 = 37, 53
    H = al.calc_H_array(args, ext[0], ext[1], ext[2], ext[3], [n0, n1])
    assert H.shape == (2, 2, n0, n1) and H.flags.writeable and not np.shares_memory(H[0, 1], H[1, 0])
    pts = oracle.grid_points(ext, n0, n1)
    want = np.array([om.hesse(x, args) for x in pts]).reshape(n0, n1, 2, 2).transpose(2, 3, 0, 1)
    compare(H, want, 1e-12, "calc_H_array with a v01 of its own")
    assert not np.array_equal(H[0, 1], H[1, 0])
    x = pts[n1 + 7]
    compare(al.calc_H(x, args), om.hesse(x, args), 1e-12, "calc_H")
