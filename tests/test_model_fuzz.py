"""Seeded random two-field models through the whole chain: symbolic stage -> transpiler (axis staging) -> the generated stage code,
compiled for the host (tests/host_twin.cpp, contraction off) -- against the CPU oracle built from the SAME model (oracle/model_c.py: the
reference's C emitter restated; gcc, no contraction).  Two statements per model, neither of which needs a tolerance model:

  * staged == unstaged, BIT FOR BIT: sharing identical nodes, moving whole sub-expressions and leading partial products / sums to the
    stage of their axes changes no rounding (inflatox_amd/staging.py; the five reference models are covered by
    tests/test_transpiler.py, this covers expression shapes nobody wrote by hand);
  * staged vs oracle: the two evaluate the same expression trees with the same libm and differ only where the transpiler spells
    pow(x, n) as a multiplication chain -- the median relative difference over a grid is at rounding level and nothing is off by
    more than 1e-9 of the value's scale; NaN patterns are equal.
Parameter numbering (the order in which the reference's printer meets the parameters) is compared as well: a user's `args` array
must mean the same thing here and there.

The generator draws potentials and diagonal metrics from a small grammar of smooth, positive-where-needed building blocks, so that
every model is well defined on its grid; with `cse=True` for every second model (the reference's per-function sympy.cse path).
"""

import functools

import numpy as np
import pytest
import sympy as sp
from host_twin import HostTwin

import oracle
from inflatox_amd import Compiler, InflationModelBuilder

x, y = sp.symbols("x y", real=True)
PARAMS = sp.symbols("a b c", positive=True)
SEEDS = tuple(range(24))


def _atom(rng, var):
    k = rng.integers(0, 9)
    p = PARAMS[rng.integers(0, 3)]
    return [var, var**2, sp.sin(var), sp.cos(p * var), sp.exp(-var / 3), sp.tanh(var), sp.log(2 + var**2), sp.sqrt(1 + p * var**2), 1 / (1 + var**2)][k]


def _term(rng):
    k = rng.integers(0, 4)
    p = PARAMS[rng.integers(0, 3)]
    if k == 0:
        return p * _atom(rng, x) * _atom(rng, y)
    if k == 1:
        return p * _atom(rng, x) ** int(rng.integers(1, 4))
    if k == 2:
        return _atom(rng, y) / (p + _atom(rng, x) ** 2)
    return p * sp.cos(x * y / 2) * _atom(rng, rng.choice([x, y]))


def _positive(rng, var):
    p = PARAMS[rng.integers(0, 3)]
    return [sp.Integer(1), 1 + var**2, sp.exp(var / 3), sp.cosh(var / 2) ** 2, (2 + sp.cos(var)) ** 2, p + var**2, p * (1 + sp.tanh(var) ** 2)][rng.integers(0, 7)]


@functools.lru_cache(maxsize=None)
def random_model(seed):
    rng = np.random.default_rng(1000 + seed)
    V = sum(_term(rng) for _ in range(int(rng.integers(2, 5)))) + PARAMS[0] * (x**2 + y**2) / 7  # the last term keeps the gradient non-zero
    G = [[_positive(rng, y if rng.random() < 0.5 else x), 0], [0, _positive(rng, x if rng.random() < 0.5 else y)]]
    model = InflationModelBuilder.new([x, y], G, V, model_name=f"fuzz{seed}", silent=True, init_sympy_printing=False, simplify=False, assertions=False).build()
    args = rng.uniform(0.5, 1.8, size=3)
    ext = (0.2 + rng.uniform(0, 0.3), 1.7 + rng.uniform(0, 0.6), -1.1 + rng.uniform(0, 0.3), 1.3 + rng.uniform(0, 0.5))
    return model, args, ext, bool(seed % 2)


def _n_used(symdict):
    return sum(1 for v in symdict.values() if v.startswith("args["))


@pytest.mark.parametrize("seed", SEEDS)
def test_random_model_through_the_transpiler(seed):
    model, args, ext, cse = random_model(seed)
    src, symdict = oracle.emit_c_source(model, cse=cse)
    om = oracle.OracleModel(oracle.compile_c_model(src))
    comp = Compiler(model, silent=True, cse=cse)
    staged = comp._generate_hip_header()
    assert comp.symbol_dict == symdict, "parameter numbering differs from the reference's print order"
    plain = Compiler(model, silent=True, cse=cse, staged=False)._generate_hip_header()
    p = args[: _n_used(symdict)]
    a, b = HostTwin(staged), HostTwin(plain)
    n0, n1 = 23, 37
    for op in (4, 0):  # the five model values, the six outputs
        got, ref = a.grid(op, p, ext, n0, n1), b.grid(op, p, ext, n0, n1)
        assert np.array_equal(got, ref, equal_nan=True), f"seed {seed}: staging changed bits of op {op}"
    got = a.grid(4, p, ext, n0, n1)
    want = om.grid_sweep(oracle.OP.RAW, p, ext, n0, n1)
    assert np.array_equal(np.isnan(got), np.isnan(want)), f"seed {seed}: NaN pattern"
    fin = np.isfinite(want)
    assert fin.mean() > 0.9
    scale = np.maximum(np.abs(want), np.nanmax(np.abs(np.where(fin, want, 0.0)), axis=(0, 1), keepdims=True) * 1e-6)
    rel = np.abs(got - want)[fin] / np.broadcast_to(scale, want.shape)[fin]
    assert np.median(rel) <= 1e-15 and rel.max() <= 1e-9, f"seed {seed}: median {np.median(rel):.2e}, max {rel.max():.2e}"


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS[:12])
def test_random_model_on_the_gpu(seed, gpu_lib):
    """The same models through hipcc and the C ABI: the sweep equals the host twin's program up to libm (OCML vs glibc), i.e. the
    oracle to a few ulps of every value's scale; fuzzed geometry (a ragged grid, a parameter batch) equals point evaluation bit for bit."""
    model, args, ext, cse = random_model(seed)
    src, symdict = oracle.emit_c_source(model, cse=cse)
    om = oracle.OracleModel(oracle.compile_c_model(src))
    art = Compiler(model, silent=True, cse=cse).compile()
    lib = gpu_lib.InflatoxDevLib(art.shared_object_path)
    p = args[: _n_used(symdict)]
    n0, n1 = 45, 333
    got = lib.sweep_host(gpu_lib.OP_RAW, p, ext, n0, n1)
    want = om.grid_sweep(oracle.OP.RAW, p, ext, n0, n1, threads=4)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    fin = np.isfinite(want)
    scale = np.maximum(np.abs(want), np.nanmax(np.abs(np.where(fin, want, 0.0)), axis=(0, 1), keepdims=True) * 1e-6)
    rel = np.abs(got - want)[fin] / np.broadcast_to(scale, want.shape)[fin]
    assert np.median(rel) <= 1e-14 and rel.max() <= 1e-9, (seed, float(np.median(rel)), float(rel.max()))
    six = lib.sweep_host(gpu_lib.OP_COMPLETE, np.stack([p, p * 1.1]), ext, n0, n1)
    pts = oracle.grid_points(ext, n0, n1)
    for k, row in enumerate((p, p * 1.1)):
        assert np.array_equal(six[k].reshape(-1, 6), lib.sweep_on_trajectory(gpu_lib.OP_COMPLETE, row, pts), equal_nan=True), (seed, k)
    # the quick point stage forced on (hoisted reciprocals + Markstein step, quick square roots, IEEE redo of rows that fail a range
    # test): exact by construction -- the same bits as the default build, on a grid that includes the rows / columns through 0
    quick = Compiler(model, silent=True, cse=cse, hoist_reciprocals=True).compile()
    lib_q = gpu_lib.InflatoxDevLib(quick.shared_object_path)
    wide = (-ext[1], ext[1], -ext[3], ext[3])
    for e, shape in ((ext, (n0, n1)), (wide, (64, 258))):
        a = lib.sweep_host(gpu_lib.OP_COMPLETE, p, e, *shape)
        b = lib_q.sweep_host(gpu_lib.OP_COMPLETE, p, e, *shape)
        assert np.array_equal(a, b, equal_nan=True), (seed, shape)
