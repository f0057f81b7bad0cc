"""The CPU oracle against the golden vectors and the reference's own known-answer test.

The goldens were produced by the reference's Python stages (symbolic + transpiler) followed by a C
compiler -- gcc, and clang (keys ``*_clang``: the reference compiles with ``zig cc`` = clang, which contracts
a*b+c where gcc -std=c17 does not) -- and oracle/sweep_oracle.c (tests/golden/make_golden.py).  Here the model
expressions come from THIS repo's symbolic stage and oracle/model_c.py instead, so bit-for-bit agreement under
both compilers pins both of those against the reference's Python half; the per-point formulas are pinned by
the reference's inequality (tests/test_doc.py:58) and by direct numpy evaluation below.
"""

import numpy as np
import pytest
from conftest import COMPILERS, MODELS, compare, golden, golden_key, oracle_model

import oracle
from oracle import OP

GRID_TAGS = {"hyperbolic": ("g16", "g64", "ragged", "off"), "doc": ("g16", "g64", "neg"), "angular": ("g16", "g64", "inner"), "egno": ("g16", "g64"), "d5": ("g16", "g64", "off")}


def test_both_reference_compilers_are_present():
    """gcc and the ROCm clang are part of the image, here and on the GPU box; a run that silently lost one of them
    would compare with half of the reference."""
    assert COMPILERS == ("gcc", "clang"), COMPILERS


@pytest.mark.parametrize("cc", COMPILERS)
@pytest.mark.parametrize("name", MODELS)
def test_oracle_matches_reference_goldens(name, cc):
    om, _ = oracle_model(name, cc)
    g = golden(name)
    for tag in GRID_TAGS[name]:
        n0, n1 = (int(v) for v in g[f"{tag}_shape"])
        ext = g[f"{tag}_extent"]
        for op, key in ((OP.RAW, "raw"), (OP.COMPLETE, "out"), (OP.CONSISTENCY, "consistency"), (OP.RAPIDTURN, "rapidturn"), (OP.EPSILON_V, "epsilon_v")):
            got = om.grid_sweep(op, g["args"], ext, n0, n1)
            # same expressions, same compiler, same libm: bit-identical
            compare(got, g[golden_key(f"{tag}_{key}", cc)], 0.0, f"{name}/{tag}/{key} [{cc}]")
        for accuracy in (1e-3, 0.5, 0.9):
            got = om.grid_sweep(OP.QDIF, g["args"], ext, n0, n1, accuracy=accuracy)
            assert np.array_equal(got, g[golden_key(f"{tag}_qdif_{accuracy}", cc)]), f"{name}/{tag}/qdif {accuracy} [{cc}]"
        # the reference's own v01 (hesse_bindings.rs:202-210): an expression of its own, not a copy of v10
        pts = oracle.grid_points(ext, n0, n1)
        v01 = np.array([om.hesse(x, g["args"])[0, 1] for x in pts]).reshape(n0, n1)
        compare(v01, g[golden_key(f"{tag}_v01", cc)], 0.0, f"{name}/{tag}/v01 [{cc}]")


def test_the_two_compilers_build_different_references():
    """What the second set of goldens is for: gcc -std=c17 (ISO: no contraction) and clang (contracts within an expression)
    return the same bits for the README hyperbolic model and measurably different numbers for the models that cancel --
    the reference differs from ITSELF by more than 1e-10 on EGNO, and in D5's NaN pattern on its singular lines."""
    rel = {}
    for name in MODELS:
        g = golden(name)
        a, b = g["g64_raw"], g["g64_raw_clang"]
        fin = np.isfinite(a) & np.isfinite(b)
        rel[name] = (np.abs(a[fin] - b[fin]) / np.maximum(np.abs(a[fin]), np.finfo(float).tiny), int((np.isnan(a) != np.isnan(b)).sum()))
    assert rel["hyperbolic"][0].max() == 0.0 and rel["hyperbolic"][1] == 0
    assert rel["doc"][0].max() < 1e-12 and rel["doc"][1] == 0
    assert np.median(rel["egno"][0]) > 1e-10  # measured: 1.7e-9 (median), 4e-5 (max)
    assert rel["d5"][1] > 0  # measured: 8 of the 64 x 48 x 5 values


@pytest.mark.parametrize("cc", COMPILERS)
def test_known_answer_doc_model(cc):
    """Reference tests/test_doc.py:50-51,58."""
    om, _ = oracle_model("doc", cc)
    x = np.array([2.0, -2.0])
    p = np.array([1.0])
    assert om.potential(x, p) == 1.9166666666666667
    assert np.allclose(om.hesse(x, p), np.array([[0.41206897, -1.05517241], [-1.05517241, -0.07873563]]))
    out = om.grid_sweep(OP.COMPLETE, p, (0.0, 2.5, 0.0, np.pi), 250, 250, threads=4)
    assert np.nanmax(out[:, :, 0]) <= 1


def test_hyperbolic_structure():
    """v10 == 0 makes `consistency` NaN everywhere, delta == 0, eta in {-3, NaN}."""
    g = golden("hyperbolic")
    out = g["g64_out"]
    assert np.isnan(out[:, :, 0]).all()
    assert (out[:, :, 4] == 0).all()
    eta = out[:, :, 3]
    assert np.all(np.isnan(eta) | (eta == -3.0))
    # nothing depends on the second field
    assert all(np.array_equal(out[:, j], out[:, 0], equal_nan=True) for j in range(out.shape[1]))


def test_per_point_formulas_in_numpy():
    """ops::complete_analysis (src/anguelova.rs:103-135) re-evaluated with numpy from the raw model values."""
    for name in ("doc", "angular"):
        g = golden(name)
        raw, out = g["g64_raw"], g["g64_out"]
        v, a, b, c, gr = (raw[..., k] for k in range(5))
        with np.errstate(all="ignore"):
            lhs = c / v
            rhs = 3.0 + 3.0 * (a / b) ** 2 + (a / v) * (b / a) ** 2
            cons = np.abs(lhs - rhs) / (np.abs(lhs) + np.abs(rhs))
            eps_v = gr / v**2
            vtt = (a * b**2 + c * a**2 - 2.0 * a * b**2) / (a**2 + b**2)
            vt2 = eps_v * (1.0 / (1.0 + (a / b) ** 2))
            eps_h = 3.0 * (eps_v - vt2) * (1.0 / (eps_v + np.abs(vtt) / v - vt2))
            delta = np.arctan(np.abs(b / a))
            omega = np.sqrt((vtt / v) * (3.0 - eps_h))
            eta = omega * np.tan(delta) - 3.0
        for k, arr in enumerate((cons, eps_v, eps_h, eta, delta, omega)):
            compare(arr, out[..., k], 1e-13, f"{name}/numpy/{k}")


def test_sweep_is_chunk_invariant_and_threaded():
    om, _ = oracle_model("doc")
    p = np.array([1.0])
    a = om.grid_sweep(OP.COMPLETE, p, (0.1, 2.5, 0.0, 3.0), 37, 29, threads=1)
    b = om.grid_sweep(OP.COMPLETE, p, (0.1, 2.5, 0.0, 3.0), 37, 29, threads=5)
    assert np.array_equal(a, b, equal_nan=True)


def test_oracle_rejects_wrong_parameter_count():
    from oracle.cpu_oracle import OracleError

    om, _ = oracle_model("hyperbolic")
    with pytest.raises(OracleError):
        om.grid_sweep(OP.COMPLETE, np.array([1.0, 1.0]), (-1, 1, -1, 1), 4, 4)
