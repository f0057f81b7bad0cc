"""The CPU oracle against the golden vectors and the reference's own known-answer test.

The goldens were produced by the reference's Python stages (symbolic + transpiler) followed by gcc
and oracle/sweep_oracle.c (tests/golden/make_golden.py).  Here the model expressions come from
THIS repo's symbolic stage and oracle/model_c.py instead, so agreement pins both of those against
the reference's Python half; the per-point formulas are pinned by the reference's inequality
(tests/test_doc.py:58) and by direct numpy evaluation below.
"""

import numpy as np
import pytest
from conftest import MODELS, compare, golden, oracle_model

from oracle import OP

GRID_TAGS = {"hyperbolic": ("g16", "g64", "ragged"), "doc": ("g16", "g64", "neg"), "angular": ("g16", "g64", "inner"), "egno": ("g16", "g64"), "d5": ("g16", "g64")}


@pytest.mark.parametrize("name", MODELS)
def test_oracle_matches_reference_goldens(name):
    om, _ = oracle_model(name)
    g = golden(name)
    for tag in GRID_TAGS[name]:
        n0, n1 = (int(v) for v in g[f"{tag}_shape"])
        ext = g[f"{tag}_extent"]
        for op, key in ((OP.RAW, "raw"), (OP.COMPLETE, "out"), (OP.CONSISTENCY, "consistency"), (OP.RAPIDTURN, "rapidturn"), (OP.EPSILON_V, "epsilon_v")):
            got = om.grid_sweep(op, g["args"], ext, n0, n1)
            # same expressions, same compiler, same libm: bit-identical
            compare(got, g[f"{tag}_{key}"], 0.0, f"{name}/{tag}/{key}")


def test_known_answer_doc_model():
    """Reference tests/test_doc.py:50-51,58."""
    om, _ = oracle_model("doc")
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
