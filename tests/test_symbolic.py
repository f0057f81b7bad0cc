"""Symbolic stage (inflatox_amd/symbolic.py) checked on models of this repository's own, NUMERICALLY: every derived
expression is lambdified and compared, at seeded points, with an independent evaluation of what it is supposed to be --
textbook Christoffel symbols of the polar plane and of the Poincare half-plane, finite differences of the potential, the
coordinate-invariant trace of the covariant Hesse matrix.  (The expressions themselves are pinned against the reference's
symbolic stage through tests/golden/: test_oracle_golden.py evaluates this stage's output bit for bit against numbers the
reference's stage produced, and test_transpiler.py compares the parameter numbering with tests/golden/symbols.json.)
"""

import numpy as np
import pytest
import sympy as sp

from inflatox_amd import InflationModelBuilder

r, th, x, y = sp.symbols("r theta x y", real=True)
a, b, k = sp.symbols("a b k", positive=True)


def polar_builder():
    """The flat plane in polar coordinates with an anisotropic potential (written in the Cartesian x = r cos, y = r sin)."""
    V = a * (r * sp.cos(th)) ** 2 / 2 + b * (r * sp.sin(th)) ** 4 / 4 + k * r * sp.cos(th)
    return InflationModelBuilder.new([r, th], [[1, 0], [0, r**2]], V, "polar plane", silent=True, init_sympy_printing=False, assertions=False, simplify=False)


def half_plane_builder():
    """Poincare half-plane ds^2 = (dx^2 + dy^2) / y^2 (constant curvature -1)."""
    V = a * x**2 / 2 + b * sp.log(y) ** 2 + k * x * y
    return InflationModelBuilder.new([x, y], [[1 / y**2, 0], [0, 1 / y**2]], V, "half plane", silent=True, init_sympy_printing=False, assertions=False, simplify=False)


PARAMS = {a: 0.7, b: 1.3, k: 0.4}


def numeric(expr, coords, pts):
    f = sp.lambdify(list(coords), sp.sympify(expr).subs(PARAMS), "numpy")
    return np.broadcast_to(np.asarray(f(*pts.T), dtype=float), (pts.shape[0],))


def points(lo, hi, n=40, seed=5):
    rng = np.random.default_rng(seed)
    return np.column_stack([rng.uniform(lo[0], hi[0], n), rng.uniform(lo[1], hi[1], n)])


def test_christoffel_symbols_of_the_polar_plane():
    gamma = polar_builder().christoffels()
    pts = points((0.3, 0.1), (2.0, 3.0))
    want = np.zeros((2, 2, 2, pts.shape[0]))
    want[0, 1, 1] = -pts[:, 0]  # Gamma^r_{theta theta} = -r
    want[1, 0, 1] = want[1, 1, 0] = 1.0 / pts[:, 0]  # Gamma^theta_{r theta} = 1 / r
    for i in range(2):
        for j in range(2):
            for m in range(2):
                assert np.allclose(numeric(gamma[i][j][m], (r, th), pts), want[i, j, m], rtol=1e-13, atol=1e-13), (i, j, m)


def test_christoffel_symbols_of_the_half_plane():
    gamma = half_plane_builder().christoffels()
    pts = points((-1.0, 0.4), (1.0, 2.5))
    inv_y = 1.0 / pts[:, 1]
    want = np.zeros((2, 2, 2, pts.shape[0]))
    want[0, 0, 1] = want[0, 1, 0] = -inv_y  # Gamma^x_{xy}
    want[1, 0, 0] = inv_y  # Gamma^y_{xx}
    want[1, 1, 1] = -inv_y  # Gamma^y_{yy}
    for i in range(2):
        for j in range(2):
            for m in range(2):
                assert np.allclose(numeric(gamma[i][j][m], (x, y), pts), want[i, j, m], rtol=1e-13, atol=1e-13), (i, j, m)


def test_inner_product_and_normalisation_use_the_metric():
    bld = half_plane_builder()
    pts = points((-1.0, 0.4), (1.0, 2.5))
    u, v = [sp.Integer(2), x], [y, sp.Integer(-1)]
    got = numeric(bld.inner_prod(u, v), (x, y), pts)
    assert np.allclose(got, (2 * pts[:, 1] - pts[:, 0]) / pts[:, 1] ** 2, rtol=1e-14)
    unit = bld.normalize(u)
    assert np.allclose(numeric(bld.inner_prod(unit, unit), (x, y), pts), 1.0, rtol=1e-13)


def test_gram_schmidt_completes_an_orthonormal_pair():
    bld = polar_builder()
    pts = points((0.3, 0.1), (2.0, 3.0))
    first = bld.normalize([sp.Integer(1), sp.sin(th)])
    second = bld.gramm_schmidt([first], [r, sp.Integer(1)])
    assert np.allclose(numeric(bld.inner_prod(second, second), (r, th), pts), 1.0, rtol=1e-12)
    assert np.allclose(numeric(bld.inner_prod(first, second), (r, th), pts), 0.0, atol=1e-12)


@pytest.mark.parametrize("make,coords,lo,hi", [(polar_builder, (r, th), (0.3, 0.1), (2.0, 3.0)), (half_plane_builder, (x, y), (-1.0, 0.4), (1.0, 2.5))])
def test_built_model_numerically(make, coords, lo, hi):
    """Everything the transpiler consumes, at 40 seeded points: the basis is orthonormal in the metric and its first vector
    points along the gradient; |dV|^2 = g^ij dV_i dV_j with the derivatives taken by central differences; the projected Hesse
    matrix is symmetric and its trace is the Laplace-Beltrami operator of V -- g^ij (d_i d_j V - Gamma^m_ij d_m V), assembled
    here from finite differences and the textbook Christoffel symbols checked above."""
    bld = make()
    model = bld.build()
    pts = points(lo, hi)
    V = sp.lambdify(list(coords), sp.sympify(model.potential).subs(PARAMS), "numpy")
    g = [[numeric(model.metric[i][j], coords, pts) for j in range(2)] for i in range(2)]
    det = g[0][0] * g[1][1] - g[0][1] * g[1][0]
    ginv = [[g[1][1] / det, -g[0][1] / det], [-g[1][0] / det, g[0][0] / det]]
    h = 1e-5
    e = np.eye(2) * h

    def d1(i):
        return (V(*(pts + e[i]).T) - V(*(pts - e[i]).T)) / (2 * h)

    def d2(i, j):
        return (V(*(pts + e[i] + e[j]).T) - V(*(pts + e[i] - e[j]).T) - V(*(pts - e[i] + e[j]).T) + V(*(pts - e[i] - e[j]).T)) / (4 * h * h)

    dV = [d1(0), d1(1)]
    # basis: orthonormal, first vector along the raised gradient
    basis = [[numeric(model.basis[n][i], coords, pts) for i in range(2)] for n in range(2)]
    for n in range(2):
        for m in range(2):
            ip = sum(g[i][j] * basis[n][i] * basis[m][j] for i in range(2) for j in range(2))
            assert np.allclose(ip, 1.0 if n == m else 0.0, atol=1e-11), (n, m)
    grad_up = [sum(ginv[i][j] * dV[j] for j in range(2)) for i in range(2)]
    norm = np.sqrt(sum(g[i][j] * grad_up[i] * grad_up[j] for i in range(2) for j in range(2)))
    for i in range(2):
        assert np.allclose(basis[0][i], grad_up[i] / norm, rtol=1e-6, atol=1e-8)
    # |dV|^2
    want_g = sum(ginv[i][j] * dV[i] * dV[j] for i in range(2) for j in range(2))
    assert np.allclose(numeric(model.gradient_square, coords, pts), want_g, rtol=1e-7)
    # projected Hesse matrix: symmetric, trace = Laplace-Beltrami of V
    hesse = [[numeric(model.hesse_cmp[n][m], coords, pts) for m in range(2)] for n in range(2)]
    assert np.allclose(hesse[0][1], hesse[1][0], rtol=1e-10, atol=1e-12)
    gamma = bld.christoffels()
    lap = 0.0
    for i in range(2):
        for j in range(2):
            cov = d2(i, j) - sum(numeric(gamma[m][i][j], coords, pts) * dV[m] for m in range(2))
            lap = lap + ginv[i][j] * cov
    assert np.allclose(hesse[0][0] + hesse[1][1], lap, rtol=2e-5, atol=1e-6)
    # ... and its (v, v) component is the second covariant derivative along the unit gradient
    vv = 0.0
    for i in range(2):
        for j in range(2):
            cov = d2(i, j) - sum(numeric(gamma[m][i][j], coords, pts) * dV[m] for m in range(2))
            vv = vv + cov * basis[0][i] * basis[0][j]
    assert np.allclose(hesse[0][0], vv, rtol=2e-5, atol=1e-6)


def test_readme_entry_points_are_the_same_objects():
    """`SymbolicCalculation.new(...).execute()` of the reference's README (renamed in its v0.10.0) are `InflationModelBuilder` / `build`."""
    import inflatox_amd

    assert inflatox_amd.SymbolicCalculation is InflationModelBuilder
    bld = polar_builder()
    got, want = bld.execute(), bld.build()
    assert sp.simplify(sp.sympify(got.potential) - sp.sympify(want.potential)) == 0 and got.dim == want.dim == 2
