"""Two-field inflation models used as sweep workloads.

These are *inputs* to the hot path (a potential V and a field-space metric G_ij written
as sympy expressions), not part of it. Each entry cites where the reference states the
model; the same (fields, metric, V) triple is fed to the reference's Python stages by
``tests/golden/make_golden.py`` and to this package's own symbolic stage, so that golden
vectors and product results start from identical input expressions.

Every factory returns a :class:`ModelSpec`; ``spec.builder_kwargs``/``spec.guesses`` are
the arguments the reference's tests pass to ``InflationModelBuilder.new(...).build(...)``
and ``spec.compiler_kwargs`` those passed to ``Compiler(...)``.
"""

from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import sympy as sp


@dataclass
class ModelSpec:
    name: str
    fields: list
    metric: list
    potential: sp.Expr
    #: parameter values in transpiler order (order of first appearance while printing)
    args: np.ndarray
    #: names of the parameters in that same order (checked against the symbol table)
    arg_names: list
    #: default sweep extent (x0_start, x0_stop, x1_start, x1_stop)
    extent: tuple
    builder_kwargs: dict = field(default_factory=dict)
    guesses: list | None = None
    compiler_kwargs: dict = field(default_factory=dict)


def hyperbolic() -> ModelSpec:
    """Quadratic potential on the hyperbolic plane, reference README.md:59-66,82,87."""
    phi, theta, L, m, phi0 = sp.symbols("φ θ L m φ0")
    V = (sp.Rational(1, 2) * m**2 * (phi - phi0) ** 2).nsimplify()
    g = [[1, 0], [0, L**2 * sp.sinh(phi / L) ** 2]]
    return ModelSpec(
        name="hyperbolic",
        fields=[phi, theta],
        metric=g,
        potential=V,
        args=np.array([1.0, 1.0, 1.0]),
        arg_names=["m", "φ0", "L"],
        extent=(-1.0, 1.0, -1.0, 1.0),
        builder_kwargs=dict(silent=True),
    )


def doc() -> ModelSpec:
    """Documentation example, reference tests/test_doc.py:27-54."""
    r, theta, m = sp.symbols("r θ m")
    V = (1 / 2 * m**2 * (theta**2 - 2 / (3 * r**2))).nsimplify()
    g = [[0.5, 0], [0, 0.5 * r**2]]
    return ModelSpec(
        name="doc",
        fields=[r, theta],
        metric=g,
        potential=V,
        args=np.array([1.0]),
        arg_names=["m"],
        extent=(0.0, 2.5, 0.0, float(np.pi)),
        builder_kwargs=dict(silent=True),
    )


def angular() -> ModelSpec:
    """Angular inflation, reference tests/test_angular.py:39-70."""
    p, x = sp.symbols("phi chi")
    mp, mx, a = sp.symbols("m_phi m_chi alpha")
    V = a / 2 * ((mp * p) ** 2 + (mx * x) ** 2).nsimplify()
    diag = 6 * a / (1 - p**2 - x**2) ** 2
    g = [[diag, 0], [0, diag]]
    alpha = 1 / 600
    m_phi = 2e-5
    m_chi = m_phi * np.sqrt(9)
    return ModelSpec(
        name="angular",
        fields=[p, x],
        metric=g,
        potential=V,
        args=np.array([alpha, m_chi, m_phi]),
        arg_names=["alpha", "m_chi", "m_phi"],
        extent=(-1.05, 1.05, -1.05, 1.05),
        builder_kwargs=dict(silent=True),
        compiler_kwargs=dict(cse=True),
    )


def egno() -> ModelSpec:
    """EGNO no-scale supergravity model, reference tests/test_egno.py:39-90."""
    alpha, m, c, a = sp.symbols("alpha m c a")
    r, theta = sp.symbols("r θ")
    Phi, PhiB, S, SB = sp.symbols("Phi Phi_B S S_B")
    # Kaehler potential; the field-space metric is its mixed second derivative at S = 0
    K = (-3 * alpha * sp.ln(Phi + PhiB - c * (Phi + PhiB - 1) ** 4) + (S * SB) / (Phi + PhiB) ** 3).nsimplify()
    K_phi_phibar = sp.diff(K, Phi, PhiB)
    g00 = K_phi_phibar.subs({Phi: r + 1j * theta, PhiB: r - 1j * theta}).nsimplify().simplify()
    g00 = g00.subs({S: 0, SB: 0}).simplify()
    g = [[g00, 0], [0, g00]]
    V = ((6 * m**2 * r**3 * ((a - r) ** 2 + theta**2)) / (a**2 * (2 * r - c * (1 - 2 * r) ** 4) ** (3 * alpha))).nsimplify()
    return ModelSpec(
        name="egno",
        fields=[r, theta],
        metric=g,
        potential=V,
        args=np.array([1e-3, 0.5, 1000.0, 1.0]),
        arg_names=["m", "a", "c", "alpha"],
        extent=(0.46, 0.50, 0.0, float(np.pi)),
        builder_kwargs=dict(silent=True, simplify=False, assertions=False),
        guesses=[[0, 1]],
        compiler_kwargs=dict(cse=True),
    )


def d5() -> ModelSpec:
    """D5-brane model, reference tests/test_d5.py:40-158."""
    from sympy.simplify.radsimp import collect_sqrt

    r, theta = sp.symbols("r θ2")
    gs, ls, N = sp.symbols("g_s l_s N")
    u, p, q = sp.symbols("u p q")
    a0, a1, b1, V0 = sp.symbols("a0 a1 b1 V0")
    pi = sp.pi

    mu5 = 1 / ((2 * pi) ** 5 * ls**6)
    T5 = mu5 / gs
    rho = r / (3 * u)
    harm = 2 / rho**2 - 2 * sp.ln(1 / rho**2 + 1)

    H = ((pi * N * gs * ls**4) / (12 * u**4) * harm).nsimplify().collect([u, r]).expand().powsimp(force=True)
    F = (H / 9 * (r**2 + 3 * u**2) ** 2 + (pi * q * ls**2) ** 2).nsimplify().collect([r, u]).expand().powsimp()
    gamma = 4 * pi**2 * ls**2 * p * q * T5 * gs
    sqrtF = sp.sqrt(F)

    g00 = collect_sqrt(4 * pi * p * T5 * sqrtF * ((r**2 + 6 * u**2) / (r**2 + p * u**2)), evaluate=True).expand().powsimp()
    g11 = (
        collect_sqrt((4 / 6) * pi * p * T5 * sqrtF * (r**2 + 6 * u**2), evaluate=True)
        .nsimplify()
        .collect([r, u])
        .expand()
        .powsimp()
    )

    Phi_min = (
        ((5 / 72) * (81 * (9 * rho**2 - 2) * rho**2 + 162 * sp.ln(9 * (rho**2 + 1)) + -9 + -160 * sp.ln(10)))
        .nsimplify()
        .collect([u])
        .expand()
        .powsimp()
    )
    Phi_h = (
        (
            a0 * harm
            + 2 * a1 * (6 + 1 / rho**2 - 2 * (2 + 3 * rho**2) * sp.ln(1 + 1 / rho**2)) * sp.cos(theta)
            + (b1 / 2) * (2 + 3 * rho**2) * sp.cos(theta)
        )
        .nsimplify()
        .collect([u, r])
        .expand()
        .powsimp()
    )
    V = V0 + (4 * pi * p * T5 / H) * (sp.sqrt(F) - (ls**2) * pi * q * gs) + gamma * (Phi_min + Phi_h)
    V = V.nsimplify().collect([ls, gs]).expand().powsimp()

    ls_v = 501.961
    return ModelSpec(
        name="d5",
        fields=[r, theta],
        metric=[[g00, 0], [0, g11]],
        potential=V,
        # [V0, a0, p, q, u, ls, a1, b1, gs, N]  (reference tests/test_d5.py:144-154)
        args=np.array([-1.17e-8, 0.001, 5.0, 1.0, 50 * ls_v, ls_v, 0.0005, 0.001, 0.01, 1000.0]),
        arg_names=["V0", "a0", "p", "q", "u", "l_s", "a1", "b1", "g_s", "N"],
        extent=(0.0, 36.0, 0.0, float(4 * np.pi)),
        builder_kwargs=dict(silent=True, simplify=False, assertions=False),
        guesses=[[1, 0]],
    )


ALL = {"hyperbolic": hyperbolic, "doc": doc, "angular": angular, "egno": egno, "d5": d5}


def get(name: str) -> ModelSpec:
    return ALL[name]()


# ---- models with special functions (the reference's GSL path, compiler.py:123-212) --------------------
BESSEL_PROBE_FUNCTIONS = (
    ("J", 0), ("J", 1), ("J", 5), ("Y", 0), ("Y", 1), ("Y", 3), ("I", 0), ("I", 1), ("I", 4), ("K", 0), ("K", 1), ("K", 2),
    ("j", 0), ("j", 1), ("j", 2), ("j", 6), ("y", 0), ("y", 1), ("y", 2), ("y", 4),
)  # fmt: skip


def bessel_probe():
    """V = sum_k c_k f_k(phi) over one function of every Bessel family and order class; a one-hot parameter
    vector turns ``calc_V`` into a probe of a single device function (tests only)."""
    phi, theta = sp.symbols("phi theta")
    fn = {"J": sp.besselj, "Y": sp.bessely, "I": sp.besseli, "K": sp.besselk, "j": sp.jn, "y": sp.yn}
    cs = sp.symbols(f"c0:{len(BESSEL_PROBE_FUNCTIONS)}")
    potential = sum(c * fn[kind](order, phi) for c, (kind, order) in zip(cs, BESSEL_PROBE_FUNCTIONS))
    return [phi, theta], [[1, 0], [0, 1]], potential


def bessel_toy():
    """A two-field model whose potential contains J0 (derivatives bring in J1, J2) -- the smallest model
    that needs ``Compiler(link_gsl=True)`` in the reference."""
    phi, theta = sp.symbols("phi theta")
    m, L = sp.symbols("m L")
    potential = m**2 * (2 + sp.besselj(0, phi)) * (1 + sp.Rational(1, 10) * sp.cos(theta))
    return [phi, theta], [[1, 0], [0, L**2 * (1 + phi**2)]], potential


def bessel_0f1():
    """Integer-order Bessel functions of both kinds and 0F1 in one potential: the functions the reference reaches
    through gsl_sf_bessel_Jn / _Kn and gsl_sf_hyperg_0F1."""
    phi, theta = sp.symbols("phi theta")
    m, c = sp.symbols("m c")
    shape = 3 + sp.besselj(2, phi) + sp.besselk(1, phi + 1) + sp.hyper([], [c], -(phi**2) / 4)
    potential = m**2 * shape * (1 + sp.cos(theta) / 10)
    return [phi, theta], [[1, 0], [0, 1 + phi**2]], potential


def bessel_real():
    """Bessel functions of REAL order -- a half-integer number and a model parameter -- of all four kinds: what the reference
    prints as gsl_sf_bessel_Jnu / _Ynu / _Inu / _Knu (compiler.py:199-212).  Differentiating shifts the orders by +-1 and
    +-2, so with nu >= 2 every order the Hesse matrix needs stays in GSL's domain nu >= 0."""
    phi, theta = sp.symbols("phi theta")
    m, nu = sp.symbols("m nu")
    shape = 4 + sp.besselj(sp.Rational(5, 2), phi) + sp.besselk(nu, phi + 1) + sp.bessely(nu, phi + 2) / 8 + sp.besseli(sp.Rational(7, 2), phi / 3) / 50
    potential = m**2 * shape * (1 + sp.cos(theta) / 10)
    return [phi, theta], [[1, 0], [0, 1 + phi**2]], potential


def hypergeometric():
    """1F1, 2F1 and 2F0 in a potential (gsl_sf_hyperg_1F1 / _2F1 / _2F0 in the reference); the parameters a and
    c of the Gauss function are model parameters."""
    phi, theta = sp.symbols("phi theta")
    m, a, c = sp.symbols("m a c")
    shape = 2 + sp.hyper([sp.Rational(1, 2)], [sp.Rational(3, 2)], -(phi**2)) + sp.hyper([a, 1], [c], phi / 10 - sp.Rational(1, 2))
    shape += sp.hyper([1, sp.Rational(5, 2)], [], -phi / 4)  # 2F0: gsl_sf_hyperg_2F0, x < 0
    potential = m**2 * shape * (1 + sp.cos(theta) / 10)
    return [phi, theta], [[1, 0], [0, 1 + phi**2]], potential
