"""Symbolic stage: from (fields, G_ij, V) to the expressions the sweep kernel evaluates.

This module is *upstream* of the hot path (it runs once per model, in seconds) and exists so
that the package is usable without the reference installed.  It keeps the public interface of
the reference's symbolic stage -- ``InflationModelBuilder.new(...).build(guesses)`` returning an
``InflationModel`` with the same attribute names (reference python/inflatox/symbolic.py:30-88,
127-138, 287-417) -- because the transpiler consumes exactly those attributes.

The mathematics (all standard Riemannian geometry on the 2-D scalar manifold):
    v^a   = G^{ab} d_b V / |dV|                      normalised potential gradient
    w^a   = unit vector orthogonal to v                (rotation of v in 2-D, or Gram-Schmidt
                                                        from a user-supplied guess)
    H_ab  = d_a d_b V - Gamma^c_{ab} d_c V             covariant Hesse matrix
    v{a}{b} = H_xy e_a^x e_b^y                         projection on the basis e_0 = v, e_1 = w
    |dV|^2 = G^{ab} d_a V d_b V
plus the background equations of motion, which are carried along for interface completeness.

Unlike the reference, no step depends on wall-clock time unless ``simplify=True`` is requested:
the reference wraps some steps in a 20 s alarm even with ``simplify=False`` (symbolic.py:629-634),
which makes its emitted expressions machine-dependent; here ``simplify=False`` means "no
simplification attempts at all" and the output is deterministic.
"""

from __future__ import annotations

import contextlib
import os
import signal
import sys
import threading

import sympy
from sympy.simplify import sqrtdenest


class InflationModel:
    """Container for every expression later stages need (reference symbolic.py:30-88)."""

    def __init__(
        self,
        model_name,
        coordinates,
        tangents,
        basis,
        eom_fields,
        eom_h,
        eom_hdot,
        potential,
        metric,
        gradient_square,
        hesse_cmp,
    ):
        n = len(coordinates)
        if any(len(row) != len(hesse_cmp) for row in hesse_cmp):
            raise Exception("The Hesse matrix is square; the provided list was not (number of columns != number of rows)")
        if any(len(row) != len(metric) for row in metric):
            raise Exception("The metric tensor is square; the provided list was not (number of columns != number of rows)")
        if len(hesse_cmp) != len(basis[0]):
            raise Exception("The provided Hesse Matrix and basis are of different dimensionality")
        if len(basis) != n:
            raise Exception("The dimension of the provided basis does not match the number of fields.")
        if len(tangents) != n:
            raise Exception("The number of coordinate symbols does not match the number of tangent symbols.")
        self.model_name = model_name
        self.coordinates = coordinates
        self.coordinate_tangents = tangents
        self.dim = n
        self.basis = basis
        self.eom_fields = eom_fields
        self.eom_h = eom_h
        self.eom_hdot = eom_hdot
        self.potential = potential
        self.metric = metric
        self.gradient_square = gradient_square
        self.hesse_cmp = hesse_cmp

    def __str__(self):
        return (
            "[Inflatox Inflation Model]\n"
            f"model name: {self.model_name}\n"
            f"dimensionality: {self.dim} field(s)\n"
            f"coordinates: {list(self.coordinates)}\n"
            f"potential: {self.potential}\n"
            f"metric: {sympy.Matrix(self.metric)}\n"
            f"basis vectors (cntr. var.): {[sympy.Matrix(v) for v in self.basis]}\n"
            f"hesse matrix: {sympy.Matrix(self.hesse_cmp)}\n"
        )


class SimplificationTimeOut(Exception):
    """Raised inside a time-limited simplification; always caught by the builder."""


@contextlib.contextmanager
def _alarm(seconds: float):
    """Raise SimplificationTimeOut in the main thread after ``seconds`` (POSIX only)."""
    usable = os.name != "nt" and threading.current_thread() is threading.main_thread() and seconds and seconds > 0
    if not usable:
        yield
        return

    def _fire(signum, frame):
        raise SimplificationTimeOut()

    previous = signal.signal(signal.SIGALRM, _fire)
    signal.setitimer(signal.ITIMER_REAL, seconds)
    try:
        yield
    finally:
        signal.setitimer(signal.ITIMER_REAL, 0)
        signal.signal(signal.SIGALRM, previous)


class InflationModelBuilder:
    """Derives an :class:`InflationModel` from a potential and a field-space metric."""

    @classmethod
    def new(
        cls,
        fields,
        field_metric,
        potential,
        model_name=None,
        silent=False,
        init_sympy_printing=True,
        assertions=True,
        simplify=True,
        simplify_timeout=None,
    ):
        """Same signature and defaults as reference symbolic.py:127-138."""
        if init_sympy_printing:
            sympy.init_printing()
        if simplify and os.name == "nt":
            print("[Inflatox Warning] cannot use simplifications on Windows. Continuing without simplifications.", file=sys.stderr)
            simplify, simplify_timeout = False, 0.0
        elif simplify_timeout is None:
            simplify_timeout = 20.0
        return cls(
            fields=fields,
            field_metric=field_metric,
            potential=potential,
            model_name=model_name if model_name is not None else "generic model",
            silent=silent,
            assertions=assertions,
            simplify=simplify,
            simplify_timeout=simplify_timeout,
        )

    def __init__(self, fields, field_metric, potential, model_name, silent, assertions, simplify, simplify_timeout):
        assert len(field_metric) == len(field_metric[0]), "field metric should be square"
        assert len(field_metric) == len(fields), "number of fields must match dimensionality of metric tensor"
        self.model_name = model_name
        self.dim = len(fields)
        self.fields = fields
        # tangent symbols get the same LaTeX-style names as the reference (symbolic.py:223) so that
        # symbol tables of the background equations agree
        self.field_derivatives = sympy.symbols([f"\\dot{{{sympy.latex(f)}}}" for f in fields])
        self.metric = field_metric
        self.V = potential
        self.assertions = assertions
        self.silent = silent
        self.simplify = simplify
        self.simplify_timeout = simplify_timeout

    # ---- time-limited rewriting helpers -----------------------------------------------------
    def _limited(self, fn, fallback):
        """Run ``fn()`` under the simplification alarm; on time-out return ``fallback``."""
        try:
            with _alarm(self.simplify_timeout):
                return fn()
        except SimplificationTimeOut:
            print(
                f"Simplification step timed out (>{self.simplify_timeout}s)!\n"
                "Consider increasing the simpliciation time-out time or turning off simplifications."
            )
            return fallback

    def simplify_expr(self, expr):
        if not self.simplify:
            return expr
        return self._limited(lambda: sympy.simplify(expr, ratio=1, inverse=True), expr)

    def expand_and_factor_expr(self, expr):
        if not self.simplify:
            return expr
        return self._limited(lambda: sympy.factor(sympy.expand(expr)), expr)

    def sqrt_and_denest_expr(self, expr):
        if not self.simplify:
            return sympy.sqrt(expr)
        return self._limited(lambda: sqrtdenest(sympy.sqrt(expr)), sympy.sqrt(expr))

    def print(self, msg):
        if not self.silent:
            print(msg)

    def display(self, expr, lhs=None):
        if self.silent:
            return
        sympy.pprint(sympy.Eq(lhs, expr, evaluate=False) if lhs is not None else expr)

    # ---- geometry ---------------------------------------------------------------------------
    def _metric_matrix(self):
        return sympy.Matrix(self.metric)

    def _gradient(self):
        return [sympy.diff(self.V, f) for f in self.fields]

    def _raise_index(self, covector):
        ginv = self._metric_matrix().inv()
        n = self.dim
        return [sum((ginv[a, b] * covector[b] for b in range(n)), sympy.Integer(0)) for a in range(n)]

    def inner_prod(self, v1, v2):
        """G_ab v1^a v2^b (reference symbolic.py:419-434)."""
        n = self.dim
        total = sympy.Integer(0)
        for a in range(n):
            for b in range(n):
                total += self.metric[a][b] * v1[a] * v2[b]
        return self.expand_and_factor_expr(total)

    def normalize(self, vec):
        """Scale ``vec`` to unit G-norm (reference symbolic.py:436-463)."""
        n = self.dim
        norm_sq = sympy.Integer(0)
        for a in range(n):
            for b in range(n):
                norm_sq += self.metric[a][b] * vec[a] * vec[b]
        if self.simplify:
            norm_sq = sympy.cancel(norm_sq)
        # sqrt(num/den) is taken as sqrt(num)/sqrt(den) so that nested roots can be denested
        num, den = sympy.fraction(norm_sq)
        root_num = self.sqrt_and_denest_expr(num)
        root_den = self.sqrt_and_denest_expr(den)
        scaled = [c * root_den / root_num for c in vec]
        return [sympy.cancel(c) for c in scaled] if self.simplify else scaled

    def christoffels(self):
        """Levi-Civita connection, indexed Gamma[up][down][down] (reference symbolic.py:465-490)."""
        n = self.dim
        g = self._metric_matrix()
        ginv = g.inv()
        x = self.fields
        gamma = [[[sympy.Integer(0)] * n for _ in range(n)] for _ in range(n)]
        for up in range(n):
            for lo1 in range(n):
                for lo2 in range(lo1 + 1):
                    acc = sympy.Integer(0)
                    for s in range(n):
                        acc += (ginv[up, s] / 2) * (sympy.diff(g[s, lo1], x[lo2]) + sympy.diff(g[s, lo2], x[lo1]) - sympy.diff(g[lo1, lo2], x[s]))
                    acc = self.simplify_expr(acc)
                    gamma[up][lo1][lo2] = acc
                    gamma[up][lo2][lo1] = acc
        return gamma

    def calc_hesse(self):
        """Twice-covariant Hesse matrix  d_a d_b V - Gamma^c_ab d_c V  (reference symbolic.py:492-530)."""
        n = self.dim
        gamma = self.christoffels()
        dV = self._gradient()
        hesse = [[None] * n for _ in range(n)]
        for a in range(n):
            for b in range(n):
                second = sympy.diff(self.V, self.fields[b], self.fields[a])
                conn = sympy.Integer(0)
                for c in range(n):
                    conn = conn + gamma[c][b][a] * dV[c]
                hesse[a][b] = self.simplify_expr(second - conn)
        return hesse

    def calc_gradient_square(self):
        """G^{ab} d_a V d_b V (reference symbolic.py:532-560)."""
        n = self.dim
        dV = self._gradient()
        ginv = self._metric_matrix().inv()
        total = 0.0  # the reference starts its accumulator from a float zero; kept for identical printing
        for a in range(n):
            for b in range(n):
                total += ginv[a, b] * dV[a] * dV[b]
        if self.simplify:
            total = self._limited(lambda: sympy.factor(sympy.expand(total)), total)
        return self.simplify_expr(total)

    def calc_v(self):
        """Unit vector along the potential gradient, contravariant (reference symbolic.py:562-583)."""
        up = self._raise_index(self._gradient())
        return [self.simplify_expr(c) for c in self.normalize(up)]

    def gramm_schmidt(self, current_basis, guess):
        """One Gram-Schmidt step (reference symbolic.py:585-636)."""
        n = len(current_basis[0])
        assert len(current_basis) < n, "current basis is already complete. No need for more vecs."
        y = list(guess)
        for e in current_basis:
            overlap = self.inner_prod(e, y)
            y = [y[a] - overlap * e[a] for a in range(n)]
        if self.simplify:
            y = self._limited(lambda: [sympy.factor(sympy.expand(c)) for c in y], y)
        return [self.simplify_expr(c) for c in self.normalize(y)]

    def project_hesse(self, hesse_matrix, v1, v2):
        """H_ab v1^a v2^b (reference symbolic.py:638-669)."""
        n = self.dim
        acc = sympy.Integer(0)
        for a in range(n):
            for b in range(n):
                acc = acc + hesse_matrix[a][b] * v1[a] * v2[b]
        return self.simplify_expr(acc)

    # ---- background equations (interface completeness; not on the sweep path) ----------------
    def compute_eom(self):
        n = self.dim
        gamma = self.christoffels()
        grad_up = self._raise_index(self._gradient())
        xd = self.field_derivatives
        out = []
        for a in range(n):
            conn = sympy.Integer(0)
            for b in range(n):
                for c in range(n):
                    conn += gamma[a][b][c] * xd[b] * xd[c]
            out.append(self.simplify_expr(self.expand_and_factor_expr(conn) + self.expand_and_factor_expr(grad_up[a])))
        return out

    def compute_eom_h(self):
        n = self.dim
        xd = self.field_derivatives
        acc = self.V
        for a in range(n):
            for b in range(n):
                acc += self.metric[a][b] * xd[a] * xd[b]
        return self.sqrt_and_denest_expr(self.expand_and_factor_expr(acc) / 3)

    def compute_eom_hdot(self):
        n = self.dim
        xd = self.field_derivatives
        acc = sympy.Integer(0)
        for a in range(n):
            for b in range(n):
                acc -= self.metric[a][b] * xd[a] * xd[b]
        return self.expand_and_factor_expr(acc / sympy.Integer(2))

    # ---- driver -----------------------------------------------------------------------------
    def build(self, guesses=None):
        """Derive all expressions (reference symbolic.py:287-417)."""
        n = self.dim
        if guesses is not None:
            assert len(guesses) == n - 1, "number of guessed vectors must equal the number of fields minus one (n-1)"

        self.print("Calculating orthonormal basis...")
        basis = [self.calc_v()]
        self.display(sympy.Matrix(basis[0]), lhs=sympy.symbols("v"))
        if guesses is None:
            if n != 2:
                raise Exception("guesses argument cannot be None if model has more than two fields")
            # in two dimensions the covector (-v^1, v^0) annihilates v; raise and normalise it
            w = self._raise_index([-basis[0][1], basis[0][0]])
            basis.append(self.normalize(w))
        else:
            for guess in guesses:
                basis.append(self.gramm_schmidt(basis, list(guess)))
        for k in range(1, n):
            self.display(sympy.Matrix(basis[k]), lhs=sympy.symbols(f"w_{k}"))

        if self.assertions:
            for a in range(n):
                for b in range(a, n):
                    target = 1 if a == b else 0
                    label = f"|w{a}|^2 = 1" if a == b else f"w{a}•w{b} = 0"
                    self.print(f"Testing if {label}")
                    try:
                        ok = sympy.Eq(target, self.inner_prod(basis[a], basis[b])).simplify()
                        assert ok, label
                    except (TypeError, AssertionError):
                        kind = "normalisation" if a == b else "orthogonality"
                        raise Exception(f"{kind} error: {label} does not hold")

        self.print("Calculating covariant Hesse matrix...")
        H = self.calc_hesse()
        self.display(sympy.Matrix(H), lhs=sympy.symbols("H"))

        self.print("Projecting the Hesse matrix on the vielbein basis...")
        H_proj = [[self.project_hesse(H, basis[a], basis[b]) for b in range(n)] for a in range(n)]

        self.print("Calculating the norm of the gradient...")
        grad_sq = self.calc_gradient_square()

        self.print("Computing the equations of motion...")
        eoms = self.compute_eom()
        eom_h = self.compute_eom_h()
        eom_hdot = self.compute_eom_hdot()

        return InflationModel(
            model_name=self.model_name,
            coordinates=self.fields,
            tangents=self.field_derivatives,
            basis=basis,
            eom_fields=eoms,
            eom_h=eom_h,
            eom_hdot=eom_hdot,
            potential=self.V,
            metric=self.metric,
            gradient_square=grad_sq,
            hesse_cmp=H_proj,
        )


    def execute(self, guesses=None):
        """The name this step has in the reference's README (README.md:72-73, ``calc.execute()``, from the releases in which
        the builder was called ``SymbolicCalculation``); the same call as :meth:`build`."""
        return self.build(guesses)


#: the builder's name in the reference's README snippet (README.md:72: ``inflatox.SymbolicCalculation.new(fields, g, V)``);
#: the reference renamed the class to ``InflationModelBuilder`` (symbolic.py:109) and left its README behind -- both names
#: work here, so that the snippet BASELINE.json cites as the definition of the headline model runs with only the import changed
SymbolicCalculation = InflationModelBuilder
