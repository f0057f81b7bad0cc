"""CPU stand-in for the reference's GSL special functions -- TEST INFRASTRUCTURE ONLY.

The reference evaluates sympy's ``besselj/bessely/besseli/besselk/jn/yn`` and ``hyper`` through GSL
(``gsl_sf_bessel_*``, ``gsl_sf_hyperg_*``; python/inflatox/compiler.py:123-212; GSL is a system library of the user's machine,
no version is pinned by the reference).  GSL is absent from this image, so generated C that calls it cannot
be compiled here: **parity against GSL itself is unpinned**.  The functions are standard, though, and any
correct double-precision implementation agrees with GSL to a few ulps of the function's local amplitude;
the oracle therefore restates them with

  * ``mpmath`` (arbitrary precision) as ground truth for the functions themselves, and
  * ``scipy.special`` (Cephes / AMOS) via ``sympy.lambdify`` for whole model expressions,

and the tests bound the device implementation (inflatox_amd/csrc/inflx_sf.h) against both.
"""

from __future__ import annotations

import numpy as np


def mp_bessel(kind: str, order: int, x: float, dps: int = 40):
    """Ground truth for one function value; ``kind`` in J, Y, I, K (cylindrical), j, y (spherical)."""
    import mpmath as mp

    with mp.workdps(dps):
        x = mp.mpf(float(x))
        if kind == "J":
            return mp.besselj(order, x)
        if kind == "Y":
            return mp.bessely(order, x)
        if kind == "I":
            return mp.besseli(order, x)
        if kind == "K":
            return mp.besselk(order, x)
        half = order + mp.mpf(1) / 2
        if kind == "j":
            return mp.sqrt(mp.pi / (2 * x)) * mp.besselj(half, x)
        if kind == "y":
            return mp.sqrt(mp.pi / (2 * x)) * mp.bessely(half, x)
    raise ValueError(kind)


def mp_amplitude(kind: str, order: int, x: float) -> float:
    """Scale against which an absolute error is judged: the modulus of the oscillating pair (J,Y) / (j,y)
    above the turning point, the magnitude of the function itself elsewhere."""
    import mpmath as mp

    if kind in "JY" and x >= order:
        return float(mp.sqrt(mp_bessel("J", order, x) ** 2 + mp_bessel("Y", order, x) ** 2))
    if kind in "jy" and x >= order + 1:
        return float(mp.sqrt(mp_bessel("j", order, x) ** 2 + mp_bessel("y", order, x) ** 2))
    return abs(float(mp_bessel(kind, order, x)))


def lambdify_raw(model):
    """numpy/scipy evaluator of the five model values (V, v00, v10, v11, |dV|^2) of a symbolic model:
    ``f(x0, x1, *params_in_symbol_dictionary_order) -> (5, ...) array``.  Parameter order is the order of
    first appearance, as everywhere (compiler.py:62-75)."""
    import sympy

    exprs = [model.potential, model.hesse_cmp[0][0], model.hesse_cmp[1][0], model.hesse_cmp[1][1], model.gradient_square]
    return exprs, lambda symbols: sympy.lambdify(symbols, exprs, modules=["scipy", "numpy"])


def raw_values(model, symbol_dictionary: dict, args, pts) -> np.ndarray:
    """(n,5) raw model values at the (n,2) points with scipy.special standing in for GSL."""
    import sympy

    exprs, make = lambdify_raw(model)
    by_name = {}
    for e in exprs:
        for s in sympy.sympify(e).free_symbols:
            by_name[sympy.printing.c.C99CodePrinter()._print_Symbol(s)] = s
    order = sorted(((slot, name) for name, slot in symbol_dictionary.items() if slot.startswith("args[")), key=lambda t: int(t[0][5:-1]))
    x0, x1 = model.coordinates
    params = [by_name.get(name, sympy.Symbol(name)) for _, name in order]
    f = make([x0, x1] + params)
    pts = np.asarray(pts, dtype=np.float64)
    with np.errstate(all="ignore"):
        vals = f(pts[:, 0], pts[:, 1], *[float(a) for a in args])
    return np.stack([np.broadcast_to(np.asarray(v, dtype=np.float64), pts.shape[:1]) for v in vals], axis=1)


def raw_values_mp(model, symbol_dictionary: dict, args, pts, dps: int = 30) -> np.ndarray:
    """The same five values through ``sympy.lambdify(..., "mpmath")`` -- for expressions scipy's printer does
    not know (hypergeometric functions, Bessel functions of real order with derivatives).  Small n only."""
    import mpmath as mp
    import sympy

    exprs, _ = lambdify_raw(model)
    by_name = {}
    for e in exprs:
        for s in sympy.sympify(e).free_symbols:
            by_name[sympy.printing.c.C99CodePrinter()._print_Symbol(s)] = s
    order = sorted(((slot, name) for name, slot in symbol_dictionary.items() if slot.startswith("args[")), key=lambda t: int(t[0][5:-1]))
    x0, x1 = model.coordinates
    params = [by_name.get(name, sympy.Symbol(name)) for _, name in order]
    f = sympy.lambdify([x0, x1] + params, exprs, modules="mpmath")
    out = np.zeros((len(pts), 5))
    with mp.workdps(dps):
        for k, (a, b) in enumerate(np.asarray(pts, dtype=np.float64)):
            vals = f(mp.mpf(float(a)), mp.mpf(float(b)), *[mp.mpf(float(v)) for v in args])
            out[k] = [float(mp.re(v)) for v in vals]
    return out
