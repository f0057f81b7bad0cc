#!/usr/bin/env python3
"""Generate inflatox_amd/csrc/inflx_sf_tables.h: Chebyshev coefficients for the Bessel functions of order
0 and 1 that inflx_sf.h evaluates on the device (J, Y, I, K; the integer orders above 1 and the spherical
functions follow from these by recurrence).

Every table is fitted here, from 50-digit mpmath values, on intervals chosen for this implementation -- no
coefficients are taken from GSL, Cephes or any other library.  The layout of the approximations is the
classical one (SLATEC/GSL use the same decomposition):

  x <= X0   the regular part of the function (after removing the logarithmic / 1/x singular terms, which
            are expressed through the companion function) is an even entire function of x: fitted in
            t = 2 x^2/X0^2 - 1
  x >  X0   J, Y: modulus M(x) and phase theta(x) with J = M cos(theta), Y = M sin(theta);
            sqrt(x) M and x (theta - x + (2 nu + 1) pi/4) are smooth in 1/x^2: fitted in t = 2 X0^2/x^2 - 1
            I, K: exp(-+x) sqrt(x) f(x) is smooth in 1/x: fitted piecewise in t linear in 1/x

usage: python scripts/make_sf_tables.py   (rewrites the header; takes a few minutes)
"""
import os
import sys

import mpmath as mp

mp.mp.dps = 50
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "inflatox_amd", "csrc", "inflx_sf_tables.h")
TOL = mp.mpf("1e-18")


def cheb_fit(f, n=64):
    """Chebyshev coefficients c_0..c_m of f on [-1, 1] (f ~ c_0/2 + sum c_j T_j), truncated where the tail
    drops below TOL relative to the largest coefficient."""
    nodes = [mp.cos(mp.pi * (k + mp.mpf(1) / 2) / n) for k in range(n)]
    vals = [f(t) for t in nodes]
    c = []
    for j in range(n):
        c.append(2 * mp.fsum(vals[k] * mp.cos(mp.pi * j * (k + mp.mpf(1) / 2) / n) for k in range(n)) / n)
    scale = max(abs(x) for x in c)
    m = n
    while m > 1 and abs(c[m - 1]) < TOL * scale:
        m -= 1
    assert m < n - 8, f"series did not converge within {n} terms (kept {m})"
    return c[:m]


def cheb_eval(c, t):
    b0 = b1 = mp.mpf(0)
    for cj in reversed(c[1:]):
        b0, b1 = 2 * t * b0 - b1 + cj, b0
    return t * b0 - b1 + c[0] / 2


def check(c, f, what):
    worst = mp.mpf(0)
    for k in range(401):
        t = mp.mpf(-1) + mp.mpf(2) * k / 400
        worst = max(worst, abs(cheb_eval(c, t) - f(t)))
    scale = max(abs(f(mp.mpf(-1))), abs(f(mp.mpf(1))), abs(f(mp.mpf(0))))
    print(f"  {what}: {len(c)} coefficients, max error {mp.nstr(worst, 3)} (scale {mp.nstr(scale, 3)})", flush=True)
    assert worst < mp.mpf("5e-17") * max(scale, 1), what


TABLES = []


def table(name, f, comment, n=64):
    c = cheb_fit(f, n)
    check(c, f, name)
    TABLES.append((name, c, comment))


def small(x0, t):
    """x from t = 2 x^2/x0^2 - 1"""
    return x0 * mp.sqrt((t + 1) / 2)


def large(x0, t):
    """x from t = 2 x0^2/x^2 - 1; t = -1 is x = infinity"""
    return x0 / mp.sqrt((t + 1) / 2) if t > -1 else mp.inf


def jy_small():
    x0 = mp.mpf(4)
    twopi = 2 / mp.pi

    def j0(t):
        return mp.besselj(0, small(x0, t))

    def j1x(t):  # J1(x)/x
        x = small(x0, t)
        return mp.besselj(1, x) / x if x != 0 else mp.mpf(1) / 2

    def y0r(t):  # Y0 - (2/pi) ln(x/2) J0
        x = small(x0, t)
        if x == 0:
            return twopi * mp.euler
        return mp.bessely(0, x) - twopi * mp.log(x / 2) * mp.besselj(0, x)

    def y1r(t):  # x (Y1 - (2/pi) ln(x/2) J1)
        x = small(x0, t)
        if x == 0:
            return -twopi
        return x * (mp.bessely(1, x) - twopi * mp.log(x / 2) * mp.besselj(1, x))

    table("INFLX_SF_J0_SMALL", j0, "J0(x), 0 <= x <= 4, t = x^2/8 - 1")
    table("INFLX_SF_J1_SMALL", j1x, "J1(x)/x, 0 <= x <= 4, t = x^2/8 - 1")
    table("INFLX_SF_Y0_SMALL", y0r, "Y0(x) - (2/pi) ln(x/2) J0(x), 0 < x <= 4, t = x^2/8 - 1")
    table("INFLX_SF_Y1_SMALL", y1r, "x (Y1(x) - (2/pi) ln(x/2) J1(x)), 0 < x <= 4, t = x^2/8 - 1")


def jy_large():
    x0 = mp.mpf(4)
    for nu in (0, 1):
        shift = (2 * nu + 1) * mp.pi / 4
        mu = 4 * nu * nu

        def amp(t, nu=nu):
            if t <= -1:
                return mp.sqrt(2 / mp.pi)
            x = large(x0, t)
            return mp.sqrt(x) * mp.sqrt(mp.besselj(nu, x) ** 2 + mp.bessely(nu, x) ** 2)

        def phase(t, nu=nu, shift=shift, mu=mu):
            if t <= -1:
                return mp.mpf(mu - 1) / 8
            x = large(x0, t)
            # at large x work with more digits: theta - x cancels log10(x) digits
            with mp.workdps(50 + int(mp.log10(x)) + 5):
                d = mp.atan2(mp.bessely(nu, x), mp.besselj(nu, x)) - (x - shift)
                d = d - 2 * mp.pi * mp.nint(d / (2 * mp.pi))
                return x * d

        table(f"INFLX_SF_AMP{nu}", amp, f"sqrt(x) |H{nu}(x)|, x >= 4, t = 32/x^2 - 1")
        table(f"INFLX_SF_PHASE{nu}", phase, f"x (arg H{nu}(x) - x + {2 * nu + 1} pi/4), x >= 4, t = 32/x^2 - 1")


def ik():
    # I0, I1: x <= 3 in t = 2x^2/9 - 1; 3 < x <= 8 in t = (48/x - 11)/5; x > 8 in t = 16/x - 1
    x0 = mp.mpf(3)
    table("INFLX_SF_I0_SMALL", lambda t: mp.besseli(0, small(x0, t)), "I0(x), 0 <= x <= 3, t = 2x^2/9 - 1")
    table("INFLX_SF_I1_SMALL", lambda t: (mp.besseli(1, small(x0, t)) / small(x0, t)) if t > -1 else mp.mpf(1) / 2, "I1(x)/x, 0 <= x <= 3, t = 2x^2/9 - 1")
    for nu in (0, 1):
        def mid(t, nu=nu):
            x = 48 / (5 * t + 11)
            return mp.exp(-x) * mp.sqrt(x) * mp.besseli(nu, x)

        def far(t, nu=nu):
            if t <= -1:
                return 1 / mp.sqrt(2 * mp.pi)
            x = 16 / (t + 1)
            return mp.exp(-x) * mp.sqrt(x) * mp.besseli(nu, x)

        table(f"INFLX_SF_I{nu}_MID", mid, f"exp(-x) sqrt(x) I{nu}(x), 3 <= x <= 8, t = (48/x - 11)/5")
        table(f"INFLX_SF_I{nu}_FAR", far, f"exp(-x) sqrt(x) I{nu}(x), x >= 8, t = 16/x - 1")
    # K0, K1: x <= 2 in t = x^2/2 - 1; 2 < x <= 8 in t = (16/x - 5)/3; x > 8 in t = 16/x - 1
    x0 = mp.mpf(2)

    def k0r(t):  # K0 + ln(x/2) I0
        x = small(x0, t)
        if x == 0:
            return -mp.euler
        return mp.besselk(0, x) + mp.log(x / 2) * mp.besseli(0, x)

    def k1r(t):  # x (K1 - ln(x/2) I1)
        x = small(x0, t)
        if x == 0:
            return mp.mpf(1)
        return x * (mp.besselk(1, x) - mp.log(x / 2) * mp.besseli(1, x))

    table("INFLX_SF_K0_SMALL", k0r, "K0(x) + ln(x/2) I0(x), 0 < x <= 2, t = x^2/2 - 1")
    table("INFLX_SF_K1_SMALL", k1r, "x (K1(x) - ln(x/2) I1(x)), 0 < x <= 2, t = x^2/2 - 1")
    for nu in (0, 1):
        def mid(t, nu=nu):
            x = 16 / (3 * t + 5)
            return mp.exp(x) * mp.sqrt(x) * mp.besselk(nu, x)

        def far(t, nu=nu):
            if t <= -1:
                return mp.sqrt(mp.pi / 2)
            x = 16 / (t + 1)
            return mp.exp(x) * mp.sqrt(x) * mp.besselk(nu, x)

        table(f"INFLX_SF_K{nu}_MID", mid, f"exp(x) sqrt(x) K{nu}(x), 2 <= x <= 8, t = (16/x - 5)/3")
        table(f"INFLX_SF_K{nu}_FAR", far, f"exp(x) sqrt(x) K{nu}(x), x >= 8, t = 16/x - 1")


def main():
    jy_small()
    jy_large()
    ik()
    lines = [
        "// Generated by scripts/make_sf_tables.py -- do not edit.",
        "// Chebyshev coefficients (f ~ c[0]/2 + sum_{j>=1} c[j] T_j(t)) fitted to 50-digit mpmath values; see the",
        "// generator for the decomposition.  Used by inflx_sf.h.",
        "#pragma once",
        "",
    ]
    for name, c, comment in TABLES:
        lines.append(f"// {comment}")
        lines.append(f"#define {name}_N {len(c)}")
        lines.append(f"#define {name}_COEFFS \\")
        body = [f"  {mp.nstr(x, 20, min_fixed=-1, max_fixed=-1)}" for x in c]
        lines.append(", \\\n".join(body))
        lines.append("")
    with open(OUT, "w") as fh:
        fh.write("\n".join(lines))
    print("wrote", OUT)


if __name__ == "__main__":
    sys.exit(main())
