"""Special functions (reference compiler.py:123-212 -> GSL): the device header
csrc/inflx_sf.h built for the host against mpmath, the two printers, and a Bessel model end to end on
the host twin against the scipy stand-in.  GSL itself is absent from this image (parity unpinned, see
oracle/special.py)."""

import ctypes as C
import hashlib
import os
import subprocess
import tempfile

import numpy as np
import pytest
import sympy

from conftest import golden  # noqa: F401  (keeps the tests directory on sys.path)
from host_twin import HostTwin
from inflatox_amd import Compiler, InflationModelBuilder
from workloads import example_models
from inflatox_amd.compiler import CInflatoxPrinter, GSLInflatoxPrinter
from oracle import special

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
DP = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def sf():
    src = os.path.join(HERE, "sf_host.cpp")
    csrc = os.path.join(ROOT, "inflatox_amd", "csrc")
    h = hashlib.sha1()
    for f in (src, os.path.join(csrc, "inflx_sf.h"), os.path.join(csrc, "inflx_sf_tables.h")):
        h.update(open(f, "rb").read())
    # INFLX_TEST_SANITIZE=1 (manual runs: LD_PRELOAD=$(g++ -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python -m pytest ...):
    # the same functions under ASan + UBSan -- table indices, shifts and integer conversions of csrc/inflx_sf.h
    sanitize = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-g"] if os.environ.get("INFLX_TEST_SANITIZE") else []
    so = os.path.join(tempfile.gettempdir(), f"inflx_sf_host_{h.hexdigest()[:12]}{'_san' if sanitize else ''}.so")
    if not os.path.exists(so):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-mfma", "-ffp-contract=off", *sanitize, f"-I{csrc}", src, "-o", so + ".tmp"], check=True)
        os.replace(so + ".tmp", so)
    return C.CDLL(so)


def call(lib, kind, order, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros_like(x)
    named = (0, 1, 2) if kind in "jy" else (0, 1)
    if order in named:
        getattr(lib, f"sf_{kind}{order}")(x.ctypes.data_as(DP), x.size, out.ctypes.data_as(DP))
    else:
        getattr(lib, f"sf_{kind}{'l' if kind in 'jy' else 'n'}")(C.c_int(order), x.ctypes.data_as(DP), x.size, out.ctypes.data_as(DP))
    return out


def points():
    rng = np.random.default_rng(7)
    return np.concatenate([rng.uniform(0, 4, 25), rng.uniform(4, 60, 50), rng.uniform(60, 400, 15), 10.0 ** rng.uniform(-8, 0, 10), [2.0, 3.0, 4.0, 8.0, 2.404825557695773]])


def bound(order, x):
    """error budget in units of the local amplitude: 2e-15 plus what the recurrence adds per order, plus
    the phase error a large argument carries (one ulp of x is an absolute phase error)"""
    return (2e-15 + 2e-16 * order) * max(1.0, x / 10.0)


@pytest.mark.parametrize("kind", ["J", "Y", "I", "K", "j", "y"])
@pytest.mark.parametrize("order", [0, 1, 2, 3, 7, 20])
def test_device_functions_on_host_against_mpmath(sf, kind, order):
    x = points()
    if kind in "IK":
        x = x[x < 300]
    got = call(sf, kind, order, x)
    for xi, g in zip(x, got):
        want = special.mp_bessel(kind, order, xi)
        if abs(want) > 1e300 or abs(want) < 1e-300:
            assert not np.isnan(g)
            continue
        err = abs(float(want - float(g)))
        assert err <= bound(order, xi) * special.mp_amplitude(kind, order, xi), (kind, order, xi, g, float(want))


def test_symmetries_domains_and_limits(sf):
    x = np.array([0.7, 3.3, 12.5])
    for n in (0, 1, 2, 5):
        assert np.array_equal(call(sf, "J", n, -x), (-1) ** n * call(sf, "J", n, x))
        assert np.array_equal(call(sf, "I", n, -x), (-1) ** n * call(sf, "I", n, x))
    out = np.zeros(3)
    for name, sign in (("Jn", -1.0), ("Yn", -1.0), ("In", 1.0), ("Kn", 1.0)):  # negative orders
        getattr(sf, f"sf_{name}")(C.c_int(-3), x.ctypes.data_as(DP), 3, out.ctypes.data_as(DP))
        assert np.array_equal(out, sign * call(sf, name[0], 3, x)), name
    bad = np.array([0.0, -1.0, np.nan])
    for kind in "YKy":  # GSL: domain error for x <= 0
        for order in (0, 1, 2, 4):
            assert np.isnan(call(sf, kind, order, bad)).all(), (kind, order)
    assert np.isnan(call(sf, "j", 3, np.array([-1.0]))).all()
    zero = np.array([0.0])
    assert call(sf, "J", 0, zero)[0] == 1.0 and call(sf, "J", 4, zero)[0] == 0.0
    assert call(sf, "I", 0, zero)[0] == 1.0 and call(sf, "I", 3, zero)[0] == 0.0
    assert call(sf, "j", 0, zero)[0] == 1.0 and call(sf, "j", 1, zero)[0] == 0.0 and call(sf, "j", 5, zero)[0] == 0.0
    assert np.isinf(call(sf, "I", 0, np.array([800.0]))[0]) and call(sf, "K", 2, np.array([800.0]))[0] == 0.0
    assert abs(call(sf, "I", 0, np.array([712.0]))[0] / 2.4684110577627523e307 - 1) < 1e-13  # no premature overflow


def test_status_word_records_domain_errors_and_refusals(sf):
    """What the reference's GSL error handler would be called for (python/inflatox/compiler.py:145-149, src/err.rs:86-103) leaves a
    bit in the status word (csrc/inflx_sf.h: INFLX_SF_EDOM = 1, INFLX_SF_EDECLINED = 2) besides the NaN; arguments inside the domain
    and NaN arguments leave none."""
    sf.sf_status_take.restype = C.c_uint
    sf.sf_status_take()
    inside = np.array([0.25, 1.0, 7.5, 40.0])
    for kind in "JYIKjy":
        for order in (0, 1, 2, 5):
            assert np.isfinite(call(sf, kind, order, inside)).all()
    for kind in "JYIK":
        assert np.isfinite(_real(sf, kind, 0.5, inside)).all() and np.isfinite(_real(sf, kind, 3.25, inside)).all()
    assert np.isfinite(_hyp(sf, "0F1", (1.5,), inside)).all() and np.isfinite(_hyp(sf, "1F1", (0.5, 1.5), inside)).all()
    assert np.isfinite(_hyp(sf, "2F1", (0.5, 1.0, 1.5), np.array([-0.9, 0.0, 0.5, 0.95]))).all()
    assert np.isfinite(_hyp(sf, "2F0", (0.5, 1.0), np.array([-3.0, -0.01, 0.0]))).all()
    assert sf.sf_status_take() == 0
    nan = np.array([np.nan])
    for kind in "YKyj":
        assert np.isnan(call(sf, kind, 1, nan)).all() and np.isnan(call(sf, kind, 4, nan)).all()
    for kind in "JYIK":
        assert np.isnan(_real(sf, kind, 0.5, nan)).all() and np.isnan(_real(sf, kind, np.nan, inside)).all()
    assert np.isnan(_hyp(sf, "2F1", (0.5, 1.0, 1.5), nan)).all() and np.isnan(_hyp(sf, "2F0", (0.5, 1.0), nan)).all()
    assert sf.sf_status_take() == 0, "a NaN argument is not a domain error"
    cases = [(lambda k=kind, n=order: call(sf, k, n, np.array([-1.0]))) for kind in "YKyj" for order in (0, 1, 2, 4)]
    cases += [(lambda k=kind: call(sf, k, 3, np.array([0.0]))) for kind in "YKy"]
    cases += [(lambda k=kind: _real(sf, k, -0.5, np.array([1.0]))) for kind in "JYIK"]  # negative order
    cases += [(lambda k=kind: _real(sf, k, 0.5, np.array([-1.0]))) for kind in "JYIK"]
    cases += [(lambda k=kind: _real(sf, k, 0.5, np.array([0.0]))) for kind in "YK"]
    cases += [lambda: _hyp(sf, "0F1", (-2.0,), np.array([1.0])), lambda: _hyp(sf, "1F1", (0.5, -1.0), np.array([1.0])),
              lambda: _hyp(sf, "2F1", (0.5, 1.0, -3.0), np.array([0.5])), lambda: _hyp(sf, "2F1", (0.5, 1.0, 1.5), np.array([1.0])),
              lambda: _hyp(sf, "2F1", (0.5, 1.0, 1.5), np.array([-1.5])), lambda: _hyp(sf, "2F0", (0.5, 1.0), np.array([0.5]))]
    for k, case in enumerate(cases):
        assert np.isnan(case()).all(), k
        assert sf.sf_status_take() == 1, k  # INFLX_SF_EDOM
        assert sf.sf_status_take() == 0, k  # reading cleared it
    # refusals: inside the domain, no answer to 1e-13
    assert np.isnan(_real(sf, "J", 2e7, np.array([1.0]))).all() and sf.sf_status_take() == 2
    assert np.isnan(_hyp(sf, "2F0", (0.1, 0.2), np.array([-5.0]))).all() and sf.sf_status_take() == 2


def _real(lib, kind, order, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros_like(x)
    getattr(lib, f"sf_{kind}nu")(C.c_double(order), x.ctypes.data_as(DP), x.size, out.ctypes.data_as(DP))
    return out


def test_gsl_printer_strings_match_the_reference():
    """The strings of the reference's tests/test_compiler.py:56-84."""
    x, y, a, b, xdot, ydot = sympy.symbols("x y a b \\dot{{x}} \\dot{{y}}")
    pr = GSLInflatoxPrinter([x, y], [xdot, ydot])
    pr.doprint(sympy.besselj(1, x))
    assert pr.BESSELH in pr.required_headers
    pr.doprint(sympy.hyper([], [1], x))
    assert pr.HYPERH in pr.required_headers
    assert pr.doprint(sympy.besselj(0, x)) == "gsl_sf_bessel_J0(x[0])"
    assert pr.doprint(sympy.besselj(1, x)) == "gsl_sf_bessel_J1(x[0])"
    assert pr.doprint(sympy.besselj(10, x)) == "gsl_sf_bessel_Jn(10, x[0])"
    assert pr.doprint(sympy.besselj(0.5, x)) == "gsl_sf_bessel_Jnu(0.50000000000000000, x[0])"
    assert pr.doprint(sympy.hyper([0, 1], [], x)) == "gsl_sf_hyperg_2F0(0, 1, x[0])"
    assert pr.doprint(sympy.hyper([0, 1], [2], x)) == "gsl_sf_hyperg_2F1(0, 1, 2, x[0])"
    assert pr.doprint(sympy.hyper([0], [1], x)) == "gsl_sf_hyperg_1F1(0, 1, x[0])"
    assert pr.doprint(sympy.hyper([], [0], x)) == "gsl_sf_hyperg_0F1(0, x[0])"
    with pytest.raises(Exception) as exc:
        pr.doprint(sympy.hyper([0, 3, 4], [1, 2], x))
    assert "Cannot compute" in str(exc.value)
    # remaining families of compiler.py:199-212
    assert pr.doprint(sympy.bessely(1, y)) == "gsl_sf_bessel_Y1(x[1])"
    assert pr.doprint(sympy.besseli(3, x)) == "gsl_sf_bessel_In(3, x[0])"
    assert pr.doprint(sympy.besselk(0, x)) == "gsl_sf_bessel_K0(x[0])"
    assert pr.doprint(sympy.jn(2, x)) == "gsl_sf_bessel_j2(x[0])"
    assert pr.doprint(sympy.yn(7, x)) == "gsl_sf_bessel_yl(7, x[0])"
    assert isinstance(pr, CInflatoxPrinter) and pr.doprint(x**2 + y) == "pow(x[0], 2) + x[1]"


def _toy():
    fields, metric, potential = example_models.bessel_toy()
    return InflationModelBuilder.new(fields, metric, potential, model_name="bessel_toy", init_sympy_printing=False, silent=True).build()


def test_device_printer_and_unsupported_functions():
    model = _toy()
    comp = Compiler(model, silent=True, link_gsl=True)
    hdr = comp._generate_hip_header()
    assert "inflx_sf_bessel_J0(x0)" in hdr and "inflx_sf_bessel_J1(x0)" in hdr and "inflx_sf_bessel_Jn(2, x0)" in hdr
    assert comp.symbol_dict == {"phi": "x[0]", "theta": "x[1]", "m": "args[0]", "L": "args[1]"}
    # Bessel functions of x[0] alone are row-stage work: none may be left in the per-point stage
    point_stage = hdr[hdr.index("inflx_stage_point") : hdr.index("inflx_basis_point")]
    assert "inflx_sf_bessel" not in point_stage
    phi, theta = model.coordinates
    nu = sympy.Symbol("nu")
    for bad in (sympy.hyper([1, 2, 3], [4, 5], phi), sympy.hyper([1], [], phi / 9), sympy.yn(nu, phi)):
        fields, metric, _ = example_models.bessel_toy()
        m2 = InflationModelBuilder.new(fields, metric, bad + 2, model_name="bad", init_sympy_printing=False, silent=True, assertions=False, simplify=False).build()
        with pytest.raises((NotImplementedError, KeyError, Exception)):  # KeyError / Exception: the reference printer's own refusals
            Compiler(m2, silent=True, link_gsl=True)._generate_hip_header()


def test_bessel_model_on_host_twin_against_scipy_stand_in():
    model = _toy()
    comp = Compiler(model, silent=True, link_gsl=True)
    tw = HostTwin(comp._generate_hip_header())
    args = np.array([1.3, 0.7])
    n0, n1, ext = 24, 20, (0.3, 14.0, 0.1, 3.0)
    got = tw.grid(4, args, ext, n0, n1).reshape(-1, 5)
    import oracle

    want = special.raw_values(model, comp.symbol_dict, args, oracle.grid_points(ext, n0, n1))
    scale = np.maximum(np.abs(want), np.abs(want).max(axis=0, keepdims=True) * 1e-3)
    assert np.isfinite(want).all()
    assert (np.abs(got - want) / scale).max() < 1e-10


@pytest.mark.parametrize("c", [0.5, 1.0, 1.5, 3.7, 25.5, -0.5, -2.3, 0.1])
def test_hyperg_0F1_on_host_against_mpmath(sf, c):
    import mpmath as mp

    rng = np.random.default_rng(23)
    x = np.concatenate([rng.uniform(-50, 50, 40), rng.uniform(-1, 1, 15), 10.0 ** rng.uniform(-8, -1, 8), -(10.0 ** rng.uniform(-8, -1, 8)), [0.0, 400.0, -400.0, -401.0, -2500.0, -1e5]])
    out = np.zeros_like(x)
    sf.sf_0F1(C.c_double(c), x.ctypes.data_as(DP), x.size, out.ctypes.data_as(DP))
    with mp.workdps(40):
        for xi, g in zip(x, out):
            want = mp.hyp0f1(c, mp.mpf(float(xi)))
            scale = abs(float(want))
            if xi < 0:  # oscillating: the envelope of the underlying Bessel pair
                z, a = 2 * mp.sqrt(-mp.mpf(float(xi))), abs(c - 1)
                if z > a:
                    scale = max(scale, float(abs(mp.gamma(c)) * (-mp.mpf(float(xi))) ** ((1 - c) / 2) * mp.sqrt(mp.besselj(a, z) ** 2 + mp.bessely(a, z) ** 2)))
            if scale > 1e300 or scale < 1e-300:
                continue
            assert abs(float(want - float(g))) <= 5e-13 * scale, (c, xi, g, float(want))
    bad = np.zeros(2)
    for pole in (0.0, -3.0):  # GSL: domain error
        sf.sf_0F1(C.c_double(pole), np.array([1.0, -1.0]).ctypes.data_as(DP), 2, bad.ctypes.data_as(DP))
        assert np.isnan(bad).all()


def test_0F1_model_on_host_twin_against_mpmath():
    fields, metric, potential = example_models.bessel_0f1()
    model = InflationModelBuilder.new(fields, metric, potential, model_name="bessel_0f1", init_sympy_printing=False, silent=True, assertions=False, simplify=False).build()
    comp = Compiler(model, silent=True, link_gsl=True)
    hdr = comp._generate_hip_header()
    for f in ("inflx_sf_bessel_Jn(", "inflx_sf_bessel_K1(", "inflx_sf_hyperg_0F1("):
        assert f in hdr
    assert comp.symbol_dict == {"phi": "x[0]", "theta": "x[1]", "m": "args[0]", "c": "args[1]"}
    tw = HostTwin(hdr)
    args = np.array([1.2, 1.5])
    n0, n1, ext = 14, 6, (0.4, 9.0, 0.2, 2.9)
    import oracle

    pts = oracle.grid_points(ext, n0, n1)
    got = tw.grid(4, args, ext, n0, n1).reshape(-1, 5)
    want = special.raw_values_mp(model, comp.symbol_dict, args, pts)
    scale = np.maximum(np.abs(want), np.abs(want).max(axis=0, keepdims=True) * 1e-3)
    assert np.isfinite(want).all() and (np.abs(got - want) / scale).max() < 1e-10
    # what the reference's printer refuses is refused here as well -- loudly
    phi = model.coordinates[0]
    for bad in (sympy.hyper([1, 2, 3], [4], phi / 9), sympy.jn(sympy.Rational(1, 2), phi)):
        m2 = InflationModelBuilder.new(fields, metric, bad + 2, model_name="bad", init_sympy_printing=False, silent=True, assertions=False, simplify=False).build()
        with pytest.raises((NotImplementedError, KeyError, Exception)):  # KeyError / Exception: the reference printer's own refusals
            Compiler(m2, silent=True, link_gsl=True)._generate_hip_header()


REAL_ORDERS = [0.0, 1e-12, 0.1, 0.25, 0.5, 0.75, 0.9999999, 1.0, 1.5, 2.3, 3.999, 7.5, 12.25, 20.0, 33.3, 75.5, 150.1]


@pytest.mark.parametrize("kind", ["J", "Y", "I", "K"])
def test_real_order_bessel_functions_on_host_against_mpmath(sf, kind):
    """gsl_sf_bessel_{J,Y,I,K}nu's counterparts (csrc/inflx_sf.h: two integral representations of DLMF chapter 10 by the
    trapezoidal rule, recurrences in their stable directions, Wronskians for the minimal solutions) against 40-digit values:
    orders from 0 to 150 including integers and near-integers, arguments from 1e-100 to 1e5.  Bound: 1e-14 of the amplitude
    sqrt(J^2 + Y^2) (of J itself where the order is above the argument and J is the minimal solution; of the function itself
    for I and K), times x/10 for the phase error a large argument carries."""
    import mpmath as mp

    rng = np.random.default_rng(3)
    xs = np.concatenate([10.0 ** rng.uniform(-12, 0, 12), rng.uniform(0, 4, 15), rng.uniform(4, 60, 25), rng.uniform(60, 400, 8), [1e-100, 1e3, 1e5]])
    if kind in "IK":
        xs = xs[xs < 300]
    fn = {"J": mp.besselj, "Y": mp.bessely, "I": mp.besseli, "K": mp.besselk}[kind]
    worst = 0.0
    with mp.workdps(40):
        for nu in REAL_ORDERS:
            out = np.zeros_like(xs)
            getattr(sf, f"sf_{kind}nu")(C.c_double(nu), xs.ctypes.data_as(DP), xs.size, out.ctypes.data_as(DP))
            for xi, g in zip(xs, out):
                want = fn(nu, mp.mpf(float(xi)))
                if abs(want) > 1e300 or abs(want) < 1e-300:
                    assert not np.isnan(g), (kind, nu, xi)
                    continue
                amp = abs(want)
                if kind in "JY" and not (kind == "J" and nu >= xi):
                    amp = mp.sqrt(mp.besselj(nu, xi) ** 2 + mp.bessely(nu, xi) ** 2)
                err = float(abs(want - mp.mpf(float(g))) / amp) / max(1.0, xi / 10.0)
                worst = max(worst, err)
                assert err <= 1e-14, (kind, nu, xi, g, float(want), err)
    print(f"{kind}nu: worst error {worst:.2e}")
    # domains as in GSL: negative orders and (Y, K) non-positive arguments are domain errors -> NaN; J, I at x = 0
    bad = np.zeros(3)
    for name, x in (("Jnu", [-1.0, np.nan, 1.0]), ("Inu", [-1.0, np.nan, 1.0])):
        getattr(sf, f"sf_{name}")(C.c_double(-0.5 if x[2] == 1.0 else 0.5), np.array(x).ctypes.data_as(DP), 3, bad.ctypes.data_as(DP))
        assert np.isnan(bad).all()
    for name in ("Ynu", "Knu"):
        getattr(sf, f"sf_{name}")(C.c_double(0.5), np.array([0.0, -2.0, np.nan]).ctypes.data_as(DP), 3, bad.ctypes.data_as(DP))
        assert np.isnan(bad).all()
    z = np.zeros(1)
    for name, nu, want in (("Jnu", 0.0, 1.0), ("Jnu", 0.5, 0.0), ("Inu", 0.0, 1.0), ("Inu", 2.5, 0.0)):
        getattr(sf, f"sf_{name}")(C.c_double(nu), np.array([0.0]).ctypes.data_as(DP), 1, z.ctypes.data_as(DP))
        assert z[0] == want


def test_real_order_bessel_functions_at_tiny_arguments(sf):
    """x below 1e-152: the steepest-descent nodes reach |u| > 355 where sinh(u)^2 overflows (the integrand's tanh and sech are
    taken by name, not from sqrt(1 + sinh^2)): Y_0(1e-160) = -234.5..., Y_{1/2}(1e-200) = -8e99, J_0 = 1, finite wherever the
    true value is."""
    import mpmath as mp

    cases = [("Ynu", 0.0, 1e-160), ("Ynu", 0.0, 1e-250), ("Ynu", 0.5, 1e-200), ("Ynu", 0.25, 1e-300), ("Jnu", 0.0, 1e-200), ("Jnu", 0.5, 1e-160), ("Ynu", 1.5, 1e-160)]
    fn = {"Jnu": mp.besselj, "Ynu": mp.bessely}
    with mp.workdps(40):
        for name, nu, x in cases:
            out = np.zeros(1)
            getattr(sf, f"sf_{name}")(C.c_double(nu), np.array([x]).ctypes.data_as(DP), 1, out.ctypes.data_as(DP))
            want = fn[name](nu, mp.mpf(x))
            if abs(want) > 1e300:
                assert out[0] == -np.inf or abs(out[0]) > 1e300, (name, nu, x, out[0])
                continue
            assert np.isfinite(out[0]) and abs(out[0] - float(want)) <= 1e-13 * abs(float(want)), (name, nu, x, out[0], float(want))


def test_real_order_bessel_model_on_host_twin_against_mpmath():
    """A model with Bessel functions of real order (a half-integer number and a model parameter; all four kinds): the
    printer emits inflx_sf_bessel_*nu where the reference emits gsl_sf_bessel_*nu, orders shifted by differentiation
    included, and the host twin's raw values agree with a 30-digit evaluation of the same sympy expressions."""
    fields, metric, potential = example_models.bessel_real()
    model = InflationModelBuilder.new(fields, metric, potential, model_name="bessel_real", init_sympy_printing=False, silent=True, assertions=False, simplify=False).build()
    comp = Compiler(model, silent=True, link_gsl=True)
    hdr = comp._generate_hip_header()
    for f in ("inflx_sf_bessel_Jnu(", "inflx_sf_bessel_Ynu(", "inflx_sf_bessel_Inu(", "inflx_sf_bessel_Knu("):
        assert f in hdr, f
    assert "inflx_sf_bessel_Knu(args[1]" in hdr or "inflx_sf_bessel_Knu(u_" in hdr  # the order is the model parameter nu (possibly staged)
    assert comp.symbol_dict == {"phi": "x[0]", "theta": "x[1]", "m": "args[0]", "nu": "args[1]"}
    # ... and the reference's printer strings for the same functions (compiler.py:199-212)
    pr = GSLInflatoxPrinter(model.coordinates, model.coordinate_tangents)
    nu = sympy.Symbol("nu")
    assert pr.doprint(sympy.besselk(nu, model.coordinates[0])) == "gsl_sf_bessel_Knu(args[0], x[0])"
    tw = HostTwin(hdr)
    args = np.array([1.2, 2.6])
    n0, n1, ext = 14, 6, (0.4, 9.0, 0.2, 2.9)
    import oracle

    pts = oracle.grid_points(ext, n0, n1)
    got = tw.grid(4, args, ext, n0, n1).reshape(-1, 5)
    want = special.raw_values_mp(model, comp.symbol_dict, args, pts)
    scale = np.maximum(np.abs(want), np.abs(want).max(axis=0, keepdims=True) * 1e-3)
    assert np.isfinite(want).all() and (np.abs(got - want) / scale).max() < 1e-10


def _hyp(lib, name, params, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros_like(x)
    getattr(lib, f"sf_{name}")(*[C.c_double(v) for v in params], x.ctypes.data_as(DP), x.size, out.ctypes.data_as(DP))
    return out


def test_hyperg_1F1_on_host_against_mpmath(sf):
    """Every value returned is right to 1e-12 (measured: 2e-14) and none is declined for |a|, |b| <= 20, |x| <= 600."""
    import mpmath as mp

    rng = np.random.default_rng(31)
    x = np.concatenate([rng.uniform(-30, 30, 25), rng.uniform(-300, 300, 15), rng.uniform(-1, 1, 6), [0.0, 1e-8, -1e-8, 600.0, -600.0]])
    params = [(0.5, 1.5), (1.0, 2.0), (2.5, 0.7), (-0.5, 1.0), (-3.0, 2.0), (-2.7, 1.3), (4.2, -1.5), (0.1, 10.0), (-4.5, -2.5), (12.5, 3.0), (-11.3, 4.0), (20.0, 21.5)]
    params += [(float(rng.uniform(-8, 8)), float(rng.uniform(-8, 8))) for _ in range(6)]
    with mp.workdps(50):
        for a, b in params:
            for xi, g in zip(x, _hyp(sf, "1F1", (a, b), x)):
                want = mp.hyp1f1(a, b, mp.mpf(float(xi)))
                if abs(want) > 1e300 or abs(want) < 1e-300:
                    continue
                assert not np.isnan(g), (a, b, xi)
                assert abs(float((mp.mpf(float(g)) - want) / want)) < 1e-12, (a, b, xi, g, float(want))
    assert np.isnan(_hyp(sf, "1F1", (1.0, -2.0), np.array([0.5])))[0]  # b = 0, -1, -2, ...: GSL's domain error
    assert _hyp(sf, "1F1", (0.0, 3.0), np.array([7.0]))[0] == 1.0 and _hyp(sf, "1F1", (2.0, 3.0), np.array([0.0]))[0] == 1.0


def test_hyperg_2F1_on_host_against_mpmath(sf):
    """Right to 1e-12 wherever a value is returned; NaN only outside -1 <= x < 1, at the poles of c, and (declined)
    next to x = 1 when c - a - b is within 1e-3 of an integer without being one."""
    import mpmath as mp

    rng = np.random.default_rng(37)
    x = np.concatenate([rng.uniform(-1, 1, 30), rng.uniform(0.9, 0.997, 8), [-1.0, -0.999, -0.5, 0.5, 0.75, 0.9, 0.95, 0.99, 0.997, 1e-9]])
    params = [(0.5, 1.0, 1.5), (1.0, 1.0, 2.0), (0.5, 0.5, 1.0), (2.0, 3.0, 4.0), (-0.5, 1.5, 2.5), (-3.0, 2.0, 1.5), (1.5, -2.0, 0.5), (0.3, 0.7, -1.5)]
    params += [(2.0, 2.0, 4.5), (0.25, 0.75, 1.0), (3.3, -1.2, 2.1), (6.5, -4.2, 1.1), (0.1, 0.2, 7.3)] + [tuple(float(v) for v in rng.uniform(-5, 5, 3)) for _ in range(8)]
    declined = 0
    with mp.workdps(50):
        for a, b, c in params:
            for xi, g in zip(x, _hyp(sf, "2F1", (a, b, c), x)):
                want = mp.hyp2f1(a, b, c, mp.mpf(float(xi)))
                if not mp.isfinite(want) or abs(want) > 1e300 or abs(want) < 1e-300:
                    continue
                if np.isnan(g):
                    declined += 1
                    continue
                assert abs(float((mp.mpf(float(g)) - want) / want)) < 1e-12, (a, b, c, xi, g, float(want))
    assert declined == 0, declined
    assert np.isnan(_hyp(sf, "2F1", (1.0, 1.0, 2.0), np.array([1.0, 1.5, -1.5]))).all()  # GSL: |x| < 1
    assert np.isnan(_hyp(sf, "2F1", (1.0, 1.0, -1.0), np.array([0.3])))[0]
    # c - a - b an integer (elliptic integrals, logarithms...): the logarithmic connection formula right up to x = 1
    near_one = np.array([0.95, 0.999, 0.999999, 1 - 1e-12])
    with mp.workdps(50):
        for a, b, c in [(0.5, 0.5, 1.0), (1.0, 1.0, 2.0), (2.0, 3.0, 4.0), (1.5, 2.5, 6.0), (-0.5, 0.5, 1.0), (3.5, 2.5, 2.0), (1.2, -0.7, -1.5)]:
            for xi, g in zip(near_one, _hyp(sf, "2F1", (a, b, c), near_one)):
                want = mp.hyp2f1(a, b, c, mp.mpf(float(xi)))
                assert abs(float((mp.mpf(float(g)) - want) / want)) < 1e-12, (a, b, c, xi, g, float(want))
    # nearly but not exactly an integer, next to x = 1: declined, not guessed
    assert np.isnan(_hyp(sf, "2F1", (1.0, 1.0, 2.0 + 1e-7), np.array([0.9999])))[0]


def test_hypergeometric_model_on_host_twin_against_mpmath():
    fields, metric, potential = example_models.hypergeometric()
    model = InflationModelBuilder.new(fields, metric, potential, model_name="hypergeometric", init_sympy_printing=False, silent=True, assertions=False, simplify=False).build()
    comp = Compiler(model, silent=True, link_gsl=True)
    hdr = comp._generate_hip_header()
    assert "inflx_sf_hyperg_1F1(" in hdr and "inflx_sf_hyperg_2F1(" in hdr and "inflx_sf_hyperg_2F0(" in hdr
    assert comp.symbol_dict == {"phi": "x[0]", "theta": "x[1]", "m": "args[0]", "a": "args[1]", "c": "args[2]"}
    tw = HostTwin(hdr)
    args = np.array([0.9, 0.7, 2.3])
    n0, n1, ext = 12, 5, (0.3, 9.0, 0.2, 2.9)
    import oracle

    pts = oracle.grid_points(ext, n0, n1)
    got = tw.grid(4, args, ext, n0, n1).reshape(-1, 5)
    want = special.raw_values_mp(model, comp.symbol_dict, args, pts)
    scale = np.maximum(np.abs(want), np.abs(want).max(axis=0, keepdims=True) * 1e-3)
    assert np.isfinite(want).all() and (np.abs(got - want) / scale).max() < 1e-10
    phi = model.coordinates[0]
    theta, m = model.coordinates[1], sympy.Symbol("m")
    v2 = m**2 * (2 + sympy.hyper([1, sympy.Rational(5, 2)], [], -phi / 4)) * (1 + sympy.cos(theta) / 10)
    m2 = InflationModelBuilder.new(fields, metric, v2, model_name="two_f_zero", init_sympy_printing=False, silent=True, assertions=False, simplify=False).build()
    c2 = Compiler(m2, silent=True, link_gsl=True)
    h2 = c2._generate_hip_header()
    assert "inflx_sf_hyperg_2F0(" in h2
    got2 = HostTwin(h2).grid(4, np.array([1.1]), ext, 6, 3).reshape(-1, 5)
    want2 = special.raw_values_mp(m2, c2.symbol_dict, np.array([1.1]), oracle.grid_points(ext, 6, 3))
    scale2 = np.maximum(np.abs(want2), np.abs(want2).max(axis=0, keepdims=True) * 1e-3)
    assert np.isfinite(want2).all() and (np.abs(got2 - want2) / scale2).max() < 1e-10


def test_hyperg_2F0_on_host_against_mpmath(sf):
    """x < 0 (GSL's domain).  Right to 1e-12 (measured 1.5e-14) wherever a value is returned; declined (NaN) only
    when neither parameter is a non-positive integer, both are below 0.25 and |x| is too large for the asymptotic
    series."""
    import mpmath as mp

    rng = np.random.default_rng(41)
    x = -np.concatenate([10.0 ** rng.uniform(-4, 4, 20), [1e-3, 0.02, 0.03, 0.05, 0.1, 0.3, 1.0, 3.0, 10.0, 100.0]])
    params = [(0.5, 0.5), (1.0, 2.0), (2.5, -1.5), (0.3, 3.7), (4.0, 0.2), (7.5, 2.0), (20.0, 3.0), (35.0, -7.5), (-2.0, 3.3), (-3.0, -4.0), (1.0, -0.5)]
    params += [(float(rng.uniform(0.25, 12)), float(rng.uniform(-6, 12))) for _ in range(8)]
    with mp.workdps(40):
        for a, b in params:
            for xi, g in zip(x, _hyp(sf, "2F0", (a, b), x)):
                want = mp.hyp2f0(a, b, mp.mpf(float(xi)))
                assert not np.isnan(g), (a, b, xi)
                assert abs(float((mp.mpf(float(g)) - want) / want)) < 1e-12, (a, b, xi, g, float(want))
    assert np.isnan(_hyp(sf, "2F0", (1.0, 2.0), np.array([0.5])))[0]  # x > 0: GSL's domain error
    assert _hyp(sf, "2F0", (1.0, 2.0), np.array([0.0]))[0] == 1.0
    assert np.isnan(_hyp(sf, "2F0", (0.1, 0.1), np.array([-1.0])))[0]  # declined
    assert abs(_hyp(sf, "2F0", (0.1, 0.1), np.array([-0.01]))[0] / float(mp.hyp2f0(0.1, 0.1, -0.01)) - 1) < 1e-13  # asymptotic range
