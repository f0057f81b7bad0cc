"""Transpiler: printer rules, parameter numbering, and exactness of the axis staging (CPU only).

The staged HIP header is compiled for the host (tests/host_twin.cpp includes the very same headers
the kernels are built from) and compared with the oracle, so the transpiler is checked on every
`-m "not gpu"` run.
"""

import json
import os

import numpy as np
import pytest
import sympy
import tolerance as tol
from conftest import GOLDEN_DIR, MODELS, compare, golden, oracle_model
from host_twin import HostTwin

import oracle
import workloads
from workloads import example_models
from inflatox_amd.compiler import CInflatoxPrinter, CompilationArtifact, Compiler

SYMBOLS = json.load(open(os.path.join(GOLDEN_DIR, "symbols.json")))


def header_for(name, **kw):
    spec = example_models.get(name)
    kwargs = dict(spec.compiler_kwargs)
    kwargs.update(kw)
    c = Compiler(workloads.model_for(name), silent=True, **kwargs)
    return c, c._generate_hip_header()


def test_printer_known_answers():
    """The strings the reference's own test-suite pins (tests/test_compiler.py:40-53)."""
    x, y, a, b, xd, yd = sympy.symbols("x y a b \\dot{{x}} \\dot{{y}}")
    pr = CInflatoxPrinter([x, y], [xd, yd])
    kat = SYMBOLS["_printer_kat"]
    assert pr._print_Symbol(x) == kat["x"] == "x[0]"
    assert pr._print_Symbol(y) == kat["y"] == "x[1]"
    assert pr._print_Symbol(a) == kat["a"] == "args[0]"
    assert pr._print_Symbol(b) == kat["b"] == "args[1]"
    assert pr._print_Symbol(xd) == kat["xdot"] == "xdot[0]"
    assert pr._print_Symbol(yd) == kat["ydot"] == "xdot[1]"
    assert pr.doprint(x**2 + y) == kat["x**2 + y"] == "pow(x[0], 2) + x[1]"
    assert pr.doprint(x * y) == kat["x*y"] == "x[0]*x[1]"
    assert pr.doprint(sympy.sqrt(a) * y) == kat["sqrt(a)*y"] == "sqrt(args[0])*x[1]"
    assert pr.doprint(sympy.sin(x)) == kat["sin(x)"] == "sin(x[0])"


@pytest.mark.parametrize("name", MODELS)
def test_symbol_table_matches_reference(name):
    """Parameter numbering is the reference's (order of first appearance in ITS print order)."""
    c, _ = header_for(name)
    want = SYMBOLS[name]
    assert c.symbol_dict == want["symbol_dictionary"]
    spec = example_models.get(name)
    assert len(c.symbol_dict) - 2 == want["n_parameters"] == len(spec.args)
    # the oracle's independent restatement of the printer agrees as well
    _, symdict = oracle_model(name)
    assert symdict == want["symbol_dictionary"]


@pytest.mark.parametrize("name", MODELS)
def test_staged_header_on_host_matches_oracle(name):
    c, hdr = header_for(name)
    tw = HostTwin(hdr)
    g = golden(name)
    assert tw.n_parameters == len(g["args"])
    for tag in ("g16", "g64"):
        n0, n1 = (int(v) for v in g[f"{tag}_shape"])
        ext = g[f"{tag}_extent"]
        pts = oracle.grid_points(ext, n0, n1)
        env, flaky = tol.reference_error(name, g["args"], pts)
        env, flaky = env.reshape(n0, n1, 5), flaky.reshape(n0, n1, 5)
        raw = tw.grid(4, g["args"], ext, n0, n1)
        tol.check(raw, g[f"{tag}_raw"], tol.allowance_raw(g[f"{tag}_raw"], env), flaky, f"{name}/{tag}/raw", model=name)
        out = tw.grid(0, g["args"], ext, n0, n1)
        tol.check(out, g[f"{tag}_out"], tol.allowance_derived(g[f"{tag}_raw"], env, tol.epilogue), flaky.any(axis=-1)[..., None], f"{name}/{tag}/out", model=name)
        if name in ("hyperbolic", "doc"):
            compare(out, g[f"{tag}_out"], 1e-13, f"{name}/{tag}/out strict")


@pytest.mark.parametrize("name", MODELS)
def test_staging_does_not_change_a_single_bit(name):
    """Sharing identical nodes and moving whole sub-expressions between stages is exact: the staged
    and the unstaged (everything per point) programs must agree bit for bit on the host."""
    _, staged = header_for(name)
    _, plain = header_for(name, staged=False)
    a, b = HostTwin(staged), HostTwin(plain)
    g = golden(name)
    n0, n1 = (int(v) for v in g["g64_shape"])
    for op in (4, 0):
        x = a.grid(op, g["args"], g["g64_extent"], n0, n1)
        y = b.grid(op, g["args"], g["g64_extent"], n0, n1)
        assert np.array_equal(x, y, equal_nan=True), name
    rng = np.random.default_rng(3)
    ext = g["g64_extent"]
    pts = np.column_stack([rng.uniform(ext[0], ext[1], 200), rng.uniform(ext[2], ext[3], 200)])
    assert np.array_equal(a.trajectory(0, g["args"], pts), b.trajectory(0, g["args"], pts), equal_nan=True)


@pytest.mark.parametrize("name", ["doc", "egno", "d5"])
def test_fast_mode_is_close_but_not_exact(name):
    """``Compiler(regroup=True)`` re-associates products (row/column/parameter factors are combined and
    divided once per row/column instead of once per point).  It is an opt-in fast mode, NOT the parity
    path: rounding differs from the reference, so it is only required to agree to 1e-9 at the typical
    point and to reproduce the NaN pattern wherever the reference's own NaN-ness is robust."""
    c, hdr = header_for(name, regroup=True)
    tw = HostTwin(hdr)
    g = golden(name)
    n0, n1 = (int(v) for v in g["g64_shape"])
    ext = g["g64_extent"]
    env, flaky = tol.reference_error(name, g["args"], oracle.grid_points(ext, n0, n1))
    flaky = flaky.reshape(n0, n1, 5) | ~np.isfinite(env.reshape(n0, n1, 5))
    raw = tw.grid(4, g["args"], ext, n0, n1)
    ref = g["g64_raw"]
    firm = ~flaky
    assert np.array_equal(np.isnan(raw)[firm], np.isnan(ref)[firm])
    ok = np.isfinite(raw) & np.isfinite(ref) & firm
    allowed = tol.allowance_raw(ref, env.reshape(n0, n1, 5))
    with np.errstate(all="ignore"):
        inside = np.abs(raw - ref)[ok] <= allowed[ok]
    assert inside.mean() >= 0.98, (name, float(inside.mean()))
    _, exact_hdr = header_for(name)

    def point_divisions(h):
        # the IEEE variant of the point stage (or the only one): rational literals such as 1.0/3.0 are not
        # divisions, a quotient by a hoisted or shared reciprocal is one
        start = h.index("void inflx_stage_point_ieee(") if "inflx_stage_point_ieee" in h else h.index("void inflx_stage_point(")
        body = h[start : h.index("}\n", start)]
        return body.count("/") - body.count(".0/") + body.count("INFLX_DIVH(") + body.count("INFLX_DIVS(") + body.count("INFLX_DIVI(")

    if name != "doc":
        assert point_divisions(hdr) < point_divisions(exact_hdr), "fast mode should divide less often per grid point"


@pytest.mark.parametrize("name", MODELS)
def test_flag_quantum_dif_on_host_matches_oracle(name):
    """ops::flag_quantum_diff (src/anguelova.rs:166-170) through the staged header vs the oracle."""
    _, hdr = header_for(name)
    tw = HostTwin(hdr)
    om, _ = oracle_model(name)
    g = golden(name)
    n0, n1 = (int(v) for v in g["g64_shape"])
    ext = g["g64_extent"]
    for accuracy in (1e-3, 0.5, 0.9):
        tw.set_accuracy(accuracy)
        got = tw.grid(5, g["args"], ext, n0, n1) != 0
        want = om.grid_sweep(oracle.OP.QDIF, g["args"], ext, n0, n1, accuracy=accuracy)
        if f"g64_qdif_{accuracy}" in g:
            assert np.array_equal(want, g[f"g64_qdif_{accuracy}"]), "oracle vs reference-generated golden"
        # a component within a few ulps of the threshold may legitimately flip; nothing else may
        assert (got != want).mean() <= 0.002, (name, accuracy, int((got != want).sum()))


def test_axis_classification():
    c, hdr = header_for("hyperbolic")
    assert c.stage_info["out_mask"] == 1 and c.stage_info["nc"] == 0  # nothing depends on x[1]
    assert c.stage_info["out_masks"][:5] == [1, 0, 0, 1, 1]  # V(x0), v00 = m^2, v10 = 0, v11(x0), g(x0)
    c, hdr = header_for("d5")
    assert c.stage_info["out_mask"] == 3
    assert c.stage_info["statements"]["3"] < 80, "D5's per-point stage should be small once r-only work is hoisted"
    assert "INFLX_FN void inflx_stage_point" in hdr


def test_reference_constants_quirk():
    """The reference computes with 12-digit M_PI (its own fallback macros, compiler.py:72-88)."""
    _, hdr = header_for("d5")
    assert "#define INFLX_M_PI 3.14159265359\n" in hdr
    _, hdr = header_for("d5", exact_constants=True)
    assert "#define INFLX_M_PI 3.14159265358979323846\n" in hdr


def test_power_strength_reduction():
    c, hdr = header_for("doc")
    assert "inflx_ipow<4>(x0)" in hdr and "pow(" not in hdr.replace("inflx_ipow", "").replace("inflx_hpow", "")
    c, hdr = header_for("egno")
    assert "pow(" in hdr.replace("inflx_ipow", "").replace("inflx_hpow", ""), "symbolic exponent 3*alpha needs the generic pow"


def test_long_double_instrument_agrees_with_50_digit_arithmetic():
    """The allowance is built on x87 extended precision; validate that instrument against the
    50-digit mpmath values stored with the goldens (tests/golden/make_golden.py: mp_truth).

    Away from a model's singular lines the two must agree to 10% of the reference's own error or
    1e-11 relative, whichever is larger -- an order of magnitude below anything the allowance has to
    resolve (RTOL = 1e-10).  The 1e-11 slack covers coefficients: the C text carries them as float64
    literals (2.0/3.0 is a float64 division even in the extended-precision build), mpmath carries
    exact rationals, and an ill-conditioned model amplifies that 1e-16 difference.  Points on a
    model's singular lines (D5: theta = k*pi/2, where a ~1e-16 residue of sin/cos multiplies 1/0-like
    factors and every finite-precision evaluation returns a different artefact) are excluded here."""
    for name in MODELS:
        g = golden(name)
        n0, n1 = (int(v) for v in g["g16_shape"])
        pts = oracle.grid_points(g["g16_extent"], n0, n1)
        om, ld_path = tol._models(name)
        t_ld = oracle.raw_long_double(ld_path, g["args"], pts).reshape(n0, n1, 5)
        t_mp = g["g16_raw_mp"]
        a = g["g16_raw"]
        ok = np.isfinite(t_ld) & np.isfinite(t_mp) & np.isfinite(a)
        with np.errstate(all="ignore"):
            instrument_err = np.abs(t_ld - t_mp)
            reference_err = np.abs(a - t_mp)
            regular = ok & (instrument_err <= 1e-3 * np.abs(t_mp))
        assert regular.sum() >= 0.75 * ok.sum(), name
        assert np.all(instrument_err[regular] <= 0.1 * reference_err[regular] + 1e-11 * np.abs(t_mp[regular])), name


def test_compiler_front_end_contract():
    spec = example_models.get("hyperbolic")
    model = workloads.model_for("hyperbolic")
    art = Compiler(model, silent=True).compile()
    assert isinstance(art, CompilationArtifact)
    assert art.n_fields == 2 and art.n_parameters == 3
    phi, theta = spec.fields
    assert art.lookup_symbol(phi) == "x[0]" and art.lookup_symbol(theta) == "x[1]"
    assert art.lookup_symbol(sympy.Symbol("L")) == "args[2]"
    path = art.shared_object_path
    assert os.path.getsize(path) > 0
    del art
    assert not os.path.exists(path), "auto_cleanup removes the artefact with the object (compiler.py:247-250)"
    keep = Compiler(model, silent=True, cleanup=False).compile()
    p2 = keep.shared_object_path
    del keep
    assert os.path.exists(p2)
    os.remove(p2)
    assert Compiler(model, link_gsl=True, silent=True).gsl is True  # accepted: sets USE_GSL (tests/test_special_functions.py)


def test_hoisted_reciprocal_division_is_the_ieee_quotient(tmp_path):
    """inflx_div_by_hoisted(a, b, RN(1/b)) against a/b, bit for bit: 20 million random quotients of moderate
    exponents, 2 million arbitrary bit patterns (denormals, overflow, NaN), every pair of special values,
    and operands with significands next to 1 and 2 (200 million + 20 million were run once: no mismatch)."""
    import subprocess

    exe = tmp_path / "div_hoisted_host"
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "inflatox_amd", "csrc")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "div_hoisted_host.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-mfma", "-ffp-contract=off", f"-I{csrc}", src, "-o", str(exe)], check=True)
    proc = subprocess.run([str(exe), "20"], capture_output=True, text=True)
    assert proc.returncode == 0, proc.stdout[-2000:]
    assert proc.stdout.count(" 0 mismatches") == 3, proc.stdout


@pytest.mark.parametrize("name", ["d5", "egno", "doc"])
def test_hoisted_reciprocals_are_used_and_change_nothing(name):
    """The per-point divisions by row/column/sweep-only denominators go through inflx_div_by_hoisted, and the
    program with them is bit-identical to the one with plain divisions on the golden grid and random points."""
    forced, with_h = header_for(name, hoist_reciprocals=True, share_reciprocals=True)
    _, without = header_for(name, hoist_reciprocals=False)
    assert "INFLX_DIVH(" in with_h and "inflx_stage_point_quick" in with_h and "INFLX_DIVH" not in without and "INFLX_DIVS" not in without
    # the automatic choice (hoist_reciprocals=None, the default): on where enough instructions leave the point stage
    _, auto = header_for(name)
    plain, with_plain = header_for(name, hoist_reciprocals=True)  # without shared per-point reciprocals (the default)
    gain = Compiler.quick_point_gain(plain.stage_info)
    # ... and below that bar: the same quotients checking themselves in the one point stage where there are enough of them (round 6)
    inline_auto, with_inline = header_for(name, hoist_reciprocals="inline")
    few = inline_auto.stage_info["inline_quotients"] < Compiler.INLINE_MIN_QUOTIENTS
    assert auto == (with_plain if gain >= Compiler.HOIST_MIN_GAIN else (without if few else with_inline)) and "= INFLX_RCPN(" not in with_plain
    assert (gain >= Compiler.HOIST_MIN_GAIN) == {"d5": True, "egno": False, "doc": False}[name], gain
    assert few == {"d5": False, "egno": False, "doc": True}[name]
    # the point stage's square roots take the guarded spelling exactly where the quick stage exists (quick_sqrt=None), or on request
    roots = {"d5": 3, "egno": 4, "doc": 1}[name]
    assert with_h.count("INFLX_SQRT(x)") == 2 and forced.stage_info["quick_square_roots"] == roots
    assert "INFLX_SQRT" not in without and "INFLX_HPOW" not in without
    asked, on_request = header_for(name, hoist_reciprocals=False, quick_sqrt=True)
    assert asked.stage_info["quick_square_roots"] == roots and asked.stage_info["hoisted_quotients"] == 0 and "inflx_stage_point_quick" in on_request
    never, _ = header_for(name, hoist_reciprocals=True, quick_sqrt=False)
    assert never.stage_info["quick_square_roots"] == 0 and never.stage_info["hoisted_quotients"] > 0
    if name == "d5":
        assert "INFLX_DIVH_PURE(" in with_h and "r_flag" in with_h and "INFLX_RANGE_CHECK(u_flag + r_flag + c_flag)" in with_h
        # four per-point denominators serve two or more quotients each: one refined reciprocal per denominator
        assert with_h.count("= INFLX_RCPN(") == 2 * 4 and forced.stage_info["shared_quotients"] >= 8
    a, b = HostTwin(with_h), HostTwin(without)
    g = golden(name)
    n0, n1 = (int(v) for v in g["g64_shape"])
    for op in (4, 0):
        assert np.array_equal(a.grid(op, g["args"], g["g64_extent"], n0, n1), b.grid(op, g["args"], g["g64_extent"], n0, n1), equal_nan=True)
    ext = example_models.get(name).extent
    assert np.array_equal(a.grid(0, g["args"], ext, 150, 130), b.grid(0, g["args"], ext, 150, 130), equal_nan=True)
    # round 6: the same quotients checking themselves in ONE point stage (hoist_reciprocals="inline")
    inline, with_i = header_for(name, hoist_reciprocals="inline")
    assert "INFLX_DIVI(" in with_i and "inflx_stage_point_quick" not in with_i and "#define INFLX_HAS_QUICK_POINT 0" in with_i
    assert inline.stage_info["inline_quotients"] == plain.stage_info["hoisted_quotients"] and inline.stage_info["quick_square_roots"] == 0
    c = HostTwin(with_i)
    for op in (4, 0):
        assert np.array_equal(c.grid(op, g["args"], g["g64_extent"], n0, n1), b.grid(op, g["args"], g["g64_extent"], n0, n1), equal_nan=True)
    assert np.array_equal(c.grid(0, g["args"], ext, 150, 130), b.grid(0, g["args"], ext, 150, 130), equal_nan=True)


def test_compiler_rejects_unknown_round6_options():
    m = workloads.model_for("hyperbolic")
    for bad in (dict(kernel_groups="everything"), dict(kernel_groups=["core", "nonsense"]), dict(contraction="fast")):
        with pytest.raises(ValueError):
            Compiler(m, silent=True, **bad)
    assert Compiler(m, silent=True, kernel_groups=["raw", "stats"]).kernel_groups == 1 | 64 | 2
    assert Compiler(m, silent=True, kernel_groups="all").kernel_groups == 511 and Compiler(m, silent=True).kernel_groups == 1


def test_abs_quotient_sign_identity_used_by_the_epilogue():
    """csrc/inflx_ops.h forms |vtt| / v (anguelova.rs:125) as copysign(|vtt / v|, v) to save one IEEE division:
    exact, because rounding to nearest is symmetric in the sign -- random operands over the whole exponent range,
    every pair of special values, bit for bit (NaN = NaN)."""
    rng = np.random.default_rng(3)
    bits = rng.integers(0, 2**64, size=2_000_000, dtype=np.uint64)
    a, b = bits[:1_000_000].view(np.float64), bits[1_000_000:].view(np.float64)
    sp = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 5e-324, -5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, -1.7976931348623157e308, 3.0, 1 / 3])
    a = np.concatenate([a, np.repeat(sp, len(sp))])
    b = np.concatenate([b, np.tile(sp, len(sp))])
    with np.errstate(all="ignore"):
        want = np.abs(a) / b
        got = np.copysign(np.abs(a / b), b)
    same = (want.view(np.uint64) == got.view(np.uint64)) | (np.isnan(want) & np.isnan(got))
    assert same.all(), (a[~same][:5], b[~same][:5])


def test_content_tag_ignores_comments_in_the_kernel_sources():
    """The cache tag of a code object (which profiles are stamped with) hashes the kernel sources without comments and
    blank lines: string literals survive, code changes change it."""
    from inflatox_amd.compiler import _code_only

    a = 'int f() { return 1; }  // one\n/* block\n comment */\nconst char* s = "a//b /* c */";\n\n'
    b = '// header\nint f() { return 1; }\nconst char* s = "a//b /* c */";  /* x */\n'
    assert _code_only(a) == _code_only(b)
    assert '"a//b /* c */"' in _code_only(a)
    assert _code_only(a) != _code_only(a.replace("return 1", "return 2"))
    # character literals do not flip the in-string parity, digit separators are not literals
    c = "char q = '\"'; // gone\nconst char* s = \"// kept\"; char e = '\\''; int n = 1'000; // gone too\n"
    assert _code_only(c) == "char q = '\"';\nconst char* s = \"// kept\"; char e = '\\''; int n = 1'000;"


@pytest.mark.parametrize("name", MODELS)
def test_staged_header_on_host_matches_oracle_for_random_parameters(name):
    """The CPU side of tests/test_parity_gpu.py::test_random_parameter_vectors_match_the_oracle: the staged header
    (value numbering, hoisted reciprocals where they are on, range flags) evaluated on the host for seeded parameter
    vectors against the oracle."""
    _, hdr = header_for(name)
    tw = HostTwin(hdr)
    om, _ = oracle_model(name)
    spec = example_models.get(name)
    rng = np.random.default_rng(20260 + len(name))
    x0a, x0b, x1a, x1b = spec.extent
    ext = (x0a + 0.021 * (x0b - x0a), x0b - 0.013 * (x0b - x0a), x1a + 0.017 * (x1b - x1a), x1b - 0.019 * (x1b - x1a))
    n0, n1 = 23, 41
    pts = oracle.grid_points(ext, n0, n1)
    for trial in range(3):
        args = np.asarray(spec.args, dtype=np.float64) * rng.uniform(0.7, 1.4, size=len(spec.args))
        env, flaky = tol.reference_error(name, args, pts)
        env, flaky = tol.neighbourhood_envelope(env.reshape(n0, n1, 5)), flaky.reshape(n0, n1, 5)
        raw = om.grid_sweep(oracle.OP.RAW, args, ext, n0, n1)
        want = om.grid_sweep(oracle.OP.COMPLETE, args, ext, n0, n1)
        tol.check(tw.grid(4, args, ext, n0, n1), raw, tol.allowance_raw(raw, env, name), flaky, f"{name}/random {trial}/raw", model=name)
        tol.check(tw.grid(0, args, ext, n0, n1), want, tol.allowance_derived(raw, env, tol.epilogue, name), flaky.any(axis=-1)[..., None], f"{name}/random {trial}/out", model=name)


@pytest.mark.parametrize("name", ["egno", "d5"])
def test_profile_guided_regrouping_measures_which_values_qualify(name):
    """Compiler(regroup="auto", sample=(args, extent)): the host instrument (inflatox_amd/_instrument.py) evaluates the
    reference's form in float64 and in extended precision and the regrouped form in float64 on a sample of the workload
    and clears a model value only if the regrouped form stays within 1e-10 relative + 4x the reference form's own
    rounding error at every sample point.  EGNO: all five values.  D5: v10 is exact in the reference's form at theta =
    k pi/4 (cancelling terms equal bit for bit) and v11 differs at a singular point -- both keep the reference's
    arithmetic, bit for bit, while V, v00 and |dV|^2 are regrouped."""
    from inflatox_amd import _instrument

    if _instrument.host_compiler() is None or _instrument.HostModel(header_for("hyperbolic")[1], long_double=True).mantissa_bits <= 53:
        pytest.skip("no host compiler / no extended-precision long double on this host")
    spec = example_models.get(name)
    c, hdr = header_for(name, regroup="auto", sample=(spec.args, spec.extent))
    chosen = set(c.stage_info["regrouped"])
    assert chosen == ({"V", "v00", "v10", "v11", "g"} if name == "egno" else {"V", "v00", "g"})
    _, exact = header_for(name)
    a, b = HostTwin(hdr), HostTwin(exact)
    g = golden(name)
    n0, n1 = (int(v) for v in g["g64_shape"])
    x, y = a.grid(4, g["args"], g["g64_extent"], n0, n1), b.grid(4, g["args"], g["g64_extent"], n0, n1)
    for k, value in enumerate(("V", "v00", "v10", "v11", "g")):
        same = np.array_equal(x[..., k], y[..., k], equal_nan=True)
        if value not in chosen:
            assert same, f"{value} was not cleared for regrouping but differs from the default build"
    assert not np.array_equal(x, y, equal_nan=True)  # something was regrouped
    # and the regrouped build meets the parity criterion of the default one on the golden grid
    env, flaky = tol.reference_error(name, g["args"], oracle.grid_points(g["g64_extent"], n0, n1))
    env, flaky = env.reshape(n0, n1, 5), flaky.reshape(n0, n1, 5)
    tol.check(x, g["g64_raw"], tol.allowance_raw(g["g64_raw"], env, name), flaky, f"{name}/g64/raw (regroup=auto)", model=name)
    with pytest.raises(ValueError, match="sample"):
        Compiler(workloads.model_for(name), silent=True, regroup="auto")


def test_an_artefact_can_rebuild_itself_profile_guided():
    """``CompilationArtifact.profile_guided(args, extent)`` = ``Compiler(model, <same arguments>, regroup="auto", sample=...)``:
    the code object workloads.artifact_for(name, tuned=True) builds by hand, found in the cache by its content tag."""
    import workloads

    spec, art = workloads.artifact_for("doc")
    _, by_hand = workloads.artifact_for("doc", tuned=True)
    again = art.profile_guided(spec.args, spec.extent)
    assert again.header_path == by_hand.header_path and again.stage_info["regrouped"] == by_hand.stage_info["regrouped"] != []
    assert not art.stage_info["regrouped"]
    from inflatox_amd.compiler import CompilationArtifact

    with pytest.raises(ValueError):
        CompilationArtifact({}, "/nonexistent", 2, 0, auto_cleanup=False).profile_guided(spec.args, spec.extent)


@pytest.mark.parametrize("name", ["angular", "d5"])
def test_expression_contraction_follows_the_references_clang_build(name):
    """The reference compiles its C with `zig cc` = clang, which fuses a*b + c inside an expression; the kernels are compiled by the
    same clang with the same rule, statement by statement.  With Compiler(contraction="expression") (the default since round 6) a
    product the stager made a variable of is spelled out in the statement of the sum it feeds, so that the same multiply-adds fuse:
    the generated header, built for the host by clang with -ffp-contract=on, reproduces more of the clang-built reference's bits than
    the "statement" style does, fewer of the gcc-built one's -- and where the two builds of the reference disagree about NaN (D5, on
    its singular lines) it takes clang's side at every such value, where the "statement" style took gcc's."""
    import shutil

    from host_twin import HostTwin

    if not (shutil.which("clang++") or os.path.exists("/opt/rocm/lib/llvm/bin/clang++")):
        pytest.skip("no clang++ for a host build with -ffp-contract=on")
    spec = example_models.get(name)
    g = golden(name)
    tally = {}
    for style in ("statement", "expression"):
        comp = Compiler(workloads.model_for(name), silent=True, contraction=style, **spec.compiler_kwargs)
        twin = HostTwin(comp._generate_hip_header(), contract="on", cxx="clang++")
        t = dict(eq_gcc=0, eq_clang=0, like_gcc=0, like_clang=0, disagreements=0)
        for tag in ("g16", "g64"):
            n0, n1 = (int(v) for v in g[f"{tag}_shape"])
            got = twin.grid(4, g["args"], g[f"{tag}_extent"], n0, n1)  # INFLX_OP_RAW
            a, b = g[f"{tag}_raw"], g[f"{tag}_raw_clang"]
            same = lambda x, y: (x == y) | (np.isnan(x) & np.isnan(y))  # noqa: E731
            nd = np.isnan(a) != np.isnan(b)
            t["eq_gcc"] += int(same(got, a).sum())
            t["eq_clang"] += int(same(got, b).sum())
            t["disagreements"] += int(nd.sum())
            t["like_gcc"] += int((np.isnan(got) == np.isnan(a))[nd].sum())
            t["like_clang"] += int((np.isnan(got) == np.isnan(b))[nd].sum())
        tally[style] = t
    st, ex = tally["statement"], tally["expression"]
    assert ex["eq_clang"] > st["eq_clang"] and ex["eq_gcc"] < st["eq_gcc"], tally
    assert ex["eq_clang"] > ex["eq_gcc"], tally
    if ex["disagreements"]:
        assert ex["like_clang"] == ex["disagreements"] and ex["like_gcc"] == 0, tally
        assert st["like_gcc"] == st["disagreements"], tally
    assert Compiler(workloads.model_for(name), silent=True).contraction == "expression"
