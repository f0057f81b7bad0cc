"""The per-point operations pinned WITHOUT a model (src/anguelova.rs:103-170).

``inflx_ops_on_values`` pushes arbitrary (V, v00, v10, v11, |dV|^2) tuples -- every pair of special values, random bit
patterns, magnitudes that straddle every validity bound of the kernels' division spelling, v10 = 0, v00 = 0, V < 0, the
one denominator of the epilogue that can cancel to zero -- through the device functions the sweep kernels inline
(csrc/inflx_ops.h) and the test compares them with the oracle's ``op_*`` functions (oracle/sweep_oracle.c) run on the
very same tuples through ``oracle/values_model.c``:

  * the kernels' spelling of complete_analysis (divisions without special-case handling where every operand is in mid
    range, else the compiler's IEEE divisions) equals the all-IEEE spelling BIT FOR BIT on every tuple, in random and in
    sorted (wave-uniform) order;
  * NaN and +-Inf patterns equal the oracle's exactly;
  * consistency, eps_V, eps_H, omega and the three single-quantity operations are bit-equal to the oracle's
    (IEEE division, sqrt, multiplication and addition are correctly rounded on both sides, contraction is off on both);
  * delta within 2 ulp (OCML atan against glibc atan) and eta within the bound that follows from it.
"""

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

SPECIALS = np.array(
    [0.0, -0.0, np.inf, -np.inf, np.nan, 5e-324, -5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, -1.7976931348623157e308, 1e-300, 1e300, 1.0, -1.0, 2.0**-120, 2.0**121, 3.0]
)


def _tuples(seed=2024):
    rng = np.random.default_rng(seed)
    blocks = []
    # every pair of positions x every pair of specials, the other three values ordinary
    pairs = []
    for i in range(5):
        for j in range(i + 1, 5):
            for a in SPECIALS:
                for b in SPECIALS:
                    rec = rng.normal(size=5) * 10.0 ** rng.integers(-3, 4, size=5)
                    rec[i], rec[j] = a, b
                    pairs.append(rec)
    blocks.append(np.array(pairs))
    # all 5-tuples over a smaller special set
    small = np.array([0.0, -0.0, np.inf, np.nan, 5e-324, 1.0, -2.5, 1e300])
    blocks.append(np.stack(np.meshgrid(*[small] * 5, indexing="ij"), axis=-1).reshape(-1, 5))
    # random bit patterns
    blocks.append(rng.integers(0, 2**64, size=(200_000, 5), dtype=np.uint64).view(np.float64))
    # magnitudes 2^e with e uniform in [-135, 135]: straddles the [2^-120, 2^121) window of the quick spelling
    e = rng.integers(-135, 136, size=(300_000, 5))
    blocks.append(np.ldexp(1.0 + rng.random((300_000, 5)), e) * rng.choice([-1.0, 1.0], size=(300_000, 5)))
    # exactly on the window's edges
    edge = np.ldexp(1.0 + rng.random((40_000, 5)) * (rng.random((40_000, 5)) < 0.5), rng.choice([-121, -120, -119, 119, 120, 121], size=(40_000, 5)))
    blocks.append(edge * rng.choice([-1.0, 1.0], size=edge.shape))
    # ordinary magnitudes, and the cases the verdict names: v10 = 0, v00 = 0, V < 0
    ordinary = rng.normal(size=(300_000, 5)) * 10.0 ** rng.uniform(-8, 8, size=(300_000, 5))
    ordinary[:30_000, 2] = 0.0
    ordinary[30_000:60_000, 1] = 0.0
    ordinary[60_000:120_000, 0] = -np.abs(ordinary[60_000:120_000, 0])
    ordinary[120_000:130_000, 2] = -0.0
    blocks.append(ordinary)
    # values close to each other (cancellation in lhs - rhs and in the numerator of vtt)
    near = 1.0 + rng.integers(-8, 9, size=(60_000, 5)) * 2.0**-52
    blocks.append(near * 10.0 ** rng.integers(-2, 3, size=(60_000, 1)))
    # the denominator eps_V + |vtt/V| - vt2 cancelling to zero (or to next to nothing): v00 << v10 makes vt2 = eps_V,
    # and a large eps_V absorbs |vtt/V|
    n = 100_000
    v = rng.choice([-1.0, 1.0], size=n) * 10.0 ** rng.uniform(-3, 3, size=n)
    v10 = 10.0 ** rng.uniform(-3, 3, size=n) * rng.choice([-1.0, 1.0], size=n)
    v00 = v10 * 10.0 ** -rng.uniform(7, 30, size=n)
    v11 = rng.normal(size=n) * 10.0 ** rng.uniform(-30, 3, size=n)
    g = v * v * 10.0 ** rng.uniform(-5, 40, size=n)
    blocks.append(np.stack([v, v00, v10, v11, g], axis=1))
    # consistency_only's own denominator |v11/V - 3| + |3 (v00/v10)^2 + (v00/V)(v10/v00)^2| cancelling to exactly zero (0/0 = NaN)
    # and to next to nothing: v10 = v00 2^k, V = -v00 2^(4k) / 3 makes the second term exactly 0, v11 = 3 V (or an ulp or a
    # few off) the first one (the quick spelling of inflx_op_consistency_only_quick tests this denominator on its own)
    n = 60_000
    k = rng.integers(-6, 7, size=n)
    mant = rng.integers(1, 2**20, size=n).astype(np.float64)
    v00 = 3.0 * mant * 2.0 ** rng.integers(-30, 30, size=n) * rng.choice([-1.0, 1.0], size=n)
    v10 = v00 * 2.0**k
    v = -(v00 / 3.0) * 2.0 ** (4 * k)
    v11 = 3.0 * v * (1.0 + rng.integers(-3, 4, size=n) * 2.0**-52 * (rng.random(n) < 0.5))
    v10 = v10 * (1.0 + rng.integers(-2, 3, size=n) * 2.0**-52 * (rng.random(n) < 0.3))
    blocks.append(np.stack([v, v00, v10, v11, np.abs(rng.normal(size=n))], axis=1))
    vals = np.concatenate(blocks)
    # the same records once more sorted by t = |v10/v00|: wavefronts whose lanes agree about every branch of atan / tan
    with np.errstate(all="ignore"):
        t = np.abs(vals[:, 2] / vals[:, 1])
    order = np.argsort(np.where(np.isnan(t), np.inf, t), kind="stable")
    return np.ascontiguousarray(np.concatenate([vals, vals[order]]))


def _bits_equal(a, b):
    """Bit for bit, any NaN equal to any NaN."""
    return (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))


def _ulps(a, b):
    return np.abs(a.view(np.int64) - b.view(np.int64))


@pytest.fixture(scope="module")
def results(gpu_lib):
    import workloads

    _, art = workloads.artifact_for("hyperbolic")
    lib = gpu_lib.InflatoxDevLib(art.shared_object_path)
    vals = _tuples()
    assert vals.shape[0] >= 2_000_000
    quick = lib.ops_on_values(vals, ieee_only=False)
    ieee = lib.ops_on_values(vals, ieee_only=True)
    want = oracle.ops_on_values(vals, threads=16)
    return vals, quick, ieee, want


def test_kernel_spelling_equals_the_ieee_spelling_bit_for_bit(results):
    vals, quick, ieee, _ = results
    same = _bits_equal(quick, ieee)
    bad = np.argwhere(~same)
    assert bad.size == 0, f"{len(bad)} values differ, first: tuple {vals[bad[0][0]]!r} output {bad[0][1]}: quick {quick[tuple(bad[0])]!r} ieee {ieee[tuple(bad[0])]!r}"
    # the test is not vacuous: a good share of the tuples takes the quick spelling (all five values normal, in the window)
    ex = np.frexp(vals)[1]
    inside = np.isfinite(vals).all(axis=1) & (vals != 0).all(axis=1) & (ex.min(axis=1) >= -119) & (ex.max(axis=1) <= 121)
    assert 0.2 < inside.mean() < 0.9


def test_nan_and_inf_patterns_equal_the_oracle(results):
    vals, _, ieee, want = results
    nan_same = np.isnan(ieee) == np.isnan(want)
    # eta = omega * tan(delta) - 3 may overflow on one side only when omega * tan(delta) sits at the very edge of the range
    k = np.argwhere(~nan_same)
    assert k.size == 0, f"NaN pattern differs at {len(k)} values, first: tuple {vals[k[0][0]]!r} output {k[0][1]}: gpu {ieee[tuple(k[0])]!r} oracle {want[tuple(k[0])]!r}"
    inf_w = np.isinf(want)
    cols = [0, 1, 2, 4, 5, 6, 7, 8]
    assert np.array_equal(np.isinf(ieee)[:, cols], inf_w[:, cols])
    assert np.array_equal(ieee[:, cols][inf_w[:, cols]], want[:, cols][inf_w[:, cols]])
    # eta: infinite on both sides or, where tan(delta) differs by an ulp next to the overflow threshold, huge on the other
    e_g, e_w = ieee[:, 3], want[:, 3]
    mism = np.isinf(e_g) != np.isinf(e_w)
    assert (np.abs(np.where(np.isinf(e_g), e_w, e_g))[mism] > 1e300).all()
    both = np.isinf(e_g) & np.isinf(e_w)
    assert np.array_equal(e_g[both], e_w[both])


def test_division_only_outputs_are_bit_equal_to_the_oracle(results):
    vals, _, ieee, want = results
    names = {0: "consistency", 1: "epsilon_V", 2: "epsilon_H", 5: "omega", 6: "consistency_only", 7: "consistency_rapidturn_only", 8: "epsilon_v_only"}
    for col, name in names.items():
        same = _bits_equal(ieee[:, col], want[:, col])
        bad = np.flatnonzero(~same)
        assert bad.size == 0, f"{name}: {bad.size} values differ, first tuple {vals[bad[0]]!r}: gpu {ieee[bad[0], col]!r} oracle {want[bad[0], col]!r}"


def test_delta_and_eta_within_the_libm_bound(results):
    vals, _, ieee, want = results
    d_g, d_w = ieee[:, 4], want[:, 4]
    fin = np.isfinite(d_w)
    assert _ulps(d_g[fin], d_w[fin]).max() <= 2, "delta: OCML atan and glibc atan differ by more than 2 ulp"
    # eta = omega * tan(delta) - 3: omega is bit-equal; delta differs by <= 2 ulp, which the tangent amplifies by
    # delta (1 + t^2) / t; the two tangents add an ulp or two of their own
    with np.errstate(all="ignore"):
        t = np.tan(d_w)
        amp = 1.0 + d_w * (1.0 + t * t) / np.maximum(t, np.finfo(float).tiny)
        product = np.abs(want[:, 5] * t)
        allowed = 8.0 * np.finfo(float).eps * amp * product + 8.0 * np.finfo(float).eps * np.abs(want[:, 3])
    ok = np.isfinite(want[:, 3]) & np.isfinite(ieee[:, 3]) & np.isfinite(allowed)
    excess = np.abs(ieee[ok, 3] - want[ok, 3]) - allowed[ok]
    worst = int(np.argmax(excess))
    assert excess.max() <= 0, f"eta differs by more than the bound: tuple {vals[ok][worst]!r} gpu {ieee[ok, 3][worst]!r} oracle {want[ok, 3][worst]!r}"
    assert ok.mean() > 0.2
