"""Shared fixtures.  `-m gpu` tests need an MI355X; everything else runs on the CPU container."""

import functools
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
MODELS = ("hyperbolic", "doc", "angular", "egno", "d5")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@functools.lru_cache(maxsize=None)
def golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, f"{name}.npz")))


def _compilers():
    import oracle

    return tuple(oracle.reference_compilers())


# The C compilers that stand in for the reference's `zig cc` (oracle/model_c.py): gcc, and the clang of the ROCm image --
# the closer stand-in, since zig cc IS clang.  gcc -std=c17 does not contract a*b+c, clang does; every reference number
# exists in both flavours (goldens: key / key + "_clang"; oracle: oracle_model(name, cc)).
COMPILERS = _compilers()


def golden_key(key, cc):
    return key if cc == "gcc" else f"{key}_{cc}"


@functools.lru_cache(maxsize=None)
def oracle_model(name, cc="gcc"):
    """The CPU oracle for an example model: this repo's symbolic stage -> oracle C emitter -> `cc` (gcc or clang)."""
    import oracle
    import workloads
    from workloads import example_models

    spec = example_models.get(name)
    src, symdict = oracle.emit_c_source(workloads.model_for(name), **spec.compiler_kwargs)
    return oracle.OracleModel(oracle.compile_c_model(src, cc=cc)), symdict


def compare(got, want, rtol, what=""):
    """NaN pattern exact, +-Inf exact, finite values within rtol relative."""
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.array_equal(np.isnan(got), np.isnan(want)), f"{what}: NaN pattern differs"
    inf = np.isinf(want)
    assert np.array_equal(np.isinf(got), inf), f"{what}: Inf pattern differs"
    assert np.array_equal(got[inf], want[inf]), f"{what}: Inf signs differ"
    fin = np.isfinite(want)
    if fin.any():
        err = np.abs(got[fin] - want[fin]) / np.maximum(np.abs(want[fin]), np.finfo(float).tiny)
        assert err.max() <= rtol, f"{what}: max relative error {err.max():.3e} > {rtol:g}"
        return float(err.max())
    return 0.0


@pytest.fixture(scope="session")
def gpu_lib():
    from inflatox_amd import _native

    _native.load_library()
    assert _native.device_count() > 0, "no HIP device visible"
    return _native


def generalised_al(art):
    """``GeneralisedAL(art)`` without the constructor's random basis check.  That check draws its parameters
    and points from the OS, and for the ill-conditioned angular model it fails in ~7 % of the draws -- on the
    reference's own C just the same (300 draws through the oracle: 20 raise) -- so tests that are about
    something else construct the object deterministically; the check itself is tested further down."""
    from inflatox_amd.consistency_conditions import GeneralisedAL, InflationCondition

    al = GeneralisedAL.__new__(GeneralisedAL)
    InflationCondition.__init__(al, art, validate_basis=False)
    return al


def pytest_sessionfinish(session, exitstatus):
    """Parity statistics of the run (per tolerance.check call: excluded fraction, worst ratio to the allowance):
    the numbers the per-model KAPPA and the exclusion caps of tests/tolerance.py are set from."""
    try:
        import json

        import tolerance

        if tolerance.STATS:
            out = os.path.join(ROOT, "gpurun_out")
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, "parity_stats.json"), "w") as fh:
                json.dump(tolerance.STATS, fh, indent=0)
    except Exception:  # noqa: BLE001 -- statistics only
        pass
