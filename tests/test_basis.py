"""Basis validation (reference src/lib.rs:141-300): the oracle's restatement
against the goldens produced from the reference's own C, and the emitted ``inflx_basis_point`` run on
the host against the oracle.  CPU only; the device kernel is covered in test_parity_gpu.py."""

import os

import numpy as np
import pytest

import tolerance as tol
from oracle import cpu_oracle

from conftest import GOLDEN_DIR, MODELS, oracle_model
from host_twin import HostTwin
import workloads
from workloads import example_models
from inflatox_amd.compiler import Compiler

BASIS = dict(np.load(os.path.join(GOLDEN_DIR, "basis.npz")))
SETS = (("inside_x", None, "inside_basis"), ("unit_x", "unit_p", "unit_basis"), ("unit_x", None, "unit_basis_args"))


@pytest.mark.parametrize("name", MODELS)
def test_oracle_basis_matches_reference_goldens(name):
    """This repo's symbolic stage -> oracle C emitter -> gcc gives the numbers of the reference's pipeline."""
    om, _ = oracle_model(name)
    for xk, pk, bk in SETS:
        p = BASIS[f"{name}_{pk}"] if pk else BASIS[f"{name}_args"]
        got = cpu_oracle.basis_on_points(om.path, p, BASIS[f"{name}_{xk}"])
        assert np.array_equal(got, BASIS[f"{name}_{bk}"], equal_nan=True), (name, bk)


@pytest.mark.parametrize("name", MODELS)
def test_emitted_basis_function_on_host(name):
    spec = example_models.get(name)
    hdr = Compiler(workloads.model_for(name), silent=True, **spec.compiler_kwargs)._generate_hip_header()
    tw = HostTwin(hdr)
    for xk, pk, bk in SETS:
        p = BASIS[f"{name}_{pk}"] if pk else BASIS[f"{name}_args"]
        x, want = BASIS[f"{name}_{xk}"], BASIS[f"{name}_{bk}"]
        tol.basis_close(tw.basis(p, x), want, f"{name}/{bk}", tol.basis_sensitivity(name, p, x, want))


@pytest.mark.parametrize("name", MODELS)
def test_example_models_are_orthonormal_inside_their_extent(name):
    b = BASIS[f"{name}_inside_basis"]
    failed = cpu_oracle.check_basis(b, BASIS[f"{name}_inside_x"], 1e-3)
    assert failed < len(b)


def test_check_basis_flags_defects_like_the_reference():
    pts = np.zeros((3, 2))
    good = np.tile([1.0, 0.0, 1.0, 1, 0, 0, 1], (3, 1))
    assert cpu_oracle.check_basis(good, pts, 1e-3) == 0
    bad = good.copy()
    bad[1, 0] = 1.002  # |v|^2 off by more than the accuracy -> BasisNorm for vector 0
    with pytest.raises(cpu_oracle.BasisDefect) as e:
        cpu_oracle.check_basis(bad, pts, 1e-3)
    assert (e.value.kind, e.value.vectors) == ("norm", (0,))
    bad = good.copy()
    bad[2, 1] = -0.01  # v.w1 -> BasisOth
    with pytest.raises(cpu_oracle.BasisDefect) as e:
        cpu_oracle.check_basis(bad, pts, 1e-3)
    assert (e.value.kind, e.value.vectors) == ("oth", (0, 1))
    # not-normal inner products are warnings (counted), not errors; a zero overlap is fine, a zero norm is not
    odd = good.copy()
    odd[0, 0] = np.nan
    odd[1, 2] = 0.0
    odd[2, 1] = np.inf
    assert cpu_oracle.check_basis(odd, pts, 1e-3) == 3
    # the tests run in the reference's order: the norm of v is looked at before the overlap
    both = good.copy()
    both[0, 1] = 0.5
    both[0, 2] = 2.0
    with pytest.raises(cpu_oracle.BasisDefect) as e:
        cpu_oracle.check_basis(both, pts, 1e-3)
    assert e.value.kind == "oth"


def test_domain_points_follow_the_reference_walk():
    """src/lib.rs:247-256: `point[axis] = stop + spacing * idx`, other coordinate at its start."""
    a, b = cpu_oracle.domain_points([4, 2], [[0.0, 2.0], [10.0, 11.0]])
    assert np.array_equal(a, [[2.0, 10.0], [2.5, 10.0], [3.0, 10.0], [3.5, 10.0]])
    assert np.array_equal(b, [[0.0, 11.0], [0.0, 11.5]])
