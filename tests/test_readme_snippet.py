"""The usage snippet of the reference's README (README.md:50-90) with only the import changed.

BASELINE.json cites that snippet as the definition of the headline model (configs[0], configs[1]).  It uses the names the
reference had before its builder was renamed (``SymbolicCalculation.new(...).execute()``); both spellings must work here.
The first half (symbolic stage + transpiler + hipcc) runs without a GPU; the second half is the ``-m gpu`` test.
"""

import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))

# README.md:50-73 as a user would type it, `inflatox` -> `inflatox_amd`; IPython's display() is replaced by a no-op
# (IPython is a notebook convenience of the snippet, not part of the package).
SNIPPET_MODEL = """
import inflatox_amd as inflatox
import sympy as sp
import numpy as np
display = lambda *a, **k: None

#define model
φ, θ, L, m, φ0 = sp.symbols('φ θ L m φ0')
fields = [φ, θ]

V = (1/2*m**2*(φ-φ0)**2).nsimplify()
g = [
  [1, 0],
  [0, L**2 * sp.sinh(φ/L)**2]
]

#print metric and potential
display(g, V)

#symbolic calculation
calc = inflatox.SymbolicCalculation.new(fields, g, V)
hesse = calc.execute()

#run the compiler
out = inflatox.Compiler(hesse).compile()
"""

# README.md:75-87
SNIPPET_SWEEP = """
#evaluate the compiled potential and Hesse matrix
from inflatox_amd.consistency_conditions import GeneralisedAL
anguelova = GeneralisedAL(out)

params = np.array([1.0, 1.0, 1.0])
x = np.array([2.0, -2.0])
print(anguelova.calc_V(x, params))
print(anguelova.calc_H(x, params))

extent = [-1., 1., -1., 1.]
consistency_condition, epsilon_V, epsilon_H, eta_H, delta, omega = anguelova.complete_analysis(params, *extent)
"""


def _golden_symbols():
    with open(os.path.join(HERE, "golden", "symbols.json")) as fh:
        return json.load(fh)["hyperbolic"]


def test_readme_snippet_builds_the_headline_model(capsys):
    import inflatox_amd

    assert inflatox_amd.SymbolicCalculation is inflatox_amd.InflationModelBuilder
    ns: dict = {}
    exec(compile(SNIPPET_MODEL, "README.md", "exec"), ns)
    out = ns["out"]
    want = _golden_symbols()
    got = {str(k): v for k, v in out.symbol_dictionary.items()}
    assert got == want["symbol_dictionary"]
    assert (out.n_fields, out.n_parameters) == (want["n_fields"], want["n_parameters"])
    assert os.path.getsize(out.shared_object_path) > 0
    # execute() is build(): same expressions as the builder's own name gives
    again = inflatox_amd.InflationModelBuilder.new(ns["fields"], ns["g"], ns["V"], silent=True, init_sympy_printing=False).build()
    assert again.potential == ns["hesse"].potential
    assert again.hesse_cmp == ns["hesse"].hesse_cmp and again.gradient_square == ns["hesse"].gradient_square


@pytest.mark.gpu
def test_readme_snippet_runs_end_to_end_on_the_gpu():
    import oracle
    import workloads

    ns: dict = {}
    exec(compile(SNIPPET_MODEL + SNIPPET_SWEEP, "README.md", "exec"), ns)
    # README.md:82-84 / tests/test_doc.py-style known answers of the hyperbolic model: V = (x0 - 1)^2 / 2 at m = phi0 = 1
    assert ns["anguelova"].calc_V(ns["x"], ns["params"]) == 0.5
    six = np.stack([ns[k] for k in ("consistency_condition", "epsilon_V", "epsilon_H", "eta_H", "delta", "omega")], axis=-1)
    assert six.shape == (1000, 1000, 6)  # the reference's default grid (consistency_conditions.py:233-234)
    spec = workloads.example_models.get("hyperbolic")
    c_src, _ = oracle.emit_c_source(workloads.model_for("hyperbolic"), **spec.compiler_kwargs)
    want = oracle.OracleModel(oracle.compile_c_model(c_src)).complete_analysis(ns["params"], tuple(ns["extent"]), 1000, 1000)
    assert np.array_equal(np.isnan(six), np.isnan(want))
    fin = np.isfinite(want)
    assert np.array_equal(six[~fin & ~np.isnan(want)], want[~fin & ~np.isnan(want)])
    rel = np.abs(six[fin] - want[fin]) / np.maximum(np.abs(want[fin]), 1e-300)
    assert rel.max() <= 1e-10  # north_star's bar, no allowance


def test_package_exports_the_reference_names(capsys):
    """python/inflatox/__init__.py:20-40: everything the reference's package exports on the sweep path exists here under the
    same name (`background`, the serial ODE solver, is out of scope); log_info / log_warn write the reference's badge lines
    (src/lib.rs:53-61,94-102) to stderr."""
    import inflatox_amd

    for name in ("CompilationArtifact", "Compiler", "InflationModel", "InflationModelBuilder", "consistency_conditions", "log_info", "log_warn", "__version__"):
        assert name in inflatox_amd.__all__, name
    for name in ("CompilationArtifact", "Compiler", "InflationModel", "InflationModelBuilder", "log_info", "log_warn", "__version__"):
        assert getattr(inflatox_amd, name) is not None
    inflatox_amd.log_info("sweep done")
    inflatox_amd.log_warn("outside the domain?")
    err = capsys.readouterr().err
    assert err == "[Inflatox Info]\nsweep done\n[Inflatox Warning]\noutside the domain?\n"
