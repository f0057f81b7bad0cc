"""Workload definitions and build-and-cache helpers for the named example models -- inputs to the hot path, kept
outside the ``inflatox_amd`` package (used by build(), bench.py, tests and scripts).

``artifact_for(name)`` runs the symbolic stage and the transpiler for one of
``example_models.ALL`` and returns ``(spec, CompilationArtifact)``.  The hipcc step is served from
the content-addressed in-tree cache (``inflatox_amd/_jit_cache``) when ``build()`` has run before;
the symbolic stage is cheap (0.2-10 s) and always re-run, so the cache key -- the generated header
text -- is recomputed rather than trusted.
"""

from __future__ import annotations

import functools

from inflatox_amd.compiler import CompilationArtifact, Compiler
from inflatox_amd.symbolic import InflationModel, InflationModelBuilder

from . import example_models


@functools.lru_cache(maxsize=None)
def model_for(name: str) -> InflationModel:
    spec = example_models.get(name)
    builder = InflationModelBuilder.new(
        spec.fields, spec.metric, spec.potential, model_name=name, init_sympy_printing=False, **spec.builder_kwargs
    )
    return builder.build(spec.guesses)


def _experiment_switches() -> dict:
    """Experiment switches for runs of the whole parity suite / the probes under another build of the example models
    (profiles/r03_experiments.txt): ``INFLX_REGROUP=1`` or a comma list of value names (``V,v00,g``), ``INFLX_TAN_SHORTCUT=T``, ``INFLX_CONTRACTION=expression``.
    They live here, in test / bench infrastructure: ``Compiler`` itself reads no environment variable."""
    import os

    out = {}
    env = os.environ.get("INFLX_REGROUP", "")
    if env:
        if env == "auto":
            raise ValueError('INFLX_REGROUP=auto: the measured choice needs a sample; use artifact_for(name, tuned=True)')
        out["regroup"] = bool(int(env)) if env.isdigit() else tuple(env.split(","))
    env = os.environ.get("INFLX_CONTRACTION", "")  # "expression": Compiler(contraction="expression"), the round-6 measurement
    if env:
        out["contraction"] = env
    env = os.environ.get("INFLX_TAN_SHORTCUT", "")
    if env:
        out["tan_shortcut"] = float(env)
    return out


def artifact_for(name: str, tuned: bool = False, **compiler_overrides) -> tuple[example_models.ModelSpec, CompilationArtifact]:
    """``tuned``: the profile-guided build -- ``Compiler(regroup="auto", sample=(spec.args, spec.extent))``: products and
    sums of the model values that qualify on a sample of the workload's own parameter values and field range are
    re-associated (inflatox_amd/_instrument.py); the default is the reference's arithmetic."""
    spec = example_models.get(name)
    kwargs = dict(spec.compiler_kwargs)
    if tuned:
        kwargs.update(regroup="auto", sample=(spec.args, spec.extent))
    else:
        kwargs.update(_experiment_switches())
    kwargs.update(compiler_overrides)
    art = Compiler(model_for(name), silent=True, **kwargs).compile()
    return spec, art
