"""inflatox_amd -- MI355X-native grid-sweep path of inflatox.

Public surface mirrors the reference package (python/inflatox/__init__.py:20-40):
``InflationModelBuilder`` -> ``Compiler`` -> ``consistency_conditions.GeneralisedAL``.
"""

from .compiler import CompilationArtifact, Compiler
from .symbolic import InflationModel, InflationModelBuilder, SymbolicCalculation
from .version import __version__

__all__ = [
    "CompilationArtifact",
    "Compiler",
    "InflationModel",
    "InflationModelBuilder",
    "SymbolicCalculation",
    "consistency_conditions",
    "log_info",
    "log_warn",
    "__version__",
]


def log_info(msg: str) -> None:
    """``libinflx_rs.log_info`` (src/lib.rs:53-56,94-97): the message under the reference's badge line, on stderr."""
    import sys

    print(f"[Inflatox Info]\n{msg}", file=sys.stderr, flush=True)


def log_warn(msg: str) -> None:
    """``libinflx_rs.log_warn`` (src/lib.rs:58-61,99-102)."""
    import sys

    print(f"[Inflatox Warning]\n{msg}", file=sys.stderr, flush=True)


def __getattr__(name):
    # consistency_conditions binds the native library on import; keep `import inflatox_amd` light
    if name == "consistency_conditions":
        import importlib

        return importlib.import_module(".consistency_conditions", __name__)
    raise AttributeError(name)
