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
    "__version__",
]


def __getattr__(name):
    # consistency_conditions binds the native library on import; keep `import inflatox_amd` light
    if name == "consistency_conditions":
        import importlib

        return importlib.import_module(".consistency_conditions", __name__)
    raise AttributeError(name)
