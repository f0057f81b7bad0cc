"""Version of this package and of the model-artefact ABI it implements.

The ABI version is the reference's (python/inflatox/version.py:22, src/lib.rs:50): artefacts carry
it in their ``VERSION`` symbol and the loader compares major.minor (src/inflatox_version.rs:48-53).
"""

__version__ = "0.1.0"
__abi_version__ = "5.0.0"
