"""CPU oracle for the grid-sweep hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import this package; the product (``inflatox_amd``) never does.

Pieces:
  * ``sweep_oracle.c``  C restatement of the reference's native sweep (see its header).
  * ``model_c.py``      restatement of the reference transpiler's C back-end: emits the
                        per-model C file with the reference's ABI and builds it with the
                        reference's compiler flags -- by gcc AND by clang (the reference's own
                        compiler is ``zig cc`` = clang; the two round differently, see model_c.py).
  * ``cpu_oracle.py``   ctypes bindings for the two above (+ the basis-validation restatement).
  * ``values_model.c``  a model artefact that returns table entries: the oracle's per-point operations on
                        arbitrary (V, v00, v10, v11, |dV|^2) tuples (``ops_on_values``).
  * ``special.py``      mpmath / scipy.special stand-in for the reference's GSL special functions
                        (GSL is absent from this image: that row is **parity unpinned**).

Parity status: pinned by the reference's known-answer test (tests/test_doc.py:50-51) and by
golden vectors produced from the reference's own Python stages (tests/golden/); the Rust
half of the reference cannot be built in the authoring container (no rustc/cargo), see
DESIGN.md "Oracle".
"""

from .cpu_oracle import OP, OracleModel, build_sweep_library, grid_points, ops_on_values, raw_long_double  # noqa: F401
from .model_c import emit_c_source, compile_c_model, reference_compilers  # noqa: F401
