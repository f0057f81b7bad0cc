"""The C-ABI library loads and exports every symbol include/inflx_hip.h declares (no compute)."""

import ctypes
import os
import re

from conftest import ROOT


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "inflx_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(inflx_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = _declared_functions()
    for must in ("inflx_open", "inflx_close", "inflx_complete_analysis", "inflx_sweep_device", "inflx_sweep_host", "inflx_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from inflatox_amd import _native

    _native.build_library()
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in _declared_functions():
        assert hasattr(lib, name), f"libinflx_hip.so lacks {name}"


def test_binding_covers_the_header():
    from inflatox_amd import _native

    assert sorted(_native.SIGNATURES) == _declared_functions()
    _native.load_library()


def test_open_fails_loudly_without_artefact_or_device():
    """No CPU fallback: a missing artefact is an IOError; with no GPU, opening a real one is a SystemError."""
    import pytest

    from inflatox_amd import _native

    with pytest.raises(IOError):
        _native.InflatoxDevLib("/nonexistent/model.hsaco")
    if _native.device_count() == 0:
        import workloads

        _, art = workloads.artifact_for("hyperbolic")
        with pytest.raises(SystemError):
            _native.InflatoxDevLib(art.shared_object_path)


def test_code_object_exports_the_model_abi():
    """The per-model artefact carries the reference's data symbols (src/dylib.rs:32-48) and the kernels."""
    import shutil
    import subprocess

    import workloads

    readelf = shutil.which("llvm-readelf") or "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(readelf):
        import pytest

        pytest.skip("llvm-readelf not available")
    _, art = workloads.artifact_for("hyperbolic")
    table = subprocess.run([readelf, "-s", "--dyn-syms", art.shared_object_path], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in table.splitlines() if " GLOBAL " in ln}
    for sym in ("VERSION", "DIM", "N_PARAMETERS", "MODEL_NAME", "USE_GSL", "INFLX_KERNEL_INFO"):
        assert sym in exported, sym
    for op in ("complete", "consistency", "rapidturn", "epsilon_v", "raw"):
        for kind in ("tile", "rows", "traj"):
            assert f"inflx_sweep_{kind}_{op}" in exported


def test_host_helpers_under_address_and_ub_sanitizers(tmp_path):
    """tests/host_units.cpp: the hand-written memory code of the host library (streaming-store fill, transfer spans, page helpers,
    the multi-device partition, the progress reporter) compiled TOGETHER WITH the library source under ASan + UBSan and run on the
    CPU -- sanitizers are a CPU-only tool on this pool, and these helpers never call into HIP."""
    import shutil
    import subprocess

    import pytest

    gxx = shutil.which("g++")
    rocm = "/opt/rocm"
    if gxx is None or not os.path.exists(os.path.join(rocm, "include", "hip", "hip_runtime.h")):
        pytest.skip("g++ or the HIP headers are not available")
    exe = str(tmp_path / "host_units")
    cmd = [gxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__",
           f"-I{rocm}/include", f"-I{os.path.join(ROOT, 'include')}", f"-I{os.path.join(ROOT, 'inflatox_amd', 'csrc')}",
           os.path.join(ROOT, "tests", "host_units.cpp"), "-o", exe, f"-L{rocm}/lib", "-lamdhip64", "-lpthread", f"-Wl,-rpath,{rocm}/lib"]  # fmt: skip
    built = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert built.returncode == 0, built.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert run.returncode == 0 and "all checks passed" in run.stdout, (run.stdout + run.stderr)[-3000:]
