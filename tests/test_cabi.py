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

    def exports(path):
        table = subprocess.run([readelf, "-s", "--dyn-syms", path], capture_output=True, text=True, check=True).stdout
        return {ln.split()[-1].removesuffix(".kd") for ln in table.splitlines() if " GLOBAL " in ln}  # (kernels also export a descriptor NAME.kd)

    data = ("VERSION", "DIM", "N_PARAMETERS", "MODEL_NAME", "USE_GSL", "INFLX_KERNEL_INFO", "INFLX_GROUPS", "MODEL_TAG")
    # what Compiler.compile() builds: the CORE object -- the reference's data symbols and every kernel complete_analysis needs
    core = exports(art.shared_object_path)
    for sym in data:
        assert sym in core, sym
    for sym in ("inflx_stage_tables", "inflx_basis_points", "inflx_sweep_rowstream6", "inflx_sweep_rowstream_planes", "inflx_sweep_colstream"):
        assert sym in core, sym
    for kind in ("tile", "rows", "rowvals", "colvals", "traj"):
        assert f"inflx_sweep_{kind}_complete" in core
    assert not any(name.startswith("inflx_sweep_tile_") and name != "inflx_sweep_tile_complete" for name in core), sorted(core)
    # every other operation: a group object of its own, built on first use, with the same data symbols (inflx_attach compares them)
    for group, kernels in (("consistency", ["inflx_sweep_tile_consistency", "inflx_sweep_traj_consistency"]), ("raw", ["inflx_sweep_tile_raw", "inflx_sweep_rows_raw"]),
                           ("stats", ["inflx_sweep_tile_complete_stats", "inflx_sweep_tile_complete_stats_nostore", "inflx_sweep_rowvals_complete_stats"]), ("values", ["inflx_ops_on_values"])):
        path = art.ensure_group(group)
        assert path == art.shared_object_path + "." + group and os.path.getsize(path) > 0
        got = exports(path)
        for sym in data + tuple(kernels):
            assert sym in got, (group, sym)
        assert "inflx_sweep_tile_complete" not in got and "inflx_stage_tables" not in got
    assert art.ensure_group("core") is None
    # a complete artefact in one file, for a C client that wants one
    from inflatox_amd.compiler import Compiler
    from workloads import example_models

    full = Compiler(workloads.model_for("hyperbolic"), silent=True, kernel_groups="all", **example_models.get("hyperbolic").compiler_kwargs).compile()
    everything = exports(full.shared_object_path)
    for op in ("complete", "consistency", "rapidturn", "epsilon_v", "raw", "qdif", "hesse"):
        for kind in ("tile", "rows", "traj"):
            assert f"inflx_sweep_{kind}_{op}" in everything
    assert "inflx_ops_on_values" in everything and "inflx_sweep_tile_complete_stats" in everything
    assert full.ensure_group("raw") is None and full.ensure_all_groups() == []
    # the sidecars go with the artefact
    paths = [art.shared_object_path] + [art.shared_object_path + "." + g for g in ("consistency", "raw", "stats", "values")]
    assert all(os.path.exists(q) for q in paths)
    del art
    import gc

    gc.collect()
    assert not any(os.path.exists(q) for q in paths)


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
