"""One call, several GPUs (include/inflx_hip.h: inflx_open_multi, inflx_sweep_host_multi, inflx_complete_analysis_multi,
inflx_sweep_stats_multi): the partition on the CPU, and on an MI355X two / three handles on the one GPU (`devices=[0, 0]`)
against the single-device call, bit for bit, on a parameter-axis split and on a row-axis split."""

import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_native_partition_equals_the_distributed_one_and_covers_the_index_space():
    """inflx_shard_plan (the in-process, several-devices form) and distributed.plan_shard (one process per GPU) are the
    same partition; blocks are contiguous, disjoint and cover (P, N0)."""
    from inflatox_amd import _native
    from inflatox_amd.distributed import plan_shard

    for P, N0, world in [(1, 8192, 8), (512, 8192, 8), (32, 4096, 8), (5, 100, 4), (3, 7, 4), (1, 3, 8), (8, 1, 8), (9, 10, 2), (1, 1, 1), (2, 5, 3)]:
        covered = np.zeros((P, N0), dtype=int)
        for rank in range(world):
            a, b = _native.shard_plan(P, N0, world, rank), plan_shard(P, N0, world, rank)
            assert (a["axis"], a["p_begin"], a["p_count"], a["row_begin"], a["row_count"]) == (b.axis, b.p_begin, b.p_count, b.row_begin, b.row_count)
            covered[a["p_begin"] : a["p_begin"] + a["p_count"], a["row_begin"] : a["row_begin"] + a["row_count"]] += 1
        assert (covered == 1).all(), (P, N0, world)
    with pytest.raises(ValueError):
        _native.shard_plan(4, 4, 2, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["hyperbolic", "doc"])
def test_two_handles_on_one_gpu_equal_the_single_device_result(name, gpu_lib):
    """Parameter-axis split (P >= devices) and row-axis split (P < devices), AoS and SoA, ragged shapes, a device count
    that does not divide the axis, every operation of the grid sweeps -- each against the single-device call."""
    import workloads

    spec, art = workloads.artifact_for(name)
    one = gpu_lib.InflatoxDevLib(art.shared_object_path)
    two = gpu_lib.InflatoxMultiLib(art.shared_object_path, [0, 0])
    three = gpu_lib.InflatoxMultiLib(art.shared_object_path, [0, 0, 0])
    assert two.n_devices == 2 and two.devices == [0, 0] and three.n_devices == 3
    rng = np.random.default_rng(7)
    rows = np.asarray(spec.args, dtype=np.float64) * rng.uniform(0.8, 1.25, size=(5, len(spec.args)))
    for multi in (two, three):
        for layout in (gpu_lib.LAYOUT_AOS, gpu_lib.LAYOUT_SOA):
            for P, n0, n1 in ((5, 37, 130), (2, 64, 96), (1, 45, 333), (1, 2, 64), (2, 7, 258)):
                want = one.sweep_host(gpu_lib.OP_COMPLETE, rows[:P], spec.extent, n0, n1, layout=layout)
                got = multi.sweep_host(gpu_lib.OP_COMPLETE, rows[:P], spec.extent, n0, n1, layout=layout)
                assert got.shape == want.shape and np.array_equal(got, want, equal_nan=True), (multi.n_devices, layout, P, n0, n1)
        for op in (gpu_lib.OP_CONSISTENCY, gpu_lib.OP_EPSILON_V, gpu_lib.OP_RAPIDTURN, gpu_lib.OP_RAW):
            want = one.sweep_host(op, rows[:1], spec.extent, 33, 70, layout=gpu_lib.LAYOUT_SOA)
            got = multi.sweep_host(op, rows[:1], spec.extent, 33, 70, layout=gpu_lib.LAYOUT_SOA)
            assert np.array_equal(got, want, equal_nan=True), op
        # max_devices = 1 is the single-device call through the multi-handle
        assert np.array_equal(multi.sweep_host(gpu_lib.OP_COMPLETE, rows[:3], spec.extent, 20, 40, max_devices=1), one.sweep_host(gpu_lib.OP_COMPLETE, rows[:3], spec.extent, 20, 40), equal_nan=True)
        # the summary: every device reduces its block inside its sweep, the host combines
        a, b = one.sweep_stats(rows[:5], spec.extent, 90, 200), multi.sweep_stats(rows[:5], spec.extent, 90, 200)
        b1 = multi.sweep_stats(rows[:1], spec.extent, 91, 200)  # row-axis split
        a1 = one.sweep_stats(rows[:1], spec.extent, 91, 200)
        for x, y in ((a, b), (a1, b1)):
            assert np.array_equal(x["min"], y["min"]) and np.array_equal(x["max"], y["max"]) and np.array_equal(x["count"], y["count"])
    # a wrong parameter count is refused with the shape error, and the handle stays usable
    with pytest.raises(gpu_lib.InflatoxShapeError):
        two.sweep_host(gpu_lib.OP_COMPLETE, np.ones((2, len(spec.args) + 1)), spec.extent, 8, 8)
    assert np.isfinite(two.sweep_host(gpu_lib.OP_RAW, rows[:1], spec.extent, 8, 8)).any()


@pytest.mark.gpu
def test_front_end_with_devices(gpu_lib):
    """GeneralisedAL(art, devices=[0, 0]): complete_analysis / the single-quantity sweeps / batch / summary equal the
    single-device object's results; threads=1 limits the call to one device."""
    import workloads
    from inflatox_amd.consistency_conditions import GeneralisedAL, InflationCondition

    spec, art = workloads.artifact_for("doc")

    def make(**kw):
        al = GeneralisedAL.__new__(GeneralisedAL)
        InflationCondition.__init__(al, art, validate_basis=False, **kw)
        return al

    single, multi = make(), make(devices=[0, 0])
    assert multi.multi is not None and multi.multi.n_devices == 2 and single.multi is None
    for n0, n1 in ((101, 77), (1000, 1000)):
        a = single.complete_analysis(spec.args, *spec.extent, n0, n1, progress=False)
        b = multi.complete_analysis(spec.args, *spec.extent, n0, n1, progress=False)
        c = multi.complete_analysis(spec.args, *spec.extent, n0, n1, progress=False, threads=1)
        assert all(np.array_equal(x, y, equal_nan=True) and np.array_equal(x, z, equal_nan=True) and y.flags.writeable for x, y, z in zip(a, b, c))
    for meth in ("consistency", "epsilon_v", "consistency_rapidturn"):
        assert np.array_equal(getattr(single, meth)(spec.args, *spec.extent, 65, 130, progress=False), getattr(multi, meth)(spec.args, *spec.extent, 65, 130, progress=False), equal_nan=True)
    rows = np.tile(spec.args, (3, 1)) * np.array([[1.0], [1.5], [0.5]])
    assert np.array_equal(single.complete_analysis_batch(rows, *spec.extent, 40, 50), multi.complete_analysis_batch(rows, *spec.extent, 40, 50), equal_nan=True)
    s1, s2 = single.complete_analysis_summary(rows, *spec.extent, 64, 64), multi.complete_analysis_summary(rows, *spec.extent, 64, 64)
    assert all(np.array_equal(s1[k], s2[k]) for k in ("min", "max", "count"))
    all_devices = make(devices="all")
    assert all_devices.multi.n_devices == gpu_lib.device_count()


_CHUNKED = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
import workloads
from inflatox_amd import _native
spec, art = workloads.artifact_for("doc")
one = _native.InflatoxDevLib(art.shared_object_path)
two = _native.InflatoxMultiLib(art.shared_object_path, [0, 0])
rows = np.tile(spec.args, (3, 1)) * np.array([[1.0], [1.5], [0.5]])
for layout in (0, 1):
    for P in (3, 1):
        want = one.sweep_host(_native.OP_COMPLETE, rows[:P], spec.extent, 300, 520, layout=layout)
        got = two.sweep_host(_native.OP_COMPLETE, rows[:P], spec.extent, 300, 520, layout=layout, progress=True)
        assert np.array_equal(got, want, equal_nan=True), (layout, P)
print("chunked ok")
"""


_SUBSET = """
import sys
sys.path.insert(0, {root!r})
import numpy as np
from inflatox_amd import _native
import workloads
spec, art = workloads.artifact_for("doc")
lib = _native.InflatoxDevLib(art.shared_object_path)
rows = np.stack([spec.args, spec.args * 1.25, spec.args * 0.8])
ext = (0.5, 2.5, 0.0, 3.0)
full = lib.sweep_host(_native.OP_RAW, rows, ext, 300, 520, row_begin=7, row_count=250, layout=_native.LAYOUT_SOA)
for first, count in ((0, 1), (1, 3), (4, 1), (0, 5), (2, 2)):
    part = lib.sweep_host_planes(_native.OP_RAW, rows, ext, 300, 520, first, count, row_begin=7, row_count=250)
    assert part.shape == (3, count, 250, 520) and np.array_equal(part, full[:, first:first + count], equal_nan=True), (first, count)
hess = lib.sweep_host(_native.OP_HESSE, rows[0], ext, 300, 520, row_begin=7, row_count=250, layout=_native.LAYOUT_SOA)  # v00, v01, v10, v11
assert hess.shape == (4, 250, 520) and np.array_equal(hess[[0, 2, 3]], full[0, 1:4], equal_nan=True)
print("subset ok")
"""


@pytest.mark.gpu
def test_plane_subsets_through_the_chunk_pipeline(gpu_lib):
    """A plane subset (calc_V_array: one plane of five) prefers the whole-result path and falls back to the chunk pipeline when
    that much HBM is not free -- forced here by switching the whole-result path off (INFLX_WHOLE_RESULT_MB=0) with 1 MiB chunks:
    every subset of a parameter batch and a row range equals the planes of the full planes-layout sweep bit for bit."""
    env = dict(os.environ, INFLX_WHOLE_RESULT_MB="0", INFLX_CHUNK_MB="1")
    proc = subprocess.run([sys.executable, "-c", _SUBSET.format(root=ROOT)], env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0 and "subset ok" in proc.stdout, proc.stdout + proc.stderr


@pytest.mark.gpu
def test_chunk_pipeline_and_progress_lines_of_a_multi_device_sweep(gpu_lib):
    """The same comparison through the chunk pipeline (results larger than the whole-result limit: forced with
    INFLX_WHOLE_RESULT_MB=0 and 1 MiB chunks), with progress reporting switched on for any size and interval: the
    reference's three figures (time to completion, operations per second, percentage; src/anguelova.rs:42-50) appear on
    stderr, the results are unchanged."""
    env = dict(os.environ, INFLX_WHOLE_RESULT_MB="0", INFLX_CHUNK_MB="1", INFLX_PROGRESS_MIN_MB="0", INFLX_PROGRESS_INTERVAL_MS="1")
    proc = subprocess.run([sys.executable, "-c", _CHUNKED.format(root=ROOT)], env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0 and "chunked ok" in proc.stdout, proc.stdout + proc.stderr
    lines = [ln for ln in proc.stderr.splitlines() if "Time to completion" in ln]
    assert lines and all("grid points/s" in ln and "%" in ln for ln in lines), proc.stderr
    assert "Calculating on 2 HIP device(s)" in proc.stderr and "Calculation finished" in proc.stderr
    # ... and the whole-result path with progress marks (slices + events), default limits
    env2 = dict(os.environ, INFLX_PROGRESS_MIN_MB="0", INFLX_PROGRESS_INTERVAL_MS="1")
    proc2 = subprocess.run([sys.executable, "-c", _CHUNKED.format(root=ROOT)], env=env2, capture_output=True, text=True, timeout=600)
    assert proc2.returncode == 0 and "chunked ok" in proc2.stdout, proc2.stdout + proc2.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["hyperbolic", "doc"])
def test_device_resident_multi_sweep_and_peer_all_gather(name, gpu_lib):
    """inflx_sweep_device_multi (every device keeps its block) and inflx_sweep_allgather_multi (every device ends up with the
    whole result: in-place sweep into the device's slice, then one hipMemcpyPeerAsync per peer) with two and three handles
    on the one GPU, parameter-axis and row-axis split, equal and unequal blocks -- against the single-device host result."""
    import torch

    import workloads

    spec, art = workloads.artifact_for(name)
    one = gpu_lib.InflatoxDevLib(art.shared_object_path)
    rng = np.random.default_rng(11)
    rows = np.asarray(spec.args, dtype=np.float64) * rng.uniform(0.8, 1.25, size=(5, len(spec.args)))
    for devices in ([0, 0], [0, 0, 0]):
        multi = gpu_lib.InflatoxMultiLib(art.shared_object_path, devices)
        n = multi.n_devices
        for P, n0, n1 in ((4, 36, 130), (5, 37, 70), (1, 64, 96), (1, 45, 333), (2, 7, 258)):
            want = one.sweep_host(gpu_lib.OP_COMPLETE, rows[:P], spec.extent, n0, n1).reshape(P, n0, n1, 6)
            # every device the whole result
            full = [torch.full((P, n0, n1, 6), -5.0, dtype=torch.float64, device="cuda:0") for _ in range(n)]
            torch.cuda.synchronize()
            multi.sweep_allgather(gpu_lib.OP_COMPLETE, rows[:P], [t.data_ptr() for t in full], full[0].numel() * 8, spec.extent, n0, n1)
            for k, t in enumerate(full):
                assert np.array_equal(t.cpu().numpy(), want, equal_nan=True), (devices, P, n0, n1, k)
            # every device its own block
            plans = [gpu_lib.shard_plan(P, n0, n, k) for k in range(n)]
            blocks = [torch.full((max(pl["p_count"], 1), max(pl["row_count"], 1), n1, 6), -5.0, dtype=torch.float64, device="cuda:0") for pl in plans]
            torch.cuda.synchronize()
            multi.sweep_device(gpu_lib.OP_COMPLETE, rows[:P], [b.data_ptr() for b in blocks], [b.numel() * 8 for b in blocks], spec.extent, n0, n1)
            torch.cuda.synchronize()  # (a device-wide wait: the handles' own streams included)
            for pl, b in zip(plans, blocks):
                if pl["p_count"] and pl["row_count"]:
                    ref = want[pl["p_begin"] : pl["p_begin"] + pl["p_count"], pl["row_begin"] : pl["row_begin"] + pl["row_count"]]
                    assert np.array_equal(b.cpu().numpy(), ref, equal_nan=True), (devices, P, n0, n1, pl)
        with pytest.raises(gpu_lib.InflatoxShapeError):
            multi.sweep_allgather(gpu_lib.OP_COMPLETE, rows[:2], [t.data_ptr() for t in full], 8, spec.extent, 8, 8)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["hyperbolic", "doc"])
def test_all_gather_through_rccl(name, gpu_lib):
    """inflx_sweep_allgather_multi_ex(..., INFLX_GATHER_RCCL): the exchange step as ONE in-place ncclAllGather per contiguous image
    (SURVEY section 8e).  A one-GPU box has one rank to offer (RCCL refuses two ranks on a device): the communicator is created
    by ncclCommInitAll, the collective is enqueued on the sweep stream behind the sweep and the result equals the host sweep --
    parameter-axis and row-axis plans, every operation width; the argument checks (unequal blocks, one device twice) answer
    with the documented errors instead of reaching RCCL."""
    import torch

    import workloads

    spec, art = workloads.artifact_for(name)
    one = gpu_lib.InflatoxDevLib(art.shared_object_path)
    rng = np.random.default_rng(12)
    rows = np.asarray(spec.args, dtype=np.float64) * rng.uniform(0.8, 1.25, size=(4, len(spec.args)))
    multi = gpu_lib.InflatoxMultiLib(art.shared_object_path, [0])
    for op, width in ((gpu_lib.OP_COMPLETE, 6), (gpu_lib.OP_RAW, 5), (gpu_lib.OP_EPSILON_V, 1)):
        for P, n0, n1 in ((3, 36, 130), (1, 64, 96)):
            want = one.sweep_host(op, rows[:P], spec.extent, n0, n1).reshape(P, n0, n1, width)
            full = torch.full((P, n0, n1, width), -5.0, dtype=torch.float64, device="cuda:0")
            torch.cuda.synchronize()
            multi.sweep_allgather(op, rows[:P], [full.data_ptr()], full.numel() * 8, spec.extent, n0, n1, gather="rccl")
            assert np.array_equal(full.cpu().numpy(), want, equal_nan=True), (op, P, n0, n1)
    # two handles on the one device: the peer-push gather takes them, the RCCL one says why it cannot
    two = gpu_lib.InflatoxMultiLib(art.shared_object_path, [0, 0])
    bufs = [torch.empty((2, 8, 8, 6), dtype=torch.float64, device="cuda:0") for _ in range(2)]
    with pytest.raises(ValueError, match="one device per handle"):
        two.sweep_allgather(gpu_lib.OP_COMPLETE, rows[:2], [b.data_ptr() for b in bufs], bufs[0].numel() * 8, spec.extent, 8, 8, gather="rccl")
    with pytest.raises(gpu_lib.InflatoxShapeError, match="equal blocks"):
        two.sweep_allgather(gpu_lib.OP_COMPLETE, rows[:3], [b.data_ptr() for b in bufs], 3 * 8 * 8 * 48, spec.extent, 8, 8, gather="rccl")
    two.sweep_allgather(gpu_lib.OP_COMPLETE, rows[:2], [b.data_ptr() for b in bufs], bufs[0].numel() * 8, spec.extent, 8, 8)  # peer pushes
    assert np.array_equal(bufs[0].cpu().numpy(), bufs[1].cpu().numpy(), equal_nan=True)


def test_parameter_grid_is_the_outer_product_of_its_axes():
    """north_star's "2D field-space x N-D parameter grid": parameter_grid builds the (len_1, ..., len_k, n_p) rows of a scan from a
    base vector and named axes (position, printed name or symbol); no device is needed for that."""
    import sympy

    import workloads
    from inflatox_amd.consistency_conditions import GeneralisedAL

    spec, art = workloads.artifact_for("hyperbolic")
    al = GeneralisedAL.__new__(GeneralisedAL)
    al.artifact = art
    names = [n for n, t in sorted(art.symbol_dictionary.items(), key=lambda kv: kv[1]) if t.startswith("args[")]
    assert names == ["m", "φ0", "L"]  # README order (tests/golden/symbols.json)
    ms, Ls = np.array([0.5, 1.0]), np.linspace(0.2, 2.0, 5)
    grid = al.parameter_grid(spec.args, {"m": ms, sympy.Symbol("L"): Ls})
    assert grid.shape == (2, 5, 3)
    for i in range(2):
        for j in range(5):
            assert np.array_equal(grid[i, j], [ms[i], spec.args[1], Ls[j]])
    assert np.array_equal(al.parameter_grid(spec.args, {2: Ls})[:, 2], Ls) and al.parameter_grid(spec.args, {}).shape == (3,)
    with pytest.raises(KeyError):
        al.parameter_grid(spec.args, {"θ": Ls})  # a field, not a parameter
    with pytest.raises(KeyError):
        al.parameter_grid(spec.args, {"m": ms, 0: ms})  # the same parameter twice
    with pytest.raises(Exception, match="expected 3 parameters"):
        al.parameter_grid(spec.args[:2], {"m": ms})


@pytest.mark.gpu
@pytest.mark.parametrize("name,axes", [("hyperbolic", {"m": [0.5, 1.0], "L": [0.2, 1.1, 2.0]}), ("doc", {0: [0.5, 1.0, 2.0, 3.0]})])
def test_nd_parameter_grid_sweep_equals_one_call_per_grid_node(name, axes, gpu_lib):
    """complete_analysis_batch over an N-D parameter grid = complete_analysis at every node of it, bit for bit, on one device and
    on two handles; the fused summary covers all nodes."""
    import workloads
    from inflatox_amd.consistency_conditions import GeneralisedAL, InflationCondition

    spec, art = workloads.artifact_for(name)

    def make(**kw):
        al = GeneralisedAL.__new__(GeneralisedAL)
        InflationCondition.__init__(al, art, validate_basis=False, **kw)
        return al

    one, two = make(), make(devices=[0, 0])
    grid = one.parameter_grid(spec.args, axes)
    lead = grid.shape[:-1]
    n0, n1 = 37, 129
    res = one.complete_analysis_batch(grid, *spec.extent, n0, n1)
    assert res.shape == lead + (n0, n1, 6) and res.flags.c_contiguous
    planes = one.complete_analysis_batch(grid, *spec.extent, n0, n1, layout="soa")
    assert planes.shape == lead + (6, n0, n1)
    assert np.array_equal(two.complete_analysis_batch(grid, *spec.extent, n0, n1), res, equal_nan=True)
    for node in np.ndindex(*lead):
        ref = one.complete_analysis(grid[node], *spec.extent, n0, n1, progress=False)
        for k in range(6):
            assert np.array_equal(res[node][:, :, k], ref[k], equal_nan=True) and np.array_equal(planes[node][k], ref[k], equal_nan=True)
    summary = one.complete_analysis_summary(grid, *spec.extent, n0, n1)
    with np.errstate(all="ignore"):
        for k in range(6):
            vals = res[..., k]
            count = int(np.count_nonzero(~np.isnan(vals)))
            assert summary["count"][k] == count
            if count:
                assert summary["min"][k] == np.nanmin(vals) and summary["max"][k] == np.nanmax(vals)
