"""The double-buffered row-table path of row-only models (csrc/inflx_hip.cpp launch_grid; the path the headline
benchmark takes): every grid point must be computed from ITS OWN parameter row (src/anguelova.rs:524-540 evaluates
each point with the `p` of its call), also when

  * one call is cut into several table batches that alternate between the two tables, and
  * several calls with different parameters are enqueued back to back without any synchronisation.

A stale or prematurely overwritten table would hand a whole grid row the six values of another parameter
row -- invisible to tests that repeat one parameter vector.  Checks: every column of a row-only model equals
column 0 (device-side, the whole array), and column 0 of every parameter row equals the CPU oracle's N0 x 1 sweep
with that row's parameters (index -> coordinate map included).
"""

import numpy as np
import pytest
from conftest import compare, oracle_model

from oracle import OP

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hyp(gpu_lib):
    import workloads

    spec, art = workloads.artifact_for("hyperbolic")
    lib = gpu_lib.InflatoxDevLib(art.shared_object_path)
    assert lib.stage_info["out_mask"] & 2 == 0
    return spec, lib


def distinct_rows(P, salt):
    """P parameter rows [m, phi0, L], no two alike and none equal to a row of another `salt`."""
    k = np.arange(P, dtype=np.float64)
    return np.column_stack([0.7 + 0.05 * k + 0.003 * salt, 1.0 + 0.011 * k - 0.002 * salt, 0.4 + 0.17 * k + 0.013 * salt])


def check_block(out, rows, extent, n0, what, op=OP.COMPLETE):
    """out: torch tensor (P, n0, n1, K) or (P, n0, n1) on the device."""
    import torch

    om, _ = oracle_model("hyperbolic")
    if out.dim() == 3:
        out = out.unsqueeze(-1)
    for k in range(out.shape[0]):
        blk = out[k]
        col0 = blk[:, :1, :]
        same = (blk == col0) | (torch.isnan(blk) & torch.isnan(col0))
        assert bool(same.all()), f"{what}: parameter row {k}: a column differs from column 0"
        want = om.grid_sweep(op, rows[k], extent, n0, 1)
        want = want[:, 0, :] if want.ndim == 3 else want[:, :1]
        compare(col0[:, 0, :].cpu().numpy(), want, 1e-10, f"{what}: parameter row {k}")
        # and it is NOT the neighbouring row's result (the defect this test exists for)
        if k + 1 < out.shape[0]:
            other = om.grid_sweep(op, rows[k + 1], extent, n0, 1)
            other = other[:, 0, :] if other.ndim == 3 else other[:, :1]
            assert not np.array_equal(other, want, equal_nan=True)


def test_three_table_batches_in_one_call(hyp, gpu_lib):
    """8192 rows x 2731 columns: 32 replicas -> 16 MiB of table per parameter row -> 4 rows per 64 MiB batch;
    P = 9 -> batches of 4 / 4 / 1 on alternating tables (9.7 GB device-resident)."""
    import torch

    spec, lib = hyp
    torch.cuda.empty_cache()
    n0, n1, P = 8192, 2731, 9
    plan = lib.sweep_plan(gpu_lib.OP_COMPLETE, P, n1, n0)
    assert {k: plan[k] for k in ("path", "batch_rows", "batches", "replicas")} == {"path": "row_stream", "batch_rows": 4, "batches": 3, "replicas": 32}, plan
    rows = distinct_rows(P, 0)
    out = torch.full((P, n0, n1, 6), -7.0, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()  # the fill ran on torch's default stream, the sweep runs on the model's own
    lib.sweep_device(gpu_lib.OP_COMPLETE, rows, out.data_ptr(), out.numel() * 8, spec.extent, n0, n1, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    check_block(out, rows, spec.extent, n0, "3 batches")
    del out


def test_many_small_batches_and_a_row_range(hyp, gpu_lib):
    """Batches of ONE parameter row (65536 grid rows x 32 replicas = 128 MiB > the 64 MiB budget): seven batches,
    each with two store-stream launches (grid.y <= 65535), on a row range that does not start at row 0."""
    import torch

    spec, lib = hyp
    torch.cuda.empty_cache()
    n0, n1, P = 70000, 2731, 7
    rb, rc = 3000, 65536
    plan = lib.sweep_plan(gpu_lib.OP_COMPLETE, P, n1, rc)
    assert plan["path"] == "row_stream" and plan["batch_rows"] == 1 and plan["batches"] == 7, plan
    rows = distinct_rows(P, 1)
    out = torch.full((P, rc, n1, 6), -7.0, dtype=torch.float64, device="cuda:0")  # 60 GB
    torch.cuda.synchronize()
    lib.sweep_device(gpu_lib.OP_COMPLETE, rows, out.data_ptr(), out.numel() * 8, spec.extent, n0, n1, row_begin=rb, row_count=rc, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    om, _ = oracle_model("hyperbolic")
    for k in range(P):
        col0 = out[k, :, :1, :]
        same = (out[k] == col0) | (torch.isnan(out[k]) & torch.isnan(col0))
        assert bool(same.all()), k
        want = om.grid_sweep(OP.COMPLETE, rows[k], spec.extent, n0, 1)[rb : rb + rc, 0, :]
        compare(col0[:, 0, :].cpu().numpy(), want, 1e-10, f"row range, parameter row {k}")
    del out


def test_plane_stream_batches(hyp, gpu_lib):
    """The single-value / SoA store stream (inflx_sweep_rowstream_planes) through several table batches:
    epsilon_V for P = 9 over 8192 x 16384 (32 replicas -> 4 rows per batch), and the six SoA planes for P = 5."""
    import torch

    spec, lib = hyp
    torch.cuda.empty_cache()
    n0, n1, P = 8192, 16384, 9
    plan = lib.sweep_plan(gpu_lib.OP_EPSILON_V, P, n1, n0)
    assert plan["path"] == "row_stream" and plan["batches"] == 3, plan
    rows = distinct_rows(P, 2)
    out = torch.full((P, n0, n1), -7.0, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()  # the fill ran on torch's default stream, the sweep runs on the model's own
    lib.sweep_device(gpu_lib.OP_EPSILON_V, rows, out.data_ptr(), out.numel() * 8, spec.extent, n0, n1, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    check_block(out, rows, spec.extent, n0, "epsilon_V planes", op=OP.EPSILON_V)
    del out
    P = 5
    assert lib.sweep_plan(gpu_lib.OP_COMPLETE, P, n1, n0, layout=gpu_lib.LAYOUT_SOA)["batches"] == 2
    rows = distinct_rows(P, 3)
    soa = torch.full((P, 6, n0, n1), -7.0, dtype=torch.float64, device="cuda:0")  # 32 GB
    torch.cuda.synchronize()
    lib.sweep_device(gpu_lib.OP_COMPLETE, rows, soa.data_ptr(), soa.numel() * 8, spec.extent, n0, n1, layout=gpu_lib.LAYOUT_SOA, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    om, _ = oracle_model("hyperbolic")
    for k in range(P):
        want = om.grid_sweep(OP.COMPLETE, rows[k], spec.extent, n0, 1)[:, 0, :]
        for q in range(6):
            plane = soa[k, q]
            col0 = plane[:, :1]
            assert bool(((plane == col0) | (torch.isnan(plane) & torch.isnan(col0))).all()), (k, q)
            compare(col0[:, 0].cpu().numpy(), want[:, q], 1e-10, f"SoA plane {q}, parameter row {k}")
    del soa


def test_back_to_back_calls_with_different_parameters(hyp, gpu_lib):
    """Eight inflx_sweep_device calls enqueued on one stream with no synchronisation in between, every one with
    different parameters, mixed shapes: single rows, a multi-batch call, a plane stream, a row range.  The
    parameter array handed to each call is overwritten right after the call returns (the library must have
    taken its copy).  Everything is checked after ONE synchronisation at the end."""
    import torch

    spec, lib = hyp
    torch.cuda.empty_cache()
    ts = torch.cuda.Stream(device="cuda:0")  # the -7 fills and the sweeps are enqueued on this one stream, in order
    stream = ts.cuda_stream
    n0, n1 = 8192, 2731
    jobs = []  # (tensor, rows, kind, extra)
    scratch = np.empty((6, 3))

    def enqueue(P, salt, op=gpu_lib.OP_COMPLETE, oop=OP.COMPLETE, rb=0, rc=None):
        rc = n0 if rc is None else rc
        rows = distinct_rows(P, salt)
        shape = (P, rc, n1, 6) if op == gpu_lib.OP_COMPLETE else (P, rc, n1)
        with torch.cuda.stream(ts):
            out = torch.full(shape, -7.0, dtype=torch.float64, device="cuda:0")
        buf = scratch[:P]
        buf[:] = rows
        lib.sweep_device(op, buf, out.data_ptr(), out.numel() * 8, spec.extent, n0, n1, row_begin=rb, row_count=rc, stream=stream)
        buf[:] = np.nan  # the caller's array is dead after the call
        jobs.append((out, rows, oop, rb, rc))

    torch.cuda.synchronize()
    enqueue(1, 10)
    enqueue(1, 11)
    enqueue(6, 12)  # two table batches (4 + 2)
    enqueue(1, 13)
    enqueue(2, 14, op=gpu_lib.OP_EPSILON_V, oop=OP.EPSILON_V)  # odd N1: the fallback row kernel, which reads its parameters on another stream
    enqueue(1, 15, rb=1000, rc=4096)
    enqueue(5, 16)
    enqueue(1, 17)
    torch.cuda.synchronize()
    om, _ = oracle_model("hyperbolic")
    for idx, (out, rows, oop, rb, rc) in enumerate(jobs):
        o4 = out if out.dim() == 4 else out.unsqueeze(-1)
        for k in range(o4.shape[0]):
            col0 = o4[k, :, :1, :]
            same = (o4[k] == col0) | (torch.isnan(o4[k]) & torch.isnan(col0))
            assert bool(same.all()), (idx, k)
            want = om.grid_sweep(oop, rows[k], spec.extent, n0, 1)
            want = want[rb : rb + rc, 0, :] if want.ndim == 3 else want[rb : rb + rc, :1]
            compare(col0[:, 0, :].cpu().numpy(), want, 1e-10, f"call {idx}, parameter row {k}")


def test_back_to_back_tile_path_calls(gpu_lib):
    """The same property on the tile path (doc model): six un-synchronised calls with different parameters, each
    bit-equal to the same sweep issued alone."""
    import torch

    import workloads

    spec, art = workloads.artifact_for("doc")
    lib = gpu_lib.InflatoxDevLib(art.shared_object_path)
    ts = torch.cuda.Stream(device="cuda:0")
    stream = ts.cuda_stream
    n0, n1 = 1500, 1300
    params = [spec.args * (1.0 + 0.07 * k) for k in range(6)]
    with torch.cuda.stream(ts):
        outs = [torch.full((n0, n1, 6), -7.0, dtype=torch.float64, device="cuda:0") for _ in params]
    buf = np.empty_like(spec.args)
    torch.cuda.synchronize()
    for p, out in zip(params, outs):
        buf[:] = p
        lib.sweep_device(gpu_lib.OP_COMPLETE, buf, out.data_ptr(), out.numel() * 8, spec.extent, n0, n1, stream=stream)
        buf[:] = np.nan
    torch.cuda.synchronize()
    for p, out in zip(params, outs):
        alone = lib.sweep_host(gpu_lib.OP_COMPLETE, p, spec.extent, n0, n1)
        assert np.array_equal(out.cpu().numpy(), alone, equal_nan=True)
    assert not np.array_equal(outs[0].cpu().numpy(), outs[1].cpu().numpy(), equal_nan=True)


def test_dominant_only_timing_refuses_several_batches(hyp, gpu_lib):
    """The timing-only mode re-runs store streams from the previous sweep's table; with several batches only the
    last table still exists, so the call is refused instead of writing other rows' values."""
    import torch

    spec, lib = hyp
    torch.cuda.empty_cache()
    n0, n1, P = 8192, 2731, 5
    out = torch.empty((P, n0, n1, 6), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    rows = distinct_rows(P, 20)
    with pytest.raises(ValueError):
        lib.sweep_device_timed(gpu_lib.OP_COMPLETE, rows, out.data_ptr(), out.numel() * 8, spec.extent, n0, n1, stream=torch.cuda.current_stream().cuda_stream, repeats=2, dominant_only=True)
    torch.cuda.synchronize()
    ms = lib.sweep_device_timed(gpu_lib.OP_COMPLETE, rows[:4], out.data_ptr(), out.numel() * 8, spec.extent, n0, n1, stream=torch.cuda.current_stream().cuda_stream, repeats=2, dominant_only=True)
    assert ms > 0
    check_block(out[:4], rows[:4], spec.extent, n0, "after dominant-only timing")
    # the in-pipeline mode times the store streams inside full sweeps -- several table batches are fine there (5 rows = 4 + 1):
    # it lies between nothing and the whole sweep, and the results are those of an ordinary sweep
    whole = lib.sweep_device_timed(gpu_lib.OP_COMPLETE, rows, out.data_ptr(), out.numel() * 8, spec.extent, n0, n1, stream=torch.cuda.current_stream().cuda_stream, repeats=2)
    inside = lib.sweep_device_timed(gpu_lib.OP_COMPLETE, rows, out.data_ptr(), out.numel() * 8, spec.extent, n0, n1, stream=torch.cuda.current_stream().cuda_stream, repeats=2, in_pipeline=True)
    assert 0.5 * whole < inside <= 1.02 * whole, (inside, whole)
    check_block(out, rows, spec.extent, n0, "after in-pipeline timing")


@pytest.mark.parametrize("name", ["hyperbolic", "doc"])
def test_more_than_65535_parameter_rows_in_one_call(name, gpu_lib):
    """A long parameter axis on a small grid (a parameter scan): 70 000 rows in one call.  grid.z / grid.y carry the
    parameter row, so the library cuts the axis into launches of at most 65 535 rows (and the store streams into batches
    of at most 65 535 images); sampled parameter rows equal the same sweep issued alone, for the AoS result, a
    single-value result and the SoA planes."""
    import workloads

    spec, art = workloads.artifact_for(name)
    lib = gpu_lib.InflatoxDevLib(art.shared_object_path)
    P, n0, n1 = 70000, 4, 6
    rng = np.random.default_rng(65535)
    rows = np.asarray(spec.args, dtype=np.float64) * rng.uniform(0.8, 1.25, size=(P, len(spec.args)))
    ss = np.array(spec.extent).reshape(2, 2)
    sample = [0, 1, 65534, 65535, 65536, P - 1] + [int(v) for v in rng.integers(0, P, 6)]
    for op, layout, shape in ((gpu_lib.OP_COMPLETE, gpu_lib.LAYOUT_AOS, (P, n0, n1, 6)), (gpu_lib.OP_EPSILON_V, gpu_lib.LAYOUT_AOS, (P, n0, n1)), (gpu_lib.OP_COMPLETE, gpu_lib.LAYOUT_SOA, (P, 6, n0, n1))):
        got = lib.sweep_host(op, rows, ss, n0, n1, layout=layout)
        assert got.shape == shape
        for k in sample:
            alone = lib.sweep_host(op, rows[k], ss, n0, n1, layout=layout)
            assert np.array_equal(got[k], alone, equal_nan=True), (name, op, layout, k)
    # the fused summary over the whole axis equals numpy over the arrays
    from inflatox_amd.distributed import numpy_summary

    stats, ref = lib.sweep_stats(rows, ss, n0, n1), numpy_summary(lib.sweep_host(gpu_lib.OP_COMPLETE, rows, ss, n0, n1))
    assert np.array_equal(stats["count"], ref["count"]) and np.array_equal(stats["min"], ref["min"]) and np.array_equal(stats["max"], ref["max"])
