"""The N>1 path on CPU: two gloo ranks shard a sweep (parameter axis, then row axis), gather it, and
the result must equal the single-process sweep.  The local compute step is the CPU oracle here; on
GPUs it is the HIP sweep (inflatox_amd.distributed.HipCompute) -- the partition and gather logic under
test is the same code."""

import os
import socket

import numpy as np
import pytest

from inflatox_amd.distributed import ShardedSweep, all_reduce_summary, block_bounds, numpy_summary, plan_shard


def test_block_bounds_cover_everything_once():
    for n in (0, 1, 5, 8, 17, 512):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                b, c = block_bounds(n, world, r)
                seen.extend(range(b, b + c))
            assert seen == list(range(n))
            sizes = [block_bounds(n, world, r)[1] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_plan_prefers_parameter_axis():
    p = plan_shard(P=512, N0=8192, world=8, rank=3)
    assert (p.axis, p.p_begin, p.p_count, p.row_begin, p.row_count) == ("param", 192, 64, 0, 8192)
    p = plan_shard(P=1, N0=8192, world=8, rank=7)
    assert (p.axis, p.p_begin, p.p_count, p.row_begin, p.row_count) == ("rows", 0, 1, 7168, 1024)
    with pytest.raises(ValueError):
        plan_shard(4, 4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, tmpdir):
    import torch.distributed as dist

    import oracle
    import workloads
    from workloads import example_models

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        name, P, N0, N1 = case
        spec = example_models.get(name)
        src, _ = oracle.emit_c_source(workloads.model_for(name), **spec.compiler_kwargs)
        om = oracle.OracleModel(oracle.compile_c_model(src))
        args = np.stack([spec.args * (1.0 + 0.1 * k) for k in range(P)])

        def compute(p_rows, row_begin, row_count):
            # rows [row_begin, row_begin+row_count) of the N0 x N1 grid: the same points as a full sweep
            x0a, x0b, x1a, x1b = spec.extent
            dx0 = (x0b - x0a) / N0
            block = np.zeros((len(p_rows), row_count, N1, 6))
            pts = oracle.grid_points(spec.extent, N0, N1).reshape(N0, N1, 2)[row_begin : row_begin + row_count].reshape(-1, 2)
            for k, p in enumerate(p_rows):
                block[k] = om.trajectory_sweep(oracle.OP.COMPLETE, p, pts).reshape(row_count, N1, 6)
            return block

        plan, full = ShardedSweep(compute, rank, world).run(args, N0, gather=True)
        want = np.stack([om.grid_sweep(oracle.OP.COMPLETE, p, spec.extent, N0, N1) for p in args])
        assert tuple(full.shape) == want.shape, (full.shape, want.shape)
        assert np.array_equal(full.numpy(), want, equal_nan=True)

        # the gather without copies: a compute step that can write into a given tensor (what HipCompute does on a GPU)
        # sweeps straight into its slice of the one result buffer, and the all-gather runs in place on it
        class WritesInPlace:
            calls = []

            def allocates(self, outer):
                import torch

                return torch.full((*outer, N1, 6), -1.0, dtype=torch.float64)

            def __call__(self, p_rows, row_begin, row_count, out=None):
                import torch

                block = torch.from_numpy(compute(p_rows, row_begin, row_count))
                self.calls.append(out is not None)
                if out is None:
                    return block
                out.copy_(block)
                return out

        step = WritesInPlace()
        plan_i, full_i = ShardedSweep(step, rank, world).run(args, N0, gather=True)
        assert plan_i == plan and np.array_equal(full_i.numpy(), want, equal_nan=True)
        n_items = P if plan.axis == "param" else N0
        assert step.calls == [n_items % world == 0 and (plan.axis == "param" or P == 1)], step.calls  # in place exactly when blocks are equal and contiguous
        plan2, local = ShardedSweep(compute, rank, world).run(args, N0, gather=False)
        if plan2.axis == "param":
            assert np.array_equal(local, want[plan2.p_begin : plan2.p_begin + plan2.p_count], equal_nan=True)
        else:
            assert np.array_equal(local, want[:, plan2.row_begin : plan2.row_begin + plan2.row_count], equal_nan=True)
        # statistics instead of arrays: per-rank summaries combined by three six-element all-reduces
        combined = all_reduce_summary(numpy_summary(local))
        whole = numpy_summary(want)
        assert np.array_equal(combined["min"], whole["min"]) and np.array_equal(combined["max"], whole["max"])
        assert np.array_equal(combined["count"], whole["count"])
        open(os.path.join(tmpdir, f"ok_{rank}"), "w").write(plan.axis)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", [("hyperbolic", 3, 12, 10), ("doc", 1, 13, 9), ("hyperbolic", 4, 6, 10), ("doc", 1, 14, 9)], ids=["param-axis", "row-axis", "param-axis-equal-blocks", "row-axis-equal-blocks"])
def test_two_gloo_ranks_shard_and_gather(case, tmp_path):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(2, port, case, str(tmp_path)), nprocs=2, join=True)
    axes = {open(tmp_path / f"ok_{r}").read() for r in range(2)}
    assert axes == ({"param"} if case[1] >= 2 else {"rows"})


def _config4_worker(rank, world, port, tmpdir):
    """One of eight gloo ranks of BASELINE configs[4]'s plan: 512 parameter rows (L in linspace(0.2, 2.0, 512), bench.py's axis)
    over 8 ranks = 64 rows each, the CPU oracle as the compute step on a reduced field grid."""
    import torch
    import torch.distributed as dist

    import oracle
    import workloads
    from workloads import example_models

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        spec = example_models.get("hyperbolic")
        src, _ = oracle.emit_c_source(workloads.model_for("hyperbolic"), **spec.compiler_kwargs)
        om = oracle.OracleModel(oracle.compile_c_model(src))
        P, N0, N1 = 512, 6, 5
        args = np.tile(np.asarray(spec.args, dtype=np.float64), (P, 1))
        args[:, -1] = 0.2 + (2.0 - 0.2) * np.arange(P) / 511.0
        pts = oracle.grid_points(spec.extent, N0, N1)

        class Compute:
            def allocates(self, outer):
                return torch.full((*outer, N1, 6), -1.0, dtype=torch.float64)

            def __call__(self, p_rows, row_begin, row_count, out=None):
                assert (row_begin, row_count) == (0, N0) and len(p_rows) == P // world  # the parameter axis is what is split
                block = np.stack([om.trajectory_sweep(oracle.OP.COMPLETE, p, pts).reshape(N0, N1, 6) for p in p_rows])
                if out is None:
                    return torch.from_numpy(block)
                out.copy_(torch.from_numpy(block))
                return out

        sweep = ShardedSweep(Compute(), rank, world)
        plan, full = sweep.run(args, N0, gather=True)
        assert (plan.axis, plan.p_begin, plan.p_count, plan.row_begin, plan.row_count) == ("param", 64 * rank, 64, 0, N0)
        assert tuple(full.shape) == (P, N0, N1, 6)
        # every rank holds every row: spot-check a row of each rank's block against this rank's own oracle, and its own block in full
        for r in range(world):
            k = 64 * r + (7 * rank) % 64
            assert np.array_equal(full[k].numpy(), om.grid_sweep(oracle.OP.COMPLETE, args[k], spec.extent, N0, N1), equal_nan=True), (rank, k)
        _, local = sweep.run(args, N0, gather=False)
        assert np.array_equal(local.numpy(), full[plan.p_begin : plan.p_begin + plan.p_count].numpy(), equal_nan=True)
        combined = all_reduce_summary(numpy_summary(local.numpy()))
        whole = numpy_summary(full.numpy())
        assert np.array_equal(combined["min"], whole["min"]) and np.array_equal(combined["max"], whole["max"]) and np.array_equal(combined["count"], whole["count"])
        assert int(combined["count"][1]) == P * N0 * N1  # epsilon_V is finite at every point of every row
        open(os.path.join(tmpdir, f"ok_{rank}"), "w").write(f"{plan.p_begin}:{plan.p_count}")
    finally:
        dist.destroy_process_group()


def test_eight_gloo_ranks_on_the_plan_of_config4(tmp_path):
    """BASELINE configs[4] (512 parameter rows over 8 GPUs) as far as a CPU container can rehearse it: eight gloo ranks, the
    partition of plan_shard (64 rows each), the in-place all-gather of ShardedSweep and the three six-element all-reduces of the
    summary -- the process-group code of the one-process-per-GPU form -- with the oracle standing in for the HIP sweep."""
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_config4_worker, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    assert sorted(open(tmp_path / f"ok_{r}").read() for r in range(8)) == sorted(f"{64 * r}:64" for r in range(8))


# ---- the same two-rank exercise with the HIP sweep as the local compute step ---------------------------
# (GPU box: both ranks use the one visible GPU and exchange CPU tensors over gloo, because RCCL refuses
# two ranks per device; partition, per-rank launch through the C ABI, gather and summary are the product's)
def _gpu_worker(rank, world, port, case, tmpdir):
    import torch
    import torch.distributed as dist

    from inflatox_amd import _native

    import workloads
    from inflatox_amd.distributed import HipCompute

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        name, P, N0, N1 = case
        spec, art = workloads.artifact_for(name)
        lib = _native.InflatoxDevLib(art.shared_object_path, device=0)
        args = np.stack([spec.args * (1.0 + 0.1 * k) for k in range(P)])
        hip = HipCompute(lib, spec.extent, N0, N1)

        def compute(p_rows, row_begin, row_count):
            block = hip(p_rows, row_begin, row_count)
            torch.cuda.synchronize()
            return block.cpu()

        plan, full = ShardedSweep(compute, rank, world).run(args, N0, gather=True)
        want = lib.sweep_host(_native.OP_COMPLETE, args, np.array(spec.extent).reshape(2, 2), N0, N1)
        assert tuple(full.shape) == want.shape, (full.shape, want.shape)
        assert np.array_equal(full.numpy(), want, equal_nan=True)
        # summaries: every rank reduces its own block on the device, the ranks combine over the process group
        if plan.axis == "param":
            local = lib.sweep_stats(args[plan.p_begin : plan.p_begin + plan.p_count], spec.extent, N0, N1)
            combined = all_reduce_summary(local)
            whole = numpy_summary(want)
            assert np.array_equal(combined["count"], whole["count"])
            assert np.array_equal(combined["min"], whole["min"]) and np.array_equal(combined["max"], whole["max"])
        open(os.path.join(tmpdir, f"ok_{rank}"), "w").write(plan.axis)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("case", [("hyperbolic", 3, 96, 80), ("doc", 1, 101, 72), ("d5", 2, 64, 64)], ids=["param-axis", "row-axis", "d5-param-axis"])
def test_two_ranks_share_the_gpu_with_hip_compute(case, tmp_path):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_gpu_worker, args=(2, port, case, str(tmp_path)), nprocs=2, join=True)
    axes = {open(tmp_path / f"ok_{r}").read() for r in range(2)}
    assert axes == ({"param"} if case[1] >= 2 else {"rows"})
