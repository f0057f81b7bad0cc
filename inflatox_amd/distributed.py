"""Sharding of a sweep over the GPUs of one node (one process per GPU, torch.distributed).

The reference has no distributed code at all: its only parallelism is rayon threads over grid
points inside one process (src/anguelova.rs:524-540).  Every grid point and every parameter row is
independent, so the sweep shards without any exchange on the data path:

  * the outermost axis is split into contiguous, balanced blocks, one per rank -- the parameter axis
    when there are at least as many parameter rows as ranks, the grid's row axis otherwise;
  * each rank sweeps its block into its own HBM (``InflatoxDevLib.sweep_device``);
  * only if the caller wants the whole result on every GPU is one ``all_gather`` issued
    (RCCL over xGMI on GPUs, i.e. torch.distributed backend "nccl"; "gloo" on CPU tensors in tests).
    The gather moves (world-1)/world of the result through each GPU's links and costs far more than
    the sweep itself (Multi-GPU), which is why it is opt-in.

The compute step is injected as a callable so that the partition/gather logic can be exercised on
CPU ranks (tests/test_distributed.py uses the CPU oracle as the callable; the product passes the
HIP sweep).
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import _native


def block_bounds(n_items: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous balanced partition: the first ``n_items % world`` ranks get one item more."""
    base, extra = divmod(n_items, world)
    begin = rank * base + min(rank, extra)
    return begin, base + (1 if rank < extra else 0)


@dataclass(frozen=True)
class ShardPlan:
    axis: str  # "param" or "rows"
    p_begin: int
    p_count: int
    row_begin: int
    row_count: int

    @property
    def empty(self) -> bool:
        return self.p_count == 0 or self.row_count == 0


def plan_shard(P: int, N0: int, world: int, rank: int) -> ShardPlan:
    """Which block of the (P, N0) outer index space does ``rank`` own?"""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of size {world}")
    if P >= world:
        b, c = block_bounds(P, world, rank)
        return ShardPlan("param", b, c, 0, N0)
    b, c = block_bounds(N0, world, rank)
    return ShardPlan("rows", 0, P, b, c)


class ShardedSweep:
    """Runs one rank's share of a (P parameter rows) x (N0 x N1 grid) sweep and optionally gathers it.

    ``compute(p_rows, row_begin, row_count) -> array (p_count, row_count, N1, K)``: the local sweep.
    ``HipCompute`` below is the product's implementation.
    """

    def __init__(self, compute, rank: int = 0, world: int = 1, process_group=None):
        self.compute = compute
        self.rank = rank
        self.world = world
        self.group = process_group

    def run(self, args2d, N0: int, gather: bool = False, force_collective: bool = False):
        """Returns ``(plan, local_block)`` or, with ``gather``, ``(plan, full)`` where ``full`` has the
        shape (P, N0, N1, K) on every rank.  A world of one rank needs no exchange and returns its block as it is;
        ``force_collective`` sends it through the all-gather all the same (how the RCCL code path is exercised on a
        one-GPU box: one rank, backend "nccl").

        Gather without copies: when every rank owns a block of the same size that is one contiguous piece of the result
        (the parameter axis divides evenly -- BASELINE configs[4]: 512 rows over 8 GPUs --, or a single parameter row
        whose grid rows divide evenly) and the compute step can write into a given tensor (``compute.allocates`` /
        ``out=``: :class:`HipCompute`), the result is allocated once, the rank sweeps straight into its slice of it and the
        all-gather runs in place on that buffer: no padded send copy, no ``world x`` receive buffer, no ``torch.cat``.
        Unequal blocks and row-split batches take the padded path."""
        import torch

        args2d = np.atleast_2d(np.asarray(args2d, dtype=np.float64))
        P = args2d.shape[0]
        plan = plan_shard(P, N0, self.world, self.rank)
        my_rows = args2d[plan.p_begin : plan.p_begin + plan.p_count]
        if not gather or (self.world == 1 and not force_collective):
            return plan, self.compute(my_rows, plan.row_begin, plan.row_count)
        import torch.distributed as dist

        n_items = P if plan.axis == "param" else N0
        axis = 0 if plan.axis == "param" else 1
        in_place = getattr(self.compute, "allocates", None) is not None and n_items % self.world == 0 and (plan.axis == "param" or P == 1)
        if in_place:
            full = self.compute.allocates((P, N0))  # (P, N0, N1, K), uninitialised, on the compute step's device
            mine = full.narrow(axis, plan.p_begin if axis == 0 else plan.row_begin, n_items // self.world)
            self.compute(my_rows, plan.row_begin, plan.row_count, out=mine)
            # rank r's block is elements [r * count, (r + 1) * count) of the flat result: the in-place form of the all-gather
            dist.all_gather_into_tensor(full.view(-1), mine.reshape(-1), group=self.group)
            return plan, full
        local = self.compute(my_rows, plan.row_begin, plan.row_count)
        local_t = local if isinstance(local, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(local))
        # equal-sized contributions: pad every block to the largest one along the sharded axis
        biggest = block_bounds(n_items, self.world, 0)[1]
        pad_shape = list(local_t.shape)
        pad_shape[axis] = biggest
        if pad_shape == list(local_t.shape):
            send = local_t.contiguous()
        else:
            send = local_t.new_zeros(pad_shape)
            send.narrow(axis, 0, local_t.shape[axis]).copy_(local_t)
        recv = send.new_empty([self.world] + pad_shape)
        # flat views: the one calling convention both RCCL ("nccl") and gloo accept
        dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=self.group)
        if axis == 0 and n_items % self.world == 0:
            return plan, recv.view([P] + pad_shape[1:])  # equal parameter blocks: the receive buffer IS the result
        pieces = []
        for r in range(self.world):
            _, count = block_bounds(n_items, self.world, r)
            pieces.append(recv[r].narrow(axis, 0, count))
        return plan, torch.cat(pieces, dim=axis)


class HipCompute:
    """The product's local sweep: one ``inflx_sweep_device`` launch into a torch CUDA tensor."""

    def __init__(self, devlib: _native.InflatoxDevLib, extent, N0: int, N1: int, op: int = _native.OP_COMPLETE):
        import torch

        self.lib = devlib
        self.extent = extent
        self.N0, self.N1, self.op = N0, N1, op
        # a torch-owned stream for the sweep: torch's default stream has the NULL handle, which the C ABI
        # reads as "use the model's own stream", and that one is not ordered with torch's work
        self.stream = torch.cuda.Stream(device=f"cuda:{devlib.device}")

    def allocates(self, outer):
        """An uninitialised result tensor ``(*outer, N1, K)`` on this object's device (the gather buffer of
        ``ShardedSweep.run``: the sweep writes every element of the slice it is given)."""
        import torch

        return torch.empty((*outer, self.N1, _native.OP_WIDTH[self.op]), dtype=torch.float64, device=torch.device(f"cuda:{self.lib.device}"))

    def __call__(self, p_rows, row_begin, row_count, out=None):
        """Sweep ``p_rows`` x grid rows [row_begin, row_begin + row_count) into a new tensor, or into ``out`` -- a
        contiguous (len(p_rows), row_count, N1, K) tensor on this device, e.g. the rank's slice of a gather buffer."""
        import torch

        k = _native.OP_WIDTH[self.op]
        device = torch.device(f"cuda:{self.lib.device}")
        shape = (len(p_rows), row_count, self.N1, k)
        if out is None:
            out = torch.empty(shape, dtype=torch.float64, device=device)
        elif tuple(out.shape) != shape or not out.is_contiguous() or out.dtype != torch.float64 or out.device != device:
            raise ValueError(f"out must be a contiguous float64 tensor of shape {shape} on {device}")
        if out.numel():
            consumer = torch.cuda.current_stream(device)
            self.stream.wait_stream(consumer)  # `out` was allocated on the consumer's stream
            self.lib.sweep_device(
                self.op,
                p_rows,
                out.data_ptr(),
                out.numel() * 8,
                self.extent,
                self.N0,
                self.N1,
                row_begin=row_begin,
                row_count=row_count,
                stream=self.stream.cuda_stream,
            )
            consumer.wait_stream(self.stream)  # whatever torch does next with `out` (e.g. the all-gather) is ordered after the sweep
            out.record_stream(self.stream)
        return out


def numpy_summary(block: np.ndarray) -> dict:
    """The summary of a (..., 6) result block computed with numpy (what the device reduction must equal)."""
    flat = np.asarray(block, dtype=np.float64).reshape(-1, 6)
    ok = ~np.isnan(flat)
    with np.errstate(all="ignore"):
        mn = np.where(ok.any(axis=0), np.nanmin(np.where(ok, flat, np.inf), axis=0), np.inf)
        mx = np.where(ok.any(axis=0), np.nanmax(np.where(ok, flat, -np.inf), axis=0), -np.inf)
    return {"min": mn, "max": mx, "count": ok.sum(axis=0).astype(np.uint64)}


def all_reduce_summary(summary: dict, device=None, group=None) -> dict:
    """Combine the per-rank summaries of a sharded sweep: three tiny all-reduces (MIN, MAX, SUM) of six
    numbers each -- the only exchange a sharded sweep needs when the caller wants statistics rather than
    the arrays (RCCL when ``device`` is a GPU, gloo on the CPU)."""
    import torch
    import torch.distributed as dist

    mn = torch.tensor(np.asarray(summary["min"], dtype=np.float64), device=device)
    mx = torch.tensor(np.asarray(summary["max"], dtype=np.float64), device=device)
    cnt = torch.tensor(np.asarray(summary["count"]).astype(np.int64), device=device)
    dist.all_reduce(mn, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
    return {"min": mn.cpu().numpy(), "max": mx.cpu().numpy(), "count": cnt.cpu().numpy().astype(np.uint64)}
