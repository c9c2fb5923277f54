"""Multi-GPU layer of the hot path: independent PDE samples shard across the
ranks of one node, weights are replicated, the forward needs NO collective; training adds ONE
flat gradient all-reduce per step (<= 5.3 M fp32 = 21 MB for the 1024^2 kernel MLP -> a single
bucket; over xGMI a ring all-reduce of 21 MB is ~0.24 ms against >= 1 s of compute per sample, so
there is nothing to overlap).  One process per GPU; backend 'nccl' is RCCL on ROCm, 'gloo' on CPU
(tests).  The reference has no distributed code at all.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*; returns
    (rank, world, local_rank).  No-op for a single process."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced shard of `n_items` independent units (samples / sub-graphs):
    sizes differ by at most one, every item belongs to exactly one rank."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


def shard(items: Sequence, rank: int, world: int) -> List:
    return [items[i] for i in shard_range(len(items), rank, world)]


def allreduce_gradients(params: Iterable[torch.nn.Parameter], world: int | None = None,
                        group=None, average: bool = True) -> int:
    """Sum (or average) `p.grad` of every parameter across ranks with ONE flat all-reduce.
    Parameters without a gradient on this rank contribute zeros so all ranks agree on the layout; one
    extra element per parameter carries "some rank had a gradient", and a parameter that had none on
    EVERY rank keeps `grad = None` - with the reference optimizer (Adam, weight_decay=5e-4,
    UAI1_full_resolution.py:242) a zero gradient would still decay the weight and create Adam state, and
    the training result would depend on the world size.  Returns the number of gradient elements reduced."""
    params = [p for p in params if p.requires_grad]
    if not params:
        return 0
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    n_grad = sum(p.numel() for p in params)
    if world == 1:
        return n_grad
    dev, dt = params[0].device, params[0].dtype
    flat = torch.zeros(n_grad + len(params), device=dev, dtype=dt)
    off = 0
    for i, p in enumerate(params):
        if p.grad is not None:
            flat[off:off + p.numel()].copy_(p.grad.reshape(-1))
            flat[n_grad + i] = 1.0
        off += p.numel()
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    # "some rank had a gradient" only needs reading for parameters WITHOUT a local gradient (a local one already contributed
    # its 1): the usual step - every parameter has a gradient on every rank - makes no device->host copy, hence no sync
    # (visible at the MGKN configurations' 24 ms steps)
    missing = [i for i, p in enumerate(params) if p.grad is None]
    has = flat[n_grad:].cpu() if missing else None
    if average:
        flat[:n_grad] /= world
    off = 0
    for i, p in enumerate(params):
        if has is None or p.grad is not None or float(has[i]) > 0.0:
            g = flat[off:off + p.numel()].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
        off += p.numel()
    return n_grad


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Replicate rank `src`'s weights (start of a data-parallel run): ONE flat broadcast per (device, dtype), then
    `t.copy_(...)` under no_grad into every parameter / buffer.  The copy is what bumps the tensors' version counters -
    a c10d collective writing into `t.detach()` does NOT (`_version` stays put), and the packed-weight /
    hidden-activation caches of graph_pde_amd.ops key on those counters: a forward that ran before the broadcast would
    otherwise keep serving the pre-broadcast packed weights on the non-source ranks.  The caches are dropped as well."""
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return
    groups = {}
    for t in list(module.parameters()) + list(module.buffers()):
        groups.setdefault((t.device, t.dtype), []).append(t)
    with torch.no_grad():
        for (dev, dt), ts in groups.items():
            flat = torch.cat([t.detach().reshape(-1) for t in ts]) if ts else None
            dist.broadcast(flat, src=src, group=group)
            off = 0
            for t in ts:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
    try:                                    # belt and braces: nothing packed from the old values survives
        from . import hidden_cache, ops
        ops.clear_caches()
        hidden_cache.clear()
    except Exception:                       # the helper is also usable on plain modules without the native library
        pass


# ----------------------------------------------------------------------------------------------------------------------
# ONE graph over several GPUs: the exchange step of SURVEY.md §8(e), way 2
# ----------------------------------------------------------------------------------------------------------------------
# A single sample too slow (or too big) for one GPU - the 241^2 graph is 95.5 M edges, 0.5 s per NNConv layer on one
# MI355X - is split by DESTINATION rows: rank r owns a contiguous node range [lo_r, hi_r) holding ~E / world in-edges
# (balanced on the in-degree prefix sum, not on node counts: boundary nodes of a radius graph have half the in-edges).
# x [N, 64] is replicated (15 MB at N = 58,081); a layer is: NNConv over the rank's own in-edges -> its rows of the
# result -> ONE all-gather of the [hi_r - lo_r, 64] blocks (RCCL over xGMI: 15 MB in total per layer at G241, ~0.1 ms
# against 63 ms of compute per rank at world 8) -> the next layer's replicated x.  The per-node arithmetic is the
# single-GPU operator's: a node's in-edges stay together, in the caller's order.
#
# Training: everything outside the conv (fc1 / fc2 / loss of the GKN stack, UAI1_full_resolution.py:27-33) is computed
# redundantly on every rank, so d loss / d (gathered rows) is the same tensor everywhere.  `_GatherRows.backward` hands
# the conv `world` x its own rows of it, `_ReplicatedInput.backward` all-reduces the dx contributions of the ranks' edge
# sets and divides by `world`: conv parameter gradients are then `world` x this rank's part, replicated parameters carry
# the full gradient on every rank, and the SAME `allreduce_gradients(average=True)` call as in the sample-sharded run
# turns both into the exact gradient (sum of parts / identical copies).


class RowPartition:
    """This rank's destination-row block of one graph (make it with `partition_rows`)."""

    def __init__(self, n_nodes, bounds, rank, edge_index, edge_attr, group=None, csr=None):
        self.n_nodes = int(n_nodes)
        self.bounds = [int(b) for b in bounds]          # world + 1 node offsets, bounds[0] = 0, bounds[-1] = N
        self.rank = int(rank)
        self.world = len(self.bounds) - 1
        self.lo, self.hi = self.bounds[self.rank], self.bounds[self.rank + 1]
        self.edge_index = edge_index                    # [2, E_r] global node ids, the caller's edge order
        self.edge_attr = edge_attr                      # [E_r, k0]
        self.group = group
        self.csr = csr                                  # partition_rows_by_position: the block as a destination CSR over all N
                                                        # nodes (ops.Csr; edge_attr rows are in its slot order) - no sort needed

    @property
    def n_edges(self) -> int:
        return int(self.edge_index.shape[1])

    def __repr__(self):
        return (f"RowPartition(rank {self.rank}/{self.world}, rows [{self.lo}, {self.hi}) of {self.n_nodes}, "
                f"{self.n_edges} in-edges)")


def row_bounds(edge_index: torch.Tensor, n_nodes: int, world: int) -> List[int]:
    """Node offsets of `world` contiguous destination ranges with (nearly) equal in-edge counts: range r ends at the
    first node where the in-degree prefix sum reaches (r + 1) E / world.  Deterministic in (edge_index, n_nodes, world):
    every rank computes the same list.  Ranges may be empty (more ranks than nodes with in-edges)."""
    if world < 1:
        raise ValueError("world must be >= 1")
    if world == 1:
        return [0, n_nodes]
    if int(edge_index.shape[1]) == 0:
        return bounds_from_degrees(torch.zeros(n_nodes, dtype=torch.int64), world)
    return bounds_from_degrees(torch.bincount(edge_index[1].reshape(-1).to(torch.int64), minlength=n_nodes), world)


def bounds_from_degrees(deg: torch.Tensor, world: int) -> List[int]:
    """`row_bounds` given the in-degree of every node ([N] integer tensor) instead of the edge list."""
    n_nodes = int(deg.numel())
    if world < 1:
        raise ValueError("world must be >= 1")
    if world == 1:
        return [0, n_nodes]
    deg = deg.reshape(-1).to(torch.int64)
    e = int(deg.sum())
    if e == 0:                                      # nothing to balance: equal node counts
        return [min(n_nodes, -(-n_nodes * r // world)) for r in range(world)] + [n_nodes]
    cum = torch.cumsum(deg, 0)
    targets = torch.tensor([-(-e * r // world) for r in range(1, world)], dtype=cum.dtype, device=cum.device)
    cuts = (torch.searchsorted(cum, targets, right=False) + 1).clamp(max=n_nodes).tolist()
    bounds = [0]
    for c in cuts:
        bounds.append(max(int(c), bounds[-1]))
    return bounds + [n_nodes]


def partition_rows(edge_index: torch.Tensor, edge_attr: torch.Tensor, n_nodes: int, rank: int | None = None,
                   world: int | None = None, group=None) -> RowPartition:
    """Keep the edges whose DESTINATION lies in this rank's row block (order preserved: a node's in-edges are summed in
    the order the single-GPU operator sums them).  `edge_index` [2, E] with row 1 = target i (nn_conv.py:271 /
    SURVEY.md App. B), `edge_attr` [E] or [E, k0].  Called with the full graph on every rank."""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world {world}")
    if edge_index.dim() != 2 or edge_index.shape[0] != 2 or edge_attr.shape[0] != edge_index.shape[1]:
        raise ValueError(f"edge_index [2, E] and edge_attr [E, ...] expected, got {tuple(edge_index.shape)} / {tuple(edge_attr.shape)}")
    bounds = row_bounds(edge_index, n_nodes, world)
    if world == 1:
        return RowPartition(n_nodes, bounds, 0, edge_index, edge_attr, group)
    # filtered in pieces of 2^24 edges: one-shot indexing of tensors above 2^26 rows is not trusted on this torch / ROCm build
    # (synth.darcy_edge_attr found zeroed rows there), and the pieces bound the nonzero / index temporaries
    e, step = int(edge_index.shape[1]), 1 << 24
    ei_parts, ea_parts = [], []
    for a0 in range(0, e, step):
        d_ = edge_index[1, a0:a0 + step]
        keep = ((d_ >= bounds[rank]) & (d_ < bounds[rank + 1])).nonzero().reshape(-1)
        ei_parts.append(edge_index[:, a0:a0 + step].index_select(1, keep))
        ea_parts.append(edge_attr[a0:a0 + step].index_select(0, keep))
    if not ei_parts:
        ei_parts, ea_parts = [edge_index[:, :0]], [edge_attr[:0]]
    return RowPartition(n_nodes, bounds, rank, torch.cat(ei_parts, 1).contiguous(), torch.cat(ea_parts, 0).contiguous(), group)


def partition_rows_by_position(pos: torch.Tensor, r: float, node_attr=None, rank: int | None = None, world: int | None = None,
                               group=None, reference_ties: bool = False, degrees_fn=None, block_fn=None) -> RowPartition:
    """This rank's destination-row block of the RADIUS graph of `pos` [N, dim] built from the positions alone - no rank ever
    holds the whole edge list (the 241^2 graph: 1.5 GB of int64 indices + 2.3 GB of attributes, then nonzero / index_select
    temporaries, per rank in `partition_rows`).  Every rank runs the COUNT pass of the cell-list builder over all nodes
    (in-degrees only: ops.radius_in_degrees), derives the same balanced bounds from them, and runs the FILL pass for its own
    destinations only (ops.radius_csr_raw(pos, r, pos_dst=pos[lo:hi])).  The block comes out as a destination CSR over all
    N nodes (rows outside [lo, hi) empty, sources ascending inside a row = the single-GPU operator's summation order).
    `node_attr`: an ops.NodeAttr (the reference's recipe edge_attr = [pos_src, pos_dst, a_src, a_dst], utilities.py:274-277):
    the block's edge attributes are materialised from it in slot order; None leaves `edge_attr` to the caller
    (`part.csr.edge_index` lists the block's edges).  `degrees_fn(pos, r)` / `block_fn(pos, r, pos_dst)` replace the two
    native passes (CPU tests)."""
    from . import ops
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world {world}")
    pos2 = pos.unsqueeze(1) if pos.dim() == 1 else pos
    n = int(pos2.shape[0])
    deg = (degrees_fn or (lambda p, rr: ops.radius_in_degrees(p, rr, reference_ties)))(pos2, r)
    bounds = bounds_from_degrees(deg.cpu(), world)
    lo, hi = bounds[rank], bounds[rank + 1]
    build = block_fn or (lambda p, rr, pd: ops.radius_csr_raw(p, rr, reference_ties, pos_dst=pd))
    dev = pos2.device
    if hi > lo:
        rp_l, src, dst_l = build(pos2, r, pos2[lo:hi])
    else:
        rp_l = torch.zeros(1, dtype=torch.int32, device=dev)
        src = torch.zeros(0, dtype=torch.int32, device=dev)
        dst_l = src.clone()
    e = int(src.numel())
    rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    rowptr[:lo + 1] = 0
    rowptr[lo:hi + 1] = rp_l.to(torch.int32)
    rowptr[hi:] = e
    dst = (dst_l + lo).to(torch.int32)
    csr = ops.Csr(n, e, rowptr, src.to(torch.int32), dst, torch.arange(e, dtype=torch.int32, device=dev))
    csr._perm_is_identity = True
    ei = csr.edge_index
    ea = None if node_attr is None else node_attr.materialize(ei)
    return RowPartition(n, bounds, rank, ei, ea, group, csr=csr)


def _gather_blocks(local: torch.Tensor, part: RowPartition) -> torch.Tensor:
    """All ranks' row blocks, in rank order, as one [N, ...] tensor: ONE flat all-gather of equal-size (padded) pieces into a
    single buffer (RCCL / gloo `all_gather_into_tensor`); the padding rows are cut out only when the blocks differ in size."""
    sizes = [part.bounds[r + 1] - part.bounds[r] for r in range(part.world)]
    mx = max(sizes)
    if local.shape[0] == mx:
        pad = local.contiguous()
    else:
        pad = local.new_zeros((mx,) + tuple(local.shape[1:]))
        pad[:local.shape[0]].copy_(local)
    buf = local.new_empty((part.world * mx,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(buf, pad, group=part.group)
    if min(sizes) == mx:
        return buf
    return torch.cat([buf[r * mx: r * mx + sizes[r]] for r in range(part.world)], 0)


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, local, part):
        ctx.part = part
        return _gather_blocks(local.contiguous(), part)

    @staticmethod
    def backward(ctx, grad_full):
        p = ctx.part
        return grad_full[p.lo:p.hi] * float(p.world), None


class _ReplicatedInput(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, part):
        ctx.part = part
        return x.view_as(x)

    @staticmethod
    def backward(ctx, grad):
        p = ctx.part
        g = grad.contiguous().clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=p.group)
        return g / float(p.world), None


def nnconv_rows(conv, x: torch.Tensor, part: RowPartition) -> torch.Tensor:
    """One NNConv layer of a row-partitioned graph: `x` [N, in] replicated on every rank -> `conv` over this rank's
    in-edges -> its rows [lo, hi) -> all-gather -> the full [N, out] on every rank.  `conv` is called as the reference
    calls it, `conv(x, edge_index, edge_attr)` (nn_conv.py:267); rows outside the block have no in-edge in the rank's
    edge set (their `x . root + bias` is computed and dropped: 64 x 64 per node).  Differentiable; with world 1 (or no
    process group) it is `conv(x, edge_index, edge_attr)`."""
    if x.shape[0] != part.n_nodes:
        raise ValueError(f"x has {x.shape[0]} rows, the partitioned graph {part.n_nodes} nodes")
    graph = part.edge_index if part.csr is None else part.csr      # a prebuilt block CSR goes to the operator as it is
    if part.world == 1:
        return conv(x, graph, part.edge_attr)
    if not dist.is_initialized():
        raise RuntimeError("nnconv_rows with world > 1 needs an initialised process group (parallel.init_from_env)")
    x_in = _ReplicatedInput.apply(x, part) if (torch.is_grad_enabled() and x.requires_grad) else x
    out = conv(x_in, graph, part.edge_attr)
    return _GatherRows.apply(out[part.lo:part.hi], part)
