"""world_size-2 `gloo` test of the sample-sharding / gradient all-reduce layer (CPU tier).
The forward of the hot path needs no collective; this covers the N>1 plumbing that bench.py and a
data-parallel training step use."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from graph_pde_amd import parallel
    r, w, _ = parallel.init_from_env("gloo")
    assert (r, w) == (rank, world)
    # 1. every sample belongs to exactly one rank, shards balanced
    n_samples = 7
    mine = list(parallel.shard_range(n_samples, rank, world))
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    flat = sorted(sum(gathered, []))
    assert flat == list(range(n_samples)), flat
    assert max(map(len, gathered)) - min(map(len, gathered)) <= 1
    # 2. replicated weights + flat gradient all-reduce == gradient of the mean loss over all samples
    torch.manual_seed(100 + rank)                      # deliberately different init per rank
    model = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.ReLU(), torch.nn.Linear(8, 3))
    v0 = [p._version for p in model.parameters()]
    before = [p.detach().clone() for p in model.parameters()]
    parallel.broadcast_parameters(model, src=0)
    # the pack / hidden caches of graph_pde_amd.ops key on the version counters: every tensor the broadcast rewrote
    # must show a new version (a c10d collective alone leaves `_version` untouched)
    assert all(p._version > v for p, v in zip(model.parameters(), v0)), [p._version for p in model.parameters()]
    if rank != 0:
        assert any(not torch.equal(a, p.detach()) for a, p in zip(before, model.parameters()))
    g = torch.Generator().manual_seed(0)
    xs = torch.randn(n_samples, 5, 6, generator=g)
    ys = torch.randn(n_samples, 5, 3, generator=g)
    loss = sum(((model(xs[i]) - ys[i]) ** 2).mean() for i in mine) / max(len(mine), 1)
    model.zero_grad()
    loss.backward()
    # weight each rank's mean by its sample count so the all-reduced result is the global mean
    for p in model.parameters():
        p.grad.mul_(len(mine) * world / n_samples)
    n = parallel.allreduce_gradients(model.parameters(), average=True)
    assert n == sum(p.numel() for p in model.parameters())
    ref = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.ReLU(), torch.nn.Linear(8, 3))
    ref.load_state_dict(model.state_dict())
    ref_loss = sum(((ref(xs[i]) - ys[i]) ** 2).mean() for i in range(n_samples)) / n_samples
    ref_loss.backward()
    err = max(float((p.grad - q_.grad).abs().max()) for p, q_ in zip(model.parameters(), ref.parameters()))
    q.put((rank, err))
    dist.destroy_process_group()


def test_sample_sharding_and_gradient_allreduce_gloo_ws2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r for r, _ in res) == [0, 1]
    assert all(err < 1e-6 for _, err in res), res


def test_shard_range_covers_everything():
    from graph_pde_amd.parallel import shard_range
    for n in (0, 1, 5, 8, 256):
        for w in (1, 2, 3, 8):
            got = sorted(i for r in range(w) for i in shard_range(n, r, w))
            assert got == list(range(n))


# ---- one graph over two ranks: destination-row partition + all-gather (way 2) ------------------------
class _TinyConv(torch.nn.Module):
    """NNConv_old's formula (nn_conv.py:267-282, mean aggregation) in plain torch ops - a CPU stand-in for the native
    operator, so that the partition / exchange / gradient logic of parallel.nnconv_rows runs in the CPU tier."""

    def __init__(self, c=4, k0=3):
        super().__init__()
        self.c = c
        self.nn = torch.nn.Sequential(torch.nn.Linear(k0, 8), torch.nn.ReLU(), torch.nn.Linear(8, c * c))
        self.root = torch.nn.Parameter(torch.randn(c, c) * 0.3)
        self.bias = torch.nn.Parameter(torch.randn(c) * 0.1)

    def forward(self, x, edge_index, edge_attr):
        w = self.nn(edge_attr).view(-1, self.c, self.c)
        m = torch.matmul(x.index_select(0, edge_index[0]).unsqueeze(1), w).squeeze(1)
        out = torch.zeros(x.size(0), self.c, dtype=x.dtype).index_add(0, edge_index[1], m)
        cnt = torch.bincount(edge_index[1], minlength=x.size(0)).clamp(min=1).to(x.dtype)
        return out / cnt.unsqueeze(1) + x @ self.root + self.bias


def _rows_problem():
    g = torch.Generator().manual_seed(5)
    n, e = 37, 400
    ei = torch.stack([torch.randint(0, n, (e,), generator=g), torch.randint(0, n - 6, (e,), generator=g)])   # last nodes: no in-edges
    ei = ei[:, torch.argsort(ei[0], stable=True)]                # the reference's order: by source
    ea = torch.randn(e, 3, generator=g)
    a_in = torch.randn(n, 5, generator=g)
    y = torch.randn(n, generator=g)
    torch.manual_seed(11)
    model = torch.nn.ModuleDict({"fc1": torch.nn.Linear(5, 4), "conv": _TinyConv(), "fc2": torch.nn.Linear(4, 1)}).double()
    return n, ei, ea.double(), a_in.double(), y.double(), model


def _rows_forward(model, a_in, layer):
    h = model["fc1"](a_in)
    for _ in range(3):
        h = torch.relu(layer(h))
    return model["fc2"](h).view(-1)


def _rows_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from graph_pde_amd import parallel
    parallel.init_from_env("gloo")
    n, ei, ea, a_in, y, model = _rows_problem()
    part = parallel.partition_rows(ei, ea, n)
    sizes = [None] * world
    dist.all_gather_object(sizes, (part.lo, part.hi, part.n_edges))
    out = _rows_forward(model, a_in, lambda h: parallel.nnconv_rows(model["conv"], h, part))
    loss = torch.norm(out - y, 1)
    loss.backward()
    parallel.allreduce_gradients(model.parameters(), average=True)
    # (numpy: pickled by value - torch tensors travel through shared-memory handles that die with this process, and the parent
    # may unpickle after it has exited: ConnectionResetError / EOFError in q.get)
    q.put((rank, sizes, out.detach().numpy().copy(), {k: p.grad.numpy().copy() for k, p in model.named_parameters()}))
    dist.barrier()
    dist.destroy_process_group()


def test_row_partitioned_graph_gather_and_gradients_gloo_ws2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rows_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n, ei, ea, a_in, y, model = _rows_problem()
    ref = _rows_forward(model, a_in, lambda h: model["conv"](h, ei, ea))
    torch.norm(ref - y, 1).backward()
    (lo0, hi0, e0), (lo1, hi1, e1) = res[0][1]
    assert (lo0, hi1) == (0, n) and hi0 == lo1 and e0 + e1 == ei.shape[1]
    assert abs(e0 - e1) <= int(torch.bincount(ei[1]).max())            # balanced on in-edges up to one node's worth
    for rank, _, out, grads in res:
        assert torch.allclose(torch.from_numpy(out), ref.detach(), rtol=1e-12, atol=1e-12), rank   # every rank holds the full result
        for k, p in model.named_parameters():
            assert torch.allclose(torch.from_numpy(grads[k]), p.grad, rtol=1e-10, atol=1e-12), (rank, k)


# ---- the same from POSITIONS: no rank holds the whole edge list (parallel.partition_rows_by_position) ------------------
def _pos_problem(n_pts):
    g = torch.Generator().manual_seed(9)
    if n_pts <= 3:
        pos = torch.rand(n_pts, 2, generator=g, dtype=torch.float64) * 0.05          # one small cluster: world 4 -> an empty block
    else:
        # density varies strongly: the row blocks balanced on in-edges hold very different node counts
        pos = torch.cat([torch.rand(n_pts // 2, 2, generator=g, dtype=torch.float64) * 0.25,
                         torch.rand(n_pts - n_pts // 2, 2, generator=g, dtype=torch.float64)])
    a = torch.randn(n_pts, generator=g, dtype=torch.float64)
    a_in = torch.randn(n_pts, 5, generator=g, dtype=torch.float64)
    y = torch.randn(n_pts, generator=g, dtype=torch.float64)
    torch.manual_seed(12)
    model = torch.nn.ModuleDict({"fc1": torch.nn.Linear(5, 4), "conv": _TinyConv(k0=6), "fc2": torch.nn.Linear(4, 1)}).double()
    return pos, a, a_in, y, model


def _cpu_degrees(pos, r):
    d2 = ((pos[:, None, :] - pos[None, :, :]) ** 2).sum(-1)
    return (d2 <= r * r).sum(0).to(torch.int32)                   # in-degree of every destination (self-loop included)


def _cpu_block(pos, r, pos_dst):
    """CPU stand-in for ops.radius_csr_raw(pos, r, pos_dst=...): rows = destinations, sources ascending inside a row."""
    d2 = ((pos_dst[:, None, :] - pos[None, :, :]) ** 2).sum(-1)   # [n_dst, n_src]
    nz = (d2 <= r * r).nonzero()
    rowptr = torch.zeros(pos_dst.shape[0] + 1, dtype=torch.int32)
    rowptr[1:] = torch.cumsum(torch.bincount(nz[:, 0], minlength=pos_dst.shape[0]), 0)
    return rowptr, nz[:, 1].to(torch.int32), nz[:, 0].to(torch.int32)


def _edge_attr(pos, a, ei):
    return torch.cat([pos[ei[0]], pos[ei[1]], a[ei[0]].unsqueeze(1), a[ei[1]].unsqueeze(1)], 1)   # utilities.py:274-277


class _TableAttr:
    """ops.NodeAttr.darcy's `materialize` in float64 (the CPU tier has no native library calls)."""

    def __init__(self, pos, a):
        self.pos, self.a = pos, a

    def materialize(self, ei):
        return _edge_attr(self.pos, self.a, ei)


def _pos_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from graph_pde_amd import ops, parallel
    parallel.init_from_env("gloo")
    res = []
    for n_pts, r in ((60, 0.2), (3, 0.2)):
        pos, a, a_in, y, model = _pos_problem(n_pts)
        part = parallel.partition_rows_by_position(pos, r, _TableAttr(pos, a), degrees_fn=_cpu_degrees, block_fn=_cpu_block)
        assert isinstance(part.csr, ops.Csr) and part.csr.n_nodes == n_pts and part.csr.n_edges == part.n_edges
        conv = lambda h, graph, ea: model["conv"](h, graph.edge_index if isinstance(graph, ops.Csr) else graph, ea)
        out = _rows_forward(model, a_in, lambda h: parallel.nnconv_rows(conv, h, part))
        loss = torch.norm(out - y, 1)
        loss.backward()
        parallel.allreduce_gradients(model.parameters(), average=True)
        # (numpy: pickled by value - torch tensors travel through shared-memory handles that die with this process)
        res.append(((part.lo, part.hi, part.n_edges), part.bounds, out.detach().numpy().copy(),
                    {k: p.grad.numpy().copy() for k, p in model.named_parameters()}))
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_row_partition_from_positions_gloo_ws4_uneven_and_empty_blocks():
    """Every rank builds ONLY its block (count pass over all nodes, fill pass over its own destinations), world 4: blocks of
    very different node counts, and - 3 points on 4 ranks - an empty one; result and every gradient equal the whole-graph
    step."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 4
    procs = [ctx.Process(target=_pos_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for case, (n_pts, r) in enumerate(((60, 0.2), (3, 0.2))):
        pos, a, a_in, y, model = _pos_problem(n_pts)
        rp, src, dst = _cpu_block(pos, r, pos)
        ei = torch.stack([src.long(), dst.long()])
        ea = _edge_attr(pos, a, ei)
        ref = _rows_forward(model, a_in, lambda h: model["conv"](h, ei, ea))
        torch.norm(ref - y, 1).backward()
        blocks = [got[rk][1][case][0] for rk in range(world)]
        bounds = got[0][1][case][1]
        assert all(got[rk][1][case][1] == bounds for rk in range(world))              # every rank derived the same bounds
        assert blocks[0][0] == 0 and blocks[-1][1] == n_pts and all(blocks[k][1] == blocks[k + 1][0] for k in range(world - 1))
        assert sum(b[2] for b in blocks) == ei.shape[1]                               # every edge on exactly one rank
        sizes = [b[1] - b[0] for b in blocks]
        if n_pts == 3:
            assert 0 in sizes                                                           # more ranks than nodes: an empty block
        else:
            assert max(sizes) >= 2 * min(sizes) and min(b[2] for b in blocks) > 0      # uneven node counts, balanced edges
            assert max(b[2] for b in blocks) - min(b[2] for b in blocks) <= 2 * int(_cpu_degrees(pos, r).max())
        for rk in range(world):
            _, _, out, grads = got[rk][1][case]
            assert torch.allclose(torch.from_numpy(out), ref.detach(), rtol=1e-12, atol=1e-12), (case, rk)
            for k, p in model.named_parameters():
                assert torch.allclose(torch.from_numpy(grads[k]), p.grad, rtol=1e-10, atol=1e-12), (case, rk, k)


def test_bounds_from_degrees_with_a_hub():
    """A node holding more than two targets' worth of in-edges leaves a block empty; the bounds stay monotone and cover N."""
    from graph_pde_amd import parallel
    b = parallel.bounds_from_degrees(torch.tensor([1, 1, 50, 1, 1]), 4)
    assert b[0] == 0 and b[-1] == 5 and all(b[i] <= b[i + 1] for i in range(4)) and any(b[i] == b[i + 1] for i in range(4))
    assert parallel.bounds_from_degrees(torch.zeros(10, dtype=torch.int64), 4) == [0, 3, 5, 8, 10]


def test_row_bounds_and_partition_single_process():
    from graph_pde_amd import parallel
    g = torch.Generator().manual_seed(2)
    n, e = 50, 1000
    ei = torch.stack([torch.randint(0, n, (e,), generator=g), torch.randint(0, n, (e,), generator=g)])
    ea = torch.arange(e, dtype=torch.float32).unsqueeze(1)
    for world in (1, 2, 3, 8, 64):
        b = parallel.row_bounds(ei, n, world)
        assert len(b) == world + 1 and b[0] == 0 and b[-1] == n and all(b[i] <= b[i + 1] for i in range(world))
        seen = []
        for r in range(world):
            part = parallel.partition_rows(ei, ea, n, rank=r, world=world)
            assert part.bounds == b and (part.lo, part.hi) == (b[r], b[r + 1])
            assert bool(((part.edge_index[1] >= part.lo) & (part.edge_index[1] < part.hi)).all())
            ids = part.edge_attr[:, 0].long()
            assert torch.equal(ids, torch.sort(ids).values)              # the caller's edge order is preserved
            assert torch.equal(part.edge_index, ei[:, ids])
            seen.append(ids)
        assert torch.equal(torch.sort(torch.cat(seen)).values, torch.arange(e))   # every edge on exactly one rank
    assert parallel.row_bounds(torch.zeros(2, 0, dtype=torch.long), 10, 4) == [0, 3, 5, 8, 10]
    # world 1 / no process group: nnconv_rows is the plain call
    part = parallel.partition_rows(ei, ea, n, rank=0, world=1)
    x = torch.randn(n, 4, generator=g)
    assert parallel.nnconv_rows(lambda x_, ei_, ea_: x_ * 2.0, x, part).equal(x * 2.0)


# ---- BASELINE config 5's bookkeeping: 256 samples over 8 ranks, 32 accumulated per rank, ONE all-reduce ---------------
def _accum_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from graph_pde_amd import parallel
    parallel.init_from_env("gloo")
    n_samples = 256
    mine = list(parallel.shard_range(n_samples, rank, world))
    torch.manual_seed(7 + rank)                            # deliberately different init per rank
    model = _TinyConv().double()
    extra = torch.nn.Linear(4, 4).double()                 # a layer NO sample touches: its gradient stays None on every rank
    parallel.broadcast_parameters(model, src=0)
    params = list(model.parameters()) + list(extra.parameters())

    def sample(i):                                         # sample i: its own small graph, attributes, input and target
        g = torch.Generator().manual_seed(1000 + i)
        n, e = 6 + i % 5, 20 + i % 7
        ei = torch.stack([torch.randint(0, n, (e,), generator=g), torch.randint(0, n, (e,), generator=g)])
        return (torch.randn(n, 4, generator=g, dtype=torch.float64), ei, torch.randn(e, 3, generator=g, dtype=torch.float64),
                torch.randn(n, 4, generator=g, dtype=torch.float64))
    for p in params:
        p.grad = None
    for i in mine:                                         # gradient accumulation over the rank's 32 samples
        x, ei, ea, y = sample(i)
        (((model(x, ei, ea) - y) ** 2).mean() / len(mine)).backward()
    calls = {"n": 0}
    real = dist.all_reduce

    def counted(*a, **k):
        calls["n"] += 1
        return real(*a, **k)
    dist.all_reduce = counted
    n_red = parallel.allreduce_gradients(params, average=True)          # ONE flat all-reduce: mean over ranks of per-rank means
    dist.all_reduce = real
    got = [None if p.grad is None else p.grad.clone() for p in params]
    ref = _TinyConv().double()
    ref.load_state_dict(model.state_dict())
    loss = sum(((ref(*sample(i)[:3]) - sample(i)[3]) ** 2).mean() for i in range(n_samples)) / n_samples
    loss.backward()
    err = max(float((g_ - p.grad).abs().max()) for g_, p in zip(got, ref.parameters()))
    q.put((rank, len(mine), calls["n"], n_red, err, [g_ is None for g_ in got[-2:]]))
    dist.destroy_process_group()


def test_batch_of_256_over_8_ranks_accumulates_32_each_and_allreduces_once_gloo_ws8():
    """GKN Darcy 241^2, batch = 256 samples sharded over 8 GPUs (BASELINE config 5; UAI1_full_resolution.py:54 trains with
    batch_size 1, the sharded batch is the data-parallel form): every rank accumulates the gradients of its 32 samples, one
    flat all-reduce averages them - the result is the gradient of the mean loss over all 256 samples under rank 0's weights,
    a parameter no sample touched keeps grad = None on every rank (Adam with weight decay must not see a zero there)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 8
    procs = [ctx.Process(target=_accum_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    n_par = sum(p.numel() for p in _TinyConv().parameters()) + 4 * 4 + 4
    for rank, n_mine, n_calls, n_red, err, none_flags in res:
        assert n_mine == 32 and n_calls == 1 and n_red == n_par, (rank, n_mine, n_calls, n_red)
        assert err < 1e-12, (rank, err)
        assert none_flags == [True, True], (rank, none_flags)
