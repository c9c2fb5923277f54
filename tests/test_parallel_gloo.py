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
    q.put((rank, sizes, out.detach(), {k: p.grad.clone() for k, p in model.named_parameters()}))
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
        assert torch.allclose(out, ref.detach(), rtol=1e-12, atol=1e-12), rank   # every rank holds the full result
        for k, p in model.named_parameters():
            assert torch.allclose(grads[k], p.grad, rtol=1e-10, atol=1e-12), (rank, k)


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
