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
