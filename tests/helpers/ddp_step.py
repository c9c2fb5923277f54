"""One data-parallel training step of the GKN stack on the native operator, run as a worker under
`python -m torch.distributed.run` (tests/test_gpu_ddp.py) or in-process (world 1, no process group).

Every rank: builds the model from ITS OWN seed, runs a warm-up forward (so the packed-weight cache holds rank-specific
weights), then `parallel.broadcast_parameters` (rank 0's weights everywhere), one forward + backward on sample `rank`,
`parallel.allreduce_gradients`.  Rank 0 saves {parameter name: gradient} and the loss values."""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def build(seed, dev):
    import graph_pde_amd as gp
    torch.manual_seed(seed)
    mlp = torch.nn.Sequential(torch.nn.Linear(6, 128), torch.nn.ReLU(), torch.nn.Linear(128, 256), torch.nn.ReLU(),
                              torch.nn.Linear(256, 4096))
    conv = gp.NNConv_old(64, 64, mlp, aggr="mean")
    return torch.nn.ModuleDict({"fc1": torch.nn.Linear(6, 64), "conv": conv, "fc2": torch.nn.Linear(64, 1)}).to(dev)


def forward(model, a_in, ei, ea, depth=3):
    h = model["fc1"](a_in)
    for _ in range(depth):
        h = torch.relu(model["conv"](h, ei, ea))
    return model["fc2"](h).view(-1)


def sample(k, dev):
    from graph_pde_amd import synth
    ei, ea, n = synth.darcy_graph(14, 0.2, device=dev, seed=10 + k)
    g = torch.Generator().manual_seed(100 + k)
    return ei, ea, torch.randn(n, 6, generator=g).to(dev), torch.randn(n, generator=g).to(dev)


def step(rank, world, dev, backend):
    from graph_pde_amd import parallel
    model = build(1000 + rank, dev)                       # deliberately different weights per rank
    ei, ea, a_in, y = sample(rank, dev)
    with torch.no_grad():
        forward(model, a_in, ei, ea)                      # packs the rank's own (pre-broadcast) weights
    if world > 1 or backend:
        parallel.broadcast_parameters(model, src=0)
    else:
        model.load_state_dict(build(1000, dev).state_dict())
    loss = torch.norm(forward(model, a_in, ei, ea) - y, 1)
    loss.backward()
    if world > 1 or backend:
        parallel.allreduce_gradients(model.parameters(), world=world, average=True)
    return model, float(loss)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    import torch.distributed as dist
    from graph_pde_amd import parallel
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(args.backend)
    model, loss = step(rank, world, dev, args.backend)
    losses = [None] * world
    dist.all_gather_object(losses, loss)
    if rank == 0:
        torch.save({"grads": {k: p.grad.detach().cpu() for k, p in model.named_parameters()}, "losses": losses,
                    "world": world, "backend": dist.get_backend()}, args.out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
