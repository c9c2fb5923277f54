"""One training step of the GKN stack on ONE graph split by destination rows over the ranks (parallel.partition_rows /
nnconv_rows: SURVEY.md §8e, way 2), run as a worker under `python -m torch.distributed.run` (tests/test_gpu_ddp.py).
Every rank builds the same model and the full graph, keeps its own in-edges, and runs fc1 -> 3 x relu(NNConv over its
rows + all-gather) -> fc2 -> L1 loss -> backward -> parallel.allreduce_gradients.  Rank 0 saves output, loss, gradients."""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from tests.helpers import ddp_step  # noqa: E402


def forward(model, a_in, layer, depth=3):
    h = model["fc1"](a_in)
    for _ in range(depth):
        h = torch.relu(layer(h))
    return model["fc2"](h).view(-1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo")
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
    model = ddp_step.build(1000, dev)
    ei, ea, a_in, y = ddp_step.sample(0, dev)
    part = parallel.partition_rows(ei, ea, a_in.size(0))
    with torch.no_grad():
        out_inf = forward(model, a_in, lambda h: parallel.nnconv_rows(model["conv"], h, part))
    out = forward(model, a_in, lambda h: parallel.nnconv_rows(model["conv"], h, part))
    loss = torch.norm(out - y, 1)
    loss.backward()
    parallel.allreduce_gradients(model.parameters(), world=world, average=True)
    parts = [None] * world
    dist.all_gather_object(parts, (part.lo, part.hi, part.n_edges))
    if rank == 0:
        torch.save({"grads": {k: p.grad.detach().cpu() for k, p in model.named_parameters()}, "loss": float(loss),
                    "out": out.detach().cpu(), "out_inference": out_inf.cpu(), "parts": parts, "world": world,
                    "backend": dist.get_backend()}, args.out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
