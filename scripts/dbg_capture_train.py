import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import graph_pde_amd as gp
from graph_pde_amd import hidden_cache, mgkn_workloads, ops
d = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "off"
hidden_cache.MODE = mode
hidden_cache.clear()
torch.manual_seed(0)
n, e = 2000, 6000
ei = torch.stack([torch.randint(0, n, (e,)), torch.randint(0, n, (e,))]).to(d)
ea = torch.randn(e, 6, device=d)
mlp = torch.nn.Sequential(torch.nn.Linear(6, 256), torch.nn.ReLU(), torch.nn.Linear(256, 256), torch.nn.ReLU(), torch.nn.Linear(256, 4096))
conv = gp.NNConv(64, 64, mlp, aggr="mean").to(d)
x0 = torch.randn(n, 64, device=d)
params = list(conv.parameters())
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2

def fb():
    for p in params:
        p.grad = None
    x = x0
    for _ in range(depth):
        x = F.relu(x + conv(x, ei, ea))
    loss = x.square().mean()
    loss.backward()
    return loss

def grads():
    torch.cuda.synchronize()
    return [p.grad.detach().clone() for p in params]
fb(); g_eager = grads()
fb(); g_eager2 = grads()
print("eager repeat equal:", [bool(torch.equal(a, b)) for a, b in zip(g_eager, g_eager2)], flush=True)
cap = gp.capture(fb, warmup=2)
for i in range(3):
    l = cap(); g = grads()
    print("replay", i, float(l), "grads == eager:", [bool(torch.equal(a, b)) for a, b in zip(g, g_eager)],
          "max rel diff", max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(g, g_eager)), flush=True)
