"""Time forward + backward of one NNConv call (developer probe)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graph_pde_amd as gp
from graph_pde_amd import ops, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "g121"
kw = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
s, r = {"g121": (121, 0.1), "g61": (61, 0.1), "g241": (241, 0.1), "g16": (16, 0.15)}[cfg]
dev = torch.device("cuda:0")
torch.manual_seed(0)
mlp = torch.nn.Sequential(torch.nn.Linear(6, kw), torch.nn.ReLU(), torch.nn.Linear(kw, kw), torch.nn.ReLU(), torch.nn.Linear(kw, 4096))
conv = gp.NNConv_old(64, 64, mlp, aggr="mean").to(dev)
ei, ea, n = synth.darcy_graph(s, r, device=dev)
x = torch.randn(n, 64, device=dev, requires_grad=True)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = conv(x, ei, ea)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    out.sum().backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{cfg} kw={kw} E={ei.shape[1]}: fwd {1e3*(t1-t0):.1f} ms, bwd {1e3*(t2-t1):.1f} ms, bwd M-edges/s {ei.shape[1]/(t2-t1)/1e6:.1f}")
