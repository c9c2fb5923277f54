"""Depth-6 training steps on the shared-H path (developer probe for rocprofv3): usage shared_h_step.py <g61|g121> [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graph_pde_amd as gp
from graph_pde_amd import ops, synth, hidden_cache
cfg = sys.argv[1] if len(sys.argv) > 1 else "g61"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
s = {"g61": 61, "g121": 121}[cfg]
dev = torch.device("cuda:0")
ei, ea, n = synth.darcy_graph(s, 0.1, device=dev)
torch.manual_seed(0)
mlp = torch.nn.Sequential(torch.nn.Linear(6, 1024), torch.nn.ReLU(), torch.nn.Linear(1024, 1024), torch.nn.ReLU(), torch.nn.Linear(1024, 4096))
conv = gp.NNConv_old(64, 64, mlp, aggr="mean").to(dev)
fc1, fc2 = torch.nn.Linear(6, 64).to(dev), torch.nn.Linear(64, 1).to(dev)
params = list(conv.parameters()) + list(fc1.parameters()) + list(fc2.parameters())
opt = torch.optim.Adam(params, lr=1e-4, weight_decay=5e-4)
a_in, y_t = torch.randn(n, 6, device=dev), torch.randn(n, device=dev)
ts = []
for it in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    h = fc1(a_in)
    for _ in range(6):
        h = torch.relu(conv(h, ei, ea))
    loss = torch.norm(fc2(h).view(-1) - y_t, 1)
    loss.backward()
    opt.step()
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(f"{cfg}: step {1e3 * sorted(ts[2:])[len(ts[2:]) // 2]:.2f} ms, {dict(hidden_cache.stats)}")
