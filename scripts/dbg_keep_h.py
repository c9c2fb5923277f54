import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graph_pde_amd as gp
from graph_pde_amd import ops, synth, hidden_cache
hidden_cache.MODE = "off"
d = torch.device("cuda:0")
ei, ea, n = synth.darcy_graph(61, 0.10, device=d)
for kw in (256, 512, 1024):
    torch.manual_seed(0)
    dims = [6, kw, kw, 4096]
    mlp = torch.nn.Sequential(*sum([[torch.nn.Linear(dims[i], dims[i + 1]), torch.nn.ReLU()] for i in range(3)], [])[:-1])
    conv = gp.NNConv_old(64, 64, mlp, aggr="mean").to(d)
    x, g = torch.randn(n, 64, device=d), torch.randn(n, 64, device=d)
    def step(gb):
        ops.SAVE_H_BYTES = gb << 30
        conv.zero_grad(set_to_none=True)
        xin = x.clone().requires_grad_(True)
        out = conv(xin, ei, ea); (out * g).sum().backward(); torch.cuda.synchronize()
        r = {"out": out.detach(), "dx": xin.grad}; r.update({k: p.grad.clone() for k, p in conv.named_parameters()}); return r
    a1, a2, b1, b2 = step(32), step(32), step(0), step(0)
    for k in a1:
        rel = float((a1[k].double() - b1[k].double()).norm() / b1[k].double().norm())
        print(kw, k, "kept run-to-run", torch.equal(a1[k], a2[k]), "rec run-to-run", torch.equal(b1[k], b2[k]), "kept vs rec", torch.equal(a1[k], b1[k]), f"{rel:.1e}")
