import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graph_pde_amd as gp
from graph_pde_amd import autograd as gpa, hidden_cache, ops, synth
DIMS = [6, 128, 128, 4096]; DEPTH = 4
torch.manual_seed(0)
mlp = torch.nn.Sequential(*sum([[torch.nn.Linear(DIMS[i], DIMS[i + 1]), torch.nn.ReLU()] for i in range(3)], [])[:-1])
conv = gp.NNConv_old(64, 64, mlp, aggr="mean").to("cuda:0")
hidden_cache.MODE = "on"; hidden_cache.WE_MODE = "auto"
d = torch.device("cuda:0")
ei, ea, n = synth.darcy_graph(24, 0.06, device=d)
x = torch.randn(n, 64, device=d); g = torch.randn(n, 64, device=d)
print("E", ei.shape[1], "N", n)
def step(flag):
    gpa.ACCUMULATE_GRAD_HIDDEN = flag
    hidden_cache.clear(); conv.zero_grad(set_to_none=True)
    xin = x.clone().requires_grad_(True)
    h = xin
    for _ in range(DEPTH): h = torch.relu(conv(h, ei, ea))
    (h * g).sum().backward(); torch.cuda.synchronize()
    return [xin.grad.clone()] + [p.grad.clone() for p in conv.parameters()]
names = ["x"] + [n_ for n_, _ in conv.named_parameters()]
a = step(False); b = step(False); c = step(True); e = step(True)
for nm, u, v, w, z in zip(names, a, b, c, e):
    print(f"{nm:14s} F/F equal {torch.equal(u, v)}  F/T equal {torch.equal(u, w)}  T/T equal {torch.equal(w, z)}  rel F/T {float((u - w).norm() / u.norm()):.2e}")
rec = {}
orig = ops.edge_weights_backward_raw
orig_b = ops.nnconv_backward_edgeweights_raw
def spy(grad_we, *a_, **k_):
    rec.setdefault("sum", []).append(grad_we.clone())
    return orig(grad_we, *a_, **k_)
def spy_b(*a_, **k_):
    r = orig_b(*a_, **k_)
    if k_.get("acc") is None:
        rec.setdefault("parts", []).append(r[1].clone())
    else:
        rec.setdefault("parts", []).append(None)
    return r
ops.edge_weights_backward_raw = spy
gpa.ops.nnconv_backward_edgeweights_raw = spy_b
step(False); sF, pF = rec["sum"][-1], rec["parts"][-4:]
step(True); sT = rec["sum"][-1]
print("sum equal", torch.equal(sF, sT), float((sF - sT).norm() / sF.norm()))
m = ((pF[0] + pF[1]) + pF[2]) + pF[3]
print("manual ((g4+g3)+g2)+g1 == autograd", torch.equal(m, sF), " == in-kernel", torch.equal(m, sT))
m2 = pF[0] + (pF[1] + (pF[2] + pF[3]))
print("manual g4+(g3+(g2+g1)) == autograd", torch.equal(m2, sF))
m3 = ((pF[3] + pF[2]) + pF[1]) + pF[0]
print("manual ((g1+g2)+g3)+g4 == autograd", torch.equal(m3, sF), " == in-kernel", torch.equal(m3, sT))
