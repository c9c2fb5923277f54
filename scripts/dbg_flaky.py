"""Developer probe: repeat one forward many times, with and without poisoning freed memory, and count results that
differ from the first (uninitialised workspace reads show up under poisoning, races show up at random)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graph_pde_amd as gp
from graph_pde_amd import ops, synth, mgkn_workloads, hidden_cache
from tests.test_host_logic import DenseNet
d = torch.device("cuda:0")
hidden_cache.MODE = "off"

def poison(val):
    t = torch.full((256 << 20,), val, device=d)      # 1 GiB of garbage, returned to the caching allocator
    del t

def stress(tag, fn, reps=60):
    y0 = fn().clone()
    bad = {}
    for mode in ("plain", "nan", "big", "neg0"):
        nb = 0
        worst = 0.0
        for i in range(reps):
            if mode == "nan": poison(float("nan"))
            if mode == "big": poison(3.0e38)
            if mode == "neg0": poison(-1.0)
            y = fn()
            if not torch.equal(y, y0):
                nb += 1
                worst = max(worst, float((y - y0).norm() / y0.norm()))
        bad[mode] = (nb, worst)
    print(tag, bad, flush=True)

torch.manual_seed(9)
s, r = 24, 0.15
ei = synth.lattice_radius_graph(s, r, d)
pos = synth.lattice_positions(s, d)
a = synth.darcy_coefficient(s, 3).to(d)
ea = synth.darcy_edge_attr(ei, pos, a)
n = s * s
na = gp.NodeAttr.darcy(pos, a)
x = torch.randn(n, 64, device=d)
for dims in ([6, 64, 128, 4096], [6, 128, 256, 4096]):
    conv = gp.NNConv_old(64, 64, DenseNet(dims, torch.nn.ReLU), aggr="mean").to(d)
    lin = ops.mlp_linears(conv.nn)
    pm = ops.pack_mlp([l.weight for l in lin], [l.bias for l in lin])
    csr = ops.csr_for(ei, n)
    for prec in ("f32", "f16split", "f16split_agg16"):
        stress(f"s24 {dims} {prec} tensor", lambda: ops.nnconv_forward_raw(x, csr, ea, pm, conv.root, conv.bias, "mean", precision=prec))
        if prec != "f32":
            stress(f"s24 {dims} {prec} nodeattr", lambda: ops.nnconv_forward_nodeattr_raw(x, csr, na, pm, conv.root, conv.bias, "mean", precision=prec))
wl = mgkn_workloads.general_darcy(d)
for i in (0, 1, 5):
    conv, xx, ei2, ea2 = wl.pairs[i]
    with torch.no_grad():
        stress(f"mgkn pair {i} E={ei2.shape[1]} direct", lambda: conv(xx, ei2, ea2))
hidden_cache.MODE = "auto"
conv, xx, ei2, ea2 = wl.pairs[0]
with torch.no_grad():
    conv(xx, ei2, ea2); conv(xx, ei2, ea2)
    stress("mgkn pair 0 cached-H", lambda: conv(xx, ei2, ea2))
