import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graph_pde_amd import ops
from oracle.nnconv_oracle import nnconv_forward, rel_l2
d = torch.device("cuda:0")
def run(dims, n=500, e=9000, pos=False, seed=0, scale=1.0):
    torch.manual_seed(seed)
    ei = torch.stack([torch.randint(0, n, (e,)), torch.randint(0, n, (e,))])
    ea = torch.randn(e, dims[0]) * scale
    if pos: ea = ea.abs()
    x = torch.randn(n, 64)
    mlp = torch.nn.Sequential(*sum([[torch.nn.Linear(dims[i], dims[i + 1]), torch.nn.ReLU()] for i in range(len(dims) - 1)], [])[:-1])
    ws_ = [l.weight.detach() for l in mlp if isinstance(l, torch.nn.Linear)]
    bs_ = [l.bias.detach() for l in mlp if isinstance(l, torch.nn.Linear)]
    y64 = nnconv_forward(x, ei, ea, ws_, bs_, None, None, aggr="mean", dtype=torch.float64)
    csr = ops.build_csr(ei.to(d), n)
    pm = ops.pack_mlp([w.to(d) for w in ws_], [b.to(d) for b in bs_])
    out = {}
    for p in ("f32", "f16split4w", "f16split"):
        y = ops.nnconv_forward_raw(x.to(d), csr, ea.to(d), pm, None, None, "mean", precision=p).cpu()
        out[p] = rel_l2(y, y64)
    print(dims, "pos" if pos else "randn", scale, {k: f"{v:.2e}" for k, v in out.items()})
run([6, 100, 200, 4096]); run([6, 100, 200, 4096], pos=True); run([6, 128, 128, 4096]); run([6, 256, 256, 4096]); run([6, 512, 512, 4096]); run([6,1024,1024,4096], e=3000)
run([6, 100, 200, 4096], scale=0.01); run([6, 100, 200, 4096], scale=100.0)
print("---- graph/attr combinations, dims [6,256,256,4096]")
from graph_pde_amd import synth
def run2(graph, attr_kind, dims=[6,256,256,4096]):
    torch.manual_seed(1)
    if graph == "lattice":
        ei, ea0, n = synth.darcy_graph(24, 0.12)
    else:
        n = 576; e = 20000
        ei = torch.stack([torch.randint(0, n, (e,)), torch.randint(0, n, (e,))]); ea0 = None
    e = ei.shape[1]
    ea = torch.randn(e, 6) if attr_kind == "randn" else (ea0 if ea0 is not None else synth.darcy_edge_attr(ei, synth.lattice_positions(24), synth.darcy_coefficient(24)))
    x = torch.randn(n, 64)
    mlp = torch.nn.Sequential(*sum([[torch.nn.Linear(dims[i], dims[i + 1]), torch.nn.ReLU()] for i in range(len(dims) - 1)], [])[:-1])
    ws_ = [l.weight.detach() for l in mlp if isinstance(l, torch.nn.Linear)]
    bs_ = [l.bias.detach() for l in mlp if isinstance(l, torch.nn.Linear)]
    y64 = nnconv_forward(x, ei, ea, ws_, bs_, None, None, aggr="mean", dtype=torch.float64)
    csr = ops.build_csr(ei.to(d), n)
    pm = ops.pack_mlp([w.to(d) for w in ws_], [b.to(d) for b in bs_])
    res = {}
    for p in ("f16split4w", "f16split"):
        y = ops.nnconv_forward_raw(x.to(d), csr, ea.to(d), pm, None, None, "mean", precision=p).cpu()
        res[p] = rel_l2(y, y64)
        if p == "f16split":
            err = (y.double() - y64).abs()
            res["max_abs"] = float(err.max()); res["frac_rows_bad"] = float((err.max(1).values > 1e-4).float().mean())
    print(graph, attr_kind, {k: (f"{v:.2e}") for k, v in res.items()})
for gk in ("lattice", "random"):
    for ak in ("randn", "darcy"):
        run2(gk, ak)
