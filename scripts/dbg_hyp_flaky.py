import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graph_pde_amd as gp
from graph_pde_amd import ops, hidden_cache
from oracle.nnconv_oracle import nnconv_forward, rel_l2
c = {'n': 2, 'e': 290, 'k0': 1, 'widths': [16], 'aggr': 'add', 'root': False, 'bias': False, 'n_dst': 1, 'seed': 0}
d = torch.device("cuda:0")
outs = []
for rep in range(12):
    g = torch.Generator().manual_seed(c["seed"])
    n, e = c["n"], c["e"]
    src = torch.randint(0, n, (e,), generator=g); dst = torch.randint(0, c["n_dst"], (e,), generator=g)
    dims = [c["k0"]] + c["widths"] + [4096]
    mlp = torch.nn.Sequential(*sum([[torch.nn.Linear(dims[i], dims[i + 1]), torch.nn.ReLU()] for i in range(len(dims) - 1)], [])[:-1])
    conv = gp.NNConv_old(64, 64, mlp, aggr=c["aggr"], root_weight=c["root"], bias=c["bias"])
    with torch.no_grad():
        for p_ in conv.parameters():
            p_.copy_(torch.empty_like(p_).uniform_(-0.125, 0.125, generator=g))
    ea = torch.randn(e, c["k0"], generator=g)
    x, gout = torch.randn(n, 64, generator=g), torch.randn(n, 64, generator=g)
    lin = ops.mlp_linears(conv.nn)
    W, B = [l.weight.detach().clone() for l in lin], [l.bias.detach().clone() for l in lin]
    ei = torch.stack([src, dst])
    ref = nnconv_forward(x, ei, ea, W, B, None, None, aggr=c["aggr"], dtype=torch.float64)
    conv = conv.to(d)
    big = torch.zeros(2, 2 * e, dtype=torch.int64, device=d); big[:, ::2] = ei.to(d); ei_d = big[:, ::2]
    xin = x.to(d).requires_grad_(True)
    out = conv(xin, ei_d, ea.to(d))
    (out * gout.to(d)).sum().backward()
    torch.cuda.synchronize()
    o = out.detach().cpu(); outs.append(o)
    print(rep, "err", rel_l2(o, ref), "equal to first", torch.equal(o, outs[0]), dict((k, v) for k, v in hidden_cache.stats.items() if v))
