"""Developer probe (end of round 6): the hypothesis case n = 2, 6,309 edges into node 0, kernel MLP [5, 267, 18, 4096], mean, no
root / bias, built several times in one process.  NOTE what it showed: the weights are set BEFORE NNConv_old(...) here, and the
constructor resets `nn` (nn_conv.py:258) with the global generator - every repetition has OTHER weights than the `ref` of repetition
0 (hence err ~ 1 from repetition 1 on).  That, not the kernels, was the "fails once, passes on replay" of tests/test_gpu_hypothesis.py,
which had the same order; the forward kernels use no floating-point atomics."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graph_pde_amd as gp
from graph_pde_amd import ops, hidden_cache
from oracle.nnconv_oracle import nnconv_forward, rel_l2
c = {'n': 2, 'e': 6309, 'k0': 5, 'widths': [267, 18], 'aggr': 'mean', 'seed': 1}
d = torch.device("cuda:0")
outs = []
for rep in range(4):
    if os.environ.get('DBG_CLEAR'):
        ops.clear_caches(); hidden_cache.clear()
    g = torch.Generator().manual_seed(c["seed"])
    n, e = c["n"], c["e"]
    src = torch.randint(0, n, (e,), generator=g); dst = torch.randint(0, 1, (e,), generator=g)
    dims = [c["k0"]] + c["widths"] + [4096]
    mlp = torch.nn.Sequential(*sum([[torch.nn.Linear(dims[i], dims[i + 1]), torch.nn.ReLU()] for i in range(len(dims) - 1)], [])[:-1])
    with torch.no_grad():
        for p_ in mlp.parameters():
            p_.copy_(torch.empty_like(p_).uniform_(-1, 1, generator=g) / (p_.shape[-1] ** 0.5))
    conv = gp.NNConv_old(64, 64, mlp, aggr=c["aggr"], root_weight=False, bias=False)
    ea = torch.randn(e, c["k0"], generator=g)
    x, gout = torch.randn(n, 64, generator=g), torch.randn(n, 64, generator=g)
    lin = ops.mlp_linears(conv.nn)
    W, B = [l.weight.detach().clone() for l in lin], [l.bias.detach().clone() for l in lin]
    ei = torch.stack([src, dst])
    if rep == 0:
        ref = nnconv_forward(x, ei, ea, W, B, None, None, aggr=c["aggr"], dtype=torch.float64)
        e32 = rel_l2(nnconv_forward(x, ei, ea, W, B, None, None, aggr=c["aggr"], dtype=torch.float32), ref)
    conv = conv.to(d)
    big = torch.zeros(2, 2 * e, dtype=torch.int64, device=d); big[:, ::2] = ei.to(d); ei_d = big[:, ::2]
    xin = x.to(d).requires_grad_(True)
    s0 = dict(hidden_cache.stats)
    if os.environ.get('DBG_NOGRAD'):
        with torch.no_grad():
            out = conv(xin.detach(), ei_d, ea.to(d))
    else:
        out = conv(xin, ei_d, ea.to(d))
    torch.cuda.synchronize()
    o = out.detach().cpu(); outs.append(o)
    print(rep, f"err {rel_l2(o, ref):.4e} (fp32 oracle {e32:.2e})", "equal to first", torch.equal(o, outs[0]), type(out.grad_fn).__name__, float(o.abs().max()),
          {k: v - s0.get(k, 0) for k, v in hidden_cache.stats.items() if v != s0.get(k, 0)}, flush=True)
