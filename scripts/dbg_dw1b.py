import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graph_pde_amd import _lib, ops
from oracle.nnconv_oracle import nnconv_grads, rel_l2
d = torch.device("cuda:0")
dims, n, deg = [6, 1024, 1024, 4096], 48, 200
torch.manual_seed(8)
e = n * deg
dst = torch.randint(0, n, (e,)); dst[: e // 8] = 7; dst[e // 8: e // 8 + 300] = 11
ei = torch.stack([torch.randint(0, n, (e,)), dst])
ea, x = torch.randn(e, 6), torch.randn(n, 64)
mlp = torch.nn.Sequential(torch.nn.Linear(6, 1024), torch.nn.ReLU(), torch.nn.Linear(1024, 1024), torch.nn.ReLU(), torch.nn.Linear(1024, 4096))
W = [l.weight.detach() for l in mlp if isinstance(l, torch.nn.Linear)]; B = [l.bias.detach() for l in mlp if isinstance(l, torch.nn.Linear)]
root = torch.empty(64, 64).uniform_(-0.125, 0.125); g = torch.randn(n, 64)
ref = nnconv_grads(x, ei, ea, W, B, root, None, "mean", g, chunk_edges=2048)
csr = ops.build_csr(ei.to(d), n)
args = (x.to(d), csr, ea.to(d), [w.to(d) for w in W], [b.to(d) for b in B], root.to(d), "mean", g.to(d))
for env in ("", "GPDE_BWD_DW1_PASS", "GPDE_BWD_H1_IMAGE", "GPDE_BWD_GEMM_F32"):
    if env: os.environ[env] = "1"
    _lib.reload_switches()
    r = ops.nnconv_backward_raw(*args); torch.cuda.synchronize()
    if env: del os.environ[env]
    print(f"{env or 'default':20s} vs float64: dW1 {rel_l2(r[1][0].cpu(), ref[1][0]):.2e} db1 {rel_l2(r[2][0].cpu(), ref[2][0]):.2e} dW2 {rel_l2(r[1][1].cpu(), ref[1][1]):.2e} db2 {rel_l2(r[2][1].cpu(), ref[2][1]):.2e}")
_lib.reload_switches()
