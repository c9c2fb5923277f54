"""debug: dW_1 / db_1 from the dU_1 GEMM's epilogue vs k_dw_first's pass (GPDE_BWD_DW1_PASS=1)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graph_pde_amd import _lib, ops, synth
d = torch.device("cuda:0")
s = int(sys.argv[1]) if len(sys.argv) > 1 else 61
ei, ea, n = synth.darcy_graph(s, 0.10, device=d)
torch.manual_seed(0)
dims = [6, 1024, 1024, 4096]
mlp = torch.nn.Sequential(torch.nn.Linear(6, 1024), torch.nn.ReLU(), torch.nn.Linear(1024, 1024), torch.nn.ReLU(), torch.nn.Linear(1024, 4096)).to(d)
lin = ops.mlp_linears(mlp)
W, B = [l.weight.detach() for l in lin], [l.bias.detach() for l in lin]
root = torch.randn(64, 64, device=d) / 8
x, g = torch.randn(n, 64, device=d), torch.randn(n, 64, device=d)
csr = ops.csr_for(ei, n)
def run():
    r = ops.nnconv_backward_raw(x, csr, ea, W, B, root, "mean", g)
    torch.cuda.synchronize()
    return r
a1 = run(); a2 = run()
os.environ["GPDE_BWD_DW1_PASS"] = "1"; _lib.reload_switches()
b = run()
del os.environ["GPDE_BWD_DW1_PASS"]; _lib.reload_switches()
def rel(u, v): return float((u - v).norm() / v.norm())
print("E", ei.shape[1], "run-to-run identical:", torch.equal(a1[1][0], a2[1][0]), torch.equal(a1[2][0], a2[2][0]))
print("dW1 epilogue vs pass:", rel(a1[1][0], b[1][0]), " db1:", rel(a1[2][0], b[2][0]), " dW2:", rel(a1[1][1], b[1][1]))
e = (a1[2][0] - b[2][0]).abs() / b[2][0].abs().max()
bad = torch.nonzero(e > 1e-4).flatten()
print("db1: columns off by > 1e-4 of the max:", bad.numel(), "of", e.numel(), "first", bad[:40].tolist())
ew = (a1[1][0] - b[1][0]).abs() / b[1][0].abs().max()
print("dW1 per-slot max err:", ew.max(dim=0).values.tolist())
print("ratio a/b on the worst columns:", (a1[2][0][bad[:8]] / b[2][0][bad[:8]]).tolist())
print("db1 (sum mode) first 8:", a1[2][0][:8].tolist(), " min/max:", float(a1[2][0].min()), float(a1[2][0].max()), " E =", ei.shape[1])
h1 = torch.relu(ea @ W[0].t() + B[0])
cnt = (h1 > 0).sum(dim=0).float()
print("active count per column (torch fp32) first 8:", cnt[:8].tolist(), " max |diff| vs db1:", float((cnt - a1[2][0]).abs().max()), " run-to-run max diff:", float((a1[2][0] - a2[2][0]).abs().max()))
