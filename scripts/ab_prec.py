"""A/B of the forward kernel variants: parity of each precision flag against the exact fp32 path,
then timing (developer probe).  usage: ab_prec.py <cfg> <prec> [<prec> ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graph_pde_amd as gp
from graph_pde_amd import ops, synth
cfg = sys.argv[1]
precs = sys.argv[2:]
s, r = {"g121": (121, 0.1), "g61": (61, 0.1), "g241": (241, 0.1), "g16": (16, 0.15), "g31": (31, 0.12)}[cfg]
dev = torch.device("cuda:0")
torch.manual_seed(0)
mlp = torch.nn.Sequential(torch.nn.Linear(6, 1024), torch.nn.ReLU(), torch.nn.Linear(1024, 1024), torch.nn.ReLU(), torch.nn.Linear(1024, 4096))
conv = gp.NNConv_old(64, 64, mlp, aggr="mean").to(dev)
ei, ea, n = synth.darcy_graph(s, r, device=dev)
x = torch.randn(n, 64, device=dev)
csr = ops.csr_for(ei, n)
lin = ops.mlp_linears(conv.nn)
pm = ops.pack_mlp([l.weight for l in lin], [l.bias for l in lin])
ws = torch.empty(ops.workspace_bytes(n, csr.n_edges, pm), dtype=torch.uint8, device=dev)
ref = torch.empty(n, 64, device=dev)
ops.nnconv_forward_raw(x, csr, ea, pm, conv.root, conv.bias, "mean", out=ref, ws=ws, precision="f32")
torch.cuda.synchronize()
for prec in precs:
    out = torch.empty(n, 64, device=dev)
    for _ in range(2):
        ops.nnconv_forward_raw(x, csr, ea, pm, conv.root, conv.bias, "mean", out=out, ws=ws, precision=prec)
    torch.cuda.synchronize()
    d = ((out - ref).double().norm() / ref.double().norm()).item()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 3
    a.record()
    for _ in range(K):
        ops.nnconv_forward_raw(x, csr, ea, pm, conv.root, conv.bias, "mean", out=out, ws=ws, precision=prec)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / K
    print(f"{cfg} {prec}: rel-L2 vs f32 {d:.2e}  {ms:.2f} ms  {csr.n_edges/ms/1e3:.1f} M-edges/s", flush=True)
