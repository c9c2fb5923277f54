"""Where does a forward step spend its time? (developer probe, not part of the test suite)"""
import sys, time, ctypes, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graph_pde_amd as gp
from graph_pde_amd import ops, synth, _lib
cfg = sys.argv[1] if len(sys.argv) > 1 else "g121"
prec = sys.argv[2] if len(sys.argv) > 2 else "f16split"
s, r = {"g121": (121, 0.1), "g61": (61, 0.1), "g241": (241, 0.1), "g16": (16, 0.15)}[cfg]
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
out = torch.empty(n, 64, device=dev)
def step():
    ops.nnconv_forward_raw(x, csr, ea, pm, conv.root, conv.bias, "mean", out=out, ws=ws, precision=prec)
for _ in range(2): step()
torch.cuda.synchronize()
for k in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); a.record(); step(); b.record(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{cfg} {prec}: host call {1e3*(t1-t0):.2f} ms, gpu events {a.elapsed_time(b):.2f} ms, wall {1e3*(t2-t0):.2f} ms, E={csr.n_edges}")
