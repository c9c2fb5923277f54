"""Headline forward with smaller destination-node chunks: does keeping the Z round trip (fused kernel -> gemm3: 256 KiB per node)
inside the 256 MB Infinity Cache buy time / energy on the power-governed kernel?  usage: time_fwd_chunks.py [g241]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graph_pde_amd as gp
from graph_pde_amd import ops, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "g241"
s, r = {"g241": (241, 0.1), "g121": (121, 0.1)}[cfg]
dev = torch.device("cuda:0")
torch.manual_seed(0)
mlp = torch.nn.Sequential(torch.nn.Linear(6, 1024), torch.nn.ReLU(), torch.nn.Linear(1024, 1024), torch.nn.ReLU(), torch.nn.Linear(1024, 4096))
conv = gp.NNConv_old(64, 64, mlp, aggr="mean").to(dev)
ei, ea, n = synth.darcy_graph(s, r, device=dev)
e = ei.shape[1]
csr = ops.csr_for(ei, n)
lin = ops.mlp_linears(conv.nn)
pm = ops.pack_mlp([l.weight for l in lin], [l.bias for l in lin])
x = torch.randn(n, 64, device=dev)
out = torch.empty(n, 64, device=dev)
full = ops.workspace_bytes(n, e, pm)
ref = None
for frac in (1.0, 0.5, 0.25, 0.125, 0.0625, 0.04):
    ws = torch.empty(int(full * frac), dtype=torch.uint8, device=dev)
    try:
        plan = ops.launch_plan(n, e, pm, ws.numel())
    except Exception as ex:
        print("workspace", ws.numel() >> 20, "MiB:", ex); continue
    for _ in range(2):
        ops.nnconv_forward_raw(x, csr, ea, pm, conv.root, conv.bias, "mean", out=out, ws=ws)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4):
        ops.nnconv_forward_raw(x, csr, ea, pm, conv.root, conv.bias, "mean", out=out, ws=ws)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 4
    if ref is None: ref = out.clone()
    print(f"{cfg} workspace {ws.numel() / 2**20:7.0f} MiB: {plan['n_chunks']} chunks of {plan['nodes_per_chunk']} nodes: {1e3 * t:.1f} ms = {e / t / 1e6:.1f} M-edges/s, "
          f"bitwise equal to the first: {bool(torch.equal(out, ref))}", flush=True)
