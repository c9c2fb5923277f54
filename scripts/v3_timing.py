"""Phase timing of the v3 forward kernel (needs libgpde_T.so built with -DGPDE_V3_TIMING):
GPDE_BUILD_SUFFIX=_T python graph-pde_amd/build.py -DGPDE_V3_TIMING ; GPDE_LIB=.../libgpde_T.so python scripts/v3_timing.py g121"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graph_pde_amd as gp
from graph_pde_amd import ops, synth, _lib
cfg = sys.argv[1] if len(sys.argv) > 1 else "g121"
s, r = {"g121": (121, 0.1), "g61": (61, 0.1), "g241": (241, 0.1)}[cfg]
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
lib = _lib.lib()
buf = (ctypes.c_ulonglong * 6)()
for _ in range(2):
    ops.nnconv_forward_raw(x, csr, ea, pm, conv.root, conv.bias, "mean", out=out, ws=ws, precision="f16split")
torch.cuda.synchronize()
lib.gpde_debug_v3_timing(buf, 1)
ops.nnconv_forward_raw(x, csr, ea, pm, conv.root, conv.bias, "mean", out=out, ws=ws, precision="f16split")
torch.cuda.synchronize()
lib.gpde_debug_v3_timing(buf, 1)
pro, loop, post, tiles, wait, bar = [int(v) for v in buf]
tot = pro + loop + post
print(f"{cfg}: wave-tiles {tiles}  clock64 ticks per wave-tile: prologue {pro/tiles:.0f}  K-loop {loop/tiles:.0f}  post+GEMM2 {post/tiles:.0f}  total {tot/tiles:.0f}")
print(f"inside the K loop, per wave-tile: vmcnt wait {wait/tiles:.0f}  s_barrier {bar/tiles:.0f}")
print(f"shares: prologue {pro/tot:.3f}  K-loop {loop/tot:.3f}  post {post/tot:.3f}")
