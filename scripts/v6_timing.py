"""Phase timing of the v6 forward kernel (needs a build with -DGPDE_V6_TIMING):
    GPDE_BUILD_SUFFIX=_T6 python graph-pde_amd/build.py -DGPDE_V6_TIMING
    GPDE_LIB=$PWD/scripts/ubench/lib/libgpde_T6.so python scripts/v6_timing.py g241
clock64 ticks per wave-tile (64 edges x 128 columns) spent in the tile prologue (attributes, H1 of chunk 0),
the K loop (32 chunks at k1 = 1024: 1664 MFMA-cycles each) and the un-scale + aggregation phase."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graph_pde_amd as gp
from graph_pde_amd import _lib, ops, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "g121"
kw = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
s, r = {"g121": (121, 0.1), "g61": (61, 0.1), "g241": (241, 0.1)}[cfg]
dev = torch.device("cuda:0")
torch.manual_seed(0)
mlp = torch.nn.Sequential(torch.nn.Linear(6, kw), torch.nn.ReLU(), torch.nn.Linear(kw, kw), torch.nn.ReLU(), torch.nn.Linear(kw, 4096))
conv = gp.NNConv_old(64, 64, mlp, aggr="mean").to(dev)
ei, ea, n = synth.darcy_graph(s, r, device=dev)
x = torch.randn(n, 64, device=dev)
csr = ops.csr_for(ei, n)
lin = ops.mlp_linears(conv.nn)
pm = ops.pack_mlp([l.weight for l in lin], [l.bias for l in lin])
ws = torch.empty(ops.workspace_bytes(n, csr.n_edges, pm), dtype=torch.uint8, device=dev)
out = torch.empty(n, 64, device=dev)
lib = _lib.lib()
fn = lib.gpde_debug_v6_timing
fn.restype, fn.argtypes = ctypes.c_int, [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 8)()
for _ in range(2):
    ops.nnconv_forward_raw(x, csr, ea, pm, conv.root, conv.bias, "mean", out=out, ws=ws, precision="f16split")
torch.cuda.synchronize()
fn(buf, 1)
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
ops.nnconv_forward_raw(x, csr, ea, pm, conv.root, conv.bias, "mean", out=out, ws=ws, precision="f16split")
t1.record()
torch.cuda.synchronize()
fn(buf, 1)
pro, loop, post, tiles, peel = [int(v) for v in buf][:5]
tot = pro + loop + post
nkc = (kw + 31) // 32
print(f"{cfg} k={kw}: {t0.elapsed_time(t1):.2f} ms  wave-tiles {tiles}  ticks per wave-tile: prologue {pro/tiles:.0f}  "
      f"K-loop {loop/tiles:.0f} ({loop/tiles/nkc:.0f} per chunk; MFMA floor 1664)  un-scale+aggregation {post/tiles:.0f}  total {tot/tiles:.0f}")
print(f"K loop split: peeled chunks 0-5 {peel/tiles/6:.0f} per chunk, steady chunks {(loop-peel)/tiles/(nkc-6):.0f} per chunk")
print(f"shares: prologue {pro/tot:.3f}  K-loop {loop/tot:.3f}  post {post/tot:.3f}")
