"""Phase timing of the split-f16 GEMM kernel inside one NNConv backward (needs a build with -DGPDE_NT_TIMING):
    GPDE_BUILD_SUFFIX=_TN python graph-pde_amd/build.py -DGPDE_NT_TIMING
    GPDE_LIB=$PWD/scripts/ubench/lib/libgpde_TN.so GPDE_HIDDEN_CACHE=off python scripts/nt_timing.py g121
clock64 ticks per wave-tile round (64 rows x 128 columns x K) in the tile prologue (scales, A chunks 0..2, first conversions),
the K loop, the drain of the tail loads and the epilogue (un-scale, mask, stores) - for the plain row tiles (dU_1 = dU_2 . W_2,
K = k2), the split-K form (dW_2 = dU_2^T . H_1, K = edges / 8) and the gather form (depth-deferred dU_2)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graph_pde_amd as gp
from graph_pde_amd import _lib, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "g121"
kw = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
s, r = {"g121": (121, 0.1), "g61": (61, 0.1), "g241": (241, 0.1)}[cfg]
dev = torch.device("cuda:0")
torch.manual_seed(0)
mlp = torch.nn.Sequential(torch.nn.Linear(6, kw), torch.nn.ReLU(), torch.nn.Linear(kw, kw), torch.nn.ReLU(), torch.nn.Linear(kw, 4096))
conv = gp.NNConv_old(64, 64, mlp, aggr="mean").to(dev)
ei, ea, n = synth.darcy_graph(s, r, device=dev)
x = torch.randn(n, 64, device=dev, requires_grad=True)
lib = _lib.lib()
fn = lib.gpde_debug_nt_timing
fn.restype, fn.argtypes = ctypes.c_int, [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()
for it in range(3):
    out = conv(x, ei, ea)
    torch.cuda.synchronize()
    fn(buf, 1)
    out.sum().backward()
    torch.cuda.synchronize()
    fn(buf, 1)
v = [int(b) for b in buf]
for name, o in (("plain row tiles (dU_1)", 0), ("split-K (dW_2)", 5), ("gather", 10)):
    pro, loop, drain, epi, tiles = v[o:o + 5]
    if not tiles:
        continue
    tot = pro + loop + drain + epi
    print(f"{cfg} {name}: wave-tile rounds {tiles}  ticks per round: prologue {pro/tiles:.0f}  K loop {loop/tiles:.0f}  drain {drain/tiles:.0f}  "
          f"epilogue {epi/tiles:.0f}  total {tot/tiles:.0f}   shares: {pro/tot:.3f} / {loop/tot:.3f} / {drain/tot:.3f} / {epi/tot:.3f}")
