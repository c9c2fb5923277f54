"""Phase timing of the staged per-edge backward kernel inside one NNConv backward (needs a build with -DGPDE_EB2_TIMING):
    GPDE_BUILD_SUFFIX=_TE python graph-pde_amd/build.py -DGPDE_EB2_TIMING
    GPDE_LIB=$PWD/scripts/ubench/lib/libgpde_TE.so GPDE_HIDDEN_CACHE=off python scripts/eb2_timing.py g121
clock64 ticks per wave and step (32 edges x 32 hidden columns: 64 fp32 MFMAs = 4096 matrix-pipe cycles) spent waiting at the
top of the step (DMA of the step's dZ / H tiles + the workgroup barrier) and in the step's products and stores."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graph_pde_amd as gp
from graph_pde_amd import _lib, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "g121"
kw = 1024
s, r = {"g121": (121, 0.1), "g61": (61, 0.1), "g241": (241, 0.1)}[cfg]
dev = torch.device("cuda:0")
torch.manual_seed(0)
mlp = torch.nn.Sequential(torch.nn.Linear(6, kw), torch.nn.ReLU(), torch.nn.Linear(kw, kw), torch.nn.ReLU(), torch.nn.Linear(kw, 4096))
conv = gp.NNConv_old(64, 64, mlp, aggr="mean").to(dev)
ei, ea, n = synth.darcy_graph(s, r, device=dev)
x = torch.randn(n, 64, device=dev, requires_grad=True)
lib = _lib.lib()
fn = lib.gpde_debug_eb2_timing
fn.restype, fn.argtypes = ctypes.c_int, [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 8)()
for it in range(3):
    out = conv(x, ei, ea)
    torch.cuda.synchronize()
    fn(buf, 1)
    out.sum().backward()
    torch.cuda.synchronize()
    fn(buf, 1)
pro, wait, work, epi, steps, waves = [int(b) for b in buf][:6]
tot = pro + wait + work + epi
print(f"{cfg}: waves {waves}  steps per wave {steps/waves:.1f}  ticks per step: wait {wait/steps:.0f}  work {work/steps:.0f}   per wave: prologue {pro/waves:.0f}  "
      f"epilogue {epi/waves:.0f}   shares: prologue {pro/tot:.3f}  wait {wait/tot:.3f}  work {work/tot:.3f}  epilogue {epi/tot:.3f}")
