"""MGKN-general (BASELINE config 4) forward with the per-edge weight cache on and the glue fused: 20 forwards after the
caches are warm - for `rocprofv3 --kernel-trace --stats` (where the remaining time of the V-cycle goes, per kernel)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graph_pde_amd import hidden_cache, mgkn_workloads
which = sys.argv[1] if len(sys.argv) > 1 else "mgkn_general_darcy2d"
dev = torch.device("cuda:0")
hidden_cache.WE_MODE = "auto"
kw = {"grouped": True} if (which == "mgkn_orthogonal_burgers1d" and len(sys.argv) > 2) else {"fused_glue": True}
wl = mgkn_workloads.WORKLOADS[which](dev, **kw)
for _ in range(4):
    wl.forward()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    wl.forward()
torch.cuda.synchronize()
print(f"{which} {kw}: {1e3 * (time.perf_counter() - t0) / n:.3f} ms per forward ({wl.calls} NNConv calls)")
