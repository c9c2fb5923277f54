"""Developer probe: the MGKN-general test flow in a loop; cached-H / fused results against the exact-fp32 direct path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graph_pde_amd import ops, mgkn_workloads, hidden_cache
d = torch.device("cuda:0")
def poison(val):
    t = torch.full((256 << 20,), val, device=d); del t
nbad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 16):
    for name in ("mgkn_general_darcy2d", "mgkn_orthogonal_burgers1d"):
        hidden_cache.MODE = "auto"; hidden_cache.clear()
        ops.DEFAULT_PRECISION = "f16split"
        wl = mgkn_workloads.WORKLOADS[name](d, seed=it)
        if it % 2: poison(float("nan"))
        wl.forward()
        ys = []
        with torch.no_grad():
            for conv, x, ei, ea in wl.pairs:
                if it % 3 == 1: poison(-3.0e38)
                ys.append(conv(x, ei, ea))
            hidden_cache.MODE = "off"
            yd = [conv(x, ei, ea) for conv, x, ei, ea in wl.pairs]
            ops.DEFAULT_PRECISION = "f32"
            yr = [conv(x, ei, ea) for conv, x, ei, ea in wl.pairs]
        for i, (a, b, r) in enumerate(zip(ys, yd, yr)):
            ea_, eb_ = float((a - r).norm() / r.norm()), float((b - r).norm() / r.norm())
            if not (ea_ < 3e-6 and eb_ < 3e-6):
                nbad += 1
                print(it, name, "pair", i, tuple(wl.pairs[i][2].shape), "cached %.2e direct %.2e" % (ea_, eb_), flush=True)
print("bad:", nbad)
