import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graph_pde_amd import mgkn_workloads, hidden_cache
d = torch.device("cuda:0")
for mode in ("off", "auto"):
    hidden_cache.MODE = mode
    hidden_cache.clear()
    wl = mgkn_workloads.general_darcy(d)
    for k, (conv, x, ei, ea) in enumerate(wl.pairs):
        with torch.no_grad():
            res = torch.randn_like(x)
            for rep in range(3):
                y1 = torch.relu(res + conv(x, ei, ea))
                y2 = conv(x, ei, ea, residual=res, activation="relu")
                print(mode, k, tuple(ei.shape), rep, bool(torch.equal(y1, y2)), float((y1 - y2).abs().max()))
