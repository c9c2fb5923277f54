"""Per NNConv application of the MGKN workloads: GPU time by kernel kind (developer probe)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graph_pde_amd import _lib, mgkn_workloads, ops
dev = torch.device("cuda:0")
for name, build in mgkn_workloads.WORKLOADS.items():
    wl = build(dev)
    wl.forward(); wl.forward()
    print(name)
    for conv, x, ei, ea in wl.pairs:
        lin = ops.mlp_linears(conv.nn)
        dims = [lin[0].weight.shape[1]] + [l.weight.shape[0] for l in lin]
        with torch.no_grad():
            conv(x, ei, ea)
            _lib.profile_begin()
            for _ in range(5):
                conv(x, ei, ea)
            k = _lib.profile_end()
        print(f"  N={x.shape[0]:6d} E={ei.shape[1]:7d} dims={dims}  " + "  ".join(f"{kk} {v[0] / 5 * 1e3:7.1f} us" for kk, v in k.items() if v[1]))
