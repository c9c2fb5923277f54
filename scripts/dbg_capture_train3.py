import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graph_pde_amd as gp
from graph_pde_amd import hidden_cache, mgkn_workloads
name = sys.argv[1] if len(sys.argv) > 1 else "mgkn_orthogonal_burgers1d"
d = torch.device("cuda:0")
hidden_cache.clear()
wa = mgkn_workloads.WORKLOADS[name](d, capturable=True)
keep = []
def step():
    loss = wa.train_step()
    c = loss.detach().clone()
    return loss, c
cap = gp.capture(step, warmup=3, updates_parameters=True)
for it in range(4):
    l, c = cap()
    torch.cuda.synchronize()
    print(f"replay {it}: loss tensor {float(l.detach()):.6f}  clone made right after the step {float(c):.6f}")
