import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graph_pde_amd as gp
from graph_pde_amd import hidden_cache, mgkn_workloads
name = "mgkn_orthogonal_burgers1d"
d = torch.device("cuda:0")
hidden_cache.clear()
wa = mgkn_workloads.WORKLOADS[name](d, capturable=True)
wb = mgkn_workloads.WORKLOADS[name](d, capturable=True)
for ma, mb in zip(wa.modules, wb.modules):
    mb.load_state_dict(ma.state_dict())
def step():
    loss = wa.train_step()
    return loss, loss.detach().clone()
cap = gp.capture(step, warmup=3, updates_parameters=True)
mode = sys.argv[1] if len(sys.argv) > 1 else "twin"
if mode == "twin":
    for _ in range(3):
        wb.train_step()
elif mode == "twin_nograd":
    for _ in range(3):
        wb.forward()
torch.cuda.synchronize()
for it in range(3):
    l, c = cap()
    torch.cuda.synchronize()
    print(f"[{mode}] replay {it}: loss tensor {float(l.detach()):.6f}  clone {float(c):.6f}")
