import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graph_pde_amd as gp
from graph_pde_amd import hidden_cache, mgkn_workloads
name = sys.argv[1] if len(sys.argv) > 1 else "mgkn_orthogonal_burgers1d"
d = torch.device("cuda:0")
hidden_cache.clear()
wa = mgkn_workloads.WORKLOADS[name](d, capturable=True)
wb = mgkn_workloads.WORKLOADS[name](d, capturable=True)
for ma, mb in zip(wa.modules, wb.modules):
    mb.load_state_dict(ma.state_dict())
cap = gp.capture(wa.train_step, warmup=3, updates_parameters=True)
for _ in range(3):
    wb.train_step()
torch.cuda.synchronize()
print("after warm-up equal:", all(torch.equal(pa, pb) for ma, mb in zip(wa.modules, wb.modules) for pa, pb in zip(ma.parameters(), mb.parameters())))
la = float(cap()); lb = float(wb.train_step()); torch.cuda.synchronize()
print("loss replay", la, "direct", lb)
for k, (ma, mb) in enumerate(zip(wa.modules, wb.modules)):
    for (n, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        e = float((pa - pb).abs().max() / pb.abs().max().clamp_min(1e-30))
        if e > 1e-6:
            print(f"module {k} {n} {tuple(pa.shape)}: max rel diff {e:.2e}")
for it in range(3):
    la = float(cap()); lb = float(wb.train_step()); torch.cuda.synchronize()
    worst = max(float((pa - pb).abs().max() / pb.abs().max().clamp_min(1e-30)) for ma, mb in zip(wa.modules, wb.modules) for pa, pb in zip(ma.parameters(), mb.parameters()))
    print(f"step {it + 2}: loss replay {la:.6f} direct {lb:.6f}  worst param diff {worst:.2e}")
