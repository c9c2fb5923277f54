"""One optimisation step of the MGKN NNConv stacks (BASELINE configs 3 / 4; graph_pde_amd/mgkn_workloads.py train_step):
wall time per step and, under rocprofv3 --kernel-trace --stats, where it goes.  usage: time_mgkn_train.py [name] [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graph_pde_amd import mgkn_workloads, hidden_cache, _lib
names = [sys.argv[1]] if len(sys.argv) > 1 and sys.argv[1] != "all" else sorted(mgkn_workloads.WORKLOADS)
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
for name in names:
    hidden_cache.clear()
    wl = mgkn_workloads.WORKLOADS[name](dev)
    for _ in range(3):
        wl.train_step()
    torch.cuda.synchronize()
    ts = []
    c0 = _lib.n_native_calls
    for _ in range(steps):
        t0 = time.perf_counter()
        loss = wl.train_step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0, t1 - t0))
    print(f"{name}: train step {1e3 * sorted(t[0] for t in ts)[len(ts) // 2]:.2f} ms (host issue time {1e3 * sorted(t[1] for t in ts)[len(ts) // 2]:.2f} ms), "
          f"{(_lib.n_native_calls - c0) / steps:.0f} native calls per step, loss {float(loss):.4g}, cache {dict(hidden_cache.stats)}", flush=True)
