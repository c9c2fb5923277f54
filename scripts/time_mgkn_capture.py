"""MGKN forwards, unmodified module calls: direct vs recorded into a HIP graph (gp.capture).  Developer probe / bench leg."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graph_pde_amd as gp
from graph_pde_amd import hidden_cache, mgkn_workloads
d = torch.device("cuda:0")
for name in sorted(mgkn_workloads.WORKLOADS):
    hidden_cache.clear()
    wl = mgkn_workloads.WORKLOADS[name](d)
    for _ in range(4):
        wl.forward()
    torch.cuda.synchronize()

    def timeit(f, n=30):
        ts = []
        for _ in range(n):
            torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        return 1e3 * sorted(ts)[len(ts) // 2]
    direct = timeit(wl.forward)
    cap = gp.capture(wl.forward)
    rep = timeit(cap)
    same = all(torch.equal(a, b) for a, b in zip(cap(), wl.forward()))
    print(f"{name}: {wl.calls} calls per forward: direct {direct:.3f} ms, captured {rep:.3f} ms, bit-identical {same}")
for name in sorted(mgkn_workloads.WORKLOADS):
    hidden_cache.clear()
    wl = mgkn_workloads.WORKLOADS[name](d, capturable=True)
    for _ in range(4):
        wl.train_step()
    torch.cuda.synchronize()
    direct = timeit(wl.train_step, 8)
    try:
        cap = gp.capture(wl.train_step, updates_parameters=True)
        rep = timeit(cap, 8)
        print(f"{name}: optimisation step: direct {direct:.2f} ms, captured {rep:.2f} ms")
    except Exception as ex:       # noqa: BLE001
        print(f"{name}: optimisation step: direct {direct:.2f} ms, capture failed: {type(ex).__name__}: {str(ex)[:300]}")
