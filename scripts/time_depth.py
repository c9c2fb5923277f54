"""depth x one conv module (the GKN pattern, UAI1_full_resolution.py:29-30): forward and training step,
hidden-activation reuse off vs on (developer probe).  usage: time_depth.py <cfg> [depth] [width]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graph_pde_amd as gp
from graph_pde_amd import ops, synth, hidden_cache
cfg = sys.argv[1] if len(sys.argv) > 1 else "g61"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 6
kw = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
s, r = {"g241": (241, 0.1), "g121": (121, 0.1), "g61": (61, 0.1), "g31": (31, 0.12), "g16": (16, 0.15)}[cfg]
FWD_ONLY = os.environ.get("FWD_ONLY", "0") == "1"       # skip the training step (G241: 41 s per step)
dev = torch.device("cuda:0")
torch.manual_seed(0)
mlp = torch.nn.Sequential(torch.nn.Linear(6, kw), torch.nn.ReLU(), torch.nn.Linear(kw, kw), torch.nn.ReLU(), torch.nn.Linear(kw, 4096))
conv = gp.NNConv_old(64, 64, mlp, aggr="mean").to(dev)
ei, ea, n = synth.darcy_graph(s, r, device=dev)
x = torch.randn(n, 64, device=dev)
e = ei.shape[1]

def model(xin):
    h = xin
    for k in range(depth):
        h = torch.relu(conv(h, ei, ea))
    return h

def timeit(fn, reps=3):
    if FWD_ONLY:
        reps = 2
        fn()
    else:
        fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

opt = torch.optim.Adam(conv.parameters(), lr=1e-4)
def fwd():
    with torch.no_grad():
        return model(x)
def step():
    opt.zero_grad()
    loss = model(x).square().mean()
    loss.backward()
    opt.step()
res = {}
for mode in os.environ.get("MODES", "off,auto").split(","):
    hidden_cache.MODE = mode
    hidden_cache.clear()
    ops.clear_caches()
    y = fwd()
    tf = timeit(fwd)
    ts = float("nan") if FWD_ONLY else timeit(step)
    res[mode] = (tf, ts, y)
    print(f"{cfg} E={e} depth={depth} k={kw} hidden_cache={mode}: forward {1e3*tf:.2f} ms "
          f"({depth*e/tf/1e6:.1f} M-edge-applications/s), training step {1e3*ts:.2f} ms "
          f"({depth*e/ts/1e6:.1f} M-edge-applications/s)  stats {hidden_cache.stats}", flush=True)
d = 0.0 if len(res) < 2 else ((res['off'][2] - res['auto'][2]).double().norm() / res['off'][2].double().norm()).item()
if len(res) == 2: print(f"speedup forward {res['off'][0]/res['auto'][0]:.2f}x  step {res['off'][1]/res['auto'][1]:.2f}x   rel-L2 between paths (after training drift differs; forward before steps): {d:.2e}")
