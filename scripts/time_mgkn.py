"""Forward time of the MGKN configurations (BASELINE configs 3 and 4) through the drop-in module:
many small NNConv calls per model forward (developer probe; numbers quoted in DESIGN.md)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import graph_pde_amd as gp
from graph_pde_amd import synth, _lib

dev = torch.device("cuda:0")

class DenseNet(torch.nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.layers = torch.nn.ModuleList()
        for j in range(len(layers) - 1):
            self.layers.append(torch.nn.Linear(layers[j], layers[j + 1]))
            if j != len(layers) - 2:
                self.layers.append(torch.nn.ReLU())
    def forward(self, x):
        for l in self.layers: x = l(x)
        return x

def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    th = (time.perf_counter() - t0) / n              # host time to enqueue
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / n
    # the same forward captured once into a HIP graph and replayed
    tg = float("nan")
    try:
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            fn()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            fn()
        graph.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): graph.replay()
        torch.cuda.synchronize()
        tg = (time.perf_counter() - t0) / n
    except Exception as ex:
        print("graph capture failed:", repr(ex)[:300])
    print(f"   eager {1e3*t:.2f} ms (host enqueue {1e3*th:.2f} ms), HIP-graph replay {1e3*tg:.2f} ms")
    return min(t, tg) if tg == tg else t

def burgers(s=8192, depth=4, ker_width=1024):
    graphs = [(ei.to(dev), ea.to(dev), n) for ei, ea, n in synth.burgers_multipole_graphs(s)]
    level = len(graphs) - 1
    convs = [gp.NNConv(64, 64, DenseNet([4, max(ker_width // 2 ** l, 16), max(ker_width // 2 ** l, 16), 4096]), aggr="mean").to(dev)
             for l in range(level + 1)]
    xs = [torch.randn(n, 64, device=dev) for _, _, n in graphs]
    edges = sum(g[0].shape[1] for g in graphs)
    def fwd():
        with torch.no_grad():
            for k in range(depth):                       # MGKN_orthogonal_burgers1d.py:65-82 call pattern
                for l in range(level + 1):
                    convs[l](xs[l], graphs[l][0], graphs[l][1])
    calls0 = _lib.n_native_calls
    t = timeit(fwd)
    ncalls = (_lib.n_native_calls - calls0) // 7
    print(f"MGKN-orthogonal Burgers s={s}: {ncalls} NNConv calls, {edges * depth} edges per forward, {1e3 * t:.2f} ms per forward, {edges * depth / t / 1e6:.2f} M-edges/s")

def darcy_general(depth=5, ker_width=256):
    m = [2400, 1600, 400, 100, 25]
    g = synth.sampled_multilevel_graphs(141, m, [0.5 / 8 * 1.41, 0.5 / 8, 0.5 / 4, 0.5 / 2, 0.5], [0.5 / 8 * 1.1, 0.5 / 8 * 1.41, 0.5 / 4 * 1.41, 0.5 / 2 * 1.41], device=dev)
    L = len(m)
    offs = [0]
    for ml in m: offs.append(offs[-1] + ml)
    x = torch.randn(offs[-1], 64, device=dev)
    inner = [gp.NNConv(64, 64, DenseNet([6, ker_width // 2 ** l, ker_width // 2 ** l, 4096]), aggr="mean", root_weight=True, bias=False).to(dev) for l in range(L)]
    down = [gp.NNConv(64, 64, DenseNet([6, ker_width // 2 ** l, 4096]), aggr="mean", root_weight=False, bias=False).to(dev) for l in range(L - 1)]
    up = [gp.NNConv(64, 64, DenseNet([6, ker_width // 2 ** l, 4096]), aggr="mean", root_weight=False, bias=False).to(dev) for l in range(L - 1)]
    gd = [(torch.stack([g["down"][l][0][0] + offs[l], g["down"][l][0][1] + offs[l + 1]]), g["down"][l][1]) for l in range(L - 1)]
    gu = [(torch.stack([g["up"][l][0][0] + offs[l + 1], g["up"][l][0][1] + offs[l]]), g["up"][l][1]) for l in range(L - 1)]
    edges = sum(g["inner"][l][0].shape[1] for l in range(L)) + 2 * sum(gd[l][0].shape[1] for l in range(L - 1))
    def fwd():
        with torch.no_grad():
            xx = x
            for t in range(depth):                        # MGKN_general_darcy2d.py:76-90 call pattern
                for l in range(L - 1):
                    xx = F.relu(xx + down[l](xx, gd[l][0], gd[l][1]))
                for l in reversed(range(L)):
                    a, b = offs[l], offs[l + 1]
                    xx = xx.clone()
                    xx[a:b] = inner[l](xx[a:b].clone(), g["inner"][l][0], g["inner"][l][1])
                    if l > 0:
                        xx = F.relu(xx + up[l - 1](xx, gu[l - 1][0], gu[l - 1][1]))
    t = timeit(fwd)
    print(f"MGKN-general Darcy m={m}: {(3 * L - 2) * depth} NNConv calls, {edges * depth} edges per forward, {1e3 * t:.2f} ms per forward, {edges * depth / t / 1e6:.2f} M-edges/s")

burgers()
darcy_general()
