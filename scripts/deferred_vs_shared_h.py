"""Depth-6 training step when H FITS memory: the shared-H path (autograd.HiddenFunction: one H node, every application's
backward writes dL/dH [E, K2P], autograd sums them, one MLP backward) against the depth-deferred pair fed with (almost) the whole H
(budget just below H: light pass per application reading H, ONE deferred pass - the sum over the applications is taken inside
the gather GEMM, no dL/dH tensors).  usage: deferred_vs_shared_h.py <g61|g121> [depth]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graph_pde_amd as gp
from graph_pde_amd import ops, synth, hidden_cache
cfg = sys.argv[1] if len(sys.argv) > 1 else "g61"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 6
s = {"g61": 61, "g121": 121}[cfg]
dev = torch.device("cuda:0")
ei, ea, n = synth.darcy_graph(s, 0.1, device=dev)
e = ei.shape[1]
hbytes = e * 1024 * 4
for tag, budget in (("shared H (budget auto)", None), ("deferred pair, 98 % of H given", int(0.985 * hbytes)), ("shared H (budget auto)", None),
                    ("deferred pair, 98 % of H given", int(0.985 * hbytes))):
    hidden_cache.BUDGET_BYTES = budget
    torch.manual_seed(0)
    mlp = torch.nn.Sequential(torch.nn.Linear(6, 1024), torch.nn.ReLU(), torch.nn.Linear(1024, 1024), torch.nn.ReLU(), torch.nn.Linear(1024, 4096))
    conv = gp.NNConv_old(64, 64, mlp, aggr="mean").to(dev)
    fc1, fc2 = torch.nn.Linear(6, 64).to(dev), torch.nn.Linear(64, 1).to(dev)
    params = list(conv.parameters()) + list(fc1.parameters()) + list(fc2.parameters())
    opt = torch.optim.Adam(params, lr=1e-4, weight_decay=5e-4)
    a_in, y_t = torch.randn(n, 6, device=dev), torch.randn(n, device=dev)
    hidden_cache.clear(); ops.clear_caches()
    for k in hidden_cache.stats: hidden_cache.stats[k] = 0
    ts, losses = [], []
    for it in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        h = fc1(a_in)
        for _ in range(depth):
            h = torch.relu(conv(h, ei, ea))
        loss = torch.norm(fc2(h).view(-1) - y_t, 1)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        loss.backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        opt.step()
        torch.cuda.synchronize()
        ts.append((t1 - t0, t2 - t1, time.perf_counter() - t0)); losses.append(float(loss.detach()))
    best = min(ts[2:], key=lambda t: t[2])
    print(f"{cfg} E={e} depth={depth} {tag:32s}: fwd {1e3 * best[0]:.1f} bwd {1e3 * best[1]:.1f} step {1e3 * best[2]:.1f} ms  losses {[f'{l:.6g}' for l in losses[:3]]}  "
          f"{ {k: v for k, v in hidden_cache.stats.items() if v} }  peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
