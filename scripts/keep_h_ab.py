"""A/B of the kept-H training forward / backward (ops.keep_hidden, round 5): GPDE_SAVE_H_GB=0 (recompute, rounds 2-4) against the
default, s=121 and s=61, module autograd with the hidden cache off; every gradient compared between the two arms."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graph_pde_amd as gp
from graph_pde_amd import ops, synth, hidden_cache
hidden_cache.MODE = "off"
dev = torch.device("cuda:0")
for cfg, s in (("g121", 121), ("g61", 61)):
    torch.manual_seed(0)
    kw = 1024
    mlp = torch.nn.Sequential(torch.nn.Linear(6, kw), torch.nn.ReLU(), torch.nn.Linear(kw, kw), torch.nn.ReLU(), torch.nn.Linear(kw, 4096))
    conv = gp.NNConv_old(64, 64, mlp, aggr="mean").to(dev)
    ei, ea, n = synth.darcy_graph(s, 0.1, device=dev)
    x = torch.randn(n, 64, device=dev, requires_grad=True)
    g = torch.randn(n, 64, device=dev)
    res = {}
    for arm, gb in (("recompute", 0), ("kept_H", 32), ("recompute", 0), ("kept_H", 32)):
        ops.SAVE_H_BYTES = gb << 30
        tf, tb = [], []
        for it in range(4):
            conv.zero_grad(set_to_none=True); x.grad = None
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = conv(x, ei, ea)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            (out * g).sum().backward()
            torch.cuda.synchronize(); t2 = time.perf_counter()
            tf.append(t1 - t0); tb.append(t2 - t1)
        grads = [out.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in conv.parameters()]
        res.setdefault(arm, grads)
        print(f"{cfg} E={ei.shape[1]} {arm:9s}: fwd {1e3 * sorted(tf[1:])[1]:.2f} ms, bwd {1e3 * sorted(tb[1:])[1]:.2f} ms, pair {1e3 * (sorted(tf[1:])[1] + sorted(tb[1:])[1]):.2f} ms, "
              f"peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    names = ["out", "dx"] + [k for k, _ in conv.named_parameters()]
    for k, a, b in zip(names, res["recompute"], res["kept_H"]):
        rel = float((a.double() - b.double()).norm() / a.double().norm())
        print(f"   {k:12s} kept_H vs recompute rel-L2 {rel:.2e} {'bitwise' if torch.equal(a, b) else ''}")
