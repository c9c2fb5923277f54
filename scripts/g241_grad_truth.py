#!/usr/bin/env python3
"""One-off evidence run (not a test: ~3 PFLOP of float64 on the GPU): the gradients of three applications of ONE conv on the FULL
241^2 graph (E = 95,539,625, kernel MLP [6,1024,1024,4096]) against FLOAT64 autograd through the reference's op chain
(nn_conv.py:273-282, utilities.py:223-227 - the chunked formulation of oracle.nnconv_grads_shared, evaluated with torch float64 ops
on the device because 3 PFLOP do not fit a CPU tier), for
  * the module's default policy at this size (partial H + gpde_nnconv_bwd_light x 3 + ONE gpde_nnconv_bwd_deferred),
  * every application's own full backward (GPDE_HIDDEN_CACHE=off; split-f16 GEMMs),
  * the same with the two k1 x k2 GEMMs on exact fp32 MFMA (GPDE_BWD_GEMM_F32=1) - what the split products cost here.
Output -> profiles/r06_g241_grad_truth.txt (tests/test_gpu_headline_train.py states its bounds from it)."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import graph_pde_amd as gp                                    # noqa: E402
from graph_pde_amd import _lib, hidden_cache, ops, synth      # noqa: E402

DIMS, APPS = [6, 1024, 1024, 4096], 3
d = torch.device("cuda:0")
ei, ea, n = synth.darcy_graph(241, 0.10, device=d, seed=0)
e = int(ei.shape[1])
torch.manual_seed(241)
mlp = torch.nn.Sequential(*sum([[torch.nn.Linear(DIMS[i], DIMS[i + 1]), torch.nn.ReLU()] for i in range(3)], [])[:-1])
conv = gp.NNConv_old(64, 64, mlp, aggr="mean").to(d)
lin = ops.mlp_linears(conv.nn)
gen = torch.Generator(device=d).manual_seed(7)
xs = [torch.randn(n, 64, device=d, generator=gen) * (0.5 + 0.4 * l) for l in range(APPS)]
gs = [torch.randn(n, 64, device=d, generator=gen) * (2.0 ** -l) for l in range(APPS)]


def rel(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float(torch.linalg.vector_norm(a - b) / torch.linalg.vector_norm(b))


def step():
    conv.zero_grad(set_to_none=True)
    xin = [x.clone().requires_grad_(True) for x in xs]
    loss = sum((conv(x, ei, ea) * g).sum() for x, g in zip(xin, gs))
    loss.backward()
    torch.cuda.synchronize()
    got = {f"dx[{l}]": x.grad.clone() for l, x in enumerate(xin)}
    got.update({"droot": conv.root.grad.clone(), "dbias": conv.bias.grad.clone()})
    for k, l in enumerate(lin):
        got[f"dW{k + 1}"], got[f"db{k + 1}"] = l.weight.grad.clone(), l.bias.grad.clone()
    conv.zero_grad(set_to_none=True)
    return got


plans = {}
hidden_cache.MODE, hidden_cache.DEFER_MODE = "auto", "auto"
step()
t0 = time.perf_counter(); plans["default policy (partial H + light x 3 + deferred)"] = step(); t_def = time.perf_counter() - t0
hidden_cache.MODE = "off"; hidden_cache.clear(); torch.cuda.empty_cache()
t0 = time.perf_counter(); plans["per-application full backward, split-f16 GEMMs"] = step(); t_own = time.perf_counter() - t0
os.environ["GPDE_BWD_GEMM_F32"] = "1"; _lib.reload_switches()
t0 = time.perf_counter(); plans["per-application full backward, exact-fp32 GEMMs"] = step(); t_f32 = time.perf_counter() - t0
del os.environ["GPDE_BWD_GEMM_F32"]; _lib.reload_switches()
hidden_cache.clear(); ops.clear_caches(); torch.cuda.empty_cache()
print(f"GPU plans done: default {t_def:.1f} s, own {t_own:.1f} s, own fp32 GEMMs {t_f32:.1f} s", flush=True)

# ---- the reference's own op chain with stock torch ops on the device: the chunked formulation of oracle.nnconv_grads_shared,
#      in float64 (the adjudicator) and in float32 (the arithmetic the reference itself runs: what ITS distance to float64 is)
src, dst = ei[0], ei[1]


def chain(dt):
    t0 = time.perf_counter()
    torch.backends.cuda.matmul.allow_tf32 = False
    Ws = [l.weight.detach().to(dt).requires_grad_(True) for l in lin]
    Bs = [l.bias.detach().to(dt).requires_grad_(True) for l in lin]
    root = conv.root.detach().to(dt).requires_grad_(True)
    bias = conv.bias.detach().to(dt).requires_grad_(True)
    xl = [x.to(dt).requires_grad_(True) for x in xs]
    cnt = torch.bincount(dst, minlength=n).clamp(min=1).to(dt).unsqueeze(1)
    gT = [g.to(dt) / cnt for g in gs]
    CH = 49152
    for lo in range(0, e, CH):
        sl = slice(lo, min(lo + CH, e))
        h = ea[sl].to(dt)
        for k in range(3):                                                   # utilities.py:223-227
            h = torch.nn.functional.linear(h, Ws[k], Bs[k])
            if k < 2:
                h = torch.relu(h)
        we = h.view(-1, 64, 64)                                              # nn_conv.py:274
        loss = 0.0
        for x, g in zip(xl, gT):
            loss = loss + (torch.matmul(x[src[sl]].unsqueeze(1), we).squeeze(1) * g[dst[sl]]).sum()       # nn_conv.py:275
        loss.backward()
    loss = 0.0
    for x, g in zip(xl, gs):
        loss = loss + ((x @ root + bias) * g.to(dt)).sum()                   # nn_conv.py:277-282
    loss.backward()
    torch.cuda.synchronize()
    out = {f"dx[{l}]": x.grad for l, x in enumerate(xl)}
    out.update({"droot": root.grad, "dbias": bias.grad})
    for k in range(3):
        out[f"dW{k + 1}"], out[f"db{k + 1}"] = Ws[k].grad, Bs[k].grad
    print(f"{dt} reference chain on the device: {time.perf_counter() - t0:.1f} s", flush=True)
    return out


ref = chain(torch.float64)
plans["the reference's op chain in float32 (stock torch ops on the device: the arithmetic the reference itself runs)"] = chain(torch.float32)
keys = [f"dx[{l}]" for l in range(APPS)] + ["dW1", "db1", "dW2", "db2", "dW3", "db3", "droot", "dbias"]
lines = [f"G241 (N={n}, E={e}), kernel MLP 6-1024-1024-4096, {APPS} applications of one conv, loss = sum_l <conv(x_l), g_l>:",
         "relative L2 of every gradient against float64 autograd through the reference's op chain (evaluated on the device)", ""]
for name, got in plans.items():
    lines.append(f"{name}:")
    lines.append("   " + "  ".join(f"{k} {rel(got[k], ref[k]):.2e}" for k in keys))
a, b = plans["default policy (partial H + light x 3 + deferred)"], plans["per-application full backward, split-f16 GEMMs"]
lines += ["", "default policy vs per-application backward (what tests/test_gpu_headline_train.py compares in the tier):",
          "   " + "  ".join(f"{k} {rel(a[k], b[k]):.2e}" for k in keys),
          f"step times on this box: default policy {t_def:.2f} s, per-application {t_own:.2f} s, with fp32 GEMMs {t_f32:.2f} s"]
out = "\n".join(lines)
print(out)
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
open(os.path.join(REPO, "gpurun_out", "r06_g241_grad_truth.txt"), "w").write(out + "\n")
