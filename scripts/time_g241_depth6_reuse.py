"""Depth-6 inference on the headline graph (G241, 95.5 M edges, kernel MLP 6-1024-1024-4096) on ONE GPU with the
cross-depth reuse of the hidden activations at different budgets: the full H is 391 GB, so the cache is
PARTIAL - the in-edges of the first nodes that fit are served from H, the rest recomputed (gpde_nnconv_fwd_mixed_keepz)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graph_pde_amd import hidden_cache, synth
import bench

dev = torch.device("cuda:0")
conv = bench.make_conv(1024, dev)
ei, ea, n = synth.darcy_graph(241, 0.10, device=dev, seed=0)
e = int(ei.shape[1])
x0 = torch.randn(n, 64, device=dev, generator=torch.Generator(device=dev).manual_seed(1000))


def forward():
    h = x0
    for _ in range(6):
        h = torch.relu(conv(h, ei, ea))
    return h


ref = None
with torch.no_grad():
    for gb in (0, 32, 96, 180):
        hidden_cache.clear()
        torch.cuda.empty_cache()
        hidden_cache.MODE = "off" if gb == 0 else "auto"
        hidden_cache.BUDGET_BYTES = gb << 30
        forward()                                   # first forward: the module is seen repeating, H is built
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        y = forward()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        ref = y if ref is None else ref
        rel = float(torch.norm(y - ref) / torch.norm(ref))
        free, total = torch.cuda.mem_get_info()
        print(f"budget {gb:3d} GB: depth-6 forward {ms:7.1f} ms = {6 * e / ms / 1e3:6.1f} M edge-applications/s; vs reuse off rel-L2 {rel:.2e}; "
              f"HBM in use {(total - free) / 2**30:.0f} GiB; stats {hidden_cache.stats}")
