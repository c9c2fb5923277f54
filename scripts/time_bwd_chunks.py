"""Does the single-call backward gain from edge chunks small enough for their 4 KiB-per-edge intermediates to stay in the
256 MB Infinity Cache between producer and consumer kernels?  (Its PMC record, profiles/traffic_r04_bwd.json, shows 416 GB of
L2-side traffic per backward at s=121.)  Times gpde_nnconv_bwd with workspaces of different sizes = different chunkings.
usage: [FRACS=6,4,2,1] time_bwd_chunks.py [g121]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graph_pde_amd as gp
from graph_pde_amd import ops, synth, _lib
cfg = sys.argv[1] if len(sys.argv) > 1 else "g121"
s, r = {"g121": (121, 0.1), "g61": (61, 0.1)}[cfg]
dev = torch.device("cuda:0")
torch.manual_seed(0)
mlp = torch.nn.Sequential(torch.nn.Linear(6, 1024), torch.nn.ReLU(), torch.nn.Linear(1024, 1024), torch.nn.ReLU(), torch.nn.Linear(1024, 4096))
conv = gp.NNConv_old(64, 64, mlp, aggr="mean").to(dev)
ei, ea, n = synth.darcy_graph(s, r, device=dev)
e = ei.shape[1]
csr = ops.csr_for(ei, n)
lin = ops.mlp_linears(conv.nn)
ws_, bs_ = [l.weight.detach() for l in lin], [l.bias.detach() for l in lin]
x, g = torch.randn(n, 64, device=dev), torch.randn(n, 64, device=dev)
dims_c = _lib.dims_array([6, 1024, 1024, 4096])
full = int(_lib.lib().gpde_nnconv_bwd_workspace_bytes(n, e, 3, dims_c))
ref = None
fracs = [float(v) for v in os.environ.get("FRACS", "1,0.5,0.25,0.12,0.06,0.03,0.015").split(",")]      # multiples of the default workspace
for frac in fracs:
    nbytes = max(int(full * frac), 600 << 20)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    ts = []
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = ops.nnconv_backward_raw(x, csr, ea, ws_, bs_, conv.root.detach(), "mean", g, ws=ws)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    if ref is None:
        ref = out
    err = max(float((a - b).norm() / b.norm()) for a, b in zip(out[1], ref[1]))
    print(f"{cfg} workspace {nbytes / 2**30:6.2f} GiB (~{int(nbytes * 0.6 / 25500 / 1000)} k edges per chunk): backward {1e3 * sorted(ts[1:])[1]:.1f} ms = "
          f"{e / sorted(ts[1:])[1] / 1e6:.1f} M-edges/s, max rel diff of the weight gradients vs the largest workspace {err:.1e}", flush=True)
    del ws
