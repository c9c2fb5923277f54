"""Per-call times of the 13 distinct NNConv calls of MGKN-general (BASELINE config 4), inference, caches warm:
the re-associated path from cached hidden activations (zagg + gemm3 + epilogue) against the cached per-edge weight
operator (gpde_weconv) - the data behind hidden_cache.edge_weights_qualify."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graph_pde_amd import hidden_cache, mgkn_workloads, ops
from graph_pde_amd.nn_conv import nnconv_group

dev = torch.device("cuda:0")
wl = mgkn_workloads.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "mgkn_general_darcy2d"](dev)


def timed(fn, n=40):
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n


qualify = hidden_cache.edge_weights_qualify
print(f"{'nodes':>6} {'dst':>6} {'edges':>7} {'deg':>6} dims                      reassoc_us  weconv_us  policy")
with torch.no_grad():
    for conv, x, ei, ea in wl.pairs:
        csr = ops.csr_for(ei, x.size(0))
        ndst = int((csr.rowptr[1:] > csr.rowptr[:-1]).sum())
        dims = [ops.mlp_linears(conv.nn)[0].in_features] + [l.out_features for l in ops.mlp_linears(conv.nn)]
        hidden_cache.WE_MODE = "off"
        t_re = timed(lambda: conv(x, ei, ea))
        hidden_cache.edge_weights_qualify = lambda c, force=False, explicit=False: c.n_edges * ops.EDGE_WEIGHT_BYTES <= (6 << 30)
        try:
            t_we = timed(lambda: nnconv_group([(conv, x, ei, ea)]))
        finally:
            hidden_cache.edge_weights_qualify = qualify
        pol = "weconv" if qualify(csr, explicit=True) else "reassoc"
        print(f"{csr.n_nodes:6d} {ndst:6d} {csr.n_edges:7d} {csr.n_edges / max(ndst, 1):6.1f} {str(dims):25s} {t_re:9.1f} {t_we:10.1f}  {pol}")
        hidden_cache.clear()
