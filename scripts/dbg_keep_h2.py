import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graph_pde_amd as gp
from graph_pde_amd import ops, synth, hidden_cache, _lib
d = torch.device("cuda:0")
ei, ea, n = synth.darcy_graph(61, 0.10, device=d)
csr = ops.build_csr(ei, n)
for kw in (256, 1024):
    torch.manual_seed(0)
    dims = [6, kw, kw, 4096]
    mlp = torch.nn.Sequential(*sum([[torch.nn.Linear(dims[i], dims[i + 1]), torch.nn.ReLU()] for i in range(3)], [])[:-1])
    conv = gp.NNConv_old(64, 64, mlp, aggr="mean").to(d)
    lin = ops.mlp_linears(conv.nn)
    W, B = [l.weight.detach() for l in lin], [l.bias.detach() for l in lin]
    pm = ops.pack_mlp(W, B)
    x, g = torch.randn(n, 64, device=d), torch.randn(n, 64, device=d)
    root = conv.root.detach()
    H, hmax = ops.hidden_forward_raw(csr, ea, pm, W, B)
    z = torch.zeros(n, 64 * ops.hidden_width(pm.dims), dtype=torch.float32, device=d)
    ops.nnconv_forward_hidden_raw(x, csr, H, pm, root, conv.bias.detach(), "mean", hmax=hmax, z_keep=z)
    lib = _lib.lib()
    dims_c = _lib.dims_array(dims)
    one = int(lib.gpde_nnconv_bwd_workspace_bytes_one_chunk(n, csr.n_edges, 3, dims_c))
    dflt = int(lib.gpde_nnconv_bwd_workspace_bytes(n, csr.n_edges, 3, dims_c))
    print(kw, "one", one, "default", dflt, "H bytes", H.numel() * 4)
    def run(keep, nbytes, fill):
        ws = torch.empty(nbytes, dtype=torch.uint8, device=d)
        ws.view(torch.float32)[: nbytes // 4].fill_(fill) if fill == fill else ws.view(torch.float32)[: nbytes // 4].fill_(float("nan"))
        out = ops.nnconv_backward_raw(x, csr, ea, W, B, root, "mean", g, ws=ws, z_saved=z, hidden_saved=H if keep else None)
        torch.cuda.synchronize()
        return [out[0]] + list(out[1]) + list(out[2]) + [out[3], out[4]]
    base = run(False, one, 0.0)
    for tag, keep, nb, fill in (("rec one NaN", False, one, float("nan")), ("rec default 0", False, dflt, 0.0), ("kept one 0", True, one, 0.0),
                                ("kept one NaN", True, one, float("nan")), ("kept one-H 0", True, one - H.numel() * 4, 0.0),
                                ("kept default 0", True, dflt, 0.0), ("kept default NaN", True, dflt, float("nan"))):
        r = run(keep, nb, fill)
        print(f"  {kw} {tag:18s}", " ".join("=" if torch.equal(a, b) else f"{float((a.double()-b.double()).norm()/b.double().norm()):.0e}" for a, b in zip(r, base)))
