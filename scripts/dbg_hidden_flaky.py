"""Developer probe: cached-H forward under memory poisoning - which intermediate goes wrong."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graph_pde_amd import ops, mgkn_workloads, hidden_cache
d = torch.device("cuda:0")
def poison(val):
    t = torch.full((256 << 20,), val, device=d); del t
hidden_cache.MODE = "off"
wl = mgkn_workloads.general_darcy(d, seed=5)
conv, x, ei, ea = wl.pairs[0]
lin = ops.mlp_linears(conv.nn)
ws_ = [l.weight.detach() for l in lin]; bs_ = [l.bias.detach() for l in lin]
pm = ops.pack_mlp(ws_, bs_)
csr = ops.csr_for(ei, x.shape[0])
ops.DEFAULT_PRECISION = "f32"
with torch.no_grad():
    yref = conv(x, ei, ea)
ops.DEFAULT_PRECISION = "f16split"
H0, hm0 = ops.hidden_forward_raw(csr, ea, pm, ws_[:-1] + [None], bs_[:-1] + [None], "f16split")
print("H max", float(H0.max()), "hmax", float(hm0), flush=True)
for it in range(40):
    vals = [1e-30, 0.5, 1e-30, 2.0]
    poison(vals[it % 4])
    H, hm = ops.hidden_forward_raw(csr, ea, pm, ws_[:-1] + [None], bs_[:-1] + [None], "f16split")
    poison(vals[(it + 1) % 4])
    y = ops.nnconv_forward_hidden_raw(x, csr, H, pm, conv.root, conv.bias, "mean", hmax=hm)
    poison(vals[(it + 2) % 4])
    y0 = ops.nnconv_forward_hidden_raw(x, csr, H0, pm, conv.root, conv.bias, "mean", hmax=hm0)
    e1, e0 = float((y - yref).norm() / yref.norm()), float((y0 - yref).norm() / yref.norm())
    flag = "" if (e1 < 3e-6 and e0 < 3e-6 and torch.equal(H, H0) and float(hm) == float(hm0)) else "  <-- BAD"
    if not torch.equal(H, H0):
        dif = (H != H0).nonzero()
        rows = dif[:, 0].unique()
        cols = dif[:, 1].unique()
        print("   differing entries", dif.shape[0], "rows", rows[:8].tolist(), "n_rows", rows.numel(), "cols", cols[:6].tolist(), "n_cols", cols.numel(),
              "new", H[dif[0, 0], dif[0, 1]].item(), "old", H0[dif[0, 0], dif[0, 1]].item(),
              "row%32", (rows[:8] % 32).tolist(), flush=True)
        r0 = int(rows[0])
        print("   row", r0, "new", H[r0, :4].tolist(), "old", H0[r0, :4].tolist(), " dst of row:", int(csr.dst[r0]), "rowptr", int(csr.rowptr[int(csr.dst[r0])]), int(csr.rowptr[int(csr.dst[r0]) + 1]))
    print(it, "H equal", bool(torch.equal(H, H0)), "hmax", float(hm), "err new-H %.2e old-H %.2e" % (e1, e0), flag, flush=True)
