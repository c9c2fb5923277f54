"""Developer probe: the store variant of the fused kernel - determinism and agreement between the two kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graph_pde_amd import ops
from tests.test_gpu_bwd import _case
d = torch.device("cuda:0")
for dims, n, e in (([6, 256, 256, 4096], 200, 9000), ([6, 1024, 1024, 4096], 300, 20000), ([6, 512, 256, 4096], 2000, 70001)):
    x, ei, ea, ws_, bs_, root, bias, gout = _case(dims, n, e, 5)
    csr = ops.build_csr(ei.to(d), n)
    wd, bd = [w.to(d) for w in ws_], [b.to(d) for b in bs_]
    pm = ops.pack_mlp(wd, bd)
    hs = []
    for it in range(6):
        t = torch.full((64 << 20,), float(it), device=d); del t
        h, hm = ops.hidden_forward_raw(csr, ea.to(d), pm, wd[:-1] + [None], bd[:-1] + [None], "f16split")
        hs.append((h.clone(), float(hm)))
    same = [bool(torch.equal(hs[0][0], h[0])) for h in hs]
    # fp32 reference of H
    a = ea.to(d)[csr.perm.long()]
    h1 = torch.relu(a @ wd[0].t() + bd[0]); h2 = torch.relu(h1 @ wd[1].t() + bd[1])
    k2 = dims[2]
    err = float((hs[0][0][:, :k2] - h2).norm() / h2.norm())
    bad = (hs[0][0][:, :k2] - h2).abs().max()
    print(dims, e, "deterministic", same, "hmax", [h[1] for h in hs][:3], "true max %.6f" % float(h2.max()), "rel err vs fp32 torch %.2e max abs %.2e" % (err, float(bad)), flush=True)
    if not all(same):
        dif = (hs[0][0] != hs[[i for i, s_ in enumerate(same) if not s_][0]][0]).nonzero()
        print("  differing", dif.shape[0], "rows", dif[:, 0].unique()[:10].tolist(), "cols", dif[:, 1].unique()[:10].tolist())
