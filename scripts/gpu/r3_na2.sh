timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import torch, graph_pde_amd as gp
from graph_pde_amd import ops, synth
from tests.test_host_logic import DenseNet
d = torch.device("cuda:0"); torch.manual_seed(12)
conv = gp.NNConv_old(64, 64, DenseNet([6, 1024, 1024, 4096], torch.nn.ReLU), aggr="mean").to(d)
lin = ops.mlp_linears(conv.nn); pm = ops.pack_mlp([l.weight for l in lin], [l.bias for l in lin])
for s in (121, 241):
    ei, ea, n = synth.darcy_graph(s, 0.10, device=d, seed=0)
    pos = synth.lattice_positions(s, d); a = synth.darcy_coefficient(s, 0).to(d)
    na = gp.NodeAttr.darcy(pos, a)
    x = torch.randn(n, 64, device=d); csr = ops.csr_for(ei, n)
    full = ops.workspace_bytes(n, csr.n_edges, pm)
    for frac in (1.0, 0.3):
        ws = torch.empty(int(full * frac), dtype=torch.uint8, device=d)
        plan = ops.launch_plan(n, csr.n_edges, pm, ws.numel())
        for prec in ("f16split", "f16split_static"):
            y_t = ops.nnconv_forward_raw(x, csr, ea, pm, conv.root, conv.bias, "mean", precision=prec, ws=ws)
            y_n = ops.nnconv_forward_nodeattr_raw(x, csr, na, pm, conv.root, conv.bias, "mean", precision=prec, ws=ws)
            y_n2 = ops.nnconv_forward_nodeattr_raw(x, csr, na, pm, conv.root, conv.bias, "mean", precision=prec, ws=ws)
            dif = (y_t != y_n).nonzero()
            rows = dif[:, 0].unique()
            print(s, plan, prec, "equal", torch.equal(y_t, y_n), "repro", torch.equal(y_n, y_n2), "ndiff", dif.shape[0], "rows", rows.numel(), rows[:8].tolist(), rows[-4:].tolist(),
                  "rel", float((y_t - y_n).norm() / y_t.norm()), flush=True)
PY
