timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import torch, graph_pde_amd as gp
from graph_pde_amd import ops, synth
from tests.test_host_logic import DenseNet
d = torch.device("cuda:0"); torch.manual_seed(12)
conv = gp.NNConv_old(64, 64, DenseNet([6, 256, 256, 4096], torch.nn.ReLU), aggr="mean").to(d)
lin = ops.mlp_linears(conv.nn); pm = ops.pack_mlp([l.weight for l in lin], [l.bias for l in lin])
s = 221
ei, ea, n = synth.darcy_graph(s, 0.10, device=d, seed=0)
pos = synth.lattice_positions(s, d); a = synth.darcy_coefficient(s, 0).to(d)
na = gp.NodeAttr.darcy(pos, a)
print("materialize equal (all edges)", torch.equal(na.materialize(ei), ea))
x = torch.randn(n, 64, device=d); csr = ops.csr_for(ei, n)
perm = csr.perm.long()
print("csr consistent: src", torch.equal(csr.src.long(), ei[0][perm]), "dst", torch.equal(csr.dst.long(), ei[1][perm]))
for prec in ("f16split", "f16split_static", "f16split_8wave", "f16split_agg32"):
    y_t = ops.nnconv_forward_raw(x, csr, ea, pm, conv.root, conv.bias, "mean", precision=prec)
    y_n = ops.nnconv_forward_nodeattr_raw(x, csr, na, pm, conv.root, conv.bias, "mean", precision=prec)
    dif = (y_t != y_n).nonzero(); rows = dif[:, 0].unique()
    print(prec, "equal", torch.equal(y_t, y_n), "rows differing", rows.numel(), rows[:6].tolist(), "rel", float((y_t - y_n).norm() / y_t.norm()), flush=True)
PY
