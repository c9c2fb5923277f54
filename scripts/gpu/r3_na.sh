O=gpurun_out/r3_06
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_v6.py -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids $O/pytest.log | grep -E "^E  |passed|failed" | head -20
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import torch, graph_pde_amd as gp
from graph_pde_amd import ops, synth
from tests.test_host_logic import DenseNet
d = torch.device("cuda:0"); torch.manual_seed(12)
s, r = 61, 0.10
ei = synth.lattice_radius_graph(s, r, d); pos = synth.lattice_positions(s, d); a = synth.darcy_coefficient(s, 5).to(d)
ea = synth.darcy_edge_attr(ei, pos, a); n = s * s; na = gp.NodeAttr.darcy(pos, a)
print("materialize equal", torch.equal(na.materialize(ei), ea))
x = torch.randn(n, 64, device=d); csr = ops.csr_for(ei, n)
conv = gp.NNConv_old(64, 64, DenseNet([6, 256, 256, 4096], torch.nn.ReLU), aggr="mean").to(d)
lin = ops.mlp_linears(conv.nn); pm = ops.pack_mlp([l.weight for l in lin], [l.bias for l in lin])
for prec in ("f16split", "f16split_static", "f16split_8wave"):
    y_t = ops.nnconv_forward_raw(x, csr, ea, pm, conv.root, conv.bias, "mean", precision=prec)
    y_n = ops.nnconv_forward_nodeattr_raw(x, csr, na, pm, conv.root, conv.bias, "mean", precision=prec)
    y_n2 = ops.nnconv_forward_nodeattr_raw(x, csr, na, pm, conv.root, conv.bias, "mean", precision=prec)
    dif = (y_t != y_n).nonzero()
    print(prec, "equal", torch.equal(y_t, y_n), "repro", torch.equal(y_n, y_n2), "ndiff", dif.shape[0], "rows", dif[:, 0].unique().numel(), "rel", float((y_t - y_n).norm() / y_t.norm()), dif[:5].tolist())
PY
