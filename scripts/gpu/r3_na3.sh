timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import torch, graph_pde_amd as gp
from graph_pde_amd import ops, synth
from tests.test_host_logic import DenseNet
d = torch.device("cuda:0"); torch.manual_seed(12)
conv = gp.NNConv_old(64, 64, DenseNet([6, 256, 256, 4096], torch.nn.ReLU), aggr="mean").to(d)
lin = ops.mlp_linears(conv.nn); pm = ops.pack_mlp([l.weight for l in lin], [l.bias for l in lin])
for s in (161, 181, 201, 221, 241):
    ei, ea, n = synth.darcy_graph(s, 0.10, device=d, seed=0)
    pos = synth.lattice_positions(s, d); a = synth.darcy_coefficient(s, 0).to(d)
    na = gp.NodeAttr.darcy(pos, a)
    x = torch.randn(n, 64, device=d); csr = ops.csr_for(ei, n)
    out = torch.full((n, 64), 7.0, device=d)
    y_t = ops.nnconv_forward_raw(x, csr, ea, pm, conv.root, conv.bias, "mean")
    y_n = ops.nnconv_forward_nodeattr_raw(x, csr, na, pm, conv.root, conv.bias, "mean", out=out)
    torch.cuda.synchronize()
    print(s, "E", csr.n_edges, "equal", torch.equal(y_t, y_n), "y_n absmax", float(y_n.abs().max()), "sevens", int((y_n == 7.0).sum()), "zeros", int((y_n == 0).sum()),
          "nan", int(y_n.isnan().sum()), "y_t absmax", float(y_t.abs().max()), "plan", ops.launch_plan(n, csr.n_edges, pm, ops.workspace_bytes(n, csr.n_edges, pm)), flush=True)
    del ei, ea, y_t, y_n
PY
