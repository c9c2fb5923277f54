"""GPU tier: SURVEY.md §8 row f3 in TRAINING - edge attributes read from node data (ops.NodeAttr: the reference's recipe
edge_attr = [pos_src, pos_dst, a_src, a_dst], graph-neural-operator/utilities.py:274-277) through every training-side native
call (`_na` entry points of include/gpde.h): forward with keep-Z, full / light / deferred backward, the hidden activations and
their backward.  The attribute values are the same floats as the materialised tensor's, so every result must be BITWISE the
tensor path's - no [E, 6] tensor, no slot-order copy, no perm."""
import pytest
import torch

import graph_pde_amd as gp
from graph_pde_amd import hidden_cache, ops, synth
from tests.test_host_logic import DenseNet

pytestmark = pytest.mark.gpu


def _setup(s=31, r=0.12, dims=(6, 256, 256, 4096), seed=0):
    d = torch.device("cuda:0")
    torch.manual_seed(seed)
    pos = synth.lattice_positions(s, d)
    a = synth.darcy_coefficient(s, seed).to(d)
    csr = ops.radius_csr(pos, r)
    na = gp.NodeAttr.darcy(pos, a)
    ea = na.materialize(csr.edge_index)
    conv = gp.NNConv_old(64, 64, DenseNet(list(dims), torch.nn.ReLU), aggr="mean").to(d)
    return d, csr, na, ea, conv, s * s


def _same(a, b, what):
    if isinstance(a, (list, tuple)):
        for k, (u, v) in enumerate(zip(a, b)):
            _same(u, v, f"{what}[{k}]")
    elif a is not None:
        assert torch.equal(a, b), (what, float((a - b).norm() / b.norm().clamp_min(1e-30)))


@pytest.mark.parametrize("s,dims", [(31, (6, 256, 256, 4096)), (61, (6, 1024, 1024, 4096))])
def test_raw_training_calls_from_node_data_are_bitwise_the_tensor_path(s, dims):
    d, csr, na, ea, conv, n = _setup(s, 0.10 if s == 61 else 0.12, dims)
    assert ops.nodeattr_train_supported(list(dims))
    lin = ops.mlp_linears(conv.nn)
    ws_, bs_ = [l.weight.detach() for l in lin], [l.bias.detach() for l in lin]
    pm = ops.pack_mlp(ws_, bs_)
    x, g = torch.randn(n, 64, device=d), torch.randn(n, 64, device=d)
    root, bias = conv.root.detach(), conv.bias.detach()
    # forward with keep-Z
    z1, z2 = ops.z_buffer(csr, pm.dims, d), ops.z_buffer(csr, pm.dims, d)
    y_t = ops.nnconv_forward_raw(x, csr, ea, pm, root, bias, "mean", z_keep=z1)
    y_n = ops.nnconv_forward_raw(x, csr, na, pm, root, bias, "mean", z_keep=z2)
    _same(y_n, y_t, "forward")
    _same(z2, z1, "Z")
    # full backward, with and without the kept Z
    _same(ops.nnconv_backward_raw(x, csr, na, ws_, bs_, root, "mean", g), ops.nnconv_backward_raw(x, csr, ea, ws_, bs_, root, "mean", g), "bwd")
    _same(ops.nnconv_backward_raw(x, csr, na, ws_, bs_, root, "mean", g, z_saved=z2),
          ops.nnconv_backward_raw(x, csr, ea, ws_, bs_, root, "mean", g, z_saved=z1), "bwd_z")
    # hidden activations (whole and partial), mixed forward, light and deferred backward with the partial H
    hn = (n // 2) // 64 * 64
    h_t, m_t = ops.hidden_forward_raw(csr, ea, pm, ws_[:-1] + [None], bs_[:-1] + [None], n_nodes_limit=hn)
    h_n, m_n = ops.hidden_forward_raw(csr, na, pm, ws_[:-1] + [None], bs_[:-1] + [None], n_nodes_limit=hn)
    _same(h_n, h_t, "partial H")
    _same(m_n, m_t, "max |H|")
    _same(ops.nnconv_forward_mixed_raw(x, csr, na, h_n, m_n, hn, pm, root, bias, "mean"),
          ops.nnconv_forward_mixed_raw(x, csr, ea, h_t, m_t, hn, pm, root, bias, "mean"), "mixed forward")
    for hp, k in ((None, 0), (h_n, hn)):
        _same(ops.nnconv_backward_light_raw(x, csr, na, ws_, bs_, root, "mean", g, hidden_part=hp, hidden_nodes=k),
              ops.nnconv_backward_light_raw(x, csr, ea, ws_, bs_, root, "mean", g, hidden_part=hp, hidden_nodes=k), f"light {k}")
        xs = [torch.randn(n, 64, device=d) for _ in range(3)]
        gs = [torch.randn(n, 64, device=d) for _ in range(3)]
        _same(ops.nnconv_backward_deferred_raw(xs, gs, csr, na, ws_, bs_, "mean", hidden_part=hp, hidden_nodes=k),
              ops.nnconv_backward_deferred_raw(xs, gs, csr, ea, ws_, bs_, "mean", hidden_part=hp, hidden_nodes=k), f"deferred {k}")
    # the hidden layers' backward from a given dL/dU (the H-cached training path)
    hf, _ = ops.hidden_forward_raw(csr, na, pm, ws_[:-1] + [None], bs_[:-1] + [None])
    gh = torch.randn_like(hf) * (hf > 0)
    _same(ops.hidden_backward_raw(csr, na, list(pm.dims), ws_[:-1], bs_[:-1], gh),
          ops.hidden_backward_raw(csr, ea, list(pm.dims), ws_[:-1], bs_[:-1], gh), "hidden backward")


class _Net(torch.nn.Module):
    def __init__(self, conv, depth):
        super().__init__()
        self.conv1, self.depth = conv, depth

    def forward(self, x, graph, attr):
        for _ in range(self.depth):
            x = torch.relu(self.conv1(x, graph, attr))
        return x


@pytest.mark.parametrize("budget", [None, 0, 0.6])
def test_module_trains_from_node_data_on_every_policy_path(budget, monkeypatch):
    """depth x one conv with gradients, attributes given as NodeAttr: default policy (H fits: shared hidden activations),
    H too large (virtual-H node, deferred backward), partly fitting (partial H + deferred) - gradients bitwise those of the
    same run on the materialised tensor; a graph given as a plain edge_index (reference edge order) as well."""
    d, csr, na, ea, conv, n = _setup(31, 0.12, (6, 256, 256, 4096), seed=3)
    net = _Net(conv, 4)
    x, tgt = torch.randn(n, 64, device=d), torch.randn(n, 64, device=d)
    if budget is not None:
        monkeypatch.setattr(hidden_cache, "BUDGET_BYTES", int(budget * csr.n_edges * 256 * 4))

    def run(graph, attr):
        hidden_cache.clear()
        outs = []
        for _ in range(2):                                   # second step: the steady-state policy
            net.zero_grad(set_to_none=True)
            xin = x.clone().requires_grad_(True)
            loss = ((net(xin, graph, attr) - tgt) ** 2).mean()
            loss.backward()
            torch.cuda.synchronize()
            with torch.no_grad():
                for p in net.parameters():
                    p.mul_(1.0)                              # an optimizer step: versions move
            outs = [xin.grad.clone()] + [p.grad.clone() for p in net.parameters()] + [loss.detach().clone()]
        return outs, dict(hidden_cache.stats)
    ref, st_t = run(csr, ea)
    got, st_n = run(csr, na)
    _same(got, ref, f"budget {budget}")
    assert {k: st_n.get(k) for k in ("builds", "hits", "deferred_builds")} == {k: st_t.get(k) for k in ("builds", "hits", "deferred_builds")}
    if budget is not None:
        assert st_n.get("deferred_builds", 0) >= 1
    ei = synth.darcy_graph(31, 0.12, device=d, seed=3)[0]   # the reference's source-major list of the same lattice graph
    csr2 = ops.csr_for(ei, n)
    if csr2.n_edges == csr.n_edges:                         # (exact-integer lattice rule vs float64 distances: equal at this r)
        got2, _ = run(ei, na)
        ref2, _ = run(ei, na.materialize(ei))
        _same(got2, ref2, "edge_index graph")
