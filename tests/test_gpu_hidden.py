"""GPU tier, SURVEY.md §8 row f4: the operator split into hidden(edge_attr) -> H and conv(x, H)
(gpde_hidden_fwd / gpde_nnconv_fwd_hidden / gpde_nnconv_bwd with `hidden` / gpde_hidden_bwd) against the
float64 oracle, and the per-module reuse policy of graph_pde_amd/hidden_cache.py.  The reference
pattern being served: one conv module applied `depth` times with the same edge_attr and weights
(graph-neural-operator/UAI1_full_resolution.py:29-30)."""
import pytest
import torch

import graph_pde_amd as gp
from graph_pde_amd import _lib, hidden_cache, ops, synth
from oracle.nnconv_oracle import densenet_forward, nnconv_forward, rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-5


def dev():
    assert torch.cuda.is_available(), "GPU tier needs an MI355X"
    return torch.device("cuda:0")


def _mlp(dims):
    return torch.nn.Sequential(*sum([[torch.nn.Linear(dims[i], dims[i + 1]), torch.nn.ReLU()]
                                     for i in range(len(dims) - 1)], [])[:-1])


def _params(mlp):
    lin = [l for l in mlp if isinstance(l, torch.nn.Linear)]
    return [l.weight.detach() for l in lin], [l.bias.detach() for l in lin]


@pytest.mark.parametrize("dims,precision", [
    ([6, 64, 128, 4096], "f16split"),       # fused f16-split kernel with the store epilogue
    ([6, 64, 128, 4096], "f32"),            # general path: fp32 MFMA GEMMs over edge chunks
    ([6, 96, 200, 4096], "f16split"),       # widths that need padding (K1P = 96, K2P = 256)
    ([6, 48, 4096], "f16split"),            # one hidden layer (MGKN inter-level kernels)
    ([4, 24, 40, 56, 4096], "f16split"),    # three hidden layers
])
def test_hidden_forward_and_conv_from_hidden(dims, precision):
    d = dev()
    torch.manual_seed(sum(dims))
    n, e = 500, 9000
    ei = torch.stack([torch.randint(0, n, (e,)), torch.randint(0, n - 10, (e,))])
    ei[1, :700] = 5
    ea, x = torch.randn(e, dims[0]), torch.randn(n, 64)
    ws_, bs_ = _params(_mlp(dims))
    root, bias = torch.empty(64, 64).uniform_(-0.125, 0.125), torch.empty(64).uniform_(-0.125, 0.125)
    csr = ops.build_csr(ei.to(d), n)
    wd, bd = [w.to(d) for w in ws_], [b.to(d) for b in bs_]
    pm = ops.pack_mlp(wd, bd)
    calls = _lib.n_native_calls
    H, hmax = ops.hidden_forward_raw(csr, ea.to(d), pm, wd, bd, precision=precision)
    y = ops.nnconv_forward_hidden_raw(x.to(d), csr, H, pm, root.to(d), bias.to(d), "mean", hmax=hmax)
    if hmax is not None:
        assert float(hmax) == float(H.max())                       # recorded by the fused kernel
    torch.cuda.synchronize()
    assert _lib.n_native_calls == calls + 2
    # H against the oracle's hidden chain (float64), rows permuted into CSR order
    h64 = ea.double()
    for l in range(len(ws_) - 1):
        h64 = torch.relu(torch.nn.functional.linear(h64, ws_[l].double(), bs_[l].double()))
    k2 = dims[-2]
    Hc = H.cpu()
    assert Hc.shape == (e, ops.hidden_width(dims))
    assert rel_l2(Hc[:, :k2], h64[csr.perm.cpu().long()]) <= TOL
    assert float(Hc[:, k2:].abs().max() if Hc.shape[1] > k2 else 0.0) == 0.0     # padding columns
    y64 = nnconv_forward(x, ei, ea, ws_, bs_, root, bias, aggr="mean", dtype=torch.float64)
    assert rel_l2(y.cpu(), y64) <= TOL


def _ref_model_grads(x0, ei, ea, ws_, bs_, root, bias, y, depth):
    """float64 autograd of `depth` applications of one conv (shared weights) with ReLU between."""
    n = x0.shape[0]
    xs = x0.double().requires_grad_(True)
    Ws = [w.double().requires_grad_(True) for w in ws_]
    Bs = [b.double().requires_grad_(True) for b in bs_]
    r, bb = root.double().requires_grad_(True), bias.double().requires_grad_(True)

    def conv(xin):
        h = ea.double()
        for l in range(len(Ws)):
            h = torch.nn.functional.linear(h, Ws[l], Bs[l])
            if l != len(Ws) - 1:
                h = torch.relu(h)
        m = torch.matmul(xin[ei[0]].unsqueeze(1), h.view(-1, 64, 64)).squeeze(1)
        o = torch.zeros(n, 64, dtype=torch.float64).index_add(0, ei[1], m)
        o = o / torch.bincount(ei[1], minlength=n).clamp(min=1).double().unsqueeze(1)
        return o + xin @ r + bb
    h = xs
    for k in range(depth):
        h = conv(h)
        if k != depth - 1:
            h = torch.relu(h)
    loss = ((h - y.double()) ** 2).mean()
    loss.backward()
    return float(loss.detach()), xs.grad, [w.grad for w in Ws], [b.grad for b in Bs], r.grad, bb.grad


@pytest.mark.parametrize("dims", [[6, 32, 64, 4096], [6, 48, 4096], [4, 24, 40, 56, 4096]])
def test_shared_hidden_training_step_matches_reference(dims, monkeypatch):
    """depth = 3 applications of ONE module: all three share one H node; autograd sums dL/dH and the MLP
    backward runs once.  Gradients against float64 autograd of the reference formulation."""
    from tests.test_host_logic import DenseNet
    d = dev()
    monkeypatch.setattr(hidden_cache, "MODE", "on")
    hidden_cache.clear()
    torch.manual_seed(21 + len(dims))
    ei, ea, n = synth.darcy_graph(12, 0.2)
    if dims[0] != 6:
        ea = torch.randn(ea.shape[0], dims[0])
    x0, y = torch.randn(n, 64), torch.randn(n, 64)
    conv = gp.NNConv_old(64, 64, DenseNet(dims, torch.nn.ReLU), aggr="mean")
    lin = ops.mlp_linears(conv.nn)
    ws_ = [l.weight.detach().clone() for l in lin]
    bs_ = [l.bias.detach().clone() for l in lin]
    root, bias = conv.root.detach().clone(), conv.bias.detach().clone()
    lref, rx, rW, rb, rroot, rbias = _ref_model_grads(x0, ei, ea, ws_, bs_, root, bias, y, depth=3)
    conv = conv.to(d)
    eid, ead = ei.to(d), ea.to(d)
    xg = x0.to(d).requires_grad_(True)
    h = xg
    for k in range(3):
        h = conv(h, eid, ead)
        if k != 2:
            h = torch.relu(h)
    loss = ((h - y.to(d)) ** 2).mean()
    assert hidden_cache.stats["builds"] == 1 and hidden_cache.stats["hits"] == 2
    loss.backward()
    tol = 2e-5
    assert abs(float(loss.detach()) - lref) <= 1e-5 * abs(lref)
    assert rel_l2(xg.grad.cpu(), rx) <= tol
    assert rel_l2(conv.root.grad.cpu(), rroot) <= tol and rel_l2(conv.bias.grad.cpu(), rbias) <= tol
    lin = ops.mlp_linears(conv.nn)
    for l in range(len(lin)):
        assert rel_l2(lin[l].weight.grad.cpu(), rW[l]) <= tol, ("dW", l, rel_l2(lin[l].weight.grad.cpu(), rW[l]))
        assert rel_l2(lin[l].bias.grad.cpu(), rb[l]) <= tol, ("db", l)
    # the H node has been consumed by backward: the next forward (same weights) must rebuild, not reuse
    out = conv(x0.to(d), eid, ead)
    assert hidden_cache.stats["builds"] == 2
    # and an optimiser step changes the weights -> new key
    torch.optim.Adam(conv.parameters(), lr=1e-3).step()
    with torch.no_grad():
        conv(x0.to(d), eid, ead)
    assert hidden_cache.stats["builds"] == 3
    assert torch.isfinite(out).all()


def test_auto_policy_learns_the_repetition(monkeypatch):
    """auto: direct path until a module is seen repeating a key; then H from the first call on; a new
    edge_attr tensor (next sample) is a new key; results identical in class to the direct path."""
    from tests.test_host_logic import DenseNet
    d = dev()
    monkeypatch.setattr(hidden_cache, "MODE", "auto")
    hidden_cache.clear()
    torch.manual_seed(5)
    ei, ea, n = synth.darcy_graph(16, 0.15)
    conv = gp.NNConv_old(64, 64, DenseNet([6, 64, 128, 4096], torch.nn.ReLU), aggr="mean").to(d)
    eid, ead, x = ei.to(d), ea.to(d), torch.randn(n, 64, device=d)
    st = hidden_cache.stats
    with torch.no_grad():
        y1 = conv(x, eid, ead)                   # first ever call: direct
        assert (st["direct"], st["builds"], st["hits"]) == (1, 0, 0)
        y2 = conv(x, eid, ead)                   # same key again: learn, build H
        assert (st["direct"], st["builds"], st["hits"]) == (1, 1, 0)
        y3 = conv(x, eid, ead)                   # hit
        assert (st["direct"], st["builds"], st["hits"]) == (1, 1, 1)
        ead2 = ead.clone()                       # next sample: new tensor, same values
        y4 = conv(x, eid, ead2)                  # learned module: builds at the first call
        assert (st["direct"], st["builds"], st["hits"]) == (1, 2, 1)
        y5 = conv(x, eid, ead2)
        assert st["hits"] == 2
        ead2.mul_(1.0)                           # in-place write bumps the version: stale H not reused
        conv(x, eid, ead2)
        assert st["builds"] == 3
    lin = ops.mlp_linears(conv.nn)
    y64 = nnconv_forward(x.cpu(), ei, ea, [l.weight.detach().cpu() for l in lin],
                         [l.bias.detach().cpu() for l in lin], conv.root.detach().cpu(),
                         conv.bias.detach().cpu(), aggr="mean", dtype=torch.float64)
    for y in (y1, y2, y3, y4, y5):
        assert rel_l2(y.cpu(), y64) <= TOL
    # a module that stops repeating goes back to the direct path
    hidden_cache.clear()
    with torch.no_grad():
        conv(x, eid, ead); conv(x, eid, ead)     # learned
        for _ in range(3):
            conv(x, eid, ead.clone())            # never repeated
    assert st["direct"] >= 2


def test_budget_and_off(monkeypatch):
    from tests.test_host_logic import DenseNet
    d = dev()
    hidden_cache.clear()
    ei, ea, n = synth.darcy_graph(12, 0.2)
    conv = gp.NNConv_old(64, 64, DenseNet([6, 64, 128, 4096], torch.nn.ReLU), aggr="mean").to(d)
    eid, ead, x = ei.to(d), ea.to(d), torch.randn(n, 64, device=d)
    monkeypatch.setattr(hidden_cache, "MODE", "on")
    monkeypatch.setattr(hidden_cache, "BUDGET_BYTES", 1024)       # H does not fit
    with torch.no_grad():
        conv(x, eid, ead); conv(x, eid, ead)
    assert hidden_cache.stats["builds"] == 0 and hidden_cache.stats["direct"] == 2
    monkeypatch.setattr(hidden_cache, "MODE", "off")
    with torch.no_grad():
        conv(x, eid, ead)
    assert hidden_cache.stats["builds"] == 0


def test_edge_attr_from_node_table_matches_materialised_tensor():
    """edge_attr = [pos_src, pos_dst, a_src, a_dst]
    (graph-neural-operator/utilities.py:274-277) read from node data inside the fused kernel; same
    output as the forward on the materialised [E,6] tensor, and within 1e-5 of the oracle."""
    from tests.test_host_logic import DenseNet
    d = dev()
    torch.manual_seed(9)
    s, r = 24, 0.15
    ei = synth.lattice_radius_graph(s, r, d)
    pos = synth.lattice_positions(s, d)
    a = synth.darcy_coefficient(s, 3).to(d)
    ea = synth.darcy_edge_attr(ei, pos, a)
    n = s * s
    na = gp.NodeAttr.darcy(pos, a)
    assert torch.equal(na.materialize(ei), ea)                       # the recipe, slot for slot
    x = torch.randn(n, 64, device=d)
    for dims in ([6, 64, 128, 4096], [6, 128, 256, 4096]):
        conv = gp.NNConv_old(64, 64, DenseNet(dims, torch.nn.ReLU), aggr="mean").to(d)
        lin = ops.mlp_linears(conv.nn)
        pm = ops.pack_mlp([l.weight for l in lin], [l.bias for l in lin])
        csr = ops.csr_for(ei, n)
        for prec in ("f16split_agg32", "f16split_agg16"):
            calls = _lib.n_native_calls
            y_t = ops.nnconv_forward_raw(x, csr, ea, pm, conv.root, conv.bias, "mean", precision=prec)
            y_n = ops.nnconv_forward_nodeattr_raw(x, csr, na, pm, conv.root, conv.bias, "mean", precision=prec)
            torch.cuda.synchronize()
            assert _lib.n_native_calls == calls + 2
            assert torch.equal(y_t, y_n), (dims, prec, rel_l2(y_n.cpu(), y_t.cpu()))
        y64 = nnconv_forward(x.cpu(), ei.cpu(), ea.cpu(), [l.weight.detach().cpu() for l in lin],
                             [l.bias.detach().cpu() for l in lin], conv.root.detach().cpu(),
                             conv.bias.detach().cpu(), aggr="mean", dtype=torch.float64)
        with torch.no_grad():
            y_mod = conv(x, ei, na)                                   # module surface, inference
        assert rel_l2(y_mod.cpu(), y64) <= TOL
        # training through the module takes the materialised tensor: gradients flow as usual
        out = conv(x, ei, na)
        out.square().mean().backward()
        assert conv.root.grad is not None and torch.isfinite(conv.root.grad).all()
    # anything but the default f16-split kernel on a 3-Linear MLP is refused, not silently rerouted
    conv2 = gp.NNConv_old(64, 64, DenseNet([6, 64, 4096], torch.nn.ReLU), aggr="mean").to(d)
    lin2 = ops.mlp_linears(conv2.nn)
    pm2 = ops.pack_mlp([l.weight for l in lin2], [l.bias for l in lin2])
    with pytest.raises(NotImplementedError):
        ops.nnconv_forward_nodeattr_raw(x, ops.csr_for(ei, n), na, pm2, conv2.root, conv2.bias, "mean")


def test_streaming_aggregation_on_split_f16_from_32768_edges():
    """gpde_nnconv_fwd_hidden with the recorded max |H|: above 32768 edges the aggregation kernel runs on
    split-f16 MFMA (global scales from max |x| and max |H|); against the float64 oracle and against the
    fp32-MFMA aggregation (hmax withheld)."""
    d = dev()
    torch.manual_seed(17)
    ei, ea, n = synth.darcy_graph(32, 0.13)
    assert ei.shape[1] >= 32768
    dims = [6, 64, 128, 4096]
    ws_, bs_ = _params(_mlp(dims))
    root, bias = torch.empty(64, 64).uniform_(-0.125, 0.125), torch.empty(64).uniform_(-0.125, 0.125)
    x = torch.randn(n, 64) * torch.logspace(-2, 2, n).unsqueeze(1)       # rows from 1e-2 to 1e2
    csr = ops.build_csr(ei.to(d), n)
    wd, bd = [w.to(d) for w in ws_], [b.to(d) for b in bs_]
    pm = ops.pack_mlp(wd, bd)
    H, hmax = ops.hidden_forward_raw(csr, ea.to(d), pm, wd, bd, precision="f16split")
    assert hmax is not None and float(hmax) == float(H.max()) > 0
    y16 = ops.nnconv_forward_hidden_raw(x.to(d), csr, H, pm, root.to(d), bias.to(d), "mean", hmax=hmax)
    y32 = ops.nnconv_forward_hidden_raw(x.to(d), csr, H, pm, root.to(d), bias.to(d), "mean", hmax=None)
    y64 = nnconv_forward(x, ei, ea, ws_, bs_, root, bias, aggr="mean", dtype=torch.float64)
    e16, e32 = rel_l2(y16.cpu(), y64), rel_l2(y32.cpu(), y64)
    assert not torch.equal(y16, y32)                                     # two different arithmetics ran
    assert e16 <= TOL and e16 <= 4 * e32 + 2e-7, (e16, e32)


def test_partial_hidden_cache_and_mixed_forward(monkeypatch):
    """H larger than the budget (391 GB at the 241^2 graph): for inference the in-edges of the leading
    nodes that fit are cached and gpde_nnconv_fwd_mixed_keepz serves those nodes from H, the rest through the
    fused kernel.  Same output class as the direct path; a call that needs gradients does not use it."""
    from tests.test_host_logic import DenseNet
    d = dev()
    torch.manual_seed(23)
    ei, ea, n = synth.darcy_graph(32, 0.13)
    conv = gp.NNConv_old(64, 64, DenseNet([6, 64, 128, 4096], torch.nn.ReLU), aggr="mean").to(d)
    eid, ead, x = ei.to(d), ea.to(d), torch.randn(n, 64, device=d)
    lin = ops.mlp_linears(conv.nn)
    y64 = nnconv_forward(x.cpu(), ei, ea, [l.weight.detach().cpu() for l in lin],
                         [l.bias.detach().cpu() for l in lin], conv.root.detach().cpu(),
                         conv.bias.detach().cpu(), aggr="mean", dtype=torch.float64)
    full_bytes = ei.shape[1] * 128 * 4
    monkeypatch.setattr(hidden_cache, "MODE", "on")
    monkeypatch.setattr(hidden_cache, "BUDGET_BYTES", full_bytes // 2)
    hidden_cache.clear()
    with torch.no_grad():
        y1 = conv(x, eid, ead)
        ent = hidden_cache._entries[conv]
        assert 0 < ent.hn < n and ent.hn % 64 == 0
        assert ent.hidden.shape[0] == int(ops.csr_for(eid, n).rowptr_host[ent.hn])
        assert ent.hidden.numel() * 4 <= full_bytes // 2
        y2 = conv(x, eid, ead)
    assert hidden_cache.stats["builds"] == 1 and hidden_cache.stats["hits"] == 1
    assert torch.equal(y1, y2)
    assert rel_l2(y1.cpu(), y64) <= TOL
    # raw entry point, arbitrary split (not a multiple of the node chunk), against the direct forward
    csr = ops.csr_for(eid, n)
    pm = ops.pack_mlp([l.weight for l in lin], [l.bias for l in lin])
    wd, bd = [l.weight.detach() for l in lin], [l.bias.detach() for l in lin]
    y_dir = ops.nnconv_forward_raw(x, csr, ead, pm, conv.root, conv.bias, "mean")
    for hn in (1, 333, n - 1):
        H, hmax = ops.hidden_forward_raw(csr, ead, pm, wd, bd, n_nodes_limit=hn)
        y_mix = ops.nnconv_forward_mixed_raw(x, csr, ead, H, hmax, hn, pm, conv.root, conv.bias, "mean")
        assert rel_l2(y_mix.cpu(), y64) <= TOL and rel_l2(y_mix.cpu(), y_dir.cpu()) <= 1e-6, hn
    # gradients needed -> no partial cache: the direct path runs and differentiates
    hidden_cache.clear()
    out = conv(x, eid, ead)
    out.square().mean().backward()
    assert hidden_cache.stats["builds"] == 0 and conv.root.grad is not None
