"""Host-side mirror of the reference module surface (no GPU)."""
import io
import os
import pickle

import pytest
import torch

import graph_pde_amd as gp
from graph_pde_amd import ops


class DenseNet(torch.nn.Module):
    """Shape of the reference kernel network (utilities.py:201-227): ModuleList `layers`."""
    def __init__(self, layers, nonlinearity):
        super().__init__()
        self.n_layers = len(layers) - 1
        self.layers = torch.nn.ModuleList()
        for j in range(self.n_layers):
            self.layers.append(torch.nn.Linear(layers[j], layers[j + 1]))
            if j != self.n_layers - 1:
                self.layers.append(nonlinearity())

    def forward(self, x):
        for l in self.layers:
            x = l(x)
        return x


def test_module_surface_matches_reference():
    conv = gp.NNConv_old(64, 64, DenseNet([6, 32, 48, 4096], torch.nn.ReLU), aggr="mean")
    assert (conv.in_channels, conv.out_channels, conv.aggr) == (64, 64, "mean")
    assert tuple(conv.root.shape) == (64, 64) and tuple(conv.bias.shape) == (64,)
    assert repr(conv) == "NNConv_old(64, 64)"
    sd = conv.state_dict()
    # names the reference checkpoints use
    for k in ("root", "bias", "nn.layers.0.weight", "nn.layers.2.bias", "nn.layers.4.weight"):
        assert k in sd
    bound = 1.0 / 8.0
    assert float(conv.root.abs().max()) <= bound and float(conv.bias.abs().max()) <= bound
    c2 = gp.NNConv(64, 64, DenseNet([6, 16, 4096], torch.nn.ReLU), aggr="mean", root_weight=False,
                   bias=False)
    assert c2.root is None and c2.bias is None and repr(c2) == "NNConv(64, 64)"
    assert "root" not in c2.state_dict()


def test_module_pickles_like_torch_save_model():
    conv = gp.NNConv_old(64, 64, DenseNet([6, 8, 4096], torch.nn.ReLU), aggr="mean")
    buf = io.BytesIO()
    torch.save(conv, buf)
    buf.seek(0)
    c2 = torch.load(buf, weights_only=False)
    assert torch.equal(c2.root, conv.root) and repr(c2) == repr(conv)
    pickle.dumps(conv)


@pytest.mark.skipif(torch.cuda.is_available(), reason="with a HIP device CPU tensors are staged to it (GPU tier)")
def test_cpu_tensors_without_a_gpu_raise_not_silently_compute():
    """CPU tensors are STAGED to the HIP device (one execution path); on a box without one the call must fail
    loudly - there is no composite / oracle fallback to fall into."""
    conv = gp.NNConv_old(64, 64, DenseNet([6, 8, 4096], torch.nn.ReLU), aggr="mean")
    x = torch.randn(5, 64)
    ei = torch.tensor([[0, 1, 2], [1, 2, 3]])
    ea = torch.randn(3, 6)
    with pytest.raises(RuntimeError, match="no CPU"):
        conv(x, ei, ea)
    with pytest.raises(RuntimeError, match="no CPU"):
        conv.message(x[:3], ea)
    with pytest.raises(RuntimeError, match="no CPU"):
        conv.update(x, x)


def test_module_surface_has_the_reference_methods():
    """nn_conv.py:234-286: __init__, reset_parameters, forward, message, update, __repr__."""
    import inspect
    for name, params in (("forward", ["self", "x", "edge_index", "edge_attr"]), ("message", ["self", "x_j", "pseudo"]),
                         ("update", ["self", "aggr_out", "x"]), ("reset_parameters", ["self"])):
        sig = inspect.signature(getattr(gp.NNConv_old, name)).parameters
        positional = [k for k, v in sig.items() if v.kind is not inspect.Parameter.KEYWORD_ONLY]
        assert positional == params, name
        # extras (the opt-in fused glue of forward) are keyword-only and default to "off"
        assert all(v.default is None for v in sig.values() if v.kind is inspect.Parameter.KEYWORD_ONLY), name


def test_inference_tensor_checksum_is_position_sensitive_and_scoped():
    """ops._content_hash / ops.ver_scope: a row permutation or a sum-preserving edit of an inference tensor moves
    the key (the old word sum did not see either); inside one operator call the checksum is computed once per tensor."""
    with torch.inference_mode():
        t = torch.arange(3 * 5000, dtype=torch.float32).reshape(5000, 3)
        k0 = ops._ver(t)
        p = t.clone()
        p[[10, 4000]] = p[[4000, 10]]                  # two rows swapped: same multiset of words
        assert ops._ver(p) != k0
        q = t.clone()
        q[7, 0] += 1.0
        q[9, 0] -= 1.0                                 # sum-preserving edit
        assert ops._ver(q) != k0
        assert ops._ver(t.clone()) == k0               # same contents elsewhere in memory: same checksum
        idx = torch.randint(0, 100, (2, 70001))        # int64, a byte count that is not a multiple of the row size
        assert ops._ver(idx) == ops._ver(idx.clone()) and ops._ver(idx) != ops._ver(idx.flip(1))
        calls = {"n": 0}
        real = ops._content_hash

        def counting(x):
            calls["n"] += 1
            return real(x)
        ops._content_hash = counting
        try:
            with ops.ver_scope():
                a = ops._ver(t)
                with ops.ver_scope():                  # nested calls (forward -> propagate) share the outer scope
                    assert ops._ver(t) == a
                assert ops._ver(t) == a
            assert calls["n"] == 1
            ops._ver(t), ops._ver(t)                   # outside a scope nothing is remembered
            assert calls["n"] == 3
        finally:
            ops._content_hash = real


def test_version_helper_accepts_inference_tensors():
    with torch.inference_mode():
        t = torch.arange(4)
        k0 = ops._ver(t)                         # reading t._version would raise: a content checksum stands in
        assert ops._ver(t) == k0
        t.mul_(2)                                # in-place writes ARE possible inside inference mode:
        assert ops._ver(t) != k0                 # the cache key must move with the contents
        f = torch.ones(5, 3)
        kf = ops._ver(f)
        f[2, 1] = 1.0000001
        assert ops._ver(f) != kf
    u = torch.zeros(3)
    v0 = ops._ver(u)
    u.add_(1)
    assert ops._ver(u) == v0 + 1
    u.data.add_(1)                               # the documented hole: .data writes do not move the counter
    assert ops._ver(u) == v0 + 1


def test_unsupported_configurations_fail_loudly():
    with pytest.raises(NotImplementedError):
        ops.mlp_linears(torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(),
                                            torch.nn.Linear(8, 4096)))
    with pytest.raises(NotImplementedError):
        ops.mlp_linears(torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Linear(8, 4096)))
    lin = ops.mlp_linears(DenseNet([6, 8, 16, 4096], torch.nn.ReLU))
    assert [tuple(l.weight.shape) for l in lin] == [(8, 6), (16, 8), (4096, 16)]
    lin = ops.mlp_linears(torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.ReLU(),
                                              torch.nn.Linear(8, 4096)))
    assert len(lin) == 2
    # round 6: a kernel network outside the Linear / ReLU chain is no longer refused by the MODULE - it is routed to nn(pseudo) + the
    # native per-edge-weight operator (NNConv_old._propagate_general_nn; GPU test: tests/test_gpu_general_nn.py); the fused kernels'
    # own parser above still refuses it
    tanh = gp.NNConv_old(64, 64, torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(), torch.nn.Linear(8, 4096)))
    assert not tanh._nn_is_linear_relu_chain()
    assert gp.NNConv_old(64, 64, DenseNet([6, 8, 16, 4096], torch.nn.ReLU))._nn_is_linear_relu_chain()


def test_synthetic_graphs_have_reference_shape():
    from graph_pde_amd import synth
    ei, ea, n = synth.darcy_graph(16, 0.15)
    assert n == 256 and ei.shape == (2, 4692) and ea.shape == (4692, 6)
    assert bool((ei[0, 1:] >= ei[0, :-1]).all())                              # sorted by source
    assert int((ei[0] == ei[1]).sum()) == n                                   # self-loops included
    # symmetric (exact integer arithmetic)
    fw = set(map(tuple, ei.t().tolist()))
    assert all((j, i) in fw for (i, j) in list(fw)[:500])
    assert synth.lattice_radius_graph(61, 0.10).shape[1] == 386221
    g = synth.burgers_multipole_graphs(8192)
    assert [x[0].shape[1] for x in g][:4] == [16384, 24570, 12282, 6138]
    assert sum(x[0].shape[1] for x in g) == 65462


def test_hidden_cache_key_and_policy_bookkeeping(monkeypatch):
    """hidden_cache.py host logic: the key pins memory + version + parameter versions;
    `off` never builds; `auto` goes direct on the first sight of a key and wants to build on the
    second -- which, without a GPU, must fail loudly inside the native path, not compute on the CPU."""
    from graph_pde_amd import hidden_cache, ops

    class FakeCsr:
        n_edges, n_nodes = 3, 5
    class FakePm:
        dims = (6, 8, 4096)
    conv = gp.NNConv_old(64, 64, DenseNet([6, 8, 4096], torch.nn.ReLU), aggr="mean")
    lin = ops.mlp_linears(conv.nn)
    w, b = [l.weight for l in lin], [l.bias for l in lin]
    ea, csr, pm = torch.randn(3, 6), FakeCsr(), FakePm()
    k1 = hidden_cache._key(ea, csr, w[:-1] + b[:-1], "f16split")
    assert k1 == hidden_cache._key(ea, csr, w[:-1] + b[:-1], "f16split")
    assert k1 != hidden_cache._key(ea[1:], csr, w[:-1] + b[:-1], "f16split")          # other view
    assert k1 != hidden_cache._key(ea, csr, w[:-1] + b[:-1], "f32")
    ea.mul_(1.0)
    assert k1 != hidden_cache._key(ea, csr, w[:-1] + b[:-1], "f16split")              # version counter
    k2 = hidden_cache._key(ea, csr, w[:-1] + b[:-1], "f16split")
    with torch.no_grad():
        w[0].add_(0.0)
    assert k2 != hidden_cache._key(ea, csr, w[:-1] + b[:-1], "f16split")              # weights touched
    with torch.no_grad():
        assert k2[-1] is True and hidden_cache._key(ea, csr, w[:-1] + b[:-1], "f16split")[-1] is False

    hidden_cache.clear()
    assert hidden_cache.lookup(conv, ea, csr, pm, w, b, mode="off") is None
    assert hidden_cache.lookup(conv, ea, csr, pm, w, b, mode="auto") is None           # first sight: direct
    assert hidden_cache.stats["direct"] == 2 and hidden_cache.stats["builds"] == 0
    with pytest.raises(RuntimeError):                                                  # second: wants H -> native
        hidden_cache.lookup(conv, ea, csr, pm, w, b, mode="auto")
    monkeypatch.setattr(hidden_cache, "BUDGET_BYTES", 16)
    assert hidden_cache.lookup(conv, ea, csr, pm, w, b, mode="on") is None             # over budget: direct
    buf = io.BytesIO()
    torch.save(conv, buf)                                                              # nothing rides on the module


def test_virtual_hidden_node_policy_host_logic(monkeypatch):
    """hidden_cache.lookup_deferred, on CPU tensors - the virtual-H node holds no numbers, so the policy runs
    without a GPU: a module seen repeating whose H is over budget gets ONE node per forward, shared by its applications until
    its backward ran; a once-per-forward module drops out; kernel MLPs outside the deferred form, `off`, no-grad calls and a
    differentiated edge_attr never get one."""
    from graph_pde_amd import hidden_cache, ops

    class FakeCsr:
        n_edges, n_nodes = 3000, 50
        rowptr_host = torch.arange(0, 3060, 60, dtype=torch.int32)
    csr = FakeCsr()
    conv = gp.NNConv_old(64, 64, DenseNet([6, 256, 256, 4096], torch.nn.ReLU), aggr="mean")
    lin = ops.mlp_linears(conv.nn)
    w, b = [l.weight for l in lin], [l.bias for l in lin]
    pm = type("Pm", (), {"dims": (6, 256, 256, 4096)})()
    ea = torch.randn(3000, 6)
    assert ops.deferred_supported([6, 256, 256, 4096]) and not ops.deferred_supported([6, 64, 128, 4096])
    assert ops.deferred_layers_padded(1) == 4 and ops.deferred_layers_padded(5) == 6 and ops.deferred_layers_padded(6) == 6
    monkeypatch.setattr(hidden_cache, "BUDGET_BYTES", 0)             # H never fits
    monkeypatch.setattr(hidden_cache, "MODE", "auto")
    monkeypatch.setattr(hidden_cache, "DEFER_MODE", "auto")
    hidden_cache.clear()
    assert hidden_cache.defer_possible(ea, pm, w, b, "mean") and not hidden_cache.defer_possible(ea, pm, w, b, "max")

    def call():
        assert hidden_cache.lookup(conv, ea, csr, pm, w, b, allow_partial=True) is None
        return hidden_cache.lookup_deferred(conv, ea, csr, pm, w, b, "mean")
    assert call() is None                                             # first sight of the key: the plain operator
    d2, d3 = call(), call()
    assert d2 is not None and d3[0] is d2[0] and d3[1] is d2[1]      # second and third application share one node
    assert d2[0].requires_grad and tuple(d2[0].shape) == (1,) and d2[1].valid and len(d2[1].stash) == 0
    assert hidden_cache.stats["deferred_builds"] == 1 and hidden_cache.stats["deferred_hits"] == 1
    d2[1].valid = False                                               # its backward ran (DeferredHiddenFunction.backward)
    d4 = call()
    assert d4 is not None and d4[1] is not d2[1]                      # next forward: a fresh node from the FIRST call on
    assert call()[1] is d4[1]
    d4[1].stash[1] = ("x", "g")                                       # a backward that ended without its deferred pass
    assert call()[1] is d4[1] and len(d4[1].stash) == 0 and hidden_cache.stats["deferred_stale_dropped"] == 1
    d4[1].valid = False
    with torch.no_grad():
        w[0].mul_(1.0)                                                # optimizer step; this forward applies the module once ...
    assert call() is not None
    hidden_cache._entries[conv].dtoken.valid = False
    with torch.no_grad():
        w[0].mul_(1.0)
    assert call() is None and not hidden_cache._entries[conv].repeats  # ... so the next forward's call is a stranger again
    # never: switched off, no gradient wanted, differentiated attributes, MLP outside the deferred form
    hidden_cache.clear()
    monkeypatch.setattr(hidden_cache, "DEFER_MODE", "off")
    call(); assert call() is None
    monkeypatch.setattr(hidden_cache, "DEFER_MODE", "auto")
    hidden_cache.clear()
    with torch.no_grad():
        call(); assert call() is None
    ea_g = ea.clone().requires_grad_(True)
    assert not hidden_cache.defer_possible(ea_g, pm, w, b, "mean")
    small = gp.NNConv_old(64, 64, DenseNet([6, 64, 128, 4096], torch.nn.ReLU), aggr="mean")
    ls = ops.mlp_linears(small.nn)
    pm_s = type("Pm", (), {"dims": (6, 64, 128, 4096)})()
    assert not hidden_cache.defer_possible(ea, pm_s, [l.weight for l in ls], [l.bias for l in ls], "mean")
    hidden_cache.clear()


def test_node_attr_recipe_is_the_reference_edge_attr():
    from graph_pde_amd import synth
    """NodeAttr.darcy(pos, a).materialize(edge_index) == [pos_src, pos_dst, a_src, a_dst]
    (SquareMeshGenerator.attributes, graph-neural-operator/utilities.py:274-277) -- host side of row f3."""
    ei = synth.lattice_radius_graph(8, 0.3)
    pos = synth.lattice_positions(8)
    a = synth.darcy_coefficient(8, 1)
    na = gp.NodeAttr.darcy(pos, a)
    assert na.k0 == 6 and na.sel == [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (1, 2)]
    assert torch.equal(na.materialize(ei), synth.darcy_edge_attr(ei, pos, a))
    with pytest.raises(ValueError):
        gp.NodeAttr(torch.zeros(4, 3), [(2, 0)])
    with pytest.raises(NotImplementedError):
        gp.NodeAttr(torch.zeros(4, 3), [(0, 0)] * 8)
    conv = gp.NNConv_old(64, 64, DenseNet([6, 8, 16, 4096], torch.nn.ReLU), aggr="mean")
    with pytest.raises(RuntimeError, match="no CPU"):       # still no CPU execution path
        conv(torch.randn(64, 64), ei, na)


def test_partial_hidden_cache_split_point(monkeypatch):
    """hidden_cache.py host logic for graphs whose H exceeds the budget: the split node hn is the largest
    multiple of 64 whose in-edges fit, partial caching needs a gradient-free caller and at least an eighth
    of the nodes, and the native builder is asked for exactly those nodes."""
    from graph_pde_amd import hidden_cache, ops

    class FakeCsr:
        n_nodes, deg = 1024, 10
        n_edges = n_nodes * deg
        rowptr_host = torch.arange(0, (n_nodes + 1) * deg, deg, dtype=torch.int32)
    class FakePm:
        dims = (6, 8, 100, 4096)                 # K2P = 128 -> 512 B per row
    conv = gp.NNConv_old(64, 64, DenseNet([6, 8, 100, 4096], torch.nn.ReLU), aggr="mean")
    lin = ops.mlp_linears(conv.nn)
    w, b = [l.weight for l in lin], [l.bias for l in lin]
    ea, csr, pm = torch.randn(FakeCsr.n_edges, 6), FakeCsr(), FakePm()
    calls = []

    def fake_hidden(csr_, attr, pm_, ws_, bs_, precision, n_nodes_limit=None):
        calls.append(n_nodes_limit)
        return torch.zeros(int(csr_.rowptr_host[n_nodes_limit]), 128), None
    monkeypatch.setattr(ops, "hidden_forward_raw", fake_hidden)
    row = 128 * 4
    monkeypatch.setattr(hidden_cache, "BUDGET_BYTES", 500 * 10 * row)      # 500 nodes' worth of rows
    hidden_cache.clear()
    with torch.no_grad():
        hit = hidden_cache.lookup(conv, ea, csr, pm, w, b, mode="on", allow_partial=True)
    assert hit is not None and hit[2] == 448 and calls == [448]            # 500 -> 448 = 7 * 64
    assert hit[0].shape[0] == 448 * 10
    # a caller that needs gradients never gets a partial H
    assert hidden_cache.lookup(conv, ea, csr, pm, w, b, mode="on", allow_partial=False) is None
    # less than an eighth of the nodes fits: not worth it
    monkeypatch.setattr(hidden_cache, "BUDGET_BYTES", 100 * 10 * row)
    hidden_cache.clear()
    with torch.no_grad():
        assert hidden_cache.lookup(conv, ea, csr, pm, w, b, mode="on", allow_partial=True) is None
    monkeypatch.setattr(hidden_cache, "PARTIAL", False)
    monkeypatch.setattr(hidden_cache, "BUDGET_BYTES", 500 * 10 * row)
    with torch.no_grad():
        assert hidden_cache.lookup(conv, ea, csr, pm, w, b, mode="on", allow_partial=True) is None


def test_partial_hidden_size_is_kept_while_it_fits(monkeypatch):
    """Round 6: the budget follows the free memory of the moment; a rebuilt partial H (the weights changed: every training step) keeps
    the PREVIOUS build's node count while that still fits and lies within 10 % of what the budget allows now - a few MiB more could
    not reuse the block the allocator just got back (on the 241^2 graph: a second 230 GiB request).  A smaller budget shrinks it, a
    budget more than 10 % larger grows it."""
    from graph_pde_amd import hidden_cache, ops

    class FakeCsr:
        n_nodes, deg = 1024, 10
        n_edges = n_nodes * deg
        rowptr_host = torch.arange(0, (n_nodes + 1) * deg, deg, dtype=torch.int32)
    class FakePm:
        dims = (6, 8, 100, 4096)
    conv = gp.NNConv_old(64, 64, DenseNet([6, 8, 100, 4096], torch.nn.ReLU), aggr="mean")
    lin = ops.mlp_linears(conv.nn)
    w, b = [l.weight for l in lin], [l.bias for l in lin]
    csr, pm = FakeCsr(), FakePm()
    calls = []

    def fake_hidden(csr_, attr, pm_, ws_, bs_, precision, n_nodes_limit=None):
        calls.append(n_nodes_limit)
        return torch.zeros(int(csr_.rowptr_host[n_nodes_limit]), 128), None
    monkeypatch.setattr(ops, "hidden_forward_raw", fake_hidden)
    row = 128 * 4
    hidden_cache.clear()

    def build(nodes_worth):
        monkeypatch.setattr(hidden_cache, "BUDGET_BYTES", nodes_worth * 10 * row)
        ea = torch.randn(FakeCsr.n_edges, 6)                    # a new edge_attr tensor: a new key, H is rebuilt
        with torch.no_grad():
            return hidden_cache.lookup(conv, ea, csr, pm, w, b, mode="on", allow_partial=True)[2]
    assert build(600) == 576                                     # 9 * 64
    assert build(660) == 576                                     # 640 would fit now - within 10 %: the previous size stays
    assert build(590) == 576                                     # still fits
    assert build(520) == 512                                     # no longer fits: shrinks
    assert build(560) == 512                                     # 10 % hysteresis around the new size
    assert build(800) == 768                                     # far more room: grows
    assert calls == [576, 576, 576, 512, 512, 768]


def test_in_place_gradient_sums_over_a_modules_applications_host_logic(monkeypatch):
    """autograd.WeConvFunction / SharedParamFunction (round 6) with the two native calls replaced by torch CPU arithmetic of the same
    contract: the first application of a backward pass returns dL/dW_e, grad_root, grad_bias and leaves them on the token, the others
    ADD to those tensors (`acc`) and return None - autograd must end up with the sums it would have formed itself, a second pass over
    a retained graph must start new tensors, and a second use of root in the loss must not lose the in-place additions."""
    from graph_pde_amd import autograd as gpa

    class FakeCsr:
        def __init__(self, src, dst, n):
            self.src, self.dst, self.n_nodes, self.n_edges = src, dst, n, int(src.numel())
    n, e = 5, 12
    g = torch.Generator().manual_seed(0)
    csr = FakeCsr(torch.randint(0, n, (e,), generator=g), torch.randint(0, n, (e,), generator=g), n)
    n_acc = {"calls": 0, "acc": 0}

    def fwd(x, csr_, we, root, bias, aggr, **kw):
        m = torch.bmm(x[csr_.src].unsqueeze(1), we.view(-1, 64, 64)).squeeze(1)
        out = torch.zeros(csr_.n_nodes, 64).index_add_(0, csr_.dst, m)
        if root is not None:
            out = out + x @ root
        return out if bias is None else out + bias

    def bwd(x, csr_, we, root, aggr, grad_out, need_root=True, need_bias=True, acc=None):
        n_acc["calls"] += 1
        gt = grad_out[csr_.dst]
        gwe = (x[csr_.src].unsqueeze(2) * gt.unsqueeze(1)).reshape(-1, 4096)
        gx = torch.zeros_like(x).index_add_(0, csr_.src, torch.bmm(we.view(-1, 64, 64), gt.unsqueeze(2)).squeeze(2))
        groot = x.t() @ grad_out if (need_root and root is not None) else None
        if root is not None:
            gx = gx + grad_out @ root.t()
        gbias = grad_out.sum(0) if need_bias else None
        if acc is not None:
            n_acc["acc"] += 1
            acc[0].add_(gwe)
            if groot is not None:
                acc[1].add_(groot)
            if gbias is not None:
                acc[2].add_(gbias)
            return gx, acc[0], acc[1], acc[2]
        return gx, gwe, groot, gbias
    monkeypatch.setattr(ops, "nnconv_forward_edgeweights_raw", fwd)
    monkeypatch.setattr(ops, "nnconv_backward_edgeweights_raw", bwd)

    def run(shared, passes=1, reg=False):
        monkeypatch.setattr(gpa, "ACCUMULATE_GRAD_HIDDEN", shared)
        gg = torch.Generator().manual_seed(1)
        x = torch.randn(n, 64, generator=gg).requires_grad_(True)
        we = (0.1 * torch.randn(e, 4096, generator=gg)).requires_grad_(True)
        root = (0.1 * torch.randn(64, 64, generator=gg)).requires_grad_(True)
        bias = torch.randn(64, generator=gg).requires_grad_(True)
        w = torch.randn(n, 64, generator=gg)
        tok = gpa.HiddenToken() if shared else None
        we_node = we * 1.0                                           # a non-leaf node shared by the applications, as W_e is
        if shared:
            tok.side_in = (gpa.SharedParamFunction.apply(root, tok, 0), gpa.SharedParamFunction.apply(bias, tok, 1), root, bias)
            r_, b_ = tok.side_in[0], tok.side_in[1]
        else:
            r_, b_ = root, bias
        h = x
        for _ in range(3):
            h = torch.tanh(gpa.WeConvFunction.apply(h, we_node, csr, r_, b_, "add", tok))
        loss = (h * w).sum() + (0.5 * root.square().sum() if reg else 0.0)
        for k in range(passes):
            loss.backward(retain_graph=k + 1 < passes)
        return [t.grad.clone() for t in (x, we, root, bias)]
    c0 = dict(n_acc)
    ref = run(False)
    assert n_acc["acc"] == c0["acc"]
    got = run(True)
    assert n_acc["acc"] - c0["acc"] == 2                              # the first application writes, the other two add
    for a, b in zip(got, ref):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    for a, b in zip(run(True, passes=2), run(False, passes=2)):       # a second pass starts its own tensors
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    for a, b in zip(run(True, reg=True), run(False, reg=True)):       # root used by the loss as well
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)


def test_bench_flop_accounting_matches_survey():
    """bench.py's roofline inputs: the reference formulation's FLOPs per edge and what the
    kernels execute."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.algorithmic_flops_per_edge([6, 1024, 1024, 4096]) == 10506304
    assert bench.algorithmic_flops_per_edge([6, 256, 256, 4096]) == 2239552
    dims = [6, 1024, 1024, 4096]
    f32, f16 = bench.executed_flops_per_edge(dims, "gpde_fused_kernel", False)
    assert (f32, f16) == (2 * 8 * 1024 * 8 + 2 * 64 * 1024 + 2 * 1024 * 1024, 0)
    f32, f16 = bench.executed_flops_per_edge(dims, "gpde_fused_f16v3_kernel", True)
    assert f32 == 0 and f16 == 3 * 2 * 1024 * 1024 + 2 * 2 * 16 * 1024 * 16 + 3 * 2 * 64 * 1024 == 7733248
    f32, f16 = bench.executed_flops_per_edge(dims, "gpde_fused_f16v3_kernel", False)
    assert f32 == 131072 and f16 == 7340032
    # the one-wave-per-SIMD kernel regenerates H1 per 128-column tile: half of the 8-wave kernel's
    f32, f16 = bench.executed_flops_per_edge(dims, "gpde_fused_f16v6_kernel", True)
    assert f32 == 0 and f16 == 3 * 2 * 1024 * 1024 + 2 * 2 * 16 * 1024 * 8 + 3 * 2 * 64 * 1024 == 7208960


def test_bench_refuses_a_traffic_record_of_another_kernel(tmp_path, monkeypatch):
    """roofline.traffic must describe the kernel that was timed: the PMC record is keyed by kernel symbol."""
    import importlib.util, json, os
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    f = tmp_path / "traffic.json"
    f.write_text(json.dumps({"config": "g241", "kernel_width": 1024, "source": "test",
                             "kernels": {"gpde_fused_f16v3_kernel": {"hbm_bytes_per_launch": 1.0e11}}}))
    monkeypatch.setattr(bench, "TRAFFIC_FILE", str(f))
    rec, why = bench.traffic_record("g241", 1024, "gpde_fused_f16v6_kernel")
    assert rec is None and "gpde_fused_f16v6_kernel" in why
    rec, why = bench.traffic_record("g241", 1024, "gpde_fused_f16v3_kernel")
    assert rec["hbm_bytes_per_launch"] == 1.0e11
    rec, why = bench.traffic_record("g121", 1024, "gpde_fused_f16v3_kernel")
    assert rec is None


def test_module_derives_message_passing_and_accepts_new_pyg_root_key():
    """The reference class derives PyG's MessagePassing and calls self.propagate (nn_conv.py:197, 242, 271); newer
    PyG state dicts name the root weight `lin.weight [out, in]`."""
    import sys
    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "graph-pde_amd", "shims")
    sys.path.insert(0, shim)
    try:
        from torch_geometric.nn.conv import MessagePassing
    finally:
        sys.path.remove(shim)
    conv = gp.NNConv_old(64, 64, DenseNet([6, 32, 64, 4096], torch.nn.ReLU), aggr="mean")
    assert isinstance(conv, MessagePassing) and isinstance(gp.NNConv(64, 64, torch.nn.Linear(6, 4096)), MessagePassing)
    assert conv.aggr == "mean" and conv.flow == "source_to_target"
    assert conv.__message_args__ == ["x_j", "pseudo"] and conv.__update_args__ == ["x"]
    assert callable(conv.propagate)
    with pytest.raises(TypeError):
        conv.propagate(torch.zeros(2, 0, dtype=torch.int64), x=torch.zeros(1, 64))          # pseudo= missing
    sd = conv.state_dict()
    new = {k: v.clone() for k, v in sd.items() if k != "root"}
    new["lin.weight"] = sd["root"].t().clone()
    other = gp.NNConv_old(64, 64, DenseNet([6, 32, 64, 4096], torch.nn.ReLU), aggr="mean")
    other.load_state_dict(new)
    assert torch.equal(other.root, conv.root) and torch.equal(other.bias, conv.bias)
    gp.NNConv_old(64, 64, torch.nn.Linear(6, 4096), aggr="max")                              # constructible (nn_conv.py:222-224)
    with pytest.raises(ValueError):
        gp.NNConv_old(64, 64, torch.nn.Linear(6, 4096), aggr="median")


def test_edge_weight_cache_policy_is_opt_in_and_bounded(monkeypatch):
    """hidden_cache.edge_weights_qualify: off by default for plain module calls; the explicit grouped API
    and aggr='max' do not consult the switch; low in-degree or small calls only; byte budget."""
    from graph_pde_amd import hidden_cache

    class C:
        def __init__(self, n, e):
            self.n_nodes, self.n_edges = n, e
    assert hidden_cache.WE_MODE in ("off", "auto")
    monkeypatch.setattr(hidden_cache, "WE_MODE", "off")
    assert not hidden_cache.edge_weights_qualify(C(8192, 16384))
    assert hidden_cache.edge_weights_qualify(C(8192, 16384), explicit=True)              # in-degree 2: Burgers level 0
    assert hidden_cache.edge_weights_qualify(C(400, 7732), explicit=True)                # <= 8192 edges whatever the degree
    assert not hidden_cache.edge_weights_qualify(C(2400, 131904), explicit=True)         # Darcy inner level 0: Z path
    assert not hidden_cache.edge_weights_qualify(C(10, 0), explicit=True, force=True)
    monkeypatch.setattr(hidden_cache, "WE_MODE", "auto")
    assert hidden_cache.edge_weights_qualify(C(8192, 16384))
    monkeypatch.setattr(hidden_cache, "WE_BUDGET_BYTES", 16384 * 1000)
    assert not hidden_cache.edge_weights_qualify(C(8192, 16384)) and hidden_cache.edge_weights_qualify(C(500, 1000))
    monkeypatch.setattr(hidden_cache, "BUDGET_BYTES", 16384 * 2000)
    assert hidden_cache.edge_weights_qualify(C(10, 2000), force=True) and not hidden_cache.edge_weights_qualify(C(10, 2001), force=True)


def test_nnconv_group_argument_checks_without_a_gpu():
    conv = gp.NNConv(64, 64, DenseNet([4, 16, 16, 4096], torch.nn.ReLU), aggr="mean")
    x, ei, ea = torch.randn(5, 64), torch.tensor([[0, 1], [1, 2]]), torch.randn(2, 4)
    with pytest.raises(ValueError, match="activation"):
        gp.nnconv_group([(conv, x, ei, ea, None, "tanh")])
    assert gp.nnconv_group([]) == []


def test_attr_slot_order_cache_and_keep_z_policy_host_logic(monkeypatch):
    """ops.attr_in_slot_order (layout-only copy of edge_attr in CSR slot order, cached per graph + edge_attr version) and
    ops.z_buffer (when the keep-Z training pair is used) - pure host logic, exercised here on CPU tensors."""
    e, n = 40000, 50
    g = torch.Generator().manual_seed(0)
    perm = torch.randperm(e, generator=g).to(torch.int32)
    dst = torch.sort(torch.randint(0, n, (e,), generator=g)).values.to(torch.int32)
    rowptr = torch.searchsorted(dst.long(), torch.arange(n + 1)).to(torch.int32)
    csr = ops.Csr(n, e, rowptr, torch.zeros(e, dtype=torch.int32), dst, perm)
    ea = torch.randn(e, 6, generator=g)
    monkeypatch.setattr(ops, "ATTR_SLOT_ORDER", True)
    # the gather itself is a native kernel (gpde_gather_rows, checked on the GPU in tests/test_gpu_v6.py); here a stand-in,
    # the subject being the cache policy around it
    monkeypatch.setattr(ops, "gather_rows", lambda rows, pm_: rows[pm_.long()])
    s1, p1 = ops.attr_in_slot_order(csr, ea)
    assert torch.equal(s1, ea[perm.long()]) and torch.equal(p1, torch.arange(e, dtype=torch.int32))
    s2, _ = ops.attr_in_slot_order(csr, ea)
    assert s2 is s1                                               # cached
    ea.mul_(2.0)                                                  # in place: version moves, a new copy is gathered
    s3, _ = ops.attr_in_slot_order(csr, ea)
    assert s3 is not s1 and torch.equal(s3, ea[perm.long()])
    monkeypatch.setattr(ops, "ATTR_SLOT_ORDER", False)
    s4, p4 = ops.attr_in_slot_order(csr, ea)
    assert s4 is ea and p4 is perm
    monkeypatch.setattr(ops, "ATTR_SLOT_ORDER", True)
    small = ops.Csr(n, 100, rowptr, torch.zeros(100, dtype=torch.int32), dst[:100], perm[:100])
    assert ops.attr_in_slot_order(small, ea[:100])[0].data_ptr() == ea[:100].data_ptr()      # below 32768 edges: not worth a copy
    ident = ops.Csr(n, e, rowptr, torch.zeros(e, dtype=torch.int32), dst, torch.arange(e, dtype=torch.int32))
    assert ops.attr_in_slot_order(ident, ea)[0] is ea            # a graph from radius_csr is in slot order already
    # keep-Z: only with >= 32 edges per node and within the byte budget
    dims = [6, 1024, 1024, 4096]
    monkeypatch.setattr(ops, "SAVE_Z_BYTES", 16 << 30)
    z = ops.z_buffer(csr, dims, "cpu")
    assert z is not None and tuple(z.shape) == (n, 64 * 1024) and float(z.abs().sum()) == 0.0
    assert ops.z_buffer(ops.Csr(2000, e, rowptr, perm, dst, perm), dims, "cpu") is None          # 20 edges per node
    monkeypatch.setattr(ops, "SAVE_Z_BYTES", n * 64 * 1024 * 4 - 1)
    assert ops.z_buffer(csr, dims, "cpu") is None
    monkeypatch.setattr(ops, "SAVE_Z_BYTES", 0)
    assert ops.z_buffer(csr, dims, "cpu") is None


def test_hidden_cache_budget_follows_the_device_unless_pinned(monkeypatch):
    """hidden_cache.budget_bytes: GPDE_HIDDEN_CACHE_GB pins it; unset -> min(70 % of the device's memory, free now (driver +
    torch's cached free blocks) + what is about to be released - the 48 GB reserve), never negative; CPU tensors (tests) get
    the old fixed 32 GB."""
    from graph_pde_amd import hidden_cache
    gb = 1 << 30
    monkeypatch.setattr(hidden_cache, "BUDGET_BYTES", 5 * gb)
    assert hidden_cache.budget_bytes("cuda:0") == 5 * gb and hidden_cache.budget_bytes(None) == 5 * gb
    monkeypatch.setattr(hidden_cache, "BUDGET_BYTES", None)
    assert hidden_cache.budget_bytes(None) == 32 * gb and hidden_cache.budget_bytes(torch.device("cpu")) == 32 * gb
    monkeypatch.setattr(torch.cuda, "memory_reserved", lambda dev=None: 0)
    monkeypatch.setattr(torch.cuda, "memory_allocated", lambda dev=None: 0)
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda dev=None: (280 * gb, 288 * gb))
    assert hidden_cache.budget_bytes("cuda:0") == int(0.7 * 288 * gb)                  # idle device: the fraction binds
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda dev=None: (100 * gb, 288 * gb))
    assert hidden_cache.budget_bytes("cuda:0") == 52 * gb                               # busy device: free - reserve
    assert hidden_cache.budget_bytes("cuda:0", releasing=20 * gb) == 72 * gb            # the H being replaced counts as free
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda dev=None: (10 * gb, 288 * gb))
    assert hidden_cache.budget_bytes("cuda:0") == 0
    # blocks torch's allocator holds free (last step's Z buffers and workspaces) count as free
    monkeypatch.setattr(torch.cuda, "memory_reserved", lambda dev=None: 150 * gb)
    monkeypatch.setattr(torch.cuda, "memory_allocated", lambda dev=None: 30 * gb)
    assert hidden_cache.budget_bytes("cuda:0") == (10 + 120 - 48) * gb


def test_workspace_allocation_drops_the_caches_once_when_the_device_is_full(monkeypatch):
    """ops._alloc_ws: the hidden-activation cache may hold most of the HBM (budget sized to the device); a workspace that
    does not fit makes it let go and is retried ONCE; with nothing to release the error is the caller's."""
    from graph_pde_amd import hidden_cache

    class M(torch.nn.Module):
        pass
    m = M()
    ent = hidden_cache._Entry()
    ent.hidden, ent.key, ent.repeats = torch.zeros(4), ("k",), True
    hidden_cache._entries[m] = ent
    real_empty, calls = torch.empty, {"n": 0}

    def flaky_empty(*a, **k):
        calls["n"] += 1
        if calls["n"] == 1:
            raise torch.OutOfMemoryError("HIP out of memory (test)")
        return real_empty(*a, **k)
    monkeypatch.setattr(torch, "empty", flaky_empty)
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda: None)
    ws = ops._alloc_ws(100, "cpu")
    assert ws.numel() == 100 and calls["n"] == 2
    assert ent.hidden is None and ent.key is None and ent.repeats       # tensors gone, what was learnt about the module stays
    calls["n"] = 0                                                        # nothing cached any more: the error propagates
    with pytest.raises(torch.OutOfMemoryError):
        ops._alloc_ws(100, "cpu")
    hidden_cache.clear()


def test_backward_workspace_is_one_chunk_or_the_default_never_in_between(monkeypatch):
    """ops.bwd_workspace_bytes: the library's default size, or its one-chunk size when that is below GPDE_BWD_WS_FRACTION of
    the free device memory - partial growth fragmented torch's cache in G241 training."""
    from graph_pde_amd import _lib, ops
    lib = _lib.lib()
    dims_c = _lib.dims_array([6, 1024, 1024, 4096])
    n, e = 14641, 5931137                                            # the s=121 graph
    small = int(lib.gpde_nnconv_bwd_workspace_bytes(n, e, 3, dims_c))
    one = int(lib.gpde_nnconv_bwd_workspace_bytes_one_chunk(n, e, 3, dims_c))
    assert one > 4 * small
    monkeypatch.setattr(ops, "BWD_WS_FRACTION", 0.6)
    monkeypatch.setattr(ops, "device_free_bytes", lambda dev: (int(one / 0.6) + (1 << 20), 288 << 30))      # just enough
    assert ops.bwd_workspace_bytes(lib, n, e, 3, dims_c, None) == one
    monkeypatch.setattr(ops, "device_free_bytes", lambda dev: (int(one / 0.6) - (1 << 30), 288 << 30))      # 1 GB short
    assert ops.bwd_workspace_bytes(lib, n, e, 3, dims_c, None) == small
    monkeypatch.setattr(ops, "device_free_bytes", lambda dev: (288 << 30, 288 << 30))
    monkeypatch.setattr(ops, "BWD_WS_FRACTION", 0.0)                                                       # switched off
    assert ops.bwd_workspace_bytes(lib, n, e, 3, dims_c, None) == small
    monkeypatch.setattr(ops, "BWD_WS_FRACTION", 0.6)
    # a small graph: the default already is one chunk
    assert ops.bwd_workspace_bytes(lib, 200, 9000, 3, _lib.dims_array([6, 256, 256, 4096]), None) == \
        int(lib.gpde_nnconv_bwd_workspace_bytes(200, 9000, 3, _lib.dims_array([6, 256, 256, 4096])))
    # the headline graph: one chunk (2.4 TB) never fits - default
    g_small = int(lib.gpde_nnconv_bwd_workspace_bytes(58081, 95539625, 3, dims_c))
    assert ops.bwd_workspace_bytes(lib, 58081, 95539625, 3, dims_c, None) == g_small


def test_keep_hidden_policy_and_its_workspace_host_logic(monkeypatch):
    """ops.keep_hidden (round 5: when a training forward keeps the last hidden activations for its own backward) and the
    one-chunk workspace of that backward, which leaves the kept tensor's bytes out - pure host logic."""
    from graph_pde_amd import _lib, ops
    n, e = 14641, 5931137                                            # the s=121 graph: 24.3 GB of H_2 at k2 = 1024
    z32 = torch.zeros(1, dtype=torch.int32)
    csr = ops.Csr(n, e, z32, z32, z32, z32)
    dims = [6, 1024, 1024, 4096]
    monkeypatch.setattr(ops, "DEFAULT_PRECISION", "f16split")
    monkeypatch.setattr(ops, "SAVE_H_BYTES", 32 << 30)
    monkeypatch.setattr(ops, "SAVE_Z_RESERVE_BYTES", 48 << 30)
    monkeypatch.setattr(ops, "device_free_bytes", lambda dev: (250 << 30, 288 << 30))
    assert ops.keep_hidden(csr, dims, None)
    assert not ops.keep_hidden(csr, [6, 1024, 4096], None)                       # 2-Linear: no store kernel of that form
    assert not ops.keep_hidden(csr, [6, 128, 1024, 4096], None)                  # < 8 first-layer chunks: the 8-wave kernel's shape
    assert not ops.keep_hidden(csr, [9, 1024, 1024, 4096], None)                 # > 8 attribute slots
    assert not ops.keep_hidden(ops.Csr(n, 200000, z32, z32, z32, z32), dims, None)            # two launches instead of one do not pay
    assert not ops.keep_hidden(ops.Csr(400000, e, z32, z32, z32, z32), dims, None)            # mean in-degree < 32: §3e's path
    assert not ops.keep_hidden(ops.Csr(58081, 95539625, z32, z32, z32, z32), dims, None)      # the headline graph: 391 GB
    monkeypatch.setattr(ops, "device_free_bytes", lambda dev: (70 << 30, 288 << 30))          # 70 - 24.3 < 48 GB reserve
    assert not ops.keep_hidden(csr, dims, None)
    monkeypatch.setattr(ops, "device_free_bytes", lambda dev: (250 << 30, 288 << 30))
    monkeypatch.setattr(ops, "SAVE_H_BYTES", 0)
    assert not ops.keep_hidden(csr, dims, None)
    monkeypatch.setattr(ops, "DEFAULT_PRECISION", "f32")
    monkeypatch.setattr(ops, "SAVE_H_BYTES", 32 << 30)
    assert not ops.keep_hidden(csr, dims, None)
    # the backward's one-chunk workspace with H given: smaller by exactly the tensor
    lib = _lib.lib()
    dims_c = _lib.dims_array(dims)
    one = int(lib.gpde_nnconv_bwd_workspace_bytes_one_chunk(n, e, 3, dims_c))
    hb = e * 1024 * 4
    monkeypatch.setattr(ops, "BWD_WS_FRACTION", 0.6)
    monkeypatch.setattr(ops, "device_free_bytes", lambda dev: (int((one - hb) / 0.6) + (1 << 20), 288 << 30))
    assert ops.bwd_workspace_bytes(lib, n, e, 3, dims_c, None, hb) == one - hb          # fits only because H is left out
    assert ops.bwd_workspace_bytes(lib, n, e, 3, dims_c, None) == int(lib.gpde_nnconv_bwd_workspace_bytes(n, e, 3, dims_c))


def test_token_of_hands_out_the_shared_h_token_only_where_h_has_one_kind_of_consumer(monkeypatch):
    """hidden_cache.token_of: the applications of a module sum their dL/dH in place on the token of its cached full H
    (autograd.NNConvHiddenFunction.backward) - not for another tensor, not after the H node's backward, and not on a graph where the
    per-edge weight form may run (its backward is a second consumer of H)."""
    from graph_pde_amd import hidden_cache, ops
    from graph_pde_amd.autograd import HiddenToken
    m = torch.nn.Linear(2, 2)
    z32 = torch.zeros(1, dtype=torch.int32)
    dense = ops.Csr(1681, 75000, z32, z32, z32, z32)               # mean in-degree 45: the re-associated path
    sparse = ops.Csr(8192, 16384, z32, z32, z32, z32)              # 2 in-edges per node: qualifies for W_e
    small = ops.Csr(100, 3200, z32, z32, z32, z32)                 # in-degree 32 but <= WE_SMALL_EDGES edges: qualifies too
    h = torch.zeros(4, 4)
    ent = hidden_cache._Entry()
    ent.hidden, ent.token = h, HiddenToken()
    monkeypatch.setitem(hidden_cache._entries, m, ent)
    monkeypatch.setattr(hidden_cache, "MODE", "auto")
    monkeypatch.setattr(hidden_cache, "WE_MODE", "auto")
    assert hidden_cache.token_of(m, h, dense) is ent.token
    assert hidden_cache.token_of(m, torch.zeros(4, 4), dense) is None
    assert hidden_cache.token_of(m, h, sparse) is None and hidden_cache.token_of(m, h, small) is None
    monkeypatch.setattr(hidden_cache, "WE_MODE", "off")
    assert hidden_cache.token_of(m, h, sparse) is ent.token
    ent.token.valid = False
    assert hidden_cache.token_of(m, h, dense) is None
    assert hidden_cache.token_of(torch.nn.Linear(2, 2), h, dense) is None
