"""GPU tier: the operator given the per-edge weights (gpde_edge_weights_fwd + gpde_nnconv_fwd_edgeweights_group;
SURVEY.md §8 row f4 second half, row a6 'max'): against the float64 oracle for add / mean / max, against the module's
other paths, grouped launch == individual calls bit for bit, the cache policy, and the MGKN-orthogonal sweep."""
import pytest
import torch

import graph_pde_amd as gp
from graph_pde_amd import _lib, hidden_cache, mgkn_workloads, ops
from oracle.nnconv_oracle import nnconv_forward, rel_l2
from tests.test_host_logic import DenseNet

pytestmark = pytest.mark.gpu


def _graph(n, e, k0, seed, dev):
    g = torch.Generator().manual_seed(seed)
    ei = torch.stack([torch.randint(0, n, (e,), generator=g), torch.randint(0, max(n - 3, 1), (e,), generator=g)])
    if e > 40:
        ei[1, :e // 8] = 2                                   # one destination with many in-edges; the last nodes have none
    return ei.to(dev), torch.randn(e, k0, generator=g).to(dev), torch.randn(n, 64, generator=g).to(dev)


def _oracle(conv, x, ei, ea, aggr):
    lin = ops.mlp_linears(conv.nn)
    return nnconv_forward(x.cpu(), ei.cpu(), ea.cpu(), [l.weight.detach().cpu() for l in lin],
                          [None if l.bias is None else l.bias.detach().cpu() for l in lin],
                          None if conv.root is None else conv.root.detach().cpu(),
                          None if conv.bias is None else conv.bias.detach().cpu(), aggr=aggr, dtype=torch.float64, chunk_edges=4096)


def _we_call(conv, x, ei, ea, **kw):
    lin = ops.mlp_linears(conv.nn)
    ws_, bs_ = [l.weight for l in lin], [l.bias for l in lin]
    pm = ops.pack_mlp(ws_, bs_)
    csr = ops.csr_for(ei, x.shape[0])
    h, _ = ops.hidden_forward_raw(csr, ea, pm, ws_[:-1] + [None], bs_[:-1] + [None], "f16split")
    we = ops.edge_weights_raw(h, pm, ws_[-1], bs_[-1])
    return dict(x=x, csr=csr, edge_weights=we, root=conv.root, bias=conv.bias, aggr=conv.aggr, **kw)


@pytest.mark.parametrize("aggr", ["add", "mean", "max"])
@pytest.mark.parametrize("dims,n,e", [([6, 256, 256, 4096], 300, 900), ([4, 1024, 512, 4096], 500, 1400),
                                      ([4, 16, 16, 4096], 40, 130), ([6, 100, 72, 4096], 64, 700), ([6, 32, 4096], 90, 250),
                                      ([5, 24, 40, 56, 4096], 50, 333)])
def test_edge_weight_operator_matches_float64_oracle(dims, n, e, aggr):
    d = torch.device("cuda:0")
    torch.manual_seed(sum(dims) + n)
    conv = gp.NNConv_old(64, 64, DenseNet(dims, torch.nn.ReLU), aggr=aggr).to(d)
    ei, ea, x = _graph(n, e, dims[0], 3 * n + e, d)
    with torch.no_grad():
        y = ops.nnconv_forward_edgeweights_group([_we_call(conv, x, ei, ea)])[0]
    torch.cuda.synchronize()
    err = rel_l2(y.cpu(), _oracle(conv, x, ei, ea, aggr))
    assert err <= 1e-5, err                                  # north_star's bar; measured ~1e-7
    assert err <= 2e-6, err
    deg = torch.bincount(ei[1].cpu(), minlength=n)
    if aggr != "add" and bool((deg == 0).any()):             # nodes without in-edges: aggregation term is 0
        i = int((deg == 0).nonzero()[0])
        ref_i = x[i].cpu().double() @ conv.root.detach().cpu().double() + conv.bias.detach().cpu().double()
        assert torch.allclose(y[i].cpu().double(), ref_i, atol=1e-5)


def test_edge_weights_are_the_reference_weight_tensor():
    """W_e = self.nn(pseudo) (nn_conv.py:274) in CSR slot order - both builders (split-f16 GEMM at k2 >= 256, fp32 GEMM below)."""
    d = torch.device("cuda:0")
    for dims in ([6, 256, 256, 4096], [6, 64, 48, 4096]):
        torch.manual_seed(5)
        mlp = DenseNet(dims, torch.nn.ReLU).to(d)
        conv = gp.NNConv_old(64, 64, mlp, aggr="mean").to(d)
        ei, ea, x = _graph(120, 400, 6, 11, d)
        call = _we_call(conv, x, ei, ea)
        perm = call["csr"].perm.long()
        ref = mlp.double()(ea[perm].double())
        err = float((call["edge_weights"].double() - ref).norm() / ref.norm())
        assert err <= 3e-7, (dims, err)


def test_grouped_launch_is_bit_identical_to_individual_calls_and_glue_is_fused():
    d = torch.device("cuda:0")
    calls, singles = [], []
    for k, (dims, n, e, aggr) in enumerate([([4, 64, 64, 4096], 200, 600, "mean"), ([4, 16, 16, 4096], 9, 20, "mean"),
                                            ([6, 128, 4096], 77, 500, "add"), ([4, 256, 256, 4096], 333, 1000, "max")] * 5):
        torch.manual_seed(100 + k)
        conv = gp.NNConv(64, 64, DenseNet(dims, torch.nn.ReLU), aggr=aggr, root_weight=k % 3 != 0, bias=k % 2 == 0).to(d)
        ei, ea, x = _graph(n, e, dims[0], 50 + k, d)
        res = torch.randn(n, 64, device=d) if k % 2 else None
        c = _we_call(conv, x, ei, ea, residual=res, relu=k % 4 == 1)
        calls.append(c)
        with torch.no_grad():
            y1 = ops.nnconv_forward_edgeweights_group([dict(c)])[0]
            base = ops.nnconv_forward_edgeweights_group([dict(c, residual=None, relu=False)])[0]
        comp = base if res is None else res + base
        comp = torch.relu(comp) if c["relu"] else comp
        assert torch.equal(y1, comp)                         # fused glue == the unfused composition, bit for bit
        singles.append(y1)
    with torch.no_grad():
        ys = ops.nnconv_forward_edgeweights_group(calls)     # 20 descriptors: two launches
    for a, b in zip(ys, singles):
        assert torch.equal(a, b)


def test_group_rejects_dependent_calls():
    d = torch.device("cuda:0")
    torch.manual_seed(0)
    conv = gp.NNConv(64, 64, DenseNet([4, 16, 16, 4096], torch.nn.ReLU), aggr="mean").to(d)
    ei, ea, x = _graph(30, 80, 4, 1, d)
    c1 = _we_call(conv, x, ei, ea)
    out = torch.empty(30, 64, device=d)
    with pytest.raises(_lib.GpdeError, match="independent"):
        ops.nnconv_forward_edgeweights_group([dict(c1, out=out), dict(c1, x=out)])


def test_module_takes_the_edge_weight_path_for_repeated_calls(monkeypatch):
    d = torch.device("cuda:0")
    monkeypatch.setattr(hidden_cache, "MODE", "auto")
    monkeypatch.setattr(hidden_cache, "WE_MODE", "auto")
    hidden_cache.clear()
    torch.manual_seed(3)
    conv = gp.NNConv(64, 64, DenseNet([4, 128, 128, 4096], torch.nn.ReLU), aggr="mean").to(d)
    ei, ea, x = _graph(500, 1500, 4, 9, d)
    ref = _oracle(conv, x, ei, ea, "mean")
    with torch.no_grad():
        ys = [conv(x, ei, ea) for _ in range(5)]             # direct, H built, W_e built, W_e hit, W_e hit
    assert hidden_cache.stats["we_builds"] == 1 and hidden_cache.stats["we_hits"] == 2
    for y in ys:
        assert rel_l2(y.cpu(), ref) <= 2e-6
    assert torch.equal(ys[3], ys[4]) and not torch.equal(ys[0], ys[4])        # another summation order than the direct path
    with torch.no_grad():                                     # a weight update invalidates W_e (version counter)
        ops.mlp_linears(conv.nn)[-1].weight.mul_(0.5)
        y6 = conv(x, ei, ea)
    assert rel_l2(y6.cpu(), _oracle(conv, x, ei, ea, "mean")) <= 2e-6
    # with gradients the module (known to repeat) builds H and W_e as autograd nodes (round 4): same value, real gradients
    b0 = hidden_cache.stats["we_builds"] + hidden_cache.stats["we_hits"]
    xg = x.clone().requires_grad_(True)
    yg = conv(xg, ei, ea)
    yg.sum().backward()
    assert hidden_cache.stats["we_builds"] + hidden_cache.stats["we_hits"] == b0 + 1 and xg.grad is not None
    assert rel_l2(yg.detach().cpu(), _oracle(conv, x, ei, ea, "mean")) <= 2e-6
    b0 += 1
    # a dense graph (in-degree > 4, > 8192 edges) keeps the re-associated path
    ei2, ea2, x2 = _graph(300, 9000, 4, 2, d)
    with torch.no_grad():
        for _ in range(4):
            conv(x2, ei2, ea2)
    assert hidden_cache.stats["we_builds"] + hidden_cache.stats["we_hits"] == b0
    hidden_cache.clear()


def test_aggr_max_through_the_module(monkeypatch):
    """nn_conv.py:222-224 names 'max'; no script uses it.  Inference runs from the per-edge weights; with gradients the module
    composes PyG's chain (gather, native message(), segment max, native update()) - checked against float64 autograd."""
    d = torch.device("cuda:0")
    hidden_cache.clear()
    torch.manual_seed(8)
    conv = gp.NNConv_old(64, 64, DenseNet([6, 64, 128, 4096], torch.nn.ReLU), aggr="max").to(d)
    ei, ea, x = _graph(150, 2000, 6, 4, d)
    with torch.no_grad():
        y = conv(x, ei, ea)
    assert rel_l2(y.cpu(), _oracle(conv, x, ei, ea, "max")) <= 2e-6
    # a freshly built module in grad mode (parameters require grad): differentiable, same value, gradients of float64 autograd
    xin = x.clone().requires_grad_(True)
    yg = conv(xin, ei, ea)
    assert rel_l2(yg.detach().cpu(), y.cpu()) <= 2e-6
    g = torch.randn_like(yg)
    (yg * g).sum().backward()
    lin = ops.mlp_linears(conv.nn)
    x64 = x.detach().cpu().double().requires_grad_(True)
    W = [l.weight.detach().cpu().double().requires_grad_(True) for l in lin]
    B = [l.bias.detach().cpu().double().requires_grad_(True) for l in lin]
    h = ea.cpu().double()
    for k in range(len(W)):
        h = torch.nn.functional.linear(h, W[k], B[k])
        if k < len(W) - 1:
            h = torch.relu(h)
    eic = ei.cpu()
    m = torch.matmul(x64[eic[0]].unsqueeze(1), h.view(-1, 64, 64)).squeeze(1)
    agg = torch.full((x.shape[0], 64), -1e9, dtype=torch.float64).scatter_reduce(0, eic[1].view(-1, 1).expand_as(m), m, "amax")
    agg = torch.where(agg == -1e9, torch.zeros_like(agg), agg)
    ref = agg + x64 @ conv.root.detach().cpu().double() + conv.bias.detach().cpu().double()
    (ref * g.cpu().double()).sum().backward()
    assert rel_l2(xin.grad.cpu(), x64.grad) <= 2e-5
    for k, l in enumerate(lin):
        assert rel_l2(l.weight.grad.cpu(), W[k].grad) <= 2e-5, k
    hidden_cache.clear()


def test_burgers_sweep_grouped_equals_fused_glue_calls(monkeypatch):
    """The 13 independent convs of an MGKN-orthogonal sweep (MGKN_orthogonal_burgers1d.py:73-82) in one grouped launch:
    identical bits to calling the modules one by one (same per-edge weight kernel, fused glue), and within rounding of
    the default module path."""
    d = torch.device("cuda:0")
    hidden_cache.clear()
    monkeypatch.setattr(hidden_cache, "WE_MODE", "off")
    wl_0 = mgkn_workloads.orthogonal_burgers(d, s=1024, seed=2)
    y0 = wl_0.forward()                                      # the fused kernels (edge-weight cache off)
    assert hidden_cache.stats["we_builds"] == 0
    monkeypatch.setattr(hidden_cache, "WE_MODE", "auto")
    wl_f = mgkn_workloads.orthogonal_burgers(d, s=1024, seed=2, fused_glue=True)
    wl_g = mgkn_workloads.orthogonal_burgers(d, s=1024, seed=2, grouped=True)
    for _ in range(2):                                       # second forward: every level served from cached W_e
        a, b = wl_f.forward(), wl_g.forward()
    assert hidden_cache.stats["we_hits"] > 0
    for u, v, w in zip(a, b, y0):
        assert torch.equal(u, v)
        assert rel_l2(u.cpu(), w.cpu()) <= 2e-6
    hidden_cache.clear()


# ---- training on the per-edge weights (gpde_nnconv_bwd_edgeweights / gpde_edge_weights_bwd) ------------------------------
@pytest.mark.parametrize("aggr", ["mean", "add"])
def test_backward_given_the_edge_weights_matches_float64(aggr):
    d = torch.device("cuda:0")
    torch.manual_seed(11)
    n, e = 300, 1100
    ei, _, x = _graph(n, e, 4, 5, d)
    we = torch.randn(e, 4096, device=d) * 0.05
    root = torch.randn(64, 64, device=d) * 0.1
    g = torch.randn(n, 64, device=d)
    csr = ops.build_csr(ei, n)
    gx, gwe, groot, gbias = ops.nnconv_backward_edgeweights_raw(x, csr, we, root, aggr, g)
    gx2, gwe2, groot2, gbias2 = ops.nnconv_backward_edgeweights_raw(x, csr, we, root, aggr, g)
    assert torch.equal(gx, gx2) and torch.equal(gwe, gwe2) and torch.equal(groot, groot2) and torch.equal(gbias, gbias2)
    # float64 autograd of the same formula, edge tensors in CSR slot order
    src, dst = csr.src.long().cpu(), csr.dst.long().cpu()
    x64 = x.cpu().double().requires_grad_(True)
    w64 = we.cpu().double().requires_grad_(True)
    r64 = root.cpu().double().requires_grad_(True)
    b64 = torch.zeros(64, dtype=torch.float64, requires_grad=True)
    m = torch.matmul(x64[src].unsqueeze(1), w64.view(-1, 64, 64)).squeeze(1)
    out = torch.zeros(n, 64, dtype=torch.float64).index_add(0, dst, m)
    if aggr == "mean":
        out = out / torch.bincount(dst, minlength=n).clamp(min=1).double().unsqueeze(1)
    out = out + x64 @ r64 + b64
    (out * g.cpu().double()).sum().backward()
    assert rel_l2(gx.cpu(), x64.grad) <= 2e-6 and rel_l2(gwe.cpu(), w64.grad) <= 2e-6
    assert rel_l2(groot.cpu(), r64.grad) <= 2e-5 and rel_l2(gbias.cpu(), b64.grad) <= 2e-5
    # the forward it differentiates
    y = ops.nnconv_forward_edgeweights_raw(x, csr, we, root, None, aggr)
    assert rel_l2(y.cpu(), (out - b64).detach()) <= 2e-6


@pytest.mark.parametrize("dims,e", [([4, 128, 256, 4096], 3000), ([4, 16, 16, 4096], 700), ([6, 64, 200, 4096], 1500)])
def test_backward_of_the_edge_weights_matches_float64(dims, e):
    """gpde_edge_weights_bwd: grad_hidden (masked), grad of the last Linear from the summed dL/dW_e - split-f16 GEMMs where the
    padded width allows, fp32 MFMA otherwise."""
    d = torch.device("cuda:0")
    torch.manual_seed(e)
    K2P = ops.hidden_width(dims)
    hidden = torch.zeros(e, K2P, device=d)
    hidden[:, :dims[-2]] = torch.relu(torch.randn(e, dims[-2], device=d))
    gwe = torch.randn(e, 4096, device=d)
    w3 = torch.randn(4096, dims[-2], device=d) / dims[-2] ** 0.5
    gh, gw, gb = ops.edge_weights_backward_raw(gwe, hidden, dims, w3)
    gh2, gw2, gb2 = ops.edge_weights_backward_raw(gwe, hidden, dims, w3)
    assert torch.equal(gh, gh2) and torch.equal(gw, gw2) and torch.equal(gb, gb2)
    h64, g64, w64 = hidden[:, :dims[-2]].cpu().double(), gwe.cpu().double(), w3.cpu().double()
    ref_h = (g64 @ w64) * (h64 > 0)
    assert rel_l2(gh[:, :dims[-2]].cpu(), ref_h) <= 2e-5 and float(gh[:, dims[-2]:].abs().sum()) == 0.0
    assert rel_l2(gw.cpu(), g64.t() @ h64) <= 2e-5 and rel_l2(gb.cpu(), g64.sum(0)) <= 2e-5


def test_training_step_on_the_shared_edge_weights_matches_the_fused_path(monkeypatch):
    """A module applied `depth` times on a low in-degree graph with gradients (the MGKN V-cycle, MGKN_orthogonal_burgers1d.py:
    65-82, :226-242): H and W_e are autograd nodes shared by the applications; every gradient equals the fused path's and
    float64 autograd through the oracle."""
    from oracle.nnconv_oracle import nnconv_grads
    d = torch.device("cuda:0")
    torch.manual_seed(13)
    dims, n, e, depth = [4, 128, 128, 4096], 600, 1500, 3
    conv = gp.NNConv(64, 64, DenseNet(dims, torch.nn.ReLU), aggr="mean").to(d)
    ei, ea, x = _graph(n, e, 4, 21, d)
    tgt = torch.randn(n, 64, device=d)

    def step():
        conv.zero_grad(set_to_none=True)
        xin = x.clone().requires_grad_(True)
        h = xin
        for _ in range(depth):
            h = torch.relu(h + conv(h, ei, ea))
        loss = (h - tgt).square().mean()
        loss.backward()
        torch.cuda.synchronize()
        return [xin.grad.clone()] + [p.grad.clone() for p in conv.parameters()], float(loss.detach())
    monkeypatch.setattr(hidden_cache, "MODE", "auto")
    monkeypatch.setattr(hidden_cache, "WE_MODE", "off")
    hidden_cache.clear()
    step()
    ref, loss_ref = step()                                   # H shared (NNConvHiddenFunction), per-call fused kernels
    monkeypatch.setattr(hidden_cache, "WE_MODE", "auto")
    hidden_cache.clear()
    step()
    got, loss = step()
    assert hidden_cache.stats["we_builds"] >= 1 and hidden_cache.stats["we_hits"] >= depth - 2
    assert abs(loss - loss_ref) <= 1e-5 * abs(loss_ref)
    names = ["x"] + [k for k, _ in conv.named_parameters()]
    for name, r, g_ in zip(names, ref, got):
        assert rel_l2(g_.cpu(), r.cpu()) <= 2e-5, (name, rel_l2(g_.cpu(), r.cpu()))
    again, _ = step()
    for a_, b_ in zip(got, again):
        assert torch.equal(a_, b_)                           # bit-reproducible
    # one application against float64 autograd through the oracle (the same W_e path: the module repeats within `step`)
    lin = ops.mlp_linears(conv.nn)
    conv.zero_grad(set_to_none=True)
    xin = x.clone().requires_grad_(True)
    gout = torch.randn(n, 64, device=d)
    y1 = conv(xin, ei, ea)
    y2 = conv(xin, ei, ea)                                   # second application of the same (edge_attr, weights): shared nodes
    ((y1 + y2) * gout).sum().backward()
    rx, rW, rb, rroot, rbias = nnconv_grads(x.cpu(), ei.cpu(), ea.cpu(), [l.weight.detach().cpu() for l in lin],
                                            [l.bias.detach().cpu() for l in lin], conv.root.detach().cpu(), conv.bias.detach().cpu(),
                                            "mean", 2 * gout.cpu())
    assert rel_l2(xin.grad.cpu(), rx) <= 2e-5
    for l, layer in enumerate(lin):
        assert rel_l2(layer.weight.grad.cpu(), rW[l]) <= 2e-5 and rel_l2(layer.bias.grad.cpu(), rb[l]) <= 2e-5, l
    assert rel_l2(conv.root.grad.cpu(), rroot) <= 2e-5 and rel_l2(conv.bias.grad.cpu(), rbias) <= 2e-5
    hidden_cache.clear()
