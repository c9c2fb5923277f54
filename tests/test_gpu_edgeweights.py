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


def test_module_takes_the_edge_weight_path_for_repeated_inference_calls_only(monkeypatch):
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
    # training calls never use it
    b0 = hidden_cache.stats["we_builds"] + hidden_cache.stats["we_hits"]
    xg = x.clone().requires_grad_(True)
    conv(xg, ei, ea).sum().backward()
    assert hidden_cache.stats["we_builds"] + hidden_cache.stats["we_hits"] == b0 and xg.grad is not None
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
    wl_0 = mgkn_workloads.orthogonal_burgers(d, s=1024, seed=2)
    y0 = wl_0.forward()                                      # default path (edge-weight cache off)
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
