"""GPU tier: the depth-deferred backward (gpde_nnconv_bwd_light / gpde_nnconv_bwd_deferred).

`KernelNN.forward` applies ONE conv `depth` times (/root/reference/graph-neural-operator/UAI1_full_resolution.py:29-30) and
`loss.backward()` (:266) sums the kernel MLP's gradients over those applications.  When the hidden activations do not fit
memory the applications run the light backward and one deferred pass differentiates the hidden layers for all of them.
Checked here: the light pass returns the bits of the full backward for everything it computes; the deferred pass equals the
sum of the per-application gradients (split-f16 / fp32 summation order: <= 2e-5) and float64 autograd through the oracle;
the module-level wiring (shared virtual-H node) gives the gradients of the direct path; bit-reproducibility; no dependence
on workgroup timing."""
import pytest
import torch

import graph_pde_amd as gp
from graph_pde_amd import hidden_cache, ops
from oracle.nnconv_oracle import rel_l2
from tests.test_gpu_bwd import _case
from tests.test_gpu_parity import _oracle_grads, dev

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _dense_case(dims, n, deg, seed):
    """A graph with `deg` in-edges per node on average and a few long rows (several 256-slot tiles per node)."""
    torch.manual_seed(seed)
    e = n * deg
    dst = torch.randint(0, n, (e,))
    dst[: e // 8] = 7                                        # one destination with many tiles
    dst[e // 8: e // 8 + 300] = 11
    ei = torch.stack([torch.randint(0, n, (e,)), dst])
    ea, x = torch.randn(e, dims[0]), torch.randn(n, 64)
    mlp = torch.nn.Sequential(*sum([[torch.nn.Linear(dims[i], dims[i + 1]), torch.nn.ReLU()]
                                    for i in range(len(dims) - 1)], [])[:-1])
    ws_ = [l.weight.detach() for l in mlp if isinstance(l, torch.nn.Linear)]
    bs_ = [l.bias.detach() for l in mlp if isinstance(l, torch.nn.Linear)]
    root = torch.empty(64, 64).uniform_(-0.125, 0.125)
    bias = torch.empty(64).uniform_(-0.125, 0.125)
    return x, ei, ea, ws_, bs_, root, bias


def _to(d, *ts):
    return [None if t is None else ([u.to(d) for u in t] if isinstance(t, (list, tuple)) else t.to(d)) for t in ts]


@pytest.mark.parametrize("dims,n,e", [([6, 256, 256, 4096], 200, 9000), ([6, 1024, 1024, 4096], 60, 900), ([4, 512, 256, 4096], 120, 1100)])
def test_light_pass_returns_the_bits_of_the_full_backward(dims, n, e):
    x, ei, ea, ws_, bs_, root, bias, gout = _case(dims, n, e, sum(dims) + 1)
    d = dev()
    csr = ops.build_csr(ei.to(d), n)
    xd, ead, wd, bd, rd, gd = _to(d, x, ea, ws_, bs_, root, gout)
    fx, fW, fb, froot, fbias = ops.nnconv_backward_raw(xd, csr, ead, wd, bd, rd, "mean", gd)
    lx, lw, lb, lroot, lbias = ops.nnconv_backward_light_raw(xd, csr, ead, wd, bd, rd, "mean", gd)
    torch.cuda.synchronize()
    assert torch.equal(lx, fx) and torch.equal(lw, fW[-1]) and torch.equal(lb, fb[-1])
    assert torch.equal(lroot, froot) and torch.equal(lbias, fbias)


@pytest.mark.parametrize("dims,n,deg,L,aggr", [([6, 256, 256, 4096], 96, 120, 6, "mean"), ([6, 256, 256, 4096], 96, 120, 3, "add"),
                                               ([6, 1024, 1024, 4096], 48, 200, 5, "mean"), ([4, 256, 384, 4096], 64, 150, 1, "mean")])
def test_deferred_pass_equals_the_sum_of_the_per_application_gradients(dims, n, deg, L, aggr):
    x0, ei, ea, ws_, bs_, root, bias = _dense_case(dims, n, deg, 3 + L)
    assert ops.deferred_supported(dims)
    d = dev()
    csr = ops.build_csr(ei.to(d), n)
    ead, wd, bd, rd = _to(d, ea, ws_, bs_, root)
    torch.manual_seed(L)
    xs = [torch.randn(n, 64) * (0.3 + l) for l in range(L)]              # layers of different magnitude
    gs = [torch.randn(n, 64) * (2.0 ** -l) for l in range(L)]
    sumW = [torch.zeros_like(w, dtype=torch.float64) for w in ws_[:-1]]
    sumb = [torch.zeros_like(b, dtype=torch.float64) for b in bs_[:-1]]
    refW = [torch.zeros_like(w, dtype=torch.float64) for w in ws_[:-1]]
    refb = [torch.zeros_like(b, dtype=torch.float64) for b in bs_[:-1]]
    for xl, gl in zip(xs, gs):
        _, gW, gb, _, _ = ops.nnconv_backward_raw(xl.to(d), csr, ead, wd, bd, rd, aggr, gl.to(d))
        _, rW, rb, _, _ = _oracle_grads(xl, ei, ea, ws_, bs_, root, bias, aggr, gl)
        for k in range(len(sumW)):
            sumW[k] += gW[k].cpu().double(); sumb[k] += gb[k].cpu().double()
            refW[k] += rW[k].double(); refb[k] += rb[k].double()
    dW, db = ops.nnconv_backward_deferred_raw([t.to(d) for t in xs], [t.to(d) for t in gs], csr, ead, wd, bd, aggr)
    dW2, db2 = ops.nnconv_backward_deferred_raw([t.to(d) for t in xs], [t.to(d) for t in gs], csr, ead, wd, bd, aggr)
    torch.cuda.synchronize()
    for k in range(len(sumW)):
        assert rel_l2(dW[k].cpu().double(), sumW[k]) <= TOL, (k, rel_l2(dW[k].cpu().double(), sumW[k]))
        assert rel_l2(db[k].cpu().double(), sumb[k]) <= TOL, (k, rel_l2(db[k].cpu().double(), sumb[k]))
        assert rel_l2(dW[k].cpu().double(), refW[k]) <= TOL, (k, rel_l2(dW[k].cpu().double(), refW[k]))
        assert rel_l2(db[k].cpu().double(), refb[k]) <= TOL, (k, rel_l2(db[k].cpu().double(), refb[k]))
        assert torch.equal(dW[k], dW2[k]) and torch.equal(db[k], db2[k])              # bit-reproducible


def test_deferred_pass_in_several_chunks_and_under_workgroup_skew(monkeypatch):
    """A workspace a third of the planned one (several node / edge chunks) and odd column slices started late
    (GPDE_DEBUG_SKEW_US) leave the result unchanged to summation order / to the bit."""
    dims, n, deg, L = [6, 256, 256, 4096], 128, 100, 4
    x0, ei, ea, ws_, bs_, root, bias = _dense_case(dims, n, deg, 9)
    d = dev()
    csr = ops.build_csr(ei.to(d), n)
    ead, wd, bd = _to(d, ea, ws_, bs_)
    torch.manual_seed(1)
    xs = [torch.randn(n, 64, device=d) for _ in range(L)]
    gs = [torch.randn(n, 64, device=d) for _ in range(L)]
    a = ops.nnconv_backward_deferred_raw(xs, gs, csr, ead, wd, bd, "mean")
    monkeypatch.setenv("GPDE_DEBUG_SKEW_US", "300")
    b = ops.nnconv_backward_deferred_raw(xs, gs, csr, ead, wd, bd, "mean")
    monkeypatch.delenv("GPDE_DEBUG_SKEW_US")
    torch.cuda.synchronize()
    for k in range(2):
        assert torch.equal(a[0][k], b[0][k]) and torch.equal(a[1][k], b[1][k])
    real = ops._alloc_ws
    monkeypatch.setattr(ops, "_alloc_ws", lambda nbytes, dev_: real(nbytes // 3 if nbytes > (64 << 20) else nbytes, dev_))
    c = ops.nnconv_backward_deferred_raw(xs, gs, csr, ead, wd, bd, "mean")
    torch.cuda.synchronize()
    for k in range(2):
        assert rel_l2(c[0][k].cpu(), a[0][k].cpu()) <= 2e-6 and rel_l2(c[1][k].cpu(), a[1][k].cpu()) <= 2e-6


class _Net(torch.nn.Module):
    """KernelNN's use of the operator (UAI1_full_resolution.py:26-33): one conv applied `depth` times."""

    def __init__(self, dims, depth):
        super().__init__()
        mlp = torch.nn.Sequential(*sum([[torch.nn.Linear(dims[i], dims[i + 1]), torch.nn.ReLU()] for i in range(len(dims) - 1)], [])[:-1])
        self.conv1 = gp.NNConv_old(64, 64, mlp, aggr="mean")
        self.depth = depth

    def forward(self, x, ei, ea):
        for _ in range(self.depth):
            x = torch.relu(self.conv1(x, ei, ea))
        return x


def _grads(net, x, ei, ea, tgt):
    net.zero_grad(set_to_none=True)
    xin = x.clone().requires_grad_(True)
    loss = ((net(xin, ei, ea) - tgt) ** 2).mean()
    loss.backward()
    torch.cuda.synchronize()
    return [xin.grad.clone()] + [p.grad.clone() for p in net.parameters()], float(loss.detach())


def test_module_shares_a_virtual_hidden_node_when_h_does_not_fit(monkeypatch):
    dims, n, deg, depth = [6, 256, 256, 4096], 96, 110, 4
    x, ei, ea, *_ = _dense_case(dims, n, deg, 21)
    d = dev()
    torch.manual_seed(0)
    net = _Net(dims, depth).to(d)
    x, ei, ea = x.to(d), ei.to(d), ea.to(d)
    tgt = torch.randn(n, 64, device=d)
    monkeypatch.setattr(hidden_cache, "MODE", "off")
    ref, loss_ref = _grads(net, x, ei, ea, tgt)                      # every application: its own full backward
    monkeypatch.setattr(hidden_cache, "MODE", "auto")
    monkeypatch.setattr(hidden_cache, "BUDGET_BYTES", 0)             # H never fits
    hidden_cache.clear()
    first, loss1 = _grads(net, x, ei, ea, tgt)                      # call 1 plain, calls 2.. on the virtual H
    b1 = hidden_cache.stats.get("deferred_builds", 0)
    second, loss2 = _grads(net, x, ei, ea, tgt)                     # all calls on the virtual H
    third, _ = _grads(net, x, ei, ea, tgt)
    assert b1 == 1 and hidden_cache.stats.get("deferred_builds", 0) == 3
    assert hidden_cache.stats.get("deferred_hits", 0) == (depth - 2) + 2 * (depth - 1)
    assert loss1 == loss_ref and loss2 == loss_ref                    # the forward is the same kernel
    names = ["x"] + [k for k, _ in net.named_parameters()]
    for name, r, a, b, c in zip(names, ref, first, second, third):
        assert rel_l2(a.cpu(), r.cpu()) <= TOL, (name, rel_l2(a.cpu(), r.cpu()))
        assert rel_l2(b.cpu(), r.cpu()) <= TOL, (name, rel_l2(b.cpu(), r.cpu()))
        assert torch.equal(b, c), name                               # bit-reproducible step
    # a module applied once per forward does not stay on the virtual H
    net1 = _Net(dims, 1).to(d)
    hidden_cache.clear()
    for _ in range(3):
        _grads(net1, x, ei, ea, tgt)
        with torch.no_grad():
            for p in net1.parameters():
                p.mul_(1.0)                                          # an optimizer step: the version counters move
    assert hidden_cache.stats.get("deferred_builds", 0) == 0
    # switched off: the direct path
    monkeypatch.setattr(hidden_cache, "DEFER_MODE", "off")
    hidden_cache.clear()
    off, _ = _grads(net, x, ei, ea, tgt)
    off, _ = _grads(net, x, ei, ea, tgt)
    assert hidden_cache.stats.get("deferred_builds", 0) == 0
    for r, a in zip(ref, off):
        assert torch.equal(r, a)


def test_partial_hidden_activations_serve_the_light_and_deferred_passes(monkeypatch):
    """The part of H that fits the budget (in-edges of the leading nodes) is read instead of recomputed: raw calls against the
    recomputing ones, then the module (mixed forward + light backward + deferred pass on the virtual-H node) against the
    direct path."""
    dims, n, deg, L = [6, 256, 256, 4096], 128, 100, 3
    x0, ei, ea, ws_, bs_, root, bias = _dense_case(dims, n, deg, 31)
    d = dev()
    csr = ops.build_csr(ei.to(d), n)
    ead, wd, bd, rd = _to(d, ea, ws_, bs_, root)
    pm = ops.pack_mlp(wd, bd)
    hn = 64
    hpart, hmax = ops.hidden_forward_raw(csr, ead, pm, wd[:-1] + [None], bd[:-1] + [None], n_nodes_limit=hn)
    torch.manual_seed(2)
    xs = [torch.randn(n, 64, device=d) for _ in range(L)]
    gs = [torch.randn(n, 64, device=d) for _ in range(L)]
    a = ops.nnconv_backward_light_raw(xs[0], csr, ead, wd, bd, rd, "mean", gs[0])
    b = ops.nnconv_backward_light_raw(xs[0], csr, ead, wd, bd, rd, "mean", gs[0], hidden_part=hpart, hidden_nodes=hn)
    for u, v in zip(a, b):
        assert rel_l2(v.cpu(), u.cpu()) <= 1e-6
    da = ops.nnconv_backward_deferred_raw(xs, gs, csr, ead, wd, bd, "mean")
    db = ops.nnconv_backward_deferred_raw(xs, gs, csr, ead, wd, bd, "mean", hidden_part=hpart, hidden_nodes=hn)
    for k in range(2):
        assert rel_l2(db[0][k].cpu(), da[0][k].cpu()) <= 2e-6 and rel_l2(db[1][k].cpu(), da[1][k].cpu()) <= 2e-6
    # module level: budget = about half of H
    depth = 4
    torch.manual_seed(0)
    net = _Net(dims, depth).to(d)
    x, eid, tgt = x0.to(d), ei.to(d), torch.randn(n, 64, device=d)
    monkeypatch.setattr(hidden_cache, "MODE", "off")
    ref, loss_ref = _grads(net, x, eid, ead, tgt)
    monkeypatch.setattr(hidden_cache, "MODE", "auto")
    monkeypatch.setattr(hidden_cache, "BUDGET_BYTES", int(csr.n_edges * 256 * 4 * 0.7))        # rows of 64 of the 128 nodes fit
    hidden_cache.clear()
    _grads(net, x, eid, ead, tgt)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(1.0)                                              # an optimizer step: the version counters move
    got, loss = _grads(net, x, eid, ead, tgt)                      # steady state: partial H from the first call on
    assert hidden_cache.stats["builds"] >= 2 and hidden_cache.stats.get("deferred_builds", 0) == 2
    ent = hidden_cache._entries[net.conv1]
    assert ent.hidden is not None and 0 < ent.hn < n and ent.hn % 64 == 0
    assert abs(loss - loss_ref) <= 1e-5 * abs(loss_ref)
    names = ["x"] + [k for k, _ in net.named_parameters()]
    for name, r, g in zip(names, ref, got):
        assert rel_l2(g.cpu(), r.cpu()) <= TOL, (name, rel_l2(g.cpu(), r.cpu()))
    # the cache gives way (ops._alloc_ws on OOM -> release_all): the hanging applications recompute, same gradients
    net.zero_grad(set_to_none=True)
    xin = x.clone().requires_grad_(True)
    loss2 = ((net(xin, eid, ead) - tgt) ** 2).mean()
    assert hidden_cache.release_all()
    loss2.backward()
    torch.cuda.synchronize()
    for name, r, g in zip(names, ref, [xin.grad] + [p.grad for p in net.parameters()]):
        assert rel_l2(g.cpu(), r.cpu()) <= TOL, (name, rel_l2(g.cpu(), r.cpu()))


def test_unsupported_kernels_keep_the_plain_backward(monkeypatch):
    """Kernel MLPs outside the deferred form ([6, 64, 128, 4096]: the checkpoint's widths) are never deferred."""
    dims = [6, 64, 128, 4096]
    assert not ops.deferred_supported(dims)
    x, ei, ea, *_ = _dense_case(dims, 64, 40, 2)
    d = dev()
    net = _Net(dims, 3).to(d)
    monkeypatch.setattr(hidden_cache, "BUDGET_BYTES", 0)
    hidden_cache.clear()
    for _ in range(2):
        _grads(net, x.to(d), ei.to(d), ea.to(d), torch.zeros(64, 64, device=d))
    assert hidden_cache.stats.get("deferred_builds", 0) == 0


def test_abandoned_or_input_only_backward_leaves_nothing_for_the_next_deferred_pass(monkeypatch):
    """a backward that never reaches the virtual-H node - `torch.autograd.grad(out, x)` (input gradients only), or a
    pass abandoned by an exception - used to leave its (x, grad_out) pairs on the shared token; with unchanged parameters the
    next forward re-used the token and the next deferred pass added the stale applications to the hidden-layer gradients."""
    dims, n, deg, depth = [6, 256, 256, 4096], 96, 110, 3
    x, ei, ea, *_ = _dense_case(dims, n, deg, 41)
    d = dev()
    torch.manual_seed(0)
    net = _Net(dims, depth).to(d)
    x, ei, ea = x.to(d), ei.to(d), ea.to(d)
    tgt = torch.randn(n, 64, device=d)
    monkeypatch.setattr(hidden_cache, "MODE", "auto")
    monkeypatch.setattr(hidden_cache, "BUDGET_BYTES", 0)             # H never fits: the virtual-H node
    hidden_cache.clear()
    _grads(net, x, ei, ea, tgt)                                      # the module is seen repeating
    ref, _ = _grads(net, x, ei, ea, tgt)                             # steady state: every application on the virtual H
    # (a) input gradients only: the light passes run, the deferred pass does not
    xin = x.clone().requires_grad_(True)
    out = net(xin, ei, ea)
    gx_only, = torch.autograd.grad(((out - tgt) ** 2).mean(), xin)
    tok = hidden_cache._entries[net.conv1].dtoken
    assert tok is not None and tok.valid and len(tok.stash) == depth
    assert rel_l2(gx_only.cpu(), ref[0].cpu()) <= 1e-6
    got, _ = _grads(net, x, ei, ea, tgt)                             # same parameters: same key, same token
    assert hidden_cache.stats.get("deferred_stale_dropped", 0) >= 1
    for name, r, g in zip(["x"] + [k for k, _ in net.named_parameters()], ref, got):
        assert torch.equal(r, g), name
    # (b) the same application differentiated twice before the deferred pass (retain_graph): it counts once
    net.zero_grad(set_to_none=True)
    xin = x.clone().requires_grad_(True)
    loss = ((net(xin, ei, ea) - tgt) ** 2).mean()
    torch.autograd.grad(loss, xin, retain_graph=True)
    loss.backward()
    torch.cuda.synchronize()
    for name, r, g in zip(["x"] + [k for k, _ in net.named_parameters()], ref, [xin.grad] + [p.grad for p in net.parameters()]):
        assert rel_l2(g.cpu(), r.cpu()) <= 1e-6, (name, rel_l2(g.cpu(), r.cpu()))
