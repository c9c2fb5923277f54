"""GPU tier: the applications of a module that shares its per-edge weights W_e sum dL/dW_e, grad_root and grad_bias INSIDE the
kernels (round 6; gpde_nnconv_bwd_edgeweights_acc, autograd.WeConvFunction.backward / SharedParamFunction).

The MGKN loops apply each NNConv `depth` times per step on graphs of low in-degree
(/root/reference/multipole-graph-neural-operator/MGKN_general_darcy2d.py:76-90, MGKN_orthogonal_burgers1d.py:69-86); on those the
operator runs on W_e as a shared autograd node, every application's backward produces dL/dW_e [E, 4096] and the gradients of the
same root / bias, and autograd adds them with one elementwise kernel per application and tensor (190 - 220 launches of an MGKN
training step).  Now the first application of a backward pass hands autograd its tensors and the others add to them in the
kernels - the same additions in the same order.  Checked:
  * all gradients BITWISE equal to the autograd-summed form (GPDE_ACCUMULATE_DLDH=0), and the path is really taken;
  * a second backward over a retained graph and a partial backward (inputs only) followed by a full one: each pass starts its own
    tensors (keyed on the autograd graph task);
  * root used a second time by the caller's loss (a regulariser): the in-place sum lives on a private node, the leaf's own sum is
    autograd's - values agree with the autograd-summed form;
  * the raw call: accumulate == write + add (unfused: the same bits), and the argument check."""
import pytest
import torch

import graph_pde_amd as gp
from graph_pde_amd import _lib
from graph_pde_amd import autograd as gpa
from graph_pde_amd import hidden_cache, ops, synth

pytestmark = pytest.mark.gpu
DIMS = [6, 128, 128, 4096]
DEPTH = 4


def _model(seed=0, bias=True):
    torch.manual_seed(seed)
    mlp = torch.nn.Sequential(*sum([[torch.nn.Linear(DIMS[i], DIMS[i + 1]), torch.nn.ReLU()] for i in range(3)], [])[:-1])
    return gp.NNConv_old(64, 64, mlp, aggr="mean", bias=bias).to("cuda:0")


def _forward(conv, x, ei, ea):
    h = x
    for _ in range(DEPTH):
        h = torch.relu(conv(h, ei, ea))
    return h


def _grads(conv, xin):
    return [xin.grad.clone()] + [p.grad.clone() for p in conv.parameters()]


@pytest.fixture()
def case(monkeypatch):
    monkeypatch.setattr(hidden_cache, "MODE", "on")           # H (and with it W_e) shared from the first application on
    monkeypatch.setattr(hidden_cache, "WE_MODE", "auto")
    hidden_cache.clear()
    d = torch.device("cuda:0")
    ei, ea, n = synth.darcy_graph(24, 0.06, device=d)          # a few thousand edges: qualifies for the per-edge weight form
    assert ei.shape[1] <= hidden_cache.WE_SMALL_EDGES or ei.shape[1] <= 4 * n
    return {"conv": _model(), "ei": ei, "ea": ea, "x": torch.randn(n, 64, device=d), "g": torch.randn(n, 64, device=d)}


def _step(c, flag, monkeypatch, passes=1, reg=False):
    monkeypatch.setattr(gpa, "ACCUMULATE_GRAD_HIDDEN", flag)
    hidden_cache.clear()
    conv = c["conv"]
    conv.zero_grad(set_to_none=True)
    xin = c["x"].clone().requires_grad_(True)
    before, we0 = ops.n_grad_hidden_accumulated, hidden_cache.stats["we_hits"] + hidden_cache.stats["we_builds"]
    loss = (_forward(conv, xin, c["ei"], c["ea"]) * c["g"]).sum()
    if reg:
        loss = loss + 0.5 * conv.root.square().sum() + conv.bias.abs().sum()
    assert hidden_cache.stats["we_hits"] + hidden_cache.stats["we_builds"] - we0 == DEPTH      # every application ran on W_e
    for k in range(passes):
        loss.backward(retain_graph=k + 1 < passes)
    torch.cuda.synchronize()
    return _grads(conv, xin), ops.n_grad_hidden_accumulated - before


def test_in_kernel_sums_are_bitwise_autograds_sums(case, monkeypatch):
    ref, n0 = _step(case, False, monkeypatch)
    acc, n1 = _step(case, True, monkeypatch)
    assert n0 == 0 and n1 == DEPTH - 1                         # the first application writes, the others add
    for a, b in zip(acc, ref):
        assert torch.equal(a, b)
    # a second pass over the retained graph starts its own tensors: .grad doubles exactly as with autograd's sums
    ref2, _ = _step(case, False, monkeypatch, passes=2)
    acc2, n2 = _step(case, True, monkeypatch, passes=2)
    assert n2 == 2 * (DEPTH - 1)
    for a, b in zip(acc2, ref2):
        assert torch.equal(a, b)


def test_no_bias_module_and_partial_pass(case, monkeypatch):
    case["conv"] = _model(seed=1, bias=False)
    ref, _ = _step(case, False, monkeypatch)
    acc, n1 = _step(case, True, monkeypatch)
    assert n1 == DEPTH - 1
    for a, b in zip(acc, ref):
        assert torch.equal(a, b)
    conv = case["conv"]

    def partial_then_full(flag):
        monkeypatch.setattr(gpa, "ACCUMULATE_GRAD_HIDDEN", flag)
        hidden_cache.clear()
        conv.zero_grad(set_to_none=True)
        xin = case["x"].clone().requires_grad_(True)
        loss = (_forward(conv, xin, case["ei"], case["ea"]) * case["g"]).sum()
        (gx,) = torch.autograd.grad(loss, xin, retain_graph=True)      # inputs only: the W_e node's backward does not run
        loss.backward()                                                  # ... then everything over the same graph
        torch.cuda.synchronize()
        return gx, _grads(conv, xin)
    gx0, seq0 = partial_then_full(False)
    gx1, seq1 = partial_then_full(True)
    assert torch.equal(gx1, ref[0]) and torch.equal(gx0, gx1)
    for a, b in zip(seq1, seq0):
        assert torch.equal(a, b)
    # ... and a fresh step afterwards is the reference step again (nothing of the partial pass is left on the token)
    again, _ = _step(case, True, monkeypatch)
    for a, b in zip(again, ref):
        assert torch.equal(a, b)


def test_a_second_use_of_root_and_bias_in_the_loss(case, monkeypatch):
    ref, _ = _step(case, False, monkeypatch, reg=True)
    acc, n1 = _step(case, True, monkeypatch, reg=True)
    assert n1 == DEPTH - 1
    for a, b in zip(acc, ref):          # (the leaf's sum is formed in another order: regulariser + in-place sum)
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6 * float(b.abs().max()))
    plain, _ = _step(case, True, monkeypatch)
    names = [n for n, _ in case["conv"].named_parameters()]
    k_root = 1 + names.index("root")
    assert not torch.equal(acc[k_root], plain[k_root])           # the regulariser's part arrived


def test_raw_accumulate_is_write_plus_add_and_is_checked(case):
    conv, ei, ea, x, g = case["conv"], case["ei"], case["ea"], case["x"], case["g"]
    csr = ops.build_csr(ei, x.shape[0])
    lin = ops.mlp_linears(conv.nn)
    W, B = [l.weight.detach() for l in lin], [l.bias.detach() for l in lin]
    pm = ops.pack_mlp(W, B)
    H, _ = ops.hidden_forward_raw(csr, ea, pm, W, B)
    we = ops.edge_weights_raw(H, pm, W[-1], B[-1])
    root = conv.root.detach()
    r1 = ops.nnconv_backward_edgeweights_raw(x, csr, we, root, "mean", g)
    x2, g2 = torch.randn_like(x), torch.randn_like(g)              # (unrelated values: a sum that rounds)
    r2 = ops.nnconv_backward_edgeweights_raw(x2, csr, we, root, "mean", g2)
    acc = tuple(t.clone() for t in r1[1:])
    r3 = ops.nnconv_backward_edgeweights_raw(x2, csr, we, root, "mean", g2, acc=acc)
    assert torch.equal(r3[0], r2[0])                               # grad_x is this application's own
    for k in range(1, 4):
        assert r3[k].data_ptr() == acc[k - 1].data_ptr() and torch.equal(r3[k], r1[k] + r2[k])
    with pytest.raises(ValueError):
        ops.nnconv_backward_edgeweights_raw(x, csr, we, root, "mean", g, acc=(acc[0][:-1], acc[1], acc[2]))
    l = _lib.lib()
    one = torch.zeros(64, device=x.device)
    assert l.gpde_nnconv_bwd_edgeweights_acc(one.data_ptr(), 0, None, 0, csr.rowptr.data_ptr(), None, None, None, None, 1,
                                             one.data_ptr(), None, None, None, None, 8, one.data_ptr(), 256, None) == -1
