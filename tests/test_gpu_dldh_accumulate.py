"""GPU tier: the applications of a module that shares its hidden activations sum their dL/dH INSIDE the per-edge kernel
(round 5; GPDE_BWD_ACCUMULATE_GRAD_HIDDEN of gpde_nnconv_bwd, autograd.NNConvHiddenFunction.backward).

`KernelNN.forward` applies ONE conv `depth` times (/root/reference/graph-neural-operator/UAI1_full_resolution.py:29-30); with the
hidden activations H shared as one autograd node, every application's backward produces dL/dH [E, K2P] and autograd adds them one by
one (4 of a 38 ms step at the script's own resolution, s=61).  Now the first application of a backward pass hands autograd its tensor
and the others add to it in the kernel - the same additions in the same order.  Checked:
  * all gradients BITWISE equal to the autograd-summed form (GPDE_ACCUMULATE_DLDH=0), and the path is really taken;
  * a second backward over a retained graph, a partial backward (inputs only) followed by a full one, and an abandoned pass followed by
    a fresh forward + backward: each pass starts its own tensor (keyed on the autograd graph task);
  * the raw call: accumulate == write + add, and the argument checks."""
import pytest
import torch

import graph_pde_amd as gp
from graph_pde_amd import autograd as gpa
from graph_pde_amd import hidden_cache, ops, synth

pytestmark = pytest.mark.gpu
DIMS = [6, 256, 256, 4096]
DEPTH = 4


def _model(seed=0):
    torch.manual_seed(seed)
    mlp = torch.nn.Sequential(*sum([[torch.nn.Linear(DIMS[i], DIMS[i + 1]), torch.nn.ReLU()] for i in range(3)], [])[:-1])
    return gp.NNConv_old(64, 64, mlp, aggr="mean").to("cuda:0")


def _forward(conv, x, ei, ea):
    h = x
    for _ in range(DEPTH):
        h = torch.relu(conv(h, ei, ea))
    return h


def _grads(conv, xin):
    return [xin.grad.clone()] + [p.grad.clone() for p in conv.parameters()]


@pytest.fixture()
def case(monkeypatch):
    monkeypatch.setattr(hidden_cache, "MODE", "on")           # H shared from the first application on
    hidden_cache.clear()
    d = torch.device("cuda:0")
    ei, ea, n = synth.darcy_graph(41, 0.10, device=d)          # 75 k edges, mean in-degree 45
    assert ei.shape[1] >= 32 * n
    return {"conv": _model(), "ei": ei, "ea": ea, "x": torch.randn(n, 64, device=d), "g": torch.randn(n, 64, device=d)}


def _step(c, flag, monkeypatch, passes=1):
    monkeypatch.setattr(gpa, "ACCUMULATE_GRAD_HIDDEN", flag)
    hidden_cache.clear()
    conv = c["conv"]
    conv.zero_grad(set_to_none=True)
    xin = c["x"].clone().requires_grad_(True)
    before = ops.n_grad_hidden_accumulated
    loss = (_forward(conv, xin, c["ei"], c["ea"]) * c["g"]).sum()
    for k in range(passes):
        loss.backward(retain_graph=k + 1 < passes)
    torch.cuda.synchronize()
    return _grads(conv, xin), ops.n_grad_hidden_accumulated - before


def test_in_kernel_sum_of_dldh_is_bitwise_autograds_sum(case, monkeypatch):
    ref, n0 = _step(case, False, monkeypatch)
    acc, n1 = _step(case, True, monkeypatch)
    assert n0 == 0 and n1 == DEPTH - 1                         # the first application writes, the others add
    for a, b in zip(acc, ref):
        assert torch.equal(a, b)
    # a second pass over the retained graph starts its own tensor: .grad doubles exactly as with autograd's sums
    ref2, _ = _step(case, False, monkeypatch, passes=2)
    acc2, n2 = _step(case, True, monkeypatch, passes=2)
    assert n2 == 2 * (DEPTH - 1)
    for a, b in zip(acc2, ref2):
        assert torch.equal(a, b)


def test_partial_and_abandoned_passes_do_not_leak_into_the_next_one(case, monkeypatch):
    ref, _ = _step(case, False, monkeypatch)
    conv = case["conv"]

    def partial_then_full(flag):
        monkeypatch.setattr(gpa, "ACCUMULATE_GRAD_HIDDEN", flag)
        hidden_cache.clear()
        conv.zero_grad(set_to_none=True)
        xin = case["x"].clone().requires_grad_(True)
        loss = (_forward(conv, xin, case["ei"], case["ea"]) * case["g"]).sum()
        # inputs only: the applications' backward passes run, the H node's does not (nothing asks for the hidden layers' gradients)
        (gx,) = torch.autograd.grad(loss, xin, retain_graph=True)
        # ... then everything, over the same graph and the same (still valid) H token.  (The second pass over a retained graph
        # re-aggregates Z - the kept one is released by the first - so it is compared with the same sequence, not with `ref`.)
        loss.backward()
        torch.cuda.synchronize()
        return gx, _grads(conv, xin)
    gx0, seq0 = partial_then_full(False)
    gx1, seq1 = partial_then_full(True)
    assert torch.equal(gx1, ref[0]) and torch.equal(gx0, gx1)
    for a, b in zip(seq1, seq0):
        assert torch.equal(a, b)
    monkeypatch.setattr(gpa, "ACCUMULATE_GRAD_HIDDEN", True)
    hidden_cache.clear()
    # an abandoned pass (an exception half way), then a fresh forward + backward with the same weights: H and its token are reused
    conv.zero_grad(set_to_none=True)
    xin2 = case["x"].clone().requires_grad_(True)
    h = _forward(conv, xin2, case["ei"], case["ea"])

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError("boom")
    h_mid = torch.relu(conv(Boom.apply(torch.relu(conv(xin2, case["ei"], case["ea"]))), case["ei"], case["ea"]))
    with pytest.raises(RuntimeError, match="boom"):
        (h_mid * case["g"]).sum().backward()
    conv.zero_grad(set_to_none=True)
    xin3 = case["x"].clone().requires_grad_(True)
    (_forward(conv, xin3, case["ei"], case["ea"]) * case["g"]).sum().backward()
    torch.cuda.synchronize()
    for a, b in zip(_grads(conv, xin3), ref):
        assert torch.equal(a, b)
    del h


def test_raw_accumulate_is_write_plus_add_and_is_checked(case):
    d = torch.device("cuda:0")
    conv, ei, ea, x, g = case["conv"], case["ei"], case["ea"], case["x"], case["g"]
    csr = ops.build_csr(ei, x.shape[0])
    lin = ops.mlp_linears(conv.nn)
    W, B = [l.weight.detach() for l in lin], [l.bias.detach() for l in lin]
    pm = ops.pack_mlp(W, B)
    H, hmax = ops.hidden_forward_raw(csr, ea, pm, W, B)
    root = conv.root.detach()
    r1 = ops.nnconv_backward_hidden_raw(x, csr, H, DIMS, W[-1], B[-1], root, "mean", g)
    r2 = ops.nnconv_backward_hidden_raw(2 * x, csr, H, DIMS, W[-1], B[-1], root, "mean", 0.5 * g)
    acc = r1[1].clone()
    r3 = ops.nnconv_backward_hidden_raw(2 * x, csr, H, DIMS, W[-1], B[-1], root, "mean", 0.5 * g, grad_hidden_acc=acc)
    torch.cuda.synchronize()
    assert r3[1] is acc and torch.equal(acc, r1[1] + r2[1])
    for a, b in zip(r3[:1] + r3[2:], r2[:1] + r2[2:]):          # everything else is untouched by the flag
        assert (a is None and b is None) or torch.equal(a, b)
    with pytest.raises(ValueError, match="grad_hidden_acc"):
        ops.nnconv_backward_hidden_raw(x, csr, H, DIMS, W[-1], B[-1], root, "mean", g, grad_hidden_acc=torch.zeros(3, 256, device=d))
