"""GPU tier: the drop-in boundary beyond forward(cuda tensors) - SURVEY.md §8(b).
CPU tensors (model.cpu() + evaluation, UAI1_full_resolution.py:287-303) are staged through the SAME HIP
kernels; `message` / `update` (nn_conv.py:273-282) exist with the reference signatures and arithmetic;
evaluation under torch.inference_mode works; cache-key contracts of ADVICE r1 are pinned."""
import pytest
import torch

import graph_pde_amd as gp
from graph_pde_amd import _lib, ops
from oracle.nnconv_oracle import nnconv_forward, rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _case(seed=0, n=60, e=700, dims=(6, 64, 128, 4096)):
    torch.manual_seed(seed)
    mods = []
    for i in range(len(dims) - 1):
        mods.append(torch.nn.Linear(dims[i], dims[i + 1]))
        if i != len(dims) - 2:
            mods.append(torch.nn.ReLU())
    conv = gp.NNConv_old(64, 64, torch.nn.Sequential(*mods), aggr="mean")
    x = torch.randn(n, 64)
    ei = torch.stack([torch.randint(0, n, (e,)), torch.randint(0, n, (e,))])
    ea = torch.randn(e, dims[0])
    return conv, x, ei, ea


def _oracle(conv, x, ei, ea, dtype=torch.float64):
    lin = ops.mlp_linears(conv.nn)
    return nnconv_forward(x, ei, ea, [l.weight.detach().cpu() for l in lin], [l.bias.detach().cpu() for l in lin],
                          None if conv.root is None else conv.root.detach().cpu(),
                          None if conv.bias is None else conv.bias.detach().cpu(), aggr=conv.aggr, dtype=dtype)


def test_cpu_tensors_run_the_hip_kernels_and_come_back():
    conv, x, ei, ea = _case()
    calls = _lib.n_native_calls
    with torch.no_grad():
        y_cpu = conv(x, ei, ea)                      # module and inputs on the CPU
    assert _lib.n_native_calls > calls              # the native entry point ran
    assert y_cpu.device.type == "cpu"
    assert rel_l2(y_cpu, _oracle(conv, x, ei, ea)) <= TOL
    d = torch.device("cuda:0")
    conv_d = torch.nn.Module.to(conv, d)
    with torch.no_grad():
        y_gpu = conv_d(x.to(d), ei.to(d), ea.to(d))
    assert torch.equal(y_gpu.cpu(), y_cpu)          # same kernels, same bits
    conv_d.cpu()


def test_cpu_tensors_gradients_reach_the_cpu_parameters():
    conv, x, ei, ea = _case(seed=1)
    x.requires_grad_(True)
    y = conv(x, ei, ea)
    g = torch.randn_like(y)
    (y * g).sum().backward()
    conv64, x64 = _case(seed=1)[0].double(), x.detach().double().requires_grad_(True)
    lin = [l for l in conv64.nn if isinstance(l, torch.nn.Linear)]
    h = ea.double()
    for i, l in enumerate(lin):
        h = l(h)
        if i != len(lin) - 1:
            h = torch.relu(h)
    m = torch.matmul(x64[ei[0]].unsqueeze(1), h.view(-1, 64, 64)).squeeze(1)
    out = torch.zeros(x.shape[0], 64, dtype=torch.float64).index_add(0, ei[1], m)
    out = out / torch.bincount(ei[1], minlength=x.shape[0]).clamp(min=1).double().unsqueeze(1)
    out = out + x64 @ conv64.root + conv64.bias
    (out * g.double()).sum().backward()
    assert x.grad.device.type == "cpu" and conv.root.grad.device.type == "cpu"
    assert rel_l2(x.grad, x64.grad) <= 2e-5
    assert rel_l2(conv.root.grad, conv64.root.grad) <= 2e-5
    lin32 = ops.mlp_linears(conv.nn)
    for a, b in zip(lin32, lin):
        assert rel_l2(a.weight.grad, b.weight.grad) <= 2e-5


@pytest.mark.parametrize("on_gpu", [True, False])
def test_message_and_update_match_the_reference_arithmetic(on_gpu):
    conv, x, ei, ea = _case(seed=2, n=40, e=300)
    d = torch.device("cuda:0") if on_gpu else torch.device("cpu")
    if on_gpu:
        conv = conv.to(d)
    x_j = x[ei[0]]
    with torch.no_grad():
        m = conv.message(x_j.to(d), ea.to(d))
        u = conv.update(torch.ones(x.shape[0], 64, device=d), x.to(d))
    assert m.device.type == d.type and u.device.type == d.type
    # reference: weight = nn(pseudo).view(-1, in, out); matmul(x_j.unsqueeze(1), weight).squeeze(1)
    c64 = _case(seed=2, n=40, e=300)[0].double()
    with torch.no_grad():
        w = c64.nn(ea.double()).view(-1, 64, 64)
        m_ref = torch.matmul(x_j.double().unsqueeze(1), w).squeeze(1)
        u_ref = torch.ones(x.shape[0], 64, dtype=torch.float64) + x.double() @ c64.root + c64.bias
    assert rel_l2(m.cpu(), m_ref) <= TOL
    assert rel_l2(u.cpu(), u_ref) <= TOL


def test_message_aggregate_update_compose_to_forward():
    """propagate = gather -> message -> scatter-mean -> update."""
    conv, x, ei, ea = _case(seed=3)
    d = torch.device("cuda:0")
    conv = conv.to(d)
    xd, eid, ead = x.to(d), ei.to(d), ea.to(d)
    with torch.no_grad():
        m = conv.message(xd[eid[0]], ead)
        agg = torch.zeros(x.shape[0], 64, device=d).index_add_(0, eid[1], m)
        agg = agg / torch.bincount(eid[1], minlength=x.shape[0]).clamp(min=1).unsqueeze(1)
        composed = conv.update(agg, xd)
        fused = conv(xd, eid, ead)
    assert rel_l2(composed.cpu(), fused.cpu()) <= 1e-6


def test_inference_mode_evaluation():
    conv, x, ei, ea = _case(seed=4)
    d = torch.device("cuda:0")
    conv = conv.to(d)
    with torch.inference_mode():
        xd, eid, ead = x.to(d), ei.to(d), ea.to(d)       # inference tensors: no version counter
        y1 = conv(xd, eid, ead)
        y2 = conv(xd, eid, ead)
    assert torch.equal(y1, y2)
    assert rel_l2(y1.cpu(), _oracle(conv, x, ei, ea)) <= TOL


def test_data_writes_need_clear_caches_and_versioned_writes_do_not():
    """the packed-weight cache is keyed on (address, version).  In-place writes under no_grad move
    the version (optimizers, parallel.broadcast_parameters); writes through `.data` do not and need
    ops.clear_caches()."""
    conv, x, ei, ea = _case(seed=5)
    d = torch.device("cuda:0")
    conv = conv.to(d)
    xd, eid, ead = x.to(d), ei.to(d), ea.to(d)
    lin = ops.mlp_linears(conv.nn)
    with torch.no_grad():
        y0 = conv(xd, eid, ead)
        lin[1].weight.mul_(1.5)                          # versioned write: seen
        y1 = conv(xd, eid, ead)
    assert rel_l2(y1.cpu(), _oracle(conv, x, ei, ea)) <= TOL and not torch.equal(y0, y1)
    with torch.no_grad():
        lin[1].weight.data.mul_(2.0)                     # .data write: NOT seen until the caches are cleared
        y_stale = conv(xd, eid, ead)
        assert torch.equal(y_stale, y1)
        ops.clear_caches()
        from graph_pde_amd import hidden_cache
        hidden_cache.clear()
        y2 = conv(xd, eid, ead)
    assert rel_l2(y2.cpu(), _oracle(conv, x, ei, ea)) <= TOL


def test_pack_cache_keeps_one_entry_per_parameter_set():
    conv, x, ei, ea = _case(seed=6)
    d = torch.device("cuda:0")
    conv = conv.to(d)
    lin = ops.mlp_linears(conv.nn)
    ops.clear_caches()
    for _ in range(5):
        with torch.no_grad():
            lin[0].weight.add_(1e-3)                     # what an optimizer step does
        ops.pack_mlp([l.weight for l in lin], [l.bias for l in lin])
    assert len(ops._pack_cache) == 1


def test_backward_is_once_differentiable():
    conv, x, ei, ea = _case(seed=7)
    d = torch.device("cuda:0")
    conv = conv.to(d)
    xd = x.to(d).requires_grad_(True)
    y = conv(xd, ei.to(d), ea.to(d))
    (gx,) = torch.autograd.grad(y.sum(), xd, create_graph=True)
    with pytest.raises(RuntimeError):
        gx.sum().backward()                              # double backward must raise, not return wrong numbers
