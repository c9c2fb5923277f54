"""GPU tier: BASELINE configs 3 and 4 (MGKN-orthogonal Burgers-1D s=8192; MGKN-general Darcy-2D, L = 5) -
every distinct NNConv application of one model forward against the float64 CPU oracle on the same tensors,
and the forward runs the native path for every call (graph_pde_amd/mgkn_workloads.py)."""
import pytest
import torch

from graph_pde_amd import _lib, mgkn_workloads, ops
from oracle.nnconv_oracle import nnconv_forward, rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.mark.parametrize("name", sorted(mgkn_workloads.WORKLOADS))
def test_mgkn_forward_calls_match_oracle(name):
    d = torch.device("cuda:0")
    wl = mgkn_workloads.WORKLOADS[name](d)
    calls = _lib.n_native_calls
    outs = wl.forward()
    torch.cuda.synchronize()
    assert _lib.n_native_calls - calls >= wl.calls            # one native call per NNConv application
    assert all(torch.isfinite(o).all() for o in outs)
    assert wl.calls == {"mgkn_orthogonal_burgers1d": 52, "mgkn_general_darcy2d": 65}[name]
    worst = 0.0
    for conv, x, ei, ea in wl.pairs:
        with torch.no_grad():
            y = conv(x, ei, ea)
        lin = ops.mlp_linears(conv.nn)
        ref = nnconv_forward(x.cpu(), ei.cpu(), ea.cpu(), [l.weight.detach().cpu() for l in lin],
                             [l.bias.detach().cpu() for l in lin],
                             None if conv.root is None else conv.root.detach().cpu(),
                             None if conv.bias is None else conv.bias.detach().cpu(), aggr=conv.aggr,
                             dtype=torch.float64, chunk_edges=8192)
        err = rel_l2(y.cpu(), ref)
        worst = max(worst, err)
        assert err <= TOL, (name, tuple(ei.shape), err)
    print(name, "max rel-L2 over", len(wl.pairs), "NNConv applications:", f"{worst:.2e}")


@pytest.mark.parametrize("mode", ["off", "auto"])
@pytest.mark.parametrize("name", sorted(mgkn_workloads.WORKLOADS))
def test_fused_glue_is_bit_identical_to_the_unfused_composition(name, mode):
    """forward(..., residual=x, activation="relu") == F.relu(x + conv(...)): the same fp32 add and max, done in
    the epilogue kernel (opt-in).  Compared on the same execution path: with the cross-depth cache
    off, and with it on once both models are past their first forward (the very first application of a module runs
    the direct kernel, later ones the cached-H kernels - same value to the last bit or two, not the same bits)."""
    from graph_pde_amd import hidden_cache
    d = torch.device("cuda:0")
    mode0 = hidden_cache.MODE
    hidden_cache.MODE = mode
    hidden_cache.clear()
    try:
        a = mgkn_workloads.WORKLOADS[name](d)
        b = mgkn_workloads.WORKLOADS[name](d, fused_glue=True)
        a.forward(), b.forward()
        for ya, yb in zip(a.forward(), b.forward()):
            assert torch.equal(ya, yb)
    finally:
        hidden_cache.MODE = mode0
        hidden_cache.clear()


def test_fused_glue_with_gradients_composes_the_unfused_operator():
    d = torch.device("cuda:0")
    torch.manual_seed(0)
    conv = mgkn_workloads.NNConv(64, 64, mgkn_workloads.dense_net([6, 32, 4096]), aggr="mean").to(d)
    n, e = 50, 400
    x = torch.randn(n, 64, device=d, requires_grad=True)
    ei = torch.stack([torch.randint(0, n, (e,)), torch.randint(0, n, (e,))]).to(d)
    ea = torch.randn(e, 6, device=d)
    from graph_pde_amd import hidden_cache
    mode0 = hidden_cache.MODE
    hidden_cache.MODE = "off"            # both calls on the direct path (a cached-H second call differs in the last bit)
    try:
        y1 = conv(x, ei, ea, residual=x, activation="relu")
        y2 = torch.relu(x + conv(x, ei, ea))
        assert torch.equal(y1, y2)
        g = torch.randn_like(y1)
        (g1,) = torch.autograd.grad((y1 * g).sum(), x)
        (g2,) = torch.autograd.grad((y2 * g).sum(), x)
        assert torch.equal(g1, g2)       # grad_x is summed per source node in slot order: reproducible to the bit
    finally:
        hidden_cache.MODE = mode0
