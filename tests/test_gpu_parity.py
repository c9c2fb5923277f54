"""GPU tier: the HIP path (through the C ABI of libgpde.so) against the reference vectors and the
CPU oracle.  Tolerance: BASELINE.json north_star = 1e-5 relative L2 (LpLoss.rel over the whole
[N,64] output); the float64 reference output is the adjudicator."""
import numpy as np
import pytest
import torch

import graph_pde_amd as gp
from graph_pde_amd import _lib, ops, synth
from oracle.nnconv_oracle import nnconv_forward, rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-5


def dev():
    assert torch.cuda.is_available(), "GPU tier needs an MI355X"
    return torch.device("cuda:0")


def run_native(x, ei, ea, weights, biases, root, bias, aggr, ws_bytes=None, precision=None):
    d = dev()
    calls = _lib.n_native_calls
    csr = ops.build_csr(ei.to(d), x.shape[0])
    pm = ops.pack_mlp([w.to(d) for w in weights], [b.to(d) for b in biases])
    ws = None
    if ws_bytes is not None:
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=d)
    y = ops.nnconv_forward_raw(x.to(d), csr, ea.to(d), pm, None if root is None else root.to(d),
                               None if bias is None else bias.to(d), aggr, ws=ws, precision=precision)
    torch.cuda.synchronize()
    assert _lib.n_native_calls == calls + 1          # the HIP entry point really ran
    return y.cpu()


def test_golden_vectors(golden):
    g = golden
    y = run_native(g["x"], g["edge_index"], g["edge_attr"], g["weights"], g["biases"], g["root"],
                   g["bias"], g["aggr"])
    assert torch.isfinite(y).all()
    e64, e32 = rel_l2(y, g["out_f64"]), rel_l2(y, g["out_f32"])
    assert e64 <= TOL and e32 <= TOL, (g["name"], e64, e32)
    # fp32 round-off class: no worse than a few times the reference's own fp32-vs-fp64 distance
    assert e64 <= 10 * max(rel_l2(g["out_f32"], g["out_f64"]), 1e-7), (g["name"], e64)


def test_csr_is_stable_and_complete():
    d = dev()
    torch.manual_seed(3)
    n, e = 300, 5000
    ei = torch.stack([torch.randint(0, n, (e,)), torch.randint(0, n - 20, (e,))])
    csr = ops.build_csr(ei.to(d), n)
    rowptr, src, dst, perm = (t.cpu().long() for t in (csr.rowptr, csr.src, csr.dst, csr.perm))
    assert rowptr[0] == 0 and rowptr[-1] == e
    assert torch.equal(torch.sort(perm).values, torch.arange(e))          # a permutation
    assert torch.equal(dst, ei[1][perm]) and torch.equal(src, ei[0][perm])
    assert bool((dst[1:] >= dst[:-1]).all())                               # sorted by target
    same = dst[1:] == dst[:-1]
    assert bool((perm[1:][same] > perm[:-1][same]).all())                  # stable within a target
    assert torch.equal(rowptr[1:] - rowptr[:-1], torch.bincount(ei[1], minlength=n))
    # strided view, as the MGKN scripts pass (edge_index[:, a:b])
    view = ei.to(d)[:, 100:3000]
    c2 = ops.build_csr(view, n)
    assert torch.equal(c2.dst.cpu().long(), ei[1, 100:3000][c2.perm.cpu().long()])
    with pytest.raises(IndexError):
        bad = ei.clone(); bad[1, 7] = n + 3
        ops.build_csr(bad.to(d), n)


@pytest.mark.parametrize("dims,aggr,use_root,use_bias", [
    ([6, 64, 4096], "mean", False, False),             # MGKN inter-level kernel
    ([6, 100, 200, 4096], "mean", True, True),          # non-multiple-of-tile widths
    ([4, 32, 32, 4096], "add", True, False),            # Burgers attributes
    ([6, 16, 32, 48, 4096], "mean", True, True),        # 4 Linear layers -> dense front layers
    ([6, 256, 256, 4096], "mean", True, True),          # secondary bench MLP
])
def test_random_graphs_against_oracle(dims, aggr, use_root, use_bias):
    torch.manual_seed(sum(dims))
    n, e = 500, 9000
    ei = torch.stack([torch.randint(0, n, (e,)), torch.randint(0, n, (e,))])
    ei[1, :700] = 17                                      # one high in-degree node (several tiles)
    ea = torch.randn(e, dims[0])
    x = torch.randn(n, 64)
    mlp = torch.nn.Sequential(*sum([[torch.nn.Linear(dims[i], dims[i + 1]), torch.nn.ReLU()]
                                    for i in range(len(dims) - 1)], [])[:-1])
    ws_ = [l.weight.detach() for l in mlp if isinstance(l, torch.nn.Linear)]
    bs_ = [l.bias.detach() for l in mlp if isinstance(l, torch.nn.Linear)]
    root = torch.empty(64, 64).uniform_(-0.125, 0.125) if use_root else None
    bias = torch.empty(64).uniform_(-0.125, 0.125) if use_bias else None
    y = run_native(x, ei, ea, ws_, bs_, root, bias, aggr)
    y64 = nnconv_forward(x, ei, ea, ws_, bs_, root, bias, aggr=aggr, dtype=torch.float64)
    y32 = nnconv_forward(x, ei, ea, ws_, bs_, root, bias, aggr=aggr, dtype=torch.float32)
    assert rel_l2(y, y64) <= TOL and rel_l2(y, y32) <= TOL, (rel_l2(y, y64), rel_l2(y, y32))


def test_darcy_lattice_g31_full_mlp():
    """Darcy-shaped lattice (s=31, r=0.10: 25,673 edges) with the headline kernel MLP
    DenseNet([6,1024,1024,4096]) (UAI1_full_resolution.py:21,57)."""
    torch.manual_seed(0)
    ei, ea, n = synth.darcy_graph(31, 0.10)
    mlp = torch.nn.Sequential(torch.nn.Linear(6, 1024), torch.nn.ReLU(), torch.nn.Linear(1024, 1024),
                              torch.nn.ReLU(), torch.nn.Linear(1024, 4096))
    ws_ = [l.weight.detach() for l in mlp if isinstance(l, torch.nn.Linear)]
    bs_ = [l.bias.detach() for l in mlp if isinstance(l, torch.nn.Linear)]
    root = torch.empty(64, 64).uniform_(-0.125, 0.125)
    bias = torch.empty(64).uniform_(-0.125, 0.125)
    x = torch.randn(n, 64)
    y = run_native(x, ei, ea, ws_, bs_, root, bias, "mean")
    y64 = nnconv_forward(x, ei, ea, ws_, bs_, root, bias, aggr="mean", dtype=torch.float64,
                         chunk_edges=4096)
    assert rel_l2(y, y64) <= TOL, rel_l2(y, y64)


def test_properties_linearity_chunking_determinism():
    """Size-independent properties: linear in x (no root/bias), invariant to the workspace
    (= node-chunk) size and to a permutation of the input edge list up to fp32 round-off,
    bit-identical run to run (no atomics)."""
    torch.manual_seed(5)
    ei, ea, n = synth.darcy_graph(24, 0.12)
    dims = [6, 96, 160, 4096]
    mlp = torch.nn.Sequential(torch.nn.Linear(6, 96), torch.nn.ReLU(), torch.nn.Linear(96, 160),
                              torch.nn.ReLU(), torch.nn.Linear(160, 4096))
    ws_ = [l.weight.detach() for l in mlp if isinstance(l, torch.nn.Linear)]
    bs_ = [l.bias.detach() for l in mlp if isinstance(l, torch.nn.Linear)]
    x1, x2 = torch.randn(n, 64), torch.randn(n, 64)
    f = lambda x, **kw: run_native(x, ei, ea, ws_, bs_, None, None, "mean", **kw)
    y1, y2, y12 = f(x1), f(x2), f(2.0 * x1 - 0.5 * x2)
    assert rel_l2(y12, 2.0 * y1 - 0.5 * y2) <= 5e-6
    assert torch.equal(f(x1), y1)                                          # deterministic
    # a workspace that forces several node chunks gives the same answer
    pm_bytes = 64 * 256 * 4 + 64 * 64 * 4
    small = f(x1, ws_bytes=70 * pm_bytes + 4096)
    assert rel_l2(small, y1) <= 1e-6
    # permuting the edge list changes only the summation order inside a destination
    p = torch.randperm(ei.shape[1])
    yp = run_native(x1, ei[:, p], ea[p], ws_, bs_, None, None, "mean")
    assert rel_l2(yp, y1) <= 2e-6
    # zero in-degree nodes: rows receive root/bias only
    ei2 = ei[:, ei[1] >= 10]
    ea2 = ea[ei[1] >= 10]
    root = torch.empty(64, 64).uniform_(-0.125, 0.125)
    bias = torch.empty(64).uniform_(-0.125, 0.125)
    y = run_native(x1, ei2, ea2, ws_, bs_, root, bias, "mean")
    assert torch.allclose(y[:10], x1[:10] @ root + bias, atol=1e-5)


def test_module_forward_drop_in():
    """The nn.Module surface end to end on the GPU (what the GKN scripts call)."""
    from tests.test_host_logic import DenseNet
    from tests.conftest import load_golden
    g = load_golden("ckpt_g16")
    d = dev()
    conv = gp.NNConv_old(64, 64, DenseNet([6, 64, 128, 4096], torch.nn.ReLU), aggr="mean")
    sd = {"root": g["root"], "bias": g["bias"]}
    for i, k in enumerate((0, 2, 4)):
        sd[f"nn.layers.{k}.weight"] = g["weights"][i]
        sd[f"nn.layers.{k}.bias"] = g["biases"][i]
    conv.load_state_dict(sd)
    conv = conv.to(d)
    with torch.no_grad():
        y = conv(g["x"].to(d), g["edge_index"].to(d), g["edge_attr"].to(d))
        y_again = conv(g["x"].to(d), g["edge_index"].to(d), g["edge_attr"].to(d))   # cached CSR/pack
    assert rel_l2(y.cpu(), g["out_f64"]) <= TOL
    assert torch.equal(y, y_again)
    # aggr='max' on a freshly built module in grad mode (parameters require grad): PyG's chain on the native message() /
    # update() (round 4; rounds 1-3 raised) - the value of the fused inference kernel
    cmax = gp.NNConv_old(64, 64, DenseNet([6, 8, 4096], torch.nn.ReLU), aggr="max").to(d)
    y_g = cmax(g["x"].to(d), g["edge_index"].to(d), g["edge_attr"].to(d))
    with torch.no_grad():
        y_n = cmax(g["x"].to(d), g["edge_index"].to(d), g["edge_attr"].to(d))
    assert y_g.requires_grad and rel_l2(y_g.detach().cpu(), y_n.cpu()) <= 2e-6


F16_VARIANTS = ["f16split", "f16split_agg16", "f16split_agg32", "f16split_8wave"]   # default + forced aggregation arithmetic / kernel


@pytest.mark.parametrize("variant", F16_VARIANTS)
@pytest.mark.parametrize("name", ["ckpt_g16", "burgers_k4"])
def test_f16split_hidden_layer_golden(name, variant):
    """GPDE_FWD_F16SPLIT (hidden layer on f16 MFMA with two-term split operands, fp32 accumulate)
    against the reference vectors: same 1e-5 bar, and within a small factor of the fp32-MFMA path."""
    from tests.conftest import load_golden
    g = load_golden(name)
    args = (g["x"], g["edge_index"], g["edge_attr"], g["weights"], g["biases"], g["root"], g["bias"], g["aggr"])
    y16 = run_native(*args, precision=variant)
    y32 = run_native(*args, precision="f32")
    e16, e32 = rel_l2(y16, g["out_f64"]), rel_l2(y32, g["out_f64"])
    assert e16 <= TOL and e16 <= 4 * e32 + 2e-7, (name, variant, e16, e32)


@pytest.mark.parametrize("variant", F16_VARIANTS)
def test_f16split_wide_dynamic_range(variant):
    """Attributes and weights spanning many binades (per-edge and per-row power-of-two scaling)."""
    torch.manual_seed(11)
    n, e = 400, 8000
    ei = torch.stack([torch.randint(0, n, (e,)), torch.randint(0, n, (e,))])
    ea = torch.randn(e, 6) * torch.logspace(-3, 3, e).unsqueeze(1)       # edges from 1e-3 to 1e3
    x = torch.randn(n, 64)
    mlp = torch.nn.Sequential(torch.nn.Linear(6, 128), torch.nn.ReLU(), torch.nn.Linear(128, 256),
                              torch.nn.ReLU(), torch.nn.Linear(256, 4096))
    with torch.no_grad():
        mlp[2].weight.mul_(torch.logspace(-4, 4, 256).unsqueeze(1))      # rows from 1e-4 to 1e4
        mlp[2].weight[:, ::7] *= 1e-3                                    # small entries inside rows
    ws_ = [l.weight.detach() for l in mlp if isinstance(l, torch.nn.Linear)]
    bs_ = [l.bias.detach() for l in mlp if isinstance(l, torch.nn.Linear)]
    y64 = nnconv_forward(x, ei, ea, ws_, bs_, None, None, aggr="mean", dtype=torch.float64)
    y16 = run_native(x, ei, ea, ws_, bs_, None, None, "mean", precision=variant)
    y32 = run_native(x, ei, ea, ws_, bs_, None, None, "mean", precision="f32")
    e16, e32 = rel_l2(y16, y64), rel_l2(y32, y64)
    assert torch.isfinite(y16).all()
    assert e16 <= TOL and e16 <= 4 * e32 + 2e-7, (variant, e16, e32)


def _oracle_grads(x, ei, ea, ws_, bs_, root, bias, aggr, gout):
    """float64 autograd through the CPU oracle = the reference's backward (same as
    oracle.nnconv_oracle.nnconv_grads, which tests/golden/*_grad.npz pin to the reference module)."""
    xs = x.double().requires_grad_(True)
    Ws = [w.double().requires_grad_(True) for w in ws_]
    Bs = [b.double().requires_grad_(True) for b in bs_]
    r = None if root is None else root.double().requires_grad_(True)
    bb = None if bias is None else bias.double().requires_grad_(True)
    src, dst = ei[0], ei[1]
    h = ea.double()
    for l in range(len(Ws)):
        h = torch.nn.functional.linear(h, Ws[l], Bs[l])
        if l != len(Ws) - 1:
            h = torch.relu(h)
    m = torch.matmul(xs[src].unsqueeze(1), h.view(-1, 64, 64)).squeeze(1)
    out = torch.zeros(x.shape[0], 64, dtype=torch.float64).index_add(0, dst, m)
    if aggr == "mean":
        out = out / torch.bincount(dst, minlength=x.shape[0]).clamp(min=1).double().unsqueeze(1)
    if r is not None:
        out = out + xs @ r
    if bb is not None:
        out = out + bb
    (out * gout.double()).sum().backward()
    return xs.grad, [w.grad for w in Ws], [b.grad for b in Bs], None if r is None else r.grad, None if bb is None else bb.grad


@pytest.mark.parametrize("dims,aggr,use_root,use_bias", [
    ([6, 40, 72, 4096], "mean", True, True),            # 3 Linear layers, non-tile widths
    ([6, 48, 4096], "mean", False, False),              # 2 Linear layers (MGKN inter-level)
    ([4, 24, 40, 56, 4096], "add", True, True),         # 4 Linear layers, Burgers attributes
])
@pytest.mark.parametrize("edge_kernel", ["1", "2", "3"])      # per-MFMA operands / staged through LDS (fp32 MFMA) / staged, split-f16 MFMA
def test_backward_against_reference_autograd(dims, aggr, use_root, use_bias, edge_kernel, monkeypatch):
    """gpde_nnconv_bwd vs float64 autograd through the oracle (the reference's loss.backward())."""
    monkeypatch.setenv("GPDE_EDGE_BWD", edge_kernel)
    d = dev()
    torch.manual_seed(100 + sum(dims))
    n, e = 300, 6000
    ei = torch.stack([torch.randint(0, n, (e,)), torch.randint(0, n - 15, (e,))])
    ei[1, :500] = 11                                       # one node spanning many tiles
    ea, x = torch.randn(e, dims[0]), torch.randn(n, 64)
    mlp = torch.nn.Sequential(*sum([[torch.nn.Linear(dims[i], dims[i + 1]), torch.nn.ReLU()]
                                    for i in range(len(dims) - 1)], [])[:-1])
    ws_ = [l.weight.detach() for l in mlp if isinstance(l, torch.nn.Linear)]
    bs_ = [l.bias.detach() for l in mlp if isinstance(l, torch.nn.Linear)]
    root = torch.empty(64, 64).uniform_(-0.125, 0.125) if use_root else None
    bias = torch.empty(64).uniform_(-0.125, 0.125) if use_bias else None
    gout = torch.randn(n, 64)
    rx, rW, rb, rroot, rbias = _oracle_grads(x, ei, ea, ws_, bs_, root, bias, aggr, gout)
    csr = ops.build_csr(ei.to(d), n)
    gx, gW, gb, groot, gbias = ops.nnconv_backward_raw(
        x.to(d), csr, ea.to(d), [w.to(d) for w in ws_], [b.to(d) for b in bs_],
        None if root is None else root.to(d), aggr, gout.to(d), need_root=use_root, need_bias=use_bias)
    torch.cuda.synchronize()
    tol = 2e-5
    assert rel_l2(gx.cpu(), rx) <= tol, ("dx", rel_l2(gx.cpu(), rx))
    for l in range(len(ws_)):
        assert rel_l2(gW[l].cpu(), rW[l]) <= tol, (f"dW{l}", rel_l2(gW[l].cpu(), rW[l]))
        assert rel_l2(gb[l].cpu(), rb[l]) <= tol, (f"db{l}", rel_l2(gb[l].cpu(), rb[l]))
    if use_root:
        assert rel_l2(groot.cpu(), rroot) <= tol
    if use_bias:
        assert rel_l2(gbias.cpu(), rbias) <= tol


@pytest.mark.parametrize("name", ["ragged_add", "mlp2_mean_noroot", "burgers_k4", "ckpt_torus_m100"])
def test_backward_against_reference_module_gradients(name):
    """gpde_nnconv_bwd against tests/golden/<name>_grad.npz: float64 autograd through the reference's OWN
    NNConv_old / DenseNet classes (make_golden.py) on the golden inputs."""
    from tests.conftest import load_golden
    from tests.test_oracle_golden import load_golden_grads
    d = dev()
    g, r = load_golden(name), load_golden_grads(name)
    n = g["x"].shape[0]
    csr = ops.build_csr(g["edge_index"].to(d), n)
    gx, gW, gb, groot, gbias = ops.nnconv_backward_raw(
        g["x"].to(d), csr, g["edge_attr"].to(d), [w.to(d) for w in g["weights"]], [b.to(d) for b in g["biases"]],
        None if g["root"] is None else g["root"].to(d), g["aggr"], r["gout"].to(d),
        need_root=g["root"] is not None, need_bias=g["bias"] is not None)
    torch.cuda.synchronize()
    tol = 2e-5
    assert rel_l2(gx.cpu(), r["gx"]) <= tol, rel_l2(gx.cpu(), r["gx"])
    for l in range(len(gW)):
        assert rel_l2(gW[l].cpu(), r["gW"][l]) <= tol, (l, rel_l2(gW[l].cpu(), r["gW"][l]))
        assert rel_l2(gb[l].cpu(), r["gb"][l]) <= tol, l
    if r["groot"] is not None:
        assert rel_l2(groot.cpu(), r["groot"]) <= tol
    if r["gbias"] is not None:
        assert rel_l2(gbias.cpu(), r["gbias"]) <= tol


def test_module_training_step_matches_reference():
    """loss.backward() + Adam step through the drop-in module (what the GKN scripts do,
    UAI1_full_resolution.py:258-273) vs the same step on the float64 oracle."""
    from tests.test_host_logic import DenseNet
    d = dev()
    torch.manual_seed(7)
    ei, ea, n = synth.darcy_graph(12, 0.2)
    x0 = torch.randn(n, 64)
    conv = gp.NNConv_old(64, 64, DenseNet([6, 32, 64, 4096], torch.nn.ReLU), aggr="mean")
    lin = ops.mlp_linears(conv.nn)
    ws_ = [l.weight.detach().clone() for l in lin]
    bs_ = [l.bias.detach().clone() for l in lin]
    root, bias = conv.root.detach().clone(), conv.bias.detach().clone()
    y = torch.randn(n, 64)
    conv = conv.to(d)
    xg = x0.to(d).requires_grad_(True)
    out = conv(xg, ei.to(d), ea.to(d))
    out2 = conv(torch.relu(out), ei.to(d), ea.to(d))       # depth 2 with shared weights (UAI1:29-30)
    loss = ((out2 - y.to(d)) ** 2).mean()
    loss.backward()
    # reference: same graph in float64 autograd
    xs = x0.double().requires_grad_(True)
    Ws = [w.double().requires_grad_(True) for w in ws_]
    Bs = [b.double().requires_grad_(True) for b in bs_]
    r, bb = root.double().requires_grad_(True), bias.double().requires_grad_(True)

    def ref_conv(xin):
        h = ea.double()
        for l in range(3):
            h = torch.nn.functional.linear(h, Ws[l], Bs[l])
            if l != 2:
                h = torch.relu(h)
        m = torch.matmul(xin[ei[0]].unsqueeze(1), h.view(-1, 64, 64)).squeeze(1)
        o = torch.zeros(n, 64, dtype=torch.float64).index_add(0, ei[1], m)
        o = o / torch.bincount(ei[1], minlength=n).clamp(min=1).double().unsqueeze(1)
        return o + xin @ r + bb
    lref = ((ref_conv(torch.relu(ref_conv(xs))) - y.double()) ** 2).mean()
    lref.backward()
    assert abs(float(loss) - float(lref)) <= 1e-5 * abs(float(lref))
    assert rel_l2(xg.grad.cpu(), xs.grad) <= 2e-5
    assert rel_l2(conv.root.grad.cpu(), r.grad) <= 2e-5 and rel_l2(conv.bias.grad.cpu(), bb.grad) <= 2e-5
    for l, k in enumerate((0, 2, 4)):
        assert rel_l2(conv.nn.layers[k].weight.grad.cpu(), Ws[l].grad) <= 2e-5, l
        assert rel_l2(conv.nn.layers[k].bias.grad.cpu(), Bs[l].grad) <= 2e-5, l
    opt = torch.optim.Adam(conv.parameters(), lr=1e-3, weight_decay=5e-4)
    opt.step()                                              # parameters updated in place -> repack
    with torch.no_grad():
        conv(x0.to(d), ei.to(d), ea.to(d))


def _dense(dims):
    from tests.test_host_logic import DenseNet
    return DenseNet(dims, torch.nn.ReLU)


def _check_module(conv, x, ei, ea, tol=TOL):
    d = dev()
    lin = ops.mlp_linears(conv.nn)
    ws_ = [l.weight.detach().cpu() for l in lin]
    bs_ = [l.bias.detach().cpu() for l in lin]
    root = None if conv.root is None else conv.root.detach().cpu()
    bias = None if conv.bias is None else conv.bias.detach().cpu()
    conv = conv.to(d)
    with torch.no_grad():
        y = conv(x.to(d), ei.to(d), ea.to(d)).cpu()
    y64 = nnconv_forward(x, ei, ea, ws_, bs_, root, bias, aggr=conv.aggr, dtype=torch.float64,
                         chunk_edges=8192)
    err = rel_l2(y, y64)
    assert err <= tol, err
    return err


def test_mgkn_orthogonal_burgers_shapes():
    """BASELINE config 3: MGKN-orthogonal Burgers, s = 8192: one NNConv per multipole level with
    kernel DenseNet([4, kl, kl, 4096]), kl = max(1024 // 2^l, 16), aggr='mean', root + bias
    (MGKN_orthogonal_burgers1d.py:33-37, 74-82) on the restated multi_pole_grid1d graphs."""
    torch.manual_seed(21)
    graphs = synth.burgers_multipole_graphs(8192)
    assert len(graphs) == 13
    for l in (0, 1, 2, 5, 9, 12):
        ei, ea, n = graphs[l]
        kl = max(1024 // (2 ** l), 16)
        conv = gp.NNConv(64, 64, _dense([4, kl, kl, 4096]), aggr="mean")
        _check_module(conv, torch.randn(n, 64), ei, ea)


def test_mgkn_general_darcy_shapes():
    """BASELINE config 4 graph family: multi-level sampled radius graphs (m = [2400,1600,400,100,25],
    radii of neurips1_MGKN.py:112-114) with the MKGN kernels: inner K_ll = DenseNet([6,kl,kl,4096])
    (root, no bias), inter-level K_{l,l+1} / K_{l+1,l} = DenseNet([6,kl,4096]) (no root, no bias),
    kl = 256 // 2^l (MGKN_general_darcy2d.py:41-61), inter graphs on the shared node index space."""
    torch.manual_seed(22)
    m = [2400, 1600, 400, 100, 25]
    g = synth.sampled_multilevel_graphs(141, m, [0.5 / 8 * 1.41, 0.5 / 8, 0.5 / 4, 0.5 / 2, 0.5],
                                        [0.5 / 8 * 1.1, 0.5 / 8 * 1.41, 0.5 / 4 * 1.41, 0.5 / 2 * 1.41])
    offs = [0]
    for ml in m:
        offs.append(offs[-1] + ml)
    ntot = offs[-1]
    xall = torch.randn(ntot, 64)
    for l in (0, 2, 4):
        ei, ea, n, _ = g["inner"][l]
        kl = 256 // (2 ** l)
        conv = gp.NNConv(64, 64, _dense([6, kl, kl, 4096]), aggr="mean", root_weight=True, bias=False)
        _check_module(conv, xall[offs[l]:offs[l + 1]].clone(), ei, ea)
    for l in (0, 3):
        kl = 256 // (2 ** l)
        ei, ea, _, _ = g["down"][l]
        ei = torch.stack([ei[0] + offs[l], ei[1] + offs[l + 1]])
        conv = gp.NNConv(64, 64, _dense([6, kl, 4096]), aggr="mean", root_weight=False, bias=False)
        _check_module(conv, xall, ei, ea)
        ei, ea, _, _ = g["up"][l]
        ei = torch.stack([ei[0] + offs[l + 1], ei[1] + offs[l]])
        conv = gp.NNConv(64, 64, _dense([6, kl, 4096]), aggr="mean", root_weight=False, bias=False)
        _check_module(conv, xall, ei, ea)


def test_native_radius_graph_matches_reference_construction():
    """gpde_radius_graph_* vs the reference construction (dense pairwise distances + np.where,
    utilities.py:250-255) on random points (no ties at the radius) and vs the exact lattice."""
    import numpy as np
    from sklearn.metrics import pairwise_distances
    d = dev()
    rng = np.random.default_rng(5)
    for n, dim, r in ((700, 2, 0.11), (300, 1, 0.03), (400, 3, 0.25)):
        pos = rng.random((n, dim))
        pwd = pairwise_distances(pos)
        ref = np.vstack(np.where(pwd <= r))                              # the reference's lines
        ei = ops.radius_graph(torch.from_numpy(pos).to(d), r).cpu().numpy()
        assert ei.shape == ref.shape and (ei == ref).all(), (n, dim, ei.shape, ref.shape)
    # the reference's own SquareMeshGenerator output (tests/golden/mesh_s12.npz), edge order included
    import os
    from tests.conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "mesh_s12.npz"))
    ei = ops.radius_graph(torch.from_numpy(g["grid"]).to(d), float(g["r"])).cpu().numpy()
    assert np.array_equal(ei, g["edge_index"])
    # exact-arithmetic lattice (synth) == native builder with a radius safely between lattice shells
    s = 31
    ei_lat = synth.lattice_radius_graph(s, 0.10)
    pos = synth.lattice_positions(s)
    ei = ops.radius_graph(pos.to(d), 0.10 + 1e-9).cpu()
    assert torch.equal(ei, ei_lat)
