"""GPU tier: the backward's ONE-PASS kernel (gpde_fused_f16v6_kernel<2>, csrc/gpde_fused_f16v6.hip).

`loss.backward()` (/root/reference/graph-neural-operator/UAI1_full_resolution.py:266) through `NNConv_old.message`
(nn_conv.py:273-275) needs, per edge, dU_2 = (x_j . dZ_i) [H_2 > 0] and dx_e = H_2 . dZ_i.  Rounds 2-4 recomputed H_2 into
memory (4 KiB per edge) and read it back in a second kernel; the one-pass kernel runs the recompute's K loop with the operands
swapped, so that H_2^T sits in the accumulators with the lane as the edge, and takes both products from there.  It is correct and
complete - and measured SLOWER than the two kernels it replaces (52.6 ms against 32.1 + 17.5 ms at s=121: a one-wave-per-SIMD kernel
hides none of the epilogue's conversions, cross-lane statistics and stores; DESIGN.md §6b, profiles/r05_onepass_ablation.txt), so it
is OPT-IN (GPDE_BWD_ONE_PASS=1; needs the forward's kept Z).  Checked here, through the C ABI:
  * every gradient of the full backward against float64 autograd through the oracle and against the two-pass form
    (the default: recompute-store + gpde_edge_bwd3_kernel) - ragged last tile, destinations with 1 .. 2000 in-edges
    (tiles spanning several nodes), nodes without in-edges, several node / edge chunks, `add` and `mean`;
  * the light pass (dx only) against the full backward's bits for everything it returns;
  * bit-reproducibility run to run, under a different chunking (grad_x), and under workgroup skew;
  * node-table attributes (row f3) bitwise equal to the tensor path."""
import pytest
import torch

from graph_pde_amd import _lib, ops
from oracle.nnconv_oracle import nnconv_grads, rel_l2
from tests.helpers.kinks import edges_off_the_kink

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _case(dims, n, e, seed, aggr="mean"):
    torch.manual_seed(seed)
    dst = torch.randint(0, n - 5, (e,))                      # the last 5 nodes: no in-edges
    dst[: e // 10] = 3                                       # one destination with ~e / 10 in-edges (dozens of tiles)
    dst[e // 10: e // 10 + 7] = 11                           # ... and short runs: several nodes inside one 64-slot tile
    ei = torch.stack([torch.randint(0, n, (e,)), dst])
    ea, x = torch.randn(e, dims[0]), torch.randn(n, 64)
    mlp = torch.nn.Sequential(*sum([[torch.nn.Linear(dims[i], dims[i + 1]), torch.nn.ReLU()] for i in range(3)], [])[:-1])
    W = [l.weight.detach() for l in mlp if isinstance(l, torch.nn.Linear)]
    B = [l.bias.detach() for l in mlp if isinstance(l, torch.nn.Linear)]
    keep = edges_off_the_kink(ea, W, B)
    ei, ea = ei[:, keep].contiguous(), ea[keep].contiguous()
    root = torch.empty(64, 64).uniform_(-0.125, 0.125)
    bias = torch.empty(64).uniform_(-0.125, 0.125)
    return x, ei, ea, W, B, root, bias, torch.randn(n, 64)


def _run(x, ei, ea, W, B, root, bias, gout, aggr="mean", light=False, ws_div=1, edge_attr=None):
    """keep-Z forward, then the backward with the kept Z (the module's training path) - raw calls."""
    d = torch.device("cuda:0")
    n = x.shape[0]
    csr = ops.build_csr(ei.to(d), n)
    Wd, Bd = [w.to(d) for w in W], [b.to(d) for b in B]
    pm = ops.pack_mlp(Wd, Bd)
    ead = ea.to(d) if edge_attr is None else edge_attr
    z = torch.zeros(n, 64 * ops.hidden_width(pm.dims), dtype=torch.float32, device=d)
    ops.nnconv_forward_raw(x.to(d), csr, ead, pm, root.to(d), bias.to(d), aggr, z_keep=z)
    ws = None
    if ws_div > 1:
        dims = [W[0].shape[1]] + [w.shape[0] for w in W]
        full = int(_lib.lib().gpde_nnconv_bwd_workspace_bytes(n, ei.shape[1], 3, _lib.dims_array(dims)))
        ws = torch.empty(full // ws_div, dtype=torch.uint8, device=d)
    if light:
        out = ops.nnconv_backward_light_raw(x.to(d), csr, ead, Wd, Bd, root.to(d), aggr, gout.to(d), z_saved=z)
    else:
        out = ops.nnconv_backward_raw(x.to(d), csr, ead, Wd, Bd, root.to(d), aggr, gout.to(d), ws=ws, z_saved=z)
    torch.cuda.synchronize()
    return out


def _flat(res):
    gx, gW, gb, groot, gbias = res
    return [("dx", gx)] + [(f"dW{l + 1}", w) for l, w in enumerate(gW)] + [(f"db{l + 1}", b) for l, b in enumerate(gb)] + \
        [("droot", groot), ("dbias", gbias)]


@pytest.mark.parametrize("dims,n,e,aggr", [([6, 256, 256, 4096], 200, 20011, "mean"), ([6, 1024, 1024, 4096], 150, 12345, "mean"),
                                           ([4, 512, 384, 4096], 120, 9000, "add"), ([6, 300, 500, 4096], 160, 14000, "mean")])
def test_one_pass_backward_matches_float64_and_the_two_pass_form(dims, n, e, aggr, monkeypatch):
    case = _case(dims, n, e, sum(dims) + e)
    x, ei, ea, W, B, root, bias, gout = case
    rx, rW, rb, rroot, rbias = nnconv_grads(x, ei, ea, W, B, root, bias, aggr, gout, chunk_edges=4096)
    ref = dict([("dx", rx)] + [(f"dW{l + 1}", w) for l, w in enumerate(rW)] + [(f"db{l + 1}", b) for l, b in enumerate(rb)] +
               [("droot", rroot), ("dbias", rbias)])
    monkeypatch.setenv("GPDE_BWD_ONE_PASS", "1")
    one = _flat(_run(*case, aggr=aggr))
    again = _flat(_run(*case, aggr=aggr))
    monkeypatch.delenv("GPDE_BWD_ONE_PASS")
    two = _flat(_run(*case, aggr=aggr))
    assert not torch.equal(one[0][1], two[0][1]), "the switch changed nothing: did the one-pass kernel run?"
    errs = {}
    for (k, a), (_, a2), (_, t) in zip(one, again, two):
        assert torch.equal(a, a2), f"{k}: not bit-reproducible"
        errs[k] = (rel_l2(a.cpu(), ref[k]), rel_l2(t.cpu(), ref[k]), rel_l2(a.cpu(), t.cpu()))
    print(dims, {k: tuple(f"{v:.1e}" for v in vs) for k, vs in errs.items()})
    for k, (e1, e2, e12) in errs.items():
        assert e1 <= TOL, (k, "one-pass vs float64", e1)
        assert e1 <= 3 * e2 + 5e-7, (k, "one-pass", e1, "two-pass", e2)
        assert e12 <= 1e-5, (k, "one-pass vs two-pass", e12)


def test_one_pass_light_pass_and_chunking(monkeypatch):
    dims, n, e = [6, 256, 256, 4096], 300, 40000
    case = _case(dims, n, e, 5)
    monkeypatch.setenv("GPDE_BWD_ONE_PASS", "1")
    fx, fW, fb, froot, fbias = _run(*case)
    lx, lw, lb, lroot, lbias = _run(*case, light=True)
    assert torch.equal(lx, fx) and torch.equal(lw, fW[-1]) and torch.equal(lb, fb[-1])
    assert torch.equal(lroot, froot) and torch.equal(lbias, fbias)
    # several node / edge chunks: grad_x keeps its bits (per-edge partial rows, one owner per element, chunks in order), the
    # weight gradients move by the split-K summation order only
    cx, cW, cb, croot, cbias = _run(*case, ws_div=2)      # (three edge chunks: the call-wide buffers of a 256-wide MLP are a quarter of `full` here)
    assert torch.equal(cx, fx)
    for l in range(3):
        assert rel_l2(cW[l].cpu(), fW[l].cpu()) <= 5e-6 and rel_l2(cb[l].cpu(), fb[l].cpu()) <= 5e-6, l
    # workgroup skew (odd column slices of the GEMMs start late) leaves every bit
    monkeypatch.setenv("GPDE_DEBUG_SKEW_US", "150")
    sx, sW, sb, sroot, sbias = _run(*case)
    monkeypatch.delenv("GPDE_DEBUG_SKEW_US")
    assert torch.equal(sx, fx) and all(torch.equal(a, b) for a, b in zip(sW, fW)) and all(torch.equal(a, b) for a, b in zip(sb, fb))


def test_one_pass_backward_with_node_table_attributes_is_bitwise_the_tensor_path(monkeypatch):
    """Row f3 in training: the attributes come from node data inside the kernel (`GpdeNodeAttr`): same floats, same bits."""
    from graph_pde_amd import synth
    monkeypatch.setenv("GPDE_BWD_ONE_PASS", "1")
    d = torch.device("cuda:0")
    s = 41
    ei, ea, n = synth.darcy_graph(s, 0.10)
    pos, a = synth.lattice_positions(s), synth.darcy_coefficient(s, 0)
    na = ops.NodeAttr.darcy(pos.to(d), a.to(d))
    torch.manual_seed(3)
    dims = [6, 256, 256, 4096]
    mlp = torch.nn.Sequential(*sum([[torch.nn.Linear(dims[i], dims[i + 1]), torch.nn.ReLU()] for i in range(3)], [])[:-1])
    W = [l.weight.detach() for l in mlp if isinstance(l, torch.nn.Linear)]
    B = [l.bias.detach() for l in mlp if isinstance(l, torch.nn.Linear)]
    x, gout = torch.randn(n, 64), torch.randn(n, 64)
    root, bias = torch.empty(64, 64).uniform_(-0.125, 0.125), torch.empty(64).uniform_(-0.125, 0.125)
    ea_t = na.materialize(ei.to(d))                               # the tensor the reference builds (utilities.py:274-277)
    assert rel_l2(ea_t.cpu(), ea) <= 1e-6
    t = _run(x, ei, ea, W, B, root, bias, gout, edge_attr=ea_t)
    v = _run(x, ei, ea, W, B, root, bias, gout, edge_attr=na)
    for (k, p), (_, q) in zip(_flat(t), _flat(v)):
        assert torch.equal(p, q), k
