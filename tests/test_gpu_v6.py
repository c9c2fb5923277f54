"""GPU tier: the one-wave-per-SIMD fused kernel (gpde_fused_f16v6_kernel, csrc/gpde_fused_f16v6.hip) --
the default from 32,768 edges on -- against the float64 CPU oracle, against the 8-wave kernel it
replaces on those graphs, and through the properties the summation structure offers (linearity in x,
invariance to node chunking, bit determinism).  Tolerance: 1e-5 relative L2 (BASELINE.json)."""
import pytest
import torch

import graph_pde_amd as gp
from graph_pde_amd import _lib, ops, synth
from oracle.nnconv_oracle import nnconv_forward, rel_l2
from tests.test_gpu_parity import run_native

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _mlp(dims, seed):
    torch.manual_seed(seed)
    layers = []
    for i in range(len(dims) - 1):
        layers.append(torch.nn.Linear(dims[i], dims[i + 1]))
    return [l.weight.detach() for l in layers], [l.bias.detach() for l in layers]


def _graph(n, e, k0, seed, skew=False):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n, (e,), generator=g)
    if skew:        # a few destinations with thousands of in-edges, many with none, the rest ragged
        w = torch.rand(n, generator=g) ** 6
        w[: n // 5] = 0
        dst = torch.multinomial(w, e, replacement=True, generator=g)
    else:
        dst = torch.randint(0, n, (e,), generator=g)
    ea = torch.randn(e, k0, generator=g)
    x = torch.randn(n, 64, generator=g)
    return x, torch.stack([src, dst]), ea


CASES = [
    # name, dims, n, e, skew, precision
    ("k256_default", [6, 256, 256, 4096], 900, 40000, False, "f16split"),
    ("k256_skewed", [6, 256, 256, 4096], 1500, 50000, True, "f16split"),
    ("k256_k0_4", [4, 256, 256, 4096], 700, 36000, True, "f16split"),
    ("ragged_widths", [6, 200, 300, 4096], 600, 34000, True, "f16split"),       # K1P = 224 (7 chunks), K2P = 384
    ("five_chunks", [6, 160, 128, 4096], 500, 33000, False, "f16split"),         # smallest supported k1
    ("k1024_forced", [6, 1024, 1024, 4096], 300, 6000, True, "f16split_agg16"),  # headline MLP, small graph
    ("tiny_forced", [6, 256, 256, 4096], 50, 130, True, "f16split_agg16"),       # fewer edges than waves
]


@pytest.mark.parametrize("name,dims,n,e,skew,precision", CASES, ids=[c[0] for c in CASES])
def test_v6_matches_oracle_and_8wave(name, dims, n, e, skew, precision):
    ws_, bs_ = _mlp(dims, 1)
    x, ei, ea = _graph(n, e, dims[0], 2, skew)
    torch.manual_seed(3)
    root, bias = torch.randn(64, 64) / 8, torch.randn(64) / 8
    y64 = nnconv_forward(x, ei, ea, ws_, bs_, root, bias, aggr="mean", dtype=torch.float64)
    y6 = run_native(x, ei, ea, ws_, bs_, root, bias, "mean", precision=precision)
    y3 = run_native(x, ei, ea, ws_, bs_, root, bias, "mean", precision="f16split_8wave")
    if precision == "f16split":
        # the block work queue (default) against one static range per wave: same per-node summation order
        ys = run_native(x, ei, ea, ws_, bs_, root, bias, "mean", precision="f16split_static")
        assert rel_l2(y6, ys) <= 5e-7, (name, rel_l2(y6, ys))
    y32 = run_native(x, ei, ea, ws_, bs_, root, bias, "mean", precision="f32")
    assert torch.isfinite(y6).all()
    e6, e3, e32 = rel_l2(y6, y64), rel_l2(y3, y64), rel_l2(y32, y64)
    assert e6 <= TOL and e6 <= 4 * e32 + 2e-7, (name, e6, e3, e32)
    assert rel_l2(y6, y3) <= 5e-7, (name, rel_l2(y6, y3))        # same products, different tile alignment


def test_v6_is_the_kernel_that_ran():
    """The default precision picks the v6 kernel from 32,768 edges on (gpde_nnconv_fwd_plan reports it)."""
    ws_, bs_ = _mlp([6, 256, 256, 4096], 1)
    d = torch.device("cuda:0")
    pm = ops.pack_mlp([w.to(d) for w in ws_], [b.to(d) for b in bs_])
    assert ops.fused_kernel_name(1000, 40000, pm, "f16split") == "gpde_fused_f16v6_kernel"
    assert ops.fused_kernel_name(1000, 30000, pm, "f16split") == "gpde_fused_f16v3_kernel"
    assert ops.fused_kernel_name(1000, 40000, pm, "f16split_8wave") == "gpde_fused_f16v3_kernel"
    assert ops.fused_kernel_name(1000, 40000, pm, "f32") == "gpde_fused_kernel"


def test_v6_deterministic_linear_and_chunk_invariant():
    dims = [6, 256, 256, 4096]
    ws_, bs_ = _mlp(dims, 5)
    x, ei, ea = _graph(1200, 45000, 6, 6, True)
    y_a = run_native(x, ei, ea, ws_, bs_, None, None, "add")
    y_b = run_native(x, ei, ea, ws_, bs_, None, None, "add")
    assert torch.equal(y_a, y_b)                                   # plain stores, fixed summation order
    # linear in x (no root / bias): power-of-two scaling of x is exact through every split and scale
    y_2 = run_native(4.0 * x, ei, ea, ws_, bs_, None, None, "add")
    assert torch.equal(y_2, 4.0 * y_a)
    # small workspace -> several destination-node chunks; chunk boundaries move tile alignment only
    d = torch.device("cuda:0")
    pm = ops.pack_mlp([w.to(d) for w in ws_], [b.to(d) for b in bs_])
    full = ops.workspace_bytes(1200, 45000, pm)
    y_c = run_native(x, ei, ea, ws_, bs_, None, None, "add", ws_bytes=full // 3)
    assert rel_l2(y_c, y_a) <= 5e-7


def test_node_table_attributes_in_the_one_wave_per_simd_kernel():
    """SURVEY.md §8 row f3 on the headline kernel (round 3): gpde_fused_f16v6_kernel<false, NODEATTR> reads the attribute
    slots from the node table through the tile's source / destination ids (no `perm`, no [E, 6] tensor) - bitwise the
    result of the same kernel on the materialised tensor (the attribute values are the same floats), queue and static
    ranges, and within the bar of the float64 oracle."""
    from tests.test_host_logic import DenseNet
    from oracle.nnconv_oracle import nnconv_forward
    d = torch.device("cuda:0")
    torch.manual_seed(12)
    s, r = 61, 0.10
    ei = synth.lattice_radius_graph(s, r, d)
    pos = synth.lattice_positions(s, d)
    a = synth.darcy_coefficient(s, 5).to(d)
    ea = synth.darcy_edge_attr(ei, pos, a)
    n = s * s
    na = gp.NodeAttr.darcy(pos, a)
    x = torch.randn(n, 64, device=d)
    csr = ops.csr_for(ei, n)
    for dims in ([6, 256, 256, 4096], [6, 1024, 1024, 4096]):
        conv = gp.NNConv_old(64, 64, DenseNet(dims, torch.nn.ReLU), aggr="mean").to(d)
        lin = ops.mlp_linears(conv.nn)
        pm = ops.pack_mlp([l.weight for l in lin], [l.bias for l in lin])
        assert ops.fused_kernel_name(n, csr.n_edges, pm, "f16split") == "gpde_fused_f16v6_kernel"
        for prec in ("f16split", "f16split_static"):
            y_t = ops.nnconv_forward_raw(x, csr, ea, pm, conv.root, conv.bias, "mean", precision=prec)
            y_n = ops.nnconv_forward_nodeattr_raw(x, csr, na, pm, conv.root, conv.bias, "mean", precision=prec)
            torch.cuda.synchronize()
            assert torch.equal(y_t, y_n), (dims, prec, rel_l2(y_n.cpu(), y_t.cpu()))
        y8 = ops.nnconv_forward_nodeattr_raw(x, csr, na, pm, conv.root, conv.bias, "mean", precision="f16split_8wave")
        assert rel_l2(y8.cpu(), y_n.cpu()) <= 1e-6                     # the round-1/2 node-table kernel (8 waves)
    rows = torch.arange(0, n, 37)
    sel = torch.isin(ei[1].cpu(), rows)
    ei_s, ea_s = ei[:, sel.to(d)], ea[sel.to(d)]
    ref = nnconv_forward(x.cpu(), ei_s.cpu(), ea_s.cpu(), [l.weight.detach().cpu() for l in lin], [l.bias.detach().cpu() for l in lin],
                         conv.root.detach().cpu(), conv.bias.detach().cpu(), aggr="mean", dtype=torch.float64, chunk_edges=8192)
    assert rel_l2(y_n.cpu()[rows], ref[rows]) <= 1e-5


def test_attributes_in_slot_order_are_a_pure_layout_change(monkeypatch):
    """ops.attr_in_slot_order (round 3): edge_attr rows gathered into CSR slot order once per (graph, edge_attr), perm -> identity.
    Same values, same summation order: forward and every gradient are bit-identical to the indirect addressing; the copy is
    cached on the CSR and follows in-place changes of edge_attr (version counter)."""
    from tests.test_host_logic import DenseNet
    d = torch.device("cuda:0")
    torch.manual_seed(21)
    ei, ea, n = synth.darcy_graph(61, 0.10, device=d, seed=3)
    x = torch.randn(n, 64, device=d)
    conv = gp.NNConv_old(64, 64, DenseNet([6, 256, 256, 4096], torch.nn.ReLU), aggr="mean").to(d)
    lin = ops.mlp_linears(conv.nn)
    ws_, bs_ = [l.weight.detach() for l in lin], [l.bias.detach() for l in lin]
    pm = ops.pack_mlp(ws_, bs_)
    g = torch.randn(n, 64, device=d)

    def run():
        csr = ops.build_csr(ei, n)
        y = ops.nnconv_forward_raw(x, csr, ea, pm, conv.root, conv.bias, "mean")
        gr = ops.nnconv_backward_raw(x, csr, ea, ws_, bs_, conv.root.detach(), "mean", g)
        return csr, y, gr
    monkeypatch.setattr(ops, "ATTR_SLOT_ORDER", False)
    _, y0, g0 = run()
    monkeypatch.setattr(ops, "ATTR_SLOT_ORDER", True)
    csr, y1, g1 = run()
    assert csr._attr_sorted is not None and len(csr._attr_sorted) == 1
    assert torch.equal(y0, y1) and torch.equal(g0[0], g1[0])
    for l in range(3):
        assert torch.equal(g0[1][l], g1[1][l]) and torch.equal(g0[2][l], g1[2][l])
    srt = next(iter(csr._attr_sorted.values()))[1]
    assert torch.equal(srt, ea[csr.perm.long()])
    ea2 = ea.clone()
    y_a = ops.nnconv_forward_raw(x, csr, ea2, pm, conv.root, conv.bias, "mean")
    ea2.mul_(1.5)                                             # in place: new version -> new gathered copy
    y_b = ops.nnconv_forward_raw(x, csr, ea2, pm, conv.root, conv.bias, "mean")
    monkeypatch.setattr(ops, "ATTR_SLOT_ORDER", False)
    y_c = ops.nnconv_forward_raw(x, csr, ea2, pm, conv.root, conv.bias, "mean")
    assert torch.equal(y_a, y1) and torch.equal(y_b, y_c) and not torch.equal(y_a, y_b)
