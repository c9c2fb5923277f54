"""GPU tier: the backward's dU_1 = (dU_2 . W_2) (.) [H_1 > 0] on split-f16 MFMA (gpde_gemm_f16s_nt_kernel,
csrc/gpde_gemm_f16s.hip; 3-Linear kernels with k2 >= 256) - gradients against float64 autograd through the oracle
(the reference's loss.backward()), against the exact-fp32 GEMM path, and bit-reproducibility of the weight
gradients."""
import pytest
import torch

from graph_pde_amd import ops
from oracle.nnconv_oracle import rel_l2
from tests.test_gpu_parity import _oracle_grads, dev

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _case(dims, n, e, seed):
    torch.manual_seed(seed)
    ei = torch.stack([torch.randint(0, n, (e,)), torch.randint(0, n - 5, (e,))])
    ei[1, : e // 10] = 3                                   # one destination spanning many tiles
    ea, x = torch.randn(e, dims[0]), torch.randn(n, 64)
    mlp = torch.nn.Sequential(*sum([[torch.nn.Linear(dims[i], dims[i + 1]), torch.nn.ReLU()]
                                    for i in range(len(dims) - 1)], [])[:-1])
    ws_ = [l.weight.detach() for l in mlp if isinstance(l, torch.nn.Linear)]
    bs_ = [l.bias.detach() for l in mlp if isinstance(l, torch.nn.Linear)]
    root = torch.empty(64, 64).uniform_(-0.125, 0.125)
    bias = torch.empty(64).uniform_(-0.125, 0.125)
    return x, ei, ea, ws_, bs_, root, bias, torch.randn(n, 64)


def _native(x, ei, ea, ws_, bs_, root, gout, aggr="mean"):
    d = dev()
    csr = ops.build_csr(ei.to(d), x.shape[0])
    out = ops.nnconv_backward_raw(x.to(d), csr, ea.to(d), [w.to(d) for w in ws_], [b.to(d) for b in bs_], root.to(d), aggr,
                                  gout.to(d))
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("dims,n,e", [([6, 256, 256, 4096], 200, 3000), ([6, 200, 300, 4096], 150, 2500),
                                      ([6, 1024, 1024, 4096], 60, 900), ([4, 512, 256, 4096], 120, 1100)])
def test_backward_with_split_f16_du1_matches_reference_autograd(dims, n, e, monkeypatch):
    x, ei, ea, ws_, bs_, root, bias, gout = _case(dims, n, e, sum(dims))
    rx, rW, rb, rroot, rbias = _oracle_grads(x, ei, ea, ws_, bs_, root, bias, "mean", gout)
    monkeypatch.delenv("GPDE_BWD_GEMM_F32", raising=False)
    gx, gW, gb, groot, gbias = _native(x, ei, ea, ws_, bs_, root, gout)
    monkeypatch.setenv("GPDE_BWD_GEMM_F32", "1")
    fx, fW, fb, froot, fbias = _native(x, ei, ea, ws_, bs_, root, gout)
    assert rel_l2(gx.cpu(), rx) <= TOL
    for l in range(3):
        e16, e32 = rel_l2(gW[l].cpu(), rW[l]), rel_l2(fW[l].cpu(), rW[l])
        assert e16 <= TOL and e16 <= 4 * e32 + 2e-7, (f"dW{l}", e16, e32)
        assert rel_l2(gb[l].cpu(), rb[l]) <= TOL, (f"db{l}", rel_l2(gb[l].cpu(), rb[l]))
    # layer 1 is where the two arithmetics differ (dW_1, db_1 come from dU_1); they must agree closely
    assert rel_l2(gW[0].cpu(), fW[0].cpu()) <= 2e-6, rel_l2(gW[0].cpu(), fW[0].cpu())
    assert torch.equal(gW[2], fW[2]) and torch.equal(gW[1], fW[1])          # untouched by the change


@pytest.mark.parametrize("variant", ["1", "2", "3"])      # per-MFMA operands / staged fp32 MFMA / staged split-f16 MFMA
def test_all_gradients_are_bit_reproducible(variant, monkeypatch):
    """Weight gradients: ordered split partials.  grad_x: per-edge contributions summed per source node in slot order
    (src_rowptr / src_slots of gpde_nnconv_bwd, from gpde_csr_source_order) instead of atomics - for both per-edge kernels, and identical
    whether the edges are processed in one chunk or in several (a smaller workspace)."""
    monkeypatch.setenv("GPDE_EDGE_BWD", variant)
    x, ei, ea, ws_, bs_, root, bias, gout = _case([6, 256, 256, 4096], 200, 9000, 5)
    a = _native(x, ei, ea, ws_, bs_, root, gout)
    b = _native(x, ei, ea, ws_, bs_, root, gout)

    def same(u, v, name):       # on failure: which gradient, how many entries, where, how far
        if not torch.equal(u, v):
            dif = (u != v).nonzero()
            raise AssertionError(f"{name}: {dif.shape[0]} of {u.numel()} entries differ between two identical calls, first "
                                 f"{dif[:6].tolist()}, rel-L2 {float((u - v).norm() / u.norm()):.2e}")
    same(a[0], b[0], "grad_x")
    for l in range(3):
        same(a[1][l], b[1][l], f"grad_W{l + 1}")
        same(a[2][l], b[2][l], f"grad_b{l + 1}")
    same(a[3], b[3], "grad_root")
    same(a[4], b[4], "grad_bias")
    # several node / edge chunks: same grad_x bits (chunks are applied in order, one owner per element)
    d = dev()
    csr = ops.build_csr(ei.to(d), x.shape[0])
    from graph_pde_amd import _lib
    dims_c = _lib.dims_array([6, 256, 256, 4096])
    full = int(_lib.lib().gpde_nnconv_bwd_workspace_bytes(x.shape[0], ei.shape[1], 3, dims_c))
    small = torch.empty(full // 2, dtype=torch.uint8, device=d)       # (below the one-chunk size: 3 - 4 edge chunks; the call-wide buffers take a third of `full` at this size)
    c = ops.nnconv_backward_raw(x.to(d), csr, ea.to(d), [w.to(d) for w in ws_], [b_.to(d) for b_ in bs_], root.to(d), "mean",
                                gout.to(d), ws=small)
    assert torch.equal(c[0], a[0])


def test_ordered_grad_x_matches_the_atomic_path(monkeypatch):
    x, ei, ea, ws_, bs_, root, bias, gout = _case([6, 256, 256, 4096], 200, 9000, 6)
    a = _native(x, ei, ea, ws_, bs_, root, gout)
    monkeypatch.setattr(ops, "DX_MODE", "atomic")
    b = _native(x, ei, ea, ws_, bs_, root, gout)
    assert rel_l2(a[0].cpu(), b[0].cpu()) <= 1e-6
    rx = _oracle_grads(x, ei, ea, ws_, bs_, root, bias, "mean", gout)[0]
    assert rel_l2(a[0].cpu(), rx) <= TOL


def test_dw2_on_the_split_f16_gemm_matches_reference_autograd(monkeypatch):
    """From 8192 edges per chunk on, dW_2 = dU_2^T . H_1 runs on the same split-f16 GEMM (operands transposed into
    K-contiguous form, 8 K splits): against float64 autograd and against the fp32 GEMM it replaces."""
    dims, n, e = [6, 256, 256, 4096], 300, 20000
    x, ei, ea, ws_, bs_, root, bias, gout = _case(dims, n, e, 77)
    rx, rW, rb, rroot, rbias = _oracle_grads(x, ei, ea, ws_, bs_, root, bias, "mean", gout)
    monkeypatch.delenv("GPDE_BWD_GEMM_F32", raising=False)
    monkeypatch.delenv("GPDE_BWD_DW2_F32", raising=False)
    gx, gW, gb, groot, gbias = _native(x, ei, ea, ws_, bs_, root, gout)
    g2 = _native(x, ei, ea, ws_, bs_, root, gout)
    monkeypatch.setenv("GPDE_BWD_DW2_F32", "1")
    fx, fW, fb, froot, fbias = _native(x, ei, ea, ws_, bs_, root, gout)
    # Against float64 row by row: with 5 M hidden activations one or two lie within rounding of the ReLU kink, where the
    # fp32-level forward and the float64 oracle pick different masks (a single flipped entry moves ITS row of dW_2 by
    # ~1/sqrt(E) - 7e-4 of the whole matrix at this size, in the fp32 path exactly as in the split path).  All other
    # rows must match to the gradient tolerance.
    rows16 = ((gW[1].cpu().double() - rW[1].double()).norm(dim=1) / rW[1].double().norm(dim=1))
    rows32 = ((fW[1].cpu().double() - rW[1].double()).norm(dim=1) / rW[1].double().norm(dim=1))
    assert int((rows16 > TOL).sum()) <= 4 and int((rows16 > TOL).sum()) <= int((rows32 > TOL).sum()) + 1, \
        (rows16.max(), rows32.max())
    assert float(rows16.median()) <= 2e-6
    assert not torch.equal(gW[1], fW[1])                    # the split path did run
    assert rel_l2(gW[1].cpu(), fW[1].cpu()) <= 2e-6
    assert torch.equal(gW[1], g2[1][1])                     # and is bit-reproducible
    assert torch.equal(gW[2], fW[2])
    # (round 6: with the split dW_2 GEMM, dW_1 is summed per 64-row tile in the dU_1 GEMM's epilogue - the fp32 form sums the same
    #  products in k_dw_first's order)
    assert torch.equal(gW[0], g2[1][0]) and rel_l2(gW[0].cpu(), fW[0].cpu()) <= 2e-6


def test_dw2_split_gemm_at_headline_widths_agrees_with_fp32(monkeypatch):
    dims, n, e = [6, 1024, 1024, 4096], 500, 30011           # a ragged K: 30011 rows pad to 8 splits of 3776
    x, ei, ea, ws_, bs_, root, bias, gout = _case(dims, n, e, 78)
    monkeypatch.delenv("GPDE_BWD_GEMM_F32", raising=False)
    monkeypatch.delenv("GPDE_BWD_DW2_F32", raising=False)
    gx, gW, gb, groot, gbias = _native(x, ei, ea, ws_, bs_, root, gout)
    monkeypatch.setenv("GPDE_BWD_DW2_F32", "1")
    fx, fW, fb, froot, fbias = _native(x, ei, ea, ws_, bs_, root, gout)
    assert not torch.equal(gW[1], fW[1])
    assert rel_l2(gW[1].cpu(), fW[1].cpu()) <= 2e-6, rel_l2(gW[1].cpu(), fW[1].cpu())
    assert torch.isfinite(gW[1]).all()


def test_first_hidden_layer_is_not_materialised_and_changes_only_the_dw2_rounding(monkeypatch):
    """Round 3: from 8192 edges per chunk on, H_1 is never written - dW_2's operand image is generated from the attribute
    slots (k_first_layer_pack, scales from an a-priori bound instead of measured column maxima) and the ReLU mask of dU_1
    travels as bits.  Same fmaf chain as the tensor it replaces: the mask, hence dU_1, dW_1, db_1, are bit-identical;
    dW_2 differs only by the rounding of differently scaled operands."""
    dims, n, e = [6, 256, 384, 4096], 300, 21000
    x, ei, ea, ws_, bs_, root, bias, gout = _case(dims, n, e, 81)
    rx, rW, rb, rroot, rbias = _oracle_grads(x, ei, ea, ws_, bs_, root, bias, "mean", gout)
    monkeypatch.delenv("GPDE_BWD_H1_MATERIALIZE", raising=False)
    gx, gW, gb, groot, gbias = _native(x, ei, ea, ws_, bs_, root, gout)
    monkeypatch.setenv("GPDE_BWD_H1_MATERIALIZE", "1")
    fx, fW, fb, froot, fbias = _native(x, ei, ea, ws_, bs_, root, gout)
    assert torch.equal(gx, fx)
    assert torch.equal(gW[2], fW[2]) and torch.equal(gb[2], fb[2])
    # round 6: H_1 comes from the split-f16 MFMA pair inside the GEMMs; a product within its error bound of zero takes its sign from
    # this very fmaf chain, so the MASK is still the materialised tensor's bit for bit - dU_1 too - while dW_1 / db_1 are now summed
    # per 64-row tile in the dU_1 GEMM's epilogue (another order of the same fp32 products)
    assert rel_l2(gW[0].cpu(), fW[0].cpu()) <= 2e-6 and rel_l2(gb[0].cpu(), fb[0].cpu()) <= 2e-6, (rel_l2(gW[0].cpu(), fW[0].cpu()), rel_l2(gb[0].cpu(), fb[0].cpu()))
    assert torch.equal(gb[1], fb[1])
    assert rel_l2(gW[1].cpu(), fW[1].cpu()) <= 2e-6
    def row_err(g, r):          # per row of dW_2; hidden units that never fire have an all-zero row in both (0 / 0)
        assert torch.equal(g.cpu()[r.norm(dim=1) == 0], r[r.norm(dim=1) == 0].float())
        live = r.norm(dim=1) > 0
        return (g.cpu().double()[live] - r.double()[live]).norm(dim=1) / r.double()[live].norm(dim=1)
    rows = row_err(gW[1], rW[1])
    assert int((rows > TOL).sum()) <= 4 and float(rows.median()) <= 2e-6, (rows.max(), rows.median())
    # wide dynamic range of the attributes: the bound-based column scales must still leave every H_1 column its accuracy
    ea2 = ea * torch.logspace(-2, 2, dims[0]).view(1, -1)
    r2 = _oracle_grads(x, ei, ea2, ws_, bs_, root, bias, "mean", gout)[1]
    monkeypatch.delenv("GPDE_BWD_H1_MATERIALIZE", raising=False)
    g2 = _native(x, ei, ea2, ws_, bs_, root, gout)[1]
    rows2 = row_err(g2[1], r2[1])
    assert int((rows2 > TOL).sum()) <= 4 and float(rows2.median()) <= 2e-6, (rows2.max(), rows2.median())


@pytest.mark.parametrize("cached", [False, True])
def test_keep_z_forward_feeds_the_backward(cached, monkeypatch):
    """gpde_nnconv_fwd_keepz / `z_saved` of gpde_nnconv_bwd (round 3): the forward leaves Z_i = sum_e x_j (x) h_e of every node for the
    backward's dW_3 instead of the backward re-aggregating it.  Through the module (direct operator and the hidden-activation
    split): output bit-identical, every gradient within rounding of the path without it and within the tolerance of float64
    autograd; nodes without in-edges and a 2-chunk-sized graph included."""
    import graph_pde_amd as gp
    from graph_pde_amd import hidden_cache
    from tests.test_host_logic import DenseNet
    d = dev()
    torch.manual_seed(31)
    n, e = 220, 9000                                            # >= 32 edges per node: the buffer is used
    ei = torch.stack([torch.randint(0, n, (e,)), torch.randint(0, n - 7, (e,))]).to(d)       # the last 7 nodes have no in-edge
    ea, x0 = torch.randn(e, 6, device=d), torch.randn(n, 64, device=d)
    conv = gp.NNConv_old(64, 64, DenseNet([6, 256, 256, 4096], torch.nn.ReLU), aggr="mean").to(d)
    gout = torch.randn(n, 64, device=d)
    monkeypatch.setattr(hidden_cache, "MODE", "on" if cached else "off")

    def run(save_bytes):
        monkeypatch.setattr(ops, "SAVE_Z_BYTES", save_bytes)
        hidden_cache.clear()
        conv.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        y = conv(x, ei, ea)
        (y * gout).sum().backward()
        return y.detach(), [x.grad.clone()] + [p.grad.clone() for p in conv.parameters()]
    seen = []
    for name in ("nnconv_backward_raw", "nnconv_backward_hidden_raw"):
        orig = getattr(ops, name)
        monkeypatch.setattr(ops, name, (lambda f: lambda *a, **k: (seen.append(k.get("z_saved") is not None), f(*a, **k))[1])(orig))
    y0, g0 = run(0)
    assert seen and not any(seen)
    del seen[:]
    y1, g1 = run(16 << 30)
    assert seen and all(seen)                                     # the backward did read the forward's Z
    assert torch.equal(y0, y1)
    for a, b in zip(g0, g1):       # (at this size both Z come from the same fp32-MFMA aggregation order: often the same bits)
        assert rel_l2(a.cpu(), b.cpu()) <= 2e-6, rel_l2(a.cpu(), b.cpu())
    lin = ops.mlp_linears(conv.nn)
    rx, rW, rb, rroot, rbias = _oracle_grads(x0.cpu(), ei.cpu(), ea.cpu(), [l.weight.detach().cpu() for l in lin],
                                             [l.bias.detach().cpu() for l in lin], conv.root.detach().cpu(), conv.bias.detach().cpu(),
                                             "mean", gout.cpu())
    assert rel_l2(g1[0].cpu(), rx) <= TOL
    names = dict(conv.named_parameters())
    assert rel_l2(names["nn.layers.4.weight"].grad.cpu(), rW[2]) <= TOL and rel_l2(names["root"].grad.cpu(), rroot) <= TOL
    assert ops.z_buffer(ops.csr_for(ei, n), [6, 256, 256, 4096], d) is not None
    monkeypatch.setattr(ops, "SAVE_Z_BYTES", 1 << 20)
    assert ops.z_buffer(ops.csr_for(ei, n), [6, 256, 256, 4096], d) is None     # over budget -> the backward aggregates itself
    hidden_cache.clear()


def test_gradient_of_the_edge_attributes_matches_float64_autograd():
    """dL/d edge_attr (`grad_edge_attr` of gpde_nnconv_bwd): the reference's `pseudo` is an ordinary autograd input (nn_conv.py:273-275);
    no script differentiates it, the module does when asked.  Rows go back in the CALLER's edge order; the other gradients
    are the bits of the call without it."""
    import graph_pde_amd as gp
    from tests.test_host_logic import DenseNet
    dims, n, e = [6, 256, 256, 4096], 200, 9000
    x, ei, ea, ws_, bs_, root, bias, gout = _case(dims, n, e, 17)
    d = dev()
    xs = x.double().requires_grad_(True)
    at = ea.double().requires_grad_(True)
    Ws = [w.double() for w in ws_]
    Bs = [b.double() for b in bs_]
    h = at
    for l in range(3):
        h = torch.nn.functional.linear(h, Ws[l], Bs[l])
        if l < 2:
            h = torch.relu(h)
    m = torch.matmul(xs[ei[0]].unsqueeze(1), h.view(-1, 64, 64)).squeeze(1)
    out = torch.zeros(n, 64, dtype=torch.float64).index_add(0, ei[1], m)
    out = out / torch.bincount(ei[1], minlength=n).clamp(min=1).double().unsqueeze(1) + xs @ root.double() + bias.double()
    (out * gout.double()).sum().backward()
    csr = ops.build_csr(ei.to(d), n)
    args = (x.to(d), csr, ea.to(d), [w.to(d) for w in ws_], [b.to(d) for b in bs_], root.to(d), "mean", gout.to(d))
    plain = ops.nnconv_backward_raw(*args)
    got = ops.nnconv_backward_raw(*args, need_attr=True)
    got2 = ops.nnconv_backward_raw(*args, need_attr=True)
    torch.cuda.synchronize()
    assert rel_l2(got[5].cpu(), at.grad) <= TOL, rel_l2(got[5].cpu(), at.grad)
    assert torch.equal(got[5], got2[5]) and torch.equal(got[0], plain[0])
    for l in (1, 2):
        assert torch.equal(got[1][l], plain[1][l]) and torch.equal(got[2][l], plain[2][l])
    # (with the attribute gradient dU_1 is materialised and dW_1 / db_1 come from k_dw_first's pass; without it they are summed per
    #  tile in the dU_1 GEMM's epilogue when the chunk has >= 8192 rows - here both take the pass: bits)
    assert rel_l2(got[1][0].cpu(), plain[1][0].cpu()) <= 2e-6 and rel_l2(got[2][0].cpu(), plain[2][0].cpu()) <= 2e-6
    # through the module: an edge_attr that requires a gradient takes the direct operator and receives it
    conv = gp.NNConv_old(64, 64, DenseNet(dims, torch.nn.ReLU), aggr="mean").to(d)
    lin = ops.mlp_linears(conv.nn)
    with torch.no_grad():
        for l, layer in enumerate(lin):
            layer.weight.copy_(ws_[l]); layer.bias.copy_(bs_[l])
        conv.root.copy_(root); conv.bias.copy_(bias)
    ea_d = ea.to(d).requires_grad_(True)
    for _ in range(3):                                      # repeated calls: the caches step aside for a differentiated edge_attr
        y = conv(x.to(d), ei.to(d), ea_d)
    (y * gout.to(d)).sum().backward()
    assert rel_l2(ea_d.grad.cpu(), at.grad) <= TOL


def test_one_chunk_backward_agrees_with_the_default_chunking_at_s121():
    """The module's backward takes a workspace of up to one chunk when the device has the room (ops.bwd_workspace_bytes):
    on the s=121 graph (5.9 M edges) that is ONE 5.9 M-row chunk instead of ten of ~640 k.  The dW_2 GEMM then contracts over
    all of them: its K split count follows the chunk (tn_ksplits, gpde_bwd.hip) so that no fp32 accumulator runs over more than
    ~100 k edges - with 8 splits the two chunkings sat 3e-5 apart in dW_2.  grad_x is the same bits in both."""
    from graph_pde_amd import _lib, synth
    d = dev()
    torch.manual_seed(3)
    ei, ea, n = synth.darcy_graph(121, 0.1, device=d, seed=0)
    e = int(ei.shape[1])
    dims = [6, 1024, 1024, 4096]
    mlp = torch.nn.Sequential(torch.nn.Linear(6, 1024), torch.nn.ReLU(), torch.nn.Linear(1024, 1024), torch.nn.ReLU(),
                              torch.nn.Linear(1024, 4096)).to(d)
    lin = ops.mlp_linears(mlp)
    ws_, bs_ = [l.weight.detach() for l in lin], [l.bias.detach() for l in lin]
    root = torch.randn(64, 64, device=d) / 8
    x, g = torch.randn(n, 64, device=d), torch.randn(n, 64, device=d)
    csr = ops.csr_for(ei, n)
    dims_c = _lib.dims_array(dims)
    lib = _lib.lib()
    small = int(lib.gpde_nnconv_bwd_workspace_bytes(n, e, 3, dims_c))
    one = int(lib.gpde_nnconv_bwd_workspace_bytes_one_chunk(n, e, 3, dims_c))
    assert one > 4 * small
    free, _ = ops.device_free_bytes(d)
    if one + (8 << 30) > free:
        pytest.skip("device has no room for the one-chunk workspace")
    outs = []
    for nbytes in (small, one):
        ws = torch.empty(nbytes, dtype=torch.uint8, device=d)
        outs.append(ops.nnconv_backward_raw(x, csr, ea, ws_, bs_, root, "mean", g, ws=ws))
        del ws
    a, b = outs
    assert torch.equal(a[0], b[0])                                   # grad_x: per-edge rows, one owner per element
    for l in range(3):
        assert rel_l2(a[1][l].cpu(), b[1][l].cpu()) <= 1.5e-5, (f"dW{l + 1}", rel_l2(a[1][l].cpu(), b[1][l].cpu()))
        assert rel_l2(a[2][l].cpu(), b[2][l].cpu()) <= 1.5e-5, (f"db{l + 1}", rel_l2(a[2][l].cpu(), b[2][l].cpu()))
