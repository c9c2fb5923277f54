"""GPU tier: the reference's LITERAL kernel-MLP shapes, forward and every gradient against float64 (VERDICT r4 missing 4).

The scripts build `DenseNet([ker_in, ker_width // 2, ker_width, width ** 2])` and friends:
  [6,  500, 1000, 4096]              UAI3_resolution.py:20,58; UAI5_sample_generalize.py:20,62; UAI6_sample_radius.py:21,67
  [6,  512, 1024, 4096]              UAI7_evaluate.py:21,75; multipole-graph-neural-operator/neurips5_GKN.py:22,81
  [6,   32,   64, 4096]              UAI4_equation_sample.py:21,64
  [6, 256, 512, 1024, 1024, 4096]    UAI8_kernel.py:21 (five Linear layers)
  [5,   64,  128, 4096]              the second shipped checkpoint (model/grain_torus_r64_radius0.4testm100; its golden
                                     vector from the reference's own module is in tests/test_gpu_parity.py)
(all under /root/reference/graph-neural-operator/ unless named).  500 / 1000 are not tile multiples: they pad to 512 / 1024,
padded units give relu(0) = 0.  Graphs: a random graph of mean in-degree 66 with one long row (the staged per-edge kernels,
the split-f16 GEMMs' >= 8192-row switch) and a low in-degree one (the per-MFMA per-edge kernel); edges on the ReLU kink are
removed (tests/helpers/kinks.py) so that every gradient meets the plain tolerance."""
import pytest
import torch

from graph_pde_amd import ops
from oracle.nnconv_oracle import nnconv_forward, nnconv_grads, rel_l2
from tests.helpers.kinks import edges_off_the_kink

pytestmark = pytest.mark.gpu
TOL_FWD, TOL_BWD = 1e-6, 2e-5          # north_star bar for the forward: 1e-5

SHAPES = [[6, 500, 1000, 4096], [6, 512, 1024, 4096], [6, 32, 64, 4096], [6, 256, 512, 1024, 1024, 4096], [5, 64, 128, 4096]]


def _case(dims, n, e, seed, long_row=True):
    torch.manual_seed(seed)
    ei = torch.stack([torch.randint(0, n, (e,)), torch.randint(0, n - 5, (e,))])      # the last 5 nodes: no in-edges
    if long_row:
        ei[1, : e // 10] = 3
    ea, x = torch.randn(e, dims[0]), torch.randn(n, 64)
    nl = len(dims) - 1
    mlp = torch.nn.Sequential(*sum([[torch.nn.Linear(dims[i], dims[i + 1]), torch.nn.ReLU()] for i in range(nl)], [])[:-1])
    W = [l.weight.detach() for l in mlp if isinstance(l, torch.nn.Linear)]
    B = [l.bias.detach() for l in mlp if isinstance(l, torch.nn.Linear)]
    keep = edges_off_the_kink(ea, W, B)
    ei, ea = ei[:, keep].contiguous(), ea[keep].contiguous()
    root = torch.empty(64, 64).uniform_(-0.125, 0.125)
    bias = torch.empty(64).uniform_(-0.125, 0.125)
    return x, ei, ea, W, B, root, bias, torch.randn(n, 64)


@pytest.mark.parametrize("n,e,long_row", [(300, 20000, True), (2000, 6000, False)], ids=["deg66", "deg3"])
@pytest.mark.parametrize("dims", SHAPES, ids=["-".join(map(str, s)) for s in SHAPES])
def test_literal_reference_shapes_forward_and_backward_vs_float64(dims, n, e, long_row):
    d = torch.device("cuda:0")
    x, ei, ea, W, B, root, bias, gout = _case(dims, n, e, sum(dims) + e, long_row)
    ref = nnconv_forward(x, ei, ea, W, B, root, bias, aggr="mean", dtype=torch.float64)
    csr = ops.build_csr(ei.to(d), n)
    Wd, Bd = [w.to(d) for w in W], [b.to(d) for b in B]
    pm = ops.pack_mlp(Wd, Bd)
    out = ops.nnconv_forward_raw(x.to(d), csr, ea.to(d), pm, root.to(d), bias.to(d), "mean")
    torch.cuda.synchronize()
    err = rel_l2(out.cpu(), ref)
    assert err <= TOL_FWD, ("forward", dims, err)
    rx, rW, rb, rroot, rbias = nnconv_grads(x, ei, ea, W, B, root, bias, "mean", gout, chunk_edges=4096)
    gx, gW, gb, groot, gbias = ops.nnconv_backward_raw(x.to(d), csr, ea.to(d), Wd, Bd, root.to(d), "mean", gout.to(d))
    torch.cuda.synchronize()
    errs = {"dx": rel_l2(gx.cpu(), rx), "droot": rel_l2(groot.cpu(), rroot), "dbias": rel_l2(gbias.cpu(), rbias)}
    for l in range(len(W)):
        errs[f"dW{l + 1}"] = rel_l2(gW[l].cpu(), rW[l])
        errs[f"db{l + 1}"] = rel_l2(gb[l].cpu(), rb[l])
    bad = {k: v for k, v in errs.items() if not v <= TOL_BWD}
    assert not bad, (dims, (n, e), bad)
