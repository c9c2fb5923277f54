"""GPU tier: the per-edge last layer for low in-degree graphs (mean in-degree <= 4, >= 4096 edges, k2 >= 256:
gpde_launch_edge_messages, csrc/gpde_gemm_f16s.hip + the branch in csrc/gpde_api.hip).  There the operator keeps the
reference's own association - W_e = nn(pseudo).view(-1, 64, 64), m_e = x_j . W_e, scatter-mean
(/root/reference/graph-neural-operator/nn_conv.py:273-275) - with W_e formed tile by tile on split-f16 MFMA and
contracted with x_j in the GEMM's epilogue.  Checked against the float64 oracle, against the re-associated path it
replaces on these graphs (flag GPDE_FWD_NO_EDGE_PATH), from given hidden activations, and for reproducibility."""
import pytest
import torch

from graph_pde_amd import ops
from oracle.nnconv_oracle import nnconv_forward, rel_l2
from tests.test_gpu_parity import dev, run_native
from tests.test_gpu_v6 import _mlp

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _low_degree_graph(n, e, k0, seed, holes=True):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n, (e,), generator=g)
    dst = torch.randint(n // 10 if holes else 0, n, (e,), generator=g)      # the first tenth of the nodes has no in-edge
    dst[: e // 50] = n - 1                                                   # and one node has many
    return torch.randn(n, 64, generator=g), torch.stack([src, dst]), torch.randn(e, k0, generator=g)


CASES = [("burgers_like", [4, 1024, 1024, 4096], 4096, 8192, "mean"),
         ("k512", [4, 512, 512, 4096], 3000, 9001, "mean"),            # ragged: 9001 rows, last tile partial
         ("k256_add", [6, 256, 256, 4096], 2000, 7777, "add"),
         ("k300_padded", [6, 192, 300, 4096], 1500, 5000, "mean")]     # K2P = 384 with zero-padded hidden units


@pytest.mark.parametrize("name,dims,n,e,aggr", CASES, ids=[c[0] for c in CASES])
def test_per_edge_path_matches_oracle_and_the_reassociated_path(name, dims, n, e, aggr):
    ws_, bs_ = _mlp(dims, 11)
    x, ei, ea = _low_degree_graph(n, e, dims[0], 12)
    torch.manual_seed(13)
    root, bias = torch.randn(64, 64) / 8, torch.randn(64) / 8
    y64 = nnconv_forward(x, ei, ea, ws_, bs_, root, bias, aggr=aggr, dtype=torch.float64)
    ye = run_native(x, ei, ea, ws_, bs_, root, bias, aggr, precision="f16split")
    yz = run_native(x, ei, ea, ws_, bs_, root, bias, aggr, precision="f16split_noedge")
    y32 = run_native(x, ei, ea, ws_, bs_, root, bias, aggr, precision="f32")
    assert not torch.equal(ye, yz)                                # a different association: the per-edge path did run
    ee, ez, e32 = rel_l2(ye, y64), rel_l2(yz, y64), rel_l2(y32, y64)
    assert ee <= TOL and ee <= 4 * e32 + 2e-7, (name, ee, ez, e32)
    assert rel_l2(ye, yz) <= 1e-6, (name, rel_l2(ye, yz))
    assert torch.equal(ye, run_native(x, ei, ea, ws_, bs_, root, bias, aggr, precision="f16split"))   # reproducible


def test_per_edge_path_without_root_and_bias_and_from_given_hidden_activations():
    dims, n, e = [6, 256, 256, 4096], 2500, 6000
    ws_, bs_ = _mlp(dims, 21)
    x, ei, ea = _low_degree_graph(n, e, 6, 22, holes=False)
    y64 = nnconv_forward(x, ei, ea, ws_, bs_, None, None, aggr="mean", dtype=torch.float64)
    assert rel_l2(run_native(x, ei, ea, ws_, bs_, None, None, "mean", precision="f16split"), y64) <= TOL
    d = dev()
    csr = ops.build_csr(ei.to(d), n)
    wd, bd = [w.to(d) for w in ws_], [b.to(d) for b in bs_]
    pm = ops.pack_mlp(wd, bd)
    h, hm = ops.hidden_forward_raw(csr, ea.to(d), pm, wd[:-1] + [None], bd[:-1] + [None], "f16split")
    yh = ops.nnconv_forward_hidden_raw(x.to(d), csr, h, pm, None, None, "mean", hmax=hm)
    yd = ops.nnconv_forward_raw(x.to(d), csr, ea.to(d), pm, None, None, "mean", precision="f16split")
    assert rel_l2(yh.cpu(), y64) <= TOL
    assert torch.equal(yh, yd)            # same H rows, same GEMM, same summation order


def test_graphs_outside_the_rule_keep_the_reassociated_path():
    # mean in-degree 15; fewer than 4096 edges; a first hidden width the fused store kernel does not cover (7 chunks)
    for dims, n, e in (([6, 256, 256, 4096], 400, 6000), ([6, 256, 256, 4096], 3000, 3000), ([6, 200, 300, 4096], 1500, 5000)):
        ws_, bs_ = _mlp(dims, 31)
        x, ei, ea = _low_degree_graph(n, e, 6, 32, holes=False)
        a = run_native(x, ei, ea, ws_, bs_, None, None, "mean", precision="f16split")
        b = run_native(x, ei, ea, ws_, bs_, None, None, "mean", precision="f16split_noedge")
        assert torch.equal(a, b), (dims, n, e)
