"""GPU tier: native radius-graph construction against the reference's own generators
(tests/golden/mesh_ties.npz, mgkn_graphs_s20.npz) and the C oracle (oracle/radius_oracle.c): integer work,
bit-exact - edge ORDER included."""
import hashlib
import os

import numpy as np
import pytest
import torch

from graph_pde_amd import ops, synth
from oracle import radius_oracle
from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _lattice(s):
    g = np.linspace(0.0, 1.0, s)
    return np.vstack([xx.ravel() for xx in np.meshgrid(g, g)]).T


def test_reference_ties_mode_reproduces_the_reference_graph():
    """The reference's default training graph (s = 61, r = 0.10: 376,471 edges, not the exact test's 383,293)."""
    d = torch.device("cuda:0")
    g = np.load(os.path.join(GOLDEN, "mesh_ties.npz"))
    r = float(g["r"])
    ei31 = ops.radius_graph(torch.from_numpy(_lattice(31)).to(d), r, reference_ties=True).cpu().numpy()
    assert np.array_equal(ei31, g["edge_index_s31"].astype(np.int64))
    ei61 = ops.radius_graph(torch.from_numpy(_lattice(61)).to(d), r, reference_ties=True).cpu().numpy()
    assert ei61.shape[1] == 376471
    assert hashlib.sha256(np.ascontiguousarray(ei61).tobytes()).hexdigest() == str(g["sha256_s61"])


@pytest.mark.parametrize("ties", [False, True])
def test_native_radius_graph_equals_the_oracle(ties):
    d = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    for dim, n, r in ((1, 300, 0.05), (2, 900, 0.11), (3, 500, 0.3)):
        p = rng.random((n, dim))
        p[::7] = np.round(p[::7] * 8) / 8                      # lattice-like points: exact-distance pairs exist
        want = radius_oracle.radius_edges(p, r, reference_ties=ties)
        got = ops.radius_graph(torch.from_numpy(p).to(d), r, reference_ties=ties).cpu().numpy()
        assert np.array_equal(got, want), (dim, ties)
        q = rng.random((n // 3, dim))
        want = radius_oracle.radius_edges(p, r, y=q, reference_ties=ties)
        got = ops.radius_graph(torch.from_numpy(p).to(d), r, reference_ties=ties, pos_dst=torch.from_numpy(q).to(d)).cpu().numpy()
        assert np.array_equal(got, want), (dim, ties, "two sets")


def test_multilevel_graphs_on_the_gpu_match_the_reference_generator():
    """RandomMultiMeshGenerator.ball_connectivity (inner / down / up, utilities.py:602-640) from the same sampled
    lattice indices: the fixture is the reference generator's own output."""
    d = torch.device("cuda:0")
    g = np.load(os.path.join(GOLDEN, "mgkn_graphs_s20.npz"))
    m = [int(v) for v in g["m"]]
    pos = synth.lattice_positions(int(g["s"])).double()
    pts = [pos[torch.from_numpy(g[f"idx{l}"]).long()].to(d) for l in range(len(m))]
    out = ops.multilevel_radius_graphs(pts, list(g["radii_inner"]), list(g["radii_inter"]), reference_ties=True)
    offs = np.concatenate([[0], np.cumsum(m)])
    for l in range(len(m)):
        lo, hi = g["range"][l]
        assert np.array_equal(out["inner"][l].cpu().numpy() + offs[l], g["edge_index"][:, lo:hi]), l
    for l in range(len(m) - 1):
        lo, hi = g["range_down"][l]
        assert np.array_equal(out["down"][l].cpu().numpy() + np.array([[offs[l]], [offs[l + 1]]]), g["edge_index_down"][:, lo:hi]), l
        assert np.array_equal(out["up"][l].cpu().numpy() + np.array([[offs[l + 1]], [offs[l]]]), g["edge_index_up"][:, lo:hi]), l


# ---- cell-list radius graph emitted as the destination CSR (gpde_radius_csr_*, round 3) ----------------------------------
@pytest.mark.parametrize("ties", [False, True])
@pytest.mark.parametrize("case", ["lattice31", "lattice61", "random2d", "random3d", "line1d", "clustered"])
def test_cell_list_csr_is_the_csr_of_the_brute_force_graph(case, ties):
    """rowptr / src / dst of ops.radius_csr == ops.csr_for(ops.radius_graph(...)) bit for bit (rows in ascending source
    order = a stable sort by destination of the reference's source-major list), for both distance arithmetics - the
    reference-ties one decides the lattice pairs at exactly distance r (tests/golden/mesh_ties.npz pins radius_graph)."""
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    if case.startswith("lattice"):
        s = int(case[7:])
        pos, r = synth.lattice_positions(s, d), 0.10
    elif case == "random2d":
        pos, r = torch.rand(3000, 2, generator=g, dtype=torch.float64).to(d), 0.07
    elif case == "random3d":
        pos, r = torch.rand(2500, 3, generator=g, dtype=torch.float64).to(d), 0.16
    elif case == "line1d":
        pos, r = torch.rand(4000, 1, generator=g, dtype=torch.float64).to(d) * 3.0 - 1.0, 0.011
    else:                                                    # most points in one cell, a few far away; a radius above the extent
        pos = torch.cat([torch.rand(900, 2, generator=g, dtype=torch.float64) * 0.01, torch.rand(40, 2, generator=g, dtype=torch.float64) * 50.0]).to(d)
        r = 0.004
    n = pos.shape[0]
    ei = ops.radius_graph(pos, r, reference_ties=ties)
    ref = ops.build_csr(ei, n)
    got = ops.radius_csr(pos, r, reference_ties=ties)
    assert got.n_edges == ref.n_edges and got.n_nodes == n
    assert torch.equal(got.rowptr, ref.rowptr) and torch.equal(got.src, ref.src) and torch.equal(got.dst, ref.dst)
    assert torch.equal(got.perm.long(), torch.arange(got.n_edges, device=d))
    assert torch.equal(got.edge_index, ei[:, ref.perm.long()])
    if case == "clustered":
        big = ops.radius_csr(pos, 100.0)                     # every pair: one cell
        assert big.n_edges == n * n


def test_cell_list_two_point_sets_and_long_rows():
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    ps = torch.rand(5000, 2, generator=g, dtype=torch.float64).to(d)
    pd = torch.rand(300, 2, generator=g, dtype=torch.float64).to(d)
    r = 0.35                                                  # rows of ~1700 sources; one destination gets > 4096 with r = 2
    rowptr, src, dst = ops.radius_csr_raw(ps, r, pos_dst=pd)
    ei = ops.radius_graph(ps, r, pos_dst=pd)                  # (source in ps, target in pd), source-major
    order = torch.argsort(ei[1] * 5000 + ei[0])
    assert int(rowptr[-1]) == ei.shape[1]
    assert torch.equal(src.long(), ei[0][order]) and torch.equal(dst.long(), ei[1][order])
    rowptr2, src2, dst2 = ops.radius_csr_raw(ps, 2.0, pos_dst=pd[:3])      # rows of 5000 > 4096: cell order, complete
    assert rowptr2.tolist() == [0, 5000, 10000, 15000]
    for k in range(3):
        assert torch.equal(torch.sort(src2[5000 * k:5000 * (k + 1)]).values.long(), torch.arange(5000, device=d))


def test_operator_on_the_cell_list_csr_matches_the_edge_index_path_bit_for_bit():
    """The headline recipe end to end without an edge list: positions -> radius_csr -> attributes by CSR slot -> NNConv."""
    from tests.test_host_logic import DenseNet
    import graph_pde_amd as gp
    d = torch.device("cuda:0")
    torch.manual_seed(4)
    s, r = 41, 0.10
    pos = synth.lattice_positions(s, d)
    a = synth.darcy_coefficient(s, 1).to(d)
    n = s * s
    ei = ops.radius_graph(pos, r)
    ea = synth.darcy_edge_attr(ei, pos, a)
    csr = ops.radius_csr(pos, r)
    ea_csr = synth.darcy_edge_attr(csr.edge_index, pos, a)
    conv = gp.NNConv_old(64, 64, DenseNet([6, 256, 256, 4096], torch.nn.ReLU), aggr="mean").to(d)
    lin = ops.mlp_linears(conv.nn)
    pm = ops.pack_mlp([l.weight for l in lin], [l.bias for l in lin])
    x = torch.randn(n, 64, device=d)
    y_ref = ops.nnconv_forward_raw(x, ops.csr_for(ei, n), ea, pm, conv.root, conv.bias, "mean")
    y_csr = ops.nnconv_forward_raw(x, csr, ea_csr, pm, conv.root, conv.bias, "mean")
    y_na = ops.nnconv_forward_nodeattr_raw(x, csr, gp.NodeAttr.darcy(pos, a), pm, conv.root, conv.bias, "mean")
    assert torch.equal(y_ref, y_csr) and torch.equal(y_ref, y_na)


def test_row_blocks_built_from_positions_equal_the_whole_graph():
    """parallel.partition_rows_by_position (SURVEY.md §8e way 2 without the whole edge list per rank): in-degree count pass
    over all nodes -> balanced bounds -> fill pass over the rank's own destinations.  The blocks tile the whole graph's CSR
    (rowptr / src / dst, attributes by slot) and the operator's rows on a block are the whole-graph rows."""
    from tests.test_host_logic import DenseNet
    import graph_pde_amd as gp
    from graph_pde_amd import parallel
    d = torch.device("cuda:0")
    torch.manual_seed(7)
    s, r = 41, 0.10
    pos = synth.lattice_positions(s, d)
    a = synth.darcy_coefficient(s, 2).to(d)
    n = s * s
    whole = ops.radius_csr(pos, r)
    na = gp.NodeAttr.darcy(pos, a)
    ea_whole = na.materialize(whole.edge_index)
    deg = ops.radius_in_degrees(pos, r)
    assert torch.equal(deg, (whole.rowptr[1:] - whole.rowptr[:-1]))
    conv = gp.NNConv_old(64, 64, DenseNet([6, 256, 256, 4096], torch.nn.ReLU), aggr="mean").to(d)
    x = torch.randn(n, 64, device=d)
    with torch.no_grad():
        y_whole = conv(x, whole, ea_whole)
    for world in (1, 3, 8):
        edges, rows = 0, 0
        for rank in range(world):
            part = parallel.partition_rows_by_position(pos, r, na, rank=rank, world=world)
            lo, hi = part.lo, part.hi
            e0, e1 = int(whole.rowptr[lo]), int(whole.rowptr[hi])
            assert part.n_edges == e1 - e0 and part.csr.n_nodes == n
            assert torch.equal(part.csr.src, whole.src[e0:e1]) and torch.equal(part.csr.dst, whole.dst[e0:e1])
            assert torch.equal(part.csr.rowptr[lo:hi + 1], whole.rowptr[lo:hi + 1] - e0)
            assert int(part.csr.rowptr[:lo + 1].abs().sum()) == 0 and bool((part.csr.rowptr[hi:] == e1 - e0).all())
            assert torch.equal(part.edge_attr, ea_whole[e0:e1])
            with torch.no_grad():
                y = conv(x, part.csr, part.edge_attr)
            err = float((y[lo:hi] - y_whole[lo:hi]).norm() / y_whole[lo:hi].norm())
            assert err <= 1e-6, (world, rank, err)           # same summation order per node; the f16-split scales are per-call maxima
            edges += part.n_edges
            rows += hi - lo
        assert edges == whole.n_edges and rows == n
        if world > 1:
            sizes = [parallel.partition_rows_by_position(pos, r, None, rank=k, world=world).n_edges for k in range(world)]
            assert max(sizes) - min(sizes) <= 2 * int(deg.max())              # balanced on in-edges
