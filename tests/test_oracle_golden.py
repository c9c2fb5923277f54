"""The CPU oracle against the committed reference vectors (CPU tier).

The vectors were produced by the reference's own NNConv_old + DenseNet classes
(tests/golden/make_golden.py); the oracle must reproduce them — that is what pins it."""
import pytest
import torch

from oracle.nnconv_oracle import nnconv_forward, rel_l2


def test_oracle_fp32_matches_reference(golden):
    g = golden
    y = nnconv_forward(g["x"], g["edge_index"], g["edge_attr"], g["weights"], g["biases"],
                       g["root"], g["bias"], aggr=g["aggr"], dtype=torch.float32)
    # same op order as the reference; only the destination scatter is chunked
    assert rel_l2(y, g["out_f32"]) <= 2e-7, rel_l2(y, g["out_f32"])


def test_oracle_fp64_matches_reference(golden):
    g = golden
    y = nnconv_forward(g["x"], g["edge_index"], g["edge_attr"], g["weights"], g["biases"],
                       g["root"], g["bias"], aggr=g["aggr"], dtype=torch.float64)
    assert rel_l2(y, g["out_f64"]) <= 1e-13, rel_l2(y, g["out_f64"])


def test_oracle_chunking_is_immaterial(golden):
    g = golden
    a = nnconv_forward(g["x"], g["edge_index"], g["edge_attr"], g["weights"], g["biases"],
                       g["root"], g["bias"], aggr=g["aggr"], dtype=torch.float64, chunk_edges=97)
    assert rel_l2(a, g["out_f64"]) <= 1e-13


def test_oracle_1d_inputs_and_isolated_nodes():
    # nn_conv.py:269-270: 1-D x / edge_attr are promoted; zero in-degree rows get root/bias only
    torch.manual_seed(0)
    n, e = 9, 20
    ei = torch.stack([torch.randint(0, n, (e,)), torch.randint(0, 5, (e,))])
    x = torch.randn(n)
    ea = torch.randn(e)
    ws = [torch.randn(4, 1), torch.randn(1, 4)]
    bs = [torch.randn(4), torch.randn(1)]
    root, bias = torch.randn(1, 1), torch.randn(1)
    y = nnconv_forward(x, ei, ea, ws, bs, root, bias, aggr="mean", in_channels=1, out_channels=1)
    assert y.shape == (n, 1)
    iso = torch.arange(5, n)
    assert torch.allclose(y[iso, 0], x[iso] * root[0, 0] + bias[0])


def test_mesh_fixture_pins_graph_and_attribute_construction():
    """tests/golden/mesh_s12.npz was produced by the reference's own SquareMeshGenerator
    (graph-neural-operator/utilities.py:228-285).  Our restatements of it must reproduce it exactly:
    the lattice positions, the radius graph in np.where order, and the edge-attribute recipe that
    NodeAttr.darcy (row f3) reads from node data."""
    import os
    import numpy as np
    import graph_pde_amd as gp
    from graph_pde_amd import synth
    from tests.conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "mesh_s12.npz"))
    s, r = int(g["s"]), float(g["r"])
    pos = synth.lattice_positions(s)
    assert np.array_equal(pos.numpy(), g["grid"])
    assert np.array_equal(pos.float().numpy(), g["grid_f32"])
    ei = synth.lattice_radius_graph(s, r)
    assert np.array_equal(ei.numpy(), g["edge_index"])
    a = torch.from_numpy(g["a"])
    assert np.array_equal(synth.darcy_edge_attr(ei, pos, a.float()).numpy(), g["edge_attr"])
    na = gp.NodeAttr.darcy(pos, a)
    assert np.array_equal(na.materialize(ei).numpy(), g["edge_attr"])


def test_burgers_graph_fixture_pins_the_multipole_graph_family():
    """tests/golden/burgers_graphs_s64.npz: the reference's own multi_pole_grid1d + get_edge_attr
    (multipole-graph-neural-operator/utilities.py:1702-1777, is_periodic=True).  The vectorised
    restatement in synth.burgers_multipole_graphs (BASELINE config 3's graph family) must reproduce every
    graph: edges in the reference's loop order and the [grid_i, grid_j, a_i, a_j] attributes."""
    import os
    import numpy as np
    from graph_pde_amd import synth
    from tests.conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "burgers_graphs_s64.npz"))
    s = int(g["s"])
    assert np.array_equal(synth.burgers_coefficient(s, 0), g["a"])
    graphs = synth.burgers_multipole_graphs(s, seed=0, periodic=True)
    assert len(graphs) == int(g["n_graphs"])
    for i, (ei, ea, n) in enumerate(graphs):
        assert n == int(g[f"n{i}"]), i
        assert np.array_equal(ei.numpy(), g[f"ei{i}"]), i
        assert np.array_equal(ea.numpy(), g[f"ea{i}"]), i


def test_mgkn_graph_fixture_pins_the_sampled_multilevel_family():
    """tests/golden/mgkn_graphs_s20.npz: the reference's own RandomMultiMeshGenerator
    (multipole-graph-neural-operator/utilities.py:546-712: sample, ball_connectivity, attributes(theta)).
    Given the same sampled lattice indices, synth.sampled_multilevel_graphs (BASELINE config 4's graph
    family) must reproduce the inner / down / up graphs and their [pos_src, pos_dst, a_src, a_dst]
    attributes, edge order included (the reference numbers nodes globally, level after level)."""
    import os
    import numpy as np
    from graph_pde_amd import synth
    from tests.conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "mgkn_graphs_s20.npz"))
    m = [int(v) for v in g["m"]]
    idx = [g[f"idx{l}"] for l in range(len(m))]
    out = synth.sampled_multilevel_graphs(int(g["s"]), m, list(g["radii_inner"]), list(g["radii_inter"]),
                                          idx=idx, a_all=g["a"])
    offs = np.concatenate([[0], np.cumsum(m)])
    for l in range(len(m)):
        lo, hi = g["range"][l]
        ei, ea, ns, nd = out["inner"][l]
        assert (ns, nd) == (m[l], m[l])
        assert np.array_equal(ei.numpy() + offs[l], g["edge_index"][:, lo:hi]), l
        assert np.array_equal(ea.numpy(), g["edge_attr"][lo:hi]), l
    for l in range(len(m) - 1):
        lo, hi = g["range_down"][l]
        ei, ea, ns, nd = out["down"][l]
        assert np.array_equal(ei.numpy() + np.array([[offs[l]], [offs[l + 1]]]), g["edge_index_down"][:, lo:hi]), l
        assert np.array_equal(ea.numpy(), g["edge_attr_down"][lo:hi]), l
        ei, ea, ns, nd = out["up"][l]
        assert np.array_equal(ei.numpy() + np.array([[offs[l + 1]], [offs[l]]]), g["edge_index_up"][:, lo:hi]), l
        assert np.array_equal(ea.numpy(), g["edge_attr_up"][lo:hi]), l


GRAD_CASES = ["ragged_add", "mlp2_mean_noroot", "burgers_k4", "ckpt_torus_m100"]


def load_golden_grads(name):
    import os
    import numpy as np
    from tests.conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, name + "_grad.npz"))
    n = sum(1 for k in z.files if k.startswith("gW"))
    return {"gout": torch.from_numpy(z["gout"]), "gx": torch.from_numpy(z["gx"]),
            "gW": [torch.from_numpy(z[f"gW{i}"]) for i in range(n)],
            "gb": [torch.from_numpy(z[f"gb{i}"]) for i in range(n)],
            "groot": torch.from_numpy(z["groot"]) if "groot" in z.files else None,
            "gbias": torch.from_numpy(z["gbias"]) if "gbias" in z.files else None}


@pytest.mark.parametrize("name", GRAD_CASES)
def test_oracle_gradients_match_autograd_through_the_reference_module(name):
    """tests/golden/<name>_grad.npz: float64 autograd through the reference's own NNConv_old / DenseNet
    (make_golden.py).  The restated operator's autograd (oracle.nnconv_grads, the checker of the native
    backward) must give the same gradients."""
    from oracle.nnconv_oracle import nnconv_grads
    from tests.conftest import load_golden
    g, r = load_golden(name), load_golden_grads(name)
    gx, gW, gb, groot, gbias = nnconv_grads(g["x"], g["edge_index"], g["edge_attr"], g["weights"], g["biases"],
                                            g["root"], g["bias"], g["aggr"], r["gout"])
    tol = 1e-12
    assert rel_l2(gx, r["gx"]) <= tol
    for l in range(len(gW)):
        assert rel_l2(gW[l], r["gW"][l]) <= tol and rel_l2(gb[l], r["gb"][l]) <= tol, l
    assert (groot is None) == (r["groot"] is None) and (gbias is None) == (r["gbias"] is None)
    if groot is not None:
        assert rel_l2(groot, r["groot"]) <= tol
    if gbias is not None:
        assert rel_l2(gbias, r["gbias"]) <= tol


@pytest.mark.parametrize("name", GRAD_CASES)
def test_oracle_gradients_in_edge_chunks_equal_the_whole(name):
    """nnconv_grads(chunk_edges=...) - the form the MGKN-scale gradient tests use (the [E, 4096] float64 tensor of a
    131 k-edge call is 4.3 GB) - is the same sum taken in pieces: equal to the pinned gradients to float64 rounding."""
    from oracle.nnconv_oracle import nnconv_grads
    from tests.conftest import load_golden
    g, r = load_golden(name), load_golden_grads(name)
    e = g["edge_index"].shape[1]
    gx, gW, gb, groot, gbias = nnconv_grads(g["x"], g["edge_index"], g["edge_attr"], g["weights"], g["biases"],
                                            g["root"], g["bias"], g["aggr"], r["gout"], chunk_edges=max(1, e // 3))
    tol = 1e-12
    assert rel_l2(gx, r["gx"]) <= tol
    for l in range(len(gW)):
        assert rel_l2(gW[l], r["gW"][l]) <= tol and rel_l2(gb[l], r["gb"][l]) <= tol, l
    if groot is not None:
        assert rel_l2(groot, r["groot"]) <= tol
    if gbias is not None:
        assert rel_l2(gbias, r["gbias"]) <= tol


@pytest.mark.parametrize("name", GRAD_CASES)
def test_oracle_shared_application_gradients_equal_the_sum_over_applications(name):
    """nnconv_grads_shared (one conv applied `depth` times, UAI1_full_resolution.py:29-30: the checker of the depth-deferred
    backward at sizes where `depth` separate float64 passes would take minutes): with ONE application it is the pinned
    gradient; with three it is the sum of three pinned-operator gradients."""
    from oracle.nnconv_oracle import nnconv_grads, nnconv_grads_shared
    from tests.conftest import load_golden
    g, r = load_golden(name), load_golden_grads(name)
    e = g["edge_index"].shape[1]
    args = (g["edge_index"], g["edge_attr"], g["weights"], g["biases"], g["root"], g["bias"], g["aggr"])
    gxs, gW, gb, groot, gbias = nnconv_grads_shared([g["x"]], *args, [r["gout"]], chunk_edges=max(1, e // 3))
    tol = 1e-12
    assert rel_l2(gxs[0], r["gx"]) <= tol
    for l in range(len(gW)):
        assert rel_l2(gW[l], r["gW"][l]) <= tol and rel_l2(gb[l], r["gb"][l]) <= tol, l
    torch.manual_seed(3)
    xs = [g["x"], torch.randn_like(g["x"]) * 0.5, torch.randn_like(g["x"]) * 2.0]
    gs = [r["gout"], torch.randn_like(r["gout"]), torch.randn_like(r["gout"]) * 0.25]
    gxs, gW, gb, groot, gbias = nnconv_grads_shared(xs, *args, gs, chunk_edges=max(1, e // 2))
    sW = [torch.zeros_like(w, dtype=torch.float64) for w in g["weights"]]
    sb = [torch.zeros_like(b, dtype=torch.float64) for b in g["biases"]]
    sroot = None if g["root"] is None else torch.zeros_like(g["root"], dtype=torch.float64)
    for i, (x, go) in enumerate(zip(xs, gs)):
        ox, oW, ob, oroot, obias = nnconv_grads(x, *args, go)
        assert rel_l2(gxs[i], ox) <= tol
        for l in range(len(sW)):
            sW[l] += oW[l]; sb[l] += ob[l]
        if sroot is not None:
            sroot += oroot
    for l in range(len(sW)):
        assert rel_l2(gW[l], sW[l]) <= tol and rel_l2(gb[l], sb[l]) <= tol, l
    if sroot is not None:
        assert rel_l2(groot, sroot) <= tol


@pytest.mark.parametrize("name", GRAD_CASES)
def test_grad_x_rows_is_the_pinned_grad_x(name):
    """oracle.nnconv_grad_x_rows - rows of d loss / d x from the out-edges of the chosen nodes only, the checker of grad_x at the
    241^2 scale (tests/test_gpu_headline_train.py) - against the gradients pinned to autograd through the reference's own module
    (tests/golden/<name>_grad.npz), on a subset of the nodes fed with exactly their out-edges."""
    from oracle.nnconv_oracle import nnconv_grad_x_rows
    from tests.conftest import load_golden
    g, r = load_golden(name), load_golden_grads(name)
    ei, n = g["edge_index"], g["x"].shape[0]
    rows = torch.arange(0, n, 3)
    keep = torch.isin(ei[0], rows)
    deg = torch.bincount(ei[1], minlength=n) if g["aggr"] == "mean" else None
    ea = g["edge_attr"] if g["edge_attr"].dim() == 2 else g["edge_attr"].unsqueeze(-1)
    gx = nnconv_grad_x_rows(rows, ei[:, keep], ea[keep], g["weights"], g["biases"], g["root"], r["gout"], deg,
                            chunk_edges=max(1, int(keep.sum()) // 3))
    ref = r["gx"] if r["gx"].dim() == 2 else r["gx"].unsqueeze(-1)
    assert rel_l2(gx, ref[rows]) <= 1e-12


def _lattice(s):
    import numpy as np
    g = np.linspace(0.0, 1.0, s)
    return np.vstack([xx.ravel() for xx in np.meshgrid(g, g)]).T          # utilities.py:247 ('xy' meshgrid)


def test_radius_oracle_reproduces_the_reference_at_tie_radii():
    """tests/golden/mesh_ties.npz comes from the reference's own SquareMeshGenerator.ball_connectivity(0.10) on the
    31^2 and 61^2 lattices, where thousands of pairs sit at exactly distance r and scikit-learn's rounding keeps only
    some of them.  oracle/radius_oracle.c (the C restatement of that arithmetic) must return the same edge list."""
    import hashlib
    import os
    import numpy as np
    from oracle import radius_oracle
    from tests.conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "mesh_ties.npz"))
    r = float(g["r"])
    ei31 = radius_oracle.radius_edges(_lattice(31), r)
    assert np.array_equal(ei31, g["edge_index_s31"].astype(np.int64))
    for s in (31, 61):
        ei = radius_oracle.radius_edges(_lattice(s), r)
        assert ei.shape[1] == int(g[f"n_edges_s{s}"]) == {31: 22951, 61: 376471}[s]
        assert np.array_equal(np.bincount(ei[0], minlength=s * s), g[f"outdeg_s{s}"])
        assert np.array_equal(np.bincount(ei[1], minlength=s * s), g[f"indeg_s{s}"])
        assert hashlib.sha256(np.ascontiguousarray(ei).tobytes()).hexdigest() == str(g[f"sha256_s{s}"])
        # the reference graph is NOT symmetric at a tie radius; the exact float64 test is, and keeps more pairs
        ex = radius_oracle.radius_edges(_lattice(s), r, reference_ties=False)
        assert ex.shape[1] > ei.shape[1]
        pairs = set(map(tuple, ex.T.tolist()))
        assert all((b, a) in pairs for a, b in list(pairs)[:2000])
    assert not np.array_equal(np.bincount(ei31[0], minlength=961), np.bincount(ei31[1], minlength=961))


def test_radius_oracle_two_point_sets_match_the_multilevel_fixture():
    """Inter-level graphs are pairwise_distances(X, Y) <= r between two sampled point sets
    (multipole-graph-neural-operator/utilities.py:617-632); the fixture is the reference generator's output."""
    import os
    import numpy as np
    from graph_pde_amd import synth
    from oracle import radius_oracle
    from tests.conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "mgkn_graphs_s20.npz"))
    m = [int(v) for v in g["m"]]
    pos = synth.lattice_positions(int(g["s"])).double().numpy()
    pts = [pos[g[f"idx{l}"]] for l in range(len(m))]
    offs = np.concatenate([[0], np.cumsum(m)])
    for l in range(len(m)):
        lo, hi = g["range"][l]
        assert np.array_equal(radius_oracle.radius_edges(pts[l], float(g["radii_inner"][l])) + offs[l], g["edge_index"][:, lo:hi])
    for l in range(len(m) - 1):
        lo, hi = g["range_down"][l]
        ei = radius_oracle.radius_edges(pts[l], float(g["radii_inter"][l]), y=pts[l + 1])
        assert np.array_equal(ei + np.array([[offs[l]], [offs[l + 1]]]), g["edge_index_down"][:, lo:hi])
