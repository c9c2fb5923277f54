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
