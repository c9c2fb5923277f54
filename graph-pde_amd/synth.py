"""Synthetic inputs of the reference's shapes (the real `.mat` datasets are not available).

Everything here follows SURVEY.md §8(d):

* lattice radius graphs in *exact integer arithmetic*, emitted in the reference's edge order
  (`np.vstack(np.where(pwd <= r))`: sorted by source, then by target; self-loops included;
  `edge_index[0]` = source j, `edge_index[1]` = target i) —
  /root/reference/graph-neural-operator/utilities.py:250-255;
* node order = `np.meshgrid(..., indexing='xy')` raveled, i.e. node id = iy*s + ix, position
  (x, y) = (lin[ix], lin[iy]) — utilities.py:236-248;
* edge attributes `[pos_src(2), pos_dst(2), a_src, a_dst]` as float32 — utilities.py:269-277;
* a piecewise-constant Darcy-like coefficient field {3, 12}, Gaussian-normalised
  (utilities.py:113-119).

torch is used for tensor plumbing only; the generators run on CPU or on the GPU (`device`).
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np
import torch


def lattice_r2(s: int, r: float) -> int:
    """Squared radius in integer lattice units for grid spacing 1/(s-1)."""
    return int(math.floor((r * (s - 1)) ** 2 + 1e-9))


def lattice_radius_graph(s: int, r: float, device="cpu") -> torch.Tensor:
    """edge_index int64 [2, E]: edge (j -> i) iff dx^2 + dy^2 <= R2 on the s x s lattice.

    Exact integer arithmetic, so the graph is symmetric (the reference's float64 dot-product
    expansion drops some pairs at exactly distance r; parity tests always feed the *same*
    edge_index to both sides).  Order: sorted by source, then target.
    """
    r2 = lattice_r2(s, r)
    rad = int(math.isqrt(r2))
    offs = [(dy, dx) for dy in range(-rad, rad + 1) for dx in range(-rad, rad + 1)
            if dx * dx + dy * dy <= r2]
    offs.sort()                                   # (dy, dx) lexicographic == increasing target id
    dy = torch.tensor([o[0] for o in offs], device=device, dtype=torch.int64)
    dx = torch.tensor([o[1] for o in offs], device=device, dtype=torch.int64)
    n = s * s
    node = torch.arange(n, device=device, dtype=torch.int64)
    ix, iy = node % s, node // s
    # process sources in slabs to bound the [n, n_off] mask
    slab = max(1, (1 << 27) // max(1, len(offs)))
    src_l, dst_l = [], []
    for a in range(0, n, slab):
        b = min(n, a + slab)
        tx = ix[a:b, None] + dx[None, :]
        ty = iy[a:b, None] + dy[None, :]
        ok = (tx >= 0) & (tx < s) & (ty >= 0) & (ty < s)
        nz = ok.nonzero(as_tuple=False)           # row-major: by source, then by offset
        src = nz[:, 0] + a
        dst = src + dy[nz[:, 1]] * s + dx[nz[:, 1]]
        src_l.append(src)
        dst_l.append(dst)
    return torch.stack([torch.cat(src_l), torch.cat(dst_l)], dim=0)


def lattice_positions(s: int, device="cpu") -> torch.Tensor:
    """float64 [N, 2] positions (x, y), node id = iy*s + ix (meshgrid 'xy')."""
    # numpy's linspace, as the reference's SquareMeshGenerator uses (utilities.py:242): torch.linspace
    # differs from it in the last bit of some float64 values
    lin = torch.from_numpy(np.linspace(0.0, 1.0, s)).to(device)
    node = torch.arange(s * s, device=device)
    return torch.stack([lin[node % s], lin[node // s]], dim=1)


def darcy_coefficient(s: int, seed: int = 0) -> torch.Tensor:
    """float32 [N]: a = 12 where a smooth Gaussian random field is > 0 else 3, then
    (a - mean) / (std + 1e-5)  (GaussianNormalizer, utilities.py:113-119)."""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    grf = gaussian_filter(rng.standard_normal((s, s)), sigma=s / 16.0)
    a = np.where(grf > 0, 12.0, 3.0).reshape(-1)
    a = (a - a.mean()) / (a.std() + 1e-5)
    return torch.from_numpy(a.astype(np.float32))


def darcy_edge_attr(edge_index: torch.Tensor, pos: torch.Tensor, a: torch.Tensor) -> torch.Tensor:
    """float32 [E, 6] = [pos_src(2), pos_dst(2), a_src, a_dst]  (utilities.py:269-277).

    Built column by column from 1-D gathers.  Rounds 1-2 wrote `out[:, 0:2] = pos[src].to(float32)` (a [E, 2] float64
    gather copied into a strided slice): on torch 2.10 + ROCm 7 that leaves ZEROS in the position columns of every row
    past the first E - 2^26 once E exceeds 2^26 - found in round 3 when the node-table kernel (row f3), which reads the
    positions itself, disagreed with the tensor path on the 241^2 graph (95.5 M edges).  tests/test_gpu_headline.py now
    checks the headline tensor against the node table over ALL edges."""
    src, dst = edge_index[0], edge_index[1]
    p32 = pos.to(torch.float32)
    a = a.to(pos.device, torch.float32)
    cols = [p32[src, 0], p32[src, 1], p32[dst, 0], p32[dst, 1], a[src], a[dst]]
    return torch.stack(cols, dim=1)


def darcy_graph(s: int, r: float, device="cpu", seed: int = 0
                ) -> Tuple[torch.Tensor, torch.Tensor, int]:
    """(edge_index [2,E] int64, edge_attr [E,6] float32, N) for the Darcy-shaped lattice."""
    ei = lattice_radius_graph(s, r, device)
    pos = lattice_positions(s, device)
    a = darcy_coefficient(s, seed)
    return ei, darcy_edge_attr(ei, pos, a), s * s


def burgers_coefficient(s: int, seed: int = 0) -> np.ndarray:
    """float32 [s]: a smooth periodic Gaussian random field, normalised (stands in for the Burgers
    initial condition `a` of burgers_data_R10.mat)."""
    rng = np.random.default_rng(seed)
    from scipy.ndimage import gaussian_filter1d
    a0 = gaussian_filter1d(rng.standard_normal(s), sigma=s / 32.0, mode="wrap")
    return ((a0 - a0.mean()) / (a0.std() + 1e-5)).astype(np.float32)


def burgers_multipole_graphs(s: int, device="cpu", seed: int = 0, periodic: bool = True):
    """1-D multipole graphs of the MGKN-orthogonal shape — a vectorised restatement of
    `multi_pole_grid1d` + `get_edge_attr`
    (/root/reference/multipole-graph-neural-operator/utilities.py:1702-1777; called with
    is_periodic=True at MGKN_orthogonal_burgers1d.py:165).

    Levels l = 1..log2(s)-1 hold s_l = s / 2^(l-1) nodes on linspace(0,1,s_l); the coefficient is
    *sub-sampled* with stride 2^(l-1).  Graph 0 = nearest neighbours (x_i -> x_i-1, x_i+1) on the
    finest level; then one "interactive neighbour" graph per level: x_j = x_i + d, 2 <= |d| <= 3,
    kept iff `abs(x_i//2 - x_j//2) % (s_l//2) <= 1`.  Edges are emitted in the reference's loop
    order (by x_i, then by offset); edge_index[0] = x_i, edge_index[1] = x_j.
    Returns [(edge_index [2,E], edge_attr [E,4] = [grid[x_i], grid[x_j], a[x_i], a[x_j]], s_l)].
    """
    a0 = burgers_coefficient(s, seed)
    level = int(np.log2(s) - 1)
    graphs = []

    def emit(xi, xj, grid, a_l, s_l):
        ei = np.stack([xi, xj]).astype(np.int64)
        ea = np.stack([grid[xi], grid[xj], a_l[xi], a_l[xj]], axis=1).astype(np.float32)
        graphs.append((torch.from_numpy(ei).to(device),
                       torch.from_numpy(ea.reshape(-1, 4)).to(device), s_l))

    for l in range(1, level + 1):
        r_l = 2 ** (l - 1)
        s_l = s // r_l
        grid = np.linspace(0.0, 1.0, s_l).astype(np.float32)
        a_l = a0[::r_l]
        xi_all = np.arange(s_l)
        if l == 1:
            xi = np.repeat(xi_all, 2)
            xj = xi + np.tile(np.array([-1, 1]), s_l)
            if periodic:
                xj = xj % s_l
            ok = (xj >= 0) & (xj < s_l)
            emit(xi[ok], xj[ok], grid, a_l, s_l)
        d = np.array([-3, -2, 2, 3])
        xi = np.repeat(xi_all, d.size)
        xj = xi + np.tile(d, s_l)
        if periodic:
            xj = xj % s_l
        ok = (xj >= 0) & (xj < s_l)
        ok &= (np.abs(xi // 2 - xj // 2) % max(s_l // 2, 1)) <= 1
        emit(xi[ok], xj[ok], grid, a_l, s_l)
    return graphs


def sampled_multilevel_graphs(s: int, m, radii_inner, radii_inter, device="cpu", seed: int = 0,
                              idx=None, a_all=None):
    """Multi-level sampled radius graphs of the MGKN-general shape
    (/root/reference/multipole-graph-neural-operator/utilities.py:546-712
    `RandomMultiMeshGenerator`, restated): level l holds m[l] points sampled without replacement
    from the s x s lattice; inner graphs connect points of one level within radii_inner[l];
    inter-level "down" graphs connect level l (source) to level l+1 (target) within
    radii_inter[l]; "up" graphs are the row-swapped down graphs (utilities.py:631).

    Returns dict with per-level inner graphs and inter graphs as
    (edge_index [2,E] (local node ids), edge_attr [E,6], n_src_nodes, n_dst_nodes).
    `idx` (per-level lattice indices) / `a_all` (coefficient on the lattice) override the sampling --
    used to pin this restatement against the reference's own generator (tests/golden).
    """
    n = s * s
    pos_all = lattice_positions(s).numpy()
    if a_all is None:
        a_all = darcy_coefficient(s, seed).numpy()
    a_all = np.asarray(a_all)
    if idx is None:
        # consecutive slices of one random permutation, unsorted, as RandomMultiMeshGenerator.sample does
        perm = np.random.default_rng(seed).permutation(n)
        idx, off = [], 0
        for ml in m:
            idx.append(perm[off:off + ml])
            off += ml
    idx = [np.asarray(i) for i in idx]
    from sklearn.metrics import pairwise_distances

    def attrs(pi, pj, ai, aj, src, dst):
        ea = np.concatenate([pi[src], pj[dst], ai[src, None], aj[dst, None]], axis=1)
        return torch.from_numpy(ea.astype(np.float32)).to(device)

    inner, down, up = [], [], []
    for l, il in enumerate(idx):
        p, a = pos_all[il], a_all[il]
        pwd = pairwise_distances(p)
        src, dst = np.where(pwd <= radii_inner[l])
        inner.append((torch.from_numpy(np.stack([src, dst]).astype(np.int64)).to(device),
                      attrs(p, p, a, a, src, dst), len(il), len(il)))
    for l in range(len(idx) - 1):
        p0, a0 = pos_all[idx[l]], a_all[idx[l]]
        p1, a1 = pos_all[idx[l + 1]], a_all[idx[l + 1]]
        pwd = pairwise_distances(p0, p1)
        src, dst = np.where(pwd <= radii_inter[l])
        down.append((torch.from_numpy(np.stack([src, dst]).astype(np.int64)).to(device),
                     attrs(p0, p1, a0, a1, src, dst), len(idx[l]), len(idx[l + 1])))
        up.append((torch.from_numpy(np.stack([dst, src]).astype(np.int64)).to(device),
                   attrs(p1, p0, a1, a0, dst, src), len(idx[l + 1]), len(idx[l])))
    return {"inner": inner, "down": down, "up": up}
