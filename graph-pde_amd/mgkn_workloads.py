"""NNConv call patterns of the two MGKN configurations of BASELINE.json (configs 3 and 4), on synthetic
graphs of the reference's generators (synth.py, pinned by tests/golden) - what bench.py's `mgkn` object and
tests/test_gpu_mgkn.py time and check.

* config 3, MGKN-orthogonal Burgers-1D, s = 8192 (/root/reference/multipole-graph-neural-operator/
  MGKN_orthogonal_burgers1d.py:27-37,59-86): level + 1 = 13 `NNConv(64, 64, DenseNet([4, k_l, k_l, 4096]))`
  modules, k_l = max(1024 // 2^l, 16); per V-cycle sweep every level does
  `x = relu(x + conv_l(phi_l, edge_index_l, edge_attr_l))`; depth 4  ->  52 NNConv calls per forward.
* config 4, MGKN-general Darcy-2D, m = [2400, 1600, 400, 100, 25] sampled from the 421^2 lattice
  (MGKN_general_darcy2d.py:43-61,69-94; neurips1_MGKN.py:112-114): inner kernels
  DenseNet([6, k_l, k_l, 4096]), inter-level kernels DenseNet([6, k_l, 4096]), k_l = 256 // 2^l, all
  aggr='mean'; per sweep: down `x = relu(x + conv_down(x))`, then per level `x[a:b] = conv_inner(x[a:b])`
  and `x = relu(x + conv_up(x))`; depth 5  ->  13 * 5 = 65 NNConv calls per forward.

Only the NNConv calls and their elementwise glue are modelled (the hot path); the
1x1 Linear lifts, the Burgers down/up-sampling and the data pipeline are outside it.
"""
from __future__ import annotations

from typing import Callable, Dict, List

import torch
import torch.nn.functional as F

from . import synth
from .nn_conv import NNConv, nnconv_group


def dense_net(layers: List[int]) -> torch.nn.Sequential:
    """Linear / ReLU chain with DenseNet's layout (utilities.py:201-227)."""
    mods: List[torch.nn.Module] = []
    for j in range(len(layers) - 1):
        mods.append(torch.nn.Linear(layers[j], layers[j + 1]))
        if j != len(layers) - 2:
            mods.append(torch.nn.ReLU())
    return torch.nn.Sequential(*mods)


class Workload:
    def __init__(self, name: str, calls: int, edge_applications: int, forward: Callable[[], List[torch.Tensor]],
                 pairs: List[tuple], description: str, train_step: Callable[[], float] = None, modules: List[torch.nn.Module] = None):
        self.name = name
        self.calls = calls                          # NNConv calls per model forward
        self.edge_applications = edge_applications  # sum over the calls of their edge counts
        self.forward = forward                      # runs one model forward (no_grad), returns the final states
        self.pairs = pairs                          # distinct (conv, x, edge_index, edge_attr) of the forward
        self.description = description
        self.train_step = train_step                # one optimisation step (forward with autograd, loss, backward, Adam): the
                                                    # scripts' inner loop (MGKN_general_darcy2d.py:260-282, MGKN_orthogonal_burgers1d.py:226-242)
        self.modules = modules or []                # the NNConv modules (their parameters are what the step updates)


def orthogonal_burgers(device, s: int = 8192, depth: int = 4, ker_width: int = 1024, seed: int = 0,
                       fused_glue: bool = False, grouped: bool = False, capturable: bool = False) -> Workload:
    """`capturable`: Adam keeps its step count on the device, so that `train_step` can be recorded by `gp.capture`."""
    torch.manual_seed(seed)
    graphs = [(ei.to(device), ea.to(device), n) for ei, ea, n in synth.burgers_multipole_graphs(s, seed=seed)]
    nlev = len(graphs)                              # level + 1 graphs: nearest neighbours + one per level
    convs = [NNConv(64, 64, dense_net([4, max(ker_width // 2 ** l, 16), max(ker_width // 2 ** l, 16), 4096]),
                    aggr="mean").to(device) for l in range(nlev)]
    phis = [torch.randn(n, 64, device=device) for _, _, n in graphs]
    edges = sum(int(g[0].shape[1]) for g in graphs)

    def sweep_all():
        xs = [p.clone() for p in phis]
        for _ in range(depth):                  # MGKN_orthogonal_burgers1d.py:65-82
            if grouped:
                # the 13 convs of a sweep read phi[l], fixed by the downward pass (:67-71): independent calls,
                # one grouped launch (nn_conv.nnconv_group), the relu(x + conv) glue inside it
                xs = nnconv_group([(convs[l], phis[l], graphs[l][0], graphs[l][1], xs[l], "relu") for l in range(nlev)])
                continue
            for l in reversed(range(nlev)):
                if fused_glue:      # opt-in: the relu(x + conv) glue inside the operator's last kernel
                    xs[l] = convs[l](phis[l], graphs[l][0], graphs[l][1], residual=xs[l], activation="relu")
                else:
                    xs[l] = F.relu(xs[l] + convs[l](phis[l], graphs[l][0], graphs[l][1]))
        return xs

    def forward():
        with torch.no_grad():
            return sweep_all()

    opt = torch.optim.Adam([p for c in convs for p in c.parameters()], lr=1e-3, weight_decay=5e-4, capturable=capturable)   # :209-211

    def train_step():
        opt.zero_grad(set_to_none=True)
        loss = sum(x_.square().mean() for x_ in sweep_all())
        loss.backward()
        opt.step()
        return loss

    pairs = [(convs[l], phis[l], graphs[l][0], graphs[l][1]) for l in range(nlev)]
    return Workload("mgkn_orthogonal_burgers1d", nlev * depth, edges * depth, forward, pairs,
                    f"MGKN-orthogonal Burgers-1D s={s}: {nlev} levels, depth {depth}, kernel widths "
                    f"max({ker_width}//2^l,16), {edges} edges per sweep", train_step, convs)


def general_darcy(device, s: int = 421, depth: int = 5, ker_width: int = 256, seed: int = 0,
                  fused_glue: bool = False, capturable: bool = False) -> Workload:
    torch.manual_seed(seed)
    m = [2400, 1600, 400, 100, 25]
    r_inner = [0.5 / 8 * 1.41, 0.5 / 8, 0.5 / 4, 0.5 / 2, 0.5]
    r_inter = [0.5 / 8 * 1.1, 0.5 / 8 * 1.41, 0.5 / 4 * 1.41, 0.5 / 2 * 1.41]
    g = synth.sampled_multilevel_graphs(s, m, r_inner, r_inter, device=device, seed=seed)
    L = len(m)
    offs = [0]
    for ml in m:
        offs.append(offs[-1] + ml)
    x0 = torch.randn(offs[-1], 64, device=device)
    inner = [NNConv(64, 64, dense_net([6, ker_width // 2 ** l, ker_width // 2 ** l, 4096]), aggr="mean",
                    root_weight=True, bias=False).to(device) for l in range(L)]
    down = [NNConv(64, 64, dense_net([6, ker_width // 2 ** (l + 1), 4096]), aggr="mean", root_weight=False,
                   bias=False).to(device) for l in range(L - 1)]
    up = [NNConv(64, 64, dense_net([6, ker_width // 2 ** (l + 1), 4096]), aggr="mean", root_weight=False,
                 bias=False).to(device) for l in range(L - 1)]
    # global node ids for the inter-level graphs (the script indexes one concatenated x)
    gd = [(torch.stack([g["down"][l][0][0] + offs[l], g["down"][l][0][1] + offs[l + 1]]), g["down"][l][1])
          for l in range(L - 1)]
    gu = [(torch.stack([g["up"][l][0][0] + offs[l + 1], g["up"][l][0][1] + offs[l]]), g["up"][l][1])
          for l in range(L - 1)]
    edges = sum(int(g["inner"][l][0].shape[1]) for l in range(L)) + \
        sum(int(gd[l][0].shape[1]) + int(gu[l][0].shape[1]) for l in range(L - 1))

    def sweep_all(train=False):
        xx = x0
        for _ in range(depth):                  # MGKN_general_darcy2d.py:76-90
            for l in range(L - 1):
                if fused_glue:
                    xx = down[l](xx, gd[l][0], gd[l][1], residual=xx, activation="relu")
                else:
                    xx = F.relu(xx + down[l](xx, gd[l][0], gd[l][1]))
            for l in reversed(range(L)):
                a, b = offs[l], offs[l + 1]
                if xx is x0 or train:           # never write into the caller's tensor (the script's x is its own fc_in output);
                    xx = xx.clone()             # with autograd: nor into a tensor F.relu saved (torch 2.x saves the OUTPUT)
                # in place on the running state, input slice cloned - as the script does (MGKN_general_darcy2d.py:84-86)
                xx[a:b] = inner[l](xx[a:b].clone(), g["inner"][l][0], g["inner"][l][1])
                if l > 0:
                    if fused_glue:
                        xx = up[l - 1](xx, gu[l - 1][0], gu[l - 1][1], residual=xx, activation="relu")
                    else:
                        xx = F.relu(xx + up[l - 1](xx, gu[l - 1][0], gu[l - 1][1]))
        return [xx]

    def forward():
        with torch.no_grad():
            return sweep_all()

    opt = torch.optim.Adam([p for c in inner + down + up for p in c.parameters()], lr=1e-3, weight_decay=5e-4,
                           capturable=capturable)   # :240-242

    def train_step():
        opt.zero_grad(set_to_none=True)
        loss = sweep_all(train=True)[0].square().mean()
        loss.backward()
        opt.step()
        return loss

    pairs = [(inner[l], x0[offs[l]:offs[l + 1]].contiguous(), g["inner"][l][0], g["inner"][l][1]) for l in range(L)]
    pairs += [(down[l], x0, gd[l][0], gd[l][1]) for l in range(L - 1)]
    pairs += [(up[l], x0, gu[l][0], gu[l][1]) for l in range(L - 1)]
    return Workload("mgkn_general_darcy2d", (3 * L - 2) * depth, edges * depth, forward, pairs,
                    f"MGKN-general Darcy-2D m={m} of the {s}^2 lattice: L={L}, depth {depth}, kernel widths "
                    f"{ker_width}//2^l, {edges} edges per sweep", train_step, inner + down + up)


WORKLOADS: Dict[str, Callable] = {"mgkn_orthogonal_burgers1d": orthogonal_burgers,
                                  "mgkn_general_darcy2d": general_darcy}
