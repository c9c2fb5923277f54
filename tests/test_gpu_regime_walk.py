"""GPU tier: ONE model walked through every execution regime `ops.py` / `hidden_cache.py` can select.

What a user gets from `conv(x, edge_index, edge_attr)` is a SELECTION - by graph size, call history and free memory - among the
direct operator (recompute backward), the direct operator that keeps H_2 for its backward, the shared-H node, the partial H + light
+ depth-deferred backward, the per-edge-weight node and a captured HIP graph of whichever of those settled.  Every path has its own
parity tests; this one pins that the ANSWER does not depend on the selection: a `KernelNN`-shaped model
(/root/reference/graph-neural-operator/UAI1_full_resolution.py:14-33: fc1, depth x relu(conv1) with ONE conv, fc2) on one graph is
trained three Adam steps (:242, weight_decay 5e-4) from identical initial weights under each regime forced in turn.  Compared:
  * the FIRST step's raw gradients (before Adam turns magnitudes into signs): pairwise <= 1e-5 and <= 2e-5 to the float64 composite
    of the reference's op chain (tests/helpers/composite_nnconv.py);
  * the three losses: pairwise <= 1e-5 relative, <= 1e-4 to float64;
  * the final weights: pairwise <= 1e-5, <= 1e-4 to float64 (relative L2 per tensor).
Counters prove each regime really ran."""
import copy

import pytest
import torch
import torch.nn.functional as F

import graph_pde_amd as gp
from graph_pde_amd import _lib, hidden_cache, ops, synth
from oracle.nnconv_oracle import rel_l2
from tests.helpers import composite_nnconv

pytestmark = pytest.mark.gpu
DIMS = [6, 256, 256, 4096]      # (k1 >= 225: the store kernel of ops.keep_hidden needs 8 k1 chunks)
DEPTH = 4
STEPS = 3


class KernelNN(torch.nn.Module):                       # the shape of UAI1_full_resolution.py:14-33
    def __init__(self):
        super().__init__()
        self.fc1 = torch.nn.Linear(6, 64)
        mlp = torch.nn.Sequential(*sum([[torch.nn.Linear(DIMS[i], DIMS[i + 1]), torch.nn.ReLU()] for i in range(3)], [])[:-1])
        self.conv1 = gp.NNConv_old(64, 64, mlp, aggr="mean")
        self.fc2 = torch.nn.Linear(64, 1)

    def forward(self, a, ei, ea):
        x = self.fc1(a)
        for _ in range(DEPTH):
            x = F.relu(self.conv1(x, ei, ea))
        return self.fc2(x)


@pytest.fixture(scope="module")
def setup():
    assert torch.cuda.is_available(), "GPU tier needs an MI355X"
    d = torch.device("cuda:0")
    ei, ea, n = synth.darcy_graph(41, 0.10, device=d)          # 75 k edges, mean in-degree 45
    torch.manual_seed(5)
    model = KernelNN().to(d)
    a, y = torch.randn(n, 6, device=d), torch.randn(n, device=d) * 0.3
    return {"d": d, "ei": ei, "ea": ea, "n": n, "state": copy.deepcopy(model.state_dict()), "a": a, "y": y}


def _train(setup, dtype=torch.float32, composite=False, captured=False):
    """STEPS optimisation steps from the common initial weights; returns (first-step gradients, losses, final weights)."""
    d = setup["d"]
    model = KernelNN().to(d)
    model.load_state_dict(setup["state"])
    model = model.to(dtype)
    if composite:
        model.conv1.forward = lambda x, ei, ea: composite_nnconv.composite_forward(model.conv1, x, ei, ea)
    a, y, ea = setup["a"].to(dtype), setup["y"].to(dtype), setup["ea"].to(dtype)
    ei = setup["ei"]
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=5e-4, capturable=True)
    names = [k for k, _ in model.named_parameters()]

    def grads_only():
        opt.zero_grad(set_to_none=True)
        loss = F.mse_loss(model(a, ei, ea).view(-1, 1), y.view(-1, 1))
        loss.backward()
        return loss

    g0, trace = None, []

    def step():
        loss = grads_only()
        opt.step()
        trace.append(loss.detach().clone())
        return loss.detach()
    losses = []
    for k in range(1 if captured else STEPS):
        loss = grads_only()
        if k == 0:
            g0 = {k_: p.grad.clone() for k_, p in model.named_parameters()}
        opt.step()
        losses.append(float(loss.detach()))
        del loss          # (nothing of a finished step's autograd graph may outlive it: see the note in capture.py)
    if captured:
        # step 1 ran directly (its raw gradients are the comparison); the recording's ONE warm-up call is step 2 (the recording
        # itself executes nothing), every replay one more step
        cap = gp.capture(step, warmup=1, updates_parameters=True)
        torch.cuda.synchronize()
        losses.append(float(trace[0]))
        calls = _lib.n_native_calls
        while len(losses) < STEPS:
            losses.append(float(cap()))
        torch.cuda.synchronize()
        assert _lib.n_native_calls == calls and cap.replays == STEPS - 2
    torch.cuda.synchronize()
    assert names == list(g0)
    ent = hidden_cache._entries.get(model.conv1)
    setup["last_hn"] = None if ent is None or ent.hidden is None else ent.hn
    return g0, losses, {k: p.detach().clone() for k, p in model.named_parameters()}


def test_the_answer_does_not_depend_on_the_regime(setup, monkeypatch):
    results, e, k2p = {}, int(setup["ei"].shape[1]), ops.hidden_width(DIMS)

    def fresh():
        hidden_cache.clear()
        ops.clear_caches()
        return dict(hidden_cache.stats), ops.n_kept_hidden, ops.n_grad_hidden_accumulated

    # 1. direct operator, backward recomputes the hidden chain
    with monkeypatch.context() as m:
        m.setattr(hidden_cache, "MODE", "off"); m.setattr(ops, "SAVE_H_BYTES", 0)
        st0, kept0, _ = fresh()
        results["direct (recompute)"] = _train(setup)
        assert ops.n_kept_hidden == kept0 and hidden_cache.stats["builds"] == 0
    # 2. direct operator keeping H_2 for its own backward
    with monkeypatch.context() as m:
        m.setattr(hidden_cache, "MODE", "off"); m.setattr(ops, "SAVE_H_MIN_EDGES", 0)
        st0, kept0, _ = fresh()
        results["direct (H_2 kept)"] = _train(setup)
        assert ops.n_kept_hidden - kept0 == STEPS * DEPTH
    # 3. shared H: one HiddenFunction node per step, DEPTH applications on it
    with monkeypatch.context() as m:
        m.setattr(hidden_cache, "MODE", "on"); m.setattr(hidden_cache, "WE_MODE", "off")
        fresh()
        results["shared H"] = _train(setup)
        assert hidden_cache.stats["builds"] == STEPS and hidden_cache.stats["hits"] == STEPS * (DEPTH - 1)
    # 4. H over budget: partial H + light passes + ONE deferred pass per step (budget pinned to half of H)
    with monkeypatch.context() as m:
        m.setattr(hidden_cache, "MODE", "auto"); m.setattr(hidden_cache, "DEFER_MODE", "auto"); m.setattr(hidden_cache, "WE_MODE", "off")
        m.setattr(hidden_cache, "BUDGET_BYTES", e * k2p * 4 // 2)
        fresh()
        results["partial H + deferred"] = _train(setup)
        assert hidden_cache.stats.get("deferred_builds", 0) >= STEPS - 1 and hidden_cache.stats.get("deferred_hits", 0) >= (STEPS - 1) * (DEPTH - 1), hidden_cache.stats
        assert setup["last_hn"] is not None and 0 < setup["last_hn"] < setup["n"], "a partial H was in use"
    # 5. per-edge weights as the shared autograd node (the MGKN training form), forced onto this graph
    with monkeypatch.context() as m:
        m.setattr(hidden_cache, "MODE", "on"); m.setattr(hidden_cache, "WE_MODE", "auto")
        m.setattr(hidden_cache, "WE_SMALL_EDGES", 1 << 20); m.setattr(hidden_cache, "WE_BUDGET_BYTES", 8 << 30)
        fresh()
        results["per-edge weights"] = _train(setup)
        assert hidden_cache.stats["we_builds"] >= STEPS, hidden_cache.stats
    # 6. the default policy recorded as ONE HIP graph per step
    fresh()
    results["captured (default policy)"] = _train(setup, captured=True)
    # 7. the reference's op chain in float64 (stock torch ops, torch autograd)
    fresh()
    ref = _train(setup, dtype=torch.float64, composite=True)

    names = list(results)
    worst = {}
    for i, p in enumerate(names):
        g, l, w = results[p]
        eg = max(rel_l2(g[k], ref[0][k]) for k in g)
        el = max(abs(a - b) / abs(b) for a, b in zip(l, ref[1]))
        ew = max(rel_l2(w[k], ref[2][k]) for k in w)
        print(f"{p:28s} vs float64: first-step gradients {eg:.1e}, losses {el:.1e}, final weights {ew:.1e}; losses {l}")
        assert eg <= 2e-5 and el <= 1e-4 and ew <= 1e-4, (p, eg, el, ew)
        for q in names[i + 1:]:
            g2, l2, w2 = results[q]
            pg = max(rel_l2(g[k], g2[k]) for k in g)
            pl = max(abs(a - b) / abs(b) for a, b in zip(l, l2))
            pw = max(rel_l2(w[k], w2[k]) for k in w)
            worst[(p, q)] = (pg, pl, pw)
            assert pg <= 1e-5 and pl <= 1e-5 and pw <= 1e-5, (p, q, pg, pl, pw)
    print("worst pair:", max(worst.items(), key=lambda kv: max(kv[1])))
