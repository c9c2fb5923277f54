"""GPU tier: GRADIENT evidence at the headline scale (BASELINE.json configs[1] / [4]: Darcy 241^2, r = 0.10, N = 58,081,
E = 95,539,625, kernel MLP [6,1024,1024,4096]) - VERDICT r5 weak 1b: until now `grads_finite` in bench.py.

`loss.backward()` (/root/reference/graph-neural-operator/UAI1_full_resolution.py:266) through three applications of ONE conv (:29-30)
under the module's DEFAULT policy.  On this graph H is 391 GB: the applications share a virtual-H node, read the part of H that fits
(~200 GiB: partial H), run gpde_nnconv_bwd_light each and ONE gpde_nnconv_bwd_deferred - the composition that was
only ever checked against float64 at s <= 61, never where the int32 offsets, the hundreds of launches per step and 200 GiB of partial
H live.  Checked here:
  * grad_x rows of >= 528 stratified SOURCE nodes (176 + corners / edge midpoints / centre per application) against the float64 oracle evaluated
    on exactly those nodes' out-edges (~0.9 M edges; the operator is linear in x, so the rows need nothing else:
    oracle.nnconv_grad_x_rows, pinned on CPU to autograd through the reference's module) - <= 2e-5;
  * dW_1..3, db_1..3, droot, dbias against the same step with GPDE_HIDDEN_CACHE=off - every application's own full backward
    (gpde_nnconv_bwd, recompute form at this size), the plan tests/test_gpu_headline_bwd.py holds to float64 - <= 2e-5; the two
    hidden layers' gradients <= 2e-4.  Why that bound: profiles/r06_g241_grad_truth.txt (scripts/g241_grad_truth.py, ~3 PFLOP of
    float64 on the device, a one-off) holds THIS step against float64 autograd through the reference's op chain on the full graph:
    grad_x 1.5e-7 on all 58,081 rows, dW_3 / db_3 / root / bias <= 8e-7, and dW_1, db_1, dW_2, db_2 4.0e-4 - 5.3e-4 for BOTH plans
    AND for the same plan with its k1 x k2 GEMMs on exact fp32 MFMA (4.0e-4) AND for the reference's own op chain run in float32
    with stock torch ops on the device (3.7e-4 - 3.8e-4): at 95.5 M edges x 1024 units the hidden layers' gradients carry an
    fp32-class floor of ~4e-4 that is the reference arithmetic's own distance to float64, not the split products' (ReLU masks of
    pre-activations inside fp32 rounding of zero differ from float64's; sums of both signs over 3 x 95.5 M edges) - the two plans
    differ from each other by 8e-5 / 1.2e-4 (dW_1 / db_1) and 1.9e-5 / 1.7e-5 (dW_2 / db_2), a fifth to a third of that floor;
  * grad_x of the two plans agree everywhere (all 58,081 rows, all applications) to 2e-5."""
import pytest
import torch

import graph_pde_amd as gp
from graph_pde_amd import hidden_cache, ops, synth
from oracle.nnconv_oracle import nnconv_grad_x_rows, rel_l2

pytestmark = pytest.mark.gpu
DIMS = [6, 1024, 1024, 4096]
TOL = 2e-5
TOL_HIDDEN = 2e-4
APPS = 3            # applications per step: with 2 the policy (rightly) stops sharing - a virtual H that served ONE application


def _stratified(s, count, offset):
    n = s * s
    special = [0, s - 1, n - s, n - 1, s // 2, (s // 2) * s, (s // 2) * s + s - 1, (s // 2) * s + s // 2, s + 1, n - 2 * s + 1]
    rows = (torch.linspace(0, n - 1, count).round().long() + offset).clamp(max=n - 1)
    return torch.cat([rows, torch.tensor(special)]).unique()


def test_g241_training_gradients_default_policy():
    d = torch.device("cuda:0")
    s = 241
    hidden_cache.clear()
    ops.clear_caches()
    torch.cuda.empty_cache()
    ei, ea, n = synth.darcy_graph(s, 0.10, device=d, seed=0)
    e = int(ei.shape[1])
    assert (n, e) == (58081, 95539625)
    torch.manual_seed(241)
    mlp = torch.nn.Sequential(*sum([[torch.nn.Linear(DIMS[i], DIMS[i + 1]), torch.nn.ReLU()] for i in range(3)], [])[:-1])
    conv = gp.NNConv_old(64, 64, mlp, aggr="mean").to(d)
    lin = ops.mlp_linears(conv.nn)
    gen = torch.Generator(device=d).manual_seed(7)
    xs = [torch.randn(n, 64, device=d, generator=gen) * (0.5 + 0.4 * l) for l in range(APPS)]
    gs = [torch.randn(n, 64, device=d, generator=gen) * (2.0 ** -l) for l in range(APPS)]

    def step():
        conv.zero_grad(set_to_none=True)
        xin = [x.clone().requires_grad_(True) for x in xs]
        loss = sum((conv(x, ei, ea) * g).sum() for x, g in zip(xin, gs))
        loss.backward()
        torch.cuda.synchronize()
        got = {"dx": [x.grad.clone() for x in xin], "droot": conv.root.grad.clone(), "dbias": conv.bias.grad.clone()}
        for k, l in enumerate(lin):
            got[f"dW{k + 1}"], got[f"db{k + 1}"] = l.weight.grad.clone(), l.bias.grad.clone()
        conv.zero_grad(set_to_none=True)
        return got

    mode0, defer0 = hidden_cache.MODE, hidden_cache.DEFER_MODE
    try:
        # ---- the default policy.  Step 1 meets the module as a stranger (first application: its own full backward); from
        # step 2 on all applications hang on the shared virtual-H node and read the partial H
        hidden_cache.MODE, hidden_cache.DEFER_MODE = "auto", "auto"
        step()
        st0 = dict(hidden_cache.stats)
        dflt = step()
        st1 = dict(hidden_cache.stats)
        ent = hidden_cache._entries.get(conv)
        hn = 0 if ent is None or ent.hidden is None else ent.hn
        print("G241 default-policy step: cache stats", {k: st1.get(k, 0) - st0.get(k, 0) for k in st1}, "nodes on partial H", hn, "of", n,
              "peak GiB", round(torch.cuda.max_memory_allocated(d) / 2 ** 30, 1))
        assert st1.get("deferred_hits", 0) + st1.get("deferred_builds", 0) - st0.get("deferred_hits", 0) - st0.get("deferred_builds", 0) == APPS, \
            "all applications of the second step run on the virtual-H node"
        assert 0 < hn < n, "a PARTIAL H serves the leading nodes at this size"
        # ---- the same step, every application differentiating itself (no shared node, no partial H)
        hidden_cache.MODE = "off"
        hidden_cache.clear()
        torch.cuda.empty_cache()
        own = step()
    finally:
        hidden_cache.MODE, hidden_cache.DEFER_MODE = mode0, defer0
        hidden_cache.clear()
    torch.cuda.empty_cache()
    assert all(bool(torch.isfinite(v).all()) for k, v in dflt.items() if k != "dx") and all(bool(torch.isfinite(g).all()) for g in dflt["dx"])
    errs = {k: rel_l2(dflt[k], own[k]) for k in dflt if k != "dx"}
    errs.update({f"dx[{l}]": rel_l2(dflt["dx"][l], own["dx"][l]) for l in range(APPS)})
    print("G241: default policy (partial H + light + deferred) vs per-application full backward:", {k: f"{v:.1e}" for k, v in errs.items()})
    bad = {k: v for k, v in errs.items() if not v <= (TOL_HIDDEN if k in ("dW1", "db1", "dW2", "db2") else TOL)}
    assert not bad, bad

    # ---- grad_x rows of stratified SOURCE nodes against float64 on exactly their out-edges
    src = ei[0]
    deg = torch.bincount(ei[1], minlength=n).cpu()
    Wc, Bc = [l.weight.detach().cpu() for l in lin], [l.bias.detach().cpu() for l in lin]
    root_c = conv.root.detach().cpu()
    torch.set_num_threads(min(64, torch.get_num_threads() if torch.get_num_threads() > 8 else 64))
    total = 0
    for l in range(APPS):
        rows = _stratified(s, 176, offset=17 * l)
        eid = torch.nonzero(torch.isin(src, rows.to(d))).squeeze(1)
        total += int(eid.numel())
        ref = nnconv_grad_x_rows(rows, ei[:, eid].cpu(), ea[eid].cpu(), Wc, Bc, root_c, gs[l].cpu(), deg, chunk_edges=32768)
        e_d, e_o = rel_l2(dflt["dx"][l][rows.to(d)].cpu(), ref), rel_l2(own["dx"][l][rows.to(d)].cpu(), ref)
        print(f"G241 grad_x rows, application {l}: {rows.numel()} source nodes, {eid.numel()} out-edges; vs float64: default policy {e_d:.2e}, "
              f"per-application backward {e_o:.2e}")
        assert e_d <= TOL and e_o <= TOL, (l, e_d, e_o)
    assert total >= 800_000
