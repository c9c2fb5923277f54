"""GPU tier: the PRODUCTION backward plan at the headline kernel MLP `[6, 1024, 1024, 4096]` against float64, directly
(VERDICT r4 weak 1a).

`loss.backward()` of the reference (/root/reference/graph-neural-operator/UAI1_full_resolution.py:258-273) differentiates
`depth` = 6 applications of ONE conv (:29-30) on the s=61, r=0.10 lattice (:39-46) with the kernel MLP of :21,57.  That is the
case here, at the reference's own training resolution: every gradient the native backward returns - grad_x of each
application, the three Linear layers, root, bias - against float64 autograd through the oracle
(`oracle.nnconv_grads_shared`, pinned on CPU to the sum of per-application `nnconv_grads`, themselves pinned to the
reference's own module), for
  * the single-call backward in its default plan (split-f16 dU_1 / dW_2 GEMMs, on-the-fly H_1, the per-edge kernel's
    by-products; one chunk) through the module's autograd - keep-Z forward that, since the second half of round 5, also KEEPS the
    last hidden activations for the backward (ops.keep_hidden: this graph has 380 k edges; tests/test_gpu_keep_hidden.py),
  * the same with a workspace that forces several node / edge chunks,
  * the light + depth-deferred pair (gpde_nnconv_bwd_light x 6, gpde_nnconv_bwd_deferred x 1).
The graph is the lattice minus the ~1 % of edges with a hidden pre-activation on the ReLU kink (tests/helpers/kinks.py): masks
then agree between the fp32-class forward and float64, and the tolerance is the plain 2e-5 on every gradient - no row-wise or
kink-level escape - with ONE stated exception: the hidden layers' gradients are sums over 3.8e5 edges of terms of both signs
(db_1 keeps ~2.5 % of the magnitude it sums: rounding noise is amplified ~40 x) computed by GEMMs whose split-f16 products carry
2^-21 each instead of fp32's 2^-24; they are held to max(2e-5, 10 x the error of the same plan with its GEMMs on exact fp32
MFMA), at most 5e-5 (`_compare`; measured: 2.6e-5 on db_1, 1.7e-5 on dW_1, against 3.4e-6 / 2.5e-6 for exact fp32)."""
import pytest
import torch

import graph_pde_amd as gp
from graph_pde_amd import _lib, hidden_cache, ops, synth
from oracle.nnconv_oracle import nnconv_grads_shared, rel_l2
from tests.helpers.kinks import edges_off_the_kink

pytestmark = pytest.mark.gpu
TOL = 2e-5
DIMS = [6, 1024, 1024, 4096]
DEPTH = 6


@pytest.fixture(scope="module")
def case():
    assert torch.cuda.is_available(), "GPU tier needs an MI355X"
    d = torch.device("cuda:0")
    ei, ea, n = synth.darcy_graph(61, 0.10)
    torch.manual_seed(61)
    mlp = torch.nn.Sequential(*sum([[torch.nn.Linear(DIMS[i], DIMS[i + 1]), torch.nn.ReLU()] for i in range(3)], [])[:-1])
    conv = gp.NNConv_old(64, 64, mlp, aggr="mean")
    lin = ops.mlp_linears(conv.nn)
    W, B = [l.weight.detach().clone() for l in lin], [l.bias.detach().clone() for l in lin]
    keep = edges_off_the_kink(ea, W, B)
    dropped = int((~keep).sum())
    assert 0 < dropped < ei.shape[1] // 20, dropped          # a thinning, not another graph
    ei, ea = ei[:, keep].contiguous(), ea[keep].contiguous()
    xs = [torch.randn(n, 64) * (0.5 + 0.3 * l) for l in range(DEPTH)]           # the applications' inputs and output gradients
    gs = [torch.randn(n, 64) * (2.0 ** -l) for l in range(DEPTH)]
    ref = nnconv_grads_shared(xs, ei, ea, W, B, conv.root.detach(), conv.bias.detach(), "mean", gs, chunk_edges=8192)
    print(f"s=61 lattice: {ei.shape[1]} edges ({dropped} on the ReLU kink removed), float64 oracle done")
    return {"d": d, "n": n, "ei": ei, "ea": ea, "conv": conv.to(d), "W": W, "B": B, "xs": xs, "gs": gs, "ref": ref}


def _errors(gxs, gW, gb, groot, gbias, ref):
    rxs, rW, rb, rroot, rbias = ref
    errs = {}
    for l, (g, r) in enumerate(zip(gxs, rxs)):
        errs[f"dx[{l}]"] = rel_l2(g.cpu(), r)
    for l in range(len(rW)):
        if gW[l] is not None:
            errs[f"dW{l + 1}"] = rel_l2(gW[l].cpu(), rW[l])
            errs[f"db{l + 1}"] = rel_l2(gb[l].cpu(), rb[l])
    if groot is not None:
        errs["droot"] = rel_l2(groot.cpu(), rroot)
    if gbias is not None:
        errs["dbias"] = rel_l2(gbias.cpu(), rbias)
    return errs


HIDDEN = ("dW1", "db1", "dW2", "db2")


def _compare(tag, gxs, gW, gb, groot, gbias, ref, e32=None):
    """Every gradient within TOL of float64.  `e32`: the errors of the SAME plan with its two k1 x k2 GEMMs on exact fp32 MFMA
    (GPDE_BWD_GEMM_F32).  The hidden layers' gradients are sums over 3.8e5 edges of terms of both signs (db_1 keeps ~2.5 % of
    the magnitude it sums: rounding noise is amplified ~40 x), and a split-f16 product carries 2^-21 where an fp32 product
    carries 2^-24: measured here, exact fp32 sits 2.5e-6 / 3.4e-6 from float64 on dW_1 / db_1 and the split GEMMs 1.7e-5 /
    2.6e-5 (7 x).  Those four gradients may be up to 10 x the exact-fp32 error from float64 - the 8 x of the arithmetic plus
    margin - and never more than 5e-5; everything else TOL."""
    errs = _errors(gxs, gW, gb, groot, gbias, ref)
    print(tag, {k: f"{v:.1e}" for k, v in errs.items()})
    bad = {}
    for k, v in errs.items():
        lim = TOL
        if e32 is not None and k in HIDDEN:
            lim = min(5e-5, max(TOL, 10 * e32[k]))
        if not v <= lim:
            bad[k] = (v, lim)
    assert not bad, (tag, bad)
    return errs


def test_single_call_backward_default_plan_vs_float64(case, monkeypatch):
    """The module's own autograd, one full backward per application (GPDE_HIDDEN_CACHE=off: no shared H), gradients summed by
    autograd over the six applications as `loss.backward()` does."""
    monkeypatch.setattr(hidden_cache, "MODE", "off")
    d, conv = case["d"], case["conv"]
    ei, ea = case["ei"].to(d), case["ea"].to(d)
    lin = ops.mlp_linears(conv.nn)

    def run():
        conv.zero_grad(set_to_none=True)
        xin = [x.to(d).requires_grad_(True) for x in case["xs"]]
        calls = _lib.n_native_calls
        loss = sum((conv(x, ei, ea) * g.to(d)).sum() for x, g in zip(xin, case["gs"]))
        loss.backward()
        torch.cuda.synchronize()
        assert _lib.n_native_calls - calls >= 2 * DEPTH
        return [x.grad for x in xin], [l.weight.grad.clone() for l in lin], [l.bias.grad.clone() for l in lin], conv.root.grad.clone(), conv.bias.grad.clone()
    monkeypatch.setenv("GPDE_BWD_GEMM_F32", "1")              # the same plan with dU_1 / dW_2 on exact fp32 MFMA: the yardstick
    e32 = _errors(*run(), case["ref"])
    print("exact-fp32 GEMMs", {k: f"{v:.1e}" for k, v in e32.items()})
    monkeypatch.delenv("GPDE_BWD_GEMM_F32")
    # the plan that runs now: >= 8192 rows per chunk, widths multiples of 128 -> split-f16 GEMMs; one chunk
    assert ei.shape[1] >= 8192 and ops.deferred_supported(DIMS)
    case["e32"] = e32
    _compare("module autograd, default plan", *run(), case["ref"], e32)


def test_single_call_backward_in_several_chunks_vs_float64(case):
    """A workspace of a third of the default: ~5 node / edge chunks of ~80 k edges (still above the 8192-row switch to the
    split-f16 GEMMs), split-K partials and the ordered dx reduction continuing across chunks."""
    d = case["d"]
    ei, ea = case["ei"].to(d), case["ea"].to(d)
    csr = ops.build_csr(ei, case["n"])
    W, B = [w.to(d) for w in case["W"]], [b.to(d) for b in case["B"]]
    root = case["conv"].root.detach()
    dims_c = _lib.dims_array(DIMS)
    full = int(_lib.lib().gpde_nnconv_bwd_workspace_bytes(case["n"], ei.shape[1], 3, dims_c))
    small = torch.empty(full // 3, dtype=torch.uint8, device=d)
    gxs, sW, sb, sroot, sbias = [], None, None, None, None
    for x, g in zip(case["xs"], case["gs"]):
        gx, gW, gb, groot, gbias = ops.nnconv_backward_raw(x.to(d), csr, ea, W, B, root, "mean", g.to(d), ws=small)
        gxs.append(gx)
        if sW is None:
            sW, sb, sroot, sbias = [w.double() for w in gW], [b.double() for b in gb], groot.double(), gbias.double()
        else:
            for k in range(3):
                sW[k] += gW[k].double(); sb[k] += gb[k].double()
            sroot += groot.double(); sbias += gbias.double()
    torch.cuda.synchronize()
    _compare("raw calls, workspace / 3", gxs, sW, sb, sroot, sbias, case["ref"], case.get("e32"))


def test_light_and_deferred_pair_vs_float64(case):
    """gpde_nnconv_bwd_light per application (grad_x, last Linear, root, bias) + ONE gpde_nnconv_bwd_deferred for the hidden
    layers of all six - what a depth-6 training step runs when H does not fit memory."""
    d = case["d"]
    ei, ea = case["ei"].to(d), case["ea"].to(d)
    csr = ops.build_csr(ei, case["n"])
    W, B = [w.to(d) for w in case["W"]], [b.to(d) for b in case["B"]]
    root = case["conv"].root.detach()
    xs, gs = [x.to(d) for x in case["xs"]], [g.to(d) for g in case["gs"]]
    gxs, w3, b3, sroot, sbias = [], None, None, None, None
    for x, g in zip(xs, gs):
        gx, gw, gb, groot, gbias = ops.nnconv_backward_light_raw(x, csr, ea, W, B, root, "mean", g)
        gxs.append(gx)
        if w3 is None:
            w3, b3, sroot, sbias = gw.double(), gb.double(), groot.double(), gbias.double()
        else:
            w3 += gw.double(); b3 += gb.double(); sroot += groot.double(); sbias += gbias.double()
    dW, db = ops.nnconv_backward_deferred_raw(xs, gs, csr, ea, W, B, "mean")
    torch.cuda.synchronize()
    _compare("light x 6 + deferred", gxs, list(dW) + [w3], list(db) + [b3], sroot, sbias, case["ref"], case.get("e32"))


def _module_step(case, d, ei, ea, xs_dev=None):
    """Six applications of the ONE module on the module's own autograd, loss = sum_l <conv(x_l), g_l>, backward: what
    `loss.backward()` differentiates at UAI1_full_resolution.py:258-273 (with the applications' inputs given)."""
    conv = case["conv"]
    lin = ops.mlp_linears(conv.nn)
    conv.zero_grad(set_to_none=True)
    xin = [x.to(d).requires_grad_(True) for x in case["xs"]] if xs_dev is None else [x.detach().requires_grad_(True) for x in xs_dev]
    loss = sum((conv(x, ei, ea) * g.to(d)).sum() for x, g in zip(xin, case["gs"]))
    loss.backward()
    return [x.grad for x in xin], [l.weight.grad for l in lin], [l.bias.grad for l in lin], conv.root.grad, conv.bias.grad


def test_shared_hidden_path_vs_float64(case, monkeypatch):
    """the DEFAULT training path of the reference's own configuration (UAI1_full_resolution.py:39-57: s=61,
    `[6,1024,1024,4096]`, depth 6) is the shared-H one - `HiddenFunction` builds H once, the six applications run
    `NNConvHiddenFunction` (gpde_nnconv_bwd with `hidden` given; from the second on their dL/dH is ADDED inside the per-edge kernel)
    and ONE gpde_hidden_bwd differentiates the hidden layers on the sum.  Held to float64 here at its own width, through the
    module's autograd; the counters prove the path."""
    monkeypatch.setattr(hidden_cache, "MODE", "on")
    hidden_cache.clear()
    d = case["d"]
    ei, ea = case["ei"].to(d), case["ea"].to(d)
    if "e32" not in case:                                      # (run alone: the yardstick of _compare from the same plan on exact fp32 GEMMs)
        monkeypatch.setenv("GPDE_BWD_GEMM_F32", "1")
        case["e32"] = _errors(*_module_step(case, d, ei, ea), case["ref"])
        monkeypatch.delenv("GPDE_BWD_GEMM_F32")
        hidden_cache.clear()
    acc0, builds0, hits0, kept0 = ops.n_grad_hidden_accumulated, hidden_cache.stats["builds"], hidden_cache.stats["hits"], ops.n_kept_hidden
    got = _module_step(case, d, ei, ea)
    torch.cuda.synchronize()
    assert hidden_cache.stats["builds"] - builds0 == 1 and hidden_cache.stats["hits"] - hits0 == DEPTH - 1      # one H, five hits
    assert ops.n_grad_hidden_accumulated - acc0 == DEPTH - 1                                                      # dL/dH summed in-kernel
    assert ops.n_kept_hidden == kept0                                                                             # not the kept-H direct operator
    _compare("module autograd, shared H (HiddenFunction + 6 x NNConvHiddenFunction)", *got, case["ref"], case["e32"])
    hidden_cache.clear()


def test_captured_training_pass_vs_float64(case, monkeypatch):
    """Same step recorded by `gp.capture` (forward + backward of the six applications as ONE HIP graph, default cache policy as it
    settles during the warm-up) and replayed on NEW inputs: the gradients a replay leaves are held to the float64 oracle, not to
    the direct step (last sentence)."""
    hidden_cache.clear()
    d, conv = case["d"], case["conv"]
    ei, ea = case["ei"].to(d), case["ea"].to(d)
    lin = ops.mlp_linears(conv.nn)
    gs = torch.stack([g.to(d) for g in case["gs"]])

    def step(X):
        conv.zero_grad(set_to_none=True)
        xin = [X[l].detach().requires_grad_(True) for l in range(DEPTH)]
        loss = sum((conv(x, ei, ea) * gs[l]).sum() for l, x in enumerate(xin))
        loss.backward()
        return [x.grad for x in xin], [l.weight.grad for l in lin], [l.bias.grad for l in lin], conv.root.grad, conv.bias.grad
    X = torch.stack([x.to(d) for x in case["xs"]])
    cap = gp.capture(step, torch.randn_like(X), copy_outputs=True)          # recorded on OTHER inputs
    calls = _lib.n_native_calls
    got = cap(X)
    torch.cuda.synchronize()
    assert _lib.n_native_calls == calls and cap.recordings == 1
    _compare("captured forward + backward, replayed on new inputs", *got, case["ref"], case.get("e32"))
    hidden_cache.clear()
