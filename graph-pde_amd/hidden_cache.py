"""Cross-depth reuse of the kernel-MLP hidden activations.

The reference applies one NNConv module `depth` times per forward with the same `edge_attr` and
the same weights (/root/reference/graph-neural-operator/UAI1_full_resolution.py:29-30; the MGKN
V-cycle, multipole-graph-neural-operator/MGKN_general_darcy2d.py:76-90), so
H = relu(L_{n-1}(... relu(L_1(edge_attr)))) -- 94 % of the operator's FLOPs at [6,1024,1024,4096] -- is
the same tensor in every one of those calls.  This module decides, per NNConv instance, when to
materialise H (one `HiddenFunction` node) and serve the remaining calls from it
(`NNConvHiddenFunction`): the hidden layer and its backward then run once per step.

Policy (`GPDE_HIDDEN_CACHE` = auto | on | off, default auto; budget `GPDE_HIDDEN_CACHE_GB`, default: sized to the device -
70 % of its HBM, at most what is free now minus a 48 GB reserve for workspaces, i.e. ~200 GB on an idle 288 GB MI355X):
  * a call whose (edge_attr memory + version, CSR, hidden-layer parameter versions, precision, grad
    mode) matches the cached H is a hit;
  * "auto" materialises H only for a module that has been SEEN repeating a key (the second call of
    the first forward, then the first call of every later forward); a module that stops repeating
    goes back to the direct fused path.  "on" always materialises, "off" never does;
  * H is E x K2P x 4 bytes; larger than the budget -> direct path, or, for inference
    (`GPDE_HIDDEN_CACHE_PARTIAL`, default on), a PARTIAL H: the in-edges of the first `hn` nodes that fit
    the budget are cached and served by the mixed forward (gpde_nnconv_fwd_mixed_keepz), the other nodes run
    the fused kernel.  G241 at k2 = 1024 needs 391 GB for the full H.
The entry holds `edge_attr` (so its memory cannot be recycled for other data while the key is
alive) and is invalidated when the backward of its H node has run.  Nothing is stored on the
module itself: modules pickle as before.
"""
from __future__ import annotations

import os
import weakref
from typing import List, Optional

import torch

from . import ops
from .autograd import DeferredHiddenFunction, DeferredToken, EdgeWeightsFunction, HiddenFunction, HiddenToken

MODE = os.environ.get("GPDE_HIDDEN_CACHE", "auto")
_env_gb = os.environ.get("GPDE_HIDDEN_CACHE_GB", "")
# None = sized to the device at the moment H is built (budget_bytes); a number pins it (tests / A-B runs set this variable)
BUDGET_BYTES: Optional[int] = None if _env_gb in ("", "auto") else int(float(_env_gb) * (1 << 30))
AUTO_FRACTION = 0.7                  # of the device's HBM.  Round 6 tried 0.8 on the 241^2 training step (30 GiB more of H instead of two kept Z:
                                     # 7.02 -> 6.79 s pinned, 6.95 s under this policy) - and the distinct-samples step ran out of memory
                                     # on its 26.9 GiB workspace with 15.7 GiB free: left at 0.7 (profiles/r06_g241_memory_budget.txt)
AUTO_RESERVE_BYTES = 48 << 30        # left free for workspaces (Z of the 241^2 graph: 15 GB), the caller's tensors, RCCL


def budget_bytes(device=None, releasing: int = 0) -> int:
    """Bytes the hidden activations of ONE module may take.  With GPDE_HIDDEN_CACHE_GB unset the budget follows the
    device: min(70 % of its memory, free now + `releasing` (the H about to be dropped) - reserve).  On the 241^2 graph
    (H = 391 GB) that caches the in-edges of ~44 % of the nodes instead of none, 1.64 x on a depth-6 inference forward
    (scripts/time_g241_depth6_reuse.py)."""
    if BUDGET_BYTES is not None:
        return BUDGET_BYTES
    dev = None if device is None else torch.device(device)
    if dev is None or dev.type != "cuda":
        return 32 << 30
    free, total = ops.device_free_bytes(dev)
    return max(0, min(int(AUTO_FRACTION * total), free + releasing - AUTO_RESERVE_BYTES))
PARTIAL = os.environ.get("GPDE_HIDDEN_CACHE_PARTIAL", "1") != "0"
# Depth-deferred backward: a module that repeats (edge_attr, weights) within a forward, needs gradients and
# whose H does NOT fit the budget shares one "virtual H" autograd node - its applications run the light backward and ONE
# deferred pass differentiates the hidden layers for all of them (autograd.DeferredHiddenFunction).  auto | off.
DEFER_MODE = os.environ.get("GPDE_DEFERRED_BWD", "auto")
# Per-edge weight cache: for calls on low in-degree / small graphs the whole
# W_e = view(nn(edge_attr_e), 64, 64) tensor is kept ([E, 4096] fp32 = 16 KiB per edge) and a call is one streaming
# kernel; with gradients W_e is an autograd node shared by the module's applications (autograd.EdgeWeightsFunction /
# WeConvFunction: the MGKN scripts TRAIN, MGKN_general_darcy2d.py:260-282).  Its summation order differs from the fused
# kernels' (same accuracy, other last bits) - like the hidden-activation cache it makes a module's output depend on its call
# history at the 1e-7 level; GPDE_HIDDEN_CACHE=off (which also disables this cache: W_e derives from H) gives
# history-independent bits.  Used by (a) nn_conv.nnconv_group - the explicit grouped API, from the first call;
# (b) aggr='max', which has no other path; (c) plain module calls of a module SEEN repeating (edge_attr, weights):
# GPDE_EDGE_WEIGHT_CACHE=auto, the default since round 4 (rounds 1-3: off); "off" keeps (a) and (b) only.
# Budget per module: GPDE_EDGE_WEIGHT_CACHE_GB (default 4).
WE_MODE = os.environ.get("GPDE_EDGE_WEIGHT_CACHE", "auto")
WE_BUDGET_BYTES = int(float(os.environ.get("GPDE_EDGE_WEIGHT_CACHE_GB", "4")) * (1 << 30))
WE_SMALL_EDGES = 8192          # calls up to this many edges qualify whatever their in-degree

stats = {"hits": 0, "builds": 0, "direct": 0, "we_hits": 0, "we_builds": 0}       # counters for tests / bench


class _Entry:
    __slots__ = ("key", "hidden", "token", "attr_ref", "csr", "last_key", "repeats", "hits_on_hidden", "hn", "we", "we_key",
                 "we_refs", "big_key", "dkey", "dtoken", "dvirtual", "dcount", "drefs", "twe", "twe_key", "twe_token", "twe_h", "last_partial")

    def __init__(self):
        self.key = None
        self.hidden = None
        self.token = None
        self.attr_ref = None        # strong reference: pins the memory behind `key`
        self.csr = None
        self.last_key = None
        self.repeats = False        # this module has been seen repeating a key
        self.hits_on_hidden = 0
        self.hn = 0                 # nodes whose in-edges the cached H covers (all of them unless partial)
        self.we = None              # per-edge weights [E, 4096] (inference) and the key they were built for
        self.we_key = None
        self.we_refs = None         # pins edge_attr / csr behind we_key
        self.big_key = None         # key of the last call whose H was wanted but exceeded the budget
        self.dkey = None            # depth-deferred backward: key, token and tensor of the current virtual H,
        self.dtoken = None          # applications hanging on it, pinned edge_attr / csr
        self.dvirtual = None
        self.dcount = 0
        self.drefs = None
        self.twe = None             # training: W_e as an autograd node shared by the applications of a step, its key,
        self.twe_key = None         # validity token (cleared by its backward) and the H tensor it was built from
        self.twe_token = None
        self.last_partial = None    # (n_edges, hn) of the last PARTIAL H built: the next build keeps that size while it still fits (see lookup)
        self.twe_h = None


_entries: "weakref.WeakKeyDictionary[torch.nn.Module, _Entry]" = weakref.WeakKeyDictionary()


def clear():
    _entries.clear()
    for k in stats:
        stats[k] = 0


def release_all() -> bool:
    """Drop every cached H / W_e tensor (the entries and what they learnt about their modules stay): called by
    ops._alloc_ws when a workspace does not fit.  Returns whether anything was released."""
    freed = False
    for ent in list(_entries.values()):
        if ent.hidden is not None or ent.we is not None:
            freed = True
        if ent.token is not None:
            ent.token.gh_acc, ent.token.gh_adds = None, 0
        ent.hidden, ent.key, ent.token, ent.attr_ref, ent.csr = None, None, None, None, None
        ent.we, ent.we_key, ent.we_refs = None, None, None
        ent.twe, ent.twe_key, ent.twe_token, ent.twe_h = None, None, None, None     # (nodes of a live graph keep their own references)
        if ent.dtoken is not None:
            ent.dtoken.hpart = None          # applications still hanging on the virtual H recompute instead (same mathematics)
    if freed:
        stats["released"] = stats.get("released", 0) + 1
    return freed


def snapshot() -> list:
    """Strong references to every tensor the entries hold right now (capture.py: a recorded graph reads H / W_e / the virtual
    node's partial H at the addresses they had at recording time; `clear()`, `release_all()` and rebuilt entries must not free
    them under it - ADVICE r5)."""
    pins = []
    for ent in list(_entries.values()):
        pins.append((ent.hidden, ent.we, ent.twe, ent.twe_h, ent.attr_ref, ent.csr, ent.we_refs, ent.drefs, ent.dvirtual,
                     None if ent.dtoken is None else ent.dtoken.hpart))
    return pins


def _key(edge_attr, csr, hidden_params: List[Optional[torch.Tensor]], precision: str):
    ops.watch(edge_attr.table if isinstance(edge_attr, ops.NodeAttr) else edge_attr, *hidden_params)
    if isinstance(edge_attr, ops.NodeAttr):      # attributes described by node data: keyed on the table's memory + version + slots
        tb = edge_attr.table
        grad = torch.is_grad_enabled() and any(p is not None and p.requires_grad for p in hidden_params)
        return (str(tb.device), tb.untyped_storage().data_ptr(), tb.storage_offset(), tuple(tb.shape), tuple(tb.stride()),
                ops._ver(tb), tuple(edge_attr.sel), id(csr),
                tuple((0, 0) if p is None else (p.data_ptr(), ops._ver(p)) for p in hidden_params), precision, grad)
    st = edge_attr.untyped_storage()
    grad = torch.is_grad_enabled() and any(p is not None and p.requires_grad for p in hidden_params)
    return (str(edge_attr.device), st.data_ptr(), edge_attr.storage_offset(), tuple(edge_attr.shape),
            tuple(edge_attr.stride()), ops._ver(edge_attr), id(csr),
            tuple((0, 0) if p is None else (p.data_ptr(), ops._ver(p)) for p in hidden_params),
            precision, grad)


def lookup(module: torch.nn.Module, edge_attr: torch.Tensor, csr, pm, weights, biases,
           precision: Optional[str] = None, mode: Optional[str] = None, allow_partial: bool = False):
    """Returns (H, hmax, hn) for this call (cached or freshly built; hmax = device scalar max |H| or None;
    hn = number of leading nodes whose in-edges H covers, = N unless partial), or None when the direct
    fused path should run.  `allow_partial`: the caller needs no gradient at all."""
    mode = MODE if mode is None else mode
    if mode == "off":
        stats["direct"] += 1
        return None
    precision = ops.DEFAULT_PRECISION if precision is None else precision
    hw, hb = list(weights[:-1]), list(biases[:-1])
    key = _key(edge_attr, csr, hw + hb, precision)
    ent = _entries.get(module)
    if ent is None:
        ent = _entries[module] = _Entry()
    if ent.hidden is not None and ent.key == key and ent.token.valid:
        ent.hits_on_hidden += 1
        # a backward pass that never reached the H node (input gradients only, an exception) left its running dL/dH sum on the
        # token: [E, K2P], 24 GB at s=121.  No pass is in flight at forward time - drop it
        ent.token.gh_acc, ent.token.gh_adds = None, 0
        if ent.hn == csr.n_nodes or allow_partial:
            stats["hits"] += 1
            if ent.hn < csr.n_nodes:
                ent.big_key = key
            return ent.hidden, ent.token.hmax, ent.hn
    repeated = ent.last_key == key
    if repeated:
        ent.repeats = True
    elif ent.hidden is not None and ent.key != key and ent.hits_on_hidden == 0:
        ent.repeats = False          # the last H was built and never reused: stop speculating
    ent.last_key = key
    row_bytes = ops.hidden_width(pm.dims) * 4
    nbytes = csr.n_edges * row_bytes
    want = mode == "on" or (mode == "auto" and ent.repeats)
    hn = csr.n_nodes
    budget = budget_bytes(edge_attr.device, 0 if ent.hidden is None else ent.hidden.numel() * ent.hidden.element_size()) if want else 0
    if want and nbytes > budget and PARTIAL and allow_partial and len(pm.dims) == 4:
        # the leading nodes whose in-edges fit the budget, in whole 64-node tiles; worth it from 1/8 on
        rp = csr.rowptr_host
        hn = int(torch.searchsorted(rp.to(torch.int64), torch.tensor(budget // row_bytes), right=True)) - 1
        hn = hn // 64 * 64
        # Keep the previous build's size while it still fits and is within 10 % of what the budget allows now: the budget follows
        # the free memory of the moment, and a partial H a few MiB LARGER than the last one cannot reuse the block the allocator
        # just got back - on the 241^2 graph a second 230 GiB request that ends in the allocator's out-of-memory retry (round 6,
        # with the budget's free-memory term binding instead of the fixed fraction: every other training step took 14.8 s instead of 6.7)
        lp = ent.last_partial
        if lp is not None and lp[0] == csr.n_edges and lp[1] <= hn and lp[1] >= 0.9 * hn:
            hn = lp[1]
        nbytes = int(rp[hn]) * row_bytes if hn >= csr.n_nodes // 8 and hn > 0 else budget + 1
    if not want or nbytes > budget or csr.n_edges == 0 or edge_attr.requires_grad:
        ent.hidden, ent.key, ent.token, ent.attr_ref, ent.csr = None, None, None, None, None
        ent.big_key = key if (want and nbytes > budget and csr.n_edges > 0 and not edge_attr.requires_grad) else None
        stats["direct"] += 1
        return None
    token = HiddenToken()
    ent.hidden = None                # release the previous H before allocating the next one
    if ent.dtoken is not None:
        ent.dtoken.hpart = None      # ... also where a finished virtual-H node still points at it
    try:
        if hn < csr.n_nodes:         # partial: no autograd node (inference, or training on the virtual-H node: lookup_deferred)
            hidden, token.hmax = ops.hidden_forward_raw(csr, edge_attr.detach(), pm, list(weights[:-1]) + [None],
                                                        list(biases[:-1]) + [None], precision, n_nodes_limit=hn)
        else:
            hidden = HiddenFunction.apply(edge_attr, csr, pm, precision, token, len(hw), *hw, *hb)
    except torch.OutOfMemoryError:
        # the budget was measured a moment ago; another module's cache, another process or the caller's own tensors may have
        # taken the room since.  H is an optimisation: drop everything cached and run the direct path.
        ent.hidden, ent.key, ent.token, ent.attr_ref, ent.csr = None, None, None, None, None
        release_all()
        torch.cuda.empty_cache()
        stats["oom_fallbacks"] = stats.get("oom_fallbacks", 0) + 1
        stats["direct"] += 1
        return None
    ent.key, ent.hidden, ent.token, ent.attr_ref, ent.csr, ent.hn = key, hidden, token, edge_attr, csr, hn
    ent.hits_on_hidden = 0
    if hn < csr.n_nodes:
        ent.big_key = key            # the whole H does not fit: a training call shares a virtual-H node (lookup_deferred)
        ent.last_partial = (csr.n_edges, hn)
    stats["builds"] += 1
    return hidden, token.hmax, hn


def defer_possible(edge_attr: torch.Tensor, pm, weights, biases, aggr: str) -> bool:
    """A call that needs gradients could hang on a virtual-H node (lookup_deferred): then `lookup` may hand it a PARTIAL H."""
    return DEFER_MODE != "off" and aggr in ("add", "mean") and torch.is_grad_enabled() and not edge_attr.requires_grad and \
        any(p is not None and p.requires_grad for p in list(weights[:-1]) + list(biases[:-1])) and ops.deferred_supported(pm.dims)


def token_of(module: torch.nn.Module, hidden: torch.Tensor, csr=None):
    """The HiddenToken of `hidden` if that is the module's cached full H (the applications sharing it sum their dL/dH on it:
    autograd.NNConvHiddenFunction.backward), else None.  None too for a graph on which the per-edge weight form may run
    (edge_weights_qualify): there W_e's backward is a second kind of consumer of H, and an application can fall from one form to the
    other when memory is short - the in-place sum wants NNConvHiddenFunction to be H's only consumer."""
    ent = _entries.get(module)
    if ent is None or ent.hidden is not hidden or ent.token is None or not ent.token.valid:
        return None
    if csr is not None and WE_MODE != "off" and edge_weights_qualify(csr, explicit=True):
        return None
    return ent.token


def lookup_deferred(module: torch.nn.Module, edge_attr: torch.Tensor, csr, pm, weights, biases, aggr: str,
                    precision: Optional[str] = None, hpart=None):
    """(virtual H tensor, token) for a call that needs gradients, right after `lookup` returned None (or a PARTIAL H: `hpart`
    = its (H, hmax, hn) - the applications then read those rows instead of recomputing them) for it - or None when
    the plain operator (autograd.NNConvFunction: its own full backward) should run.  Deferred when: the module has been seen
    repeating this (edge_attr, weights) key, its H was wanted but does not fit the budget, the hidden layers require a
    gradient and the kernel MLP is in the deferred form (ops.deferred_supported).  All applications of one forward share the
    returned node until its backward has run."""
    if DEFER_MODE == "off" or aggr not in ("add", "mean") or not torch.is_grad_enabled():
        return None
    ent = _entries.get(module)
    if ent is None:
        return None
    precision = ops.DEFAULT_PRECISION if precision is None else precision
    hw, hb = list(weights[:-1]), list(biases[:-1])
    if not any(p is not None and p.requires_grad for p in hw + hb) or edge_attr.requires_grad:
        return None
    key = _key(edge_attr, csr, hw + hb, precision)
    if ent.dtoken is not None and ent.dkey == key and ent.dtoken.valid:
        ent.dcount += 1
        ent.dtoken.hpart = hpart
        if ent.dtoken.drop_stale():      # a backward over this (still valid) node ended without its deferred pass
            stats["deferred_stale_dropped"] = stats.get("deferred_stale_dropped", 0) + 1
        stats["deferred_hits"] = stats.get("deferred_hits", 0) + 1
        return ent.dvirtual, ent.dtoken
    if ent.dtoken is not None and ent.dcount <= 1:
        ent.repeats = False             # the last virtual H served a single application: stop speculating
    if ent.dtoken is not None:
        ent.dtoken.hpart = None
    ent.dkey, ent.dtoken, ent.dvirtual, ent.drefs, ent.dcount = None, None, None, None, 0
    if not ent.repeats or ent.big_key != key or not ops.deferred_supported(pm.dims):
        return None
    token = DeferredToken()
    token.hpart = hpart
    n = len(weights)
    w_last, b_last = weights[-1], biases[-1]
    virtual = DeferredHiddenFunction.apply(edge_attr, csr, aggr, token, n, *hw, w_last.detach(),
                                           *hb, None if b_last is None else b_last.detach())
    ent.dkey, ent.dtoken, ent.dvirtual, ent.drefs, ent.dcount = key, token, virtual, (edge_attr, csr), 1
    stats["deferred_builds"] = stats.get("deferred_builds", 0) + 1
    return virtual, token


def edge_weights_qualify(csr, force: bool = False, explicit: bool = False) -> bool:
    """The per-edge weight form pays where the re-association of DESIGN.md §2 does not: mean in-degree <= 4, or a call
    of at most WE_SMALL_EDGES edges (launch-bound anyway).  `explicit`: the caller opted in (nnconv_group) - WE_MODE is
    not consulted; `force` (aggr='max': no other path) only needs the budget."""
    e, n = csr.n_edges, csr.n_nodes
    if e == 0:
        return False
    if force:
        return e * ops.EDGE_WEIGHT_BYTES <= max(WE_BUDGET_BYTES, budget_bytes(getattr(getattr(csr, "rowptr", None), "device", None)))
    # (plain module calls: only while the hidden-activation cache it derives from is on - GPDE_HIDDEN_CACHE=off is the ONE switch
    # for history-independent bits)
    return (explicit or (WE_MODE == "auto" and MODE != "off")) and (e <= 4 * n or e <= WE_SMALL_EDGES) and \
        e * ops.EDGE_WEIGHT_BYTES <= WE_BUDGET_BYTES


def we_token_of(module: torch.nn.Module, we: torch.Tensor):
    """The token of the module's shared W_e node when `we` is that node (WeConvFunction sums the applications' gradients in place
    on it), else None."""
    ent = _entries.get(module)
    return ent.twe_token if ent is not None and ent.twe is we else None


def lookup_edge_weights_train(module: torch.nn.Module, hidden: torch.Tensor, csr, pm, weights, biases):
    """W_e [E, 4096] as an AUTOGRAD node for a call that needs gradients and was just handed the module's full cached H
    (`hidden`: the HiddenFunction output `lookup` returned) - or None when the graph does not qualify / the policy is off.
    All applications of the step share the node (autograd sums their dL/dW_e; EdgeWeightsFunction.backward runs the
    4096 x k2 products once) until its backward has run or H was rebuilt."""
    if WE_MODE != "auto" or not edge_weights_qualify(csr) or not torch.is_grad_enabled():
        return None
    ent = _entries.get(module)
    if ent is None or ent.hidden is not hidden:
        return None
    w_last, b_last = weights[-1], biases[-1]
    key = ((w_last.data_ptr(), ops._ver(w_last)), (0, 0) if b_last is None else (b_last.data_ptr(), ops._ver(b_last)))
    if ent.twe is not None and ent.twe_h is hidden and ent.twe_key == key and ent.twe_token.valid:
        stats["we_hits"] += 1
        # (a backward pass that never reached the W_e node left its running sums on the token: no pass is in flight at forward time)
        ent.twe_token.gh_acc, ent.twe_token.gh_adds, ent.twe_token.side_acc = None, 0, None
        return ent.twe
    ent.twe, ent.twe_key, ent.twe_token, ent.twe_h = None, None, None, None
    # memory of the training form: W_e, one dL/dW_e per application in flight and autograd's running sum of them - 16 KiB per
    # edge each (the budget bounded W_e only).  When three of them do not fit what is free, the H path runs.
    free, _ = ops.device_free_bytes(hidden.device)
    if 3 * csr.n_edges * 4096 * 4 > free:
        stats["we_train_no_room"] = stats.get("we_train_no_room", 0) + 1
        return None
    token = HiddenToken()
    try:
        we = EdgeWeightsFunction.apply(hidden, pm, w_last, b_last, token)
    except torch.OutOfMemoryError:
        torch.cuda.empty_cache()
        stats["oom_fallbacks"] = stats.get("oom_fallbacks", 0) + 1
        return None
    ent.twe, ent.twe_key, ent.twe_token, ent.twe_h = we, key, token, hidden
    stats["we_builds"] += 1
    return we


def lookup_edge_weights(module: torch.nn.Module, edge_attr: torch.Tensor, csr, pm, weights, biases,
                        precision: Optional[str] = None, force: bool = False, explicit: bool = False):
    """W_e [E, 4096] for this call, or None when another path should run.  ONLY for calls that need no gradient at
    all.  `force` / `explicit`: built at once (H too).  Otherwise (GPDE_EDGE_WEIGHT_CACHE=auto) it is derived from the
    module's cached hidden activations the first time they are served without gradient - i.e. only for a module that
    has been seen repeating (edge_attr, weights)."""
    if not edge_weights_qualify(csr, force, explicit) or edge_attr.requires_grad:
        return None
    precision = ops.DEFAULT_PRECISION if precision is None else precision
    ent = _entries.get(module)
    w_last, b_last = weights[-1], biases[-1]
    key = (_key(edge_attr, csr, list(weights[:-1]) + list(biases[:-1]), precision)[:-1],
           (w_last.data_ptr(), ops._ver(w_last)), (0, 0) if b_last is None else (b_last.data_ptr(), ops._ver(b_last)))
    if ent is not None and ent.we is not None and ent.we_key == key:
        stats["we_hits"] += 1
        return ent.we
    now = force or explicit
    if not now:
        # peek only: the H policy (who repeats, who gets built) is driven by ONE lookup() per call, made by the caller's
        # ordinary path.  W_e is derived from an H that is already there - i.e. from the call after H was built.
        hkey = _key(edge_attr, csr, list(weights[:-1]) + list(biases[:-1]), precision)
        if ent is None or ent.hidden is None or ent.key != hkey or not ent.token.valid or ent.hn != csr.n_nodes:
            return None
    hit = lookup(module, edge_attr, csr, pm, weights, biases, precision, mode="on" if now else None, allow_partial=False)
    ent = _entries.get(module)
    if hit is None or hit[2] != csr.n_nodes:
        if ent is not None:
            ent.we, ent.we_key, ent.we_refs = None, None, None
        return None
    ent.we = None                       # release the previous tensor before allocating the next one
    try:
        ent.we = ops.edge_weights_raw(hit[0].detach(), pm, w_last, b_last)
    except torch.OutOfMemoryError:      # 16 KiB per edge: an optimisation like H - on OOM the ordinary path runs
        ent.we, ent.we_key, ent.we_refs = None, None, None
        torch.cuda.empty_cache()
        stats["oom_fallbacks"] = stats.get("oom_fallbacks", 0) + 1
        return None
    ent.we_key, ent.we_refs = key, (edge_attr, csr)
    stats["we_builds"] += 1
    return ent.we
