"""CPU oracle for the edge-conditioned graph convolution (NNConv) forward.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (`graph-pde_amd/`) may import this
module; only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` do,
and there only as the checker / the timed CPU baseline.

This is a plain-PyTorch *restatement* (not a copy) of the reference operator's arithmetic, in
the reference's own op order, so its fp32 round-off behaviour is that of the reference CPU path:

* kernel MLP  `DenseNet.forward`          -> /root/reference/graph-neural-operator/utilities.py:223-227
  (a chain of `torch.nn.Linear` (weight `[out,in]`) with ReLU between, no final non-linearity,
  built at utilities.py:201-221)
* `NNConv_old.forward`                    -> /root/reference/graph-neural-operator/nn_conv.py:267-271
  (1-D `x` / `edge_attr` are promoted to 2-D, then `propagate(edge_index, x=x, pseudo=edge_attr)`)
* `NNConv_old.message`                    -> nn_conv.py:273-275
  (`weight = nn(pseudo).view(-1, in, out)`; `m = matmul(x_j.unsqueeze(1), weight).squeeze(1)`)
* `NNConv_old.update`                     -> nn_conv.py:277-282  (`+ mm(x, root)`, `+ bias`)
* `MessagePassing.propagate` is third-party (torch_geometric, unpinned, NOT vendored in the
  reference and not installable here).  Its published semantics for flow='source_to_target'
  (SURVEY.md Appendix B): `x_j = x.index_select(0, edge_index[0])`, aggregation over
  `edge_index[1]` with `dim_size = N`; 'add' = index_add, 'mean' = sum / clamp(count, min=1)
  (zero in-degree -> 0), 'max' = segment max with empty -> 0.

Parity pinning: the reference ships no tests / golden vectors for this path.
This oracle is pinned against the reference's *own* `nn_conv.NNConv_old` + `utilities.DenseNet`
classes executed in the build container under import stubs (see `tests/golden/make_golden.py`,
which commits the resulting vectors under `tests/golden/`); the only restated third-party piece
is `propagate`, per the semantics quoted above.

The `[E, in*out]` per-edge weight tensor is 16 KiB/edge at width 64, so edges are processed in
chunks; the chunking does not change any per-element arithmetic except the order of the
destination scatter-add, which is kept in ascending edge order exactly like a sequential
`index_add_` on CPU.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch


def densenet_forward(edge_attr: torch.Tensor, weights: Sequence[torch.Tensor],
                     biases: Sequence[Optional[torch.Tensor]]) -> torch.Tensor:
    """Linear -> ReLU -> ... -> Linear  (utilities.py:223-227; no out_nonlinearity, no BatchNorm:
    neither is ever enabled by the reference scripts)."""
    h = edge_attr
    n = len(weights)
    for l in range(n):
        h = torch.nn.functional.linear(h, weights[l], biases[l])
        if l != n - 1:
            h = torch.relu(h)
    return h


def nnconv_forward(x: torch.Tensor, edge_index: torch.Tensor, edge_attr: torch.Tensor,
                   weights: Sequence[torch.Tensor], biases: Sequence[Optional[torch.Tensor]],
                   root: Optional[torch.Tensor], bias: Optional[torch.Tensor],
                   aggr: str = "mean", in_channels: Optional[int] = None,
                   out_channels: Optional[int] = None, dtype: torch.dtype = torch.float32,
                   chunk_edges: int = 65536) -> torch.Tensor:
    """Reference-order forward of NNConv_old / torch_geometric.nn.NNConv on CPU.

    x [N,in] (or [N]), edge_index [2,E] int64 (row 0 = source j, row 1 = target i),
    edge_attr [E,k0] (or [E]); returns [N,out] in `dtype` (float32 = the reference's arithmetic,
    float64 = adjudicator).
    """
    x = x.detach().to("cpu", dtype)
    edge_attr = edge_attr.detach().to("cpu", dtype)
    edge_index = edge_index.detach().to("cpu", torch.int64)
    weights = [w.detach().to("cpu", dtype) for w in weights]
    biases = [None if b is None else b.detach().to("cpu", dtype) for b in biases]
    root = None if root is None else root.detach().to("cpu", dtype)
    bias = None if bias is None else bias.detach().to("cpu", dtype)

    # nn_conv.py:269-270
    if x.dim() == 1:
        x = x.unsqueeze(-1)
    if edge_attr.dim() == 1:
        edge_attr = edge_attr.unsqueeze(-1)
    n_nodes = x.size(0)
    cin = x.size(1) if in_channels is None else in_channels
    cout = (weights[-1].size(0) // cin) if out_channels is None else out_channels
    n_edges = edge_index.size(1)
    src, dst = edge_index[0], edge_index[1]

    if aggr in ("add", "mean"):
        out = torch.zeros(n_nodes, cout, dtype=dtype)
    elif aggr == "max":
        out = torch.full((n_nodes, cout), float("-inf"), dtype=dtype)
    else:
        raise ValueError(f"unknown aggr {aggr!r}")

    for e0 in range(0, n_edges, chunk_edges):
        e1 = min(e0 + chunk_edges, n_edges)
        x_j = x.index_select(0, src[e0:e1])                              # propagate: gather
        w = densenet_forward(edge_attr[e0:e1], weights, biases)          # nn_conv.py:274
        w = w.view(-1, cin, cout)
        m = torch.matmul(x_j.unsqueeze(1), w).squeeze(1)                 # nn_conv.py:275
        if aggr == "max":
            out = out.scatter_reduce(0, dst[e0:e1].unsqueeze(1).expand_as(m), m, reduce="amax",
                                     include_self=True)
        else:
            out.index_add_(0, dst[e0:e1], m)                             # scatter add
    count = torch.bincount(dst, minlength=n_nodes)
    if aggr == "mean":
        out = out / count.clamp(min=1).to(dtype).unsqueeze(1)
    elif aggr == "max":
        out = torch.where(count.unsqueeze(1) > 0, out, torch.zeros_like(out))
    # nn_conv.py:277-282
    if root is not None:
        out = out + torch.mm(x, root)
    if bias is not None:
        out = out + bias
    return out


def nnconv_grads(x: torch.Tensor, edge_index: torch.Tensor, edge_attr: torch.Tensor,
                 weights: Sequence[torch.Tensor], biases: Sequence[Optional[torch.Tensor]],
                 root: Optional[torch.Tensor], bias: Optional[torch.Tensor], aggr: str,
                 grad_out: torch.Tensor, chunk_edges: Optional[int] = None):
    """float64 autograd through the restated operator = the reference's `loss.backward()`
    (UAI1_full_resolution.py:266) for loss = sum(out * grad_out).  Pinned against autograd through
    the reference's own module by tests/golden/*_grad.npz.
    `chunk_edges`: the loss is a sum over edges (out_i = sum_{e -> i} m_e / deg_i + x_i root + bias), so the edges may be
    differentiated in pieces and the gradients accumulated - same mathematics, bounds the [E, 4096] float64 tensor of the
    MGKN-scale calls (tests/test_oracle_golden.py checks chunked == whole).
    Returns (grad_x, [grad_W], [grad_b], grad_root or None, grad_bias or None)."""
    n = x.shape[0]
    xs = x.double().requires_grad_(True)
    Ws = [w.double().requires_grad_(True) for w in weights]
    Bs = [None if b is None else b.double().requires_grad_(True) for b in biases]
    r = None if root is None else root.double().requires_grad_(True)
    bb = None if bias is None else bias.double().requires_grad_(True)
    src, dst = edge_index[0], edge_index[1]
    if aggr not in ("add", "mean"):
        raise ValueError(aggr)
    e = int(src.numel())
    if chunk_edges is None or e <= chunk_edges:
        h = densenet_forward(edge_attr.double(), Ws, Bs)
        m = torch.matmul(xs[src].unsqueeze(1), h.view(-1, xs.shape[1], h.shape[1] // xs.shape[1])).squeeze(1)
        out = torch.zeros(n, m.shape[1], dtype=torch.float64).index_add(0, dst, m)
        if aggr == "mean":
            out = out / torch.bincount(dst, minlength=n).clamp(min=1).double().unsqueeze(1)
        if r is not None:
            out = out + xs @ r
        if bb is not None:
            out = out + bb
        (out * grad_out.double()).sum().backward()
    else:
        gT = grad_out.double()
        if aggr == "mean":
            gT = gT / torch.bincount(dst, minlength=n).clamp(min=1).double().unsqueeze(1)
        for lo in range(0, e, chunk_edges):
            sl = slice(lo, lo + chunk_edges)
            h = densenet_forward(edge_attr[sl].double(), Ws, Bs)
            m = torch.matmul(xs[src[sl]].unsqueeze(1), h.view(-1, xs.shape[1], h.shape[1] // xs.shape[1])).squeeze(1)
            (m * gT[dst[sl]]).sum().backward()
        node = torch.zeros(n, grad_out.shape[1], dtype=torch.float64)
        if r is not None:
            node = node + xs @ r
        if bb is not None:
            node = node + bb
        if r is not None or bb is not None:
            (node * grad_out.double()).sum().backward()
    zero = lambda t: torch.zeros_like(t) if t.grad is None else t.grad
    return (zero(xs), [zero(w) for w in Ws], [None if b is None else zero(b) for b in Bs], None if r is None else zero(r),
            None if bb is None else zero(bb))


def nnconv_grads_shared(xs: Sequence[torch.Tensor], edge_index: torch.Tensor, edge_attr: torch.Tensor,
                        weights: Sequence[torch.Tensor], biases: Sequence[Optional[torch.Tensor]],
                        root: Optional[torch.Tensor], bias: Optional[torch.Tensor], aggr: str,
                        grad_outs: Sequence[torch.Tensor], chunk_edges: int = 16384):
    """`KernelNN.forward` applies ONE conv `depth` times with the same edge_attr and weights
    (/root/reference/graph-neural-operator/UAI1_full_resolution.py:29-30) and `loss.backward()` (:266) sums the parameter
    gradients over the applications.  This is float64 autograd of  loss = sum_l sum(conv(x_l) * g_l)  with the applications'
    inputs x_l given (independent leaves): the kernel MLP (nn_conv.py:274, utilities.py:223-227) is evaluated once per edge
    chunk and shared by the `depth` messages, exactly the sharing the reference's graph has.  Equal to the sum of
    `nnconv_grads` over the applications (tests/test_oracle_golden.py), at 1/depth of the cost.
    Returns ([grad_x_l], [grad_W], [grad_b], grad_root or None, grad_bias or None)."""
    if aggr not in ("add", "mean"):
        raise ValueError(aggr)
    n = xs[0].shape[0]
    xl = [x.double().requires_grad_(True) for x in xs]
    Ws = [w.double().requires_grad_(True) for w in weights]
    Bs = [None if b is None else b.double().requires_grad_(True) for b in biases]
    r = None if root is None else root.double().requires_grad_(True)
    bb = None if bias is None else bias.double().requires_grad_(True)
    src, dst = edge_index[0], edge_index[1]
    e = int(src.numel())
    gT = [g.double() for g in grad_outs]
    if aggr == "mean":
        cnt = torch.bincount(dst, minlength=n).clamp(min=1).double().unsqueeze(1)
        gT = [g / cnt for g in gT]
    for lo in range(0, e, max(1, chunk_edges)):
        sl = slice(lo, lo + chunk_edges)
        h = densenet_forward(edge_attr[sl].double(), Ws, Bs)
        we = h.view(-1, xl[0].shape[1], h.shape[1] // xl[0].shape[1])                       # nn_conv.py:274
        loss = 0.0
        for x, g in zip(xl, gT):
            m = torch.matmul(x[src[sl]].unsqueeze(1), we).squeeze(1)                       # nn_conv.py:275
            loss = loss + (m * g[dst[sl]]).sum()
        loss.backward()
    if r is not None or bb is not None:
        loss = 0.0
        for x, g in zip(xl, grad_outs):
            node = torch.zeros(n, g.shape[1], dtype=torch.float64)
            if r is not None:
                node = node + x @ r                                                        # nn_conv.py:277-282
            if bb is not None:
                node = node + bb
            loss = loss + (node * g.double()).sum()
        loss.backward()
    zero = lambda t: torch.zeros_like(t) if t.grad is None else t.grad
    return ([zero(x) for x in xl], [zero(w) for w in Ws], [None if b is None else zero(b) for b in Bs],
            None if r is None else zero(r), None if bb is None else zero(bb))


def nnconv_grad_x_rows(rows: torch.Tensor, edge_index_out: torch.Tensor, edge_attr_out: torch.Tensor,
                       weights: Sequence[torch.Tensor], biases: Sequence[Optional[torch.Tensor]],
                       root: Optional[torch.Tensor], grad_out: torch.Tensor, in_degree: Optional[torch.Tensor],
                       chunk_edges: int = 16384) -> torch.Tensor:
    """float64 rows `rows` of d loss / d x for loss = sum(out * grad_out), given ALL out-edges of those nodes
    (`edge_index_out` [2, E'] with the graph's own node ids, `edge_attr_out` [E', k0]) - for graphs whose full backward does not
    fit a CPU (the 241^2 graph: 95.5 M edges).  The operator is linear in x (nn_conv.py:273-282:
    out_i = sum_{e: j -> i} x_j . W_e / deg_i + x_i . root + bias with W_e = view(nn(pseudo_e), in, out)), so
        d loss / d x_j = sum_{e: j -> i} W_e . (grad_out_i / deg_i) + root . grad_out_j
    needs the out-edges of j only and no x.  `in_degree`: the FULL graph's in-degree per node for aggr='mean'
    (clamp(count, 1)), None for 'add'.  Pinned on CPU to autograd through the restated forward
    (tests/test_oracle_golden.py::test_grad_x_rows_is_autograds_grad_x)."""
    src, dst = edge_index_out[0], edge_index_out[1]
    gT = grad_out.double()
    if in_degree is not None:
        gT = gT / in_degree.clamp(min=1).double().unsqueeze(1)
    Ws = [w.double() for w in weights]
    Bs = [None if b is None else b.double() for b in biases]
    n, cin = grad_out.shape[0], Ws[-1].shape[0] // grad_out.shape[1]
    dx = torch.zeros(n, cin, dtype=torch.float64)
    e = int(src.numel())
    with torch.no_grad():
        for lo in range(0, e, max(1, chunk_edges)):
            sl = slice(lo, lo + chunk_edges)
            we = densenet_forward(edge_attr_out[sl].double(), Ws, Bs).view(-1, cin, grad_out.shape[1])      # nn_conv.py:274
            dx.index_add_(0, src[sl], torch.matmul(we, gT[dst[sl]].unsqueeze(2)).squeeze(2))
        if root is not None:
            dx += grad_out.double() @ root.double().t()
    return dx[rows]


def rel_l2(y: torch.Tensor, y_ref: torch.Tensor) -> float:
    """Relative L2 over the whole output, the `LpLoss.rel` formula for one sample
    (/root/reference/graph-neural-operator/utilities.py:184-196)."""
    y = y.detach().to("cpu", torch.float64).reshape(-1)
    y_ref = y_ref.detach().to("cpu", torch.float64).reshape(-1)
    denom = torch.linalg.vector_norm(y_ref)
    if denom == 0:
        return float(torch.linalg.vector_norm(y - y_ref))
    return float(torch.linalg.vector_norm(y - y_ref) / denom)


def mlp_params(mlp: torch.nn.Module):
    """Pull (weights, biases) of the Linear layers out of a DenseNet-like module, checking that
    the module really is Linear/ReLU alternation (what the oracle restates)."""
    ws: List[torch.Tensor] = []
    bs: List[Optional[torch.Tensor]] = []
    layers = list(mlp.layers) if hasattr(mlp, "layers") else list(mlp)
    expect_linear = True
    for l in layers:
        if isinstance(l, torch.nn.Linear):
            assert expect_linear, "two Linear layers without a non-linearity"
            ws.append(l.weight)
            bs.append(l.bias)
            expect_linear = False
        elif isinstance(l, torch.nn.ReLU):
            assert not expect_linear
            expect_linear = True
        else:
            raise TypeError(f"oracle restates Linear/ReLU chains only, got {type(l).__name__}")
    assert not expect_linear, "DenseNet ends with a Linear layer in every reference script"
    return ws, bs
