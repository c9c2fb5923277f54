"""Drop-in modules with the reference's surface:

* `NNConv_old`  -- /root/reference/graph-neural-operator/nn_conv.py:197-286 (full w x w edge kernel;
  the class every GKN script instantiates), and
* `NNConv`      -- the name `torch_geometric.nn.NNConv` resolves to for the MGKN scripts
  (MGKN_general_darcy2d.py:8,45,53,61; MGKN_orthogonal_burgers1d.py:8,37); same math.

Same constructor, attributes (`in_channels, out_channels, nn, aggr, root, bias`), parameter
names / shapes (`root [in,out]`, `bias [out]`), `reset_parameters`, `forward(x, edge_index,
edge_attr)`, `__repr__`; instances pickle with `torch.save(model)` (no native handles on the
module).  `forward` runs the fused HIP operator of libgpde.so; there is no other execution path
(CPU tensors raise).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch.nn import Parameter

from . import hidden_cache, ops
from .autograd import NNConvFunction, NNConvHiddenFunction


def _reset(nn):
    """`torch_geometric.nn.inits.reset` (imported at nn_conv.py:4): call reset_parameters() on every
    leaf that has one."""
    def _r(item):
        if hasattr(item, "reset_parameters"):
            item.reset_parameters()
    if nn is not None:
        children = list(nn.children()) if hasattr(nn, "children") else []
        if children:
            for item in children:
                _reset(item)
        else:
            _r(nn)


def _uniform(size, tensor):
    """`torch_geometric.nn.inits.uniform`: U(-1/sqrt(size), 1/sqrt(size))."""
    if tensor is not None:
        bound = 1.0 / math.sqrt(size)
        tensor.data.uniform_(-bound, bound)


class NNConv_old(torch.nn.Module):
    r"""x'_i = Theta x_i + aggr_{j in N(i)} x_j . h_Theta(e_ij)   with h_Theta a kernel MLP emitting
    in_channels*out_channels values per edge (nn_conv.py:197-232)."""

    def __init__(self, in_channels, out_channels, nn, aggr="add", root_weight=True, bias=True,
                 **kwargs):
        super().__init__()
        flow = kwargs.pop("flow", "source_to_target")
        if flow != "source_to_target":
            raise NotImplementedError("only flow='source_to_target' (the reference default) is built")
        if kwargs:
            raise TypeError(f"unexpected arguments {sorted(kwargs)}")
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.nn = nn
        self.aggr = aggr
        self.flow = flow
        if root_weight:
            self.root = Parameter(torch.Tensor(in_channels, out_channels))
        else:
            self.register_parameter("root", None)
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):                       # nn_conv.py:261-265
        _reset(self.nn)
        size = self.in_channels
        _uniform(size, self.root)
        _uniform(size, self.bias)

    def forward(self, x, edge_index, edge_attr):      # nn_conv.py:267-271
        x = x.unsqueeze(-1) if x.dim() == 1 else x
        if isinstance(edge_attr, ops.NodeAttr):
            # opt-in: attributes read from node data inside the kernel.  Inference on
            # the default f16-split kernel; anything else takes the tensor the reference would build.
            lin = ops.mlp_linears(self.nn)
            needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
            if not needs_grad and len(lin) == 3 and ops.DEFAULT_PRECISION == "f16split" and \
                    self.in_channels == ops.WIDTH and self.out_channels == ops.WIDTH:
                csr = ops.csr_for(edge_index, x.size(0))
                pm = ops.pack_mlp([l.weight for l in lin], [l.bias for l in lin])
                return ops.nnconv_forward_nodeattr_raw(x, csr, edge_attr, pm, self.root, self.bias, self.aggr)
            edge_attr = edge_attr.materialize(edge_index)
        pseudo = edge_attr.unsqueeze(-1) if edge_attr.dim() == 1 else edge_attr
        if self.in_channels != ops.WIDTH or self.out_channels != ops.WIDTH:
            raise NotImplementedError(
                f"the MI355X operator is built for in_channels = out_channels = {ops.WIDTH} (the "
                f"width of every reference configuration), got {self.in_channels}->{self.out_channels}")
        lin = ops.mlp_linears(self.nn)
        weights = [l.weight for l in lin]
        biases = [l.bias for l in lin]
        # cross-depth reuse (hidden_cache.py): this module applied again with the same edge_attr and
        # weights shares one hidden-activation tensor with the earlier applications
        if hidden_cache.MODE != "off" and x.is_cuda and self.aggr in ("add", "mean") and \
                pseudo.dtype == torch.float32 and x.dtype == torch.float32:
            csr = ops.csr_for(edge_index, x.size(0))
            pm = ops.pack_mlp(weights, biases)
            no_grad = not (torch.is_grad_enabled() and
                           (x.requires_grad or any(p.requires_grad for p in self.parameters())))
            hit = hidden_cache.lookup(self, pseudo, csr, pm, weights, biases, allow_partial=no_grad)
            if hit is not None:
                hidden, hmax, hn = hit
                if hn < csr.n_nodes:        # H of the leading nodes only: mixed forward (inference)
                    return ops.nnconv_forward_mixed_raw(x, csr, pseudo, hidden, hmax, hn, pm, self.root,
                                                        self.bias, self.aggr)
                return NNConvHiddenFunction.apply(x, hidden, csr, pm, weights[-1], biases[-1],
                                                  self.root, self.bias, self.aggr, hmax)
        return NNConvFunction.apply(x, edge_index, pseudo, self.root, self.bias, self.aggr,
                                    len(weights), *weights, *biases)

    def __repr__(self):                               # nn_conv.py:284-286
        return "{}({}, {})".format(self.__class__.__name__, self.in_channels, self.out_channels)


class NNConv(NNConv_old):
    """`torch_geometric.nn.NNConv` as the MGKN scripts use it (full edge kernel, same math as
    NNConv_old).  Note: the dead-code diagonal variant that
    graph-neural-operator/nn_conv.py:8-96 also calls `NNConv` is never instantiated by any script
    and is not built."""
    pass


ECConv = NNConv
