"""`torch.autograd.Function` around the fused NNConv forward (libgpde.so).

Forward = one gpde_nnconv_fwd call (include/gpde.h).  Backward is SURVEY.md §8 row f1 ("next"):
until the native backward lands it raises instead of silently falling back to a composite path.
"""
from __future__ import annotations

import torch

from . import ops


class NNConvFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, edge_index, edge_attr, root, bias, aggr, n_layers, *params):
        weights = list(params[:n_layers])
        biases = list(params[n_layers:])
        ops._require_cuda(x, "x")
        csr = ops.csr_for(edge_index, x.size(0))
        pm = ops.pack_mlp(weights, biases)
        out = ops.nnconv_forward_raw(x.detach(), csr, edge_attr.detach(), pm, root, bias, aggr)
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        raise NotImplementedError(
            "backward of the fused MI355X NNConv is not built yet (SURVEY.md §8 row f1); run "
            "inference under torch.no_grad() or detach the inputs")
