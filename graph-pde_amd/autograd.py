"""`torch.autograd.Function` around the fused NNConv (libgpde.so): forward = one gpde_nnconv_fwd
call, backward = one gpde_nnconv_bwd call (include/gpde.h).  Nothing but the inputs is saved: the
backward recomputes the hidden activations chunk by chunk.
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
        ctx.csr, ctx.aggr, ctx.n_layers = csr, aggr, n_layers
        ctx.has_bias = bias is not None
        ctx.attr_needs_grad = edge_attr.requires_grad
        ctx.save_for_backward(x, edge_attr, root, *params)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.attr_needs_grad:
            raise NotImplementedError(
                "gradient with respect to edge_attr is not built (no reference script needs it)")
        x, edge_attr, root, *params = ctx.saved_tensors
        n = ctx.n_layers
        weights, biases = list(params[:n]), list(params[n:])
        gx, gW, gb, groot, gbias = ops.nnconv_backward_raw(
            x, ctx.csr, edge_attr, weights, biases, root, ctx.aggr, grad_out,
            need_root=root is not None, need_bias=ctx.has_bias)
        return (gx, None, None, groot, gbias if ctx.has_bias else None, None, None, *gW, *gb)
