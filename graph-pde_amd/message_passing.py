"""`MessagePassing` - the base class the reference's operator derives from
(/root/reference/graph-neural-operator/nn_conv.py:3 `from torch_geometric.nn.conv import MessagePassing`,
`class NNConv_old(MessagePassing)` :197, `super().__init__(aggr=aggr, **kwargs)` :242, `self.propagate(...)` :271).
PyG is not installable here, so the class lives in this package and the import facade
`shims/torch_geometric/nn/conv` re-exports it: `isinstance(conv, torch_geometric.nn.conv.MessagePassing)` holds for
the native modules under the facade exactly as it does for the reference's under PyG.

`graph_pde_amd.nn_conv.NNConv_old` OVERRIDES `propagate` with the fused HIP operator; the generic `propagate` below
(PyG ~1.3 semantics) serves only third-party subclasses that are not the hot path.
"""
import inspect

import torch


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", flow="source_to_target"):
        super().__init__()
        assert aggr in ("add", "mean", "max") and flow in ("source_to_target", "target_to_source")
        self.aggr, self.flow = aggr, flow
        self.__message_args__ = inspect.getfullargspec(self.message)[0][1:]
        self.__update_args__ = inspect.getfullargspec(self.update)[0][2:]

    def propagate(self, edge_index, size=None, **kwargs):
        i, j = (1, 0) if self.flow == "source_to_target" else (0, 1)
        n = None
        margs = []
        for arg in self.__message_args__:
            if arg.endswith("_i") or arg.endswith("_j"):
                t = kwargs[arg[:-2]]
                n = t.size(0) if n is None else n
                margs.append(t.index_select(0, edge_index[i if arg.endswith("_i") else j]))
            else:
                margs.append(kwargs[arg])
        out = self.message(*margs)
        n = n if size is None else (size[i] if isinstance(size, (list, tuple)) else size)
        idx = edge_index[i]
        if self.aggr in ("add", "mean"):
            res = torch.zeros(n, *out.shape[1:], dtype=out.dtype, device=out.device).index_add_(0, idx, out)
            if self.aggr == "mean":
                cnt = torch.bincount(idx, minlength=n).clamp(min=1).to(out.dtype)
                res = res / cnt.view(-1, *([1] * (out.dim() - 1)))
        else:
            res = torch.full((n, *out.shape[1:]), -1e9, dtype=out.dtype, device=out.device)
            res = res.scatter_reduce(0, idx.view(-1, *([1] * (out.dim() - 1))).expand_as(out), out, "amax")
            res = torch.where(res == -1e9, torch.zeros_like(res), res)      # (not in place: amax saved its output for the backward)
        return self.update(res, *[kwargs[a] for a in self.__update_args__])

    def message(self, x_j):
        return x_j

    def update(self, aggr_out):
        return aggr_out
