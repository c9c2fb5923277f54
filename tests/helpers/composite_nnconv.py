"""Test infrastructure: the reference operator as a differentiable composite of stock torch ops, for script-level parity.

`forward(x, edge_index, edge_attr)` below is the reference's own chain - `DenseNet.forward` (oracle.nnconv_oracle.
densenet_forward = /root/reference/graph-neural-operator/utilities.py:223-227), `view(-1, in, out)` + `matmul`
(nn_conv.py:274-275), PyG's gather by `edge_index[0]` / scatter-mean over `edge_index[1]`, `update`
(nn_conv.py:277-282) - on whatever device / dtype the script's tensors live, differentiated by torch autograd: what the
scripts ran on before this operator existed.  `tests/test_gpu_reference_scripts.py` runs every staged script twice, once on
libgpde.so and once with the modules' `forward` replaced by this, and compares the numbers the script prints.
Only `scripts/run_reference_script.py --composite` (a test harness) installs it; the product never imports it."""
import torch

from oracle.nnconv_oracle import densenet_forward, mlp_params


def composite_forward(self, x, edge_index, edge_attr, **kw):
    if kw.get("residual") is not None or kw.get("activation") is not None:
        raise NotImplementedError("the reference scripts never pass residual / activation")
    x = x.unsqueeze(-1) if x.dim() == 1 else x                                            # nn_conv.py:269-270
    pseudo = edge_attr.unsqueeze(-1) if edge_attr.dim() == 1 else edge_attr
    ws, bs = mlp_params(self.nn)
    src, dst = edge_index[0], edge_index[1]
    n = x.size(0)
    weight = densenet_forward(pseudo, ws, bs).view(-1, self.in_channels, self.out_channels)  # nn_conv.py:274
    m = torch.matmul(x.index_select(0, src).unsqueeze(1), weight).squeeze(1)               # nn_conv.py:275
    out = torch.zeros(n, self.out_channels, dtype=m.dtype, device=m.device).index_add(0, dst, m)
    if self.aggr == "mean":
        cnt = torch.bincount(dst, minlength=n).clamp(min=1).to(m.dtype).unsqueeze(1)
        out = out / cnt
    elif self.aggr != "add":
        raise NotImplementedError(self.aggr)
    if self.root is not None:                                                              # nn_conv.py:277-282
        out = out + torch.mm(x, self.root)
    if self.bias is not None:
        out = out + self.bias
    return out


def install():
    """Replace `forward` of the native modules (graph_pde_amd.nn_conv.NNConv_old / NNConv, the classes the shims re-export)
    by the composite.  Returns a counter dict: 'calls' moves with every composite forward."""
    from graph_pde_amd import nn_conv
    counter = {"calls": 0}

    def fwd(self, x, edge_index, edge_attr, **kw):
        counter["calls"] += 1
        return composite_forward(self, x, edge_index, edge_attr, **kw)
    for cls in {nn_conv.NNConv_old, getattr(nn_conv, "NNConv", nn_conv.NNConv_old)}:
        cls.forward = fwd
    return counter
