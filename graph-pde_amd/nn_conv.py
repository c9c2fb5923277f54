"""Drop-in modules with the reference's surface:

* `NNConv_old`  -- /root/reference/graph-neural-operator/nn_conv.py:197-286 (full w x w edge kernel;
  the class every GKN script instantiates), and
* `NNConv`      -- the name `torch_geometric.nn.NNConv` resolves to for the MGKN scripts
  (MGKN_general_darcy2d.py:8,45,53,61; MGKN_orthogonal_burgers1d.py:8,37); same math.

Same constructor, attributes (`in_channels, out_channels, nn, aggr, root, bias`), parameter
names / shapes (`root [in,out]`, `bias [out]`), `reset_parameters`, `forward(x, edge_index,
edge_attr)`, `message(x_j, pseudo)`, `update(aggr_out, x)`, `__repr__`; instances pickle with
`torch.save(model)` (no native handles on the module).  `forward` runs the fused HIP operator of
libgpde.so; there is no other execution path: CPU tensors (a model moved back with `model.cpu()`,
UAI1_full_resolution.py:287-303) are staged to the current HIP device, run through the same kernels and
the result is copied back - without a HIP device the call raises.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch.nn import Parameter

from . import hidden_cache, ops
from . import autograd as _ag
from .autograd import NNConvDeferredFunction, NNConvFunction, NNConvHiddenFunction, SharedParamFunction, WeConvFunction
from .message_passing import MessagePassing


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
    """`torch_geometric.nn.inits.uniform`: U(-1/sqrt(size), 1/sqrt(size)).  Written under no_grad on the
    parameter itself (not through `.data`): the in-place version counter moves, so packed-weight caches
    keyed on it cannot serve the old values."""
    if tensor is not None:
        bound = 1.0 / math.sqrt(size)
        with torch.no_grad():
            tensor.uniform_(-bound, bound)


class NNConv_old(MessagePassing):
    r"""x'_i = Theta x_i + aggr_{j in N(i)} x_j . h_Theta(e_ij)   with h_Theta a kernel MLP emitting
    in_channels*out_channels values per edge (nn_conv.py:197-232).  Derives `MessagePassing` like the reference
    (nn_conv.py:197, 242) and overrides `propagate` with the fused HIP operator."""

    def __init__(self, in_channels, out_channels, nn, aggr="add", root_weight=True, bias=True,
                 **kwargs):
        flow = kwargs.pop("flow", "source_to_target")
        if flow != "source_to_target":
            raise NotImplementedError("only flow='source_to_target' (the reference default) is built")
        if kwargs:
            raise TypeError(f"unexpected arguments {sorted(kwargs)}")
        if aggr not in ("add", "mean", "max"):
            raise ValueError(f"aggr must be 'add', 'mean' or 'max' (nn_conv.py:222-224), got {aggr!r}")
        super().__init__(aggr=aggr, flow=flow)                     # nn_conv.py:242
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.nn = nn
        if root_weight:
            self.root = Parameter(torch.Tensor(in_channels, out_channels))
        else:
            self.register_parameter("root", None)
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        """State dicts written by newer PyG releases store the root weight as `lin.weight [out, in]` (a bias-free
        Linear) instead of `root [in, out]`: accepted and transposed."""
        k_new, k_old = prefix + "lin.weight", prefix + "root"
        if k_new in state_dict and k_old not in state_dict:
            state_dict[k_old] = state_dict.pop(k_new).t().contiguous()
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def reset_parameters(self):                       # nn_conv.py:261-265
        _reset(self.nn)
        size = self.in_channels
        _uniform(size, self.root)
        _uniform(size, self.bias)

    def forward(self, x, edge_index, edge_attr, *, residual=None, activation=None):      # nn_conv.py:267-271
        """The reference signature `forward(x, edge_index, edge_attr)`.  Two keyword-only, opt-in extras fuse the
        callers' elementwise glue into the operator's last kernel: `residual` (a [N, 64] tensor
        added to the result) and `activation="relu"` - `conv(x, ei, ea, residual=x, activation="relu")` equals
        `F.relu(x + conv(x, ei, ea))` (MGKN_general_darcy2d.py:79-80).  Fused for inference on device tensors;
        when a gradient is needed the same value is composed from the unfused operator and torch ops."""
        if activation not in (None, "relu"):
            raise ValueError(f"activation must be None or 'relu', got {activation!r}")
        with ops.ver_scope():        # (inference tensors: one content checksum per tensor and call, shared by all cache keys)
            return self._forward(x, edge_index, edge_attr, residual, activation)

    def _forward(self, x, edge_index, edge_attr, residual, activation):
        if residual is not None or activation is not None:
            return self._forward_act(x, edge_index, edge_attr, residual, activation == "relu")
        x = x.unsqueeze(-1) if x.dim() == 1 else x
        if not x.is_cuda:
            return self._forward_staged(x, edge_index, edge_attr)
        if isinstance(edge_attr, ops.NodeAttr) and not self._nn_is_linear_relu_chain():
            edge_attr = edge_attr.materialize(edge_index.edge_index if isinstance(edge_index, ops.Csr) else edge_index)
        if isinstance(edge_attr, ops.NodeAttr):
            # opt-in: attributes read from node data inside the kernel.  Inference on
            # the default f16-split kernel; anything else takes the tensor the reference would build.
            lin = ops.mlp_linears(self.nn)
            needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
            if torch.is_grad_enabled() and edge_attr.table.requires_grad:
                # a node table that wants a gradient (learned positions / coefficients): the node-table kernels do not differentiate
                # the table, the materialised tensor does (differentiable gather -> grad_edge_attr of gpde_nnconv_bwd) - ADVICE r4
                edge_attr = edge_attr.materialize(edge_index.edge_index if isinstance(edge_index, ops.Csr) else edge_index)
                pseudo = edge_attr.unsqueeze(-1) if edge_attr.dim() == 1 else edge_attr
                return self.propagate(edge_index, x=x, pseudo=pseudo)
            if not needs_grad and self.aggr != "max" and len(lin) == 3 and ops.DEFAULT_PRECISION == "f16split" and \
                    self.in_channels == ops.WIDTH and self.out_channels == ops.WIDTH:
                csr = ops.csr_for(edge_index, x.size(0))
                pm = ops.pack_mlp([l.weight for l in lin], [l.bias for l in lin])
                return ops.nnconv_forward_nodeattr_raw(x, csr, edge_attr, pm, self.root, self.bias, self.aggr)
            if needs_grad and self.aggr in ("add", "mean") and x.dtype == torch.float32 and self.in_channels == ops.WIDTH and \
                    self.out_channels == ops.WIDTH and all(l.bias is not None for l in lin) and \
                    ops.nodeattr_train_supported([lin[0].in_features] + [l.out_features for l in lin]):
                # training from node data (round 4): the descriptor travels the ordinary path - direct operator, shared hidden
                # activations or the virtual-H node - and every native call reads the table (`node_attr` of the entry points)
                return self.propagate(edge_index, x=x, pseudo=edge_attr)
            edge_attr = edge_attr.materialize(edge_index.edge_index if isinstance(edge_index, ops.Csr) else edge_index)
        pseudo = edge_attr.unsqueeze(-1) if edge_attr.dim() == 1 else edge_attr
        return self.propagate(edge_index, x=x, pseudo=pseudo)                            # nn_conv.py:271

    def propagate(self, edge_index, size=None, **kwargs):
        """`propagate(edge_index, x=x, pseudo=pseudo)` (the reference's call, nn_conv.py:271): gather x_j, `message`,
        aggregate over `edge_index[1]`, `update` - as ONE native operator (gpde_nnconv_fwd) instead of PyG's
        index_select / message / scatter / update chain.  `size` may only restate the node count."""
        if set(kwargs) != {"x", "pseudo"}:
            raise TypeError(f"propagate() takes x= and pseudo= (nn_conv.py:271), got {sorted(kwargs)}")
        x, pseudo = kwargs["x"], kwargs["pseudo"]
        x = x.unsqueeze(-1) if x.dim() == 1 else x
        pseudo = pseudo.unsqueeze(-1) if pseudo.dim() == 1 else pseudo
        if size is not None:
            sz = list(size) if isinstance(size, (list, tuple)) else [size, size]
            if any(v is not None and int(v) != x.size(0) for v in sz):
                raise NotImplementedError("bipartite propagate (size != [N, N]) is not built: no graph-pde script uses it")
        if not x.is_cuda:
            return self._forward_staged(x, edge_index, pseudo)
        if not self._nn_is_linear_relu_chain():
            return self._propagate_general_nn(x, edge_index, pseudo)
        lin = ops.mlp_linears(self.nn)
        weights = [l.weight for l in lin]
        biases = [l.bias for l in lin]
        return self._propagate(x, edge_index, pseudo, weights, biases, self.root, self.bias, use_hidden_cache=True)

    def _nn_is_linear_relu_chain(self) -> bool:
        try:
            ops.mlp_linears(self.nn)
            return True
        except NotImplementedError:
            return False

    def _propagate_general_nn(self, x, edge_index, pseudo):
        """`nn` is "a neural network h_Theta ... e.g. torch.nn.Sequential" (nn_conv.py:217-221): anything that is not the
        Linear / ReLU chain the fused kernels re-associate - `DenseNet(normalize=True)` (BatchNorm1d) or `out_nonlinearity`
        (utilities.py:207-221), `DenseNet_sin` (multipole utilities.py:233-252), a user's own module.  The reference's own order
        then: `weight = self.nn(pseudo).view(-1, in, out)` (nn_conv.py:274) by the caller's module as torch ops on the device -
        16 KiB per edge, exactly the tensor the reference materialises - and message / aggregate / update (nn_conv.py:275-282) as
        ONE native kernel over it (gpde_nnconv_fwd_edgeweights; backward gpde_nnconv_bwd_edgeweights, whose dL/dW_e autograd
        carries back into `nn`).  The operator itself never leaves libgpde.so; what cannot be fused is the caller's network."""
        self._check_width()
        if self.aggr not in ("add", "mean"):
            return MessagePassing.propagate(self, edge_index.edge_index if isinstance(edge_index, ops.Csr) else edge_index, x=x, pseudo=pseudo)
        if x.dtype != torch.float32:
            raise NotImplementedError(f"a kernel network outside the Linear / ReLU chain: float32 only (x is {x.dtype})")
        csr = ops.csr_for(edge_index, x.size(0))
        need = csr.n_edges * ops.WIDTH * ops.WIDTH * 4
        free, _ = ops.device_free_bytes(x.device)
        if 2 * need > free:
            raise RuntimeError(f"kernel network {type(self.nn).__name__} is not a Linear / ReLU chain: its per-edge weights are materialised as in the "
                               f"reference (nn_conv.py:274) - {csr.n_edges} edges x 16 KiB = {need / 2**30:.1f} GiB (twice that with gradients), "
                               f"{free / 2**30:.1f} GiB free")
        perm = csr.perm.long()
        # rows in CSR slot order (the order the kernels address W_e in); `nn` acts row by row, batch statistics are order-free
        pseudo_s = pseudo if bool(getattr(csr, "_perm_is_identity", False)) else pseudo.index_select(0, perm)
        weight = self.nn(pseudo_s)
        if weight.dim() != 2 or weight.size(0) != csr.n_edges or weight.size(1) != ops.WIDTH * ops.WIDTH:
            raise ValueError(f"nn(pseudo) must be [E, {ops.WIDTH * ops.WIDTH}] (in_channels * out_channels, nn_conv.py:274), got {tuple(weight.shape)}")
        weight = weight.float().contiguous()
        if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or any(p is not None and p.requires_grad for p in (self.root, self.bias))):
            return WeConvFunction.apply(x, weight, csr, self.root, self.bias, self.aggr, None)
        return ops.nnconv_forward_edgeweights_raw(x, csr, weight.detach(), self.root, self.bias, self.aggr)

    def _forward_act(self, x, edge_index, edge_attr, residual, relu):
        attr_grad = (not isinstance(edge_attr, ops.NodeAttr)) and torch.is_tensor(edge_attr) and edge_attr.requires_grad
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or attr_grad or
                                                  (residual is not None and residual.requires_grad) or
                                                  any(p.requires_grad for p in self.parameters()))
        fusable = x.is_cuda and x.dim() == 2 and not isinstance(edge_attr, ops.NodeAttr) and not needs_grad and \
            x.dtype == torch.float32 and (residual is None or residual.device == x.device) and self._nn_is_linear_relu_chain()
        if not fusable:
            y = self.forward(x, edge_index, edge_attr)
            if residual is not None:
                y = residual + y
            return torch.relu(y) if relu else y
        self._check_width()
        pseudo = edge_attr.unsqueeze(-1) if edge_attr.dim() == 1 else edge_attr
        lin = ops.mlp_linears(self.nn)
        weights = [l.weight for l in lin]
        biases = [l.bias for l in lin]
        csr = ops.csr_for(edge_index, x.size(0))
        pm = ops.pack_mlp(weights, biases)
        call = self._edge_weights_call(x, csr, pseudo, pm, weights, biases, self.root, self.bias, residual, relu)
        if call is not None:
            return ops.nnconv_forward_edgeweights_group([call])[0]
        if hidden_cache.MODE != "off" and self.aggr in ("add", "mean") and pseudo.dtype == torch.float32:
            hit = hidden_cache.lookup(self, pseudo, csr, pm, weights, biases, allow_partial=False)
            if hit is not None and hit[2] == csr.n_nodes:
                return ops.nnconv_forward_hidden_raw(x, csr, hit[0].detach(), pm, self.root, self.bias, self.aggr,
                                                     hmax=hit[1], residual=residual, relu=relu)
        return ops.nnconv_forward_raw(x, csr, pseudo, pm, self.root, self.bias, self.aggr, residual=residual, relu=relu)

    def _check_width(self):
        if self.in_channels != ops.WIDTH or self.out_channels != ops.WIDTH:
            raise NotImplementedError(
                f"the MI355X operator is built for in_channels = out_channels = {ops.WIDTH} (the "
                f"width of every reference configuration), got {self.in_channels}->{self.out_channels}")

    def _propagate(self, x, edge_index, pseudo, weights, biases, root, bias, use_hidden_cache):
        """propagate() of the reference (gather, message, aggregate, update) as ONE native operator on device
        tensors; `weights / biases / root / bias` are the tensors to use (the module's own, or staged copies)."""
        self._check_width()
        no_grad = not (torch.is_grad_enabled() and
                       (x.requires_grad or pseudo.requires_grad or any(p.requires_grad for p in self.parameters())))
        if self.aggr == "max" and not no_grad:
            # a gradient through 'max' (no graph-pde script uses it; a freshly built module in grad mode lands here): PyG's own
            # chain - gather, `message`, segment max, `update` - with the native `message()` / `update()`,
            # both differentiable; the fused inference kernel serves no_grad calls
            if isinstance(edge_index, ops.Csr):
                edge_index = edge_index.edge_index
            return MessagePassing.propagate(self, edge_index, x=x, pseudo=pseudo)
        if no_grad and pseudo.dtype == torch.float32 and x.dtype == torch.float32 and (use_hidden_cache or self.aggr == "max"):
            # inference on a low in-degree / small graph (or aggr='max'): one streaming kernel over the cached per-edge
            # weights (hidden_cache.lookup_edge_weights)
            csr = ops.csr_for(edge_index, x.size(0))
            pm = ops.pack_mlp(weights, biases)
            call = self._edge_weights_call(x, csr, pseudo, pm, weights, biases, root, bias, None, False)
            if call is not None:
                return ops.nnconv_forward_edgeweights_group([call])[0]
        if self.aggr == "max":
            if pseudo.dtype != torch.float32 or x.dtype != torch.float32:
                raise NotImplementedError(f"aggr='max': float32 only (got x {x.dtype}, edge_attr {pseudo.dtype})")
            raise NotImplementedError(
                f"aggr='max': the per-edge weights of this call ({ops.csr_for(edge_index, x.size(0)).n_edges} edges x 16 KiB) "
                "exceed the cache budget (GPDE_HIDDEN_CACHE_GB / GPDE_EDGE_WEIGHT_CACHE_GB)")
        # cross-depth reuse (hidden_cache.py): this module applied again with the same edge_attr and
        # weights shares one hidden-activation tensor with the earlier applications
        if use_hidden_cache and hidden_cache.MODE != "off" and self.aggr in ("add", "mean") and \
                pseudo.dtype == torch.float32 and x.dtype == torch.float32:
            csr = ops.csr_for(edge_index, x.size(0))
            pm = ops.pack_mlp(weights, biases)
            defer_ok = (not no_grad) and hidden_cache.defer_possible(pseudo, pm, weights, biases, self.aggr)
            hit = hidden_cache.lookup(self, pseudo, csr, pm, weights, biases, allow_partial=no_grad or defer_ok)
            if hit is not None and (hit[2] == csr.n_nodes or no_grad):
                hidden, hmax, hn = hit
                if hn < csr.n_nodes:        # H of the leading nodes only: mixed forward (inference)
                    return ops.nnconv_forward_mixed_raw(x, csr, pseudo, hidden, hmax, hn, pm, root,
                                                        bias, self.aggr)
                if not no_grad:
                    # low in-degree / small graph (the MGKN V-cycles): the per-edge weights as a shared autograd node, the
                    # call itself one streaming kernel forward and one backward
                    we = hidden_cache.lookup_edge_weights_train(self, hidden, csr, pm, weights, biases)
                    if we is not None:
                        tok = hidden_cache.we_token_of(self, we)
                        if tok is not None and _ag.ACCUMULATE_GRAD_HIDDEN:
                            # root / bias behind private identity nodes, one per step and module: the applications sum their
                            # gradients in place there (autograd.SharedParamFunction)
                            if tok.side_in is None or tok.side_in[2] is not root or tok.side_in[3] is not bias:
                                tok.side_in = (None if root is None else SharedParamFunction.apply(root, tok, 0),
                                               None if bias is None else SharedParamFunction.apply(bias, tok, 1), root, bias)
                            return WeConvFunction.apply(x, we, csr, tok.side_in[0], tok.side_in[1], self.aggr, tok)
                        return WeConvFunction.apply(x, we, csr, root, bias, self.aggr, None)
                return NNConvHiddenFunction.apply(x, hidden, csr, pm, weights[-1], biases[-1],
                                                  root, bias, self.aggr, hmax, hidden_cache.token_of(self, hidden, csr))
            if not no_grad:
                # H wanted but too large for the device (the 241^2 graph: 391 GB): the applications of this forward share a
                # "virtual H" node instead - light backward per application, ONE deferred pass for the hidden layers; the
                # part of H that does fit (`hit`: the in-edges of the leading nodes) is read instead of recomputed
                d = hidden_cache.lookup_deferred(self, pseudo, csr, pm, weights, biases, self.aggr, hpart=hit)
                if d is not None:
                    return NNConvDeferredFunction.apply(x, d[0], csr, pseudo, root, bias, self.aggr, d[1],
                                                        len(weights), *weights, *biases)
        return NNConvFunction.apply(x, edge_index, pseudo, root, bias, self.aggr,
                                    len(weights), *weights, *biases)

    def _edge_weights_call(self, x, csr, pseudo, pm, weights, biases, root, bias, residual, relu, explicit=False):
        """Descriptor of this call for ops.nnconv_forward_edgeweights_group, or None when the per-edge weight form does
        not apply (graph too dense / too large; not opted in - hidden_cache.WE_MODE - or, in `auto`, module not seen
        repeating yet).  `explicit`: the caller opted in (nnconv_group).  Callers need NO gradient."""
        force = self.aggr == "max"
        if not hidden_cache.edge_weights_qualify(csr, force, explicit):
            return None
        we = hidden_cache.lookup_edge_weights(self, pseudo, csr, pm, weights, biases, force=force, explicit=explicit)
        if we is None:
            return None
        return dict(x=x, csr=csr, edge_weights=we, root=root, bias=bias, aggr=self.aggr, residual=residual, relu=relu)

    def _params_on(self, dev, need_grad):
        """The module's parameters as tensors on `dev`: (kernel-MLP weights, biases, root, bias).  Parameters already
        there are returned AS THEY ARE (same objects: the pack cache of ops.pack_mlp recognises them).  CPU parameters
        (a model moved back with `model.cpu()`): a differentiable `.to(dev)` when a gradient is needed, otherwise the
        device copy cached by ops.stage_const on the CPU tensor's storage + version - repeated evaluation calls then
        share one copy and one pack instead of re-packing 24 MB per call."""
        def one(t):
            if t is None or t.device == dev:
                return t
            if need_grad and t.requires_grad:
                return t.to(dev)
            return ops.stage_const(t, dev)
        lin = ops.mlp_linears(self.nn)
        return [one(l.weight) for l in lin], [one(l.bias) for l in lin], one(self.root), one(self.bias)

    def _forward_staged(self, x, edge_index, edge_attr):
        """CPU tensors (`model.cpu()` + evaluation, UAI1_full_resolution.py:287-303): inputs and
        parameters are copied to the current HIP device (`.to()` is differentiable, so gradients flow back to
        the CPU parameters), the SAME kernels run, the output returns to the caller's device."""
        dev = ops.staging_device()
        if isinstance(edge_attr, ops.NodeAttr):
            edge_attr = edge_attr.materialize(edge_index.edge_index if isinstance(edge_index, ops.Csr) else edge_index)
        pseudo = edge_attr.unsqueeze(-1) if edge_attr.dim() == 1 else edge_attr
        ei_d = edge_index if isinstance(edge_index, ops.Csr) else ops.stage_const(edge_index, dev)
        ea_d = pseudo.to(dev) if pseudo.requires_grad else ops.stage_const(pseudo, dev)
        weights, biases, root, bias = self._params_on(dev, torch.is_grad_enabled())
        out = self._propagate(x.to(dev), ei_d, ea_d, weights, biases, root, bias, use_hidden_cache=False)
        return out.to(x.device)

    def message(self, x_j, pseudo):                    # nn_conv.py:273-275
        """m_e = x_j[e] . W(pseudo_e)  ([E, in] x [E, in, out] -> [E, out]).  Computed by the same native
        operator on the graph in which every edge has its own target (E nodes, edge e: e -> e, aggr='add', no
        root / bias): out[e] = x_j[e] . h_Theta(pseudo_e), exactly the reference's message."""
        x_j = x_j.unsqueeze(-1) if x_j.dim() == 1 else x_j
        pseudo = pseudo.unsqueeze(-1) if pseudo.dim() == 1 else pseudo
        e = x_j.size(0)
        dev = x_j.device if x_j.is_cuda else ops.staging_device()
        ar = torch.arange(e, device=dev, dtype=torch.int64)
        ident = torch.stack([ar, ar])
        weights, biases, _, _ = self._params_on(dev, torch.is_grad_enabled())
        self._check_width()
        out = NNConvFunction.apply(x_j.to(dev), ops.build_csr(ident, e), pseudo.to(dev), None, None, "add", len(weights),
                                   *weights, *biases)
        return out.to(x_j.device)

    def update(self, aggr_out, x):                     # nn_conv.py:277-282
        """aggr_out + x . root + bias.  The `x . root + bias` term is the native operator on the graph without
        edges (its epilogue kernel); the sum with aggr_out is one elementwise add."""
        if self.root is None and self.bias is None:
            return aggr_out
        self._check_width()
        dev = x.device if x.is_cuda else ops.staging_device()
        empty = torch.empty(2, 0, dtype=torch.int64, device=dev)
        need_grad = torch.is_grad_enabled()
        weights, biases, root, bias = self._params_on(dev, need_grad)
        k0 = weights[0].size(1)
        if not (need_grad and (x.requires_grad or any(t is not None and t.requires_grad for t in (root, bias)))):
            csr = ops.build_csr(empty, x.size(0))      # edgeless: no sort, nothing cached
            pm = ops.pack_mlp(weights, biases)
            term = ops.nnconv_forward_raw(x.to(dev), csr, torch.empty(0, k0, device=dev), pm, root, bias, "add")
        else:
            # the kernel MLP takes no part in update(): its (staged) parameters enter detached
            term = NNConvFunction.apply(x.to(dev), ops.build_csr(empty, x.size(0)), torch.empty(0, k0, device=dev), root, bias, "add", len(weights),
                                        *[w.detach() for w in weights], *[None if b is None else b.detach() for b in biases])
        return aggr_out + term.to(aggr_out.device)

    def __repr__(self):                               # nn_conv.py:284-286
        return "{}({}, {})".format(self.__class__.__name__, self.in_channels, self.out_channels)


class NNConv(NNConv_old):
    """`torch_geometric.nn.NNConv` as the MGKN scripts use it (full edge kernel, same math as
    NNConv_old).  Note: the dead-code diagonal variant that
    graph-neural-operator/nn_conv.py:8-96 also calls `NNConv` is never instantiated by any script
    and is not built."""
    pass


ECConv = NNConv


def nnconv_group(calls):
    """Run several INDEPENDENT NNConv calls - e.g. the 13 `conv_list[l](phi[l], ...)` of one upward sweep of the
    MGKN-orthogonal V-cycle, whose inputs the downward pass fixed beforehand (MGKN_orthogonal_burgers1d.py:65-82) - and
    return their outputs in order.  `calls`: sequence of (conv, x, edge_index, edge_attr[, residual[, activation]]).
    Calls that can run from cached per-edge weights (inference, low in-degree / small graphs: DESIGN.md §6d) share ONE
    kernel launch per 16 (gpde_nnconv_fwd_edgeweights_group); every other call runs as `conv(x, edge_index, edge_attr,
    residual=, activation=)`.  The result of each call is bit-identical to calling the module on its own."""
    outs = [None] * len(calls)
    batch, where = [], []
    for k, c in enumerate(calls):
        conv, x, edge_index, edge_attr = c[0], c[1], c[2], c[3]
        residual = c[4] if len(c) > 4 else None
        activation = c[5] if len(c) > 5 else None
        if activation not in (None, "relu"):
            raise ValueError(f"activation must be None or 'relu', got {activation!r}")
        call = None
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or (residual is not None and residual.requires_grad) or
                                                  (torch.is_tensor(edge_attr) and edge_attr.requires_grad) or
                                                  any(p.requires_grad for p in conv.parameters()))
        if not needs_grad and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and torch.is_tensor(edge_attr) and \
                edge_attr.dtype == torch.float32 and (residual is None or residual.device == x.device):
            conv._check_width()
            pseudo = edge_attr.unsqueeze(-1) if edge_attr.dim() == 1 else edge_attr
            lin = ops.mlp_linears(conv.nn)
            weights, biases = [l.weight for l in lin], [l.bias for l in lin]
            csr = ops.csr_for(edge_index, x.size(0))
            call = conv._edge_weights_call(x, csr, pseudo, ops.pack_mlp(weights, biases), weights, biases, conv.root, conv.bias,
                                           residual, activation == "relu", explicit=True)
        if call is not None:
            batch.append(call)
            where.append(k)
        elif residual is not None or activation is not None:
            outs[k] = conv(x, edge_index, edge_attr, residual=residual, activation=activation)
        else:
            outs[k] = conv(x, edge_index, edge_attr)
    for k, o in zip(where, ops.nnconv_forward_edgeweights_group(batch)):
        outs[k] = o
    return outs
