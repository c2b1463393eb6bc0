"""GAT / GATConv with the reference's Python API, running on hand-written HIP kernels.

Mirrors ``model/gat.py`` of the reference: same class names, constructor arguments,
parameter names, construction order (so seeds and state dicts carry over) and
``forward`` signatures.  The arithmetic of ``GATConv.forward`` (``model/gat.py:264-340``)
is re-expressed for the MI355X:

* node phase -- ``fc`` (``:284``), the per-head scores ``eh``/``et`` (``:291-292``) and the
  residual projection ``res_fc`` (``:323``) are one matrix-core GEMM (fp32-grade split-bf16 products) against an augmented
  weight ``W_aug = [W_fc ; fold(W_fc, attn_h) ; fold(W_fc, attn_t) ; W_res]`` where
  ``fold(W, a)[h, :] = sum_d a[h, d] * W[h*D + d, :]`` (``eh = (h W^T) . attn_h`` is linear
  in ``h``, so it is a GEMM column).  ``_FoldFn`` builds ``W_aug`` in one launch and maps the
  single ``d W_aug`` back to the gradients of ``fc.weight``, ``attn_h``, ``attn_t`` and
  ``res_fc.weight`` in one launch.
* edge phase -- ``edge_fc`` + ``attn_e`` (``:297-299``) collapse the same way into a
  ``[e_dim, H]`` table indexed by edge type when ``edata['attr']`` is one-hot (it always is
  in MAGIC: ``utils/loaddata.py:259``), so the reference's ``[E, H*D]`` temporary never
  exists; ``:302-335`` (logits, leaky ReLU, edge softmax, aggregation, bias, residual,
  PReLU) are the fused HIP kernels of ``csrc/gat_fwd.hip`` / ``csrc/gat_bwd.hip``.

There is no CPU implementation here: tensors must live on the GPU.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _lib, ops
from ..graph import Graph
from ..utils.utils import create_activation


def expand_as_pair(in_dim):
    """``dgl.utils.expand_as_pair`` for feature sizes (``model/gat.py:177``)."""
    if isinstance(in_dim, tuple):
        return in_dim
    return in_dim, in_dim


def _pad_head_dim(d: int) -> int:
    """Head widths the kernels are compiled for: 4 floats per lane, power-of-two lanes."""
    for p in (4, 8, 16, 32, 64):
        if d <= p:
            return p
    raise _lib.MagicHipError(f"head width {d} > 64 has no compiled kernel")


class _FoldFn(torch.autograd.Function):
    """Parameters of one layer -> (augmented weight, edge-score table, padded flat bias): one
    launch forward, one launch backward (``csrc/fold.hip``)."""

    @staticmethod
    def forward(ctx, w_fc, attn_h, attn_t, w_res, w_efc, attn_e, bias, H, D, Dp):
        c = lambda t: None if t is None else t.contiguous()  # noqa: E731
        w_fc, attn_h, attn_t, w_res, w_efc, attn_e, bias = map(c, (w_fc, attn_h, attn_t, w_res, w_efc, attn_e, bias))
        w_aug, table, bias_p = _lib.fold_fwd(H, D, Dp, w_fc, attn_h, attn_t, w_res, w_efc, attn_e, bias)
        ctx.save_for_backward(w_fc, attn_h, attn_t, w_efc, attn_e)
        ctx.dims = (H, D, Dp, w_res is not None, bias is not None)
        ctx.set_materialize_grads(False)
        if bias_p is None:
            return w_aug, table
        return w_aug, table, bias_p

    @staticmethod
    def backward(ctx, d_w_aug, d_table, d_bias_p=None):
        w_fc, attn_h, attn_t, w_efc, attn_e = ctx.saved_tensors
        H, D, Dp, has_res, has_bias = ctx.dims
        c = lambda t: None if t is None else t.contiguous()  # noqa: E731
        d_w_fc, d_ah, d_at, d_w_res, d_w_efc, d_ae, d_bias = _lib.fold_bwd(
            H, D, Dp, w_fc, attn_h, attn_t, w_efc, attn_e, c(d_w_aug), c(d_table), c(d_bias_p), has_res, has_bias)
        return d_w_fc, d_ah, d_at, d_w_res, d_w_efc, d_ae, d_bias, None, None, None


class _FoldAllFn(torch.autograd.Function):
    """``_FoldFn`` for several layers at once: one launch forward, one backward.  Inputs: ``dims`` = [(H, D,
    Dp), ...] and seven tensors per layer (``w_fc, attn_h, attn_t, w_res | None, w_efc, attn_e, bias | None``);
    outputs: three per layer (``w_aug, table, bias_p | None``)."""

    @staticmethod
    def forward(ctx, dims, *tensors):
        c = lambda t: None if t is None else t.contiguous()  # noqa: E731
        layers = [tuple(d) + tuple(map(c, tensors[7 * i:7 * i + 7])) for i, d in enumerate(dims)]
        outs = _lib.fold_fwd_multi(layers)
        ctx.dims = [(lay[0], lay[1], lay[2], lay[6] is not None, lay[9] is not None) for lay in layers]
        ctx.save_for_backward(*[t for lay in layers for t in (lay[3], lay[4], lay[5], lay[7], lay[8])])
        ctx.set_materialize_grads(False)
        return tuple(t for out in outs for t in out)

    @staticmethod
    def backward(ctx, *grads):
        saved = ctx.saved_tensors
        c = lambda t: None if t is None else t.contiguous()  # noqa: E731
        layers = []
        for i, (H, D, Dp, has_res, has_bias) in enumerate(ctx.dims):
            w_fc, attn_h, attn_t, w_efc, attn_e = saved[5 * i:5 * i + 5]
            # (the residual weight / bias themselves are not read by the adjoint: only whether they exist)
            layers.append((H, D, Dp, w_fc, attn_h, attn_t, w_fc if has_res else None, w_efc, attn_e,
                           attn_h if has_bias else None))
        outs = _lib.fold_bwd_multi(layers, [tuple(map(c, grads[3 * i:3 * i + 3])) for i in range(len(layers))])
        return (None,) + tuple(t for out in outs for t in out)


def fold_layers(layers, pads):
    """(augmented weight, edge score table, flat bias | None) of every ``GATConv`` in ``layers`` (head widths
    padded to ``pads``) from one launch."""
    dims, tensors = [], []
    for layer, Dp in zip(layers, pads):
        dims.append((layer.n_heads, layer.out_feat, Dp))
        res_w = layer.res_fc.weight if isinstance(layer.res_fc, nn.Linear) else None
        tensors += [layer.fc.weight, layer.attn_h, layer.attn_t, res_w, layer.edge_fc.weight, layer.attn_e, layer.bias]
    out = _FoldAllFn.apply(dims, *tensors)
    return [tuple(out[3 * i:3 * i + 3]) for i in range(len(dims))]


class _GATLayerFn(torch.autograd.Function):
    """proj GEMM + fused edge phase; backward = HIP edge-phase backward + two GEMMs."""

    @staticmethod
    def forward(ctx, h, w_aug, edge_term, bias, prelu_w, struct, cfg):
        H, Dp, C = cfg["H"], cfg["Dp"], cfg["C"]
        res_c = cfg["res_c"]
        h = h.contiguous() if h.stride(-1) != 1 else h
        hs, score, res = ops.proj_fwd(h, w_aug, C, 2 * H, res_c, attn=cfg.get("attn"))
        if cfg["res_identity"]:
            res = h
        etype = cfg["etype_csc"]
        if etype is not None:
            table, ee_csc = edge_term.contiguous(), None
        else:
            table, ee_csc = None, edge_term[struct.in_eid.long()].contiguous()
        need_grad = cfg["need_grad"]
        want_pre = prelu_w is not None and need_grad
        out, pre, attn, stat = ops.gat_fwd(struct, H, Dp, hs, score[:, :H], score[:, H:], table, etype, ee_csc,
                                            cfg["slope"], bias, res, prelu_w, want_pre, want_attn=cfg["want_attn"],
                                            want_stat=need_grad)
        ctx.struct, ctx.cfg = struct, cfg
        ctx.save_for_backward(h, w_aug, hs, score, stat, table, ee_csc, pre, prelu_w,
                              out if prelu_w is not None else None)
        if attn is None:
            attn = hs.new_empty(0)
        ctx.mark_non_differentiable(attn)
        return out, attn

    @staticmethod
    def backward(ctx, d_out, _d_attn):
        h, w_aug, hs, score, stat, table, ee_csc, pre, prelu_w, act_out = ctx.saved_tensors
        struct, cfg = ctx.struct, ctx.cfg
        H, Dp, C, res_c = cfg["H"], cfg["Dp"], cfg["C"], cfg["res_c"]
        N = h.shape[0]
        Ctot = w_aug.shape[0]
        ld = (Ctot + 3) // 4 * 4  # float4 row stores need 16-byte aligned rows
        d_y = torch.empty(N, ld, dtype=torch.float32, device=h.device)[:, :Ctot]
        if res_c and (C + 2 * H) % 4 == 0:
            d_pre = d_y[:, C + 2 * H:]
        else:
            d_pre = torch.empty(N, C, dtype=torch.float32, device=h.device)
        dense = table is None
        d_table, rec, d_bias, d_prelu = ops.gat_bwd(
            struct, H, Dp, hs, score[:, :H], stat, table, cfg["etype_csr"], ee_csc, cfg["e_dim"], cfg["slope"], pre,
            act_out, prelu_w, d_out, d_pre, d_y, cfg["has_bias"], want_dlogit=dense and ctx.needs_input_grad[2])
        if res_c and d_pre.data_ptr() != d_y[:, C + 2 * H:].data_ptr():
            d_y[:, C + 2 * H:] = d_pre
        d_h = d_w = d_edge = None
        if ctx.needs_input_grad[0]:
            d_h = ops.proj_bwd_input(d_y, w_aug, add=d_pre if cfg["res_identity"] else None)[0]
        if ctx.needs_input_grad[1]:
            d_w = ops.proj_bwd_weight(d_y, h)
        if ctx.needs_input_grad[2]:
            if not dense:
                d_edge = d_table
            else:
                d_edge = torch.empty(struct.num_edges, H, dtype=torch.float32, device=h.device)
                if struct.num_edges:
                    d_edge[struct.in_eid.long()] = rec
        return d_h, d_w, d_edge, d_bias, d_prelu, None, None


class _LinearNoBiasFn(torch.autograd.Function):
    """y = x W^T on the projection kernels of this library (forward, input and weight gradients)."""

    @staticmethod
    def forward(ctx, x, w):
        y, _, _ = ops.proj_fwd(x, w, w.shape[0], 0, 0)
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = ops.proj_bwd_input(dy, w)[0] if ctx.needs_input_grad[0] else None
        dw = ops.proj_bwd_weight(dy, x) if ctx.needs_input_grad[1] else None
        return dx, dw


class _LinearFn(torch.autograd.Function):
    """``F.linear(x, w, b)`` whose weight gradient (a reduction over the rows of x) runs on the
    streaming dW kernel; forward and input gradient stay on the BLAS."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return torch.nn.functional.linear(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dy @ w if ctx.needs_input_grad[0] else None
        dw = ops.proj_bwd_weight(dy, x) if ctx.needs_input_grad[1] else None
        db = dy.sum(0) if ctx.needs_input_grad[2] else None
        return dx, dw, db


def linear_nobias(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """``torch.nn.functional.linear(x, w)`` for GPU tensors with up to 320 output features."""
    return _LinearNoBiasFn.apply(x, w)


class GATConv(nn.Module):
    """Edge-feature-aware graph attention layer (``model/gat.py:136-340``)."""

    def __init__(self, in_dim, e_dim, out_dim, n_heads, feat_drop=0.0, attn_drop=0.0, negative_slope=0.2,
                 residual=False, activation=None, allow_zero_in_degree=False, bias=True, norm=None,
                 concat_out=True):
        super().__init__()
        self.n_heads = n_heads
        self.src_feat, self.dst_feat = expand_as_pair(in_dim)
        self.edge_feat = e_dim
        self.out_feat = out_dim
        self.allow_zero_in_degree = allow_zero_in_degree
        self.concat_out = concat_out
        if isinstance(in_dim, tuple):
            raise NotImplementedError("bipartite (src, dst) inputs are not used by MAGIC and have no HIP kernel")
        # same construction order as model/gat.py:189-210 (RNG stream and state-dict keys)
        self.fc_node_embedding = nn.Linear(self.src_feat, out_dim * n_heads, bias=False)
        self.fc = nn.Linear(self.src_feat, out_dim * n_heads, bias=False)
        self.edge_fc = nn.Linear(e_dim, out_dim * n_heads, bias=False)
        self.attn_h = nn.Parameter(torch.FloatTensor(size=(1, n_heads, out_dim)))
        self.attn_e = nn.Parameter(torch.FloatTensor(size=(1, n_heads, out_dim)))
        self.attn_t = nn.Parameter(torch.FloatTensor(size=(1, n_heads, out_dim)))
        self.feat_drop = nn.Dropout(feat_drop)
        self.attn_drop = nn.Dropout(attn_drop)
        self.leaky_relu = nn.LeakyReLU(negative_slope)
        if bias:
            self.bias = nn.Parameter(torch.FloatTensor(size=(1, n_heads, out_dim)))
        else:
            self.register_buffer("bias", None)
        if residual:
            if self.dst_feat != n_heads * out_dim:
                self.res_fc = nn.Linear(self.dst_feat, n_heads * out_dim, bias=False)
            else:
                self.res_fc = nn.Identity()
        else:
            self.register_buffer("res_fc", None)
        self.reset_parameters()
        self.activation = activation
        self.norm = norm
        if norm is not None:
            self.norm = norm(n_heads * out_dim)

    def reset_parameters(self):
        """``model/gat.py:227-240``."""
        gain = nn.init.calculate_gain("relu")
        nn.init.xavier_normal_(self.edge_fc.weight, gain=gain)
        nn.init.xavier_normal_(self.fc.weight, gain=gain)
        nn.init.xavier_normal_(self.attn_h, gain=gain)
        nn.init.xavier_normal_(self.attn_e, gain=gain)
        nn.init.xavier_normal_(self.attn_t, gain=gain)
        if self.bias is not None:
            nn.init.constant_(self.bias, 0)
        if isinstance(self.res_fc, nn.Linear):
            nn.init.xavier_normal_(self.res_fc.weight, gain=gain)

    def set_allow_zero_in_degree(self, set_value):
        self.allow_zero_in_degree = set_value

    # ---------------------------------------------------------------------------------
    def _operands(self, Dp: int):
        """(augmented weight, [e_dim, H] edge score table, head-padded flat bias | None)."""
        res_w = self.res_fc.weight if isinstance(self.res_fc, nn.Linear) else None
        out = _FoldFn.apply(self.fc.weight, self.attn_h, self.attn_t, res_w, self.edge_fc.weight, self.attn_e,
                            self.bias, self.n_heads, self.out_feat, Dp)
        return out if len(out) == 3 else (out[0], out[1], None)

    def fused_params(self, Dp: int):
        """(augmented weight, edge score table, flat bias, PReLU slope | None): the differentiable
        operands the whole-network node in ``fused.py`` consumes."""
        w_aug, table, bias = self._operands(Dp)
        prelu_w = self.activation.weight if isinstance(self.activation, nn.PReLU) else None
        return w_aug, table, bias, prelu_w

    def onehot_input_ok(self, graph: Graph) -> bool:
        """True when ``forward_onehot`` serves this layer: inference (no autograd, dropout inactive), the encoder's
        four heads with a linear residual and a fused PReLU, typed edges, at most 13 input columns."""
        return (not torch.is_grad_enabled() and (not self.training or self.feat_drop.p == 0.0)
                and self.n_heads == 4 and _pad_head_dim(self.out_feat) == self.out_feat
                and self.out_feat in (8, 16, 32, 64) and isinstance(self.res_fc, nn.Linear)
                and isinstance(self.activation, nn.PReLU) and self.activation.weight.numel() == 1
                and self.norm is None and self.concat_out and self.bias is not None
                and self.src_feat <= _lib.CODED_MAX_DIM and graph.edge_types_csc() is not None)

    def forward_onehot(self, graph: Graph, node_types: torch.Tensor):
        """``forward(graph, one_hot(node_types))`` without the one-hot matrix, the projection launch or the
        projected [N, C] array: the layer's row of a node is a row of the [n_dim + ..., M] table of its augmented
        weight, fetched by the node's type inside the sweep (``csrc/coded.hip``).  Inference only."""
        H, D = self.n_heads, self.out_feat
        struct = graph.structure.ensure_csr()
        w_aug, table, bias = self._operands(D)
        token = torch.zeros(self.src_feat, dtype=torch.float32, device=w_aug.device)
        tab = torch.ops.magic.coded_table(w_aug, token, 1.0)
        code = torch.ops.magic.node_code(node_types, self.src_feat, None, None)
        out, _, _ = ops.gat_fwd_coded(struct, H, D, ops.Coded(code, tab, token, 1.0), table.contiguous(),
                                      graph.edge_types_csc(), self.leaky_relu.negative_slope, bias,
                                      self.activation.weight, False)
        return out

    def forward(self, graph: Graph, feat, get_attention=False):
        if isinstance(feat, tuple):
            raise NotImplementedError("bipartite (src, dst) inputs are not used by MAGIC and have no HIP kernel")
        if feat.device.type != "cuda":
            raise _lib.MagicHipError("magic_amd.GATConv runs on the MI355X only (no CPU fallback); "
                                     "move the graph and the model to 'cuda'")
        if self.attn_drop.p != 0.0 and self.training:
            raise NotImplementedError("attn_drop > 0 is never used by MAGIC (model/autoencoder.py:143,162)")
        H, D = self.n_heads, self.out_feat
        Dp = _pad_head_dim(D)
        C = H * Dp
        struct = graph.structure.ensure_csr()
        h = self.feat_drop(feat)                                       # gat.py:283
        res_identity = isinstance(self.res_fc, nn.Identity)
        if res_identity and Dp != D:
            raise _lib.MagicHipError("identity residual with a padded head width is not supported")
        # gat.py:284-299, :323 collapsed into GEMM operands
        w_aug, table, bias = self._operands(Dp)
        # edge score term: [e_dim, H] table (one-hot attributes) or dense [E, H]
        etype_csc = graph.edge_types_csc()
        if etype_csc is not None:
            edge_term = table.contiguous()
        else:
            edge_term = graph.edata["attr"].to(torch.float32) @ table
        fuse_act = isinstance(self.activation, nn.PReLU) and self.activation.weight.numel() == 1 \
            and self.norm is None and self.concat_out
        prelu_w = self.activation.weight if fuse_act else None
        need_grad = torch.is_grad_enabled()
        cfg = dict(H=H, Dp=Dp, C=C, res_c=C if isinstance(self.res_fc, nn.Linear) else 0,
                   res_identity=res_identity, etype_csc=etype_csc,
                   etype_csr=graph.edge_types_csr() if need_grad else None, e_dim=self.edge_feat,
                   slope=self.leaky_relu.negative_slope, has_bias=bias is not None, need_grad=need_grad,
                   want_attn=bool(get_attention),
                   attn=(self.attn_h, self.attn_t) if D == 64 and Dp == D else None)
        out, attn = _GATLayerFn.apply(h, w_aug, edge_term, bias, prelu_w, struct, cfg)
        rst = out
        if Dp != D:
            rst = out.view(-1, H, Dp)[:, :, :D].reshape(-1, H * D)
        if not self.concat_out:                                        # gat.py:328-329
            rst = rst.reshape(-1, H, D).mean(dim=1)
        if self.norm is not None:                                      # gat.py:331-332
            rst = self.norm(rst)
        if self.activation is not None and not fuse_act:               # gat.py:334-335
            rst = self.activation(rst)
        if get_attention:
            a = torch.empty_like(attn)
            a[struct.in_eid.long()] = attn.abs()
            return rst, a.unsqueeze(-1)
        return rst


class GAT(nn.Module):
    """Multi-layer GAT (``model/gat.py:23-133``)."""

    def __init__(self, n_dim, e_dim, hidden_dim, out_dim, n_layers, n_heads, n_heads_out, activation,
                 feat_drop, attn_drop, negative_slope, residual, norm, concat_out=False, encoding=False):
        super().__init__()
        self.out_dim = out_dim
        self.n_heads = n_heads
        self.n_layers = n_layers
        self.gats = nn.ModuleList()
        self.concat_out = concat_out
        last_activation = create_activation(activation) if encoding else None
        last_residual = (encoding and residual)
        last_norm = norm if encoding else None
        if self.n_layers == 1:
            self.gats.append(GATConv(n_dim, e_dim, out_dim, n_heads_out, feat_drop, attn_drop, negative_slope,
                                     last_residual, norm=last_norm, concat_out=self.concat_out))
        else:
            self.gats.append(GATConv(n_dim, e_dim, hidden_dim, n_heads, feat_drop, attn_drop, negative_slope,
                                     residual, create_activation(activation), norm=norm,
                                     concat_out=self.concat_out))
            for _ in range(1, self.n_layers - 1):
                self.gats.append(GATConv(hidden_dim * self.n_heads, e_dim, hidden_dim, n_heads, feat_drop,
                                         attn_drop, negative_slope, residual, create_activation(activation),
                                         norm=norm, concat_out=self.concat_out))
            self.gats.append(GATConv(hidden_dim * self.n_heads, e_dim, out_dim, n_heads_out, feat_drop,
                                     attn_drop, negative_slope, last_residual, last_activation, norm=last_norm,
                                     concat_out=self.concat_out))
        self.head = nn.Identity()

    def forward(self, g, input_feature, return_hidden=False, node_types=None):
        """``node_types``: the caller's word that ``input_feature`` is ``one_hot(node_types)``; the first layer then
        reads the types instead of the matrix where it can (inference)."""
        h = input_feature
        hidden_list = []
        for layer in range(self.n_layers):
            if layer == 0 and node_types is not None and self.n_layers > 1 and self.gats[0].onehot_input_ok(g):
                h = self.gats[0].forward_onehot(g, node_types)
            else:
                h = self.gats[layer](g, h)
            hidden_list.append(h)
        if return_hidden:
            return self.head(h), hidden_list
        return self.head(h)

    def reset_classifier(self, num_classes):
        self.head = nn.Linear(self.n_heads * self.out_dim, num_classes)
