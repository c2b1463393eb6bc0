"""Whole-network autograd node for the masked-GAT auto-encoder.

``GMAEModel.compute_loss`` (``model/autoencoder.py:228-245`` of the reference) runs the
encoder layers, ``torch.cat`` of their outputs, ``encoder_to_decoder`` and the decoder layer.
Driving that through one ``torch.autograd.Function`` per layer leaves the elementwise glue to
torch (dropout, gradient accumulation of tensors used twice, ``cat``, PReLU adjoints ...),
which costs more HBM traffic than the sparse kernels themselves.  This node owns every buffer
of the chain instead:

* layer outputs are written straight into their column block of the ``[N, L*C]`` hidden
  buffer (no ``cat``), the next layer's projection reads that block in place;
* dropout (``model/gat.py:283``) is a uint8 keep mask applied where the operand is loaded
  (projection input, residual input, weight-gradient input) and where the input gradient is
  stored -- the dropped tensor is never materialised;
* in backward, the projection-input GEMM of layer l+1 finishes the gradient of layer l's
  pre-activation in its epilogue (dropout mask, + gradient arriving through the hidden-state
  concatenation, PReLU adjoint, bias / slope gradient sums), so the sparse sweeps of layer l
  start from it directly.

The arithmetic is the one of the per-layer path (``model/gat.py`` in this package); only the
order of floating-point additions of gradient contributions differs.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from .. import _lib, ops


class FusedCfg:
    """Static description of one auto-encoder evaluation (built per call by GMAEModel)."""

    def __init__(self, struct, etype_csc, etype_csr, e_dim, n_layers, n_heads, head_dim, n_dim, dec_dim_pad, slope,
                 res_kinds, masks: List[Optional[torch.Tensor]], drop_scale: float,
                 token_rows: Optional[torch.Tensor] = None, node_types: Optional[torch.Tensor] = None):
        # edge types in CSC order (forward sweep) and in CSR order (backward sweeps)
        self.struct, self.etype, self.etype_csr, self.e_dim = struct, etype_csc, etype_csr, e_dim
        self.L, self.H, self.D, self.C = n_layers, n_heads, head_dim, n_heads * head_dim
        self.n_dim, self.Dp = n_dim, dec_dim_pad
        self.slope = slope
        self.res_kinds = res_kinds          # per encoder layer: "linear" | "identity" | None
        self.masks = masks                  # L encoder masks + 1 decoder mask (uint8 [N,F] or None)
        self.scale = drop_scale
        # rows of x that hold the (learnable) mask token: with them given, the token's gradient is
        # produced directly and the input gradient is never materialised
        self.token_rows = token_rows
        # node types [N] when the layer-0 input is known to be one_hot(type) outside the token rows: the first
        # layer then runs on code words (type, masked, dropout keep bits) and a [n_dim, M] table instead of a
        # projected [N, C] array -- no projection launch, no gathers of 4C-byte rows, no [N, n_dim] operand
        self.node_types = node_types
        # uint8 [N]: the rows of the reconstruction the caller reads (the masked nodes of the loss); the decoder's
        # sweeps then visit the edges into those rows only.  None: every row is produced.
        self.recon_rows = None
        # per encoder layer (attn_h, attn_t) or None: lets the projection of a wide layer form the head scores from
        # its output rows (values only; their gradients come out of the augmented weight's folded rows)
        self.attn = None

    def coded(self, token) -> bool:
        return (self.node_types is not None and token is not None and self.res_kinds[0] == "linear" and self.H == 4
                and self.L >= 2
                and self.n_dim <= _lib.CODED_MAX_DIM and self.D in (8, 16, 32, 64))


def _aligned_rows(n, cols, device):
    """[n, cols] float32 view whose rows start 16-byte aligned."""
    ld = (cols + 3) // 4 * 4
    return torch.empty(n, ld, dtype=torch.float32, device=device)[:, :cols]


class FusedGMAEFn(torch.autograd.Function):
    """inputs: x [N,n_dim], sample_idx [S] (rows of the hidden buffer to gather), cfg, the mask
    token [1,n_dim] (or None; only its gradient is produced here, x already contains it), then per
    encoder layer (w_aug, table, bias, prelu_w), then w_e2d, then decoder (w_aug, table, bias).
    outputs: hidden [N, L*C] (not differentiable from outside), recon [N, Dp], samples [S, L*C]."""

    @staticmethod
    def forward(ctx, x, sample_idx, cfg: FusedCfg, token, *params):
        L, H, D, C = cfg.L, cfg.H, cfg.D, cfg.C
        s = cfg.struct
        N = x.shape[0]
        dev = x.device
        enc = torch.empty(N, L * C, dtype=torch.float32, device=dev)
        inp = x if x.stride(-1) == 1 else x.contiguous()
        x_in = inp
        saved = []
        coded = cfg.coded(token)
        for l in range(L):
            w_aug, table, bias, prelu_w = params[4 * l:4 * l + 4]
            mask = cfg.masks[l]
            kind = cfg.res_kinds[l]
            out_l = enc[:, l * C:(l + 1) * C]
            if l == 0 and coded:
                # every value the projected input row can take, and the row each node maps to
                scale0 = cfg.scale if mask is not None else 1.0
                tok = token.detach().reshape(-1)
                tab = torch.ops.magic.coded_table(w_aug, tok, scale0)
                code = torch.ops.magic.node_code(cfg.node_types, cfg.n_dim, mask, cfg.token_rows)
                _, pre, stat = ops.gat_fwd_coded(s, H, D, ops.Coded(code, tab, tok, scale0), table.contiguous(),
                                                 cfg.etype, cfg.slope, bias, prelu_w, True, out=out_l)
                saved += [code, tab, stat, pre]
                inp = out_l
                continue
            hs, score, res = ops.proj_fwd(inp, w_aug, C, 2 * H, C if kind == "linear" else 0,
                                           x_mask=mask, x_scale=cfg.scale, attn=cfg.attn[l] if cfg.attn else None)
            res_mask = None
            if kind == "identity":
                res, res_mask = inp, mask
            _, pre, _, stat = ops.gat_fwd(s, H, D, hs, score[:, :H], score[:, H:], table.contiguous(), cfg.etype,
                                          None, cfg.slope, bias, res, prelu_w, True, res_mask=res_mask,
                                          res_scale=cfg.scale, out=out_l, want_attn=False, want_stat=True)
            saved += [hs, score, stat, pre]
            inp = out_l
        w_e2d = params[4 * L]
        w_aug_d, table_d, bias_d = params[4 * L + 1:4 * L + 4]
        rep = ops.proj_fwd(enc, w_e2d, w_e2d.shape[0], 0, 0)[0]
        mask_d = cfg.masks[L]
        hs_d, score_d, _ = ops.proj_fwd(rep, w_aug_d, cfg.Dp, 2, 0, x_mask=mask_d, x_scale=cfg.scale)
        recon, _, _, stat_d = ops.gat_fwd(s, 1, cfg.Dp, hs_d, score_d[:, :1], score_d[:, 1:], table_d.contiguous(),
                                          cfg.etype, None, cfg.slope, bias_d, None, None, False, want_attn=False,
                                          want_stat=True, dst_active=cfg.recon_rows)
        samples = enc[sample_idx]
        ctx.cfg = cfg
        ctx.coded = coded
        ctx.token = token.detach() if coded else None
        ctx.set_materialize_grads(False)  # no zero-filled [N, L*C] gradient for the hidden buffer
        ctx.save_for_backward(x_in, sample_idx, enc, rep, hs_d, score_d, stat_d, *saved, *params)
        ctx.mark_non_differentiable(enc)
        return enc, recon, samples

    @staticmethod
    def backward(ctx, _d_enc, d_recon, d_samples):
        cfg = ctx.cfg
        L, H, D, C = cfg.L, cfg.H, cfg.D, cfg.C
        s = cfg.struct
        t = ctx.saved_tensors
        x_in, sample_idx, enc, rep, hs_d, score_d, stat_d = t[:7]
        saved = t[7:7 + 4 * L]
        params = t[7 + 4 * L:]
        N = x_in.shape[0]
        dev = x_in.device
        grads = [None] * len(params)
        w_e2d = params[4 * L]
        w_aug_d = params[4 * L + 1]
        mask_d = cfg.masks[L]
        Dp = cfg.Dp

        # ---- decoder layer (no activation: the incoming gradient is the pre-activation gradient)
        if d_recon is None:
            d_recon = torch.zeros(N, Dp, dtype=torch.float32, device=dev)
        d_recon = d_recon.contiguous()
        d_y = _aligned_rows(N, Dp + 2, dev)
        table_d = params[4 * L + 2].contiguous()
        d_table, _, d_bias, _ = ops.gat_bwd(s, 1, Dp, hs_d, score_d[:, :1], stat_d, table_d, cfg.etype_csr, None,
                                             cfg.e_dim, cfg.slope, None, None, None, d_recon, d_recon, d_y, True,
                                             dst_active=cfg.recon_rows)
        grads[4 * L + 1] = ops.proj_bwd_weight(d_y, rep, x_mask=mask_d, x_scale=cfg.scale)
        grads[4 * L + 2] = d_table
        grads[4 * L + 3] = d_bias
        d_rep = ops.proj_bwd_input(d_y, w_aug_d, out_mask=mask_d, out_scale=cfg.scale)[0]
        # ---- encoder_to_decoder
        grads[4 * L] = ops.proj_bwd_weight(d_rep, enc)
        d_enc = ops.proj_bwd_input(d_rep, w_e2d)[0]                   # [N, L*C]
        if d_samples is not None:
            # rows used by the edge-reconstruction head (stable sort + run heads: reproducible)
            _lib.rows_scatter_add(d_enc, sample_idx, d_samples)

        # ---- encoder layers, last to first.  State handed from layer l+1 to layer l: the finished
        # pre-activation gradient d_pre of layer l, the d_y buffer it may live in, and the bias /
        # PReLU-slope gradients that came out of the same epilogue.
        d_x = d_token = None
        d_pre = d_y_next = bias_grad = prelu_grad = None
        off = C + 2 * H
        for l in range(L - 1, -1, -1):
            w_aug, table, bias, prelu_w = params[4 * l:4 * l + 4]
            hs, score, stat, pre = saved[4 * l:4 * l + 4]
            table = table.contiguous()
            kind = cfg.res_kinds[l]
            mask = cfg.masks[l]
            inp = x_in if l == 0 else enc[:, (l - 1) * C:l * C]
            if l == 0 and ctx.coded:
                # hs / score hold the code words and the table here
                cd = ops.Coded(hs, score, ctx.token, cfg.scale if mask is not None else 1.0)
                d_y = d_y_next      # (its residual columns hold d_pre, finished by layer 1's input-gradient epilogue)
                d_table = ops.gat_bwd_coded(s, H, D, cd, stat, table, cfg.etype_csr, cfg.e_dim, cfg.slope, None,
                                            None, None, d_pre, d_pre, d_y, False)[0]
                grads[0], d_token = ops.coded_weight_grad(cd, d_y, w_aug)
                grads[1] = d_table
                grads[2] = bias_grad
                grads[3] = prelu_grad.view_as(prelu_w)
                break
            if l == L - 1:
                # gradient arrives only through the hidden-state concatenation: the destination
                # sweep applies the PReLU adjoint and sums the bias / slope gradients itself
                d_y = _aligned_rows(N, w_aug.shape[0], dev)
                d_pre = d_y[:, off:] if kind == "linear" else torch.empty(N, C, dtype=torch.float32, device=dev)
                d_table, _, bias_grad, prelu_grad = ops.gat_bwd(
                    s, H, D, hs, score[:, :H], stat, table, cfg.etype_csr, None, cfg.e_dim, cfg.slope, pre,
                    enc[:, l * C:(l + 1) * C], prelu_w, d_enc[:, l * C:(l + 1) * C], d_pre, d_y, True)
            else:
                d_y = d_y_next if d_y_next is not None else _aligned_rows(N, w_aug.shape[0], dev)
                d_table = ops.gat_bwd(s, H, D, hs, score[:, :H], stat, table, cfg.etype_csr, None, cfg.e_dim,
                                       cfg.slope, None, None, None, d_pre, d_pre, d_y, False)[0]
            token_path = (l == 0 and cfg.token_rows is not None and ctx.needs_input_grad[3]
                          and not ctx.needs_input_grad[0] and kind != "identity")
            if not token_path:  # (the token path below produces this layer's weight gradient itself)
                grads[4 * l] = ops.proj_bwd_weight(d_y, inp, x_mask=mask, x_scale=cfg.scale)
            grads[4 * l + 1] = d_table
            grads[4 * l + 2] = bias_grad
            grads[4 * l + 3] = prelu_grad.view_as(prelu_w)
            add = d_pre if kind == "identity" else None
            if l > 0:
                # input gradient of layer l, finished into layer l-1's pre-activation gradient
                d_y_next = target = None
                if cfg.res_kinds[l - 1] == "linear":  # lands in the residual columns of layer l-1's d_y
                    d_y_next = _aligned_rows(N, params[4 * (l - 1)].shape[0], dev)
                    target = d_y_next[:, off:]
                d_pre, bias_grad, prelu_grad = ops.proj_bwd_input(
                    d_y, w_aug, add=add, out_mask=mask, out_scale=cfg.scale, add2=d_enc[:, (l - 1) * C:l * C],
                    act_out=inp, act_pre=saved[4 * (l - 1) + 3], prelu_w=params[4 * (l - 1) + 3],
                    want_col_sum=True, out=target)
            else:
                want_token = cfg.token_rows is not None and ctx.needs_input_grad[3]
                if want_token and not ctx.needs_input_grad[0] and add is None:
                    # d_token[c] = sum over token rows of d_x[:, c] without forming d_x: append the
                    # row indicator (broadcast over the n_dim columns, same dropout mask) to the
                    # weight-gradient operand, G = d_y^T (ind * mask); d_token = colsum(G * W)
                    F0 = inp.shape[1]
                    ind = torch.zeros(N, 1, dtype=torch.float32, device=dev)
                    ind.index_fill_(0, cfg.token_rows, 1.0)
                    x2 = torch.cat([inp, ind.expand(N, F0)], 1)
                    m2 = torch.cat([mask, mask], 1) if mask is not None else None
                    dw2 = ops.proj_bwd_weight(d_y, x2, x_mask=m2, x_scale=cfg.scale)
                    grads[0] = dw2[:, :F0]
                    d_token = (dw2[:, F0:] * w_aug).sum(0, keepdim=True)
                elif ctx.needs_input_grad[0] or want_token:
                    d_x = ops.proj_bwd_input(d_y, w_aug, add=add)[0]
                    if mask is not None:  # [N, n_dim] is narrow: not worth an epilogue variant
                        d_x = d_x * mask * cfg.scale
                    if want_token:
                        d_token = d_x[cfg.token_rows].sum(0, keepdim=True)
                    if not ctx.needs_input_grad[0]:
                        d_x = None
        return (d_x, None, None, d_token, *grads)
