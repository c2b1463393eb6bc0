"""Graph masked auto-encoder with the reference's API (``model/autoencoder.py``).

Same class, attribute and parameter names, construction order and method signatures as the
reference, so ``train.py`` / ``eval.py`` and reference checkpoints work unchanged.  The GAT
layers and the masked SCE loss run on the HIP kernels; the small edge-reconstruction MLP
(``model/autoencoder.py:248-266``) stays plain torch on the GPU.

Randomness is injectable so that a run can be compared step by step with a CPU run of the
reference recipe: ``compute_loss(g, perm=..., pos_idx=..., neg_pairs=...)``.  Without
overrides the behaviour follows the reference: ``torch.randperm`` (drawn from the CPU
generator when ``mask_rng='cpu'``, the default, so that masks equal the reference's
CPU run bit for bit; ``'device'`` draws on the GPU), ``random.sample`` for positive edges,
and a uniform non-edge sampler in place of ``dgl.sampling.global_uniform_negative_sampling``.
"""
from __future__ import annotations

import os
import random
from functools import partial
from itertools import chain

import torch
import torch.nn as nn

from .. import _lib, ops  # noqa: F401  (ops registers torch.ops.magic.*)
from ..graph import Graph
from ..utils.utils import create_norm, sample_range
from .fused import FusedCfg, FusedGMAEFn
from .gat import GAT, _LinearFn, _pad_head_dim, fold_layers, linear_nobias
from .loss_func import sce_loss, sce_loss_masked


def build_model(args):
    """``model/autoencoder.py:18-52``."""
    return GMAEModel(
        n_dim=args.n_dim, e_dim=args.e_dim, hidden_dim=args.num_hidden, n_layers=args.num_layers,
        n_heads=4, activation="prelu", feat_drop=0.1, negative_slope=args.negative_slope, residual=True,
        mask_rate=args.mask_rate, norm="BatchNorm", loss_fn="sce", alpha_l=args.alpha_l)


def _sorted_edge_keys(g: Graph) -> torch.Tensor:
    """src * N + dst of every edge, ascending (cached on the structure: a graph's edges never change)."""
    s = g.structure
    keys = getattr(s, "_edge_keys", None)
    if keys is None:
        src, dst = g.edges()
        keys = s._edge_keys = torch.sort(src * g.num_nodes() + dst).values
    return keys


def negative_candidate_count(num_samples: int) -> int:
    """Candidates drawn for ``num_samples`` negative pairs: a quarter more, as DGL's sampler oversamples
    (its ``redundancy``) so that rejected and repeated draws rarely leave it short."""
    return num_samples + (num_samples + 3) // 4


def _accept_negatives(g: Graph, u: torch.Tensor, v: torch.Tensor, num_samples: int) -> torch.Tensor:
    """Mask that accepts, in order, the first ``num_samples`` candidate pairs that are no edge of ``g``, no
    self-loop and no repeat of an earlier candidate -- with tensor operations (any device)."""
    n = g.num_nodes()
    m = u.numel()
    ekey = _sorted_edge_keys(g)
    key = u * n + v
    pos = torch.searchsorted(ekey, key).clamp_(max=max(ekey.numel() - 1, 0))
    is_edge = (ekey[pos] == key) if ekey.numel() else torch.zeros_like(key, dtype=torch.bool)
    ok = (u != v) & ~is_edge
    # repeats: stable sort by key, every element but the first of a run of equal keys is a repeat
    skey, order = torch.sort(key, stable=True)
    repeat_sorted = torch.zeros(m, dtype=torch.bool, device=u.device)
    repeat_sorted[1:] = skey[1:] == skey[:-1]
    repeat = torch.empty_like(repeat_sorted)
    repeat[order] = repeat_sorted
    ok &= ~repeat
    ok &= torch.cumsum(ok, 0) <= num_samples
    return ok


def _negative_candidates(g: Graph, num_samples: int, generator: torch.Generator = None):
    """Candidate (src, dst) pairs for ``dgl.sampling.global_uniform_negative_sampling(g, num_samples)``
    (``model/autoencoder.py:251``) with fixed shapes, so that nothing is read back to the host:
    ``negative_candidate_count(num_samples)`` uniformly drawn pairs and a mask that accepts, in draw
    order, the first ``num_samples`` of them that are no edge of ``g``, no self-loop and no repeat of an
    earlier accepted pair (DGL samples without replacement).  On the GPU the mask is one call of the
    library (``magic_negative_filter``: segment scans and an LDS table instead of an edge-key sort, a
    search and a sort of the candidates)."""
    n = g.num_nodes()
    dev = g.device
    m = negative_candidate_count(num_samples)
    u = torch.randint(0, n, (m,), device=dev, generator=generator)
    v = torch.randint(0, n, (m,), device=dev, generator=generator)
    if dev.type == "cuda" and m <= _lib.NEGATIVE_FILTER_MAX:
        s = g.structure.ensure_csr()
        return u, v, torch.ops.magic.negative_filter(s.out_ptr, s.out_dst, s.in_ptr, s.in_src, u, v, num_samples)
    return u, v, _accept_negatives(g, u, v, num_samples)


def global_uniform_negative_sampling(g: Graph, num_samples: int, generator: torch.Generator = None):
    """At most ``num_samples`` uniformly drawn (src, dst) pairs that are neither edges of ``g``
    nor self-loops -- the contract of ``dgl.sampling.global_uniform_negative_sampling(g, k)``
    as called at ``model/autoencoder.py:251`` (one redraw-free trial, so fewer than k pairs may
    come back).  DGL draws from a private RNG the reference never seeds, so its exact sequence
    cannot be matched; parity tests inject the pairs instead."""
    u, v, ok = _negative_candidates(g, num_samples, generator)
    return u[ok], v[ok]


class _EdgeHeadLossFn(torch.autograd.Function):
    """``BCELoss(Sigmoid(Linear(LeakyReLU(h + b1))), [1]*n_pos + [0]*n_neg)`` with the negative pairs weighted
    by ``neg_valid`` -- the tail of the edge-reconstruction loss (``model/autoencoder.py:257-266`` through the
    head of ``:114-119``) as two passes over ``h`` (``magic_edge_head_fwd`` / ``_bwd``) instead of ~25 small
    tensor operations.  ``h`` is the first layer's output without its bias ``b1``."""

    @staticmethod
    def forward(ctx, h, b1, w2, b2, neg_valid, n_pos, slope):
        loss, d_logit, wsum = torch.ops.magic.edge_head_fwd(h, b1, w2, b2, neg_valid, int(n_pos), float(slope))
        ctx.save_for_backward(h, b1, w2, d_logit, wsum)
        ctx.slope = float(slope)
        ctx.w2_shape, ctx.b2_shape = w2.shape, b2.shape
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        h, b1, w2, d_logit, wsum = ctx.saved_tensors
        d_h, d_b1, d_w2, d_b2 = torch.ops.magic.edge_head_bwd(h, b1, w2, d_logit, wsum, g.contiguous().float(),
                                                               ctx.slope)
        return d_h, (d_b1.view(b1.shape) if b1 is not None else None), d_w2.view(ctx.w2_shape), \
            d_b2.view(ctx.b2_shape), None, None, None


class GMAEModel(nn.Module):
    """``model/autoencoder.py:55-305``."""

    def __init__(self, n_dim, e_dim, hidden_dim, n_layers, n_heads, activation, feat_drop, negative_slope,
                 residual, norm, mask_rate=0.5, loss_fn="sce", alpha_l=2, mask_rng="cpu", alias_clone=True):
        super().__init__()
        self.alias_clone = alias_clone
        # run the first encoder layer on code words (node type, masked, dropout bits) when the input is known to
        # be one-hot node types; MAGIC_CODED_INPUT=0 (or this attribute) keeps the dense projection instead
        self.coded_input = os.environ.get("MAGIC_CODED_INPUT", "1") != "0"
        # decoder sweeps restricted to the rows the loss reads (MAGIC_DECODER_ROWS=0: every row, as the reference)
        self.decoder_rows = os.environ.get("MAGIC_DECODER_ROWS", "1") != "0"
        # (a smaller graph's node rows are served from the caches whichever edges are visited: nothing to gain there)
        self.decoder_rows_min_nodes = 200_000
        self._mask_rate = mask_rate
        self._output_hidden_size = hidden_dim
        self._alpha_l = alpha_l
        self.mask_rng = mask_rng
        self.recon_loss = nn.BCELoss(reduction="mean")

        def init_weights(m):
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                nn.init.constant_(m.bias, 0)

        self.edge_recon_fc = nn.Sequential(
            nn.Linear(hidden_dim * n_layers * 2, hidden_dim), nn.LeakyReLU(negative_slope),
            nn.Linear(hidden_dim, 1), nn.Sigmoid())
        self.edge_recon_fc.apply(init_weights)
        assert hidden_dim % n_heads == 0
        enc_num_hidden = hidden_dim // n_heads
        self.encoder = GAT(n_dim=n_dim, e_dim=e_dim, hidden_dim=enc_num_hidden, out_dim=enc_num_hidden,
                           n_layers=n_layers, n_heads=n_heads, n_heads_out=n_heads, concat_out=True,
                           activation=activation, feat_drop=feat_drop, attn_drop=0.0,
                           negative_slope=negative_slope, residual=residual, norm=create_norm(norm),
                           encoding=True)
        self.decoder = GAT(n_dim=hidden_dim, e_dim=e_dim, hidden_dim=hidden_dim, out_dim=n_dim, n_layers=1,
                           n_heads=n_heads, n_heads_out=1, concat_out=True, activation=activation,
                           feat_drop=feat_drop, attn_drop=0.0, negative_slope=negative_slope,
                           residual=residual, norm=create_norm(norm), encoding=False)
        self.enc_mask_token = nn.Parameter(torch.zeros(1, n_dim))
        self.encoder_to_decoder = nn.Linear(hidden_dim * n_layers, hidden_dim, bias=False)
        self.criterion = self.setup_loss_fn(loss_fn, alpha_l)

    @property
    def output_hidden_dim(self):
        return self._output_hidden_size

    def setup_loss_fn(self, loss_fn, alpha_l):
        if loss_fn == "sce":
            return partial(sce_loss, alpha=alpha_l)
        raise NotImplementedError

    def _randperm(self, n, device):
        if self.mask_rng == "cpu":
            # CPU generator: same stream as a CPU reference run.  Drawn into pinned memory and copied
            # asynchronously: a pageable copy would make the host wait for everything queued so far
            pin = torch.device(device).type == "cuda"
            return torch.randperm(n, pin_memory=pin).to(device, non_blocking=True)
        return torch.randperm(n, device=device)        # model/autoencoder.py:201 on the GPU

    def encoding_mask_noise(self, g: Graph, mask_rate=0.3, perm=None):
        """``model/autoencoder.py:188-212``.  ``g.clone()`` shares the feature tensors, so the
        token write below also lands in ``g.ndata['attr']`` -- DGL 1.0 behaviour the reference's
        loss depends on (its SCE target at ``:242`` is read after this write)."""
        new_g = g.clone()
        if not self.alias_clone:  # paper-intended variant: keep the original rows as SCE target
            new_g.ndata["attr"] = new_g.ndata["attr"].clone()
        num_nodes = g.num_nodes()
        if perm is None:
            perm = self._randperm(num_nodes, g.device)
        num_mask_nodes = int(mask_rate * num_nodes)
        mask_nodes = perm[:num_mask_nodes]
        keep_nodes = perm[num_mask_nodes:]
        new_g.ndata["attr"][mask_nodes] = self.enc_mask_token
        return new_g, (mask_nodes, keep_nodes)

    def forward(self, g, **overrides):
        return self.compute_loss(g, **overrides)

    def _dropout_masks(self, widths, n, device, drop_masks, drop_seeds=None):
        """One uint8 keep mask per GAT layer input (``model/gat.py:283``), or Nones when dropout
        is inactive.  ``drop_masks`` (0/1 tensors, any dtype) replaces the Bernoulli draws;
        ``drop_seeds`` (int64 tensor on the GPU, one word per mask) replaces the seeds drawn from
        torch's CPU generator and is read when the kernels run (HIP-graph replays draw new masks)."""
        p = self.encoder.gats[0].feat_drop.p
        if drop_masks is not None:
            assert len(drop_masks) == len(widths)
            return [m.to(device=device, dtype=torch.uint8).contiguous() for m in drop_masks], 1.0 / (1.0 - p)
        if not self.training or p == 0.0:
            return [None] * len(widths), 1.0
        if drop_seeds is not None:
            assert drop_seeds.numel() >= len(widths) and drop_seeds.dtype == torch.int64
            return [_lib.dropout_mask_dev((n, w), 1.0 - p, drop_seeds[i:i + 1], device)
                    for i, w in enumerate(widths)], 1.0 / (1.0 - p)
        # one 63-bit seed per mask from torch's CPU generator (reproducible under torch.manual_seed,
        # no device synchronisation); the masks themselves are drawn on the GPU
        seeds = torch.randint(0, 2 ** 62, (len(widths),)).tolist()
        return [_lib.dropout_mask((n, w), 1.0 - p, s, device) for w, s in zip(widths, seeds)], 1.0 / (1.0 - p)

    def _fused_cfg(self, g: Graph, drop_masks, drop_seeds=None):
        """Description of the whole encoder -> decoder chain for ``FusedGMAEFn``, or None when a
        layer needs the generic per-layer path (norm layers, non-PReLU activations, head widths
        that need padding inside the encoder, dense edge attributes ...)."""
        enc, dec = self.encoder.gats, self.decoder.gats
        etype = g.edge_types_csc()
        if etype is None or len(dec) != 1 or not torch.is_grad_enabled():
            return None
        H, D = enc[0].n_heads, enc[0].out_feat
        if _pad_head_dim(D) != D or (2 * H) % 4 != 0:
            return None
        p = enc[0].feat_drop.p
        res_kinds = []
        for i, layer in enumerate(enc):
            ok = (layer.n_heads == H and layer.out_feat == D and layer.norm is None and layer.concat_out
                  and isinstance(layer.activation, nn.PReLU) and layer.activation.weight.numel() == 1
                  and layer.bias is not None and layer.attn_drop.p == 0.0 and layer.feat_drop.p == p
                  and (i == 0 or layer.src_feat == H * D))
            if not ok:
                return None
            res_kinds.append("linear" if isinstance(layer.res_fc, nn.Linear)
                             else "identity" if isinstance(layer.res_fc, nn.Identity) else None)
        d = dec[0]
        if not (d.n_heads == 1 and d.norm is None and d.activation is None and d.res_fc is None
                and d.bias is not None and d.concat_out and d.attn_drop.p == 0.0 and d.feat_drop.p == p):
            return None
        n = g.num_nodes()
        widths = [enc[0].src_feat] + [H * D] * (len(enc) - 1) + [d.src_feat]
        masks, scale = self._dropout_masks(widths, n, g.device, drop_masks, drop_seeds)
        cfg = FusedCfg(g.structure.ensure_csr(), etype, g.edge_types_csr(), enc[0].edge_feat, len(enc), H, D, d.out_feat,
                       _pad_head_dim(d.out_feat), enc[0].leaky_relu.negative_slope, res_kinds, masks, scale)
        if D == 64 and os.environ.get("MAGIC_PROJ_ATTN", "1") != "0":   # (head width the scores-from-the-output-row
            # shortcut of the projection serves)
            cfg.attn = [(layer.attn_h, layer.attn_t) for layer in enc]
        return cfg

    def _edge_recon_loss(self, pairs, n_pos, n_neg, neg_valid):
        """The BCE of ``model/autoencoder.py:257-266`` on the library: the head's first layer as one projection of
        the pair matrix, the rest as the loss tail of ``csrc/head.hip``.  None when the head is not the one the
        reference builds (``:114-119``) or the operands are not float32 rows on the GPU."""
        fc1, act, fc2, sig = self.edge_recon_fc
        if not (pairs.is_cuda and pairs.dtype == torch.float32 and isinstance(act, nn.LeakyReLU)
                and isinstance(sig, nn.Sigmoid) and isinstance(fc2, nn.Linear) and fc2.out_features == 1
                and fc2.bias is not None and fc1.bias is not None and fc2.in_features % 4 == 0
                and fc2.in_features <= 320 and isinstance(self.recon_loss, nn.BCELoss)
                and self.recon_loss.weight is None and self.recon_loss.reduction == "mean" and n_pos + n_neg > 0):
            return None
        if neg_valid is not None:
            neg_valid = torch.as_tensor(neg_valid).to(pairs.device)
            if neg_valid.dtype not in (torch.bool, torch.uint8):
                neg_valid = neg_valid != 0
        h = linear_nobias(pairs, fc1.weight)                                   # (the bias joins in the tail)
        return _EdgeHeadLossFn.apply(h, fc1.bias, fc2.weight, fc2.bias, neg_valid, n_pos, act.negative_slope)

    def _edge_recon(self, pairs):
        """``edge_recon_fc`` (``model/autoencoder.py:75-78``) with tensor operations, the weight gradients of its
        two linear layers on the streaming dW kernel (a [rows, features] reduction over ~20k rows that the BLAS
        heuristics serve badly)."""
        fc1, act, fc2, sig = self.edge_recon_fc
        h = act(_LinearFn.apply(pairs, fc1.weight, fc1.bias))
        return sig(_LinearFn.apply(h, fc2.weight, fc2.bias))

    def compute_loss(self, g: Graph, perm=None, pos_idx=None, neg_pairs=None, drop_masks=None, return_parts=False,
                     fused=True, neg_valid=None, drop_seeds=None):
        """``model/autoencoder.py:218-267``.  ``fused=False`` forces the layer-by-layer path.
        ``neg_valid`` (bool mask) marks which of the injected ``neg_pairs`` count (the rest get weight 0
        in the BCE, as the candidates this method draws itself); ``drop_seeds``: see ``_dropout_masks``."""
        feat_has_grad = g.ndata["attr"].requires_grad
        # (read before the masking step writes the token into the attribute tensor)
        node_types = g.node_types_if_onehot() if self.coded_input else None
        pre_use_g, (mask_nodes, keep_nodes) = self.encoding_mask_noise(g, self._mask_rate, perm)
        pre_use_x = pre_use_g.ndata["attr"].to(pre_use_g.device)
        use_g = pre_use_g
        # autoencoder.py:248-256: node pairs of the sampled edge reconstruction (drawn first: the
        # fused chain gathers their hidden rows itself)
        threshold = min(10000, g.num_nodes())
        if neg_pairs is None:
            # all k candidates go through the head, the rejected ones with weight 0 in the loss: the
            # same mean over the accepted pairs as selecting them first, without a host round trip
            # for their count
            nu, nv, neg_valid = _negative_candidates(g, threshold)
            neg_pairs = (nu, nv)
        if pos_idx is None:
            pos_idx = sample_range(g.number_of_edges(), threshold)   # = random.sample(range(E), T)
        pos_idx = torch.as_tensor(pos_idx).long()
        if pos_idx.device != g.device:
            if pos_idx.device.type == "cpu" and g.device.type == "cuda":
                pos_idx = pos_idx.pin_memory()
            pos_idx = pos_idx.to(g.device, non_blocking=True)
        src, dst = g.edges()
        pos = (src[pos_idx], dst[pos_idx])
        neg = (neg_pairs[0].to(g.device), neg_pairs[1].to(g.device))
        n_pairs = len(pos[0]) + len(neg[0])
        # (src_0, dst_0, src_1, dst_1, ...): the gathered rows, viewed as [n_pairs, 2 * width], ARE the pair matrix
        # torch.cat([h[src], h[dst]], dim=-1) of autoencoder.py:257-258 -- no second copy of it is made
        sample_idx = torch.cat([pos[0], neg[0], pos[1], neg[1]]).view(2, n_pairs).t().reshape(-1)

        cfg = self._fused_cfg(use_g, drop_masks, drop_seeds) if fused else None
        if cfg is not None:
            # operands of every layer from one launch (and their adjoints from one launch in backward)
            enc_layers = list(self.encoder.gats)
            folded = fold_layers(enc_layers + [self.decoder.gats[0]], [cfg.D] * len(enc_layers) + [cfg.Dp])
            params = []
            for layer, (w_aug, table, bias) in zip(enc_layers, folded):
                params += [w_aug, table, bias, layer.activation.weight]
            params.append(self.encoder_to_decoder.weight)
            params += list(folded[-1])
            if (self.decoder_rows and not return_parts and len(mask_nodes) < g.num_nodes()
                    and g.num_nodes() >= self.decoder_rows_min_nodes):
                # only recon[mask_nodes] is read below (autoencoder.py:242-245): the decoder skips the other rows
                cfg.recon_rows = torch.zeros(g.num_nodes(), dtype=torch.uint8, device=g.device)
                cfg.recon_rows.index_fill_(0, mask_nodes, 1)
            if feat_has_grad:
                enc_rep, recon_p, samples = FusedGMAEFn.apply(pre_use_x, sample_idx, cfg, None, *params)
            else:
                # the features' only differentiable content is the token written into the masked
                # rows: hand the chain the plain values and let it return the token's gradient
                cfg.token_rows = mask_nodes
                cfg.node_types = node_types      # one-hot input: the first layer runs on code words
                enc_rep, recon_p, samples = FusedGMAEFn.apply(pre_use_x.detach(), sample_idx, cfg,
                                                              self.enc_mask_token, *params)
            recon = recon_p[:, :cfg.n_dim]
            all_hidden = [enc_rep[:, i * cfg.C:(i + 1) * cfg.C] for i in range(cfg.L)]
        else:
            if drop_masks is not None:
                raise NotImplementedError("injected dropout masks need the fused chain")
            enc_rep, all_hidden = self.encoder(use_g, pre_use_x, return_hidden=True)
            enc_rep = torch.cat(all_hidden, dim=1)
            rep = linear_nobias(enc_rep, self.encoder_to_decoder.weight)   # autoencoder.py:237
            recon = recon_p = self.decoder(pre_use_g, rep)
            samples = enc_rep[sample_idx]
        # autoencoder.py:242-245: gather of the masked rows fused into the loss kernel
        x_target = g.ndata["attr"]
        # (the loss kernel takes the head-padded reconstruction as it is and leaves zeros in the
        # padding columns of its gradient)
        n_dim = recon.shape[1]
        if x_target.data_ptr() == pre_use_x.data_ptr():
            # shared storage: every masked row of the target IS the mask token
            loss_sce = sce_loss_masked(recon_p, self.enc_mask_token, mask_nodes, self._alpha_l, n_dim=n_dim)
        else:
            loss_sce = sce_loss_masked(recon_p, x_target, mask_nodes, self._alpha_l, n_dim=n_dim)
        loss = loss_sce
        # autoencoder.py:257-266: sampled edge reconstruction
        pairs = samples.view(n_pairs, -1)
        loss_bce = self._edge_recon_loss(pairs, len(pos[0]), len(neg[0]), neg_valid)
        if loss_bce is None:
            y_pred = self._edge_recon(pairs).squeeze(-1)
            y = torch.cat([torch.ones(len(pos[0]), device=g.device), torch.zeros(len(neg[0]), device=g.device)])
            if neg_valid is None:
                loss_bce = self.recon_loss(y_pred, y)
            else:
                w = torch.cat([torch.ones(len(pos[0]), device=g.device),
                               torch.as_tensor(neg_valid).to(g.device).float()])
                loss_bce = nn.functional.binary_cross_entropy(y_pred, y, weight=w, reduction="sum") / w.sum()
        loss = loss + loss_bce
        if return_parts:
            return loss, dict(sce=loss_sce, bce=loss_bce, recon=recon, hidden=all_hidden, mask_nodes=mask_nodes)
        return loss

    def embed(self, g: Graph):
        """``model/autoencoder.py:269-282``."""
        x = g.ndata["attr"].to(g.device)
        return self.encoder(g, x, node_types=g.node_types_if_onehot() if self.coded_input else None)

    @property
    def enc_params(self):
        return self.encoder.parameters()

    @property
    def dec_params(self):
        return chain(*[self.encoder_to_decoder.parameters(), self.decoder.parameters()])
