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

import random
from functools import partial
from itertools import chain

import torch
import torch.nn as nn

from ..graph import Graph
from ..utils.utils import create_norm
from .gat import GAT, linear_nobias
from .loss_func import sce_loss, sce_loss_masked


def build_model(args):
    """``model/autoencoder.py:18-52``."""
    return GMAEModel(
        n_dim=args.n_dim, e_dim=args.e_dim, hidden_dim=args.num_hidden, n_layers=args.num_layers,
        n_heads=4, activation="prelu", feat_drop=0.1, negative_slope=args.negative_slope, residual=True,
        mask_rate=args.mask_rate, norm="BatchNorm", loss_fn="sce", alpha_l=args.alpha_l)


def global_uniform_negative_sampling(g: Graph, num_samples: int, generator: torch.Generator = None):
    """At most ``num_samples`` uniformly drawn (src, dst) pairs that are neither edges of ``g``
    nor self-loops -- the contract of ``dgl.sampling.global_uniform_negative_sampling(g, k)``
    as called at ``model/autoencoder.py:251`` (one redraw-free trial, so fewer than k pairs may
    come back).  DGL draws from a private RNG the reference never seeds, so its exact sequence
    cannot be matched; parity tests inject the pairs instead."""
    n = g.num_nodes()
    dev = g.device
    u = torch.randint(0, n, (num_samples,), device=dev, generator=generator)
    v = torch.randint(0, n, (num_samples,), device=dev, generator=generator)
    src, dst = g.edges()
    ekey = torch.sort(src * n + dst).values
    key = u * n + v
    pos = torch.searchsorted(ekey, key).clamp_(max=max(ekey.numel() - 1, 0))
    is_edge = (ekey[pos] == key) if ekey.numel() else torch.zeros_like(key, dtype=torch.bool)
    ok = (u != v) & ~is_edge
    return u[ok], v[ok]


class GMAEModel(nn.Module):
    """``model/autoencoder.py:55-305``."""

    def __init__(self, n_dim, e_dim, hidden_dim, n_layers, n_heads, activation, feat_drop, negative_slope,
                 residual, norm, mask_rate=0.5, loss_fn="sce", alpha_l=2, mask_rng="cpu", alias_clone=True):
        super().__init__()
        self.alias_clone = alias_clone
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
            return torch.randperm(n).to(device)       # CPU generator: same stream as a CPU reference run
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

    def compute_loss(self, g: Graph, perm=None, pos_idx=None, neg_pairs=None, return_parts=False):
        """``model/autoencoder.py:218-267``."""
        pre_use_g, (mask_nodes, keep_nodes) = self.encoding_mask_noise(g, self._mask_rate, perm)
        pre_use_x = pre_use_g.ndata["attr"].to(pre_use_g.device)
        use_g = pre_use_g
        enc_rep, all_hidden = self.encoder(use_g, pre_use_x, return_hidden=True)
        enc_rep = torch.cat(all_hidden, dim=1)
        rep = linear_nobias(enc_rep, self.encoder_to_decoder.weight)   # autoencoder.py:237
        recon = self.decoder(pre_use_g, rep)
        # autoencoder.py:242-245: gather of the masked rows fused into the loss kernel
        x_target = g.ndata["attr"]
        if x_target.data_ptr() == pre_use_x.data_ptr():
            # shared storage: every masked row of the target IS the mask token
            loss_sce = sce_loss_masked(recon, self.enc_mask_token, mask_nodes, self._alpha_l)
        else:
            loss_sce = sce_loss_masked(recon, x_target, mask_nodes, self._alpha_l)
        loss = loss_sce
        # autoencoder.py:248-266: sampled edge reconstruction
        threshold = min(10000, g.num_nodes())
        if neg_pairs is None:
            neg_pairs = global_uniform_negative_sampling(g, threshold)
        if pos_idx is None:
            pos_idx = random.sample(range(g.number_of_edges()), threshold)
        pos_idx = torch.as_tensor(pos_idx, device=g.device).long()
        src, dst = g.edges()
        pos = (src[pos_idx], dst[pos_idx])
        neg = (neg_pairs[0].to(g.device), neg_pairs[1].to(g.device))
        sample_src = enc_rep[torch.cat([pos[0], neg[0]])]
        sample_dst = enc_rep[torch.cat([pos[1], neg[1]])]
        y_pred = self.edge_recon_fc(torch.cat([sample_src, sample_dst], dim=-1)).squeeze(-1)
        y = torch.cat([torch.ones(len(pos[0]), device=g.device), torch.zeros(len(neg[0]), device=g.device)])
        loss_bce = self.recon_loss(y_pred, y)
        loss = loss + loss_bce
        if return_parts:
            return loss, dict(sce=loss_sce, bce=loss_bce, recon=recon, hidden=all_hidden, mask_nodes=mask_nodes)
        return loss

    def embed(self, g: Graph):
        """``model/autoencoder.py:269-282``."""
        x = g.ndata["attr"].to(g.device)
        return self.encoder(g, x)

    @property
    def enc_params(self):
        return self.encoder.parameters()

    @property
    def dec_params(self):
        return chain(*[self.encoder_to_decoder.parameters(), self.decoder.parameters()])
