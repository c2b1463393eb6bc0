"""Torch-CPU restatement of MAGIC's masked-GAT auto-encoder.  TEST INFRASTRUCTURE ONLY.

Parity unpinned: the arithmetic restated here is DGL 1.0.0's (g-SDDMM, edge_softmax,
g-SpMM), which cannot be imported in this pipeline; every function cites the
reference line it follows.  Works in fp32 and fp64 (``model.double()``); the fp64
run is the tolerance anchor for the fp32 HIP kernels.

Module construction order and parameter names equal the reference's, so a
reference ``state_dict`` loads and ``torch.manual_seed`` produces the same initial
weights (``model/gat.py:175-240``, ``model/autoencoder.py:100-175``).
"""
from __future__ import annotations

import random
from functools import partial
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------- graph
class OGraph:
    """Plain COO graph.  ``clone()`` shares feature tensors with the original, which is
    what DGL 1.0 ``DGLGraph.clone`` does (its ``Frame.clone`` docstring: "they still
    share the tensor contents so any mutable operation on the column tensor are
    visible to each other")."""

    is_block = False

    def __init__(self, src, dst, num_nodes):
        self.src = torch.as_tensor(src).long()
        self.dst = torch.as_tensor(dst).long()
        self.n = int(num_nodes)
        self.ndata = {}
        self.edata = {}

    def num_nodes(self):
        return self.n

    def number_of_edges(self):
        return int(self.src.numel())

    def edges(self):
        return self.src, self.dst

    def clone(self, alias=True):
        g = OGraph(self.src, self.dst, self.n)
        if alias:
            g.ndata, g.edata = dict(self.ndata), dict(self.edata)
        else:
            g.ndata = {k: v.clone() for k, v in self.ndata.items()}
            g.edata = {k: v.clone() for k, v in self.edata.items()}
        return g


def obatch(graphs):
    """``dgl.batch`` (``model/train.py:39``): concatenate with node-id offsets."""
    off, srcs, dsts = 0, [], []
    for g in graphs:
        srcs.append(g.src + off)
        dsts.append(g.dst + off)
        off += g.n
    out = OGraph(torch.cat(srcs), torch.cat(dsts), off)
    for k in graphs[0].ndata:
        out.ndata[k] = torch.cat([g.ndata[k] for g in graphs], 0)
    for k in graphs[0].edata:
        out.edata[k] = torch.cat([g.edata[k] for g in graphs], 0)
    return out


def transform_graph(g: OGraph, n_dim: int, e_dim: int) -> OGraph:
    """``utils/loaddata.py:257-260``: one-hot node / edge attributes from the type ids."""
    new_g = g.clone()
    new_g.ndata["attr"] = F.one_hot(g.ndata["type"].view(-1), num_classes=n_dim).float()
    new_g.edata["attr"] = F.one_hot(g.edata["type"].view(-1), num_classes=e_dim).float()
    return new_g


# ----------------------------------------------------------------------------- DGL ops
def edge_softmax(dst: torch.Tensor, e: torch.Tensor, n: int) -> torch.Tensor:
    """``dgl.ops.edge_softmax(graph, e)`` with norm_by='dst' (``model/gat.py:308``):
    per destination node and head, ``exp(e - max) / sum(exp(e - max))``."""
    idx = dst.view(-1, *([1] * (e.dim() - 1))).expand_as(e)
    mx = torch.full((n,) + e.shape[1:], float("-inf"), dtype=e.dtype)
    mx = mx.scatter_reduce(0, idx, e, reduce="amax", include_self=True)
    ex = torch.exp(e - mx[dst])
    sm = torch.zeros((n,) + e.shape[1:], dtype=e.dtype).index_add_(0, dst, ex)
    return ex / sm[dst]


def u_mul_e_sum(src, dst, hs, a, n):
    """``update_all(fn.u_mul_e('hs','a','m'), fn.sum('m','hs'))`` (``model/gat.py:311-312``);
    nodes without in-edges get zeros."""
    m = hs[src] * a
    return torch.zeros((n,) + hs.shape[1:], dtype=hs.dtype).index_add_(0, dst, m)


# ----------------------------------------------------------------------------- utils
def create_activation(name):
    """``utils/utils.py:77-101``."""
    if name == "relu":
        return nn.ReLU()
    if name == "gelu":
        return nn.GELU()
    if name == "prelu":
        return nn.PReLU()
    if name is None:
        return nn.Identity()
    if name == "elu":
        return nn.ELU()
    raise NotImplementedError(f"{name} is not implemented.")


def create_norm(name):
    """``utils/utils.py:104-121``: only lower-case names match, so the model's
    ``norm='BatchNorm'`` (``model/autoencoder.py:48``) yields no normalisation."""
    if name == "layernorm":
        return nn.LayerNorm
    if name == "batchnorm":
        return nn.BatchNorm1d
    if name == "graphnorm":
        raise NotImplementedError("graphnorm is not on the hot path")
    return None


def sce_loss(x, y, alpha=3):
    """``model/loss_func.py:28-31``."""
    x = F.normalize(x, p=2, dim=-1)
    y = F.normalize(y, p=2, dim=-1)
    loss = (1 - (x * y).sum(dim=-1)).pow_(alpha)
    return loss.mean()


# ----------------------------------------------------------------------------- model
class GATConv(nn.Module):
    """``model/gat.py:136-340`` (non-tuple ``in_dim`` branch, the only one MAGIC uses)."""

    def __init__(self, in_dim, e_dim, out_dim, n_heads, feat_drop=0.0, attn_drop=0.0,
                 negative_slope=0.2, residual=False, activation=None, allow_zero_in_degree=False,
                 bias=True, norm=None, concat_out=True):
        super().__init__()
        self.n_heads = n_heads
        self.src_feat = self.dst_feat = in_dim
        self.edge_feat = e_dim
        self.out_feat = out_dim
        self.allow_zero_in_degree = allow_zero_in_degree
        self.concat_out = concat_out
        # gat.py:189-192 (fc_node_embedding is never used in forward but is constructed,
        # consumes RNG and appears in the state dict)
        self.fc_node_embedding = nn.Linear(in_dim, out_dim * n_heads, bias=False)
        self.fc = nn.Linear(in_dim, out_dim * n_heads, bias=False)
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

    def forward(self, graph: OGraph, feat, get_attention=False, drop_mask=None):
        """``model/gat.py:264-340``.  ``drop_mask`` (0/1 keep mask, already scaled or not:
        it is multiplied by 1/(1-p)) replaces the Bernoulli draw of ``feat_drop`` so that
        a device run can be compared under identical dropout."""
        src, dst, n = graph.src, graph.dst, graph.n
        edge_feature = graph.edata["attr"]
        H, D = self.n_heads, self.out_feat
        if drop_mask is not None:
            h_src = h_dst = feat * drop_mask * (1.0 / (1.0 - self.feat_drop.p))
        else:
            h_src = h_dst = self.feat_drop(feat)                        # :283
        feat_src = feat_dst = self.fc(h_src).view(-1, H, D)             # :284
        eh = (feat_src * self.attn_h).sum(-1).unsqueeze(-1)             # :291
        et = (feat_dst * self.attn_t).sum(-1).unsqueeze(-1)             # :292
        feat_edge = self.edge_fc(edge_feature).view(-1, H, D)           # :297
        ee = (feat_edge * self.attn_e).sum(-1).unsqueeze(-1)            # :299
        e = eh[src] + ee                                                # :302 u_add_e
        e = e + et[dst]                                                 # :303 e_add_v
        e = self.leaky_relu(e)                                          # :307
        a = self.attn_drop(edge_softmax(dst, e, n))                     # :308
        rst = u_mul_e_sum(src, dst, feat_src, a, n).view(-1, H, D)      # :311-314
        if self.bias is not None:
            rst = rst + self.bias.view(1, H, D)                         # :317
        if self.res_fc is not None:
            resval = self.res_fc(h_dst).view(h_dst.shape[0], -1, D)     # :323
            rst = rst + resval
        if self.concat_out:
            rst = rst.flatten(1)                                        # :327
        else:
            rst = torch.mean(rst, dim=1)
        if self.norm is not None:
            rst = self.norm(rst)
        if self.activation:
            rst = self.activation(rst)                                  # :335
        if get_attention:
            return rst, a
        return rst


class GAT(nn.Module):
    """``model/gat.py:23-124``."""

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
        if n_layers == 1:
            self.gats.append(GATConv(n_dim, e_dim, out_dim, n_heads_out, feat_drop, attn_drop,
                                     negative_slope, last_residual, norm=last_norm, concat_out=concat_out))
        else:
            self.gats.append(GATConv(n_dim, e_dim, hidden_dim, n_heads, feat_drop, attn_drop, negative_slope,
                                     residual, create_activation(activation), norm=norm, concat_out=concat_out))
            for _ in range(1, n_layers - 1):
                self.gats.append(GATConv(hidden_dim * n_heads, e_dim, hidden_dim, n_heads, feat_drop,
                                         attn_drop, negative_slope, residual, create_activation(activation),
                                         norm=norm, concat_out=concat_out))
            self.gats.append(GATConv(hidden_dim * n_heads, e_dim, out_dim, n_heads_out, feat_drop, attn_drop,
                                     negative_slope, last_residual, last_activation, norm=last_norm,
                                     concat_out=concat_out))
        self.head = nn.Identity()

    def forward(self, g, input_feature, return_hidden=False, drop_masks=None):
        h = input_feature
        hidden_list = []
        for layer in range(self.n_layers):
            dm = None if drop_masks is None else drop_masks[layer]
            h = self.gats[layer](g, h, drop_mask=dm)
            hidden_list.append(h)
        if return_hidden:
            return self.head(h), hidden_list
        return self.head(h)


class GMAEModel(nn.Module):
    """``model/autoencoder.py:55-305``.

    Extra keyword hooks of ``compute_loss`` (all default to the reference behaviour)
    let a test replay the exact random choices on another device:
    ``perm`` (node permutation instead of ``torch.randperm``), ``pos_idx`` (positive
    edge ids instead of ``random.sample``), ``neg_pairs`` (negative (src, dst) instead
    of DGL's sampler, whose private RNG the reference never seeds), ``drop_masks``.
    ``alias_clone=True`` reproduces DGL's shared feature storage (the SCE target then
    is the mask token itself); ``False`` keeps the original one-hot rows as target.
    """

    def __init__(self, n_dim, e_dim, hidden_dim, n_layers, n_heads, activation, feat_drop, negative_slope,
                 residual, norm, mask_rate=0.5, loss_fn="sce", alpha_l=2, alias_clone=True):
        super().__init__()
        self._mask_rate = mask_rate
        self._output_hidden_size = hidden_dim
        self.alias_clone = alias_clone
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
        self.criterion = partial(sce_loss, alpha=alpha_l)

    @property
    def output_hidden_dim(self):
        return self._output_hidden_size

    def encoding_mask_noise(self, g: OGraph, mask_rate=0.3, perm=None):
        """``model/autoencoder.py:198-212``."""
        new_g = g.clone(alias=self.alias_clone)
        num_nodes = g.num_nodes()
        if perm is None:
            perm = torch.randperm(num_nodes)
        num_mask_nodes = int(mask_rate * num_nodes)
        mask_nodes = perm[:num_mask_nodes]
        keep_nodes = perm[num_mask_nodes:]
        # in-place on the column tensor: with alias=True this also rewrites g.ndata['attr']
        x = new_g.ndata["attr"]
        x[mask_nodes] = self.enc_mask_token.to(x.dtype)
        return new_g, (mask_nodes, keep_nodes)

    def forward(self, g, **kw):
        return self.compute_loss(g, **kw)

    def compute_loss(self, g: OGraph, perm=None, pos_idx=None, neg_pairs=None, drop_masks=None,
                     return_parts=False):
        """``model/autoencoder.py:228-267``."""
        pre_use_g, (mask_nodes, keep_nodes) = self.encoding_mask_noise(g, self._mask_rate, perm)
        pre_use_x = pre_use_g.ndata["attr"]
        enc_dm = None if drop_masks is None else drop_masks[:-1]
        dec_dm = None if drop_masks is None else drop_masks[-1:]
        enc_rep, all_hidden = self.encoder(pre_use_g, pre_use_x, return_hidden=True, drop_masks=enc_dm)
        enc_rep = torch.cat(all_hidden, dim=1)
        rep = self.encoder_to_decoder(enc_rep)
        recon = self.decoder(pre_use_g, rep, drop_masks=dec_dm)
        x_init = g.ndata["attr"][mask_nodes]
        x_rec = recon[mask_nodes]
        loss_sce = self.criterion(x_rec, x_init)
        threshold = min(10000, g.num_nodes())
        if neg_pairs is None:
            neg_pairs = uniform_negative_pairs(g, threshold)
        if pos_idx is None:
            pos_idx = random.sample(range(g.number_of_edges()), threshold)
        pos_idx = torch.as_tensor(pos_idx).long()
        pos = (g.src[pos_idx], g.dst[pos_idx])
        sample_src = enc_rep[torch.cat([pos[0], neg_pairs[0]])]
        sample_dst = enc_rep[torch.cat([pos[1], neg_pairs[1]])]
        y_pred = self.edge_recon_fc(torch.cat([sample_src, sample_dst], dim=-1)).squeeze(-1)
        y = torch.cat([torch.ones(len(pos[0])), torch.zeros(len(neg_pairs[0]))]).to(y_pred.dtype)
        loss_bce = self.recon_loss(y_pred, y)
        loss = loss_sce + loss_bce
        if return_parts:
            return loss, dict(sce=loss_sce, bce=loss_bce, recon=recon, hidden=all_hidden,
                              mask_nodes=mask_nodes)
        return loss

    def embed(self, g: OGraph):
        """``model/autoencoder.py:279-282``."""
        return self.encoder(g, g.ndata["attr"])


def uniform_negative_pairs(g: OGraph, k: int, seed: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Stand-in for ``dgl.sampling.global_uniform_negative_sampling(g, k)``
    (``model/autoencoder.py:251``): at most k uniformly drawn (u, v) pairs that are
    neither edges nor self-loops.  DGL draws from a private, unseeded RNG, so no
    sequence can be reproduced; tests inject the pairs on both sides."""
    gen = torch.Generator().manual_seed(seed if seed is not None else random.getrandbits(31))
    n = g.n
    u = torch.randint(0, n, (k,), generator=gen)
    v = torch.randint(0, n, (k,), generator=gen)
    key = u * n + v
    ekey = g.src * n + g.dst
    ok = (u != v) & ~torch.isin(key, ekey)
    return u[ok], v[ok]


def build_model(n_dim, e_dim, num_hidden, num_layers, negative_slope=0.2, mask_rate=0.5, alpha_l=3,
                alias_clone=True):
    """``model/autoencoder.py:18-52`` with the hard-coded hyper-parameters."""
    return GMAEModel(n_dim=n_dim, e_dim=e_dim, hidden_dim=num_hidden, n_layers=num_layers, n_heads=4,
                     activation="prelu", feat_drop=0.1, negative_slope=negative_slope, residual=True,
                     mask_rate=mask_rate, norm="BatchNorm", loss_fn="sce", alpha_l=alpha_l,
                     alias_clone=alias_clone)
