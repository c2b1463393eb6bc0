"""Shared helpers for the parity tests: seeded graphs on both sides, model twins."""
import numpy as np
import torch

from magic_amd import synth
from magic_amd.graph import Graph
from oracle import gat_oracle as O


def small_graph(n, e, n_dim, e_dim, seed, hub_deg=0, hub_out=0):
    """Synthetic graph plus optional extra hub (in-degree hub_deg, out-degree hub_out) so that
    the chunked heavy-segment kernels run; also guarantees self-loops and zero in-degree nodes."""
    rng = np.random.default_rng(seed)
    src, dst = synth.power_law_edges(n, e, rng)
    extra_s, extra_d = [], []
    if hub_deg:
        s = rng.permutation(n)[:hub_deg]
        extra_s.append(s)
        extra_d.append(np.full(hub_deg, 3))
    if hub_out:
        d = rng.permutation(n)[:hub_out]
        extra_s.append(np.full(hub_out, 5))
        extra_d.append(d)
    if extra_s:
        src = np.concatenate([src] + extra_s)
        dst = np.concatenate([dst] + extra_d)
        key = src.astype(np.int64) * n + dst
        _, first = np.unique(key, return_index=True)
        first.sort()
        src, dst = src[first], dst[first]
    ntype = synth.categorical(n, n_dim, rng)
    etype = synth.categorical(src.size, e_dim, rng)
    return src.astype(np.int64), dst.astype(np.int64), ntype, etype


def to_ograph(src, dst, n, ntype, etype, n_dim, e_dim, dtype=torch.float32):
    g = O.OGraph(torch.from_numpy(src), torch.from_numpy(dst), n)
    g.ndata["type"] = torch.from_numpy(ntype)
    g.edata["type"] = torch.from_numpy(etype)
    g.ndata["attr"] = torch.nn.functional.one_hot(g.ndata["type"], n_dim).to(dtype)
    g.edata["attr"] = torch.nn.functional.one_hot(g.edata["type"], e_dim).to(dtype)
    return g


def to_graph(src, dst, n, ntype, etype, n_dim, e_dim, device):
    g = Graph(torch.from_numpy(src), torch.from_numpy(dst), n)
    g.ndata["type"] = torch.from_numpy(ntype)
    g.edata["type"] = torch.from_numpy(etype)
    g.ndata["attr"] = torch.nn.functional.one_hot(g.ndata["type"], n_dim).float()
    g.edata["attr"] = torch.nn.functional.one_hot(g.edata["type"], e_dim).float()
    g.mark_node_attr_onehot()           # (what the loaders' transform_graph does: the coded first layer needs it)
    return g.to(device)


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def row_rel_err(a, b, floor=1e-3):
    """Per-row relative error: max over rows of ||a_row - b_row|| / ||b_row||, taken over the rows
    whose reference norm is at least ``floor`` times the largest row norm (rows that are numerically
    zero have no meaningful relative error).  Stricter than ``rel_err``, which normalises by the
    largest entry of the whole tensor."""
    a = a.detach().double().cpu().reshape(a.shape[0], -1)
    b = b.detach().double().cpu().reshape(b.shape[0], -1)
    nb = b.norm(dim=1)
    keep = nb >= floor * float(nb.max() + 1e-300)
    if not bool(keep.any()):
        return 0.0
    return float(((a - b).norm(dim=1)[keep] / nb[keep]).max())
