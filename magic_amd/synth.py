"""Synthetic provenance-graph generator for tests and benchmarks.

Shapes follow the datasets MAGIC is evaluated on (node / edge counts and feature widths
from the reference's ``analysis_result/analysis_*.txt``): a simple directed graph without
duplicate (src, dst) pairs (the reference's parsers de-duplicate edges), a skewed in-degree
distribution with hub nodes, a share of nodes without in-edges, categorical node and edge
types with one-hot attributes (``utils/loaddata.py:258-259``).
"""
from __future__ import annotations

import numpy as np
import torch

from .graph import Graph

# name -> (n_nodes, n_edges, n_dim, e_dim, hidden, layers)
SHAPES = {
    "streamspot": (8000, 170000, 8, 20, 256, 4),        # one StreamSpot-sized graph
    "streamspot_real": (8407, 12715, 8, 20, 256, 4),    # average processed StreamSpot graph
    "wget": (36812, 96018, 8, 4, 256, 4),               # average Unicorn-Wget graph
    "wget_max": (126356, 351388, 8, 4, 256, 4),
    "theia": (700000, 3000000, 5, 17, 64, 3),           # E3-THEIA-shaped full graph
    "trace": (785330, 1054972, 11, 23, 64, 3),          # E3-Trace train graph 0
}


def power_law_edges(n: int, e: int, rng: np.random.Generator, zero_in_frac: float = 0.35,
                    exponent: float = 0.75, self_loop_frac: float = 0.001):
    """Duplicate-free edge list: dst ~ rank^-exponent over a random subset of the nodes
    (the rest keep in-degree 0), src uniform; sorted by (src, draw order) like the edge
    iteration order of a networkx DiGraph."""
    n_active = max(1, int(round(n * (1.0 - zero_in_frac))))
    active = rng.permutation(n)[:n_active]
    w = (np.arange(1, n_active + 1, dtype=np.float64)) ** (-exponent)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    src_all = np.empty(0, np.int64)
    dst_all = np.empty(0, np.int64)
    need = e
    while need > 0:
        m = int(need * 1.15) + 16
        dst = active[np.searchsorted(cdf, rng.random(m), side="right").clip(max=n_active - 1)]
        src = rng.integers(0, n, m)
        loops = rng.random(m) < self_loop_frac
        src = np.where(loops, dst, src)
        keep = loops | (src != dst)
        src_all = np.concatenate([src_all, src[keep]])
        dst_all = np.concatenate([dst_all, dst[keep]])
        key = src_all * n + dst_all
        _, first = np.unique(key, return_index=True)
        first.sort()
        src_all, dst_all = src_all[first], dst_all[first]
        need = e - src_all.size
    src_all, dst_all = src_all[:e], dst_all[:e]
    order = np.argsort(src_all, kind="stable")
    return src_all[order], dst_all[order]


def categorical(n: int, k: int, rng: np.random.Generator):
    p = 1.0 / np.arange(1, k + 1)
    p /= p.sum()
    t = rng.choice(k, size=n, p=p)
    if n >= k:
        t[:k] = np.arange(k)  # every class present, as in the real datasets' metadata
    return t


def make_graph(n: int, e: int, n_dim: int, e_dim: int, seed: int, device="cpu", attrs: bool = True) -> Graph:
    rng = np.random.default_rng(seed)
    src, dst = power_law_edges(n, e, rng)
    g = Graph(torch.from_numpy(src), torch.from_numpy(dst), n)
    g.ndata["type"] = torch.from_numpy(categorical(n, n_dim, rng))
    g.edata["type"] = torch.from_numpy(categorical(e, e_dim, rng))
    if attrs:
        g.ndata["attr"] = torch.nn.functional.one_hot(g.ndata["type"], n_dim).float()
        g.edata["attr"] = torch.nn.functional.one_hot(g.edata["type"], e_dim).float()
        g.mark_node_attr_onehot()
    return g.to(device) if str(device) != "cpu" else g


def make_shape(name: str, seed: int, device="cpu", scale: float = 1.0) -> Graph:
    n, e, n_dim, e_dim, _, _ = SHAPES[name]
    return make_graph(max(16, int(n * scale)), max(16, int(e * scale)), n_dim, e_dim, seed, device)


def gaussian_mixture(n: int, d: int, seed: int, clusters: int = 200, sigma: float = 0.05,
                     dup_frac: float = 0.0) -> np.ndarray:
    """Embedding-like float32 rows: a mixture of tight clusters, optionally with exact
    duplicates (provenance embeddings contain many)."""
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((clusters, d)).astype(np.float32)
    which = rng.integers(0, clusters, n)
    x = centers[which] + sigma * rng.standard_normal((n, d)).astype(np.float32)
    if dup_frac > 0:
        m = int(n * dup_frac)
        x[rng.integers(0, n, m)] = x[rng.integers(0, n, m)]
    return x.astype(np.float32)
