"""Brute-force Euclidean nearest neighbours on the MI355X.

``NearestNeighbors`` offers the two calls MAGIC makes on scikit-learn's class of the same
name (``model/eval.py:150-158``, ``:294-311``): ``fit(x_train)`` and
``kneighbors(x, n_neighbors)`` returning ``(distances float64 [Q,k], indices int64 [Q,k])``
in ascending distance order.  Distances are formed in float64 from float32 inputs, as
scikit-learn's float32 brute-force path does; ties are ordered by train index.

``ShardedNearestNeighbors`` splits the train rows over the ranks of a ``torch.distributed``
group: every rank searches its shard for all queries, the per-shard lists are exchanged and
merged, and every rank ends with the global result (identical to the single-GPU one).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import _lib, ops  # noqa: F401  (ops registers torch.ops.magic.*)
from .parallel import collectives_off


def _as_device_f32(x, device) -> torch.Tensor:
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    return x.to(device=device, dtype=torch.float32).contiguous()


class NearestNeighbors:
    def __init__(self, n_neighbors: int = 5, n_jobs=None, device="cuda", query_block: int = 1 << 20):
        self.n_neighbors = n_neighbors
        self.device = torch.device(device)
        self.query_block = query_block
        self._train: Optional[torch.Tensor] = None

    def fit(self, x):
        self._train = _as_device_f32(x, self.device)
        return self

    def kneighbors_device(self, x: torch.Tensor, n_neighbors: Optional[int] = None, index_offset: int = 0):
        """Device tensors in, device tensors out (no host round trip)."""
        k = n_neighbors or self.n_neighbors
        if self._train is None:
            raise RuntimeError("call fit() first")
        if k > self._train.shape[0]:
            raise ValueError(f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {k}, "
                             f"n_samples_fit = {self._train.shape[0]}")
        dists, idxs = [], []
        block = self.query_block
        if self._train.shape[1] > 64:
            # rows wider than 64 take the kernel without matrix-core tiling, which keeps a
            # [queries, train rows] float64 key matrix in its workspace: bound that matrix to 1 GiB
            block = max(1, min(block, (1 << 30) // (8 * self._train.shape[0])))
        for s in range(0, x.shape[0], block):
            d, i = torch.ops.magic.knn_topk(self._train, x[s:s + block].contiguous(), k, index_offset)
            dists.append(d)
            idxs.append(i)
        if len(dists) == 1:
            return dists[0], idxs[0]
        return torch.cat(dists), torch.cat(idxs)

    def mean_distances_device(self, x: torch.Tensor, n_neighbors: Optional[int] = None) -> torch.Tensor:
        """Per-query mean of the k nearest distances (``distances.mean(axis=1)``, ``model/eval.py:313``)."""
        return self.kneighbors_device(x, n_neighbors)[0].mean(dim=1)

    def kneighbors(self, x, n_neighbors: Optional[int] = None, return_distance: bool = True):
        was_numpy = isinstance(x, np.ndarray)
        xq = _as_device_f32(x, self.device)
        d, i = self.kneighbors_device(xq, n_neighbors)
        if was_numpy:
            d, i = d.cpu().numpy(), i.cpu().numpy()
        return (d, i) if return_distance else i


def merge_topk(part_dist: torch.Tensor, part_idx: torch.Tensor):
    """[P,Q,k] per-shard results -> [Q,k] global k best, ordered by (distance, index)."""
    if part_dist.device.type == "cuda":
        return torch.ops.magic.knn_merge(part_dist, part_idx)
    # host tensors (used by the gloo tests of the exchange logic): same order rule
    P, Q, k = part_dist.shape
    d = part_dist.permute(1, 0, 2).reshape(Q, P * k)
    i = part_idx.permute(1, 0, 2).reshape(Q, P * k)
    i_key = torch.where(i < 0, torch.full_like(i, torch.iinfo(torch.int64).max), i)
    order = torch.from_numpy(np.lexsort((i_key.numpy(), d.numpy()), axis=1))[:, :k]
    return torch.gather(d, 1, order), torch.gather(i, 1, order)


class ShardedNearestNeighbors:
    """Train rows sharded over the process group, queries replicated.

    Rank r keeps rows ``[r*ceil(N/G), (r+1)*ceil(N/G))``.  Every rank computes the local top-k of
    all queries with global indices; one all-to-all hands rank r the G partial lists of query
    block r, which it merges in shard order -- the single-device (distance, index) ordering
    exactly.  ``mean_distances_device`` then all-gathers only the ``[Q]`` per-query means;
    ``kneighbors_device`` all-gathers the merged lists for callers that need them everywhere.
    """

    def __init__(self, n_neighbors: int, group=None, device="cuda", local=None):
        """``local``: the per-shard searcher (``fit(x)``, ``kneighbors_device(x, k, index_offset)``, a
        ``_train`` attribute); the HIP search unless a test injects a host one."""
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.k = n_neighbors
        self.device = torch.device(device)
        self.local = local if local is not None else NearestNeighbors(n_neighbors, device=device)
        self.offset = 0

    @staticmethod
    def shard_bounds(n: int, world: int, rank: int):
        per = (n + world - 1) // world
        lo = min(rank * per, n)
        return lo, min(lo + per, n)

    def fit(self, x_full=None, x_local=None, offset: Optional[int] = None):
        """Either the full train matrix (each rank slices its rows) or the local shard."""
        if x_full is not None:
            lo, hi = self.shard_bounds(x_full.shape[0], self.world, self.rank)
            x_local, offset = x_full[lo:hi], lo
        self.offset = int(offset or 0)
        self.local.fit(x_local)
        return self

    def kneighbors(self, x, n_neighbors: Optional[int] = None, return_distance: bool = True):
        """scikit-learn's call shape (as ``NearestNeighbors.kneighbors``), so that the sharded search
        can be handed to ``model.eval.entity_level_distances(..., nbrs=...)``: every rank passes the
        same replicated queries and gets the same global result."""
        was_numpy = isinstance(x, np.ndarray)
        d, i = self.kneighbors_device(_as_device_f32(x, self.device), n_neighbors)
        if was_numpy:
            d, i = d.cpu().numpy(), i.cpu().numpy()
        return (d, i) if return_distance else i

    def local_topk(self, xq: torch.Tensor, k: int):
        n_local = self.local._train.shape[0]
        kk = min(k, n_local)
        d, i = self.local.kneighbors_device(xq, kk, self.offset)
        if kk < k:  # shard smaller than k: pad with "none"
            pad_d = torch.full((xq.shape[0], k - kk), float("inf"), dtype=d.dtype, device=d.device)
            pad_i = torch.full((xq.shape[0], k - kk), -1, dtype=i.dtype, device=i.device)
            d, i = torch.cat([d, pad_d], 1), torch.cat([i, pad_i], 1)
        return d, i

    def kneighbors_device(self, xq: torch.Tensor, n_neighbors: Optional[int] = None):
        """Global ``(distances, indices)`` of all queries on every rank."""
        k = n_neighbors or self.k
        d, i = self.local_topk(xq, k)
        md, mi, q = exchange_and_merge_blocks(d, i, self.group)
        return gather_blocks(md, q, self.group), gather_blocks(mi, q, self.group)

    def mean_distances_device(self, xq: torch.Tensor, n_neighbors: Optional[int] = None) -> torch.Tensor:
        """Per-query mean of the k nearest distances (``distances.mean(axis=1)`` at ``model/eval.py:313``)
        on every rank: the merged lists stay with the rank that merged them, only the ``[Q]`` means travel."""
        k = n_neighbors or self.k
        d, i = self.local_topk(xq, k)
        md, _, q = exchange_and_merge_blocks(d, i, self.group)
        return gather_blocks(md.mean(dim=1), q, self.group)


def exchange_and_merge_blocks(d: torch.Tensor, i: torch.Tensor, group=None):
    """One all-to-all: the queries are cut into ``world`` blocks of ``ceil(Q / world)`` rows and rank r
    receives every rank's partial top-k lists of block r (``(Q / world) * k * 16`` bytes per peer, all
    xGMI links busy at once), then merges them in shard order -- the single-device (distance, index)
    order.  Returns the merged ``[ceil(Q / world), k]`` lists of this rank's block (rows past Q are
    padding) and Q."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    q, k = d.shape
    if collectives_off(world):
        return d, i, q
    per = (q + world - 1) // world
    pad = per * world - q
    if pad:
        d = torch.cat([d, torch.full((pad, k), float("inf"), dtype=d.dtype, device=d.device)])
        i = torch.cat([i, torch.full((pad, k), -1, dtype=i.dtype, device=i.device)])
    rd, ri = torch.empty_like(d), torch.empty_like(i)
    dist.all_to_all_single(rd, d.contiguous(), group=group)     # chunk j of the send buffer goes to rank j
    dist.all_to_all_single(ri, i.contiguous(), group=group)
    md, mi = merge_topk(rd.view(world, per, k), ri.view(world, per, k))
    return md, mi, q


def gather_blocks(block: torch.Tensor, q: int, group=None) -> torch.Tensor:
    """All-gather of the ranks' row blocks, cut back to the first ``q`` rows."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if collectives_off(world):
        return block[:q]
    out = torch.empty((world * block.shape[0],) + tuple(block.shape[1:]), dtype=block.dtype, device=block.device)
    dist.all_gather_into_tensor(out, block.contiguous(), group=group)
    return out[:q]


def exchange_and_merge(d: torch.Tensor, i: torch.Tensor, group=None):
    """Global ``[Q,k]`` lists on every rank from the per-rank partial lists (all-to-all, merge of
    Q / world queries per rank, all-gather of the merged blocks)."""
    md, mi, q = exchange_and_merge_blocks(d, i, group)
    return gather_blocks(md, q, group), gather_blocks(mi, q, group)
