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
        for s in range(0, x.shape[0], self.query_block):
            d, i = torch.ops.magic.knn_topk(self._train, x[s:s + self.query_block].contiguous(), k, index_offset)
            dists.append(d)
            idxs.append(i)
        if len(dists) == 1:
            return dists[0], idxs[0]
        return torch.cat(dists), torch.cat(idxs)

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

    Rank r keeps rows ``[r*ceil(N/G), (r+1)*ceil(N/G))``.  ``kneighbors_device`` computes the
    local top-k with global indices, all-gathers the ``[Q,k]`` lists (Q*k*16 bytes per rank;
    one collective, all xGMI links busy at once) and merges them in shard order, which
    reproduces the single-device (distance, index) ordering exactly.
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
        k = n_neighbors or self.k
        d, i = self.local_topk(xq, k)
        return exchange_and_merge(d, i, self.group)


def exchange_and_merge(d: torch.Tensor, i: torch.Tensor, group=None):
    """All-gather the per-rank ``[Q,k]`` lists and merge them (works for gloo/CPU and RCCL)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1:
        return d, i
    q, k = d.shape
    pd = torch.empty((world * q, k), dtype=d.dtype, device=d.device)
    pi = torch.empty((world * q, k), dtype=i.dtype, device=i.device)
    dist.all_gather_into_tensor(pd, d.contiguous(), group=group)
    dist.all_gather_into_tensor(pi, i.contiguous(), group=group)
    return merge_topk(pd.view(world, q, k), pi.view(world, q, k))
