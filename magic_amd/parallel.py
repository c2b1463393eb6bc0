"""Data parallelism for the batch-level training loop: one process per GPU, identical model
replicas, independent graphs (or graph batches) per rank, and ONE all-reduce per step over a
single flat fp32 gradient buffer.

The reference trains in one process (``model/train.py:33-48``, ``train.py:107-118``); the
gradient payload of its models is tiny (0.24 MB for the entity-level model, 4.9 MB for the
batch-level one), so the exchange is latency-bound: bucketing or overlapping it with
backward would only add launches.  Gradients live as views into the flat buffer, so no
pack / unpack copies surround the collective.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch
import torch.distributed as dist


class FlatGradAllReduce:
    def __init__(self, params: Iterable[torch.nn.Parameter], group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        """Replacement for ``optimizer.zero_grad()`` that keeps the gradient views alive."""
        self.flat.zero_()

    def check_views(self):
        for p in self.params:
            if p.grad is None or p.grad.untyped_storage().data_ptr() != self.flat.untyped_storage().data_ptr():
                raise RuntimeError("a gradient left the flat buffer (zero_grad(set_to_none=True)?); "
                                   "use FlatGradAllReduce.zero() instead")

    def allreduce(self):
        """Average the gradients over the ranks: one collective on the whole buffer."""
        if self.world == 1:
            return
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.div_(self.world)


def broadcast_parameters(model: torch.nn.Module, src: int = 0, group=None):
    """Make every replica start from rank ``src``'s weights (one flat broadcast)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    tensors = [p.data for p in model.parameters()] + [b.data for b in model.buffers()]
    flat = torch.cat([t.reshape(-1).float() for t in tensors])
    dist.broadcast(flat, src=src, group=group)
    off = 0
    for t in tensors:
        t.copy_(flat[off:off + t.numel()].view_as(t).to(t.dtype))
        off += t.numel()


def shard_indices(n_items: int, world: int, rank: int):
    """Round-robin assignment of independent items (graphs) to ranks."""
    return list(range(rank, n_items, world))
