"""Data parallelism for the batch-level training loop: one process per GPU, identical model
replicas, independent graphs (or graph batches) per rank, and ONE all-reduce per step over a
single flat fp32 gradient buffer.

The reference trains in one process (``model/train.py:33-48``, ``train.py:107-118``); the
gradient payload of its models is tiny (0.24 MB for the entity-level model, 4.9 MB for the
batch-level one), so the exchange is latency-bound: bucketing or overlapping it with
backward would only add launches.  The ~40 gradient tensors are packed into the flat buffer
with one ``cat`` and unpacked with one multi-tensor copy; a single-rank run does neither.
"""
from __future__ import annotations

import os
from typing import Iterable, Optional

import torch
import torch.distributed as dist


def collectives_off(world: int) -> bool:
    """A single-rank group skips its collectives (they are identities) -- unless MAGIC_FORCE_COLLECTIVES=1 asks
    for them anyway, which is how the RCCL call path (dtypes, shapes, in-place semantics) is exercised on a
    one-GPU machine."""
    return world == 1 and os.environ.get("MAGIC_FORCE_COLLECTIVES") != "1"


class FlatGradAllReduce:
    def __init__(self, params: Iterable[torch.nn.Parameter], group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.sizes = [p.numel() for p in self.params]
        dev = self.params[0].device
        self.flat = torch.zeros(sum(self.sizes), dtype=torch.float32, device=dev)
        self._comm_stream = None

    def zero(self):
        """``optimizer.zero_grad(set_to_none=True)``: autograd then stores each gradient instead
        of accumulating into a zeroed tensor (no launches)."""
        for p in self.params:
            p.grad = None

    def pack(self) -> torch.Tensor:
        """All gradients side by side in ``self.flat`` (parameter order; a missing gradient is zero)."""
        parts = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.params]
        torch.cat(parts, out=self.flat)
        return self.flat

    def _exchange(self):
        self.pack()
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.div_(self.world)
        views = self.flat.split(self.sizes)
        for p, v in zip(self.params, views):
            if p.grad is None:
                p.grad = torch.empty_like(p)
        torch._foreach_copy_([p.grad for p in self.params], [v.view_as(p) for p, v in zip(self.params, views)])

    def allreduce(self):
        """Average the gradients over the ranks: one collective on the whole buffer.

        On the GPU the exchange runs on a stream of its own, ordered after the caller's stream and before its
        later work.  The collective leaves completion events behind on the stream it ran on, which the process
        group's watchdog thread polls for some time afterwards; polling an event of a stream that has meanwhile
        started recording a HIP graph is an error, so the exchange must never run on a stream that steps are
        recorded on (``model.step.CapturedTrainStep`` records on the stream its eager steps ran on)."""
        if collectives_off(self.world):
            return
        if not self.flat.is_cuda:
            return self._exchange()
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=self.flat.device)
        cur = torch.cuda.current_stream(self.flat.device)
        self._comm_stream.wait_stream(cur)
        with torch.cuda.stream(self._comm_stream):
            self._exchange()
        cur.wait_stream(self._comm_stream)


def broadcast_parameters(model: torch.nn.Module, src: int = 0, group=None):
    """Make every replica start from rank ``src``'s weights (one flat broadcast)."""
    if not (dist.is_available() and dist.is_initialized()) or collectives_off(dist.get_world_size(group)):
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


def embed_graphs(embed_fn, graphs, out_width: int, device, group=None):
    """Embedding extraction over independent graphs (``eval.py:80-97`` of the reference embeds the
    train and test graphs one after the other): graph i is embedded by rank ``i % world`` and its
    ``[N_i, out_width]`` result broadcast from there, so every rank ends with all embeddings, each
    bit-identical to what a single process computes for that graph.

    ``embed_fn(g) -> [g.num_nodes(), out_width]`` float32 tensor on ``device``."""
    multi = dist.is_available() and dist.is_initialized() and not collectives_off(dist.get_world_size(group))
    world = dist.get_world_size(group) if multi else 1
    rank = dist.get_rank(group) if multi else 0
    outs = []
    for i, g in enumerate(graphs):
        owner = i % world
        if owner == rank:
            emb = embed_fn(g).to(device=device, dtype=torch.float32).contiguous()
            assert emb.shape == (g.num_nodes(), out_width)
        else:
            emb = torch.empty(g.num_nodes(), out_width, dtype=torch.float32, device=device)
        if multi:
            dist.broadcast(emb, src=dist.get_global_rank(group, owner) if group is not None else owner, group=group)
        outs.append(emb)
    return outs


def gather_rows(rows: torch.Tensor, counts, group=None) -> torch.Tensor:
    """Concatenation, in rank order, of row blocks of different heights: rank r contributes ``counts[r]``
    rows of ``rows`` (its first ``counts[r]``; the counts are known everywhere).  One equal-sized
    all-gather of zero-padded blocks -- RCCL and gloo both want equal shares -- then the padding is cut."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = [int(c) for c in counts]
    assert len(counts) == world and rows.shape[0] >= counts[rank]
    if collectives_off(world):
        return rows[:counts[0]]
    cmax = max(counts)
    send = torch.zeros((cmax,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    send[:counts[rank]] = rows[:counts[rank]]
    recv = torch.empty((world * cmax,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    return torch.cat([recv[r * cmax:r * cmax + counts[r]] for r in range(world)])
