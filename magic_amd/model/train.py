"""Batch-level training loop with the reference's signature (``model/train.py:18-48``).

Same statements as the reference -- per batch: ``transform_graph`` on every graph, ``batch``,
``model(batch_g)``, ``zero_grad``, ``backward``, ``step`` -- with two differences that do not
change the arithmetic: the loss of a batch stays on the device until the epoch's mean is
printed (the reference's ``loss.item()`` per batch is a host synchronisation per step), and an
optional data-parallel gradient exchange between ``backward`` and ``step``.
"""
from __future__ import annotations

from typing import Optional

import torch

from ..graph import batch as batch_graphs
from ..utils.loaddata import transform_graph


def batch_level_train(model, graphs, train_loader, optimizer, max_epoch, device, n_dim=0, e_dim=0,
                      grad_exchange=None, progress: bool = True):
    """``grad_exchange``: a :class:`magic_amd.parallel.FlatGradAllReduce` when several ranks train
    replicas on different batches (its ``zero()`` replaces ``optimizer.zero_grad()``)."""
    epochs = range(max_epoch)
    bar: Optional[object] = None
    if progress:
        try:
            from tqdm import tqdm
            bar = epochs = tqdm(epochs)
        except ImportError:  # pragma: no cover - tqdm is optional
            bar = None
    for epoch in epochs:
        model.train()
        losses = []
        for _, batch in enumerate(train_loader):
            batch_g = [transform_graph(graphs[idx][0], n_dim, e_dim).to(device) for idx in batch]
            batch_g = batch_graphs(batch_g)                     # model/train.py:39
            model.train()
            loss = model(batch_g)
            if grad_exchange is not None:
                grad_exchange.zero()
            else:
                optimizer.zero_grad()
            loss.backward()
            if grad_exchange is not None:
                grad_exchange.allreduce()
            optimizer.step()
            losses.append(loss.detach())
            del batch_g
        mean_loss = float(torch.stack(losses).mean()) if losses else float("nan")
        if bar is not None:
            bar.set_description(f"Epoch {epoch} | train_loss: {mean_loss:.4f}")
    return model
