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
                      grad_exchange=None, progress: bool = True, step_hook=None):
    """``grad_exchange``: a :class:`magic_amd.parallel.FlatGradAllReduce` when several ranks train
    replicas on different batches (its ``zero()`` replaces ``optimizer.zero_grad()``).
    ``step_hook(epoch, step, batch_g)`` may return keyword overrides for ``model(batch_g, ...)``
    (injected permutation / samples / dropout masks, see ``GMAEModel.compute_loss``)."""
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
        for it, batch in enumerate(train_loader):
            batch_g = [transform_graph(graphs[idx][0], n_dim, e_dim).to(device) for idx in batch]
            batch_g = batch_graphs(batch_g)                     # model/train.py:39
            if not model.training:                              # model/train.py:40 (a walk over ~60 modules per call)
                model.train()
            loss = model(batch_g, **(step_hook(epoch, it, batch_g) or {})) if step_hook else model(batch_g)
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


def entity_level_train(model, load_graph, n_train, optimizer, max_epoch, device, n_dim=0, e_dim=0,
                       progress: bool = True, step_hook=None, capture: bool = False):
    """The entity-level loop of the reference's driver (``train.py:107-118``): per epoch, for each of
    the ``n_train`` training graphs ``g = load_graph(i).to(device)``, ``loss = model(g) / n_train``,
    ``zero_grad``, ``backward``, ``step`` -- sequential Adam updates, one graph per step.

    ``load_graph(i)`` returns the i-th training graph (the reference unpickles it from disk every
    step, ``train.py:110``); graphs that carry integer types only get their one-hot attributes from
    ``transform_graph`` (``n_dim`` / ``e_dim``).  The epoch's loss stays on the device until it is
    printed.  ``step_hook(epoch, i, g)`` may return keyword overrides for ``model(g, ...)``.
    ``capture=True`` records the step of every training graph as a HIP graph on its first epochs and
    replays it afterwards (:class:`magic_amd.model.step.CapturedTrainStep`; ``optimizer`` must then be
    capturable, e.g. ``step.make_adam``, and ``load_graph`` must return the same structure per index)."""
    from .step import CapturedTrainStep
    epochs = range(max_epoch)
    bar = None
    if progress:
        try:
            from tqdm import tqdm
            bar = epochs = tqdm(epochs)
        except ImportError:  # pragma: no cover
            bar = None
    steps = {}
    for epoch in epochs:
        epoch_loss = []
        for i in range(n_train):
            model.train()
            if capture:
                if i not in steps:
                    g = load_graph(i).to(device)
                    if "attr" not in g.edata:
                        g = transform_graph(g, n_dim, e_dim)
                    steps[i] = CapturedTrainStep(model, optimizer, g, n_dim or g.ndata["attr"].shape[1],
                                                 loss_scale=1.0 / n_train)
                loss = steps[i](**((step_hook(epoch, i, steps[i].g) or {}) if step_hook else {})).clone()
            else:
                g = load_graph(i).to(device)
                if "attr" not in g.ndata or "attr" not in g.edata:
                    g = transform_graph(g, n_dim, e_dim)
                loss = model(g, **(step_hook(epoch, i, g) or {})) if step_hook else model(g)
                loss = loss / n_train                           # train.py:113
                optimizer.zero_grad()
                loss.backward()
                optimizer.step()
                del g
            epoch_loss.append(loss.detach())
        total = float(torch.stack(epoch_loss).sum()) if epoch_loss else float("nan")
        if bar is not None:
            bar.set_description(f"Epoch {epoch} | train_loss: {total:.4f}")
    return model
