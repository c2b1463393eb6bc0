"""Factories and seeding helpers with the reference's names and behaviour (``utils/utils.py``)."""
import random
from functools import partial

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim


def create_optimizer(opt, model, lr, weight_decay):
    """``utils/utils.py:9-43``: Adam with L2-in-gradient weight decay is MAGIC's default."""
    opt_lower = opt.lower().split("_")[-1]
    parameters = model.parameters()
    opt_args = dict(lr=lr, weight_decay=weight_decay)
    if opt_lower == "adam":
        return optim.Adam(parameters, **opt_args)
    if opt_lower == "adamw":
        return optim.AdamW(parameters, **opt_args)
    if opt_lower == "adadelta":
        return optim.Adadelta(parameters, **opt_args)
    if opt_lower == "radam":
        return optim.RAdam(parameters, **opt_args)
    if opt_lower == "sgd":
        opt_args["momentum"] = 0.9
        return optim.SGD(parameters, **opt_args)
    raise ValueError(f"Invalid optimizer {opt!r}")


def random_shuffle(x, y):
    """``utils/utils.py:46-59``."""
    idx = list(range(len(x)))
    random.shuffle(idx)
    return x[idx], y[idx]


def set_random_seed(seed):
    """``utils/utils.py:62-74``: seeds Python, numpy and torch (CPU and every GPU)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)


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
    """``utils/utils.py:104-121``.  Only lower-case names match; the model passes
    ``'BatchNorm'`` (``model/autoencoder.py:48``), which therefore yields ``None``."""
    if name == "layernorm":
        return nn.LayerNorm
    if name == "batchnorm":
        return nn.BatchNorm1d
    if name == "graphnorm":
        return partial(NormLayer, norm_type="groupnorm")
    return None


class NormLayer(nn.Module):
    """Per-graph normalisation hook of the reference (``utils/utils.py:124-196``).

    ``create_norm('graphnorm')`` binds ``norm_type='groupnorm'``, a name the reference's
    constructor rejects (``utils/utils.py:141-152`` accept batchnorm / layernorm / graphnorm
    only), so selecting it raises there as it does here.  The batchnorm / layernorm forms
    simply wrap the torch layer.
    """

    def __init__(self, hidden_dim, norm_type):
        super().__init__()
        wrapped = {"batchnorm": nn.BatchNorm1d, "layernorm": nn.LayerNorm}.get(norm_type)
        if wrapped is None:
            raise NotImplementedError(f"norm_type {norm_type!r}")
        self.norm = wrapped(hidden_dim)

    def forward(self, graph, x):
        return self.norm(x)
