"""Graph preparation used by the training / evaluation loops (``utils/loaddata.py``).

Only the part on the hot path lives here: ``transform_graph`` (one-hot node and edge
attributes from the integer types, ``utils/loaddata.py:242-260``).  The dataset readers of
the reference unpickle DGL objects and are not part of this package; the batch-level
entry point accepts an in-memory dataset instead.
"""
import torch
import torch.nn.functional as F

from ..graph import Graph

_REGISTRY = {}


def register_batch_level_dataset(name, graphs_and_labels):
    """Make ``[(Graph, label), ...]`` available to ``load_batch_level_dataset(name)``."""
    _REGISTRY[name] = list(graphs_and_labels)


def load_batch_level_dataset(dataset_name):
    """Same result dictionary as ``utils/loaddata.py:209-239`` for a registered dataset."""
    if dataset_name not in _REGISTRY:
        raise FileNotFoundError(
            f"dataset {dataset_name!r} is not registered; the reference's on-disk caches are pickled DGL "
            "objects and cannot be read without DGL -- build Graph objects and call "
            "register_batch_level_dataset()")
    dataset = _REGISTRY[dataset_name]
    node_feature_dim = max(int(g.ndata["type"].max().item()) for g, _ in dataset) + 1
    edge_feature_dim = max(int(g.edata["type"].max().item()) for g, _ in dataset) + 1
    return {"dataset": dataset,
            "train_index": [i for i in range(len(dataset)) if dataset[i][1] == 0],
            "full_index": list(range(len(dataset))),
            "n_feat": node_feature_dim,
            "e_feat": edge_feature_dim}


def transform_graph(g: Graph, node_feature_dim: int, edge_feature_dim: int) -> Graph:
    """``utils/loaddata.py:257-260``: fresh one-hot float attributes on a clone."""
    new_g = g.clone()
    new_g.ndata["attr"] = F.one_hot(g.ndata["type"].view(-1).long(), num_classes=node_feature_dim).float()
    new_g.edata["attr"] = F.one_hot(g.edata["type"].view(-1).long(), num_classes=edge_feature_dim).float()
    new_g.mark_edge_attr_onehot()
    return new_g
