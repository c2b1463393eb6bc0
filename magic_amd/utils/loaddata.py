"""Graph preparation used by the training / evaluation loops (``utils/loaddata.py``).

``transform_graph`` (one-hot node and edge attributes from the integer types,
``utils/loaddata.py:242-260``) is the part on the hot path.  Around it: a DGL-free reader of the
node-link JSON graphs the reference's parsers write (``utils/streamspot_parser.py:56-57``,
``utils/wget_parser.py:1074-1075``), producing the graph ``dgl.from_networkx(nx.node_link_graph(...),
node_attrs=['type'], edge_attrs=['type'])`` builds at ``utils/loaddata.py:62-66`` and ``:141-145``
-- same node numbering, same edge order -- and the two batch-level datasets on top of it.  The
reference's ``graphs.pkl`` / ``train{i}.pkl`` caches are pickled DGL objects and are not read.
"""
import json
import os

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


def graph_from_node_link(data: dict) -> Graph:
    """The graph ``dgl.from_networkx(nx.node_link_graph(data), node_attrs=['type'],
    edge_attrs=['type'])`` returns, without networkx or DGL:

    * node ids are the ranks of the node labels in sorted order (DGL relabels with
      ``convert_node_labels_to_integers(ordering='sorted')``);
    * edges come in the iteration order of a networkx DiGraph: grouped by source in the order the
      nodes were first added (the ``nodes`` list, then endpoints first seen in ``links``), inside
      a group in the order the (source, target) pairs first appeared; a repeated pair keeps its
      place and takes the attributes of its last occurrence (``multigraph: false``).
    """
    if data.get("multigraph", False):
        raise NotImplementedError("multigraph node-link data: MAGIC's parsers write simple digraphs")
    if not data.get("directed", False):
        raise NotImplementedError("undirected node-link data: MAGIC's parsers write directed graphs")
    links = data["links"] if "links" in data else data["edges"]
    first_seen = {}        # label -> position in node insertion order
    node_type = {}
    for nd in data["nodes"]:
        first_seen.setdefault(nd["id"], len(first_seen))
        node_type[nd["id"]] = nd.get("type")
    for e in links:
        for lab in (e["source"], e["target"]):
            first_seen.setdefault(lab, len(first_seen))
    new_id = {lab: i for i, lab in enumerate(sorted(first_seen))}
    pair_pos, pair_type = {}, {}
    for e in links:
        key = (e["source"], e["target"])
        pair_pos.setdefault(key, len(pair_pos))
        pair_type[key] = e.get("type")
    order = sorted(pair_pos, key=lambda k: (first_seen[k[0]], pair_pos[k]))
    missing = [lab for lab in first_seen if node_type.get(lab) is None]
    if missing or any(pair_type[k] is None for k in order):
        raise KeyError("node-link data: every node and every link needs an integer 'type'")
    src = torch.tensor([new_id[k[0]] for k in order], dtype=torch.int64)
    dst = torch.tensor([new_id[k[1]] for k in order], dtype=torch.int64)
    g = Graph(src, dst, len(new_id))
    ntype = torch.empty(len(new_id), dtype=torch.int64)
    for lab, i in new_id.items():
        ntype[i] = int(node_type[lab])
    g.ndata["type"] = ntype
    g.edata["type"] = torch.tensor([int(pair_type[k]) for k in order], dtype=torch.int64)
    return g


def load_node_link_json(path: str) -> Graph:
    with open(path) as f:
        return graph_from_node_link(json.load(f))


def load_json_dataset(name: str, root: str = "./data"):
    """``[(Graph, label), ...]`` of a batch-level dataset from its JSON files, files and labels as at
    ``utils/loaddata.py:54-71`` (StreamSpot: ``<root>/streamspot/{1..600}.json``, graphs 300-399
    attacks) and ``:133-150`` (Wget: ``<root>/wget/final/{0..149}.json``, graphs 0-24 attacks);
    registers it so that ``load_batch_level_dataset(name)`` finds it."""
    if name == "streamspot":
        files = [os.path.join(root, "streamspot", f"{i + 1}.json") for i in range(600)]
        labels = [1 if 300 <= i <= 399 else 0 for i in range(600)]
    elif name == "wget":
        files = [os.path.join(root, "wget", "final", f"{i}.json") for i in range(150)]
        labels = [1 if i <= 24 else 0 for i in range(150)]
    else:
        raise NotImplementedError(name)
    dataset = [(load_node_link_json(f), y) for f, y in zip(files, labels)]
    register_batch_level_dataset(name, dataset)
    return dataset
