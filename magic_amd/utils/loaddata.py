"""Graph preparation used by the training / evaluation loops (``utils/loaddata.py``).

``transform_graph`` (one-hot node and edge attributes from the integer types,
``utils/loaddata.py:242-260``) is the part on the hot path.  Around it: a DGL-free reader of the
node-link JSON graphs the reference's parsers write (``utils/streamspot_parser.py:56-57``,
``utils/wget_parser.py:1074-1075``), producing the graph ``dgl.from_networkx(nx.node_link_graph(...),
node_attrs=['type'], edge_attrs=['type'])`` builds at ``utils/loaddata.py:62-66`` and ``:141-145``
-- same node numbering, same edge order -- and the two batch-level datasets on top of it.  The
reference's ``graphs.pkl`` / ``train{i}.pkl`` caches are pickled DGL objects and are not read.

Entity-level datasets (DARPA E3 Trace / THEIA / CADETS): the trace parser leaves ``train.pkl`` and
``test.pkl`` (lists of ``networkx.node_link_data`` dictionaries) and ``malicious.pkl`` (ids and names of
the malicious test nodes) under ``./data/<name>/`` (``utils/trace_parser.py:235-237``).  They hold plain
containers, numbers and strings only, so they are read with an unpickler that refuses every global
(no code in the file can run).  ``preload_entity_level_dataset`` / ``load_metadata`` /
``load_entity_level_dataset`` keep the reference's names and ``metadata.json`` layout
(``utils/loaddata.py:263-363``); the per-graph cache is an ``.npz`` file of int32 arrays
(edge list, types, and the CSC / CSR arrays once they were built) instead of a pickled DGL object.
"""
import json
import os
import pickle
import weakref

import numpy as np
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


# One-hot edge attributes kept for reuse across calls: a budget per process, MAGIC_EDGE_ATTR_CACHE_MB megabytes
# (default 4096; 0 switches the cache off).  A StreamSpot-sized graph holds 13.6 MB this way, so the default keeps
# about 300 training graphs' attributes beside the working set of a step (under 2 % of the 288 GB of an MI355X).
_EDGE_ATTR_CACHE_BYTES = int(float(os.environ.get("MAGIC_EDGE_ATTR_CACHE_MB", "4096")) * (1 << 20))
_edge_attr_bytes = 0


def _release_edge_attr(nbytes: int):
    global _edge_attr_bytes
    _edge_attr_bytes -= nbytes


def _edge_one_hot(g: Graph, edge_feature_dim: int) -> torch.Tensor:
    """``one_hot(edata['type']).float()``, kept on the source graph while the type tensor and the attribute
    tensor stay untouched (version counters): the training loops call ``transform_graph`` on the same
    graphs every epoch, nothing on the training path writes into the edge attributes, and at the
    StreamSpot shape the twelve one-hot matrices of a batch are 160 MB and 24 launches per step.
    The budget is given back when an entry is replaced and when the tensor that holds it is freed (with its graph)."""
    global _edge_attr_bytes
    t = g.edata["type"]
    hit = g._cache.get("edge_onehot")
    if hit is not None:
        t_ref, t_ver, dim, attr, a_ver, fin = hit
        if t_ref() is t and t_ver == t._version and dim == edge_feature_dim and a_ver == attr._version:
            return attr
        fin()                                   # gives the bytes back now (a finalizer runs once)
        del g._cache["edge_onehot"]
    attr = _one_hot_float(t, edge_feature_dim)
    nbytes = attr.numel() * 4
    if t.is_cuda and _edge_attr_bytes + nbytes <= _EDGE_ATTR_CACHE_BYTES:
        fin = weakref.finalize(attr, _release_edge_attr, nbytes)
        g._cache["edge_onehot"] = (weakref.ref(t), t._version, edge_feature_dim, attr, attr._version, fin)
        _edge_attr_bytes += nbytes
    return attr


def _one_hot_float(ids: torch.Tensor, num_classes: int) -> torch.Tensor:
    """``F.one_hot(ids, num_classes).float()``; on the GPU written directly as float32 (two launches instead of
    the int64 one-hot, its fill and a conversion)."""
    ids = ids.view(-1).long()
    if not ids.is_cuda:
        return F.one_hot(ids, num_classes=num_classes).float()
    out = torch.zeros(ids.numel(), num_classes, dtype=torch.float32, device=ids.device)
    return out.scatter_(1, ids.view(-1, 1), 1.0)


def transform_graph(g: Graph, node_feature_dim: int, edge_feature_dim: int) -> Graph:
    """``utils/loaddata.py:257-260``: one-hot float attributes on a clone (the node attributes fresh on every
    call: the masking step writes into them)."""
    new_g = g.clone()
    new_g.ndata["attr"] = _one_hot_float(g.ndata["type"], node_feature_dim)
    new_g.edata["attr"] = _edge_one_hot(g, edge_feature_dim)
    new_g.mark_edge_attr_onehot()
    new_g.mark_node_attr_onehot()
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


# ------------------------------------------------------------------------------------- entity level
class _PlainUnpickler(pickle.Unpickler):
    """Unpickler for files that hold only containers, numbers and strings: any reference to a class or
    function (the only way a pickle can run code) is refused."""

    def find_class(self, module, name):
        raise pickle.UnpicklingError(f"global '{module}.{name}' is not allowed in a MAGIC data file")


def load_plain_pickle(path: str):
    with open(path, "rb") as f:
        return _PlainUnpickler(f).load()


def save_graph_npz(g: Graph, path: str):
    """int32 binary cache of one graph: edge list, node / edge types and, if built, the CSC / CSR arrays."""
    s = g.structure
    arrays = {"num_nodes": np.int64(s.num_nodes), "src": s.src.cpu().numpy().astype(np.int32),
              "dst": s.dst.cpu().numpy().astype(np.int32),
              "ntype": g.ndata["type"].cpu().numpy().astype(np.int32),
              "etype": g.edata["type"].cpu().numpy().astype(np.int32)}
    if s.has_csr():
        for k in ("in_ptr", "in_src", "in_eid", "out_ptr", "out_dst", "out_eid"):
            arrays[k] = getattr(s, k).cpu().numpy().astype(np.int32)
    np.savez(path, **arrays)


def load_graph_npz(path: str) -> Graph:
    with np.load(path, allow_pickle=False) as z:
        g = Graph(torch.from_numpy(z["src"].astype(np.int64)), torch.from_numpy(z["dst"].astype(np.int64)),
                  int(z["num_nodes"]))
        g.ndata["type"] = torch.from_numpy(z["ntype"].astype(np.int64))
        g.edata["type"] = torch.from_numpy(z["etype"].astype(np.int64))
        if "in_ptr" in z.files:
            g.structure.set_csr(*(torch.from_numpy(z[k]) for k in
                                  ("in_ptr", "in_src", "in_eid", "out_ptr", "out_dst", "out_eid")))
    return g


def preload_entity_level_dataset(path, root: str = "./data"):
    """``utils/loaddata.py:263-326``: on first use turn ``train.pkl`` / ``test.pkl`` / ``malicious.pkl``
    into ``metadata.json`` (same keys as the reference writes) and one ``train{i}.npz`` / ``test{i}.npz``
    per graph."""
    base = os.path.join(root, path)
    if os.path.exists(os.path.join(base, "metadata.json")) and os.path.exists(os.path.join(base, "train0.npz")):
        return
    train_gs = [graph_from_node_link(d) for d in load_plain_pickle(os.path.join(base, "train.pkl"))]
    test_gs = [graph_from_node_link(d) for d in load_plain_pickle(os.path.join(base, "test.pkl"))]
    malicious = load_plain_pickle(os.path.join(base, "malicious.pkl"))
    node_feature_dim = max(int(g.ndata["type"].max()) for g in train_gs + test_gs) + 1
    edge_feature_dim = max(int(g.edata["type"].max()) for g in train_gs + test_gs) + 1
    metadata = {"node_feature_dim": node_feature_dim, "edge_feature_dim": edge_feature_dim,
                "malicious": [list(malicious[0]), list(malicious[1])],
                "n_train": len(train_gs), "n_test": len(test_gs)}
    for kind, gs in (("train", train_gs), ("test", test_gs)):
        for i, g in enumerate(gs):
            save_graph_npz(g, os.path.join(base, f"{kind}{i}.npz"))
    with open(os.path.join(base, "metadata.json"), "w", encoding="utf-8") as f:
        json.dump(metadata, f)


def load_metadata(path, root: str = "./data"):
    """``utils/loaddata.py:328-344``."""
    preload_entity_level_dataset(path, root)
    with open(os.path.join(root, path, "metadata.json"), "r", encoding="utf-8") as f:
        return json.load(f)


def load_entity_level_dataset(path, t, n, root: str = "./data") -> Graph:
    """``utils/loaddata.py:346-363``: the n-th ``t`` ('train' | 'test') graph with one-hot attributes
    (the reference caches its graphs after ``transform_graph``)."""
    meta = load_metadata(path, root)
    g = load_graph_npz(os.path.join(root, path, f"{t}{n}.npz"))
    return transform_graph(g, meta["node_feature_dim"], meta["edge_feature_dim"])
