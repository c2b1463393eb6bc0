"""KNN outlier detector with the reference's API (``model/eval.py``).

Signatures, printed lines and metric tail follow the reference; the distance search --
scikit-learn's ``NearestNeighbors`` there (``model/eval.py:150-158``, ``:294-311``) -- runs on the
MI355X (``magic_amd.knn``).  With numpy inputs (the reference's call shape) standardisation is done
in numpy exactly as the reference does it, so the search sees bit-identical inputs; with embeddings
that already live on the GPU (``entity_level_distances`` on CUDA tensors) standardisation, both
searches and the row means stay on the device.

The result cache of the entity-level evaluator (``model/eval.py:298-320``) is kept, but as an
``.npz`` file that is read without unpickling and only accepted for the very arrays it was computed
from; the reference's ``.pkl`` cache files are never opened.
"""
from __future__ import annotations

import hashlib
import os
import random

import numpy as np
import torch

from ..knn import NearestNeighbors
from ..utils.utils import set_random_seed


def _metrics():
    from sklearn.metrics import precision_recall_curve, roc_auc_score
    return roc_auc_score, precision_recall_curve


def _confusion(y_test, score, best_thres):
    """Counts of the reference's Python loops (``model/eval.py:176-185``, ``:347-356``)."""
    y = np.asarray(y_test)
    hit = np.asarray(score) >= best_thres
    tp = int(np.sum((y == 1.0) & hit))
    fn = int(np.sum((y == 1.0) & ~hit))
    tn = int(np.sum((y == 0.0) & ~hit))
    fp = int(np.sum((y == 0.0) & hit))
    return tn, fn, tp, fp


def batch_level_evaluation(model, pooler, device, method, dataset, n_dim=0, e_dim=0, data=None):
    """``model/eval.py:17-83``.  ``data`` (a dict with 'dataset', 'full_index', 'n_feat') may be
    passed directly; otherwise it is loaded through ``utils.loaddata``."""
    from ..utils.loaddata import load_batch_level_dataset, transform_graph
    model.eval()
    x_list, y_list = [], []
    if data is None:
        data = load_batch_level_dataset(dataset)
    full, graphs = data["full_index"], data["dataset"]
    with torch.no_grad():
        for i in full:
            g = transform_graph(graphs[i][0], n_dim, e_dim).to(device)
            label = graphs[i][1]
            out = model.embed(g)
            if dataset != "wget":
                out = pooler(g, out).cpu().numpy()
            else:
                out = pooler(g, out, n_types=data["n_feat"]).cpu().numpy()
            y_list.append(label)
            x_list.append(out)
    x = np.concatenate(x_list, axis=0)
    y = np.array(y_list)
    if "knn" in method:
        return evaluate_batch_level_using_knn(100, dataset, x, y)
    raise NotImplementedError


def _batch_level_once(x, y, benign_idx, attack_idx, train_count, n_neighbors, eps):
    """One split of ``model/eval.py:139-159``; returns (y_test, score)."""
    x_train = x[benign_idx[:train_count]]
    x_test = np.concatenate([x[benign_idx[train_count:]], x[attack_idx]], axis=0)
    y_test = np.concatenate([y[benign_idx[train_count:]], y[attack_idx]], axis=0)
    x_train_mean = x_train.mean(axis=0)
    x_train_std = x_train.std(axis=0)
    x_train = (x_train - x_train_mean) / (x_train_std + eps)
    x_test = (x_test - x_train_mean) / (x_train_std + eps)
    nbrs = NearestNeighbors(n_neighbors=n_neighbors)
    nbrs.fit(x_train)
    distances, _ = nbrs.kneighbors(x_train, n_neighbors=n_neighbors)
    mean_distance = distances.mean() * n_neighbors / (n_neighbors - 1)
    distances, _ = nbrs.kneighbors(x_test, n_neighbors=n_neighbors)
    return y_test, distances.mean(axis=1) / mean_distance


def evaluate_batch_level_using_knn(repeat, dataset, embeddings, labels):
    """``model/eval.py:86-263``."""
    roc_auc_score, precision_recall_curve = _metrics()
    x, y = embeddings, labels
    train_count = 400 if dataset == "streamspot" else 100
    n_neighbors = min(int(train_count * 0.02), 10)
    benign_idx = np.where(y == 0)[0]
    attack_idx = np.where(y == 1)[0]
    if repeat != -1:
        prec_list, rec_list, f1_list = [], [], []
        tp_list, fp_list, tn_list, fn_list, auc_list = [], [], [], [], []
        for s in range(repeat):
            set_random_seed(s)
            np.random.shuffle(benign_idx)
            np.random.shuffle(attack_idx)
            y_test, score = _batch_level_once(x, y, benign_idx, attack_idx, train_count, n_neighbors, 1e-6)
            auc = roc_auc_score(y_test, score)
            prec, rec, threshold = precision_recall_curve(y_test, score)
            f1 = 2 * prec * rec / (rec + prec + 1e-9)
            max_f1_idx = np.argmax(f1)
            best_thres = threshold[max_f1_idx]
            prec_list.append(prec[max_f1_idx])
            rec_list.append(rec[max_f1_idx])
            f1_list.append(f1[max_f1_idx])
            tn, fn, tp, fp = _confusion(y_test, score, best_thres)
            tp_list.append(tp)
            fp_list.append(fp)
            fn_list.append(fn)
            tn_list.append(tn)
            auc_list.append(auc)
        print('AUC: {}+{}'.format(np.mean(auc_list), np.std(auc_list)))
        print('F1: {}+{}'.format(np.mean(f1_list), np.std(f1_list)))
        print('PRECISION: {}+{}'.format(np.mean(prec_list), np.std(prec_list)))
        print('RECALL: {}+{}'.format(np.mean(rec_list), np.std(rec_list)))
        print('TN: {}+{}'.format(np.mean(tn_list), np.std(tn_list)))
        print('FN: {}+{}'.format(np.mean(fn_list), np.std(fn_list)))
        print('TP: {}+{}'.format(np.mean(tp_list), np.std(tp_list)))
        print('FP: {}+{}'.format(np.mean(fp_list), np.std(fp_list)))
        return np.mean(auc_list), np.std(auc_list)
    set_random_seed(0)
    np.random.shuffle(benign_idx)
    np.random.shuffle(attack_idx)
    y_test, score = _batch_level_once(x, y, benign_idx, attack_idx, train_count, n_neighbors, 0.0)
    auc = roc_auc_score(y_test, score)
    prec, rec, threshold = precision_recall_curve(y_test, score)
    f1 = 2 * prec * rec / (rec + prec + 1e-9)
    best_idx = np.argmax(f1)
    best_thres = threshold[best_idx]
    tn, fn, tp, fp = _confusion(y_test, score, best_thres)
    print('AUC: {}'.format(auc))
    print('F1: {}'.format(f1[best_idx]))
    print('PRECISION: {}'.format(prec[best_idx]))
    print('RECALL: {}'.format(rec[best_idx]))
    print('TN: {}'.format(tn))
    print('FN: {}'.format(fn))
    print('TP: {}'.format(tp))
    print('FP: {}'.format(fp))
    return auc, 0.0


def _mean_k_distance(nbrs, x, n_neighbors):
    """Per-row mean of the k nearest distances, on the device."""
    if hasattr(nbrs, "mean_distances_device"):
        return nbrs.mean_distances_device(x, n_neighbors)
    return nbrs.kneighbors_device(x, n_neighbors)[0].mean(dim=1)


def standardise_device(x_train: torch.Tensor, x_test: torch.Tensor, sharded: bool = False, group=None):
    """``(x - mean) / std`` with the train columns' mean and (population) standard deviation
    (``model/eval.py:282-285``: no epsilon), statistics accumulated in float64 and rounded to float32
    once, the division in float32 as numpy does it.  With ``sharded`` the rows of ``x_train`` are one
    rank's shard and the column sums are all-reduced over ``group`` (two ``[d]`` float64 vectors)."""
    n = torch.tensor([float(x_train.shape[0])], dtype=torch.float64, device=x_train.device)
    s1 = x_train.sum(0, dtype=torch.float64)
    if sharded:
        import torch.distributed as dist
        dist.all_reduce(s1, group=group)
        dist.all_reduce(n, group=group)
    mean = (s1 / n).float()
    s2 = (x_train - mean).square_().sum(0, dtype=torch.float64)
    if sharded:
        dist.all_reduce(s2, group=group)
    std = (s2 / n).sqrt().float()
    return (x_train - mean) / std, (x_test - mean) / std


def _entity_level_distances_device(x_train, x_test, n_neighbors, nbrs=None, sample_idx=None):
    x_train, x_test = standardise_device(x_train.float(), x_test.float())
    if nbrs is None:
        nbrs = NearestNeighbors(n_neighbors=n_neighbors, device=x_train.device)
    nbrs.fit(x_train)
    if sample_idx is None:
        idx = list(range(x_train.shape[0]))
        random.shuffle(idx)                                     # model/eval.py:303-304
        sample_idx = idx[:min(50000, x_train.shape[0])]
    sample_idx = torch.as_tensor(sample_idx, dtype=torch.int64).to(x_train.device, non_blocking=True)
    mean_distance = _mean_k_distance(nbrs, x_train[sample_idx], n_neighbors).mean()
    return mean_distance, _mean_k_distance(nbrs, x_test, n_neighbors)


def entity_level_distances(x_train, x_test, n_neighbors, nbrs=None, sample_idx=None):
    """Hot lines of ``model/eval.py:282-313``: standardise, mean k-NN distance of at most
    50,000 shuffled train rows, per-test-row mean k-NN distance.

    numpy arrays in (the reference's call shape): numpy standardisation, numpy results.  CUDA tensors
    in: everything stays on the device and the results are float64 device tensors.  ``sample_idx``
    replaces the ``random.shuffle`` draw of the train rows behind ``mean_distance``."""
    if isinstance(x_train, torch.Tensor) and x_train.device.type == "cuda":
        return _entity_level_distances_device(x_train, x_test, n_neighbors, nbrs, sample_idx)
    x_train_mean = x_train.mean(axis=0)
    x_train_std = x_train.std(axis=0)
    x_train = (x_train - x_train_mean) / x_train_std
    x_test = (x_test - x_train_mean) / x_train_std
    if nbrs is None:
        nbrs = NearestNeighbors(n_neighbors=n_neighbors, n_jobs=-1)
    nbrs.fit(x_train)
    if sample_idx is None:
        idx = list(range(x_train.shape[0]))
        random.shuffle(idx)
        sample_idx = idx[:min(50000, x_train.shape[0])]
    distances, _ = nbrs.kneighbors(x_train[sample_idx], n_neighbors=n_neighbors)
    mean_distance = distances.mean()
    distances, _ = nbrs.kneighbors(x_test, n_neighbors=n_neighbors)
    return mean_distance, distances.mean(axis=1)


def _fingerprint(x_train, x_test, n_neighbors) -> str:
    """Identity of the arrays a cached result belongs to (shapes, dtypes, neighbour count, bytes)."""
    h = hashlib.sha256()
    for a in (x_train, x_test):
        a = np.ascontiguousarray(a)
        h.update(repr((a.shape, str(a.dtype), int(n_neighbors))).encode())
        h.update(memoryview(a).cast("B"))
    return h.hexdigest()


def _load_distance_cache(path, fingerprint, n_test):
    """(mean_distance, distances) from a cache this module wrote for exactly these inputs, else None.
    ``numpy.load`` without pickle support: nothing in the file is executed."""
    if not os.path.exists(path):
        return None
    try:
        with np.load(path, allow_pickle=False) as z:
            if str(z["fingerprint"]) != fingerprint or z["distances"].shape != (n_test,):
                return None
            return float(z["mean_distance"]), np.array(z["distances"], dtype=np.float64)
    except (OSError, ValueError, KeyError):
        return None


def _print_entity_metrics(m):
    """The lines ``model/eval.py:359-366`` prints (scraped by the reference's web wrapper)."""
    print('AUC: {}'.format(m["auc"]))
    print('F1: {}'.format(m["f1"]))
    print('PRECISION: {}'.format(m["precision"]))
    print('RECALL: {}'.format(m["recall"]))
    print('TN: {}'.format(m["tn"]))
    print('FN: {}'.format(m["fn"]))
    print('TP: {}'.format(m["tp"]))
    print('FP: {}'.format(m["fp"]))


def evaluate_entity_level_using_knn(dataset, x_train, x_test, y_test):
    """``model/eval.py:265-367``.

    numpy arrays in (the reference's call shape): numpy standardisation, the fingerprinted ``.npz``
    result cache, scikit-learn's curve functions.  CUDA tensors in: standardisation, both searches,
    the score and the whole metric tail (AUC, precision-recall curve, recall floor, confusion counts:
    ``model/metrics.py``) run on the device, nothing is cached, and only the printed numbers reach
    the host."""
    n_neighbors = 200 if dataset == 'cadets' else 10
    if isinstance(x_train, torch.Tensor) and x_train.device.type == "cuda":
        from .metrics import entity_level_detection_metrics
        mean_distance, distances = entity_level_distances(x_train, x_test, n_neighbors)
        m = entity_level_detection_metrics(dataset, torch.as_tensor(y_test), distances / mean_distance)
        _print_entity_metrics(m)
        return m["auc"], 0.0, None, None
    roc_auc_score, precision_recall_curve = _metrics()
    save_dict_path = './eval_result/distance_save_{}.npz'.format(dataset)
    fingerprint = _fingerprint(x_train, x_test, n_neighbors)
    cached = _load_distance_cache(save_dict_path, fingerprint, len(y_test))
    if cached is None:
        mean_distance, distances = entity_level_distances(x_train, x_test, n_neighbors)
        os.makedirs(os.path.dirname(save_dict_path), exist_ok=True)
        np.savez(save_dict_path, mean_distance=np.float64(mean_distance), distances=np.asarray(distances, np.float64),
                 fingerprint=np.str_(fingerprint))
    else:
        mean_distance, distances = cached
    score = distances / mean_distance
    del distances
    auc = roc_auc_score(y_test, score)
    prec, rec, threshold = precision_recall_curve(y_test, score)
    f1 = 2 * prec * rec / (rec + prec + 1e-9)
    best_idx = -1
    floors = {'trace': 0.99979, 'theia': 0.99996, 'cadets': 0.9976}
    for i in range(len(f1)):
        if dataset in floors and rec[i] < floors[dataset]:
            best_idx = i - 1
            break
    best_thres = threshold[best_idx]
    tn, fn, tp, fp = _confusion(y_test, score, best_thres)
    _print_entity_metrics(dict(auc=auc, f1=f1[best_idx], precision=prec[best_idx], recall=rec[best_idx],
                               tn=tn, fn=fn, tp=tp, fp=fp))
    return auc, 0.0, None, None


def entity_level_evaluation(model, dataset, device, metadata=None, load_graph=None):
    """The entity-level branch of the reference's evaluation driver (``eval.py:62-121``) without a
    host round trip: embed every train and test graph (``model.embed``), build the test set
    (``eval.py:99-114``: the last test graph's nodes plus every malicious node), score it against the
    train embeddings and print the metrics -- embeddings, search and metric tail stay on the GPU.

    ``metadata`` / ``load_graph(kind, i)`` default to ``utils.loaddata.load_metadata(dataset)`` and
    ``load_entity_level_dataset(dataset, kind, i)``.  Returns (test_auc, test_std)."""
    from ..utils import loaddata
    from .metrics import entity_level_test_set
    if metadata is None:
        metadata = loaddata.load_metadata(dataset)
    if load_graph is None:
        def load_graph(kind, i):
            return loaddata.load_entity_level_dataset(dataset, kind, i)
    model.eval()
    malicious, _ = metadata['malicious']
    n_train, n_test = metadata['n_train'], metadata['n_test']
    with torch.no_grad():
        x_train = torch.cat([model.embed(load_graph('train', i).to(device)) for i in range(n_train)], 0)
        skip_benign = 0
        x_test = []
        for i in range(n_test):
            g = load_graph('test', i).to(device)
            if i != n_test - 1:
                skip_benign += g.number_of_nodes()
            x_test.append(model.embed(g))
            del g
        x_test, y_test = entity_level_test_set(torch.cat(x_test, 0), malicious, skip_benign)
        test_auc, test_std, _, _ = evaluate_entity_level_using_knn(dataset, x_train, x_test, y_test)
    print(f"#Test_AUC: {test_auc:.4f}±{test_std:.4f}")
    return test_auc, test_std
