"""KNN outlier detector with the reference's API (``model/eval.py``).

Signatures, printed lines, cache file and metric tail follow the reference; the distance
search -- scikit-learn's ``NearestNeighbors`` there (``model/eval.py:150-158``, ``:294-311``)
-- runs on the MI355X (``magic_amd.knn``).  Standardisation is done in numpy exactly as
the reference does it, so the search sees bit-identical inputs.
"""
from __future__ import annotations

import os
import pickle as pkl
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


def entity_level_distances(x_train, x_test, n_neighbors, nbrs=None):
    """Hot lines of ``model/eval.py:282-313``: standardise, mean k-NN distance of at most
    50,000 shuffled train rows, per-test-row mean k-NN distance."""
    x_train_mean = x_train.mean(axis=0)
    x_train_std = x_train.std(axis=0)
    x_train = (x_train - x_train_mean) / x_train_std
    x_test = (x_test - x_train_mean) / x_train_std
    if nbrs is None:
        nbrs = NearestNeighbors(n_neighbors=n_neighbors, n_jobs=-1)
    nbrs.fit(x_train)
    idx = list(range(x_train.shape[0]))
    random.shuffle(idx)
    distances, _ = nbrs.kneighbors(x_train[idx][:min(50000, x_train.shape[0])], n_neighbors=n_neighbors)
    mean_distance = distances.mean()
    distances, _ = nbrs.kneighbors(x_test, n_neighbors=n_neighbors)
    return mean_distance, distances.mean(axis=1)


def evaluate_entity_level_using_knn(dataset, x_train, x_test, y_test):
    """``model/eval.py:265-367``."""
    roc_auc_score, precision_recall_curve = _metrics()
    n_neighbors = 200 if dataset == 'cadets' else 10
    save_dict_path = './eval_result/distance_save_{}.pkl'.format(dataset)
    if not os.path.exists(save_dict_path):
        mean_distance, distances = entity_level_distances(x_train, x_test, n_neighbors)
        os.makedirs(os.path.dirname(save_dict_path), exist_ok=True)
        with open(save_dict_path, 'wb') as f:
            pkl.dump([mean_distance, distances], f)
    else:
        with open(save_dict_path, 'rb') as f:  # a cache this function wrote itself
            mean_distance, distances = pkl.load(f)
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
    print('AUC: {}'.format(auc))
    print('F1: {}'.format(f1[best_idx]))
    print('PRECISION: {}'.format(prec[best_idx]))
    print('RECALL: {}'.format(rec[best_idx]))
    print('TN: {}'.format(tn))
    print('FN: {}'.format(fn))
    print('TP: {}'.format(tp))
    print('FP: {}'.format(fp))
    return auc, 0.0, None, None
