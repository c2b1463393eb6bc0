"""KNN outlier-scoring oracle.  TEST INFRASTRUCTURE ONLY.

Restates the hot lines of ``model/eval.py`` (``:139-159`` batch level, ``:282-323``
entity level).  The distance search there is ``sklearn.neighbors.NearestNeighbors``
(``requirements.txt:6`` pins scikit-learn 1.2.2; it is not part of the reference
tree): brute force for d >= 16, squared distances expanded as |x|^2 - 2xy + |y|^2 in
float64, k smallest per query, ascending, then sqrt.  ``exact_topk`` below is the
direct float64 (x - y)^2 form of the same definition; on tie-free data its indices
equal sklearn's, its distances agree to ~1e-6 abs (sklearn's expansion leaves
~1e-7 instead of 0 for a point's distance to itself).

Pinned by ``tests/golden/knn_*.npz``: outputs of the reference's own
``evaluate_entity_level_using_knn`` / ``evaluate_batch_level_using_knn`` captured in
the authoring container (``tests/golden/make_golden.py``).
"""
import random

import numpy as np


def exact_topk(train: np.ndarray, query: np.ndarray, k: int, block: int = 512):
    """k nearest train rows per query by exact float64 Euclidean distance.
    Ties are ordered by ascending train index.  Returns (dist[Q,k] f64, idx[Q,k] i64)."""
    t = np.asarray(train, dtype=np.float64)
    q = np.asarray(query, dtype=np.float64)
    Q = q.shape[0]
    dist = np.empty((Q, k), np.float64)
    idx = np.empty((Q, k), np.int64)
    for s in range(0, Q, block):
        qb = q[s:s + block]
        d2 = ((qb[:, None, :] - t[None, :, :]) ** 2).sum(-1) if t.shape[0] * qb.shape[0] * t.shape[1] < 2e8 \
            else np.maximum((qb * qb).sum(1)[:, None] - 2.0 * qb @ t.T + (t * t).sum(1)[None, :], 0.0)
        order = np.lexsort((np.broadcast_to(np.arange(t.shape[0]), d2.shape), d2), axis=1)[:, :k]
        idx[s:s + block] = order
        dist[s:s + block] = np.sqrt(np.take_along_axis(d2, order, 1))
    return dist, idx


def standardize_entity(x_train, x_test):
    """``model/eval.py:282-285``: column mean / std of the train set, no epsilon, in the
    input dtype (float32 embeddings stay float32)."""
    m = x_train.mean(axis=0)
    s = x_train.std(axis=0)
    return (x_train - m) / s, (x_test - m) / s


def standardize_batch(x_train, x_test):
    """``model/eval.py:144-147`` (repeat mode): epsilon 1e-6 on the std."""
    m = x_train.mean(axis=0)
    s = x_train.std(axis=0)
    return (x_train - m) / (s + 1e-6), (x_test - m) / (s + 1e-6)


def entity_scores(x_train, x_test, k, topk=exact_topk, shuffle_seed=None):
    """``model/eval.py:282-323`` without the metric tail: returns
    (mean_distance, per-test-row mean k-distance, score).  ``random.shuffle`` of the
    train index list uses Python's global RNG exactly like the reference (``:303-306``);
    seed it before the call for a reproducible subset."""
    x_train, x_test = standardize_entity(x_train, x_test)
    idx = list(range(x_train.shape[0]))
    if shuffle_seed is not None:
        random.seed(shuffle_seed)
    random.shuffle(idx)
    sub = x_train[idx][:min(50000, x_train.shape[0])]
    d, _ = topk(x_train, sub, k)
    mean_distance = d.mean()
    d, _ = topk(x_train, x_test, k)
    dm = d.mean(axis=1)
    return mean_distance, dm, dm / mean_distance


def batch_scores(x_train, x_test, k, topk=exact_topk):
    """``model/eval.py:144-159``: self-corrected mean distance ``mean * k / (k - 1)``."""
    x_train, x_test = standardize_batch(x_train, x_test)
    d, _ = topk(x_train, x_train, k)
    mean_distance = d.mean() * k / (k - 1)
    d, _ = topk(x_train, x_test, k)
    return mean_distance, d.mean(axis=1) / mean_distance
