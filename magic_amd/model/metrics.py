"""Detection tail of the entity-level evaluator on the device (``model/eval.py:323-357`` and the
test-set construction of ``eval.py:99-114`` of the reference).

The reference hands the scores to scikit-learn (``roc_auc_score``, ``precision_recall_curve``) and
counts the confusion matrix in a Python loop over every test node.  Here the scores stay on the GPU:
one descending sort, prefix sums of the labels, and the curves read off at the distinct score values
-- the same definitions as scikit-learn's ``_binary_clf_curve`` -- so that only the handful of printed
numbers travels to the host.  Written with torch's device sort / scan primitives; the functions take
tensors on any device, which is how the CPU suite checks them against scikit-learn itself.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

# recall floors below which the reference stops walking the precision-recall curve (model/eval.py:332-340)
RECALL_FLOOR = {"trace": 0.99979, "theia": 0.99996, "cadets": 0.9976}


def binary_clf_curve(y_true: torch.Tensor, score: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(fps, tps, thresholds) at every distinct score, thresholds descending: the counts of false /
    true positives among the samples with ``score >= threshold`` (scikit-learn ``_binary_clf_curve``)."""
    score = score.reshape(-1).double()
    y = y_true.reshape(-1).double()
    order = torch.sort(score, descending=True, stable=True).indices
    s, y = score[order], y[order]
    last_of_run = torch.ones_like(s, dtype=torch.bool)
    last_of_run[:-1] = s[1:] != s[:-1]
    idx = torch.nonzero(last_of_run).reshape(-1)
    tps = torch.cumsum(y, 0)[idx]
    fps = (idx + 1).double() - tps
    return fps, tps, s[idx]


def roc_auc(y_true: torch.Tensor, score: torch.Tensor) -> torch.Tensor:
    """Area under the ROC curve (``sklearn.metrics.roc_auc_score`` for binary labels), a 0-dim float64
    tensor on the scores' device."""
    fps, tps, _ = binary_clf_curve(y_true, score)
    if float(tps[-1]) == 0.0 or float(fps[-1]) == 0.0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    zero = torch.zeros(1, dtype=torch.float64, device=fps.device)
    fpr = torch.cat([zero, fps / fps[-1]])
    tpr = torch.cat([zero, tps / tps[-1]])
    return torch.trapezoid(tpr, fpr)


def precision_recall_curve(y_true: torch.Tensor, score: torch.Tensor):
    """(precision, recall, thresholds) as ``sklearn.metrics.precision_recall_curve`` returns them:
    thresholds ascending, ``precision[i]`` / ``recall[i]`` for ``score >= thresholds[i]``, one final
    point (1, 0) without a threshold."""
    fps, tps, thr = binary_clf_curve(y_true, score)
    ps = tps + fps
    precision = torch.where(ps != 0, tps / ps.clamp(min=1.0), torch.zeros_like(ps))
    recall = tps / tps[-1] if float(tps[-1]) != 0.0 else torch.ones_like(tps)
    one = torch.ones(1, dtype=torch.float64, device=fps.device)
    return (torch.cat([precision.flip(0), one]), torch.cat([recall.flip(0), one * 0.0]), thr.flip(0))


def entity_level_detection_metrics(dataset: str, y_test: torch.Tensor, score: torch.Tensor) -> Dict[str, float]:
    """The numbers ``model/eval.py:327-366`` prints: AUC; then, at the last threshold before the recall
    drops below the dataset's floor (the last threshold of the curve for other datasets), F1, precision,
    recall and the confusion counts with ``score >= threshold`` as the positive call."""
    y = y_test.reshape(-1).to(score.device).double()
    score = score.reshape(-1).double()
    auc = roc_auc(y, score)
    prec, rec, thr = precision_recall_curve(y, score)
    f1 = 2 * prec * rec / (rec + prec + 1e-9)
    best_idx = -1
    floor: Optional[float] = RECALL_FLOOR.get(dataset)
    if floor is not None:
        below = rec < floor
        if bool(below.any()):
            best_idx = int(torch.nonzero(below)[0]) - 1      # first i with rec[i] < floor, minus one
    best_thres = thr[best_idx]
    hit = score >= best_thres
    pos = y == 1.0
    neg = y == 0.0
    counts = torch.stack([(neg & ~hit).sum(), (pos & ~hit).sum(), (pos & hit).sum(), (neg & hit).sum()])
    vals = torch.stack([auc, f1[best_idx], prec[best_idx], rec[best_idx]]).cpu().tolist()   # one copy to the host
    tn, fn, tp, fp = counts.cpu().tolist()
    return {"auc": vals[0], "f1": vals[1], "precision": vals[2], "recall": vals[3],
            "tn": int(tn), "fn": int(fn), "tp": int(tp), "fp": int(fp), "threshold": float(best_thres)}


def entity_level_test_set(x_test: torch.Tensor, malicious, skip_benign: int):
    """``eval.py:99-114``: label the malicious node ids 1, keep the nodes of the last test graph (index
    >= ``skip_benign``) plus every malicious node; returns (x_test[kept], y_test[kept])."""
    n = x_test.shape[0]
    y = torch.zeros(n, dtype=torch.float64, device=x_test.device)
    mal = torch.as_tensor(malicious, dtype=torch.int64).to(x_test.device)
    if mal.numel():
        y[mal] = 1.0
    keep = (torch.arange(n, device=x_test.device) >= skip_benign) | (y == 1.0)
    return x_test[keep], y[keep]
