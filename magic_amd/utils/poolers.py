"""Graph read-out with the reference's interface (``utils/poolers.py``), on the MI355X.

``Pooling(pooler)(graph, feat, n_types=None)``: mean or sum of the node embeddings, over all
nodes (``[1, hidden]``) or, with ``n_types``, per value of ``graph.ndata['type']`` with the per-type
rows side by side (``[1, n_types * hidden]``, zeros for a type without nodes) -- what
``model/eval.py:64-68`` feeds the batch-level detector (Unicorn-Wget: 8 x 256).  One segmented pass
over the embeddings (``magic_readout``): no Python loop over the types and no host round trip per
type.
"""
import torch
import torch.nn as nn

from .. import _lib, ops  # noqa: F401  (ops registers torch.ops.magic.*)


class Pooling(nn.Module):
    def __init__(self, pooler):
        super().__init__()
        if pooler not in ("mean", "sum", "max"):
            raise NotImplementedError(pooler)
        self.pooler = pooler

    def forward(self, graph, feat, n_types=None):
        if self.pooler == "max":
            # utils/poolers.py:54-55,69 returns torch's (values, indices) pair there, which none of its
            # callers can use; no kernel is provided for it
            raise NotImplementedError("'max' read-out: the reference returns a (values, indices) tuple "
                                      "its callers cannot consume; use 'mean' or 'sum'")
        if feat.device.type != "cuda":
            raise _lib.MagicHipError("magic_amd.Pooling runs on the MI355X only (no CPU fallback)")
        types = graph.ndata["type"] if n_types else None
        return torch.ops.magic.readout(feat, types, int(n_types or 0), self.pooler == "mean")
