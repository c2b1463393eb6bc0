"""Graph read-out (``utils/poolers.py``): mean / sum / max over nodes, optionally per node type."""
import torch
import torch.nn as nn


class Pooling(nn.Module):
    def __init__(self, pooler):
        super().__init__()
        self.pooler = pooler

    def forward(self, graph, feat, n_types=None):
        """``utils/poolers.py:48-73``.  With ``n_types`` the result is the concatenation of the
        per-type read-outs ``[1, n_types * hidden]`` (zeros for an absent type)."""
        with graph.local_scope():
            if not n_types:
                if self.pooler == "mean":
                    return feat.mean(0, keepdim=True)
                if self.pooler == "sum":
                    return feat.sum(0, keepdim=True)
                if self.pooler == "max":
                    return feat.max(0, keepdim=True)
                raise NotImplementedError
            result = []
            types = graph.ndata["type"]
            for i in range(n_types):
                mask = (types == i)
                if not mask.any():
                    result.append(torch.zeros((1, feat.shape[-1]), device=feat.device))
                elif self.pooler == "mean":
                    result.append(feat[mask].mean(0, keepdim=True))
                elif self.pooler == "sum":
                    result.append(feat[mask].sum(0, keepdim=True))
                elif self.pooler == "max":
                    result.append(feat[mask].max(0, keepdim=True))
                else:
                    raise NotImplementedError
            return torch.cat(result, dim=-1)
