"""Adam on the library's multi-tensor kernel (``magic_adam_step``).

The reference trains with ``torch.optim.Adam(model.parameters(), lr, weight_decay)``
(``utils/utils.py:24-31``): the L2 term goes into the gradient, no amsgrad.  :class:`FusedAdam` is that update
rule as one launch over all parameter tensors with the step count kept on the device, so the step can be recorded
into a HIP graph; parameters, gradients and moments must be float32 CUDA tensors.
"""
from __future__ import annotations

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    """``torch.optim.Adam(params, lr, betas, eps, weight_decay)`` (no amsgrad / maximize) for CUDA parameters.

    State per parameter, as torch's Adam names it: ``exp_avg``, ``exp_avg_sq`` and ``step`` -- the latter one
    int64 device tensor shared by the parameters of a group."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            rows, counter = [], None
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise _lib.MagicHipError("FusedAdam updates contiguous float32 parameters on the GPU only")
                if g.is_sparse or g.dtype != torch.float32 or g.device != p.device:
                    raise _lib.MagicHipError("FusedAdam needs dense float32 gradients on the parameter's device")
                if not g.is_contiguous():
                    g = g.contiguous()
                st = self.state[p]
                if not st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["step"] = None
                rows.append((p, g, st["exp_avg"], st["exp_avg_sq"]))
                if st["step"] is not None and counter is None:
                    counter = st["step"]
            if not rows:
                continue
            if counter is None:
                counter = torch.zeros(1, dtype=torch.int64, device=rows[0][0].device)
            for p, *_ in rows:
                if self.state[p]["step"] is None:
                    self.state[p]["step"] = counter
            b1, b2 = group["betas"]
            _lib.adam_step(rows, counter, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                           float(group["weight_decay"]))
        return loss
