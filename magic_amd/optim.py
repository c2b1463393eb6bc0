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

    State per parameter, as torch's Adam names it: ``exp_avg``, ``exp_avg_sq`` and ``step``.  ``step`` is an int64
    device tensor that parameters which have always stepped together SHARE (one tensor and one launch for a whole
    model in the usual case); a parameter whose count differs -- its first gradient arrived later, or it sat out a
    step -- gets a tensor of its own, so every parameter is bias-corrected with its own count as in torch."""

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
            fresh, by_counter = [], {}
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
                row = (p, g, st["exp_avg"], st["exp_avg_sq"])
                if st["step"] is None:
                    fresh.append(row)
                else:
                    by_counter.setdefault(id(st["step"]), (st["step"], []))[1].append(row)
            if not fresh and not by_counter:
                continue
            # a parameter that sits this step out must not be counted with the ones it shares a tensor with
            for p in group["params"]:
                if p.grad is None:
                    c = self.state.get(p, {}).get("step")
                    if c is not None and id(c) in by_counter:
                        self.state[p]["step"] = c.clone()
            if fresh:
                counter = torch.zeros(1, dtype=torch.int64, device=fresh[0][0].device)
                for p, *_ in fresh:
                    self.state[p]["step"] = counter
                by_counter[id(counter)] = (counter, fresh)
            b1, b2 = group["betas"]
            for counter, rows in by_counter.values():
                _lib.adam_step(rows, counter, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                               float(group["weight_decay"]))
        return loss

    def load_state_dict(self, state_dict):
        """Accepts what this class and what ``torch.optim.Adam`` write (``step`` as a float or integer tensor on
        any device, or a number): every count becomes an int64 tensor on its parameter's device, one tensor per
        distinct count and group."""
        super().load_state_dict(state_dict)
        for group in self.param_groups:
            shared = {}
            for p in group["params"]:
                st = self.state.get(p)
                if not st or st.get("step") is None:
                    continue
                count = int(round(float(st["step"])))
                key = (count, p.device)
                if key not in shared:
                    shared[key] = torch.full((1,), count, dtype=torch.int64, device=p.device)
                st["step"] = shared[key]
                for name in ("exp_avg", "exp_avg_sq"):
                    st[name] = st[name].to(device=p.device, dtype=torch.float32).contiguous()
