"""Scaled-cosine-error loss (``model/loss_func.py:16-32``) on the fused HIP kernel."""
from __future__ import annotations

import torch

from .. import _lib, ops


class _SCEMaskedFn(torch.autograd.Function):
    """loss = mean_i (1 - cos(x[mask_i], target_i))^alpha with the gradient produced in the
    same kernel launch (``csrc/sce.hip``).  ``target`` is either ``[N, n]`` (row ``mask_i`` is
    the target of masked row i) or a single ``[1, n]`` row shared by all masked rows."""

    @staticmethod
    def forward(ctx, x, target, mask_idx, alpha, n_dim):
        single = target.dim() == 2 and target.shape[0] == 1 and x.shape[0] != 1
        if x.stride(-1) != 1:
            x = x.contiguous()
        if not single and target.stride(-1) != 1:
            target = target.contiguous()
        loss, d_x, d_t = ops.sce_fwd_bwd(x, n_dim, target, single, mask_idx, alpha)
        ctx.single = single
        ctx.save_for_backward(d_x, d_t)
        ctx.t_shape = target.shape
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        d_x, d_t = ctx.saved_tensors
        gx = d_x * g if ctx.needs_input_grad[0] else None
        gt = None
        if ctx.needs_input_grad[1]:
            if not ctx.single:
                raise NotImplementedError("gradient w.r.t. a per-row SCE target is not needed by MAGIC")
            gt = (d_t * g).view(ctx.t_shape)
        return gx, gt, None, None, None


def sce_loss_masked(x, target, mask_idx, alpha=3, n_dim=None):
    """SCE over the rows ``mask_idx`` of ``x`` without materialising the gathered rows.  With
    ``n_dim`` only the first ``n_dim`` columns of ``x`` (and of the target) are features, the rest
    is padding whose gradient is zero."""
    return _SCEMaskedFn.apply(x, target, mask_idx, float(alpha), int(n_dim if n_dim is not None else x.shape[1]))


def sce_loss(x, y, alpha=3):
    """Reference signature (``model/loss_func.py:16``): SCE between row-aligned ``x`` and ``y``."""
    if x.device.type != "cuda":
        raise _lib.MagicHipError("magic_amd.sce_loss runs on the MI355X only (no CPU fallback)")
    idx = torch.arange(x.shape[0], device=x.device)
    if y.requires_grad:
        raise NotImplementedError("gradient w.r.t. a per-row SCE target is not needed by MAGIC")
    return _SCEMaskedFn.apply(x, y, idx, float(alpha), x.shape[1])
