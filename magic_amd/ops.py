"""``torch.ops.magic.*``: the C ABI of ``libmagic_hip.so`` registered as PyTorch custom operators.

Every operator has a schema, a CUDA (= ROCm) implementation that forwards raw device pointers to the
library through ``magic_amd._lib``, and a fake (meta) implementation that only computes output
shapes, so the operators can be traced / shape-checked without a GPU.  There is no CPU
implementation: calling one with CPU tensors raises PyTorch's own "no kernel for backend CPU" error.
Operators whose results land in caller-provided buffers carry a trailing underscore and declare those
arguments as mutated (``Tensor(a!)``).  Absent optional outputs come back as empty tensors.

Differentiable entry points built on them live next to the reference-shaped modules
(``model/gat.py``, ``model/fused.py``, ``model/loss_func.py``); two operators also carry a registered
autograd formula so that they can be used on their own: ``magic::sce_loss`` and
``magic::linear_nobias``.

The segment plan of a graph direction (``graph.SegmentPlan``) travels as eight int32 tensors:
``ptr, heavy_node, heavy_cptr, chunk_node, chunk_beg, chunk_end, order, xcd_start``.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import _lib

_L = torch.library.Library("magic", "DEF")
_PLAN = "Tensor {0}ptr, Tensor {0}heavy_node, Tensor {0}heavy_cptr, Tensor {0}chunk_node, Tensor {0}chunk_beg, " \
        "Tensor {0}chunk_end, Tensor {0}order, Tensor {0}xcd_start"


def _register(name: str, schema: str, impl, fake):
    _L.define(f"{name}{schema}")
    _L.impl(name, impl, "CUDA")
    torch.library.register_fake(f"magic::{name}", fake)


def _none(t: Optional[Tensor]) -> Optional[Tensor]:
    return None if t is None or t.numel() == 0 else t


def _empty(like: Tensor, dtype=torch.float32) -> Tensor:
    return torch.empty(0, dtype=dtype, device=like.device)


def _plan(ptr, heavy_node, heavy_cptr, chunk_node, chunk_beg, chunk_end, order, xcd_start):
    from .graph import SegmentPlan
    return SegmentPlan(ptr, heavy_node, heavy_cptr, chunk_node, chunk_beg, chunk_end,
                       order=order if order.numel() else None, xcd_start=xcd_start if xcd_start.numel() else None)


def plan_tensors(plan) -> List[Tensor]:
    """The eight tensors of a ``graph.SegmentPlan`` in operator-argument order."""
    none = torch.empty(0, dtype=torch.int32, device=plan.ptr.device)
    order = plan.order if plan.order is not None else none
    xcd_start = plan.xcd_start if plan.xcd_start is not None else none
    return [plan.ptr, plan.heavy_node, plan.heavy_cptr, plan.chunk_node, plan.chunk_beg, plan.chunk_end, order,
            xcd_start]


# ------------------------------------------------------------------------------------- csr_build
def _csr_build(src: Tensor, dst: Tensor, num_nodes: int):
    return _lib.csr_build(src, dst, num_nodes)


def _csr_build_fake(src, dst, num_nodes):
    e = src.shape[0]
    i32 = dict(dtype=torch.int32, device=src.device)
    return (torch.empty(num_nodes + 1, **i32), torch.empty(e, **i32), torch.empty(e, **i32),
            torch.empty(num_nodes + 1, **i32), torch.empty(e, **i32), torch.empty(e, **i32))


_register("csr_build", "(Tensor src, Tensor dst, int num_nodes) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)",
          _csr_build, _csr_build_fake)


_register("negative_filter", "(Tensor out_ptr, Tensor out_dst, Tensor in_ptr, Tensor in_src, Tensor u, Tensor v, "
          "int num_samples) -> Tensor",
          lambda out_ptr, out_dst, in_ptr, in_src, u, v, num_samples:
          _lib.negative_filter(out_ptr, out_dst, in_ptr, in_src, u, v, num_samples),
          lambda out_ptr, out_dst, in_ptr, in_src, u, v, num_samples: torch.empty_like(u, dtype=torch.bool))


# ------------------------------------------------------------------------------------- projections
def _proj_fwd(x, w, out_c: int, n_score: int, res_c: int, x_mask, x_scale: float, attn_src, attn_dst):
    hs, score, res = _lib.proj_fwd(x, w, out_c, n_score, res_c, x_mask=x_mask, x_scale=x_scale, attn_src=attn_src,
                                   attn_dst=attn_dst)
    return hs, score if score is not None else _empty(x), res if res is not None else _empty(x)


def _proj_fwd_fake(x, w, out_c, n_score, res_c, x_mask, x_scale, attn_src, attn_dst):
    n = x.shape[0]
    return (x.new_empty(n, out_c), x.new_empty((n, n_score) if n_score else (0,)),
            x.new_empty((n, res_c) if res_c else (0,)))


_register("proj_fwd", "(Tensor x, Tensor w, int out_c, int n_score, int res_c, Tensor? x_mask, float x_scale, "
          "Tensor? attn_src, Tensor? attn_dst) -> (Tensor, Tensor, Tensor)", _proj_fwd, _proj_fwd_fake)


def _proj_bwd_input(dy, w, add, out_mask, out_scale: float, add2, act_out, act_pre, prelu_w, want_col_sum: bool):
    dx, cs, pg = _lib.proj_bwd_input(dy, w, add=add, out_mask=out_mask, out_scale=out_scale, add2=add2,
                                     act_out=act_out, act_pre=act_pre, prelu_w=prelu_w, want_col_sum=want_col_sum)
    return dx, cs if cs is not None else _empty(dy), pg if pg is not None else _empty(dy)


def _proj_bwd_input_fake(dy, w, add, out_mask, out_scale, add2, act_out, act_pre, prelu_w, want_col_sum):
    f = w.shape[1]
    return (dy.new_empty(dy.shape[0], f), dy.new_empty(f if want_col_sum else 0),
            dy.new_empty(1 if (want_col_sum and act_out is not None) else 0))


_EPI = "Tensor? add, Tensor? out_mask, float out_scale, Tensor? add2, Tensor? act_out, Tensor? act_pre, " \
       "Tensor? prelu_w, bool want_col_sum"
_register("proj_bwd_input", f"(Tensor dy, Tensor w, {_EPI}) -> (Tensor, Tensor, Tensor)",
          _proj_bwd_input, _proj_bwd_input_fake)


def _proj_bwd_input_(dy, w, add, out_mask, out_scale: float, add2, act_out, act_pre, prelu_w, want_col_sum: bool, out):
    _, cs, pg = _lib.proj_bwd_input(dy, w, add=add, out_mask=out_mask, out_scale=out_scale, add2=add2,
                                    act_out=act_out, act_pre=act_pre, prelu_w=prelu_w, want_col_sum=want_col_sum,
                                    out=out)
    return cs if cs is not None else _empty(dy), pg if pg is not None else _empty(dy)


def _proj_bwd_input__fake(dy, w, add, out_mask, out_scale, add2, act_out, act_pre, prelu_w, want_col_sum, out):
    f = w.shape[1]
    return dy.new_empty(f if want_col_sum else 0), dy.new_empty(1 if (want_col_sum and act_out is not None) else 0)


_register("proj_bwd_input_", f"(Tensor dy, Tensor w, {_EPI}, Tensor(a!) out) -> (Tensor, Tensor)",
          _proj_bwd_input_, _proj_bwd_input__fake)

_register("proj_bwd_weight", "(Tensor dy, Tensor x, Tensor? x_mask, float x_scale) -> Tensor",
          lambda dy, x, x_mask, x_scale: _lib.proj_bwd_weight(dy, x, x_mask=x_mask, x_scale=x_scale),
          lambda dy, x, x_mask, x_scale: dy.new_empty(dy.shape[1], x.shape[1]))


# ------------------------------------------------------------------------------------- GAT edge phase
def _struct_in(n_nodes, plan7, in_src, in_dst):
    return SimpleNamespace(num_nodes=n_nodes, num_edges=int(in_src.numel()), plan_in=_plan(*plan7), in_src=in_src,
                           in_dst=in_dst)


def _gat_fwd_(p0, p1, p2, p3, p4, p5, p6, p7, in_src, in_dst, n_heads: int, head_dim: int, hs, eh, et, ee_table,
              etype_csc, ee_csc, slope: float, bias, res, prelu_w, want_pre: bool, res_mask, res_scale: float,
              want_attn: bool, want_stat: bool, dst_active, out):
    s = _struct_in(out.shape[0], (p0, p1, p2, p3, p4, p5, p6, p7), in_src, in_dst)
    _, pre, attn, stat = _lib.gat_fwd(s, n_heads, head_dim, hs, eh, et, ee_table, etype_csc, ee_csc, slope, bias, res,
                                      prelu_w, want_pre, res_mask=res_mask, res_scale=res_scale, out=out,
                                      want_attn=want_attn, want_stat=want_stat, dst_active=dst_active)
    e = _empty(hs)
    return pre if pre is not None else e, attn if attn is not None else e, stat if stat is not None else e


def _gat_fwd__fake(p0, p1, p2, p3, p4, p5, p6, p7, in_src, in_dst, n_heads, head_dim, hs, eh, et, ee_table, etype_csc,
                   ee_csc, slope, bias, res, prelu_w, want_pre, res_mask, res_scale, want_attn, want_stat, dst_active,
                   out):
    n, c = out.shape[0], n_heads * head_dim
    return (hs.new_empty((n, c) if want_pre else (0,)),
            hs.new_empty((in_src.shape[0], n_heads) if want_attn else (0,)),
            hs.new_empty((n, n_heads, 2) if want_stat else (0,)))


_register("gat_fwd_", "(" + _PLAN.format("in_") + ", Tensor in_src, Tensor in_dst, int n_heads, int head_dim, "
          "Tensor hs, Tensor eh, Tensor et, Tensor? ee_table, Tensor? etype_csc, Tensor? ee_csc, float slope, "
          "Tensor? bias, Tensor? res, Tensor? prelu_w, bool want_pre, Tensor? res_mask, float res_scale, "
          "bool want_attn, bool want_stat, Tensor? dst_active, Tensor(a!) out) -> (Tensor, Tensor, Tensor)",
          _gat_fwd_, _gat_fwd__fake)


def _gat_bwd_(i0, i1, i2, i3, i4, i5, i6, i7, o0, o1, o2, o3, o4, o5, o6, o7, in_pos, out_dst, out_pos, n_heads: int,
              head_dim: int, hs, eh, stat, ee_table, etype_csr, ee_csc, e_dim: int, slope: float, pre, act_out,
              prelu_w, d_out, d_pre, d_y, want_bias: bool, want_dlogit: bool, dst_active):
    s = SimpleNamespace(num_nodes=hs.shape[0], num_edges=int(out_dst.numel()), plan_in=_plan(i0, i1, i2, i3, i4, i5, i6, i7),
                        plan_out=_plan(o0, o1, o2, o3, o4, o5, o6, o7), in_pos=in_pos, out_dst=out_dst, out_pos=out_pos)
    d_table, d_logit, d_bias, d_prelu = _lib.gat_bwd(s, n_heads, head_dim, hs, eh, stat, ee_table, etype_csr, ee_csc,
                                                     e_dim, slope, pre, act_out, prelu_w, d_out, d_pre, d_y,
                                                     want_bias, want_dlogit, dst_active=dst_active)
    e = _empty(hs)
    return (d_table if d_table is not None else e, d_logit if d_logit is not None else e,
            d_bias if d_bias is not None else e, d_prelu if d_prelu is not None else e)


def _gat_bwd__fake(i0, i1, i2, i3, i4, i5, i6, i7, o0, o1, o2, o3, o4, o5, o6, o7, in_pos, out_dst, out_pos, n_heads,
                   head_dim, hs, eh, stat, ee_table, etype_csr, ee_csc, e_dim, slope, pre, act_out, prelu_w, d_out,
                   d_pre, d_y, want_bias, want_dlogit, dst_active):
    return (hs.new_empty((e_dim, n_heads) if ee_table is not None else (0,)),
            hs.new_empty((out_dst.shape[0], n_heads) if want_dlogit else (0,)),
            hs.new_empty(n_heads * head_dim if want_bias else 0), hs.new_empty(1 if prelu_w is not None else 0))


# d_pre may be the very tensor passed as d_out when prelu_w is None: both are declared mutable
_register("gat_bwd_", "(" + _PLAN.format("in_") + ", " + _PLAN.format("out_") + ", Tensor in_pos, Tensor out_dst, "
          "Tensor out_pos, int n_heads, int head_dim, Tensor hs, Tensor eh, Tensor stat, Tensor? ee_table, "
          "Tensor? etype_csr, Tensor? ee_csc, int e_dim, float slope, Tensor? pre, Tensor? act_out, "
          "Tensor? prelu_w, Tensor(a!) d_out, Tensor(b!) d_pre, Tensor(c!) d_y, bool want_bias, bool want_dlogit, "
          "Tensor? dst_active) "
          "-> (Tensor, Tensor, Tensor, Tensor)", _gat_bwd_, _gat_bwd__fake)


def gat_fwd(struct, H, D, hs, eh, et, ee_table, etype_csc, ee_csc, slope, bias, res, prelu_w, want_pre,
            res_mask=None, res_scale=1.0, out=None, want_attn=True, want_stat=False, dst_active=None):
    """``torch.ops.magic.gat_fwd_`` with the calling convention of ``_lib.gat_fwd`` (graph structure
    object in, ``None`` for absent outputs): returns (out, pre, attn, stat)."""
    if out is None:
        out = torch.empty(struct.num_nodes, H * D, dtype=torch.float32, device=hs.device)
    pre, attn, stat = torch.ops.magic.gat_fwd_(
        *plan_tensors(struct.plan_in), struct.in_src, struct.in_dst, H, D, hs, eh, et, ee_table, etype_csc, ee_csc,
        float(slope), bias, res, prelu_w, bool(want_pre), res_mask, float(res_scale), bool(want_attn),
        bool(want_stat), dst_active, out)
    return out, _none(pre), _none(attn), _none(stat)


def gat_bwd(struct, H, D, hs, eh, stat, ee_table, etype_csr, ee_csc, e_dim, slope, pre, act_out, prelu_w, d_out,
            d_pre, d_y, want_bias, want_dlogit=False, dst_active=None):
    d_table, d_logit, d_bias, d_prelu = torch.ops.magic.gat_bwd_(
        *plan_tensors(struct.plan_in), *plan_tensors(struct.plan_out), struct.in_pos, struct.out_dst, struct.out_pos,
        H, D, hs, eh, stat, ee_table, etype_csr, ee_csc, int(e_dim), float(slope), pre, act_out, prelu_w, d_out,
        d_pre, d_y, bool(want_bias), bool(want_dlogit), dst_active)
    return _none(d_table), _none(d_logit), _none(d_bias), _none(d_prelu)


# ---- coded first-layer input (csrc/coded.hip): table of the values a one-hot / mask-token row can take, the row
# index per node, the sweeps that fetch rows by it and the grouped-sum weight gradient ------------------------
def _coded(code, table, token, scale):
    return _lib.CodedInput(code, table, token, token.numel(), scale)


def _coded_table(w_aug, token, scale: float):
    return _lib.coded_table(w_aug, token, scale)


def _coded_table_fake(w_aug, token, scale):
    n_dim = w_aug.shape[1]
    return w_aug.new_empty((n_dim + 1 + (1 << n_dim), w_aug.shape[0]))


_register("coded_table", "(Tensor w_aug, Tensor token, float scale) -> Tensor", _coded_table, _coded_table_fake)


def _node_code(node_type, n_dim: int, keep_mask, mask_rows):
    return _lib.node_code(node_type, n_dim, keep_mask, mask_rows)


def _node_code_fake(node_type, n_dim, keep_mask, mask_rows):
    return node_type.new_empty((node_type.numel(),), dtype=torch.int32)


_register("node_code", "(Tensor node_type, int n_dim, Tensor? keep_mask, Tensor? mask_rows) -> Tensor", _node_code,
          _node_code_fake)


def _gat_fwd_coded_(p0, p1, p2, p3, p4, p5, p6, p7, in_src, n_heads: int, head_dim: int, code, table, token,
                    scale: float, ee_table, etype_csc, slope: float, bias, prelu_w, want_pre: bool, out):
    s = _struct_in(out.shape[0], (p0, p1, p2, p3, p4, p5, p6, p7), in_src, None)
    _, pre, stat = _lib.gat_fwd_coded(s, n_heads, head_dim, _coded(code, table, token, scale), ee_table, etype_csc,
                                      slope, bias, prelu_w, want_pre, out=out)
    return pre if pre is not None else _empty(table), stat


def _gat_fwd_coded__fake(p0, p1, p2, p3, p4, p5, p6, p7, in_src, n_heads, head_dim, code, table, token, scale,
                         ee_table, etype_csc, slope, bias, prelu_w, want_pre, out):
    n, c = out.shape[0], n_heads * head_dim
    return table.new_empty((n, c) if want_pre else (0,)), table.new_empty((n, n_heads, 2))


_register("gat_fwd_coded_", "(" + _PLAN.format("in_") + ", Tensor in_src, int n_heads, int head_dim, Tensor code, "
          "Tensor table, Tensor token, float scale, Tensor ee_table, Tensor etype_csc, float slope, Tensor? bias, "
          "Tensor? prelu_w, bool want_pre, Tensor(a!) out) -> (Tensor, Tensor)", _gat_fwd_coded_, _gat_fwd_coded__fake)


def _gat_bwd_coded_(i0, i1, i2, i3, i4, i5, i6, i7, o0, o1, o2, o3, o4, o5, o6, o7, in_pos, out_dst, out_pos,
                    n_heads: int, head_dim: int, code, table, token, scale: float, stat, ee_table, etype_csr, e_dim: int,
                    slope: float, pre, act_out, prelu_w, d_out, d_pre, d_y, want_bias: bool):
    s = SimpleNamespace(num_nodes=code.shape[0], num_edges=int(out_dst.numel()),
                        plan_in=_plan(i0, i1, i2, i3, i4, i5, i6, i7), plan_out=_plan(o0, o1, o2, o3, o4, o5, o6, o7),
                        in_pos=in_pos, out_dst=out_dst, out_pos=out_pos)
    d_table, d_bias, d_prelu = _lib.gat_bwd_coded(s, n_heads, head_dim, _coded(code, table, token, scale), stat, ee_table,
                                                  etype_csr, e_dim, slope, pre, act_out, prelu_w, d_out, d_pre, d_y,
                                                  want_bias)
    e = _empty(table)
    return d_table, d_bias if d_bias is not None else e, d_prelu if d_prelu is not None else e


def _gat_bwd_coded__fake(i0, i1, i2, i3, i4, i5, i6, i7, o0, o1, o2, o3, o4, o5, o6, o7, in_pos, out_dst, out_pos,
                         n_heads, head_dim, code, table, token, scale, stat, ee_table, etype_csr, e_dim, slope, pre,
                         act_out, prelu_w, d_out, d_pre, d_y, want_bias):
    return (table.new_empty((e_dim, n_heads)), table.new_empty(n_heads * head_dim if want_bias else 0),
            table.new_empty(1 if prelu_w is not None else 0))


_register("gat_bwd_coded_", "(" + _PLAN.format("in_") + ", " + _PLAN.format("out_") + ", Tensor in_pos, Tensor out_dst, "
          "Tensor out_pos, int n_heads, int head_dim, Tensor code, Tensor table, Tensor token, float scale, Tensor stat, "
          "Tensor ee_table, Tensor etype_csr, int e_dim, float slope, Tensor? pre, Tensor? act_out, Tensor? prelu_w, "
          "Tensor(a!) d_out, Tensor(b!) d_pre, Tensor(c!) d_y, bool want_bias) -> (Tensor, Tensor, Tensor)",
          _gat_bwd_coded_, _gat_bwd_coded__fake)


def _coded_weight_grad(code, table, token, scale: float, d_y, w_aug):
    return _lib.coded_weight_grad(_coded(code, table, token, scale), d_y, w_aug)


def _coded_weight_grad_fake(code, table, token, scale, d_y, w_aug):
    return w_aug.new_empty(w_aug.shape), w_aug.new_empty((1, w_aug.shape[1]))


_register("coded_weight_grad", "(Tensor code, Tensor table, Tensor token, float scale, Tensor d_y, Tensor w_aug) "
          "-> (Tensor, Tensor)", _coded_weight_grad, _coded_weight_grad_fake)


class Coded:
    """Operands of a coded-input layer as the operators take them (code words, table, token, dropout scale)."""

    def __init__(self, code, table, token, scale):
        self.code, self.table, self.scale = code, table, float(scale)
        self.token = token.detach().reshape(-1).contiguous()


def gat_fwd_coded(struct, H, D, cd: Coded, ee_table, etype_csc, slope, bias, prelu_w, want_pre, out=None):
    """(out, pre | None, stat) of ``torch.ops.magic.gat_fwd_coded_``."""
    if out is None:
        out = torch.empty(struct.num_nodes, H * D, dtype=torch.float32, device=cd.code.device)
    pre, stat = torch.ops.magic.gat_fwd_coded_(*plan_tensors(struct.plan_in), struct.in_src, H, D, cd.code, cd.table,
                                               cd.token, cd.scale, ee_table, etype_csc, float(slope), bias, prelu_w,
                                               bool(want_pre), out)
    return out, _none(pre), stat


def gat_bwd_coded(struct, H, D, cd: Coded, stat, ee_table, etype_csr, e_dim, slope, pre, act_out, prelu_w, d_out, d_pre,
                  d_y, want_bias):
    d_table, d_bias, d_prelu = torch.ops.magic.gat_bwd_coded_(
        *plan_tensors(struct.plan_in), *plan_tensors(struct.plan_out), struct.in_pos, struct.out_dst, struct.out_pos,
        H, D, cd.code, cd.table, cd.token, cd.scale, stat, ee_table, etype_csr, int(e_dim), float(slope), pre, act_out,
        prelu_w, d_out, d_pre, d_y, bool(want_bias))
    return d_table, _none(d_bias), _none(d_prelu)


def coded_weight_grad(cd: Coded, d_y, w_aug):
    return torch.ops.magic.coded_weight_grad(cd.code, cd.table, cd.token, cd.scale, d_y, w_aug)


def proj_fwd(x, w, out_c, n_score, res_c, x_mask=None, x_scale=1.0, attn=None):
    """``attn``: (attn_h, attn_t) of the layer, any shape with out_c elements (see ``_lib.proj_fwd``)."""
    a_s, a_d = (None, None) if attn is None else (attn[0].detach().reshape(-1), attn[1].detach().reshape(-1))
    hs, score, res = torch.ops.magic.proj_fwd(x, w, out_c, n_score, res_c, x_mask, float(x_scale), a_s, a_d)
    return hs, _none(score), _none(res)


def proj_bwd_input(dy, w, add=None, out_mask=None, out_scale=1.0, add2=None, act_out=None, act_pre=None,
                   prelu_w=None, want_col_sum=False, out=None):
    if out is None:
        dx, cs, pg = torch.ops.magic.proj_bwd_input(dy, w, add, out_mask, float(out_scale), add2, act_out, act_pre,
                                                    prelu_w, bool(want_col_sum))
    else:
        dx = out
        cs, pg = torch.ops.magic.proj_bwd_input_(dy, w, add, out_mask, float(out_scale), add2, act_out, act_pre,
                                                 prelu_w, bool(want_col_sum), out)
    return dx, _none(cs), _none(pg)


def proj_bwd_weight(dy, x, x_mask=None, x_scale=1.0):
    return torch.ops.magic.proj_bwd_weight(dy, x, x_mask, float(x_scale))


# ------------------------------------------------------------------------------------- SCE
def _sce_fwd_bwd(x, n_dim: int, target, single_row: bool, mask_idx, alpha: float):
    loss, d_x, d_t = _lib.sce_fwd_bwd(x, n_dim, target, single_row, mask_idx, alpha)
    return loss, d_x, d_t if d_t is not None else _empty(x)


_register("sce_fwd_bwd", "(Tensor x, int n_dim, Tensor target, bool single_row, Tensor mask_idx, float alpha) "
          "-> (Tensor, Tensor, Tensor)", _sce_fwd_bwd,
          lambda x, n_dim, target, single_row, mask_idx, alpha:
          (x.new_empty(1), x.new_empty(x.shape), x.new_empty(n_dim if single_row else 0)))


def sce_fwd_bwd(x, n_dim, target, single_row, mask_idx, alpha):
    loss, d_x, d_t = torch.ops.magic.sce_fwd_bwd(x, int(n_dim), target, bool(single_row), mask_idx, float(alpha))
    return loss, d_x, _none(d_t)


# magic::sce_loss: the masked SCE as a differentiable operator of its own (gradient w.r.t. x)
def _sce_loss(x, target, mask_idx, alpha: float):
    single = target.shape[0] == 1 and x.shape[0] != 1
    return _lib.sce_fwd_bwd(x.contiguous(), x.shape[1], target.contiguous(), single, mask_idx, alpha)[0].view(())


_register("sce_loss", "(Tensor x, Tensor target, Tensor mask_idx, float alpha) -> Tensor", _sce_loss,
          lambda x, target, mask_idx, alpha: x.new_empty(()))


def _sce_loss_setup(ctx, inputs, output):
    x, target, mask_idx, alpha = inputs
    ctx.save_for_backward(x, target, mask_idx)
    ctx.alpha = alpha


def _sce_loss_backward(ctx, g):
    x, target, mask_idx = ctx.saved_tensors
    single = target.shape[0] == 1 and x.shape[0] != 1
    d_x = torch.ops.magic.sce_fwd_bwd(x.contiguous(), x.shape[1], target.contiguous(), single, mask_idx, ctx.alpha)[1]
    return d_x * g, None, None, None


torch.library.register_autograd("magic::sce_loss", _sce_loss_backward, setup_context=_sce_loss_setup)


# magic::linear_nobias: y = x W^T on the projection kernels, differentiable in x and W
_register("linear_nobias", "(Tensor x, Tensor w) -> Tensor",
          lambda x, w: _lib.proj_fwd(x, w, w.shape[0], 0, 0)[0], lambda x, w: x.new_empty(x.shape[0], w.shape[0]))


def _linear_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _linear_backward(ctx, dy):
    x, w = ctx.saved_tensors
    dy = dy.contiguous()
    dx = torch.ops.magic.proj_bwd_input(dy, w, None, None, 1.0, None, None, None, None, False)[0] \
        if ctx.needs_input_grad[0] else None
    dw = torch.ops.magic.proj_bwd_weight(dy, x, None, 1.0) if ctx.needs_input_grad[1] else None
    return dx, dw


torch.library.register_autograd("magic::linear_nobias", _linear_backward, setup_context=_linear_setup)


# ------------------------------------------------------------------------------------- edge-reconstruction loss tail
_register("edge_head_fwd", "(Tensor h_pre, Tensor? b1, Tensor w2, Tensor b2, Tensor? neg_valid, int n_pos, float slope) "
          "-> (Tensor, Tensor, Tensor)",
          lambda h_pre, b1, w2, b2, neg_valid, n_pos, slope:
          _lib.edge_head_fwd(h_pre, b1, w2, b2, neg_valid, n_pos, slope),
          lambda h_pre, b1, w2, b2, neg_valid, n_pos, slope:
          (h_pre.new_empty(1), h_pre.new_empty(h_pre.shape[0]), h_pre.new_empty(1)))


def _edge_head_bwd(h_pre, b1, w2, d_logit, weight_sum, grad_out, slope: float):
    d_h, d_b1, d_w2, d_b2 = _lib.edge_head_bwd(h_pre, b1, w2, d_logit, weight_sum, grad_out, slope)
    return d_h, d_b1 if d_b1 is not None else _empty(h_pre), d_w2, d_b2


_register("edge_head_bwd", "(Tensor h_pre, Tensor? b1, Tensor w2, Tensor d_logit, Tensor weight_sum, Tensor grad_out, "
          "float slope) -> (Tensor, Tensor, Tensor, Tensor)", _edge_head_bwd,
          lambda h_pre, b1, w2, d_logit, weight_sum, grad_out, slope:
          (h_pre.new_empty(h_pre.shape), h_pre.new_empty(h_pre.shape[1] if b1 is not None else 0),
           h_pre.new_empty(h_pre.shape[1]), h_pre.new_empty(1)))


# ------------------------------------------------------------------------------------- KNN, read-out
def _knn_fake(train, query, k, index_offset):
    q = query.shape[0]
    return (torch.empty(q, k, dtype=torch.float64, device=query.device),
            torch.empty(q, k, dtype=torch.int64, device=query.device))


_register("knn_topk", "(Tensor train, Tensor query, int k, int index_offset) -> (Tensor, Tensor)",
          lambda train, query, k, index_offset: _lib.knn_topk(train, query, k, index_offset), _knn_fake)
_register("knn_merge", "(Tensor part_dist, Tensor part_idx) -> (Tensor, Tensor)",
          lambda part_dist, part_idx: _lib.knn_merge(part_dist, part_idx),
          lambda part_dist, part_idx: (part_dist.new_empty(part_dist.shape[1:]), part_idx.new_empty(part_idx.shape[1:])))
_register("readout", "(Tensor feat, Tensor? node_type, int n_types, bool mean) -> Tensor",
          lambda feat, node_type, n_types, mean: _lib.readout(feat, node_type, n_types, mean),
          lambda feat, node_type, n_types, mean: feat.new_empty(1, max(n_types, 1) * feat.shape[1]))
