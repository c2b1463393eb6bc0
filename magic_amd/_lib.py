"""ctypes binding of ``libmagic_hip.so`` (C ABI declared in ``include/magic_hip.h``).

PyTorch is used here only as the owner of device memory and of the HIP stream:
every call passes raw device pointers, sizes and the current stream handle.
There is no CPU fallback: if the shared library is missing, or a tensor is not on
the GPU, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MAGIC_HIP_LIBRARY: another build of the library (kernel tuning: two builds timed in turn on one machine)
_LIB_PATH = os.environ.get("MAGIC_HIP_LIBRARY") or os.path.join(_HERE, "libmagic_hip.so")
_lib = None

c_i32, c_i64, c_f32, c_vp, c_sz = C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_size_t


class MagicPlan(C.Structure):
    _fields_ = [("ptr", c_vp), ("light_max", c_i32), ("n_heavy", c_i32), ("heavy_node", c_vp),
                ("heavy_cptr", c_vp), ("n_chunks", c_i32), ("chunk_node", c_vp), ("chunk_beg", c_vp),
                ("chunk_end", c_vp), ("node_order", c_vp), ("xcd_start", c_vp)]


class MagicCodedInput(C.Structure):
    """``magic_coded_input`` of include/magic_hip.h."""
    _fields_ = [("code", c_vp), ("table", c_vp), ("ld_table", c_i64), ("token", c_vp), ("n_dim", c_i32),
                ("scale", c_f32)]


# name -> (restype, argtypes); must list every symbol include/magic_hip.h declares
SYMBOLS = {
    "magic_abi_version": (c_i32, []),
    "magic_status_string": (C.c_char_p, [c_i32]),
    "magic_prof_enable": (c_i32, [c_i32]),
    "magic_prof_reset": (c_i32, []),
    "magic_prof_read": (c_i32, [c_i32, C.c_char_p, C.POINTER(c_i64), C.POINTER(C.c_double)]),
    "magic_prof_sequence": (c_i32, [c_i32, C.c_char_p]),
    "magic_csr_build_workspace_bytes": (c_i32, [c_i64, c_i64, C.POINTER(c_sz)]),
    "magic_csr_build": (c_i32, [c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "magic_concat_offset": (c_i32, [c_i32, c_i32, c_vp, c_i64, c_vp]),
    "magic_walk_order_workspace_bytes": (c_sz, [c_i64]),
    "magic_walk_order": (c_i32, [c_i32, C.POINTER(c_vp), C.POINTER(c_i32), c_i32, C.POINTER(c_vp), C.POINTER(c_vp),
                                 c_vp, c_sz, c_vp]),
    "magic_negative_filter": (c_i32, [c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "magic_edge_head_workspace_bytes": (c_sz, [c_i64, c_i32]),
    "magic_edge_head_fwd": (c_i32, [c_i64, c_i64, c_i32, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp,
                                    c_vp, c_sz, c_vp]),
    "magic_edge_head_bwd": (c_i32, [c_i64, c_i32, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_i64, c_vp,
                                    c_vp, c_vp, c_vp, c_sz, c_vp]),
    "magic_adam_step": (c_i32, [c_i32, c_vp, c_vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, c_vp]),
    "magic_proj_fwd": (c_i32, [c_i64, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp, c_f32, c_vp, c_vp, c_i64,
                               c_vp, c_vp, c_i64, c_vp, c_sz, c_vp]),
    "magic_proj_fwd_attn": (c_i32, [c_i64, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp, c_f32, c_vp, c_vp, c_i64,
                                    c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "magic_proj_fwd_workspace_bytes": (c_sz, [c_i32, c_i32]),
    "magic_proj_bwd_input_workspace_bytes": (c_sz, [c_i64, c_i32, c_i32]),
    "magic_proj_bwd_input": (c_i32, [c_i64, c_i32, c_i32, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_sz, c_vp]),
    "magic_proj_bwd_weight_workspace_bytes": (c_sz, [c_i64, c_i32, c_i32]),
    "magic_proj_bwd_weight": (c_i32, [c_i64, c_i32, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp, c_f32, c_vp, c_vp, c_sz,
                                      c_vp]),
    "magic_rows_scatter_add_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "magic_rows_scatter_add": (c_i32, [c_i64, c_i64, c_i32, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_sz, c_vp]),
    "magic_dropout_mask": (c_i32, [c_i64, c_f32, C.c_uint64, c_vp, c_vp]),
    "magic_dropout_mask_dev": (c_i32, [c_i64, c_f32, c_vp, c_vp, c_vp]),
    "magic_fold_fwd": (c_i32, [c_i32] * 5 + [c_vp] * 10 + [c_vp]),
    "magic_fold_bwd": (c_i32, [c_i32] * 5 + [c_vp] * 15 + [c_vp]),
    "magic_fold_fwd_multi": (c_i32, [c_i32, c_vp, c_vp]),
    "magic_fold_bwd_multi": (c_i32, [c_i32, c_vp, c_vp]),
    "magic_gat_fwd_workspace_bytes": (c_sz, [c_i32, c_i32, c_i32, c_i32]),
    "magic_gat_fwd": (c_i32, [c_i32, c_i32, c_i32, c_i32, C.POINTER(MagicPlan), c_vp, c_vp,
                              c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_i64, c_vp, c_f32, c_vp,
                              c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "magic_gat_fwd_active": (c_i32, [c_i32, c_i32, c_i32, c_i32, C.POINTER(MagicPlan), c_vp, c_vp,
                                     c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_i64, c_vp, c_f32,
                                     c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "magic_gat_bwd_workspace_bytes": (c_sz, [c_i32] * 7),
    "magic_gat_bwd": (c_i32, [c_i32, c_i32, c_i32, c_i32, C.POINTER(MagicPlan), c_vp, C.POINTER(MagicPlan), c_vp,
                              c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_f32, c_vp, c_vp, c_i64,
                              c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "magic_gat_bwd_active": (c_i32, [c_i32, c_i32, c_i32, c_i32, C.POINTER(MagicPlan), c_vp, C.POINTER(MagicPlan), c_vp,
                                     c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_f32, c_vp, c_vp, c_i64,
                                     c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "magic_coded_table_rows": (c_i64, [c_i32]),
    "magic_coded_table": (c_i32, [c_i32, c_i32, c_vp, c_i64, c_vp, c_f32, c_vp, c_i64, c_vp]),
    "magic_node_code": (c_i32, [c_i64, c_i32, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "magic_gat_fwd_coded": (c_i32, [c_i32, c_i32, c_i32, c_i32, C.POINTER(MagicPlan), c_vp, C.POINTER(MagicCodedInput),
                                    c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "magic_gat_bwd_coded": (c_i32, [c_i32, c_i32, c_i32, c_i32, C.POINTER(MagicPlan), c_vp, C.POINTER(MagicPlan), c_vp,
                                    c_vp, C.POINTER(MagicCodedInput), c_vp, c_vp, c_vp, c_i32, c_f32, c_vp, c_vp, c_i64,
                                    c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "magic_coded_weight_grad_workspace_bytes": (c_sz, [c_i32, c_i32]),
    "magic_coded_weight_grad": (c_i32, [c_i64, c_i32, C.POINTER(MagicCodedInput), c_vp, c_i64, c_vp, c_i64, c_vp, c_i64,
                                        c_vp, c_vp, c_sz, c_vp]),
    "magic_sce_workspace_bytes": (c_sz, [c_i64, c_i32]),
    "magic_sce_fwd_bwd": (c_i32, [c_i64, c_i32, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_f32, c_f32,
                                  c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "magic_sce_fwd_bwd_static": (c_i32, [c_i64, c_i32, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_f32, c_f32,
                                         c_vp, c_vp, c_vp, c_vp]),
    "magic_readout_workspace_bytes": (c_sz, [c_i64, c_i32, c_i32]),
    "magic_readout": (c_i32, [c_i64, c_i32, c_vp, c_i64, c_vp, c_i32, c_i32, c_vp, c_vp, c_sz, c_vp]),
    "magic_knn_workspace_bytes": (c_sz, [c_i64, c_i64, c_i32, c_i32]),
    "magic_knn_topk": (c_i32, [c_i64, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp, c_i64, c_i64, c_vp, c_vp,
                               c_vp, c_sz, c_vp]),
    "magic_knn_merge": (c_i32, [c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
}


class MagicHipError(RuntimeError):
    pass


def lib_path() -> str:
    return _LIB_PATH


def load():
    """dlopen the library once and set every prototype; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise MagicHipError(
            f"{_LIB_PATH} not found: build it with `make -C magic_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback")
    lib = C.CDLL(_LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(status: int, what: str):
    if status != 0:
        msg = load().magic_status_string(status).decode()
        raise MagicHipError(f"{what} failed: {msg} (status {status})")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """Handle of the current HIP stream of the current device.  The raw accessors cost a fraction of a
    microsecond; ``torch.cuda.current_stream()`` builds a Python object per call (about 9 us, several
    hundred times per training step)."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _dev(t: Optional[torch.Tensor], dtype=None, name="tensor") -> Optional[int]:
    if t is None:
        return None
    if t.device.type != "cuda":
        raise MagicHipError(f"{name} must live on the GPU (got {t.device}); there is no CPU fallback")
    if dtype is not None and t.dtype != dtype:
        raise MagicHipError(f"{name} must be {dtype} (got {t.dtype})")
    return t.data_ptr()


_ws_cache: Dict[Tuple[int, int], torch.Tensor] = {}


def workspace(nbytes: int, device, slot: int = 0) -> torch.Tensor:
    """Grow-only scratch buffer per (device, slot); kernels on one stream run in order, so
    consecutive calls can share it."""
    device = torch.device(device)
    if device.type != "cuda":
        raise MagicHipError(f"operands must live on the GPU (got {device}); there is no CPU fallback")
    key = (device.index if device.index is not None else torch.cuda.current_device(), slot)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


# ------------------------------------------------------------------------------------- profiling
def prof_enable(on: bool):
    _check(load().magic_prof_enable(1 if on else 0), "magic_prof_enable")


def prof_reset():
    _check(load().magic_prof_reset(), "magic_prof_reset")


def prof_read(max_entries: int = 64):
    names = C.create_string_buffer(max_entries * 64)
    launches = (c_i64 * max_entries)()
    total = (C.c_double * max_entries)()
    n = load().magic_prof_read(max_entries, names, launches, total)
    if n < 0:
        _check(n, "magic_prof_read")
    out = {}
    for i in range(n):
        nm = names.raw[i * 64:(i + 1) * 64].split(b"\0", 1)[0].decode()
        out[nm] = (int(launches[i]), float(total[i]))
    return out


def prof_sequence(max_entries: int = 4096):
    """Names of the timed launches in launch order since the last reset."""
    names = C.create_string_buffer(max_entries * 64)
    n = load().magic_prof_sequence(max_entries, names)
    if n < 0:
        _check(n, "magic_prof_sequence")
    return [names.raw[i * 64:(i + 1) * 64].split(b"\0", 1)[0].decode() for i in range(min(n, max_entries))]


# ------------------------------------------------------------------------------------- CSR
def csr_build(src: torch.Tensor, dst: torch.Tensor, num_nodes: int):
    lib = load()
    E = int(src.numel())
    dev = src.device
    src = src.contiguous().to(torch.int64)
    dst = dst.contiguous().to(torch.int64)
    i32 = torch.int32
    in_ptr = torch.empty(num_nodes + 1, dtype=i32, device=dev)
    out_ptr = torch.empty(num_nodes + 1, dtype=i32, device=dev)
    in_src, in_eid, out_dst, out_eid = (torch.empty(E, dtype=i32, device=dev) for _ in range(4))
    nbytes = c_sz(0)
    _check(lib.magic_csr_build_workspace_bytes(num_nodes, E, C.byref(nbytes)), "magic_csr_build_workspace_bytes")
    ws = workspace(nbytes.value, dev)
    _check(lib.magic_csr_build(num_nodes, E, _dev(src, torch.int64, "src"), _dev(dst, torch.int64, "dst"),
                               _dev(in_ptr), _dev(in_src), _dev(in_eid), _dev(out_ptr), _dev(out_dst),
                               _dev(out_eid), ws.data_ptr(), ws.numel(), _stream()), "magic_csr_build")
    return in_ptr, in_src, in_eid, out_ptr, out_dst, out_eid


# ------------------------------------------------------------------------------------- union graphs
def concat_offset(groups, device) -> None:
    """``groups``: [(jobs, elem_bytes), ...] with ``jobs`` a list of (source address or 0, destination address,
    length, addend); runs destination[i] = source[i] + addend for every job, one launch per group.  The job
    tables of all groups travel to the device in one copy."""
    lib = load()
    groups = [(jobs, int(width)) for jobs, width in groups if len(jobs)]
    if not groups:
        return
    flat = [row for jobs, _ in groups for row in jobs]
    # pinned + asynchronous: a pageable copy would make the host wait for everything queued so far
    table = torch.tensor(flat, dtype=torch.int64).pin_memory().to(device, non_blocking=True)
    base, stream = table.data_ptr(), _stream()
    first = 0
    for jobs, width in groups:
        _check(lib.magic_concat_offset(width, len(jobs), base + 32 * first, max(row[2] for row in jobs), stream),
               "magic_concat_offset")
        first += len(jobs)


def walk_order(ptrs, light_max: int):
    """Walk order and XCD range starts of up to four device segment arrays (int32 ``ptr`` tensors with at
    least one node each); returns [(order, xcd_start), ...]."""
    lib = load()
    k = len(ptrs)
    dev = ptrs[0].device
    ns = [int(p.numel()) - 1 for p in ptrs]
    outs = [(torch.empty(n, dtype=torch.int32, device=dev), torch.empty(9, dtype=torch.int32, device=dev)) for n in ns]
    ws = workspace(lib.magic_walk_order_workspace_bytes(sum(ns)), dev)
    arr_p = (c_vp * k)(*[_dev(p, torch.int32, "ptr") for p in ptrs])
    arr_n = (c_i32 * k)(*ns)
    arr_o = (c_vp * k)(*[o.data_ptr() for o, _ in outs])
    arr_x = (c_vp * k)(*[x.data_ptr() for _, x in outs])
    _check(lib.magic_walk_order(k, arr_p, arr_n, int(light_max), arr_o, arr_x, ws.data_ptr(), ws.numel(), _stream()),
           "magic_walk_order")
    return outs


def edge_head_fwd(h_pre: torch.Tensor, b1: Optional[torch.Tensor], w2: torch.Tensor, b2: torch.Tensor,
                  neg_valid: Optional[torch.Tensor], n_pos: int, slope: float):
    """Returns (loss [1], d_logit [P] before the division by the weight sum, weight_sum [1]); ``b1`` is the
    first layer's bias when ``h_pre`` comes without it."""
    lib = load()
    P, W = h_pre.shape
    dev = h_pre.device
    h_pre, w2 = _rows(h_pre), w2.contiguous().view(-1)
    b1 = None if b1 is None else b1.contiguous().view(-1)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    d_logit = torch.empty(P, dtype=torch.float32, device=dev)
    wsum = torch.empty(1, dtype=torch.float32, device=dev)
    if neg_valid is not None:
        neg_valid = neg_valid.contiguous()
        if neg_valid.dtype not in (torch.bool, torch.uint8) or neg_valid.numel() != P - n_pos:
            raise MagicHipError("neg_valid must be a bool / uint8 tensor with one entry per negative pair")
    ws = workspace(lib.magic_edge_head_workspace_bytes(P, W), dev)
    _check(lib.magic_edge_head_fwd(P, int(n_pos), W, _dev(h_pre, torch.float32, "h_pre"), h_pre.stride(0),
                                   _dev(b1, torch.float32, "b1"), _dev(w2, torch.float32, "w2"),
                                   _dev(b2, torch.float32, "b2"), _dev(neg_valid), float(slope), _dev(loss),
                                   _dev(d_logit), _dev(wsum), ws.data_ptr(), ws.numel(), _stream()),
           "magic_edge_head_fwd")
    return loss, d_logit, wsum


def edge_head_bwd(h_pre: torch.Tensor, b1: Optional[torch.Tensor], w2: torch.Tensor, d_logit: torch.Tensor,
                  wsum: torch.Tensor, grad_out: torch.Tensor, slope: float):
    """Returns (d_h_pre [P, W], d_b1 [W] | None, d_w2 [W], d_b2 [1])."""
    lib = load()
    P, W = h_pre.shape
    dev = h_pre.device
    h_pre, w2 = _rows(h_pre), w2.contiguous().view(-1)
    b1 = None if b1 is None else b1.contiguous().view(-1)
    grad_out = grad_out.contiguous().view(-1)
    d_h = torch.empty(P, W, dtype=torch.float32, device=dev)
    d_w2 = torch.empty(W, dtype=torch.float32, device=dev)
    d_b2 = torch.empty(1, dtype=torch.float32, device=dev)
    d_b1 = torch.empty(W, dtype=torch.float32, device=dev) if b1 is not None else None
    ws = workspace(lib.magic_edge_head_workspace_bytes(P, W), dev)
    _check(lib.magic_edge_head_bwd(P, W, _dev(h_pre, torch.float32, "h_pre"), h_pre.stride(0),
                                   _dev(b1, torch.float32, "b1"), _dev(w2, torch.float32, "w2"),
                                   _dev(d_logit, torch.float32, "d_logit"), _dev(wsum, torch.float32, "weight_sum"),
                                   _dev(grad_out, torch.float32, "grad_out"), float(slope), _dev(d_h), d_h.stride(0),
                                   _dev(d_b1), _dev(d_w2), _dev(d_b2), ws.data_ptr(), ws.numel(), _stream()),
           "magic_edge_head_bwd")
    return d_h, d_b1, d_w2, d_b2


class MagicAdamSlot(C.Structure):
    """``magic_adam_slot`` of include/magic_hip.h."""
    _fields_ = [("param", c_vp), ("grad", c_vp), ("exp_avg", c_vp), ("exp_avg_sq", c_vp), ("n", c_i64)]


def adam_step(rows, step: torch.Tensor, lr: float, beta1: float, beta2: float, eps: float, weight_decay: float):
    """One Adam step over ``rows`` = [(param, grad, exp_avg, exp_avg_sq), ...] (contiguous float32 CUDA
    tensors); ``step`` is the int64 device counter the call advances."""
    n = len(rows)
    slots = (MagicAdamSlot * max(n, 1))()
    for i, (p, g, m, v) in enumerate(rows):
        sl = slots[i]
        sl.param, sl.grad, sl.exp_avg, sl.exp_avg_sq, sl.n = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
    _check(load().magic_adam_step(n, C.cast(slots, c_vp), _dev(step, torch.int64, "step"), lr, beta1, beta2, eps,
                                  weight_decay, _stream()), "magic_adam_step")


NEGATIVE_FILTER_MAX = 16384  # MAGIC_NEGATIVE_FILTER_MAX


def negative_filter(out_ptr: torch.Tensor, out_dst: torch.Tensor, in_ptr: torch.Tensor, in_src: torch.Tensor,
                    u: torch.Tensor, v: torch.Tensor, num_samples: int) -> torch.Tensor:
    """Bool mask over the candidate pairs (u[i], v[i]): in order, the first ``num_samples`` that are no edge
    of the graph (given as its int32 segment arrays in both directions), no self-loop and no repeat."""
    m = int(u.numel())
    u = u.contiguous()
    v = v.contiguous()
    ok = torch.empty(m, dtype=torch.bool, device=u.device)
    _check(load().magic_negative_filter(int(out_ptr.numel()) - 1, _dev(out_ptr, torch.int32, "out_ptr"),
                                        _dev(out_dst, torch.int32, "out_dst"), _dev(in_ptr, torch.int32, "in_ptr"),
                                        _dev(in_src, torch.int32, "in_src"), m, _dev(u, torch.int64, "u"),
                                        _dev(v, torch.int64, "v"), int(num_samples), _dev(ok), _stream()),
           "magic_negative_filter")
    return ok


def make_plan(plan, light_max: int) -> MagicPlan:
    """ctypes view of a graph.SegmentPlan (keeps no references: the caller holds the tensors)."""
    p = MagicPlan()
    p.ptr = plan.ptr.data_ptr()
    p.light_max = light_max
    p.n_heavy = plan.n_heavy
    p.heavy_node = plan.heavy_node.data_ptr() if plan.n_heavy else None
    p.heavy_cptr = plan.heavy_cptr.data_ptr() if plan.n_heavy else None
    p.n_chunks = plan.n_chunks
    p.chunk_node = plan.chunk_node.data_ptr() if plan.n_chunks else None
    p.chunk_beg = plan.chunk_beg.data_ptr() if plan.n_chunks else None
    p.chunk_end = plan.chunk_end.data_ptr() if plan.n_chunks else None
    p.node_order = plan.order.data_ptr() if plan.order is not None else None
    xs = getattr(plan, "xcd_start", None)
    p.xcd_start = xs.data_ptr() if xs is not None and p.node_order is not None else None
    return p


# ------------------------------------------------------------------------------------- projection
class ProjEpilogue(C.Structure):
    _fields_ = [("add", c_vp), ("ld_add", c_i64), ("out_mask", c_vp), ("out_scale", c_f32),
                ("add2", c_vp), ("ld_add2", c_i64), ("act_out", c_vp), ("ld_act", c_i64),
                ("act_pre", c_vp), ("prelu_w", c_vp), ("col_sum", c_vp), ("prelu_grad", c_vp)]


def _rows(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """2-D operand with unit inner stride (row stride free)."""
    if t is None:
        return None
    return t if t.stride(-1) == 1 else t.contiguous()


def proj_fwd(x: torch.Tensor, w: torch.Tensor, out_c: int, n_score: int, res_c: int,
             x_mask: Optional[torch.Tensor] = None, x_scale: float = 1.0,
             attn_src: Optional[torch.Tensor] = None, attn_dst: Optional[torch.Tensor] = None):
    """hs [N,out_c], score [N,n_score] | None, res [N,res_c] | None  =  split((x * mask * scale) @ w.T).
    ``attn_src`` / ``attn_dst`` ([out_c] each, the layer's ``attn_h`` / ``attn_t``): lets wide layers form the score
    columns from the output rows (``magic_proj_fwd_attn``); ``w`` stays the full augmented weight."""
    lib = load()
    assert x.dim() == 2 and w.dim() == 2 and w.shape[1] == x.shape[1] and w.shape[0] == out_c + n_score + res_c
    x = _rows(x)
    w = w.contiguous()
    N, F = x.shape
    dev = x.device
    if x_mask is not None:
        assert x_mask.dtype == torch.uint8 and x_mask.shape == x.shape and x_mask.is_contiguous()
    hs = torch.empty(N, out_c, dtype=torch.float32, device=dev)
    score = torch.empty(N, n_score, dtype=torch.float32, device=dev) if n_score else None
    res = torch.empty(N, res_c, dtype=torch.float32, device=dev) if res_c else None
    ws = workspace(lib.magic_proj_fwd_workspace_bytes(F, out_c + n_score + res_c), dev)
    head = (N, F, out_c, n_score, res_c, _dev(x, torch.float32, "x"), x.stride(0), _dev(x_mask), float(x_scale),
            _dev(w, torch.float32, "w"), _dev(hs), out_c, _dev(score), _dev(res), res_c)
    if attn_src is not None and attn_dst is not None:
        attn_src, attn_dst = attn_src.detach().reshape(-1).contiguous(), attn_dst.detach().reshape(-1).contiguous()
        if attn_src.numel() != out_c or attn_dst.numel() != out_c:
            raise MagicHipError("attn_src / attn_dst must hold one entry per output column")
        _check(lib.magic_proj_fwd_attn(*head, _dev(attn_src, torch.float32, "attn_src"),
                                       _dev(attn_dst, torch.float32, "attn_dst"), ws.data_ptr(), ws.numel(), _stream()),
               "magic_proj_fwd_attn")
    else:
        _check(lib.magic_proj_fwd(*head, ws.data_ptr(), ws.numel(), _stream()), "magic_proj_fwd")
    return hs, score, res


def proj_bwd_input(dy: torch.Tensor, w: torch.Tensor, add: Optional[torch.Tensor] = None,
                   out_mask: Optional[torch.Tensor] = None, out_scale: float = 1.0,
                   add2: Optional[torch.Tensor] = None, act_out: Optional[torch.Tensor] = None,
                   act_pre: Optional[torch.Tensor] = None, prelu_w: Optional[torch.Tensor] = None,
                   want_col_sum: bool = False, out: Optional[torch.Tensor] = None):
    """dX = epilogue(dy @ w); dy [N,K], w [K,F].  Returns (dx, col_sum | None, prelu_grad | None).
    See ``magic_proj_epilogue`` in include/magic_hip.h for the epilogue order."""
    lib = load()
    assert dy.dim() == 2 and w.dim() == 2 and dy.shape[1] == w.shape[0]
    dy = _rows(dy)
    w = w.contiguous()
    add, add2, act_out = _rows(add), _rows(add2), _rows(act_out)
    N, K = dy.shape
    F = w.shape[1]
    dev = dy.device
    dx = out if out is not None else torch.empty(N, F, dtype=torch.float32, device=dev)
    assert dx.stride(1) == 1 and dx.shape == (N, F)
    col_sum = torch.empty(F, dtype=torch.float32, device=dev) if want_col_sum else None
    prelu_grad = torch.empty(1, dtype=torch.float32, device=dev) if (act_out is not None and want_col_sum) else None
    ep = ProjEpilogue()
    ep.add, ep.ld_add = _dev(add), (add.stride(0) if add is not None else 0)
    if out_mask is not None:
        assert out_mask.dtype == torch.uint8 and out_mask.shape == (N, F) and out_mask.is_contiguous()
    ep.out_mask, ep.out_scale = _dev(out_mask), float(out_scale)
    ep.add2, ep.ld_add2 = _dev(add2), (add2.stride(0) if add2 is not None else 0)
    ep.act_out, ep.ld_act = _dev(act_out), (act_out.stride(0) if act_out is not None else 0)
    if act_pre is not None:
        assert act_pre.is_contiguous() and act_pre.shape == (N, F)
    ep.act_pre, ep.prelu_w = _dev(act_pre), _dev(prelu_w)
    ep.col_sum, ep.prelu_grad = _dev(col_sum), _dev(prelu_grad)
    nbytes = lib.magic_proj_bwd_input_workspace_bytes(N, K, F)
    ws = workspace(nbytes, dev)
    _check(lib.magic_proj_bwd_input(N, K, F, _dev(dy, torch.float32, "dy"), dy.stride(0),
                                    _dev(w, torch.float32, "w"), C.byref(ep), _dev(dx), dx.stride(0),
                                    ws.data_ptr(), ws.numel(), _stream()), "magic_proj_bwd_input")
    return dx, col_sum, prelu_grad


def proj_bwd_weight(dy: torch.Tensor, x: torch.Tensor, x_mask: Optional[torch.Tensor] = None,
                    x_scale: float = 1.0) -> torch.Tensor:
    """dW = dy.T @ (x * mask * scale); dy [N,M], x [N,F] (unit inner strides) -> [M,F]."""
    lib = load()
    assert dy.dim() == 2 and x.dim() == 2 and dy.shape[0] == x.shape[0]
    dy, x = _rows(dy), _rows(x)
    N, M = dy.shape
    F = x.shape[1]
    if x_mask is not None:
        assert x_mask.dtype == torch.uint8 and x_mask.shape == x.shape and x_mask.is_contiguous()
    dw = torch.empty(M, F, dtype=torch.float32, device=dy.device)
    nbytes = lib.magic_proj_bwd_weight_workspace_bytes(N, M, F)
    ws = workspace(nbytes, dy.device)
    _check(lib.magic_proj_bwd_weight(N, M, F, _dev(dy, torch.float32, "dy"), dy.stride(0),
                                     _dev(x, torch.float32, "x"), x.stride(0), _dev(x_mask), float(x_scale),
                                     _dev(dw), ws.data_ptr(), ws.numel(), _stream()), "magic_proj_bwd_weight")
    return dw


# ------------------------------------------------------------------------------------- scatter
def rows_scatter_add(dst: torch.Tensor, idx: torch.Tensor, src: torch.Tensor):
    """dst[idx[s]] += src[s] (repeated indices allowed, all in range), bitwise reproducible."""
    lib = load()
    assert dst.dim() == 2 and src.dim() == 2 and src.shape[1] == dst.shape[1] and idx.numel() == src.shape[0]
    assert dst.stride(1) == 1
    src = _rows(src)
    idx = idx.to(torch.int64).contiguous()
    ws = workspace(lib.magic_rows_scatter_add_workspace_bytes(src.shape[0], dst.shape[0]), dst.device)
    _check(lib.magic_rows_scatter_add(src.shape[0], dst.shape[0], src.shape[1], _dev(idx),
                                      _dev(src, torch.float32, "src"), src.stride(0),
                                      _dev(dst, torch.float32, "dst"), dst.stride(0), ws.data_ptr(), ws.numel(),
                                      _stream()), "magic_rows_scatter_add")
    return dst


# ------------------------------------------------------------------------------------- dropout
def dropout_mask(shape, keep_prob: float, seed: int, device) -> torch.Tensor:
    """uint8 keep mask of the given shape (1 = keep with probability ``keep_prob``)."""
    lib = load()
    m = torch.empty(shape, dtype=torch.uint8, device=device)
    _check(lib.magic_dropout_mask(m.numel(), float(keep_prob), int(seed) & 0xFFFFFFFFFFFFFFFF, _dev(m), _stream()),
           "magic_dropout_mask")
    return m


def dropout_mask_dev(shape, keep_prob: float, seed_dev: torch.Tensor, device) -> torch.Tensor:
    """Same as :func:`dropout_mask` with the seed taken from ``seed_dev[0]`` (int64, on the GPU) when the
    kernel runs, so the launch can be recorded in a HIP graph."""
    lib = load()
    assert seed_dev.dtype == torch.int64 and seed_dev.numel() >= 1
    m = torch.empty(shape, dtype=torch.uint8, device=device)
    _check(lib.magic_dropout_mask_dev(m.numel(), float(keep_prob), _dev(seed_dev), _dev(m), _stream()),
           "magic_dropout_mask_dev")
    return m


# ------------------------------------------------------------------------------------- folds
def fold_fwd(H: int, D: int, Dp: int, w_fc, attn_h, attn_t, w_res, w_efc, attn_e, bias):
    """(w_aug, table [e_dim,H], bias_p | None) from the layer's parameters; see include/magic_hip.h."""
    lib = load()
    F, e_dim = w_fc.shape[1], w_efc.shape[1]
    dev = w_fc.device
    Cp = H * Dp
    rows = Cp + 2 * H + (Cp if w_res is not None else 0)
    w_aug = torch.empty(rows, F, dtype=torch.float32, device=dev)
    table = torch.empty(e_dim, H, dtype=torch.float32, device=dev)
    bias_p = torch.empty(Cp, dtype=torch.float32, device=dev) if bias is not None else None
    _check(lib.magic_fold_fwd(H, D, Dp, F, e_dim, _dev(w_fc, torch.float32, "w_fc"), _dev(attn_h), _dev(attn_t),
                              _dev(w_res), _dev(w_efc), _dev(attn_e), _dev(bias), _dev(w_aug), _dev(table),
                              _dev(bias_p), _stream()), "magic_fold_fwd")
    return w_aug, table, bias_p


def fold_bwd(H: int, D: int, Dp: int, w_fc, attn_h, attn_t, w_efc, attn_e, d_w_aug, d_table, d_bias_p,
             has_res: bool, has_bias: bool):
    lib = load()
    F, e_dim = w_fc.shape[1], w_efc.shape[1]
    dev = w_fc.device
    C = H * D
    new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)  # noqa: E731
    d_w_fc, d_ah, d_at, d_w_efc, d_ae = new(C, F), new(1, H, D), new(1, H, D), new(C, e_dim), new(1, H, D)
    d_w_res = new(C, F) if has_res else None
    d_bias = new(1, H, D) if has_bias else None
    _check(lib.magic_fold_bwd(H, D, Dp, F, e_dim, _dev(w_fc), _dev(attn_h), _dev(attn_t), _dev(w_efc), _dev(attn_e),
                              _dev(d_w_aug), _dev(d_table), _dev(d_bias_p), _dev(d_w_fc), _dev(d_ah), _dev(d_at),
                              _dev(d_w_res), _dev(d_w_efc), _dev(d_ae), _dev(d_bias), _stream()), "magic_fold_bwd")
    return d_w_fc, d_ah, d_at, d_w_res, d_w_efc, d_ae, d_bias


class MagicFoldLayer(C.Structure):
    """``magic_fold_layer`` of include/magic_hip.h."""
    _fields_ = [(n, c_i32) for n in ("H", "D", "Dp", "F", "e_dim", "reserved")] + \
               [(n, c_vp) for n in ("w_fc", "attn_h", "attn_t", "w_res", "w_efc", "attn_e", "bias", "w_aug", "table",
                                    "bias_p", "d_w_aug", "d_table", "d_bias_p", "d_w_fc", "d_attn_h", "d_attn_t",
                                    "d_w_res", "d_w_efc", "d_attn_e", "d_bias")]


def _fold_slots(layers):
    """ctypes array for ``layers`` = [(H, D, Dp, w_fc, attn_h, attn_t, w_res, w_efc, attn_e, bias), ...]."""
    arr = (MagicFoldLayer * len(layers))()
    for sl, (H, D, Dp, w_fc, attn_h, attn_t, w_res, w_efc, attn_e, bias) in zip(arr, layers):
        sl.H, sl.D, sl.Dp, sl.F, sl.e_dim = H, D, Dp, w_fc.shape[1], w_efc.shape[1]
        sl.w_fc, sl.attn_h, sl.attn_t = _dev(w_fc, torch.float32, "w_fc"), _dev(attn_h), _dev(attn_t)
        sl.w_res, sl.w_efc, sl.attn_e, sl.bias = _dev(w_res), _dev(w_efc), _dev(attn_e), _dev(bias)
    return arr


def fold_fwd_multi(layers):
    """``fold_fwd`` for every layer of ``layers`` (see ``_fold_slots``) in one launch; returns
    [(w_aug, table, bias_p | None), ...]."""
    arr = _fold_slots(layers)
    outs = []
    for sl, (H, D, Dp, w_fc, attn_h, attn_t, w_res, w_efc, attn_e, bias) in zip(arr, layers):
        dev, Cp = w_fc.device, H * Dp
        rows = Cp + 2 * H + (Cp if w_res is not None else 0)
        w_aug = torch.empty(rows, sl.F, dtype=torch.float32, device=dev)
        table = torch.empty(sl.e_dim, H, dtype=torch.float32, device=dev)
        bias_p = torch.empty(Cp, dtype=torch.float32, device=dev) if bias is not None else None
        sl.w_aug, sl.table, sl.bias_p = _dev(w_aug), _dev(table), _dev(bias_p)
        outs.append((w_aug, table, bias_p))
    _check(load().magic_fold_fwd_multi(len(layers), C.cast(arr, c_vp), _stream()), "magic_fold_fwd_multi")
    return outs


def fold_bwd_multi(layers, grads):
    """``fold_bwd`` for every layer in one launch; ``grads`` = [(d_w_aug | None, d_table | None, d_bias_p | None),
    ...]; returns [(d_w_fc, d_attn_h, d_attn_t, d_w_res | None, d_w_efc, d_attn_e, d_bias | None), ...]."""
    arr = _fold_slots(layers)
    outs = []
    for sl, (H, D, Dp, w_fc, attn_h, attn_t, w_res, w_efc, attn_e, bias), (d_w_aug, d_table, d_bias_p) in \
            zip(arr, layers, grads):
        dev, Cn = w_fc.device, H * D
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)  # noqa: E731
        d_w_fc, d_ah, d_at, d_w_efc, d_ae = new(Cn, sl.F), new(1, H, D), new(1, H, D), new(Cn, sl.e_dim), new(1, H, D)
        d_w_res = new(Cn, sl.F) if w_res is not None else None
        d_bias = new(1, H, D) if bias is not None else None
        sl.d_w_aug, sl.d_table, sl.d_bias_p = _dev(d_w_aug), _dev(d_table), _dev(d_bias_p)
        sl.d_w_fc, sl.d_attn_h, sl.d_attn_t, sl.d_w_res = _dev(d_w_fc), _dev(d_ah), _dev(d_at), _dev(d_w_res)
        sl.d_w_efc, sl.d_attn_e, sl.d_bias = _dev(d_w_efc), _dev(d_ae), _dev(d_bias)
        outs.append((d_w_fc, d_ah, d_at, d_w_res, d_w_efc, d_ae, d_bias))
    _check(load().magic_fold_bwd_multi(len(layers), C.cast(arr, c_vp), _stream()), "magic_fold_bwd_multi")
    return outs


# ------------------------------------------------------------------------------------- GAT
def gat_fwd(struct, H: int, D: int, hs, eh, et, ee_table, etype_csc, ee_csc, slope: float, bias, res,
            prelu_w, want_pre: bool, res_mask: Optional[torch.Tensor] = None, res_scale: float = 1.0,
            out: Optional[torch.Tensor] = None, want_attn: bool = True, want_stat: bool = False,
            dst_active: Optional[torch.Tensor] = None):
    """``out`` may be a preallocated [N,C] view (unit inner stride, 16-byte aligned rows), e.g. a column
    block of the concatenated hidden-state buffer.  ``pre`` is only written by the kernel when the
    PReLU slope is <= 0 (otherwise backward recovers it from ``out``).  ``dst_active`` [N] uint8 marks the
    destinations whose output row is read (``magic_gat_fwd_active``; the others get bias + residual).  Returns
    (out, pre | None, attn [E,H] | None, stat [N,H,2] | None); ``stat`` is what ``gat_bwd`` needs."""
    lib = load()
    from .graph import LIGHT_MAX
    N, E = struct.num_nodes, struct.num_edges
    dev = hs.device
    Cc = H * D
    if out is None:
        out = torch.empty(N, Cc, dtype=torch.float32, device=dev)
    assert out.shape == (N, Cc) and out.stride(1) == 1
    pre = torch.empty(N, Cc, dtype=torch.float32, device=dev) if want_pre else None
    attn = torch.empty(E, H, dtype=torch.float32, device=dev) if want_attn else None
    stat = torch.empty(N, H, 2, dtype=torch.float32, device=dev) if want_stat else None
    plan = make_plan(struct.plan_in, LIGHT_MAX)
    nbytes = lib.magic_gat_fwd_workspace_bytes(E, struct.plan_in.n_chunks, H, D)
    ws = workspace(nbytes, dev)
    res = _rows(res)
    assert eh.stride(0) == et.stride(0)
    if res_mask is not None:
        assert res_mask.dtype == torch.uint8 and res_mask.shape == (N, Cc) and res_mask.is_contiguous()
    head = (N, E, H, D, C.byref(plan), _dev(struct.in_src), _dev(struct.in_dst),
            _dev(hs, torch.float32, "hs"), hs.stride(0), _dev(eh, torch.float32, "eh"),
            _dev(et, torch.float32, "et"), eh.stride(0), _dev(ee_table), _dev(etype_csc), _dev(ee_csc),
            float(slope), _dev(bias), _dev(res), res.stride(0) if res is not None else 0,
            _dev(res_mask), float(res_scale), _dev(prelu_w), _dev(out), out.stride(0), _dev(pre),
            _dev(attn), _dev(stat))
    if dst_active is not None:
        _check_active(dst_active, N)
        _check(lib.magic_gat_fwd_active(*head, _dev(dst_active), ws.data_ptr(), ws.numel(), _stream()),
               "magic_gat_fwd_active")
    else:
        _check(lib.magic_gat_fwd(*head, ws.data_ptr(), ws.numel(), _stream()), "magic_gat_fwd")
    return out, pre, attn, stat


def _check_active(dst_active, n):
    if dst_active.dtype != torch.uint8 or dst_active.shape != (n,) or not dst_active.is_contiguous():
        raise MagicHipError("dst_active must be a contiguous uint8 tensor with one entry per node")


def gat_bwd(struct, H: int, D: int, hs, eh, stat, ee_table, etype_csr, ee_csc, e_dim: int, slope: float, pre,
            act_out, prelu_w, d_out, d_pre, d_y, want_bias: bool, want_dlogit: bool = False,
            dst_active: Optional[torch.Tensor] = None):
    """Fills d_pre [N,>=C] and d_y [N,>=C+2H] (views allowed, unit inner stride); with
    ``prelu_w is None`` ``d_pre`` may be the very tensor passed as ``d_out`` (no copy is made).
    ``dst_active`` [N] uint8: the destinations at which ``d_out`` can be non-zero (``magic_gat_bwd_active``).
    ``eh`` [N,H] (row stride allowed), ``stat`` [N,H,2] from ``gat_fwd(want_stat=True)``; the edge term
    as in forward, with the edge types in CSR order.
    Returns (d_ee_table | None, d_logit [E,H] in CSC order | None, d_bias | None, d_prelu_w | None)."""
    lib = load()
    from .graph import LIGHT_MAX
    N, E = struct.num_nodes, struct.num_edges
    dev = hs.device
    Cc = H * D
    rec = torch.empty(E, H, dtype=torch.float32, device=dev) if want_dlogit else None
    d_table = torch.empty(e_dim, H, dtype=torch.float32, device=dev) if ee_table is not None else None
    d_bias = torch.empty(Cc, dtype=torch.float32, device=dev) if want_bias else None
    d_prelu = torch.empty(1, dtype=torch.float32, device=dev) if prelu_w is not None else None
    pin = make_plan(struct.plan_in, LIGHT_MAX)
    pout = make_plan(struct.plan_out, LIGHT_MAX)
    nbytes = lib.magic_gat_bwd_workspace_bytes(N, E, struct.plan_in.n_chunks, struct.plan_out.n_chunks, H, D,
                                               max(e_dim, 1))
    ws = workspace(nbytes, dev)
    d_out = _rows(d_out)
    act_out = _rows(act_out)
    assert stat.shape == (N, H, 2) and stat.is_contiguous() and eh.stride(-1) == 1
    head = (N, E, H, D, C.byref(pin), _dev(struct.in_pos), C.byref(pout), _dev(struct.out_dst),
            _dev(struct.out_pos), _dev(hs, torch.float32, "hs"), hs.stride(0),
            _dev(eh, torch.float32, "eh"), eh.stride(0), _dev(stat, torch.float32, "stat"),
            _dev(ee_table), _dev(etype_csr), _dev(ee_csc), int(e_dim), float(slope),
            _dev(pre), _dev(act_out), act_out.stride(0) if act_out is not None else 0,
            _dev(prelu_w), _dev(d_out, torch.float32, "d_out"), d_out.stride(0),
            _dev(d_pre), d_pre.stride(0), _dev(d_y), d_y.stride(0), _dev(d_table))
    if dst_active is not None:
        _check_active(dst_active, N)
        if want_dlogit:
            raise MagicHipError("d_logit is not produced together with dst_active")
        _check(lib.magic_gat_bwd_active(*head, _dev(d_bias), _dev(d_prelu), _dev(dst_active), ws.data_ptr(),
                                        ws.numel(), _stream()), "magic_gat_bwd_active")
    else:
        _check(lib.magic_gat_bwd(*head, _dev(rec), _dev(d_bias), _dev(d_prelu), ws.data_ptr(), ws.numel(),
                                 _stream()), "magic_gat_bwd")
    return d_table, rec, d_bias, d_prelu


# ------------------------------------------------------------------------------------- coded layer input
CODED_MAX_DIM = 13  # MAGIC_CODED_MAX_DIM


class CodedInput:
    """The first encoder layer's input as code words (``magic_coded_input``): ``code`` [N] (int32 holding the table
    row of every node), ``table`` [n_dim + 1 + 2^n_dim, M] (``coded_table``), ``token`` [n_dim], the dropout scale.
    Holds the tensors alive; ``struct()`` is the ctypes view."""

    def __init__(self, code: torch.Tensor, table: torch.Tensor, token: torch.Tensor, n_dim: int, scale: float):
        assert code.dtype == torch.int32 and code.is_contiguous()
        assert table.dtype == torch.float32 and table.stride(1) == 1 and table.stride(0) % 4 == 0
        assert table.data_ptr() % 16 == 0 and table.shape[0] == n_dim + 1 + (1 << n_dim)
        token = token.detach().reshape(-1).contiguous()
        assert token.dtype == torch.float32 and token.numel() == n_dim
        self.code, self.table, self.token, self.n_dim, self.scale = code, table, token, int(n_dim), float(scale)

    def struct(self) -> MagicCodedInput:
        ci = MagicCodedInput()
        ci.code, ci.table, ci.ld_table = self.code.data_ptr(), self.table.data_ptr(), self.table.stride(0)
        ci.token, ci.n_dim, ci.scale = self.token.data_ptr(), self.n_dim, self.scale
        return ci


def coded_table(w_aug: torch.Tensor, token: torch.Tensor, scale: float) -> torch.Tensor:
    """[n_dim + 1 + 2^n_dim, M] table of the rows ``x . w_aug^T`` a coded input row can produce (w_aug [M, n_dim])."""
    lib = load()
    M, n_dim = w_aug.shape
    assert w_aug.stride(1) == 1
    rows = int(lib.magic_coded_table_rows(n_dim))
    if rows == 0:
        raise MagicHipError(f"coded input supports at most {CODED_MAX_DIM} input columns (got {n_dim})")
    ld = (M + 3) // 4 * 4
    tab = torch.empty(rows, ld, dtype=torch.float32, device=w_aug.device)[:, :M]
    token = token.detach().reshape(-1).contiguous()
    _check(lib.magic_coded_table(n_dim, M, _dev(w_aug, torch.float32, "w_aug"), w_aug.stride(0),
                                 _dev(token, torch.float32, "token"), float(scale), _dev(tab), tab.stride(0), _stream()),
           "magic_coded_table")
    return tab


def node_code(node_type: torch.Tensor, n_dim: int, keep_mask: Optional[torch.Tensor],
              mask_rows: Optional[torch.Tensor]) -> torch.Tensor:
    """Code words [N] (int32 holding the uint32 bit patterns) from node types, the uint8 keep mask [N, n_dim] of the
    layer's dropout (None: keep all) and the ids of the rows that hold the mask token (None: none)."""
    node_type = node_type.view(-1).to(torch.int64).contiguous()
    n = node_type.numel()
    code = torch.empty(n, dtype=torch.int32, device=node_type.device)
    if keep_mask is not None:
        assert keep_mask.dtype == torch.uint8 and keep_mask.shape == (n, n_dim) and keep_mask.is_contiguous()
    n_mask = 0
    if mask_rows is not None:
        mask_rows = mask_rows.view(-1).to(torch.int64).contiguous()
        n_mask = mask_rows.numel()
    _check(load().magic_node_code(n, int(n_dim), _dev(node_type), _dev(keep_mask), n_mask, _dev(mask_rows), _dev(code),
                                  _stream()), "magic_node_code")
    return code


def gat_fwd_coded(struct, H: int, D: int, ci: CodedInput, ee_table, etype_csc, slope: float, bias, prelu_w,
                  want_pre: bool, out: Optional[torch.Tensor] = None):
    """``gat_fwd`` for a layer whose input is given as code words: returns (out, pre | None, stat [N,H,2])."""
    lib = load()
    from .graph import LIGHT_MAX
    N, E = struct.num_nodes, struct.num_edges
    dev = ci.code.device
    Cc = H * D
    assert ci.table.shape[1] >= 2 * Cc + 2 * H
    if out is None:
        out = torch.empty(N, Cc, dtype=torch.float32, device=dev)
    assert out.shape == (N, Cc) and out.stride(1) == 1
    pre = torch.empty(N, Cc, dtype=torch.float32, device=dev) if want_pre else None
    stat = torch.empty(N, H, 2, dtype=torch.float32, device=dev)
    plan = make_plan(struct.plan_in, LIGHT_MAX)
    ws = workspace(lib.magic_gat_fwd_workspace_bytes(E, struct.plan_in.n_chunks, H, D), dev)
    cs = ci.struct()
    _check(lib.magic_gat_fwd_coded(N, E, H, D, C.byref(plan), _dev(struct.in_src), C.byref(cs), _dev(ee_table),
                                   _dev(etype_csc), float(slope), _dev(bias), _dev(prelu_w), _dev(out), out.stride(0),
                                   _dev(pre), _dev(stat), ws.data_ptr(), ws.numel(), _stream()), "magic_gat_fwd_coded")
    return out, pre, stat


def gat_bwd_coded(struct, H: int, D: int, ci: CodedInput, stat, ee_table, etype_csr, e_dim: int, slope: float, pre,
                  act_out, prelu_w, d_out, d_pre, d_y, want_bias: bool):
    """``gat_bwd`` for a coded-input layer.  Returns (d_ee_table, d_bias | None, d_prelu_w | None)."""
    lib = load()
    from .graph import LIGHT_MAX
    N, E = struct.num_nodes, struct.num_edges
    dev = ci.code.device
    Cc = H * D
    d_table = torch.empty(e_dim, H, dtype=torch.float32, device=dev)
    d_bias = torch.empty(Cc, dtype=torch.float32, device=dev) if want_bias else None
    d_prelu = torch.empty(1, dtype=torch.float32, device=dev) if prelu_w is not None else None
    pin = make_plan(struct.plan_in, LIGHT_MAX)
    pout = make_plan(struct.plan_out, LIGHT_MAX)
    ws = workspace(lib.magic_gat_bwd_workspace_bytes(N, E, struct.plan_in.n_chunks, struct.plan_out.n_chunks, H, D,
                                                     max(e_dim, 1)), dev)
    d_out = _rows(d_out)
    act_out = _rows(act_out)
    assert stat.shape == (N, H, 2) and stat.is_contiguous()
    cs = ci.struct()
    _check(lib.magic_gat_bwd_coded(N, E, H, D, C.byref(pin), _dev(struct.in_pos), C.byref(pout), _dev(struct.out_dst),
                                   _dev(struct.out_pos), C.byref(cs), _dev(stat, torch.float32, "stat"), _dev(ee_table),
                                   _dev(etype_csr), int(e_dim), float(slope), _dev(pre), _dev(act_out),
                                   act_out.stride(0) if act_out is not None else 0, _dev(prelu_w),
                                   _dev(d_out, torch.float32, "d_out"), d_out.stride(0), _dev(d_pre), d_pre.stride(0),
                                   _dev(d_y), d_y.stride(0), _dev(d_table), _dev(d_bias), _dev(d_prelu), ws.data_ptr(),
                                   ws.numel(), _stream()), "magic_gat_bwd_coded")
    return d_table, d_bias, d_prelu


def coded_weight_grad(ci: CodedInput, d_y: torch.Tensor, w_aug: torch.Tensor):
    """(dW [M, n_dim], d_token [1, n_dim]) of a coded-input layer from d_y [N, M] and its augmented weight [M, n_dim]."""
    lib = load()
    d_y = _rows(d_y)
    N, M = d_y.shape
    assert w_aug.shape == (M, ci.n_dim) and w_aug.stride(1) == 1
    dev = d_y.device
    dw = torch.empty(M, ci.n_dim, dtype=torch.float32, device=dev)
    d_token = torch.empty(1, ci.n_dim, dtype=torch.float32, device=dev)
    ws = workspace(lib.magic_coded_weight_grad_workspace_bytes(M, ci.n_dim), dev)
    cs = ci.struct()
    _check(lib.magic_coded_weight_grad(N, M, C.byref(cs), _dev(d_y, torch.float32, "d_y"), d_y.stride(0),
                                       _dev(w_aug, torch.float32, "w_aug"), w_aug.stride(0), _dev(dw), dw.stride(0),
                                       _dev(d_token), ws.data_ptr(), ws.numel(), _stream()), "magic_coded_weight_grad")
    return dw, d_token


# ------------------------------------------------------------------------------------- SCE
def sce_fwd_bwd(x: torch.Tensor, n_dim: int, target: torch.Tensor, single_row: bool, mask_idx: torch.Tensor,
                alpha: float):
    """Returns (loss[1], d_x like x, d_target_row[n_dim] | None)."""
    lib = load()
    dev = x.device
    assert x.stride(1) == 1
    N = x.shape[0]
    mask_idx = mask_idx.contiguous().to(torch.int64)
    M = int(mask_idx.numel())
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    d_x = torch.empty_strided((N, x.stride(0)), (x.stride(0), 1), dtype=torch.float32, device=dev)
    d_t = torch.empty(n_dim, dtype=torch.float32, device=dev) if single_row else None
    if single_row:
        target = target.contiguous().view(-1)
        ld_t = 0
    else:
        assert target.stride(1) == 1
        ld_t = target.stride(0)
    nbytes = lib.magic_sce_workspace_bytes(M, n_dim)
    ws = workspace(nbytes, dev)
    _check(lib.magic_sce_fwd_bwd(N, n_dim, M, _dev(mask_idx), _dev(x, torch.float32, "x"), x.stride(0),
                                 _dev(target, torch.float32, "target"), ld_t, float(alpha), 1.0, _dev(loss),
                                 _dev(d_x), _dev(d_t), ws.data_ptr(), ws.numel(), _stream()), "magic_sce_fwd_bwd")
    return loss, d_x[:, :x.shape[1]], d_t


# ------------------------------------------------------------------------------------- read-out
def readout(feat: torch.Tensor, node_type: Optional[torch.Tensor], n_types: int, mean: bool) -> torch.Tensor:
    """[1, max(n_types, 1) * width] mean / sum read-out of feat [N, width] (per node type when n_types > 0)."""
    lib = load()
    assert feat.dim() == 2
    feat = _rows(feat)
    N, W = feat.shape
    dev = feat.device
    if n_types:
        node_type = node_type.view(-1).to(device=dev, dtype=torch.int64).contiguous()
        assert node_type.numel() == N
    out = torch.empty(1, max(n_types, 1) * W, dtype=torch.float32, device=dev)
    ws = workspace(lib.magic_readout_workspace_bytes(N, W, n_types), dev)
    _check(lib.magic_readout(N, W, _dev(feat, torch.float32, "feat"), feat.stride(0),
                             _dev(node_type) if n_types else None, int(n_types), 1 if mean else 0, _dev(out),
                             ws.data_ptr(), ws.numel(), _stream()), "magic_readout")
    return out


# ------------------------------------------------------------------------------------- KNN
def knn_topk(train: torch.Tensor, query: torch.Tensor, k: int, index_offset: int = 0):
    lib = load()
    assert train.dim() == 2 and query.dim() == 2 and train.shape[1] == query.shape[1]
    train = train.contiguous()
    query = query.contiguous()
    dev = train.device
    Nt, d = train.shape
    Q = query.shape[0]
    dist = torch.empty(Q, k, dtype=torch.float64, device=dev)
    idx = torch.empty(Q, k, dtype=torch.int64, device=dev)
    nbytes = lib.magic_knn_workspace_bytes(Nt, Q, d, k)
    ws = workspace(nbytes, dev, slot=1)
    _check(lib.magic_knn_topk(Nt, Q, d, k, _dev(train, torch.float32, "train"), train.stride(0),
                              _dev(query, torch.float32, "query"), query.stride(0), int(index_offset),
                              _dev(dist), _dev(idx), ws.data_ptr(), ws.numel(), _stream()), "magic_knn_topk")
    return dist, idx


def knn_merge(part_dist: torch.Tensor, part_idx: torch.Tensor):
    """[P,Q,k] partial results (each ascending) -> global [Q,k]."""
    lib = load()
    P, Q, k = part_dist.shape
    part_dist = part_dist.contiguous()
    part_idx = part_idx.contiguous()
    dist = torch.empty(Q, k, dtype=torch.float64, device=part_dist.device)
    idx = torch.empty(Q, k, dtype=torch.int64, device=part_dist.device)
    _check(lib.magic_knn_merge(Q, k, P, _dev(part_dist, torch.float64, "part_dist"),
                               _dev(part_idx, torch.int64, "part_idx"), _dev(dist), _dev(idx), _stream()),
           "magic_knn_merge")
    return dist, idx
