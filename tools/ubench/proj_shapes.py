#!/usr/bin/env python3
"""Time the dense projection kernels on the shapes of one THEIA-shaped training step.

Prints microseconds per launch and the algorithmic HBM rate (bytes every operand is read or
written exactly once / time).  Run on the GPU box:  python tools/ubench/proj_shapes.py [N]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from magic_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 700_000
REPS = 20
dev = torch.device("cuda:0")


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(REPS):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / REPS * 1e3


def report(name, us, nbytes):
    print(f"{name:44s} {us:8.1f} us  {nbytes / 1e6:8.1f} MB  {nbytes / us / 1e6:6.2f} TB/s", flush=True)


def rnd(*shape):
    return torch.randn(*shape, device=dev)


def mask(*shape):
    return (torch.rand(*shape, device=dev) < 0.9).to(torch.uint8)


C, H = 64, 4
x5 = rnd(N, 5)
w0 = rnd(C + 2 * H + C, 5)
report("fwd L0 5->136", timeit(lambda: _lib.proj_fwd(x5, w0, C, 2 * H, C)), N * 4 * (5 + 136))
enc = rnd(N, 3 * C)
x64 = enc[:, C:2 * C]
m64 = mask(N, C)
w1 = rnd(C + 2 * H, C)
report("fwd L1 64->72 (strided in, mask)", timeit(lambda: _lib.proj_fwd(x64, w1, C, 2 * H, 0, x_mask=m64, x_scale=1.1)),
       N * (4 * (64 + 72) + 64))
report("fwd L1 64->72 (strided in)", timeit(lambda: _lib.proj_fwd(x64, w1, C, 2 * H, 0)), N * 4 * (64 + 72))
we = rnd(C, 3 * C)
report("fwd e2d 192->64", timeit(lambda: _lib.proj_fwd(enc, we, C, 0, 0)), N * 4 * (192 + 64))
rep = rnd(N, C)
wd = rnd(16 + 2, C)
report("fwd dec 64->18 (mask)", timeit(lambda: _lib.proj_fwd(rep, wd, 16, 2, 0, x_mask=m64, x_scale=1.1)),
       N * (4 * (64 + 18) + 64))

dy20 = rnd(N, 20)[:, :18]
report("bwd_in dec 18->64 (mask)", timeit(lambda: _lib.proj_bwd_input(dy20, wd, out_mask=m64, out_scale=1.1)),
       N * (4 * (18 + 64) + 64))
drep = rnd(N, C)
report("bwd_in e2d 64->192", timeit(lambda: _lib.proj_bwd_input(drep, we)), N * 4 * (64 + 192))
dy72 = rnd(N, 72)
dpre = rnd(N, C)
denc = rnd(N, 3 * C)
pre = rnd(N, C)
pw = torch.full((1,), 0.25, device=dev)
report("bwd_in L 72->64 plain", timeit(lambda: _lib.proj_bwd_input(dy72, w1)), N * 4 * (72 + 64))
report("bwd_in L 72->64 fused (add,mask,add2,prelu,sums)",
       timeit(lambda: _lib.proj_bwd_input(dy72, w1, add=dpre, out_mask=m64, out_scale=1.1, add2=denc[:, :C],
                                          act_out=enc[:, :C], act_pre=pre, prelu_w=pw, want_col_sum=True)),
       N * (4 * (72 + 64 * 4) + 64))
dy136 = rnd(N, 136)
report("bwd_in L0 136->5", timeit(lambda: _lib.proj_bwd_input(dy136, w0)), N * 4 * (136 + 5))

report("dW dec 18x64 (mask)", timeit(lambda: _lib.proj_bwd_weight(dy20, rep, x_mask=m64, x_scale=1.1)),
       N * (4 * (18 + 64) + 64))
report("dW e2d 64x192", timeit(lambda: _lib.proj_bwd_weight(drep, enc)), N * 4 * (64 + 192))
report("dW L 72x64 (mask, strided x)", timeit(lambda: _lib.proj_bwd_weight(dy72, x64, x_mask=m64, x_scale=1.1)),
       N * (4 * (72 + 64) + 64))
report("dW L 72x64", timeit(lambda: _lib.proj_bwd_weight(dy72, rep)), N * 4 * (72 + 64))
m5 = mask(N, 5)
report("dW L0 136x5 (mask)", timeit(lambda: _lib.proj_bwd_weight(dy136, x5, x_mask=m5, x_scale=1.1)),
       N * (4 * (136 + 5) + 5))
