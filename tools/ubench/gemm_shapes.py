#!/usr/bin/env python3
"""fp32 GEMM rates of the library BLAS (torch.mm) and of the projection kernels of this repo on
the shapes of the hidden=256 configurations (StreamSpot batch of 12 graphs: N = 96,000)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from magic_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 96_000
dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_tf32 = False


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for (k, m) in [(8, 520), (256, 264), (256, 256), (264, 256), (1024, 256), (256, 1024), (256, 10), (64, 72), (72, 64)]:
    x = torch.randn(N, k, device=dev)
    w = torch.randn(m, k, device=dev)
    fl = 2.0 * N * k * m
    t_blas = timeit(lambda: x @ w.t())
    try:
        t_mine = timeit(lambda: _lib.proj_fwd(x, w, m, 0, 0))
    except Exception as ex:  # noqa: BLE001
        t_mine = float("nan")
    wt = w.t().contiguous()  # [k, m]: dX = dY @ W form
    t_blas_b = timeit(lambda: x @ wt)
    try:
        t_mine_b = timeit(lambda: _lib.proj_bwd_input(x, wt))
    except Exception as ex:  # noqa: BLE001
        t_mine_b = float("nan")
    print(f"[{N},{k}]x[{k},{m}]  blas {t_blas:7.1f} us {fl / t_blas / 1e6:6.1f} TF | proj_fwd {t_mine:7.1f} us "
          f"{fl / t_mine / 1e6:6.1f} TF | blas(nn) {t_blas_b:7.1f} us | proj_bwd_input {t_mine_b:7.1f} us", flush=True)

print("weight gradients dW[M,F] = dY[N,M]^T X[N,F]")
for (m, f) in [(264, 256), (256, 1024), (520, 8), (10, 256), (72, 64)]:
    dy = torch.randn(N, m, device=dev)
    x = torch.randn(N, f, device=dev)
    fl = 2.0 * N * m * f
    t_blas = timeit(lambda: dy.t() @ x)
    t_mine = timeit(lambda: _lib.proj_bwd_weight(dy, x))
    print(f"[{m},{N}]x[{N},{f}]  blas {t_blas:7.1f} us {fl / t_blas / 1e6:6.1f} TF | proj_bwd_weight {t_mine:7.1f} us "
          f"{fl / t_mine / 1e6:6.1f} TF", flush=True)
