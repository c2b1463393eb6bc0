#!/usr/bin/env python3
"""What plain streaming passes reach on this GPU at the sizes of the THEIA-shaped step (tensor-library kernels:
copy = 1 read + 1 write, add = 2 reads + 1 write, sum = read only), as a yardstick for the projection kernels."""
import torch

dev = torch.device("cuda:0")
N = 700_000


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


for width in (64, 192, 1024):
    x = torch.randn(N, width, device=dev)
    y = torch.randn(N, width, device=dev)
    z = torch.empty_like(x)
    by = N * width * 4
    us = timeit(lambda: z.copy_(x))
    print(f"copy  [N,{width}]  {us:8.1f} us  {2 * by / us / 1e6:6.2f} TB/s")
    us = timeit(lambda: torch.add(x, y, out=z))
    print(f"add   [N,{width}]  {us:8.1f} us  {3 * by / us / 1e6:6.2f} TB/s")
    us = timeit(lambda: x.sum())
    print(f"sum   [N,{width}]  {us:8.1f} us  {by / us / 1e6:6.2f} TB/s")
    us = timeit(lambda: z.fill_(1.0))
    print(f"fill  [N,{width}]  {us:8.1f} us  {by / us / 1e6:6.2f} TB/s")
    if width == 192:
        v = x[:, 64:128]
        o = torch.empty(N, 64, device=dev)
        us = timeit(lambda: o.copy_(v))
        print(f"copy of a [N,64] column block of [N,192]  {us:8.1f} us  {2 * N * 256 / us / 1e6:6.2f} TB/s")
