#!/usr/bin/env python3
"""Time one k-nearest-neighbour search (train rows x queries, d = 64, k = 10) and print per-kernel times.
usage: knn_time.py [n_train] [n_query] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from magic_amd import _lib  # noqa: E402

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 2604530
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1005)
centers = torch.randn(200, 64, device=dev, generator=gen)
train = centers[torch.randint(0, 200, (nt,), device=dev, generator=gen)] + 0.05 * torch.randn(nt, 64, device=dev, generator=gen)
query = centers[torch.randint(0, 200, (nq,), device=dev, generator=gen)] + 0.05 * torch.randn(nq, 64, device=dev, generator=gen)
_lib.knn_topk(train, query, 10)
torch.cuda.synchronize()
_lib.prof_reset()
_lib.prof_enable(True)
for _ in range(reps):
    _lib.knn_topk(train, query, 10)
torch.cuda.synchronize()
_lib.prof_enable(False)
for k, (cnt, ms) in sorted(_lib.prof_read(64).items(), key=lambda kv: -kv[1][1]):
    extra = ""
    if "filter" in k:
        extra = f"  {2.0 * nt * nq * 64 / (ms / cnt * 1e-3) / 1e12:.1f} TFLOP/s (algorithmic)"
    print(f"  {k:28s} {ms / cnt:10.3f} ms x{cnt // reps}{extra}", flush=True)
