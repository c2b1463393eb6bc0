#!/usr/bin/env python3
"""A few launches of one projection shape (for counter passes): gemm_one.py N K M [fwd|bwd]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from magic_amd import _lib  # noqa: E402

n, k, m = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode = sys.argv[4] if len(sys.argv) > 4 else "fwd"
dev = torch.device("cuda:0")
x = torch.randn(n, k, device=dev)
w = torch.randn(m, k, device=dev)
wt = w.t().contiguous()
for _ in range(5):
    if mode == "fwd":
        _lib.proj_fwd(x, w, m, 0, 0)
    else:
        _lib.proj_bwd_input(x, wt)
torch.cuda.synchronize()
