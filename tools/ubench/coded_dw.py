"""Times the weight gradient of the coded first layer (grouped sum of d_y rows by code word) at the THEIA and
StreamSpot-batch shapes and checks it against a float64 sum; MAGIC_CODED_DW_COLS=1 in the environment selects
the column-per-thread kernel (the fallback for unaligned rows) for comparison.
usage: python tools/ubench/coded_dw.py ; MAGIC_CODED_DW_COLS=1 python tools/ubench/coded_dw.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from magic_amd import _lib  # noqa: E402


def run(n, n_dim, hidden, iters=30):
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    M = 2 * hidden + 8
    w = torch.randn(M, n_dim, generator=g).to(dev)
    tok = torch.randn(n_dim, generator=g).to(dev)
    types = torch.randint(0, n_dim, (n,), generator=g).to(dev)
    keep = (torch.rand(n, n_dim, generator=g) < 0.9).to(torch.uint8).to(dev)
    rows = torch.randperm(n, generator=g)[: n // 2].to(dev)
    code = _lib.node_code(types, n_dim, keep, rows)
    tab = _lib.coded_table(w, tok, 1.0 / 0.9)
    ci = _lib.CodedInput(code, tab, tok, n_dim, 1.0 / 0.9)
    dy = torch.randn(n, M, generator=g).to(dev)
    name = "cols" if os.environ.get("MAGIC_CODED_DW_COLS") else "quads"
    for _ in range(3):
        dw, dt = _lib.coded_weight_grad(ci, dy, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        dw, dt = _lib.coded_weight_grad(ci, dy, w)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"N={n} n_dim={n_dim} M={M} {name}: {us:.1f} us  {(n * M * 4 + n * 4) / us / 1e6:.2f} TB/s", flush=True)
    # float64 reference: x[u, j] as the first layer sees it, dW = d_y^T x
    x = torch.nn.functional.one_hot(types, n_dim).double()
    x[rows] = tok.double()
    x = x * keep.double() / 0.9
    ref = dy.double().t() @ x
    print("  max |dW - float64| / max|dW| =", float((dw.double() - ref).abs().max() / ref.abs().max()))


if __name__ == "__main__":
    run(700_000, 5, 64)
    run(96_000, 8, 256)
    run(785_330, 11, 64)
