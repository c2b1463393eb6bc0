#!/usr/bin/env python3
"""Time the sparse sweeps of one GAT layer (forward + backward edge phase) on a synthetic graph.

Per-kernel microseconds from the library's own event brackets.  Run on the GPU box:
    python tools/ubench/gat_sweeps.py theia|streamspot_batch [reps] [d_y row width in floats]
Shapes timed: the encoder layer (H=4, D=hidden/4) and the decoder layer (H=1, D=8) of that workload.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from magic_amd import _lib, ops  # noqa: E402
from magic_amd.synth import SHAPES, make_graph  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "theia"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dy_width = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # 0: the model's packing (C + 2H rounded up to 4)
dev = torch.device("cuda:0")
n, e, n_dim, e_dim, hidden, _ = SHAPES["streamspot" if name == "streamspot_batch" else name]
if name == "streamspot_batch":
    from magic_amd.graph import batch
    gs = [make_graph(n, e, n_dim, e_dim, 1002 + i, attrs=False).to(dev) for i in range(12)]
    for g in gs:
        g.structure.ensure_csr()
        g.edata["attr"] = torch.nn.functional.one_hot(g.edata["type"], e_dim).float()
        g.mark_edge_attr_onehot()
    g = batch(gs)
else:
    g = make_graph(n, e, n_dim, e_dim, 1004, attrs=False).to(dev)
    g.edata["attr"] = torch.nn.functional.one_hot(g.edata["type"], e_dim).float()
    g.mark_edge_attr_onehot()
s = g.structure.ensure_csr()
N, E = s.num_nodes, s.num_edges
et_csc, et_csr = g.edge_types_csc(), g.edge_types_csr()
print(f"{name}: N={N} E={E}", flush=True)


def layer(H, D, with_act):
    C = H * D
    torch.manual_seed(0)
    hs = torch.randn(N, C, device=dev)
    score = torch.randn(N, 2 * H, device=dev)
    table = torch.randn(e_dim, H, device=dev)
    bias = torch.randn(C, device=dev)
    res = torch.randn(N, C, device=dev) if with_act else None
    prelu = torch.full((1,), 0.25, device=dev) if with_act else None
    d_out = torch.randn(N, C, device=dev)
    d_y = torch.empty(N, max(dy_width, (C + 2 * H + 3) // 4 * 4), device=dev)

    def once():
        out, pre, _, stat = ops.gat_fwd(s, H, D, hs, score[:, :H], score[:, H:], table, et_csc, None, 0.2, bias, res,
                                        prelu, False, want_attn=False, want_stat=True)
        ops.gat_bwd(s, H, D, hs, score[:, :H], stat, table, et_csr, None, e_dim, 0.2, None, None, None, d_out, d_out,
                    d_y, False)
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    _lib.prof_reset()
    _lib.prof_enable(True)
    for _ in range(reps):
        once()
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    prof = _lib.prof_read(128)
    _lib.prof_reset()
    tot = 0.0
    for k, (cnt, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:34s} {ms / cnt * 1e3:8.1f} us x{cnt // reps}", flush=True)
        tot += ms / reps
    print(f"  total {tot * 1e3:.1f} us per layer (H={H}, D={D})", flush=True)


layer(4, hidden // 4, True)
layer(1, 8, False)
