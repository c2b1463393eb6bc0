#!/usr/bin/env python3
"""cProfile of the host side of one training step on a small graph (launch-bound regime)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from magic_amd import synth  # noqa: E402
from magic_amd.model.autoencoder import GMAEModel  # noqa: E402
from magic_amd.utils.utils import create_optimizer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "streamspot_real"
if name == "e2e":
    # the reference's batch-level loop itself (model/train.py:18-48) on a batch of 12 StreamSpot-sized graphs
    from magic_amd.model.train import batch_level_train
    n, e, n_dim, e_dim, hidden, layers = synth.SHAPES["streamspot"]
    dev = torch.device("cuda:0")
    parts = [synth.make_graph(n, e, n_dim, e_dim, seed=1002 + i, attrs=False).to(dev) for i in range(12)]
    for p_ in parts:
        p_.structure.ensure_csr()
    model = GMAEModel(n_dim, e_dim, hidden, layers, 4, "prelu", 0.1, 0.2, True, "BatchNorm", 0.5, "sce", 3).to(dev)
    opt = create_optimizer("adam", model, 1e-3, 5e-4)
    pool = [(p_, 0) for p_ in parts]
    loader = [list(range(12))]
    batch_level_train(model, pool, loader, opt, 3, dev, n_dim, e_dim, progress=False)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    batch_level_train(model, pool, loader * 20, opt, 1, dev, n_dim, e_dim, progress=False)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(70)
    sys.exit(0)
n, e, n_dim, e_dim, hidden, layers = synth.SHAPES[name]
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = GMAEModel(n_dim, e_dim, hidden, layers, 4, "prelu", 0.1, 0.2, True, "BatchNorm", 0.5, "sce", 3).to(dev).train()
opt = create_optimizer("adam", model, 1e-3, 5e-4)
g = synth.make_graph(n, e, n_dim, e_dim, seed=1, attrs=True).to(dev)
g.structure.ensure_csr()
g.edge_types_csc()
thr = min(10000, n)
gen = torch.Generator(device=dev).manual_seed(1)
perm = torch.randperm(n, device=dev, generator=gen)
pos = torch.randint(0, e, (thr,), device=dev, generator=gen)
neg = (torch.randint(0, n, (thr,), device=dev, generator=gen), torch.randint(0, n, (thr,), device=dev, generator=gen))


def step():
    g.ndata["attr"] = torch.nn.functional.one_hot(g.ndata["type"], n_dim).float()
    loss = model.compute_loss(g, perm=perm, pos_idx=pos, neg_pairs=neg)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
