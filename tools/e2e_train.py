#!/usr/bin/env python3
"""End-to-end time of the reference's batch-level training loop (model/train.py:18-48) on this
package: resident StreamSpot-sized graphs, batches of 12, everything the loop does per step
(transform_graph, batch, masking / sampling inside the model, backward, Adam).
usage: e2e_train.py [n_graphs] [epochs] [--profile]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from magic_amd import synth  # noqa: E402
from magic_amd.model.autoencoder import GMAEModel  # noqa: E402
from magic_amd.model.train import batch_level_train  # noqa: E402
from magic_amd.utils.utils import create_optimizer, set_random_seed  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n_graphs = int(args[0]) if args else 48
epochs = int(args[1]) if len(args) > 1 else 3
dev = torch.device("cuda:0")
n, e, n_dim, e_dim, hidden, layers = synth.SHAPES["streamspot"]
set_random_seed(0)
model = GMAEModel(n_dim, e_dim, hidden, layers, 4, "prelu", 0.1, 0.2, True, "BatchNorm", 0.5, "sce", 3).to(dev)
opt = create_optimizer("adam", model, 1e-3, 5e-4)
graphs = []
for i in range(n_graphs):
    g = synth.make_graph(n, e, n_dim, e_dim, seed=1002 + i, attrs=False).to(dev)
    g.structure.ensure_csr()
    graphs.append((g, 0))
loader = [list(range(i, i + 12)) for i in range(0, n_graphs, 12)]
batch_level_train(model, graphs, loader, opt, 1, dev, n_dim, e_dim, progress=False)   # warm-up
torch.cuda.synchronize()
steps = epochs * len(loader)
if "--profile" in sys.argv:
    pr = cProfile.Profile()
    pr.enable()
t0 = time.perf_counter()
batch_level_train(model, graphs, loader, opt, epochs, dev, n_dim, e_dim, progress=False)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
if "--profile" in sys.argv:
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
edges = sum(graphs[i][0].num_edges() for b in loader for i in b) * epochs
print(f"batch_level_train: {steps} steps, {dt / steps * 1e3:.2f} ms/step end to end, {edges / dt / 1e6:.1f} M edges/s")
