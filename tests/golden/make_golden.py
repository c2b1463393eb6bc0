#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference itself.

Run in the authoring container only (it reads /root/reference; nothing at test time does):

    python tests/golden/make_golden.py

What is captured
  sce_*.npz         model/loss_func.py::sce_loss outputs and input gradients (torch only)
  knn_entity.npz    model/eval.py::evaluate_entity_level_using_knn on seeded arrays: the cache
                    it writes ([mean_distance, per-row mean k-distance]) and the returned AUC
  knn_batch.npz     model/eval.py::evaluate_batch_level_using_knn (repeat=-1 and repeat=3)
  knn_raw_*.npz     sklearn NearestNeighbors.kneighbors (the reference's search) on seeded data
  pool.npz          utils/poolers.py::Pooling outputs
  checkpoint_manifest.json   parameter names / shapes of the four shipped checkpoints
  checkpoint_theia.npz       the E3-THEIA checkpoint's tensors (data file, loaded weights_only)
  randperm.json     torch.manual_seed(s); torch.randperm(n) prefixes (mask indices of
                    model/autoencoder.py:201-206 on the CPU generator)

model/eval.py imports utils/loaddata.py, which subclasses dgl.data.DGLDataset at import time;
an inert placeholder module named `dgl` (no arithmetic, only that attribute) lets the import
succeed.  The GAT files themselves cannot be exercised this way: their arithmetic IS DGL.
"""
import contextlib
import io
import json
import os
import random
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)


def inert_dgl():
    dgl = types.ModuleType("dgl")
    data = types.ModuleType("dgl.data")

    class DGLDataset:  # attribute looked up by `class StreamspotDataset(dgl.data.DGLDataset)`
        pass
    data.DGLDataset = DGLDataset
    dgl.data = data
    sys.modules["dgl"] = dgl
    sys.modules["dgl.data"] = data


def gen_sce():
    from model.loss_func import sce_loss
    for tag, n, d, alpha, seed in [("a", 257, 5, 3, 0), ("b", 100, 11, 2, 1), ("c", 64, 8, 1, 2)]:
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(n, d, generator=g, requires_grad=True)
        y = torch.nn.functional.one_hot(torch.randint(0, d, (n,), generator=g), d).float()
        loss = sce_loss(x, y, alpha=alpha)
        loss.backward()
        np.savez(os.path.join(OUT, f"sce_{tag}.npz"), x=x.detach().numpy(), y=y.numpy(), alpha=alpha,
                 loss=loss.detach().numpy(), grad_x=x.grad.numpy())


def mixture(n, d, seed, clusters=20, sigma=0.3):
    rng = np.random.default_rng(seed)
    c = rng.standard_normal((clusters, d)).astype(np.float32)
    return (c[rng.integers(0, clusters, n)] + sigma * rng.standard_normal((n, d))).astype(np.float32)


def gen_knn():
    inert_dgl()
    import pickle
    from model import eval as ref_eval
    from sklearn.neighbors import NearestNeighbors
    # --- entity level (k=10 for 'trace', k=200 for 'cadets')
    out = {}
    for ds, ntr, nte, d in [("trace", 3000, 700, 64), ("cadets", 1500, 300, 64)]:
        x_train = mixture(ntr, d, 10)
        x_test = np.concatenate([mixture(nte - 60, d, 11), mixture(60, d, 12) + 1.5]).astype(np.float32)
        y_test = np.concatenate([np.zeros(nte - 60), np.ones(60)])
        cwd = os.getcwd()
        with tempfile.TemporaryDirectory() as tmp:
            os.makedirs(os.path.join(tmp, "eval_result"))
            os.chdir(tmp)
            try:
                random.seed(1234)     # the function shuffles the train index list with `random`
                buf = io.StringIO()
                with contextlib.redirect_stdout(buf):
                    auc = ref_eval.evaluate_entity_level_using_knn(ds, x_train.copy(), x_test.copy(), y_test)[0]
                with open(f"./eval_result/distance_save_{ds}.pkl", "rb") as f:   # written by this very call
                    mean_distance, dists = pickle.load(f)
            finally:
                os.chdir(cwd)
        out.update({f"{ds}_x_train": x_train, f"{ds}_x_test": x_test, f"{ds}_y_test": y_test,
                    f"{ds}_auc": auc, f"{ds}_mean_distance": mean_distance, f"{ds}_dists": dists,
                    f"{ds}_stdout": np.array(buf.getvalue())})
    np.savez_compressed(os.path.join(OUT, "knn_entity.npz"), **out)
    # --- batch level
    out = {}
    for ds, n, d, n_attack in [("streamspot", 600, 256, 100), ("wget", 150, 2048, 25)]:
        emb = mixture(n, d, 20, clusters=5, sigma=0.5)
        labels = np.zeros(n, dtype=np.int64)
        labels[-n_attack:] = 1
        emb[-n_attack:] += 0.8
        if d > 1000:   # stored as float16 to keep the fixture small: make the values exact in it
            emb = emb.astype(np.float16).astype(np.float32)
        res = {}
        for rep in (-1, 3):
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                auc, std = ref_eval.evaluate_batch_level_using_knn(rep, ds, emb.copy(), labels.copy())
            res[rep] = (auc, std, buf.getvalue())
        out.update({f"{ds}_emb": emb.astype(np.float16) if d > 1000 else emb, f"{ds}_labels": labels,
                    f"{ds}_auc_once": res[-1][0], f"{ds}_stdout_once": np.array(res[-1][2]),
                    f"{ds}_auc_rep3": res[3][0], f"{ds}_std_rep3": res[3][1],
                    f"{ds}_stdout_rep3": np.array(res[3][2])})
    np.savez_compressed(os.path.join(OUT, "knn_batch.npz"), **out)
    # --- raw searches
    for tag, ntr, nq, d, k, dup in [("d64k10", 4000, 500, 64, 10, False), ("d256k8", 400, 200, 256, 8, False),
                                    ("d2048k2", 100, 50, 2048, 2, False), ("dups", 600, 100, 64, 10, True)]:
        tr = mixture(ntr, d, 30)
        q = mixture(nq, d, 31)
        if d > 1000:   # stored as float16: make the values exact in it
            tr, q = tr.astype(np.float16).astype(np.float32), q.astype(np.float16).astype(np.float32)
        if dup:
            tr[100:400] = tr[:300]
        nb = NearestNeighbors(n_neighbors=k, n_jobs=-1).fit(tr)
        dist, idx = nb.kneighbors(q, n_neighbors=k)
        np.savez_compressed(os.path.join(OUT, f"knn_raw_{tag}.npz"),
                            train=tr.astype(np.float16) if d > 1000 else tr,
                            query=q.astype(np.float16) if d > 1000 else q, k=k, dist=dist, idx=idx,
                            algorithm=np.array(nb._fit_method))


def gen_pool():
    from utils.poolers import Pooling

    class G:
        ndata = {}

        @contextlib.contextmanager
        def local_scope(self):
            yield
    g = G()
    gen = torch.Generator().manual_seed(3)
    feat = torch.randn(500, 16, generator=gen)
    g.ndata["type"] = torch.randint(0, 8, (500,), generator=gen)
    g.ndata["type"][g.ndata["type"] == 5] = 4          # leave type 5 empty
    np.savez(os.path.join(OUT, "pool.npz"), feat=feat.numpy(), type=g.ndata["type"].numpy(),
             mean=Pooling("mean")(g, feat).numpy(), sum=Pooling("sum")(g, feat).numpy(),
             mean_typed=Pooling("mean")(g, feat, n_types=8).numpy(),
             sum_typed=Pooling("sum")(g, feat, n_types=8).numpy())


def gen_checkpoints():
    man = {}
    for ds in ("theia", "trace", "cadets", "fivedirections"):
        sd = torch.load(os.path.join(REF, "checkpoints", f"checkpoint-{ds}.pt"), map_location="cpu",
                        weights_only=True)
        man[ds] = {k: list(v.shape) for k, v in sd.items()}
        if ds == "theia":
            np.savez_compressed(os.path.join(OUT, "checkpoint_theia.npz"), **{k: v.numpy() for k, v in sd.items()})
    json.dump(man, open(os.path.join(OUT, "checkpoint_manifest.json"), "w"), indent=1)


def gen_randperm():
    out = {}
    for seed in (0, 1, 7):
        for n in (10, 8407, 100890, 700000):
            torch.manual_seed(seed)
            p = torch.randperm(n)
            out[f"{seed}_{n}"] = {"head": p[:16].tolist(), "tail": p[-4:].tolist(), "sum_head_half": int(p[: n // 2].sum())}
    json.dump(out, open(os.path.join(OUT, "randperm.json"), "w"), indent=1)


if __name__ == "__main__":
    gen_sce()
    gen_knn()
    gen_pool()
    gen_checkpoints()
    gen_randperm()
    print("golden fixtures written to", OUT)
