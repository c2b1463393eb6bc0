"""HIP path against the golden vectors captured from the reference (tests/golden/): the
scikit-learn searches and the reference's own KNN evaluators, its sce_loss, and a forward
pass with the shipped E3-THEIA checkpoint."""
import contextlib
import io
import json
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


@pytest.mark.parametrize("tag", ["d64k10", "d256k8", "d2048k2"])
def test_knn_equals_sklearn_fixture(cuda, tag):
    from magic_amd.knn import NearestNeighbors
    z = load(f"knn_raw_{tag}.npz")
    tr, q, k = z["train"].astype(np.float32), z["query"].astype(np.float32), int(z["k"])
    d, i = NearestNeighbors(k).fit(tr).kneighbors(q, k)
    assert d.dtype == np.float64 and i.dtype == np.int64
    assert np.array_equal(i, z["idx"])                       # bit-exact neighbour indices
    assert (np.abs(d - z["dist"]) < 2e-6 + 2e-7 * z["dist"]).all()


def test_knn_duplicates_fixture(cuda):
    from magic_amd.knn import NearestNeighbors
    z = load("knn_raw_dups.npz")
    d, i = NearestNeighbors(10).fit(z["train"]).kneighbors(z["query"], 10)
    assert (np.abs(d - z["dist"]) < 2e-6 + 2e-7 * z["dist"]).all()   # ties: distances only
    for row_i, row_w in zip(i, z["idx"]):                            # same neighbour rows up to ties
        assert np.array_equal(np.sort(z["train"][row_i], axis=0), np.sort(z["train"][row_w], axis=0))


@pytest.mark.parametrize("ds", ["trace", "cadets"])
def test_entity_level_evaluator_equals_reference(cuda, ds, tmp_path, monkeypatch):
    from magic_amd.model import eval as ev
    z = load("knn_entity.npz")
    monkeypatch.chdir(tmp_path)
    random.seed(1234)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        auc = ev.evaluate_entity_level_using_knn(ds, z[f"{ds}_x_train"].copy(), z[f"{ds}_x_test"].copy(),
                                                 z[f"{ds}_y_test"])[0]
    assert abs(auc - float(z[f"{ds}_auc"])) < 1e-9
    with np.load(tmp_path / "eval_result" / f"distance_save_{ds}.npz", allow_pickle=False) as c:   # written by the call above
        mean_distance, dists = float(c["mean_distance"]), c["distances"]
    assert abs(mean_distance - float(z[f"{ds}_mean_distance"])) < 1e-6
    assert np.abs(dists - z[f"{ds}_dists"]).max() < 2e-6
    # a second call is served from the cache; a cache of other arrays, or a reference-style .pkl file, is not used
    with contextlib.redirect_stdout(io.StringIO()):
        assert ev.evaluate_entity_level_using_knn(ds, z[f"{ds}_x_train"].copy(), z[f"{ds}_x_test"].copy(),
                                                  z[f"{ds}_y_test"])[0] == auc
        (tmp_path / "eval_result" / f"distance_save_{ds}.pkl").write_bytes(b"not a pickle this code would ever open")
        x_other = z[f"{ds}_x_test"].copy()
        x_other[0, 0] += 1.0
        random.seed(1234)
        auc_other = ev.evaluate_entity_level_using_knn(ds, z[f"{ds}_x_train"].copy(), x_other, z[f"{ds}_y_test"])[0]
    assert np.isfinite(auc_other)
    with np.load(tmp_path / "eval_result" / f"distance_save_{ds}.npz", allow_pickle=False) as c:
        assert not np.array_equal(c["distances"], dists)          # recomputed for the new arrays
    # same printed metric lines (scraped by the reference's web wrapper)
    want = str(z[f"{ds}_stdout"]).strip().splitlines()
    got = buf.getvalue().strip().splitlines()
    assert [l.split(":")[0] for l in got] == [l.split(":")[0] for l in want]
    for lg, lw in zip(got, want):
        if lg.split(":")[0] in ("TN", "FN", "TP", "FP"):
            assert lg == lw


@pytest.mark.parametrize("ds", ["streamspot", "wget"])
def test_batch_level_evaluator_equals_reference(cuda, ds):
    from magic_amd.model import eval as ev
    z = load("knn_batch.npz")
    x, y = z[f"{ds}_emb"].astype(np.float32), z[f"{ds}_labels"]
    with contextlib.redirect_stdout(io.StringIO()):
        auc, std = ev.evaluate_batch_level_using_knn(3, ds, x.copy(), y.copy())
        auc1, _ = ev.evaluate_batch_level_using_knn(-1, ds, x.copy(), y.copy())
    assert abs(auc - float(z[f"{ds}_auc_rep3"])) < 1e-9 and abs(std - float(z[f"{ds}_std_rep3"])) < 1e-9
    assert abs(auc1 - float(z[f"{ds}_auc_once"])) < 1e-9


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_sce_loss_equals_reference(cuda, tag):
    from magic_amd.model.loss_func import sce_loss
    z = load(f"sce_{tag}.npz")
    x = torch.from_numpy(z["x"]).to(cuda).requires_grad_(True)
    loss = sce_loss(x, torch.from_numpy(z["y"]).to(cuda), alpha=int(z["alpha"]))
    loss.backward()
    assert abs(float(loss.detach()) - float(z["loss"])) < 1e-6
    assert np.abs(x.grad.cpu().numpy() - z["grad_x"]).max() < 1e-6 * max(1.0, np.abs(z["grad_x"]).max())


def test_theia_checkpoint_embeddings(cuda):
    """Shipped E3-THEIA weights on a THEIA-shaped synthetic graph: embeddings within 1e-4
    (relative to the largest magnitude) of the fp64 oracle."""
    import copy
    from magic_amd.model.autoencoder import GMAEModel
    from oracle import gat_oracle as O
    from tests.util import rel_err, small_graph, to_graph, to_ograph
    z = load("checkpoint_theia.npz")
    sd = {k: torch.from_numpy(z[k]) for k in z.files}
    mine = GMAEModel(5, 17, 64, 3, 4, "prelu", 0.1, 0.2, True, "BatchNorm", 0.5, "sce", 3)
    mine.load_state_dict(sd)
    ref = O.build_model(5, 17, 64, 3)
    ref.load_state_dict(sd)
    n, e = 3000, 6000
    src, dst, nt, et = small_graph(n, e, 5, 17, seed=77, hub_deg=700)
    with torch.no_grad():
        want = copy.deepcopy(ref).double().eval().embed(to_ograph(src, dst, n, nt, et, 5, 17, torch.float64))
        got = mine.to(cuda).eval().embed(to_graph(src, dst, n, nt, et, 5, 17, cuda))
    assert rel_err(got, want) < 1e-4


def test_entity_level_evaluation_on_device_equals_host_pipeline(cuda, tmp_path, monkeypatch):
    """eval.py:62-121 end to end: embeddings -> test set -> KNN scores -> metric tail, everything on the
    GPU, against the same model's embeddings pushed through the numpy-shaped evaluator (whose arithmetic
    the reference-generated fixtures above pin): same confusion counts, same AUC."""
    from magic_amd import synth
    from magic_amd.model import eval as ev
    from magic_amd.model.autoencoder import GMAEModel
    monkeypatch.chdir(tmp_path)
    n_dim, e_dim = 5, 17
    torch.manual_seed(2)
    model = GMAEModel(n_dim, e_dim, 64, 3, 4, "prelu", 0.1, 0.2, True, "BatchNorm", 0.5, "sce", 3).to(cuda).eval()
    graphs = {("train", 0): synth.make_graph(3000, 9000, n_dim, e_dim, seed=1),
              ("train", 1): synth.make_graph(2500, 8000, n_dim, e_dim, seed=2),
              ("test", 0): synth.make_graph(2000, 6000, n_dim, e_dim, seed=3),
              ("test", 1): synth.make_graph(2600, 8000, n_dim, e_dim, seed=4)}
    rng = np.random.default_rng(0)
    malicious = sorted(rng.choice(4600, 60, replace=False).tolist())
    meta = {"malicious": [malicious, [str(m) for m in malicious]], "n_train": 2, "n_test": 2}
    random.seed(5)
    out_dev = io.StringIO()
    with contextlib.redirect_stdout(out_dev):
        auc_dev, _ = ev.entity_level_evaluation(model, "trace", cuda, metadata=meta, load_graph=lambda k, i: graphs[(k, i)])
    # the same pipeline piece by piece: device embeddings and test set, device scores, and scikit-learn
    # (the reference's metric functions) on those very scores
    from sklearn.metrics import roc_auc_score
    from magic_amd.model.metrics import entity_level_test_set
    with torch.no_grad():
        x_train = torch.cat([model.embed(graphs[("train", i)].to(cuda)) for i in range(2)])
        x_test = torch.cat([model.embed(graphs[("test", i)].to(cuda)) for i in range(2)])
    xs, ys = entity_level_test_set(x_test, malicious, 2000)
    y = np.zeros(x_test.shape[0])
    y[malicious] = 1.0
    keep = [i for i in range(x_test.shape[0]) if i >= 2000 or y[i] == 1.0]          # eval.py:107-110
    assert np.array_equal(ys.cpu().numpy(), y[keep]) and torch.equal(xs, x_test[torch.tensor(keep, device=cuda)])
    random.seed(5)
    m_dev, d_dev = ev.entity_level_distances(x_train, xs, 10)
    score = (d_dev / m_dev).cpu().numpy()
    assert abs(auc_dev - roc_auc_score(y[keep], score)) < 1e-12
    # ... and the numpy-shaped evaluator (pinned by the reference-generated fixtures above) sees the same
    # distances up to the float32 rounding of the standardisation
    random.seed(5)
    m_host, d_host = ev.entity_level_distances(x_train.cpu().numpy(), xs.cpu().numpy(), 10)
    assert abs(float(m_dev) - m_host) < 1e-5 * m_host
    assert np.abs(d_dev.cpu().numpy() - d_host).max() < 1e-5 * np.abs(d_host).max()
    dev_lines = out_dev.getvalue().strip().splitlines()
    assert [l.split(":")[0] for l in dev_lines] == ["AUC", "F1", "PRECISION", "RECALL", "TN", "FN", "TP", "FP", "#Test_AUC"]
    counts = {l.split(":")[0]: int(l.split(":")[1]) for l in dev_lines[4:8]}
    assert counts["TP"] + counts["FN"] == len(malicious) and sum(counts.values()) == len(keep)
