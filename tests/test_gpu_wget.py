"""Unicorn-Wget configuration (train.py:53-56, model/eval.py:59-68 of the reference): n_dim 8, e_dim 4,
hidden 256, 4 layers, one graph per step, graph embedding = per-node-type mean read-out [1, 8*256].
Small graphs against the float64 oracle and the reference-generated read-out fixture; the largest
Wget graph size (126,356 nodes / 351,388 edges) through size-independent properties."""
import copy
import os

import numpy as np
import pytest
import torch

from magic_amd import synth
from oracle import gat_oracle as O
from tests.util import rel_err, row_rel_err, small_graph, to_graph, to_ograph

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
N_DIM, E_DIM, HIDDEN, LAYERS = 8, 4, 256, 4


def _typed_readout_f64(feat, types, n_types, mean=True):
    out = np.zeros((1, n_types * feat.shape[1]))
    for t in range(n_types):
        rows = feat[types == t]
        if rows.shape[0]:
            out[0, t * feat.shape[1]:(t + 1) * feat.shape[1]] = rows.mean(0) if mean else rows.sum(0)
    return out


def test_readout_equals_reference_fixture(cuda):
    """pool.npz holds outputs of the reference's own utils/poolers.py::Pooling."""
    from magic_amd.graph import Graph
    from magic_amd.utils.poolers import Pooling
    z = np.load(os.path.join(GOLD, "pool.npz"))
    g = Graph(torch.zeros(1, dtype=torch.long), torch.zeros(1, dtype=torch.long), 500).to(cuda)
    g.ndata["type"] = torch.from_numpy(z["type"]).to(cuda)
    feat = torch.from_numpy(z["feat"]).to(cuda)
    assert np.allclose(Pooling("mean")(g, feat).cpu().numpy(), z["mean"], atol=1e-6)
    assert np.allclose(Pooling("sum")(g, feat).cpu().numpy(), z["sum"], atol=1e-5)
    mt = Pooling("mean")(g, feat, n_types=8)
    assert mt.shape == (1, 8 * 16)
    assert np.allclose(mt.cpu().numpy(), z["mean_typed"], atol=1e-6)
    assert np.allclose(Pooling("sum")(g, feat, n_types=8).cpu().numpy(), z["sum_typed"], atol=1e-5)
    assert bool((mt[0, 5 * 16:6 * 16] == 0).all())            # type 5 has no node: zeros
    with pytest.raises(NotImplementedError):
        Pooling("max")(g, feat)


@pytest.mark.parametrize("n,width,n_types", [(1, 256, 8), (127, 256, 8), (129, 64, 0), (5000, 256, 8),
                                             (3001, 40, 3), (70000, 256, 8), (999, 300, 32), (2000, 64, 33), (4000, 24, 70)])
def test_readout_against_float64(cuda, n, width, n_types):
    from magic_amd import _lib
    rng = np.random.default_rng(n + width)
    feat = rng.standard_normal((n, width)).astype(np.float32)
    types = rng.integers(0, max(n_types, 1), n)
    if n_types > 2:
        types[types == 1] = 0                                  # an absent type
    ft = torch.from_numpy(feat).to(cuda)
    tt = torch.from_numpy(types).to(cuda)
    for mean in (True, False):
        got = _lib.readout(ft, tt if n_types else None, n_types, mean)
        again = _lib.readout(ft, tt if n_types else None, n_types, mean)
        want = _typed_readout_f64(feat.astype(np.float64), types if n_types else np.zeros(n, np.int64),
                                  max(n_types, 1), mean)
        scale = np.abs(want).max() + 1e-30
        assert np.abs(got.cpu().numpy() - want).max() / scale < 2e-6
        assert torch.equal(got, again)                         # fixed summation order
    # a strided view (a column block of a wider buffer) and type ids outside [0, n_types)
    wide = torch.from_numpy(rng.standard_normal((n, width + 24)).astype(np.float32)).to(cuda)
    view = wide[:, 8:8 + width]
    t2 = tt.clone()
    t2[::7] = -1
    if n_types:
        got = _lib.readout(view, t2, n_types, True)
        ty = types.copy()
        ty[::7] = -1
        want = _typed_readout_f64(view.cpu().numpy().astype(np.float64), ty, n_types, True)
        assert np.abs(got.cpu().numpy() - want).max() / (np.abs(want).max() + 1e-30) < 2e-6
    # no rows: torch's mean over an empty node set is NaN, its sum and the per-type read-outs are zeros
    empty = ft[:0]
    assert bool(torch.isnan(_lib.readout(empty, None, 0, True)).all())
    assert not bool(_lib.readout(empty, None, 0, False).any())
    assert not bool(_lib.readout(empty, tt[:0], 3, True).any())


def test_wget_pooled_graph_embedding_matches_oracle(cuda):
    """model/eval.py:59-68: transform_graph -> embed -> Pooling(g, out, n_types=8)."""
    from magic_amd.model.autoencoder import GMAEModel
    from magic_amd.utils.poolers import Pooling
    n, e = 2500, 6500
    src, dst, ntype, etype = small_graph(n, e, N_DIM, E_DIM, seed=77, hub_deg=300, hub_out=120)
    ntype[ntype == 6] = 2                                      # Wget graphs do not hold every type
    torch.manual_seed(3)
    ref = O.build_model(N_DIM, E_DIM, HIDDEN, LAYERS).eval()
    torch.manual_seed(3)
    mine = GMAEModel(N_DIM, E_DIM, HIDDEN, LAYERS, 4, "prelu", 0.1, 0.2, True, "BatchNorm", 0.5, "sce", 3).to(cuda).eval()
    og = to_ograph(src, dst, n, ntype, etype, N_DIM, E_DIM, torch.float64)
    g = to_graph(src, dst, n, ntype, etype, N_DIM, E_DIM, cuda)
    with torch.no_grad():
        eo = copy.deepcopy(ref).double().embed(og)
        em = mine.embed(g)
        pooled = Pooling("mean")(g, em, n_types=N_DIM)
    assert rel_err(em, eo) < 1e-4 and row_rel_err(em, eo) < 1e-4
    want = _typed_readout_f64(eo.numpy(), ntype, N_DIM)
    assert pooled.shape == (1, N_DIM * HIDDEN)
    assert np.abs(pooled.cpu().numpy() - want).max() / np.abs(want).max() < 1e-4
    assert bool((pooled[0, 6 * HIDDEN:7 * HIDDEN] == 0).all())


@pytest.fixture(scope="module")
def wget_max(cuda):
    n, e = synth.SHAPES["wget_max"][:2]
    g = synth.make_graph(n, e, N_DIM, E_DIM, seed=1003, attrs=True).to(cuda)
    g.structure.ensure_csr()
    return g


def test_wget_largest_graph_properties(wget_max, cuda):
    """126,356 nodes / 351,388 edges (analysis_result/analysis_wget.txt): attention rows sum to one,
    the read-out equals a float64 index_add of the same embeddings, a training step is bitwise
    reproducible and its gradient predicts a central difference of the loss."""
    from magic_amd.model.autoencoder import GMAEModel
    from magic_amd.utils.poolers import Pooling
    g = wget_max
    n, e = g.num_nodes(), g.num_edges()
    torch.manual_seed(0)
    model = GMAEModel(N_DIM, E_DIM, HIDDEN, LAYERS, 4, "prelu", 0.1, 0.2, True, "BatchNorm", 0.5, "sce", 3).to(cuda)
    with torch.no_grad():
        model.enc_mask_token.normal_(0, 0.3)
    model.eval()
    with torch.no_grad():
        out, attn = model.encoder.gats[0](g, g.ndata["attr"], get_attention=True)
        emb = model.embed(g)
        pooled = Pooling("mean")(g, emb, n_types=N_DIM)
    _, dst = g.edges()
    sums = torch.zeros(n, 4, device=cuda, dtype=torch.float64).index_add_(0, dst, attn[:, :, 0].double())
    deg = torch.bincount(dst, minlength=n)
    assert float((sums[deg > 0] - 1).abs().max()) < 1e-5
    ty = g.ndata["type"]
    ref = torch.zeros(N_DIM, HIDDEN, device=cuda, dtype=torch.float64).index_add_(0, ty, emb.double())
    cnt = torch.bincount(ty, minlength=N_DIM).clamp(min=1).double().view(-1, 1)
    ref = (ref / cnt).view(1, -1)
    assert float((pooled.double() - ref).abs().max() / ref.abs().max()) < 1e-5
    # training step: reproducible, finite, directional derivative
    model.train()
    gen = torch.Generator().manual_seed(4)
    perm = torch.randperm(n, generator=gen).to(cuda)
    pos = torch.randint(0, e, (10000,), generator=gen)
    neg = (torch.randint(0, n, (10000,), generator=gen), torch.randint(0, n, (10000,), generator=gen))
    widths = [N_DIM] + [HIDDEN] * LAYERS
    masks = [(torch.rand(n, w, device=cuda) < 0.9) for w in widths]

    def run():
        g.ndata["attr"] = torch.nn.functional.one_hot(g.ndata["type"], N_DIM).float()
        model.zero_grad(set_to_none=True)
        loss = model.compute_loss(g, perm=perm, pos_idx=pos, neg_pairs=neg, drop_masks=masks)
        loss.backward()
        return loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    l1, g1 = run()
    l2, g2 = run()
    assert torch.equal(l1, l2)
    for k in g1:
        assert torch.equal(g1[k], g2[k]) and bool(torch.isfinite(g1[k]).all()), k
    names = ["encoder.gats.2.fc.weight", "encoder.gats.0.attn_e", "decoder.gats.0.fc.weight", "encoder_to_decoder.weight"]
    params = dict(model.named_parameters())
    norm = sum(float((g1[k].double() ** 2).sum()) for k in names) ** 0.5
    eps = 0.05 / norm      # a step that moves the O(1) loss by about +-0.05: far above fp32 noise, little curvature

    def shift(scale):
        with torch.no_grad():
            for k in names:
                params[k].add_(g1[k] * (scale * eps / norm))
    shift(1)
    lp = float(run()[0])
    shift(-2)
    lm = float(run()[0])
    shift(1)
    assert abs((lp - lm) / (2 * eps) - norm) / norm < 0.05


def test_edge_attribute_cache_returns_its_budget(cuda):
    """transform_graph keeps the one-hot edge attributes of a GPU graph for the next call; the bytes count
    against a per-process budget and come back when the entry is replaced or the graph is freed."""
    import gc
    from magic_amd.utils import loaddata as L
    src, dst, nt, et = small_graph(300, 1500, 8, 4, seed=21)
    g = to_graph(src, dst, 300, nt, et, 8, 4, cuda)
    before = L._edge_attr_bytes
    a1 = L.transform_graph(g, 8, 4).edata["attr"]
    a2 = L.transform_graph(g, 8, 4).edata["attr"]
    assert a1 is a2 and L._edge_attr_bytes == before + 1500 * 4 * 4
    g.edata["type"] = g.edata["type"].clone()            # another type tensor: the entry is rebuilt, not leaked
    a3 = L.transform_graph(g, 8, 4).edata["attr"]
    assert a3 is not a1 and torch.equal(a3, a1) and L._edge_attr_bytes == before + 1500 * 4 * 4
    del a1, a2, a3, g
    gc.collect()
    assert L._edge_attr_bytes == before
