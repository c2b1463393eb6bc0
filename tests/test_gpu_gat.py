"""HIP GAT layer / auto-encoder vs the torch-CPU oracle (fp32 and fp64 anchor).

Tolerance: north star asks embeddings / loss within 1e-4 relative in fp32; layer outputs and
gradients are compared against the fp64 oracle at 2e-4 of the tensor's max magnitude (the
fp32 oracle itself sits at ~1e-6 from the fp64 one)."""
import copy

import numpy as np
import pytest
import torch

from tests.util import rel_err, row_rel_err, small_graph, to_graph, to_ograph
from oracle import gat_oracle as O

pytestmark = pytest.mark.gpu
TOL = 2e-4


def _layer_pair(in_dim, e_dim, out_dim, heads, residual, act, seed, dev):
    from magic_amd.model.gat import GATConv
    torch.manual_seed(seed)
    act_o = torch.nn.PReLU() if act else None
    ref = O.GATConv(in_dim, e_dim, out_dim, heads, 0.0, 0.0, 0.2, residual, act_o, norm=None, concat_out=True)
    with torch.no_grad():
        if ref.bias is not None:
            ref.bias.normal_(0, 0.1)
        if act:
            ref.activation.weight.fill_(0.3)
    mine = GATConv(in_dim, e_dim, out_dim, heads, 0.0, 0.0, 0.2, residual, torch.nn.PReLU() if act else None,
                   norm=None, concat_out=True)
    mine.load_state_dict(ref.state_dict())
    return ref.double(), mine.to(dev)


@pytest.mark.parametrize("cfg", [
    # in_dim, e_dim, out_dim(per head), heads, residual, prelu, n, e, hub_in, hub_out
    (64, 17, 16, 4, True, True, 600, 2500, 300, 200),     # entity-level encoder layer (identity residual)
    (5, 17, 16, 4, True, True, 500, 2000, 150, 0),        # layer 0: linear residual, tiny input width
    (64, 17, 5, 1, False, False, 400, 1500, 100, 90),     # decoder: 1 head, width 5 (padded to 8)
    (256, 20, 64, 4, True, True, 300, 1800, 280, 270),    # batch-level encoder layer
    (64, 23, 11, 1, False, False, 300, 900, 0, 0),        # decoder width 11 (padded to 16)
    (8, 4, 64, 4, True, True, 64, 0, 0, 0),               # no edges at all
])
def test_gatconv_forward_backward(cuda, cfg):
    in_dim, e_dim, out_dim, heads, residual, act, n, e, hub_in, hub_out = cfg
    if e == 0:
        src = dst = np.zeros(0, np.int64)
        rng = np.random.default_rng(0)
        ntype, etype = rng.integers(0, 4, n), np.zeros(0, np.int64)
    else:
        src, dst, ntype, etype = small_graph(n, e, 4, e_dim, seed=7, hub_deg=hub_in, hub_out=hub_out)
    og = to_ograph(src, dst, n, ntype, etype, 4, e_dim, torch.float64)
    g = to_graph(src, dst, n, ntype, etype, 4, e_dim, cuda)
    ref, mine = _layer_pair(in_dim, e_dim, out_dim, heads, residual, act, 3, cuda)
    torch.manual_seed(11)
    x = torch.randn(n, in_dim)
    xo = x.double().requires_grad_(True)
    xm = x.to(cuda).requires_grad_(True)
    yo, ao = ref(og, xo, get_attention=True)
    ym, am = mine(g, xm, get_attention=True)
    assert ym.shape == yo.shape
    assert rel_err(ym, yo) < TOL
    if e:
        assert rel_err(am, ao) < TOL
    w = torch.randn_like(yo)
    (yo * w).sum().backward()
    (ym * w.float().to(cuda)).sum().backward()
    assert rel_err(xm.grad, xo.grad) < TOL
    for (k, po), (_, pm) in zip(ref.named_parameters(), mine.named_parameters()):
        if po.grad is None:
            assert pm.grad is None or float(pm.grad.abs().max()) == 0.0, k
            continue
        assert pm.grad is not None, k
        assert rel_err(pm.grad, po.grad) < TOL, k


@pytest.mark.parametrize("in_dim,out_dim,heads", [(64, 16, 4), (256, 64, 4), (64, 8, 1), (64, 16, 1), (32, 8, 4)])
def test_degree_ladder_every_segment_length_class(cuda, in_dim, out_dim, heads):
    """In-degrees 0 .. 70 and a few hubs, one node each: the register path (<= 4 edges), the LDS strip with
    one to several edges per lane, the sub-streams of narrow rows, the light / heavy boundary (64 / 65) and
    chunked hubs (256 / 257 / 600), out-degrees spread alike.  Every ROW is compared, not the tensor's
    maximum, so a wrong segment class cannot hide behind the others."""
    degs = list(range(0, 71)) + [127, 128, 129, 255, 256, 257, 600]
    n = 800
    rng = np.random.default_rng(3)
    dst = np.concatenate([np.full(d, v, np.int64) for v, d in enumerate(degs)])
    src = np.concatenate([rng.choice(n, d, replace=False) for d in degs]).astype(np.int64)
    ntype, etype = rng.integers(0, 4, n), rng.integers(0, 7, src.size)
    og = to_ograph(src, dst, n, ntype, etype, 4, 7, torch.float64)
    g = to_graph(src, dst, n, ntype, etype, 4, 7, cuda)
    ref, mine = _layer_pair(in_dim, 7, out_dim, heads, heads > 1, heads > 1, 5, cuda)
    torch.manual_seed(13)
    x = torch.randn(n, in_dim)
    xo, xm = x.double().requires_grad_(True), x.to(cuda).requires_grad_(True)
    yo, ao = ref(og, xo, get_attention=True)
    ym, am = mine(g, xm, get_attention=True)
    scale = float(yo.detach().abs().max())
    row_err = (ym.detach().double().cpu() - yo.detach()).abs().amax(1) / scale
    assert float(row_err.max()) < TOL, int(row_err.argmax())
    assert rel_err(am, ao) < TOL
    w = torch.randn_like(yo)
    (yo * w).sum().backward()
    (ym * w.float().to(cuda)).sum().backward()
    gscale = float(xo.grad.abs().max())
    grow = (xm.grad.double().cpu() - xo.grad).abs().amax(1) / gscale
    assert float(grow.max()) < TOL, int(grow.argmax())
    for (k, po), (_, pm) in zip(ref.named_parameters(), mine.named_parameters()):
        if po.grad is not None:
            assert rel_err(pm.grad, po.grad) < TOL, k


def test_dense_edge_attr_path(cuda):
    """edata['attr'] that is not one-hot takes the dense per-edge score path."""
    n, e_dim = 300, 6
    src, dst, ntype, etype = small_graph(n, 1200, 4, e_dim, seed=5, hub_deg=120)
    og = to_ograph(src, dst, n, ntype, etype, 4, e_dim, torch.float64)
    g = to_graph(src, dst, n, ntype, etype, 4, e_dim, cuda)
    torch.manual_seed(2)
    dense = torch.rand(src.size, e_dim)
    og.edata["attr"] = dense.double()
    g.edata["attr"] = dense.to(cuda)
    del g.edata["type"]
    ref, mine = _layer_pair(64, e_dim, 16, 4, True, True, 4, cuda)
    x = torch.randn(n, 64)
    xo, xm = x.double().requires_grad_(True), x.to(cuda).requires_grad_(True)
    yo, ym = ref(og, xo), mine(g, xm)
    assert rel_err(ym, yo) < TOL
    yo.sum().backward()
    ym.sum().backward()
    assert rel_err(xm.grad, xo.grad) < TOL
    assert rel_err(mine.edge_fc.weight.grad, ref.edge_fc.weight.grad) < TOL
    assert rel_err(mine.attn_e.grad, ref.attn_e.grad) < TOL


def _model_pair(n_dim, e_dim, hidden, layers, dev, seed=0, alias=True):
    from magic_amd.model.autoencoder import GMAEModel
    torch.manual_seed(seed)
    ref = O.build_model(n_dim, e_dim, hidden, layers, alias_clone=alias)
    torch.manual_seed(seed)
    mine = GMAEModel(n_dim, e_dim, hidden, layers, 4, "prelu", 0.1, 0.2, True, "BatchNorm", 0.5, "sce", 3,
                     alias_clone=alias)
    # identical construction order and RNG consumption => identical initial weights
    for (k1, v1), (k2, v2) in zip(ref.state_dict().items(), mine.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2), k1
    with torch.no_grad():  # a non-zero mask token so that both SCE branches are exercised
        ref.enc_mask_token.normal_(0, 0.3)
        mine.enc_mask_token.copy_(ref.enc_mask_token)
    return ref, mine.to(dev)


@pytest.mark.parametrize("shape,alias,mode", [
    ((5, 17, 64, 3), True, "eval"), ((5, 17, 64, 3), False, "eval"), ((8, 20, 256, 4), True, "eval"),
    ((5, 17, 64, 3), True, "perlayer"),      # layer-by-layer autograd path
    ((5, 17, 64, 3), True, "drop"),          # training mode, dropout masks injected on both sides
    ((12, 9, 128, 2), False, "drop"),
    ((5, 17, 64, 3), True, "negslope"),      # PReLU slopes <= 0: pre-activation must come from storage
    ((6, 7, 32, 2), True, "drop"),
    ((8, 4, 256, 4), True, "drop"),          # Unicorn-Wget dims (train.py:53-56)
    ((8, 4, 256, 4), False, "eval"),
    ((5, 17, 64, 3), True, "tiny"),          # 33 nodes / 70 edges: ragged last row tile everywhere
])
def test_autoencoder_loss_and_grads(cuda, shape, alias, mode):
    n_dim, e_dim, hidden, layers = shape
    n, e = 700, 3000
    if mode == "tiny":
        n, e = 33, 70
        src, dst, ntype, etype = small_graph(n, e, n_dim, e_dim, seed=22)
    else:
        src, dst, ntype, etype = small_graph(n, e, n_dim, e_dim, seed=21, hub_deg=200, hub_out=150)
    ref, mine = _model_pair(n_dim, e_dim, hidden, layers, cuda, alias=alias)
    gen = torch.Generator().manual_seed(5)
    drop_o = drop_m = None
    if mode in ("drop", "tiny"):
        ref.train()
        mine.train()
        widths = [n_dim] + [hidden] * (layers - 1) + [hidden]
        keep = [(torch.rand(n, w, generator=gen) < 0.9) for w in widths]
        drop_o = [k.double() for k in keep]
        drop_m = [k.to(cuda) for k in keep]
    else:
        ref.eval()
        mine.eval()   # dropout off on both sides; everything else is the training graph
    if mode == "negslope":
        with torch.no_grad():
            for i, (lo_, lm_) in enumerate(zip(ref.encoder.gats, mine.encoder.gats)):
                v = (-0.3, 0.0, 0.25)[i % 3]
                lo_.activation.weight.fill_(v)
                lm_.activation.weight.fill_(v)
    perm = torch.randperm(n, generator=gen)
    pos_idx = torch.randperm(src.size, generator=gen)[:n]
    og64 = to_ograph(src, dst, n, ntype, etype, n_dim, e_dim, torch.float64)
    neg = O.uniform_negative_pairs(og64, n, seed=9)
    ref64 = copy.deepcopy(ref).double()
    lo, parts_o = ref64.compute_loss(og64, perm=perm, pos_idx=pos_idx, neg_pairs=neg, drop_masks=drop_o,
                                     return_parts=True)
    g = to_graph(src, dst, n, ntype, etype, n_dim, e_dim, cuda)
    lm, parts_m = mine.compute_loss(g, perm=perm.to(cuda), pos_idx=pos_idx, neg_pairs=neg, drop_masks=drop_m,
                                    return_parts=True, fused=mode != "perlayer")
    assert abs(float(lm) - float(lo)) / abs(float(lo)) < 1e-4
    assert abs(float(parts_m["sce"]) - float(parts_o["sce"])) / abs(float(parts_o["sce"])) < 1e-4
    for hm, ho in zip(parts_m["hidden"], parts_o["hidden"]):
        assert rel_err(hm, ho) < 1e-4
        assert row_rel_err(hm, ho) < 1e-4          # every node's embedding row, relative to itself
    assert rel_err(parts_m["recon"], parts_o["recon"]) < 1e-4
    assert row_rel_err(parts_m["recon"], parts_o["recon"]) < 1e-4
    lo.backward()
    lm.backward()
    for (k, po), (_, pm) in zip(ref64.named_parameters(), mine.named_parameters()):
        if po.grad is None or float(po.grad.abs().max()) == 0.0:
            continue
        assert pm.grad is not None, k
        assert rel_err(pm.grad, po.grad) < 5e-4, k


def test_fused_chain_matches_layer_path_and_is_reproducible(cuda):
    """Same arithmetic in both paths: results agree to rounding, and the fused chain is bitwise
    reproducible run to run (fixed-order reductions, no atomics)."""
    n_dim, e_dim, hidden, layers = 9, 13, 64, 3
    n, e = 1500, 9000
    src, dst, ntype, etype = small_graph(n, e, n_dim, e_dim, seed=4, hub_deg=500, hub_out=300)
    _, mine = _model_pair(n_dim, e_dim, hidden, layers, cuda)
    mine.eval()
    gen = torch.Generator().manual_seed(1)
    perm = torch.randperm(n, generator=gen).to(cuda)
    pos_idx = torch.randperm(src.size, generator=gen)[:n]
    neg = (torch.randint(0, n, (n,), generator=gen), torch.randint(0, n, (n,), generator=gen))
    outs = []
    for fused in (True, True, False):
        # a fresh graph per pass: the token write of the masking step lands in the graph's own
        # feature tensor (shared storage), which then carries autograd history
        g = to_graph(src, dst, n, ntype, etype, n_dim, e_dim, cuda)
        mine.zero_grad(set_to_none=True)
        loss = mine.compute_loss(g, perm=perm, pos_idx=pos_idx, neg_pairs=neg, fused=fused)
        loss.backward()
        outs.append((loss.detach().clone(), [p.grad.clone() for p in mine.parameters() if p.grad is not None]))
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)
    assert abs(float(outs[0][0]) - float(outs[2][0])) < 1e-5 * abs(float(outs[2][0]))
    for a, b in zip(outs[0][1], outs[2][1]):
        assert rel_err(a, b) < 1e-4


@pytest.mark.parametrize("shape", [(5, 17, 64, 3), (13, 9, 256, 2), (8, 4, 128, 2), (11, 5, 32, 2)])
def test_coded_first_layer_equals_dense_projection(cuda, shape):
    """One-hot node types in, the first encoder layer either on code words + a [n_dim, M] table (no projection
    launch, no [N, C] operand) or on the dense projection of the materialised one-hot matrix: same loss, hidden
    states and gradients to rounding, with dropout masks and a non-zero mask token; the coded path must really be
    the one that ran (its kernels show up in the library's launch log) and is bitwise reproducible."""
    from magic_amd import _lib
    n_dim, e_dim, hidden, layers = shape
    n, e = 900, 5000
    src, dst, ntype, etype = small_graph(n, e, n_dim, e_dim, seed=31, hub_deg=300, hub_out=200)
    _, mine = _model_pair(n_dim, e_dim, hidden, layers, cuda)
    mine.train()
    gen = torch.Generator().manual_seed(2)
    widths = [n_dim] + [hidden] * (layers - 1) + [hidden]
    drop = [(torch.rand(n, w, generator=gen) < 0.9).to(cuda) for w in widths]
    perm = torch.randperm(n, generator=gen).to(cuda)
    pos_idx = torch.randperm(src.size, generator=gen)[:n]
    neg = (torch.randint(0, n, (n,), generator=gen), torch.randint(0, n, (n,), generator=gen))
    outs = []
    for coded in (True, True, False):
        mine.coded_input = coded
        g = to_graph(src, dst, n, ntype, etype, n_dim, e_dim, cuda)
        mine.zero_grad(set_to_none=True)
        _lib.prof_reset()
        _lib.prof_enable(True)
        loss, parts = mine.compute_loss(g, perm=perm, pos_idx=pos_idx, neg_pairs=neg, drop_masks=drop, return_parts=True)
        loss.backward()
        _lib.prof_enable(False)
        names = set(_lib.prof_read(160))
        _lib.prof_reset()
        assert ("node_code" in names) == coded and ("coded_dw_partial" in names) == coded, names
        outs.append((loss.detach().clone(), [h.detach().clone() for h in parts["hidden"]],
                     {k: p.grad.clone() for k, p in mine.named_parameters() if p.grad is not None}))
    mine.coded_input = True
    assert torch.equal(outs[0][0], outs[1][0])
    for k in outs[0][2]:
        assert torch.equal(outs[0][2][k], outs[1][2][k]), k
    assert abs(float(outs[0][0]) - float(outs[2][0])) < 1e-5 * abs(float(outs[2][0]))
    for a, b in zip(outs[0][1], outs[2][1]):
        assert rel_err(a, b) < 1e-5 and row_rel_err(a, b) < 1e-4
    assert set(outs[0][2]) == set(outs[2][2])
    for k in outs[0][2]:
        assert rel_err(outs[0][2][k], outs[2][2][k]) < 1e-4, k
    assert float(outs[0][2]["enc_mask_token"].abs().max()) > 0.0


def test_embed_matches_oracle_and_is_deterministic(cuda):
    n_dim, e_dim = 11, 23
    n, e = 900, 2600
    src, dst, ntype, etype = small_graph(n, e, n_dim, e_dim, seed=33, hub_deg=400)
    ref, mine = _model_pair(n_dim, e_dim, 64, 3, cuda, seed=1)
    ref.eval()
    mine.eval()
    og = to_ograph(src, dst, n, ntype, etype, n_dim, e_dim, torch.float64)
    g = to_graph(src, dst, n, ntype, etype, n_dim, e_dim, cuda)
    with torch.no_grad():
        eo = copy.deepcopy(ref).double().embed(og)
        e1 = mine.embed(g)
        e2 = mine.embed(g)
    assert rel_err(e1, eo) < 1e-4
    assert row_rel_err(e1, eo) < 1e-4   # per node, relative to that node's own embedding
    assert torch.equal(e1, e2)   # no atomics anywhere: bitwise reproducible
    # one-hot node types in: the first layer ran on the type ids (no one-hot matrix, no projection launch); the
    # dense projection of the materialised matrix gives the same embeddings to rounding
    from magic_amd import _lib
    _lib.prof_reset()
    _lib.prof_enable(True)
    with torch.no_grad():
        mine.embed(g)
        coded_names = set(_lib.prof_read(160))
        _lib.prof_reset()
        mine.coded_input = False
        e3 = mine.embed(g)
        dense_names = set(_lib.prof_read(160))
    _lib.prof_enable(False)
    _lib.prof_reset()
    mine.coded_input = True
    assert "node_code" in coded_names and "node_code" not in dense_names
    assert rel_err(e1, e3) < 1e-5 and row_rel_err(e1, e3) < 1e-4


def test_batch_level_train_loop(cuda):
    """``batch_level_train`` (model/train.py:18-48 of the reference) on a handful of small graphs:
    the loop runs on batched graphs, moves the parameters and lowers the loss."""
    import random as pyrandom
    from magic_amd import synth
    from magic_amd.model.autoencoder import GMAEModel
    from magic_amd.model.train import batch_level_train
    n_dim, e_dim = 8, 20
    torch.manual_seed(0)
    pyrandom.seed(0)
    graphs = [(synth.make_graph(300 + 20 * i, 12000 + 100 * i, n_dim, e_dim, seed=70 + i, attrs=False), 0)
              for i in range(6)]
    for g, _ in graphs:
        g.to(cuda).structure.ensure_csr()
    graphs = [(g.to(cuda), y) for g, y in graphs]
    model = GMAEModel(n_dim, e_dim, 64, 2, 4, "prelu", 0.1, 0.2, True, "BatchNorm", 0.5, "sce", 3).to(cuda)
    opt = torch.optim.Adam(model.parameters(), lr=5e-3, weight_decay=5e-4)
    loader = [[0, 1, 2], [3, 4, 5]]

    def epoch_loss():
        model.eval()
        tot = 0.0
        gen = torch.Generator().manual_seed(1)
        from magic_amd.utils.loaddata import transform_graph
        import magic_amd
        for b in loader:
            bg = magic_amd.batch([transform_graph(graphs[i][0], n_dim, e_dim) for i in b])
            n, e = bg.num_nodes(), bg.num_edges()
            with torch.enable_grad():
                tot += float(model.compute_loss(bg, perm=torch.randperm(n, generator=gen).to(cuda),
                                                pos_idx=torch.randint(0, e, (n,), generator=gen),
                                                neg_pairs=(torch.randint(0, n, (n,), generator=gen),
                                                           torch.randint(0, n, (n,), generator=gen))))
        return tot

    before = epoch_loss()
    w0 = model.encoder.gats[0].fc.weight.detach().clone()
    batch_level_train(model, graphs, loader, opt, 8, cuda, n_dim, e_dim, progress=False)
    after = epoch_loss()
    assert not torch.equal(w0, model.encoder.gats[0].fc.weight)
    assert after < before, (before, after)


def test_embed_on_edgeless_graph(cuda):
    """No edges at all: every node is 'bias + residual' through the layers (zero in-degree is allowed,
    model/gat.py:180 never checks the flag)."""
    n_dim, e_dim, n = 5, 17, 50
    ref, mine = _model_pair(n_dim, e_dim, 64, 3, cuda, seed=2)
    ref.eval()
    mine.eval()
    src = np.zeros(0, dtype=np.int64)
    rng = np.random.default_rng(0)
    ntype = rng.integers(0, n_dim, n)
    etype = np.zeros(0, dtype=np.int64)
    og = to_ograph(src, src, n, ntype, etype, n_dim, e_dim, torch.float64)
    g = to_graph(src, src, n, ntype, etype, n_dim, e_dim, cuda)
    with torch.no_grad():
        eo = copy.deepcopy(ref).double().embed(og)
        em = mine.embed(g)
    assert em.shape == (n, 64)
    assert rel_err(em, eo) < 1e-5


def test_default_sampling_paths_equal_injected_ones(cuda):
    """compute_loss without injected randomness: (a) the rejected negative candidates carry weight 0
    in the reconstruction loss instead of being filtered out first -- same loss and gradients as
    passing the accepted pairs; (b) the positive edges are random.sample(range(E), T), drawn in
    array form; (c) a batch assembled from marked one-hot attributes takes the type-table path
    without checking the attribute values."""
    import random as pyrandom
    import magic_amd
    from magic_amd import synth
    from magic_amd.model.autoencoder import GMAEModel, global_uniform_negative_sampling
    from magic_amd.utils.loaddata import transform_graph
    n_dim, e_dim = 8, 20
    parts = [synth.make_graph(400 + 30 * i, 30000 + 500 * i, n_dim, e_dim, seed=80 + i, attrs=False).to(cuda)
             for i in range(3)]
    for g in parts:
        g.structure.ensure_csr()
    torch.manual_seed(0)
    model = GMAEModel(n_dim, e_dim, 64, 2, 4, "prelu", 0.1, 0.2, True, "BatchNorm", 0.5, "sce", 3).to(cuda).eval()

    def union():
        bg = magic_amd.batch([transform_graph(g, n_dim, e_dim) for g in parts])
        assert bg._edge_attr_marked() and bg.edge_types_csc() is not None
        return bg
    bg = union()
    n, e = bg.num_nodes(), bg.num_edges()
    thr = min(10000, n)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(5)).to(cuda)

    def grads():
        out = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None])
        model.zero_grad(set_to_none=True)
        return out
    # default path
    torch.manual_seed(11)
    pyrandom.seed(12)
    with torch.enable_grad():
        l_def = model.compute_loss(union(), perm=perm)
    l_def.backward()
    g_def = grads()
    # the same choices, injected
    torch.manual_seed(11)
    neg = global_uniform_negative_sampling(bg, thr)
    assert 0 < neg[0].numel() <= thr
    pyrandom.seed(12)
    pos = pyrandom.sample(range(e), thr)
    with torch.enable_grad():
        l_inj = model.compute_loss(union(), perm=perm, pos_idx=pos, neg_pairs=neg)
    l_inj.backward()
    g_inj = grads()
    assert abs(float(l_def) - float(l_inj)) <= 2e-6 * abs(float(l_inj))
    assert rel_err(g_def, g_inj) < 1e-4
    # a write to the attribute tensor drops the mark
    bg2 = union()
    bg2.edata["attr"][0, 0] += 0.0
    assert not bg2._edge_attr_marked()


def test_custom_operators_called_directly(cuda):
    """torch.ops.magic.gat_fwd_ / knn_topk / sce_loss / linear_nobias called as operators (not through
    the model classes): same results as the oracle, autograd formulas of the two differentiable ones."""
    import magic_amd
    from magic_amd import ops
    H, D, n_dim, e_dim = 4, 16, 5, 17
    n, e = 400, 1700
    src, dst, ntype, etype = small_graph(n, e, n_dim, e_dim, seed=12, hub_deg=120)
    g = to_graph(src, dst, n, ntype, etype, n_dim, e_dim, cuda)
    s = g.structure.ensure_csr()
    torch.manual_seed(4)
    hs = torch.randn(n, H * D, device=cuda)
    eh, et = torch.randn(n, H, device=cuda), torch.randn(n, H, device=cuda)
    table = torch.randn(e_dim, H, device=cuda)
    bias = torch.randn(H * D, device=cuda)
    out = torch.empty(n, H * D, device=cuda)
    pre, attn, stat = torch.ops.magic.gat_fwd_(*ops.plan_tensors(s.plan_in), s.in_src, s.in_dst, H, D, hs, eh, et,
                                               table, g.edge_types_csc(), None, 0.2, bias, None, None, False, None,
                                               1.0, True, True, None, out)
    assert pre.numel() == 0 and attn.shape == (src.size, H) and stat.shape == (n, H, 2)
    # oracle statement of model/gat.py:302-324 on the same operands
    S, T = torch.from_numpy(src), torch.from_numpy(dst)
    x = (eh.cpu().double()[S] + table.cpu().double()[torch.from_numpy(etype)]) + et.cpu().double()[T]
    lx = torch.nn.functional.leaky_relu(x, 0.2)
    a = O.edge_softmax(T, lx.unsqueeze(-1), n)
    want = O.u_mul_e_sum(S, T, hs.cpu().double().view(n, H, D), a, n).reshape(n, H * D) + bias.cpu().double()
    assert rel_err(out, want) < 1e-5
    # attention weights (CSC order, sign bit = logit not positive) and the per-destination statistics
    eid = s.in_eid.long().cpu()
    assert (attn.cpu().abs().double() - a.squeeze(-1)[eid]).abs().max() < 1e-6
    assert torch.equal(attn.cpu() < 0, (x[eid] <= 0) & (attn.cpu() != 0))
    has_in = torch.zeros(n, dtype=torch.bool).index_fill_(0, T, True)
    lse = torch.full((n, H), -float("inf"), dtype=torch.float64)
    lse = lse.scatter_reduce(0, T.view(-1, 1).expand(-1, H), lx, "amax")
    lse = lse + torch.log(torch.zeros(n, H, dtype=torch.float64).index_add_(0, T, torch.exp(lx - lse[T])))
    assert torch.equal(stat[:, :, 0].cpu(), et.cpu())
    assert (stat[:, :, 1].cpu().double() - lse)[has_in].abs().max() < 1e-5
    assert (stat[:, :, 1].cpu()[~has_in] == 0).all()
    # differentiable operators
    xw = torch.randn(300, 24, device=cuda, requires_grad=True)
    w = torch.randn(40, 24, device=cuda, requires_grad=True)
    y = torch.ops.magic.linear_nobias(xw, w)
    y.square().sum().backward()
    xr, wr = xw.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    (xr @ wr.t()).square().sum().backward()
    assert rel_err(y, xr @ wr.t()) < 1e-5 and rel_err(xw.grad, xr.grad) < 1e-4 and rel_err(w.grad, wr.grad) < 1e-4
    xs = torch.randn(200, 7, device=cuda, requires_grad=True)
    tgt = torch.nn.functional.one_hot(torch.randint(0, 7, (200,)), 7).float().to(cuda)
    mask = torch.randperm(200)[:90].to(cuda)
    loss = torch.ops.magic.sce_loss(xs, tgt, mask, 3.0)
    loss.backward()
    xo = xs.detach().cpu().double().requires_grad_(True)
    lo = O.sce_loss(xo[mask.cpu()], tgt.cpu().double()[mask.cpu()], alpha=3)
    lo.backward()
    assert abs(float(loss) - float(lo)) < 1e-5 and rel_err(xs.grad, xo.grad) < 1e-4
    assert magic_amd.__version__


def test_negative_candidates_are_distinct_non_edges_without_self_loops(cuda):
    """The stand-in for dgl.sampling.global_uniform_negative_sampling (model/autoencoder.py:251) on a
    small dense graph, where rejections and repeats are frequent: the accepted pairs are distinct, no
    edges, no self-loops, at most k, and k exactly whenever enough candidates survive."""
    from magic_amd.graph import Graph
    from magic_amd.model.autoencoder import _negative_candidates, negative_candidate_count
    n = 40
    rng = np.random.default_rng(0)
    dense = rng.random((n, n)) < 0.3
    src, dst = np.nonzero(dense)
    g = Graph(torch.from_numpy(src), torch.from_numpy(dst), n).to(cuda)
    edges = set(zip(src.tolist(), dst.tolist()))
    k = 300
    gen = torch.Generator(device=cuda).manual_seed(1)
    u, v, ok = _negative_candidates(g, k, generator=gen)
    assert u.shape == v.shape == ok.shape == (negative_candidate_count(k),)
    pairs = list(zip(u[ok].tolist(), v[ok].tolist()))
    assert 0 < len(pairs) <= k and len(set(pairs)) == len(pairs)
    assert all(a != b and (a, b) not in edges for a, b in pairs)
    # every rejected candidate before the last accepted one is an edge, a self-loop or a repeat
    last = int(torch.nonzero(ok)[-1])
    seen = set()
    for j in range(last + 1):
        p = (int(u[j]), int(v[j]))
        good = p[0] != p[1] and p not in edges and p not in seen
        assert bool(ok[j]) == good
        seen.add(p)
    u2, v2, ok2 = _negative_candidates(g, 5, generator=gen)
    assert int(ok2.sum()) == 5          # plenty of candidates: exactly k are accepted


def test_edge_type_cache_follows_tensor_identity_and_endpoints_are_checked(cuda):
    from magic_amd.graph import Graph
    n_dim, e_dim = 5, 7
    src, dst, nt, et = small_graph(120, 500, n_dim, e_dim, seed=2)
    g = to_graph(src, dst, 120, nt, et, n_dim, e_dim, cuda)
    first = g.edge_types_csc().clone()
    c = g.clone()
    # a new attribute tensor of the same shape (possibly at a recycled address) with other types
    new_types = (g.edata["type"] + 1) % e_dim
    c.edata["type"] = new_types
    c.edata["attr"] = torch.nn.functional.one_hot(new_types, e_dim).float()
    got = c.edge_types_csc()
    assert torch.equal(got, new_types.to(torch.int32)[g.structure.in_eid.long()])
    assert torch.equal(g.edge_types_csc(), first)                    # the original graph keeps its own answer
    moved = g.to("cpu")
    assert not moved._edge_attr_marked()                             # to_graph builds attr without the mark ...
    from magic_amd.utils.loaddata import transform_graph
    assert transform_graph(moved, n_dim, e_dim).to(cuda)._edge_attr_marked()   # ... transform_graph's mark survives .to()
    with pytest.raises(ValueError):
        Graph(torch.tensor([0, 5]), torch.tensor([1, 2]), 5)          # host endpoints: checked at construction
    bad = Graph(torch.tensor([0, 1], device=cuda), torch.tensor([1, -1], device=cuda), 5)
    with pytest.raises(ValueError):
        bad.structure.ensure_csr()                                    # device endpoints: checked before the sort


@pytest.mark.parametrize("hidden,n_pos,n_neg,weighted", [(64, 300, 375, True), (256, 1000, 1250, True),
                                                        (64, 37, 5, False), (128, 1, 0, False)])
def test_edge_reconstruction_loss_tail_equals_tensor_path(cuda, hidden, n_pos, n_neg, weighted):
    """magic_edge_head_fwd / _bwd (LeakyReLU -> Linear(hidden, 1) -> Sigmoid -> BCE in two passes) against the
    same head evaluated with tensor operations in float64: loss, gradients of the gathered rows, of both linear
    layers and their biases; negative pairs weighted 0 / 1, a scaled upstream gradient, and saturated logits
    (the -100 clamp of BCELoss)."""
    from magic_amd.model.autoencoder import GMAEModel
    torch.manual_seed(hidden + n_pos)
    layers = 2
    m = GMAEModel(8, 4, hidden, layers, 4, "prelu", 0.1, 0.2, True, "BatchNorm", 0.5, "sce", 3).to(cuda)
    P = n_pos + n_neg
    rows = torch.randn(P, 2 * hidden * layers).to(cuda).requires_grad_(True)   # the pair matrix [h[src] | h[dst]]
    valid = (torch.rand(n_neg) > 0.3).to(cuda) if weighted else None
    with torch.no_grad():                      # two pairs far into saturation in float32 and float64 alike
        m.edge_recon_fc[0].bias.normal_()      # (zero-initialised by the model; the tail adds it on the fly)
        m.edge_recon_fc[2].bias.fill_(0.3)
        rows[0] *= 1e4
        rows[P - 1] *= -1e4
    loss = m._edge_recon_loss(rows, n_pos, n_neg, valid)
    assert loss is not None and loss.shape == ()
    (loss * 0.25).backward()
    got = dict(rows=rows.grad.clone(), **{n: p.grad.clone() for n, p in m.edge_recon_fc.named_parameters()})
    # float64 evaluation of the reference's statements (model/autoencoder.py:257-266)
    ref = copy.deepcopy(m.edge_recon_fc).double().cpu()
    r64 = rows.detach().double().cpu().requires_grad_(True)
    y_pred = ref(r64).squeeze(-1)
    y = torch.cat([torch.ones(n_pos), torch.zeros(n_neg)]).double()
    if weighted:
        w = torch.cat([torch.ones(n_pos), valid.cpu().double()])
        want = torch.nn.functional.binary_cross_entropy(y_pred, y, weight=w, reduction="sum") / w.sum()
    else:
        want = torch.nn.BCELoss()(y_pred, y)
    (want * 0.25).backward()
    assert abs(float(loss) - float(want)) <= 1e-5 * max(1.0, abs(float(want)))
    assert rel_err(got["rows"].cpu().double(), r64.grad) < 5e-5
    for n, p in ref.named_parameters():
        assert rel_err(got[n].cpu().double(), p.grad) < 5e-5, n


def test_fold_of_all_layers_in_one_launch_equals_the_per_layer_calls(cuda):
    """magic_fold_fwd_multi / _bwd_multi against magic_fold_fwd / _bwd layer by layer, bit for bit: layers with and
    without residual projection and bias, padded head widths, more layers than one launch holds (10 > 8), absent
    incoming gradients."""
    from magic_amd import _lib
    gen = torch.Generator().manual_seed(3)
    r = lambda *s: torch.randn(*s, generator=gen).to(cuda)   # noqa: E731
    specs = [(4, 16, 16, 8, 5, True, True), (4, 16, 16, 64, 5, False, True), (1, 5, 8, 64, 5, False, True),
             (4, 64, 64, 256, 20, False, False), (2, 3, 4, 7, 3, True, False)] * 2
    layers = []
    for H, D, Dp, F, e_dim, has_res, has_bias in specs:
        layers.append((H, D, Dp, r(H * D, F), r(1, H, D), r(1, H, D), r(H * D, F) if has_res else None,
                       r(H * D, e_dim), r(1, H, D), r(1, H, D) if has_bias else None))
    multi = _lib.fold_fwd_multi(layers)
    grads = []
    for i, (lay, out) in enumerate(zip(layers, multi)):
        H, D, Dp, w_fc, ah, at, w_res, w_efc, ae, bias = lay
        single = _lib.fold_fwd(H, D, Dp, w_fc, ah, at, w_res, w_efc, ae, bias)
        for a, b in zip(out, single):
            assert (a is None) == (b is None) and (a is None or torch.equal(a, b)), i
        g = [r(*out[0].shape), r(*out[1].shape), r(*out[2].shape) if out[2] is not None else None]
        if i == 3:
            g[1] = None          # no gradient reached the edge-score table
        if i == 6:
            g[0] = None          # ... or the augmented weight
        grads.append(tuple(g))
    back = _lib.fold_bwd_multi(layers, grads)
    for i, (lay, g, out) in enumerate(zip(layers, grads, back)):
        H, D, Dp, w_fc, ah, at, w_res, w_efc, ae, bias = lay
        single = _lib.fold_bwd(H, D, Dp, w_fc, ah, at, w_efc, ae, g[0], g[1], g[2], w_res is not None, bias is not None)
        assert len(out) == len(single) == 7
        for a, b in zip(out, single):
            assert (a is None) == (b is None) and (a is None or torch.equal(a, b)), i


@pytest.mark.parametrize("H,D", [(1, 8), (1, 16), (4, 16), (4, 64)])
@pytest.mark.parametrize("hub_on", [False, True])
def test_sweeps_restricted_to_marked_destinations(H, D, hub_on, cuda):
    """magic_gat_fwd_active / magic_gat_bwd_active (the decoder, whose output is read at the masked rows only,
    model/autoencoder.py:240-245): marked rows bit for bit those of the full sweeps, unmarked rows = bias with a
    +inf log-sum-exp; with a gradient that is zero outside the marked rows every backward output equals the
    full backward's."""
    from magic_amd import ops
    n, e, n_dim, e_dim = 3000, 30000, 5, 17
    src, dst, ntype, etype = small_graph(n, e, n_dim, e_dim, seed=21, hub_deg=700, hub_out=400)
    g = to_graph(src, dst, n, ntype, etype, n_dim, e_dim, cuda)
    s = g.structure.ensure_csr()
    assert s.plan_in.n_chunks > 3 and s.plan_out.n_chunks > 0
    torch.manual_seed(8)
    C = H * D
    hs = torch.randn(n, C, device=cuda)
    eh, et = torch.randn(n, H, device=cuda), torch.randn(n, H, device=cuda)
    table = torch.randn(e_dim, H, device=cuda)
    bias = torch.randn(C, device=cuda)
    flags = (torch.rand(n) < 0.5).to(torch.uint8)
    flags[3] = int(hub_on)          # the hub of small_graph
    flags[:2] = 1                   # (the node the unmarked edges are pointed at is not node 0)
    flags = flags.to(cuda)
    on = flags.bool()

    def fwd(active):
        return ops.gat_fwd(s, H, D, hs, eh, et, table, g.edge_types_csc(), None, 0.2, bias, None, None, False,
                           want_attn=False, want_stat=True, dst_active=active)
    out_all, _, _, stat_all = fwd(None)
    out_on, _, _, stat_on = fwd(flags)
    assert torch.equal(out_on[on], out_all[on]) and torch.equal(stat_on[on], stat_all[on])
    assert torch.equal(out_on[~on], bias.expand(int((~on).sum()), C))
    assert torch.isinf(stat_on[~on][:, :, 1]).all() and (stat_on[~on][:, :, 1] > 0).all()

    d_out = torch.randn(n, C, device=cuda) * on[:, None]

    def bwd(stat, active):
        d = d_out.clone()
        d_y = torch.full((n, (C + 2 * H + 3) // 4 * 4), float("nan"), device=cuda)[:, :C + 2 * H]   # float4 rows
        d_table, _, d_bias, _ = ops.gat_bwd(s, H, D, hs, eh, stat, table, g.edge_types_csr(), None, e_dim, 0.2, None,
                                            None, None, d, d, d_y, True, dst_active=active)
        return d_y.contiguous(), d_table, d_bias
    ref = bwd(stat_all, None)
    for stat in (stat_on, stat_all):     # (the statistics of either forward serve)
        got = bwd(stat, flags)
        for a, b in zip(got, ref):
            assert torch.equal(a, b)
    # all marked / none marked
    ones = torch.ones(n, dtype=torch.uint8, device=cuda)
    assert torch.equal(fwd(ones)[0], out_all)
    for a, b in zip(bwd(stat_all, ones), ref):
        assert torch.equal(a, b)
    zeros = torch.zeros(n, dtype=torch.uint8, device=cuda)
    assert torch.equal(fwd(zeros)[0], bias.expand(n, C))
    with pytest.raises(Exception):
        ops.gat_fwd(s, H, D, hs, eh, et, table, g.edge_types_csc(), None, 0.2, bias, None, None, False,
                    want_attn=True, dst_active=flags)


@pytest.mark.parametrize("shape", [(5, 17, 64, 3), (8, 20, 256, 2)])
def test_decoder_restricted_to_the_masked_rows_changes_nothing(cuda, shape):
    """compute_loss reads recon[mask_nodes] only (model/autoencoder.py:242-245): with the decoder's sweeps restricted
    to those rows the loss and every parameter gradient are bit for bit those of the full decoder."""
    from magic_amd import _lib
    n_dim, e_dim, hidden, layers = shape
    n, e = 1500, 12000
    src, dst, ntype, etype = small_graph(n, e, n_dim, e_dim, seed=41, hub_deg=500, hub_out=300)
    _, mine = _model_pair(n_dim, e_dim, hidden, layers, cuda)
    mine.train()
    gen = torch.Generator().manual_seed(3)
    widths = [n_dim] + [hidden] * (layers - 1) + [hidden]
    drop = [(torch.rand(n, w, generator=gen) < 0.9).to(cuda) for w in widths]
    perm = torch.randperm(n, generator=gen).to(cuda)
    pos_idx = torch.randperm(src.size, generator=gen)[:n]
    neg = (torch.randint(0, n, (n,), generator=gen), torch.randint(0, n, (n,), generator=gen))
    outs = []
    mine.decoder_rows_min_nodes = 0
    for rows in (True, False):
        mine.decoder_rows = rows
        g = to_graph(src, dst, n, ntype, etype, n_dim, e_dim, cuda)
        mine.zero_grad(set_to_none=True)
        _lib.prof_reset()
        _lib.prof_enable(True)
        loss = mine.compute_loss(g, perm=perm, pos_idx=pos_idx, neg_pairs=neg, drop_masks=drop)
        loss.backward()
        _lib.prof_enable(False)
        names = set(_lib.prof_read(200))
        _lib.prof_reset()
        assert ("gat_bwd_active_dst" in names) == rows, names
        outs.append((loss.detach().clone(), {k: p.grad.clone() for k, p in mine.named_parameters() if p.grad is not None}))
    mine.decoder_rows = True
    assert torch.equal(outs[0][0], outs[1][0])
    assert set(outs[0][1]) == set(outs[1][1])
    for k in outs[0][1]:
        assert torch.equal(outs[0][1][k], outs[1][1][k]), k


def test_onehot_node_mark_dies_with_writes(cuda):
    """The coded first layer trusts ``ndata['attr'] == one_hot(ndata['type'])`` only while both tensors are the very
    objects that were marked, unwritten; clones and device copies keep the mark."""
    n, e, n_dim, e_dim = 200, 600, 5, 7
    src, dst, ntype, etype = small_graph(n, e, n_dim, e_dim, seed=2)
    g = to_graph(src, dst, n, ntype, etype, n_dim, e_dim, cuda)
    assert torch.equal(g.node_types_if_onehot().cpu(), torch.from_numpy(ntype))
    assert g.clone().node_types_if_onehot() is not None
    g1 = to_graph(src, dst, n, ntype, etype, n_dim, e_dim, cuda)
    g1.ndata["attr"][0, 0] += 0.0
    assert g1.node_types_if_onehot() is None
    g2 = to_graph(src, dst, n, ntype, etype, n_dim, e_dim, cuda)
    g2.ndata["type"][0] = (g2.ndata["type"][0] + 1) % n_dim
    assert g2.node_types_if_onehot() is None
    g3 = to_graph(src, dst, n, ntype, etype, n_dim, e_dim, cuda)
    g3.ndata["type"] = g3.ndata["type"].clone()
    assert g3.node_types_if_onehot() is None
