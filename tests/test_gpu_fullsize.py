"""Size-independent properties at the benchmark's full sizes (THEIA-shaped graph: 700,000 nodes,
3,000,000 edges; Trace-shaped KNN: 2,604,530 train rows), where the CPU oracle would take minutes:
softmax normalisation, the zero-in-degree closed form, bitwise reproducibility, a directional
derivative of the loss, sortedness / self-match / shard-merge invariance of the neighbour search,
and the CSR invariants."""
import numpy as np
import pytest
import torch

from magic_amd import synth

pytestmark = pytest.mark.gpu

N, E, N_DIM, E_DIM, HIDDEN, LAYERS = synth.SHAPES["theia"]


@pytest.fixture(scope="module")
def theia(cuda):
    g = synth.make_graph(N, E, N_DIM, E_DIM, seed=1004, attrs=True).to(cuda)
    g.structure.ensure_csr()
    return g


def _model(cuda, seed=0):
    from magic_amd.model.autoencoder import GMAEModel
    torch.manual_seed(seed)
    m = GMAEModel(N_DIM, E_DIM, HIDDEN, LAYERS, 4, "prelu", 0.1, 0.2, True, "BatchNorm", 0.5, "sce", 3).to(cuda)
    with torch.no_grad():
        m.enc_mask_token.normal_(0, 0.3)
    return m


def test_csr_invariants_full_size(theia):
    s = theia.structure
    src, dst = theia.edges()
    n, e = theia.num_nodes(), theia.num_edges()
    for ptr, ids, eid, key in ((s.in_ptr, s.in_src, s.in_eid, dst), (s.out_ptr, s.out_dst, s.out_eid, src)):
        assert int(ptr[0]) == 0 and int(ptr[-1]) == e and bool((ptr[1:] >= ptr[:-1]).all())
        eid = eid.long()
        assert torch.equal(torch.sort(eid).values, torch.arange(e, device=eid.device))   # a permutation
        k = key[eid]
        assert bool((k[1:] >= k[:-1]).all())                                             # sorted by key
        same = k[1:] == k[:-1]
        assert bool((eid[1:][same] > eid[:-1][same]).all())                              # stable
        assert torch.equal(torch.bincount(key, minlength=n), (ptr[1:] - ptr[:-1]).long())
    assert torch.equal(s.in_src.long(), src[s.in_eid.long()])
    assert torch.equal(s.in_eid.long()[s.out_pos.long()], s.out_eid.long())              # CSR slot -> CSC slot
    assert torch.equal(s.out_eid.long()[s.in_pos.long()], s.in_eid.long())               # CSC slot -> CSR slot


def test_attention_rows_sum_to_one_and_isolated_nodes(theia, cuda):
    model = _model(cuda).eval()
    layer = model.encoder.gats[0]
    x = theia.ndata["attr"]
    with torch.no_grad():
        out, attn = layer(theia, x, get_attention=True)
    assert attn.shape == (E, 4, 1) and bool((attn >= 0).all())
    _, dst = theia.edges()
    sums = torch.zeros(N, 4, device=cuda, dtype=torch.float64).index_add_(0, dst, attn[:, :, 0].double())
    deg = torch.bincount(dst, minlength=N)
    assert float((sums[deg > 0] - 1).abs().max()) < 1e-5
    # no in-edges: out = PReLU(bias + res_fc(x)) exactly as model/gat.py:317-335 leaves it
    iso = (deg == 0).nonzero().flatten()[:50_000]
    ref = torch.nn.functional.prelu(layer.bias.view(1, -1) + x[iso] @ layer.res_fc.weight.t(), layer.activation.weight)
    assert float((out[iso] - ref).detach().abs().max()) < 1e-5


def test_training_step_bitwise_reproducible_and_directional_derivative(theia, cuda):
    model = _model(cuda).train()
    gen = torch.Generator().manual_seed(3)
    perm = torch.randperm(N, generator=gen).to(cuda)
    pos = torch.randint(0, E, (10000,), generator=gen)
    neg = (torch.randint(0, N, (10000,), generator=gen), torch.randint(0, N, (10000,), generator=gen))
    widths = [N_DIM] + [HIDDEN] * (LAYERS - 1) + [HIDDEN]
    masks = [(torch.rand(N, w, device=cuda) < 0.9) for w in widths]

    def run(mdl):
        theia.ndata["attr"] = torch.nn.functional.one_hot(theia.ndata["type"], N_DIM).float()
        mdl.zero_grad(set_to_none=True)
        loss = mdl.compute_loss(theia, perm=perm, pos_idx=pos, neg_pairs=neg, drop_masks=masks)
        loss.backward()
        return loss.detach().clone(), {k: p.grad.clone() for k, p in mdl.named_parameters() if p.grad is not None}

    l1, g1 = run(model)
    l2, g2 = run(model)
    assert torch.equal(l1, l2)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k          # fixed-order reductions everywhere, no atomics
        assert bool(torch.isfinite(g1[k]).all()), k
    # central difference of the loss along the normalised gradient direction of a few tensors
    names = ["encoder.gats.1.fc.weight", "encoder.gats.0.attn_h", "decoder.gats.0.fc.weight", "encoder_to_decoder.weight"]
    params = dict(model.named_parameters())
    eps = 2e-2
    norm = sum(float((g1[k].double() ** 2).sum()) for k in names) ** 0.5

    def shift(scale):
        with torch.no_grad():
            for k in names:
                params[k].add_(g1[k] * (scale * eps / norm))

    shift(1)
    lp = float(run(model)[0])
    shift(-2)
    lm = float(run(model)[0])
    shift(1)
    fd = (lp - lm) / (2 * eps)
    assert abs(fd - norm) / norm < 0.05, (fd, norm)   # fp32 loss of O(1): the difference carries ~1e-5 noise


def test_knn_properties_full_size(cuda):
    from magic_amd.knn import NearestNeighbors, merge_topk
    nt, nq, d, k = 2_604_530, 16_384, 64, 10
    gen = torch.Generator(device=cuda).manual_seed(5)
    centers = torch.randn(200, d, device=cuda, generator=gen)
    train = centers[torch.randint(0, 200, (nt,), device=cuda, generator=gen)] + \
        0.05 * torch.randn(nt, d, device=cuda, generator=gen)
    rows = torch.randint(0, nt, (nq,), device=cuda, generator=gen)
    query = train[rows].clone()
    nn_full = NearestNeighbors(k, device=cuda).fit(train)
    dist, idx = nn_full.kneighbors_device(query, k)
    assert dist.dtype == torch.float64 and idx.dtype == torch.int64
    assert bool((dist[:, 1:] >= dist[:, :-1]).all())                    # ascending
    assert torch.equal(idx[:, 0], rows)                                 # a train row finds itself first
    assert float(dist[:, 0].max()) < 1e-3
    # distances are what the indices say (float64 recomputation of the returned pairs)
    chk = (query.double().unsqueeze(1) - train[idx].double()).pow(2).sum(-1).sqrt()
    assert float((chk - dist).abs().max()) < 1e-5
    # nothing closer was missed: a random sample of train rows is never nearer than the k-th neighbour
    samp = train[torch.randint(0, nt, (4096,), device=cuda, generator=gen)].double()
    dmin = torch.cdist(query[:512].double(), samp)
    assert bool((dmin >= dist[:512, :1] - 1e-6).all())     # (the sample may hold the query's own row)
    assert bool(((dmin < dist[:512, -1:] - 1e-9).sum(1) <= k).all())
    # two half shards merged in shard order give the single-search answer bit for bit
    half = nt // 2
    parts = []
    for lo, hi in ((0, half), (half, nt)):
        dd, ii = NearestNeighbors(k, device=cuda).fit(train[lo:hi]).kneighbors_device(query, k, index_offset=lo)
        parts.append((dd, ii))
    md, mi = merge_topk(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    assert torch.equal(mi, idx) and torch.equal(md, dist)


def test_knn_long_lists_properties_at_cadets_size(cuda):
    """k = 200 (model/eval.py:289) at the size of a cadets evaluation: 350,000 train rows, 32,768 queries.  Ascending
    lists, a train row finds itself, the distances are what the indices say, a float64 brute force of a query sample
    gives the same lists, and two half shards merge into the single-search answer bit for bit."""
    from magic_amd.knn import NearestNeighbors, merge_topk
    nt, nq, d, k = 350_000, 32_768, 64, 200
    gen = torch.Generator(device=cuda).manual_seed(6)
    centers = torch.randn(200, d, device=cuda, generator=gen)
    train = centers[torch.randint(0, 200, (nt,), device=cuda, generator=gen)] + \
        0.05 * torch.randn(nt, d, device=cuda, generator=gen)
    rows = torch.randint(0, nt, (nq,), device=cuda, generator=gen)
    query = train[rows].clone()
    dist, idx = NearestNeighbors(k, device=cuda).fit(train).kneighbors_device(query, k)
    assert bool((dist[:, 1:] >= dist[:, :-1]).all())
    assert torch.equal(idx[:, 0], rows) and float(dist[:, 0].max()) < 1e-3
    assert bool((idx.sort(1).values[:, 1:] != idx.sort(1).values[:, :-1]).all())      # no row twice
    sel = torch.arange(0, nq, 257, device=cuda)
    chk = (query[sel].double().unsqueeze(1) - train[idx[sel]].double()).pow(2).sum(-1).sqrt()
    assert float((chk - dist[sel]).abs().max()) < 1e-5
    full = torch.cdist(query[sel].double(), train.double())                          # [128, 350000] float64
    want_d, want_i = full.topk(k, dim=1, largest=False)
    assert float((want_d - dist[sel]).abs().max()) < 1e-6
    assert float((want_i == idx[sel]).double().mean()) > 0.999      # (float64 near-ties of the two formulas may swap)
    half = nt // 2
    parts = []
    for lo, hi in ((0, half), (half, nt)):
        parts.append(NearestNeighbors(k, device=cuda).fit(train[lo:hi]).kneighbors_device(query, k, index_offset=lo))
    md, mi = merge_topk(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    assert torch.equal(mi, idx) and torch.equal(md, dist)


def test_streamspot_batch_step_reproducible_and_directional_derivative(cuda):
    """The benchmark's default workload (a batch of 12 StreamSpot-sized graphs, hidden 256, 4 layers:
    the wide-layer kernels): union CSR equal to a direct sort, bitwise reproducible step, and the
    loss changing along the gradient as the gradient says."""
    import magic_amd
    from magic_amd.graph import Structure
    from magic_amd.model.autoencoder import GMAEModel
    n1, e1, n_dim, e_dim, hidden, layers = synth.SHAPES["streamspot"]
    parts = [synth.make_graph(n1, e1, n_dim, e_dim, seed=1002 + i, attrs=False).to(cuda) for i in range(12)]
    for p in parts:
        p.structure.ensure_csr()
    g = magic_amd.batch(parts)
    n, e = g.num_nodes(), g.num_edges()
    assert (n, e) == (12 * n1, sum(p.num_edges() for p in parts))
    direct = Structure(g.edges()[0], g.edges()[1], n).ensure_csr()
    for name in ("in_ptr", "in_src", "in_eid", "out_ptr", "out_dst", "out_eid", "out_pos", "in_pos"):
        assert torch.equal(getattr(g.structure, name), getattr(direct, name)), name
    g.edata["attr"] = torch.nn.functional.one_hot(g.edata["type"], e_dim).float()
    torch.manual_seed(0)
    model = GMAEModel(n_dim, e_dim, hidden, layers, 4, "prelu", 0.1, 0.2, True, "BatchNorm", 0.5, "sce", 3).to(cuda)
    with torch.no_grad():
        model.enc_mask_token.normal_(0, 0.3)
    model.train()
    gen = torch.Generator().manual_seed(3)
    perm = torch.randperm(n, generator=gen).to(cuda)
    pos = torch.randint(0, e, (10000,), generator=gen)
    neg = (torch.randint(0, n, (10000,), generator=gen), torch.randint(0, n, (10000,), generator=gen))
    widths = [n_dim] + [hidden] * (layers - 1) + [hidden]
    masks = [(torch.rand(n, w, device=cuda) < 0.9) for w in widths]

    def run():
        g.ndata["attr"] = torch.nn.functional.one_hot(g.ndata["type"], n_dim).float()
        model.zero_grad(set_to_none=True)
        loss = model.compute_loss(g, perm=perm, pos_idx=pos, neg_pairs=neg, drop_masks=masks)
        loss.backward()
        return loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

    l1, g1 = run()
    l2, g2 = run()
    assert torch.equal(l1, l2)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k
        assert bool(torch.isfinite(g1[k]).all()), k
    names = ["encoder.gats.2.fc.weight", "encoder.gats.0.attn_h", "decoder.gats.0.fc.weight",
             "encoder_to_decoder.weight", "edge_recon_fc.0.weight"]
    params = dict(model.named_parameters())
    # central difference along each tensor's own gradient direction, step sized for a predicted loss
    # change of 0.01 per side (large enough against the ~1e-5 noise of an fp32 loss of O(1), small
    # enough for the third-order term)
    for k in names:
        norm = float((g1[k].double() ** 2).sum()) ** 0.5
        eps = 0.01 / norm

        def shift(scale):
            with torch.no_grad():
                params[k].add_(g1[k] * (scale * eps / norm))
        shift(1)
        lp = float(run()[0])
        shift(-2)
        lm = float(run()[0])
        shift(1)
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - norm) / norm < 0.05, (k, fd, norm)
