"""CSR build (bit-exact), KNN search (bit-exact indices on tie-free data) and the SCE kernel
against their oracles, through the C ABI."""
import numpy as np
import pytest
import torch

from oracle import csr_oracle, knn_oracle
from oracle import gat_oracle as O
from tests.util import rel_err, small_graph

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,e,hub", [(1, 0, 0), (50, 0, 0), (300, 1500, 0), (5000, 40000, 3000), (70000, 300000, 0)])
def test_csr_build_bit_exact(cuda, n, e, hub):
    from magic_amd import _lib
    from magic_amd.graph import Structure, LIGHT_MAX, CHUNK
    if e:
        src, dst, _, _ = small_graph(n, e, 4, 4, seed=n, hub_deg=hub, hub_out=hub // 2)
    else:
        src = dst = np.zeros(0, np.int64)
    s = Structure(torch.from_numpy(src).to(cuda), torch.from_numpy(dst).to(cuda), n).ensure_csr()
    want = csr_oracle.build(src, dst, n)
    got = (s.in_ptr, s.in_src, s.in_eid, s.out_ptr, s.out_dst, s.out_eid)
    for w, g_ in zip(want, got):
        assert g_.dtype == torch.int32
        assert np.array_equal(w, g_.cpu().numpy())
    # edge position map and the chunk plan
    inv = np.empty(src.size, np.int32)
    inv[want[2]] = np.arange(src.size, dtype=np.int32)
    assert np.array_equal(inv[want[5]], s.out_pos.cpu().numpy())
    for ptr, plan in ((want[0], s.plan_in), (want[3], s.plan_out)):
        hv, cp, cn, cb, ce = csr_oracle.segment_plan(ptr, LIGHT_MAX, CHUNK)
        assert np.array_equal(hv, plan.heavy_node.cpu().numpy())
        assert np.array_equal(cp, plan.heavy_cptr.cpu().numpy())
        assert np.array_equal(cn, plan.chunk_node.cpu().numpy())
        assert np.array_equal(cb, plan.chunk_beg.cpu().numpy())
        assert np.array_equal(ce, plan.chunk_end.cpu().numpy())


def test_batch_csr_equals_direct_sort(cuda):
    from magic_amd.graph import Graph, batch, Structure
    gs = []
    for i in range(5):
        src, dst, nt, et = small_graph(200 + 37 * i, 900 + 100 * i, 4, 4, seed=100 + i, hub_deg=150, hub_out=100)
        g = Graph(torch.from_numpy(src), torch.from_numpy(dst), 200 + 37 * i).to(cuda)
        g.structure.ensure_csr()
        gs.append(g)
    bg = batch(gs)
    direct = Structure(bg.edges()[0], bg.edges()[1], bg.num_nodes()).ensure_csr()
    for name in ("in_ptr", "in_src", "in_eid", "in_dst", "out_ptr", "out_dst", "out_eid", "out_pos", "in_pos"):
        assert torch.equal(getattr(bg.structure, name), getattr(direct, name)), name
    # the union's segment plans, assembled from the components' plans, equal the ones built from scratch
    for which in ("plan_in", "plan_out"):
        a, b = getattr(bg.structure, which), getattr(direct, which)
        assert (a.n_heavy, a.n_chunks) == (b.n_heavy, b.n_chunks) and a.n_heavy > 0
        for name in ("ptr", "heavy_node", "heavy_cptr", "chunk_node", "chunk_beg", "chunk_end", "order", "xcd_start"):
            ta, tb = getattr(a, name), getattr(b, name)
            assert ta.dtype == tb.dtype and torch.equal(ta, tb), (which, name)


def test_batch_by_the_library_equals_the_tensor_path(cuda, monkeypatch):
    """The union graph assembled by the library (job-table concatenation, device walk order) against the
    same union assembled with torch tensors: every array of the structure and of both segment plans, for
    components with hubs, without edges, with a single node, and for a union without any heavy segment."""
    import importlib
    G = importlib.import_module("magic_amd.graph")
    from magic_amd.graph import Graph, batch

    def graphs(spec):
        gs = []
        for i, (n, e, hub_in, hub_out) in enumerate(spec):
            if e:
                src, dst, _, _ = small_graph(n, e, 4, 4, seed=300 + i, hub_deg=hub_in, hub_out=hub_out)
                g = Graph(torch.from_numpy(src), torch.from_numpy(dst), n).to(cuda)
            else:
                g = Graph(torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int64), n).to(cuda)
            g.structure.ensure_csr()
            gs.append(g)
        return gs

    for spec in ([(300, 1500, 200, 120), (50, 0, 0, 0), (1, 0, 0, 0), (4000, 30000, 900, 700), (64, 64, 0, 0)],
                 [(100, 150, 0, 0), (80, 100, 0, 0)],
                 [(2000, 9000, 300, 0)]):
        gs = graphs(spec)
        calls = []
        real = G._union_structure_device

        def spy(*a, **k):
            r = real(*a, **k)
            calls.append(r is not None)
            return r
        monkeypatch.setattr(G, "_union_structure_device", spy)
        lib = batch(gs).structure
        assert calls == [True]
        monkeypatch.setattr(G, "_union_structure_device", lambda *a, **k: None)
        ref = batch(gs).structure
        monkeypatch.setattr(G, "_union_structure_device", real)
        assert lib.num_nodes == ref.num_nodes and lib.num_edges == ref.num_edges
        for name in ("src", "dst", "in_ptr", "in_src", "in_eid", "in_dst", "out_ptr", "out_dst", "out_eid", "out_pos",
                     "in_pos"):
            ta, tb = getattr(lib, name), getattr(ref, name)
            assert ta.dtype == tb.dtype and torch.equal(ta, tb), name
        for which in ("plan_in", "plan_out"):
            a, b = getattr(lib, which), getattr(ref, which)
            assert (a.n_heavy, a.n_chunks) == (b.n_heavy, b.n_chunks)
            for name in ("ptr", "heavy_node", "heavy_cptr", "chunk_node", "chunk_beg", "chunk_end", "order", "xcd_start"):
                ta, tb = getattr(a, name), getattr(b, name)
                assert ta.dtype == tb.dtype and torch.equal(ta, tb), (which, name)


def test_negative_pair_filter_equals_the_tensor_path(cuda):
    """magic_negative_filter (segment scans + LDS table, one launch) against the same acceptance rule written
    with tensor operations (sorted edge keys, search, stable sort of the candidates), bit for bit: candidates
    that are edges (of hubs in both directions too), self-loops, repeats of accepted and of rejected pairs, a
    quota smaller and larger than the number of passing pairs, an edgeless graph, and a python-loop check of
    the rule itself on the small case."""
    from magic_amd import _lib
    from magic_amd.graph import Graph
    from magic_amd.model.autoencoder import _accept_negatives, _negative_candidates, negative_candidate_count

    gen = torch.Generator().manual_seed(11)
    for n, e, hub_in, hub_out, m, quota in [(60, 300, 40, 30, 500, 100), (5000, 60000, 3000, 2500, 12500, 10000),
                                            (5000, 60000, 3000, 2500, 16384, 20000), (300, 0, 0, 0, 375, 300),
                                            (1, 0, 0, 0, 2, 1)]:
        if e:
            src, dst, _, _ = small_graph(n, e, 4, 4, seed=n + e, hub_deg=hub_in, hub_out=hub_out)
            src, dst = torch.from_numpy(src), torch.from_numpy(dst)
        else:
            src = dst = torch.zeros(0, dtype=torch.int64)
        g = Graph(src, dst, n).to(cuda)
        s = g.structure.ensure_csr()
        u = torch.randint(0, n, (m,), generator=gen)
        v = torch.randint(0, n, (m,), generator=gen)
        if e:
            take = torch.randint(0, e, (m // 5,), generator=gen)   # a fifth of the candidates are edges,
            u[:m // 5], v[:m // 5] = src[take], dst[take]
            mix = torch.randperm(m, generator=gen)                  # spread over the draw order
            u, v = u[mix], v[mix]
        u[3 % m] = v[3 % m]                                         # a self-loop
        if m > 20:
            u[m - 5:], v[m - 5:] = u[5:10].clone(), v[5:10].clone()  # repeats of earlier candidates
        u, v = u.to(cuda), v.to(cuda)
        got = _lib.negative_filter(s.out_ptr, s.out_dst, s.in_ptr, s.in_src, u, v, quota)
        want = _accept_negatives(g, u, v, quota)
        assert got.dtype == torch.bool and torch.equal(got, want), (n, e, m)
        assert int(got.sum()) <= quota
        if n == 60:
            edges = set(zip(src.tolist(), dst.tolist()))
            seen, acc, ref = set(), 0, []
            for a, b in zip(u.tolist(), v.tolist()):
                good = a != b and (a, b) not in edges and (a, b) not in seen and acc < quota
                seen.add((a, b))
                acc += good
                ref.append(good)
            assert got.tolist() == ref
    # the model's own draw goes through the library on the GPU
    nu, nv, ok = _negative_candidates(g, 1)
    assert nu.numel() == negative_candidate_count(1) and ok.dtype == torch.bool and not bool(ok.any())  # one node
    with pytest.raises(_lib.MagicHipError):
        big = torch.zeros(_lib.NEGATIVE_FILTER_MAX + 1, dtype=torch.int64, device=cuda)
        _lib.negative_filter(s.out_ptr, s.out_dst, s.in_ptr, s.in_src, big, big, 1)


@pytest.mark.parametrize("nt,nq,d,k", [(5000, 777, 64, 10), (300, 130, 64, 10), (20000, 300, 64, 200),
                                        (400, 200, 256, 8), (100, 50, 2048, 2), (3000, 257, 32, 5)])
def test_knn_matches_exact_oracle(cuda, nt, nq, d, k):
    from magic_amd import _lib
    rng = np.random.default_rng(nt + d)
    train = rng.standard_normal((nt, d)).astype(np.float32)
    query = rng.standard_normal((nq, d)).astype(np.float32)
    query[: min(20, nq)] = train[: min(20, nq)]   # exact hits: distance 0 must come first
    dist, idx = _lib.knn_topk(torch.from_numpy(train).to(cuda), torch.from_numpy(query).to(cuda), k)
    wd, wi = knn_oracle.exact_topk(train, query, k)
    assert np.array_equal(idx.cpu().numpy(), wi)            # tie-free data: bit-exact indices
    assert np.abs(dist.cpu().numpy() - wd).max() < 1e-6     # |q|^2-2qt+|t|^2 vs (q-t)^2 in f64


@pytest.mark.parametrize("k", [40, 100, 200, 250, 496, 600])
@pytest.mark.parametrize("kind", ["random", "duplicates"])
def test_knn_long_lists(cuda, k, kind):
    """Lists too long for LDS (the cadets evaluation searches k = 200, model/eval.py:289): candidates are appended to
    per-query buffers and sorted by their wave when a buffer fills.  Exact float64 answer, ties by row index, for
    every buffer size the kernel is compiled for (and the sorted global lists beyond them); few queries against
    many rows, so the train rows are cut into several splits that are merged."""
    from magic_amd import _lib
    rng = np.random.default_rng(k)
    nt, nq, d = 30000, 333, 64
    if kind == "random":
        train = rng.standard_normal((nt, d)).astype(np.float32)
        query = rng.standard_normal((nq, d)).astype(np.float32)
    else:
        base = rng.standard_normal((4000, d)).astype(np.float32)
        train = base[rng.integers(0, 4000, nt)]                     # every row about 7 times: ties by index
        query = base[rng.integers(0, 4000, nq)] + 0.01
    dist, idx = _lib.knn_topk(torch.from_numpy(train).to(cuda), torch.from_numpy(query).to(cuda), k)
    wd, wi = knn_oracle.exact_topk(train, query, k, block=200)
    assert np.array_equal(idx.cpu().numpy(), wi)
    assert np.abs(dist.cpu().numpy() - wd).max() < 1e-6
    # descending train order: every row is a candidate when it arrives (the buffers fill at the highest rate)
    order = np.argsort(-((train.astype(np.float64) - query[0].astype(np.float64)) ** 2).sum(1), kind="stable")
    d2, i2 = _lib.knn_topk(torch.from_numpy(train[order]).to(cuda), torch.from_numpy(query[:1]).to(cuda), k)
    w2d, w2i = knn_oracle.exact_topk(train[order], query[:1], k, block=200)
    assert np.array_equal(i2.cpu().numpy(), w2i) and np.abs(d2.cpu().numpy() - w2d).max() < 1e-6


def test_knn_ties_ordered_by_index_and_merge(cuda):
    from magic_amd import _lib
    rng = np.random.default_rng(1)
    base = rng.standard_normal((50, 64)).astype(np.float32)
    train = np.concatenate([base, base, base])               # every row three times
    query = base[:17] + 0.01
    dist, idx = _lib.knn_topk(torch.from_numpy(train).to(cuda), torch.from_numpy(query).to(cuda), 6)
    wd, wi = knn_oracle.exact_topk(train, query, 6)
    assert np.array_equal(idx.cpu().numpy(), wi)
    # shards + merge == single search
    parts_d, parts_i = [], []
    for lo in (0, 60, 120):
        hi = min(lo + 60, 150)
        d_, i_ = _lib.knn_topk(torch.from_numpy(train[lo:hi]).to(cuda), torch.from_numpy(query).to(cuda), 6, lo)
        parts_d.append(d_)
        parts_i.append(i_)
    md, mi = _lib.knn_merge(torch.stack(parts_d), torch.stack(parts_i))
    assert torch.equal(mi, idx) and torch.equal(md, dist)


@pytest.mark.parametrize("kind", ["clusters", "duplicates", "near_ties"])
def test_knn_two_stage_search_is_exact(cuda, kind):
    """Problems large enough for the float32 filter + float64 refine path: the answer must be the exact
    float64 one whether the proof passes (well separated neighbours) or the query falls back to the
    float64 sweep (many rows at the same or nearly the same distance as the k-th neighbour)."""
    from magic_amd import _lib
    rng = np.random.default_rng(11)
    d, k, nt, nq = 64, 10, 12000, 600
    centers = rng.standard_normal((20, d)).astype(np.float32)
    if kind == "clusters":
        train = centers[rng.integers(0, 20, nt)] + 0.05 * rng.standard_normal((nt, d)).astype(np.float32)
        query = centers[rng.integers(0, 20, nq)] + 0.05 * rng.standard_normal((nq, d)).astype(np.float32)
    elif kind == "duplicates":
        base = centers[rng.integers(0, 20, 300)] + 0.05 * rng.standard_normal((300, d)).astype(np.float32)
        train = base[rng.integers(0, 300, nt)]                       # every row about 40 times
        query = base[rng.integers(0, 300, nq)] + (rng.random((nq, 1)) < 0.5) * 0.01
    else:
        base = centers[rng.integers(0, 20, 300)] + 0.05 * rng.standard_normal((300, d)).astype(np.float32)
        train = base[rng.integers(0, 300, nt)] + 1e-6 * rng.standard_normal((nt, d)).astype(np.float32)
        query = base[rng.integers(0, 300, nq)] + 0.01
    train, query = train.astype(np.float32), query.astype(np.float32)
    t, q = torch.from_numpy(train).to(cuda), torch.from_numpy(query).to(cuda)
    dist, idx = _lib.knn_topk(t, q, k)
    # blocks small enough for the oracle's direct (q - t)^2 form: identical rows get identical distances there
    wd, wi = knn_oracle.exact_topk(train, query, k, block=200)
    got_i, got_d = idx.cpu().numpy(), dist.cpu().numpy()
    assert np.abs(got_d - wd).max() < 1e-6
    if kind == "near_ties":
        # keys that differ by less than float64 rounding of the two distance formulas may swap: compare the
        # distances of the rows each side picked instead of the ids
        mine = np.sqrt(((query[:, None, :].astype(np.float64) - train[got_i].astype(np.float64)) ** 2).sum(-1))
        assert np.abs(mine - wd).max() < 1e-6
    else:
        assert np.array_equal(got_i, wi)
    # the same search sharded over three train ranges and merged gives the single-search result bit for bit
    parts_d, parts_i = [], []
    for lo in (0, 5000, 10000):
        hi = min(lo + 5000, nt)
        d_, i_ = _lib.knn_topk(t[lo:hi], q, k, lo)
        parts_d.append(d_)
        parts_i.append(i_)
    md, mi = _lib.knn_merge(torch.stack(parts_d), torch.stack(parts_i))
    assert torch.equal(mi, idx) and torch.equal(md, dist)


def test_knn_unsplit_rounds_and_split_tail_agree(cuda):
    """More than 131,072 queries: the first 512 query blocks sweep the train rows unsplit, the remaining
    blocks are cut over the train rows.  Both parts must give what the same queries give when they are
    searched in chunks small enough to take the split path only (that path is checked against the float64
    oracle above), bit for bit, and a sample from each part is checked against the oracle directly.  A third
    of the queries sit on rows that occur about 30 times, so the float64 sweep for rejected queries runs too."""
    from magic_amd import _lib
    rng = np.random.default_rng(23)
    d, k, nt, nq = 24, 5, 9000, 131072 + 30000
    centers = rng.standard_normal((30, d)).astype(np.float32)
    base = centers[rng.integers(0, 30, 100)] + 0.05 * rng.standard_normal((100, d)).astype(np.float32)
    train = centers[rng.integers(0, 30, nt)] + 0.05 * rng.standard_normal((nt, d)).astype(np.float32)
    train[:3000] = base[rng.integers(0, 100, 3000)]
    query = centers[rng.integers(0, 30, nq)] + 0.05 * rng.standard_normal((nq, d)).astype(np.float32)
    dup = rng.random(nq) < 0.33
    query[dup] = base[rng.integers(0, 100, int(dup.sum()))]
    t, q = torch.from_numpy(train).to(cuda), torch.from_numpy(query).to(cuda)
    dist, idx = _lib.knn_topk(t, q, k)
    for lo in range(0, nq, 40000):
        d_, i_ = _lib.knn_topk(t, q[lo:lo + 40000].contiguous(), k)
        assert torch.equal(i_, idx[lo:lo + 40000]) and torch.equal(d_, dist[lo:lo + 40000]), lo
    pick = np.concatenate([rng.integers(0, 131072, 300), rng.integers(131072, nq, 300)])
    wd, wi = knn_oracle.exact_topk(train, query[pick], k, block=200)
    got_d = dist.cpu().numpy()[pick]
    assert np.abs(got_d - wd).max() < 1e-6
    clear = ~dup[pick]                      # rows without exact ties: the ids are determined
    assert np.array_equal(idx.cpu().numpy()[pick][clear], wi[clear])


@pytest.mark.parametrize("n_dim,alpha,single", [(5, 3, False), (11, 3, True), (8, 2, False), (8, 1.5, True),
                                                 (1, 3, True), (16, 3, False), (17, 3, True), (40, 2, False), (64, 3, True)])
def test_sce_kernel(cuda, n_dim, alpha, single):
    from magic_amd.model.loss_func import sce_loss_masked
    torch.manual_seed(0)
    n = 1000
    x = torch.randn(n, n_dim)
    x[3] = 0.0                                    # zero row: F.normalize's eps clamp
    tgt = torch.randn(1, n_dim) * 0.1 if single else torch.nn.functional.one_hot(
        torch.randint(0, n_dim, (n,)), n_dim).float()
    mask = torch.randperm(n)[: n // 2]
    if not bool((mask == 3).any()):     # masks come from a permutation: no duplicates
        mask[0] = 3
    xo = x.double().requires_grad_(True)
    to = tgt.double().requires_grad_(single)
    lo = O.sce_loss(xo[mask], to.expand(mask.numel(), -1) if single else to[mask], alpha=alpha)
    lo.backward()
    xm = x.to(cuda).requires_grad_(True)
    tm = tgt.to(cuda).requires_grad_(single)
    lm = sce_loss_masked(xm, tm, mask.to(cuda), alpha)
    (lm * 1.0).backward()
    assert abs(float(lm) - float(lo)) < 1e-5 * max(1.0, abs(float(lo)))
    assert rel_err(xm.grad, xo.grad) < 1e-4
    if single:
        assert rel_err(tm.grad, to.grad) < 1e-4


def test_sce_zero_token_target(cuda):
    """Zero-initialised mask token as target (model/autoencoder.py:170): cos = 0, loss = 1,
    gradient through the eps clamp of F.normalize."""
    from magic_amd.model.loss_func import sce_loss_masked
    torch.manual_seed(1)
    x = torch.randn(64, 5)
    mask = torch.arange(0, 64, 2)
    to = torch.zeros(1, 5, dtype=torch.float64, requires_grad=True)
    xo = x.double().requires_grad_(True)
    lo = O.sce_loss(xo[mask], to.expand(mask.numel(), -1), alpha=3)
    lo.backward()
    tm = torch.zeros(1, 5, device=cuda, requires_grad=True)
    xm = x.to(cuda).requires_grad_(True)
    lm = sce_loss_masked(xm, tm, mask.to(cuda), 3)
    lm.backward()
    assert float(lm) == pytest.approx(1.0, abs=1e-6)
    assert rel_err(tm.grad, to.grad) < 1e-4
    assert rel_err(xm.grad, xo.grad) < 1e-4 or float(xo.grad.abs().max()) == 0.0


def test_knn_wide_rows_many_train_rows_are_searched_in_blocks(cuda):
    """d = 128 takes the kernel that keeps a [queries, train rows] key matrix; NearestNeighbors cuts the
    queries so that this matrix stays bounded (60,000 train rows: 2,236 queries per call)."""
    from magic_amd.knn import NearestNeighbors
    rng = np.random.default_rng(5)
    train = rng.standard_normal((60000, 128)).astype(np.float32)
    query = rng.standard_normal((5000, 128)).astype(np.float32)
    nn_ = NearestNeighbors(10, device=cuda).fit(train)
    torch.cuda.reset_peak_memory_stats()
    d, i = nn_.kneighbors(query[:5000], 10)
    assert torch.cuda.max_memory_allocated() < 3 * (1 << 30)
    wd, wi = knn_oracle.exact_topk(train, query[:300], 10)
    assert np.array_equal(i[:300], wi) and np.abs(d[:300] - wd).max() < 1e-6
    assert (np.diff(d, axis=1) >= 0).all()


def test_entity_level_scoring_on_device_equals_host_standardisation(cuda):
    """model/eval.py:282-313 with CUDA tensors in (standardise + both searches + row means on the
    device) against the numpy-standardised path of the same function."""
    import random
    from magic_amd import synth
    from magic_amd.model.eval import entity_level_distances
    x_train = synth.gaussian_mixture(30000, 64, seed=3) * 2.5 + 0.7
    x_test = synth.gaussian_mixture(4000, 64, seed=4) * 2.5 + 0.7
    random.seed(9)
    m_host, d_host = entity_level_distances(x_train, x_test, 10)
    random.seed(9)
    m_dev, d_dev = entity_level_distances(torch.from_numpy(x_train).to(cuda), torch.from_numpy(x_test).to(cuda), 10)
    assert d_dev.dtype == torch.float64 and d_dev.device.type == "cuda"
    assert abs(float(m_dev) - float(m_host)) < 1e-5 * float(m_host)
    assert np.abs(d_dev.cpu().numpy() - d_host).max() < 1e-5 * np.abs(d_host).max()
