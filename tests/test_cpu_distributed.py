"""CPU suite, part 2: the multi-rank paths with world_size 2 over gloo.

The collectives and the host logic around them (flat gradient buffer, shard bounds, list
exchange and merge) are device independent; on the GPU box the same code runs over RCCL.
The local top-k lists come from the numpy oracle here (the HIP search needs a GPU)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import knn_oracle


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _dp_worker(rank, world, port, out):
    from magic_amd.parallel import FlatGradAllReduce, broadcast_parameters
    from oracle import gat_oracle as O
    from tests.util import small_graph, to_ograph
    _init(rank, world, port)
    torch.manual_seed(100 + rank)          # different initial weights on purpose
    model = O.build_model(5, 17, 64, 3).eval()
    broadcast_parameters(model)            # ... made identical here
    flat = FlatGradAllReduce(model.parameters())
    n = 150
    src, dst, nt, et = small_graph(n, 600, 5, 17, seed=50 + rank)
    g = to_ograph(src, dst, n, nt, et, 5, 17)
    gen = torch.Generator().manual_seed(rank)
    perm = torch.randperm(n, generator=gen)
    neg = O.uniform_negative_pairs(g, n, seed=rank)
    loss = model.compute_loss(g, perm=perm, pos_idx=torch.arange(n), neg_pairs=neg)
    flat.zero()
    loss.backward()
    local = flat.pack().clone()
    flat.allreduce()
    after = torch.cat([p.grad.reshape(-1) for p in flat.params])     # the averaged gradients landed in .grad
    assert torch.equal(after, flat.flat)
    torch.save({"local": local, "avg": flat.flat.clone(),
                "w0": torch.cat([p.detach().reshape(-1) for p in model.parameters()])}, f"{out}/dp{rank}.pt")
    dist.destroy_process_group()


def test_flat_gradient_allreduce_world2(tmp_path):
    port = _free_port()
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "dp0.pt")
    r1 = torch.load(tmp_path / "dp1.pt")
    assert torch.equal(r0["w0"], r1["w0"])                       # broadcast made replicas identical
    assert not torch.equal(r0["local"], r1["local"])             # different graphs, different grads
    want = (r0["local"] + r1["local"]) / 2
    assert torch.equal(r0["avg"], r1["avg"])                     # every rank holds the same average
    assert torch.allclose(r0["avg"], want, rtol=1e-6, atol=1e-30)


def _knn_worker(rank, world, port, out):
    from magic_amd.knn import ShardedNearestNeighbors, exchange_and_merge
    _init(rank, world, port)
    rng = np.random.default_rng(7)
    base = rng.standard_normal((301, 64)).astype(np.float32)
    train = np.concatenate([base, base[:50]])                    # duplicates straddle the shards
    query = rng.standard_normal((40, 64)).astype(np.float32)
    k = 6
    lo, hi = ShardedNearestNeighbors.shard_bounds(train.shape[0], world, rank)
    d, i = knn_oracle.exact_topk(train[lo:hi], query, k)
    d_t, i_t = torch.from_numpy(d), torch.from_numpy(i + lo)
    md, mi = exchange_and_merge(d_t, i_t)
    np.savez(f"{out}/knn{rank}.npz", d=md.numpy(), i=mi.numpy(), lo=lo, hi=hi)
    dist.destroy_process_group()


def test_sharded_knn_merge_equals_single_search_world2(tmp_path):
    port = _free_port()
    mp.spawn(_knn_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    rng = np.random.default_rng(7)
    base = rng.standard_normal((301, 64)).astype(np.float32)
    train = np.concatenate([base, base[:50]])
    query = rng.standard_normal((40, 64)).astype(np.float32)
    wd, wi = knn_oracle.exact_topk(train, query, 6)
    r0, r1 = np.load(tmp_path / "knn0.npz"), np.load(tmp_path / "knn1.npz")
    assert (int(r0["lo"]), int(r0["hi"]), int(r1["lo"]), int(r1["hi"])) == (0, 176, 176, 351)
    for r in (r0, r1):
        assert np.array_equal(r["i"], wi)            # ties resolved by index, as in one search
        assert np.array_equal(r["d"], wd)


def test_shard_bounds_cover_everything():
    from magic_amd.knn import ShardedNearestNeighbors as S
    from magic_amd.parallel import shard_indices
    for n in (0, 1, 7, 8, 2604530):
        for world in (1, 2, 4, 8):
            b = [S.shard_bounds(n, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[r][1] == b[r + 1][0] for r in range(world - 1))
    assert sorted(sum((shard_indices(10, 4, r) for r in range(4)), [])) == list(range(10))


def test_merge_topk_host_order_rule():
    from magic_amd.knn import merge_topk
    d = torch.tensor([[[0.0, 1.0, 2.0]], [[0.0, 1.0, float("inf")]]], dtype=torch.float64)   # [P=2,Q=1,k=3]
    i = torch.tensor([[[5, 1, 9]], [[2, 7, -1]]])
    md, mi = merge_topk(d, i)
    assert md.tolist() == [[0.0, 0.0, 1.0]] and mi.tolist() == [[2, 5, 1]]


class _HostSearch:
    """Per-shard searcher backed by the numpy oracle (stands in for the HIP search on CPU)."""

    def __init__(self):
        self._train = None

    def fit(self, x):
        self._train = torch.as_tensor(np.asarray(x, dtype=np.float32))
        return self

    def kneighbors_device(self, x, k, index_offset=0):
        d, i = knn_oracle.exact_topk(self._train.numpy(), np.asarray(x, dtype=np.float32), k)
        return torch.from_numpy(d), torch.from_numpy(i + index_offset)

    def kneighbors(self, x, n_neighbors):
        d, i = self.kneighbors_device(x, n_neighbors)
        return d.numpy(), i.numpy()


def _eval_data():
    rng = np.random.default_rng(11)
    return (rng.standard_normal((403, 16)).astype(np.float32) * 3 + 1,
            rng.standard_normal((57, 16)).astype(np.float32) * 3 + 1)


def _entity_worker(rank, world, port, out):
    import random
    from magic_amd.knn import ShardedNearestNeighbors
    from magic_amd.model.eval import entity_level_distances
    _init(rank, world, port)
    x_train, x_test = _eval_data()
    random.seed(3)
    nbrs = ShardedNearestNeighbors(10, device="cpu", local=_HostSearch())
    mean_distance, distances = entity_level_distances(x_train, x_test, 10, nbrs=nbrs)
    # per-query means only (all-to-all, merge of this rank's query block, all-gather of [Q] means)
    xs = (x_test - x_train.mean(0)) / x_train.std(0)
    means = nbrs.mean_distances_device(torch.from_numpy(xs), 10).numpy()
    np.savez(f"{out}/ent{rank}.npz", m=mean_distance, d=distances, means=means)
    dist.destroy_process_group()


def test_entity_level_scores_sharded_equal_single_world2(tmp_path):
    """model/eval.py:282-313 with the train rows sharded over two ranks: same mean distance and
    same per-row distances as one search over all rows."""
    import random
    from magic_amd.model.eval import entity_level_distances
    port = _free_port()
    mp.spawn(_entity_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    x_train, x_test = _eval_data()
    random.seed(3)
    m, d = entity_level_distances(x_train, x_test, 10, nbrs=_HostSearch())
    for r in range(2):
        got = np.load(tmp_path / f"ent{r}.npz")
        assert float(got["m"]) == float(m)
        assert np.array_equal(got["d"], d)
        assert got["means"].shape == (57,) and np.allclose(got["means"], d, rtol=1e-13, atol=0)   # 57 queries: ragged blocks


class _FakeGraph:
    def __init__(self, n):
        self.n = n

    def num_nodes(self):
        return self.n


def _embed_worker(rank, world, port, out):
    from magic_amd.parallel import embed_graphs
    _init(rank, world, port)
    graphs = [_FakeGraph(n) for n in (5, 9, 1, 12, 7)]
    calls = []

    def embed_fn(g):
        calls.append(g.n)
        return torch.arange(g.n * 3, dtype=torch.float32).reshape(g.n, 3) * (g.n + 1)
    outs = embed_graphs(embed_fn, graphs, 3, torch.device("cpu"))
    torch.save({"outs": outs, "calls": calls}, f"{out}/emb{rank}.pt")
    dist.destroy_process_group()


def test_embedding_extraction_round_robin_world2(tmp_path):
    port = _free_port()
    mp.spawn(_embed_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "emb0.pt"), torch.load(tmp_path / "emb1.pt")
    assert r0["calls"] == [5, 1, 7] and r1["calls"] == [9, 12]       # graph i on rank i % 2
    for a, b, n in zip(r0["outs"], r1["outs"], (5, 9, 1, 12, 7)):
        want = torch.arange(n * 3, dtype=torch.float32).reshape(n, 3) * (n + 1)
        assert torch.equal(a, want) and torch.equal(b, want)


def _gather_rows_worker(rank, world, port, out):
    from magic_amd.parallel import gather_rows
    _init(rank, world, port)
    counts = [3, 5]
    mine = torch.arange(counts[rank] * 4, dtype=torch.float32).reshape(counts[rank], 4) + 100 * rank
    got = gather_rows(mine, counts)
    torch.save(got, f"{out}/rows{rank}.pt")
    dist.destroy_process_group()


def test_gather_rows_of_different_heights_world2(tmp_path):
    """Ranks that hold different numbers of rows (the sampled train rows of the sharded KNN scoring):
    every rank ends with all rows in rank order, no padding left."""
    port = _free_port()
    mp.spawn(_gather_rows_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    want = torch.cat([torch.arange(12, dtype=torch.float32).reshape(3, 4),
                      torch.arange(20, dtype=torch.float32).reshape(5, 4) + 100])
    for r in range(2):
        assert torch.equal(torch.load(tmp_path / f"rows{r}.pt"), want)


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` without a launcher must start two ranks itself (before touching a
    GPU) and report n_gpus = 2; --dry-run keeps the launch plumbing and drops the GPU work."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3"],
                         capture_output=True, text=True, timeout=180, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout                      # one JSON line, from rank 0 only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["dry_run"] is True
    # a launcher that gives a different world size is refused instead of mislabelled
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"],
                         capture_output=True, text=True, timeout=60, env=env2, cwd=root)
    assert bad.returncode != 0 and "refusing" in (bad.stderr + bad.stdout)
