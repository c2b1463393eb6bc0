"""Child process of tests/test_gpu_rccl.py: a process group of ONE rank on RCCL, with the single-rank shortcuts
switched off (MAGIC_FORCE_COLLECTIVES=1), so that every collective of the data-parallel and sharded-KNN paths
runs through RCCL with its real dtypes and shapes.  With one rank every collective is an identity: each result
must equal the plain single-process result bit for bit.  Prints "rccl world1 ok" on success."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MAGIC_FORCE_COLLECTIVES"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29531")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", world_size=1, rank=0, device_id=dev)
    from magic_amd import knn as K
    from magic_amd import parallel as P
    from magic_amd.model.autoencoder import GMAEModel
    from magic_amd.model.eval import standardise_device
    assert not P.collectives_off(1)
    calls = {"n": 0}
    for name in ("all_reduce", "broadcast", "all_gather_into_tensor", "all_to_all_single"):
        real = getattr(dist, name)

        def counted(*a, _real=real, **kw):
            calls["n"] += 1
            return _real(*a, **kw)
        setattr(dist, name, counted)

    # ---- flat gradient all-reduce + parameter broadcast on a real model
    torch.manual_seed(0)
    model = GMAEModel(5, 17, 64, 3, 4, "prelu", 0.1, 0.2, True, "BatchNorm", 0.5, "sce", 3).to(dev)
    before = [p.detach().clone() for p in model.parameters()]
    P.broadcast_parameters(model)
    assert all(torch.equal(a, p) for a, p in zip(before, model.parameters()))
    gen = torch.Generator(device=dev).manual_seed(1)
    for p in model.parameters():
        p.grad = torch.randn(p.shape, device=dev, generator=gen)
    grads = [p.grad.clone() for p in model.parameters()]
    flat = P.FlatGradAllReduce(model.parameters())
    flat.allreduce()
    assert all(torch.equal(a, p.grad) for a, p in zip(grads, model.parameters()))
    n_after_dp = calls["n"]
    assert n_after_dp == 2, n_after_dp                      # one broadcast, ONE all-reduce

    # ---- sharded KNN: all-to-all of f64 distances and int64 ids, merge, gathers
    tr = torch.randn(9000, 64, device=dev, generator=gen)
    qu = torch.randn(1001, 64, device=dev, generator=gen)       # not a multiple of anything
    xt, xq = standardise_device(tr, qu, sharded=True)
    xt1, xq1 = standardise_device(tr, qu, sharded=False)
    assert torch.equal(xt, xt1) and torch.equal(xq, xq1)
    single = K.NearestNeighbors(10, device=dev).fit(xt)
    d1, i1 = single.kneighbors_device(xq, 10)
    sh = K.ShardedNearestNeighbors(10, device=dev).fit(x_local=xt, offset=0)
    d2, i2 = sh.kneighbors_device(xq, 10)
    assert d2.dtype == torch.float64 and i2.dtype == torch.int64
    assert torch.equal(d1, d2) and torch.equal(i1, i2)
    assert torch.equal(sh.mean_distances_device(xq, 10), d1.mean(dim=1))
    md, mi, q = K.exchange_and_merge_blocks(d1, i1)
    assert q == 1001 and torch.equal(md[:q], d1) and torch.equal(mi[:q], i1)
    assert torch.equal(K.gather_blocks(md, q), d1)
    rows = torch.randn(37, 64, device=dev, generator=gen)
    assert torch.equal(P.gather_rows(rows, [30]), rows[:30])
    # embedding extraction: graph i from rank i % world, broadcast from there
    outs = P.embed_graphs(lambda g: g, [_FakeGraph(rows)], 64, dev)
    assert torch.equal(outs[0], rows)
    assert calls["n"] >= n_after_dp + 3 + 2 + 2 + 2 + 1 + 1 + 1, calls["n"]
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("rccl world1 ok: %d collectives" % calls["n"])


class _FakeGraph(torch.Tensor):
    """A tensor that answers num_nodes(), so that an identity embed function can stand in for the model."""
    @staticmethod
    def __new__(cls, t):
        return torch.Tensor._make_subclass(cls, t)

    def num_nodes(self):
        return self.shape[0]


if __name__ == "__main__":
    main()
