"""CPU suite, part 1: the oracle against the golden vectors captured from the reference
(tests/golden/make_golden.py), plus host logic that needs no GPU."""
import json
import os
import random
import re

import numpy as np
import pytest
import torch

from oracle import csr_oracle, knn_oracle
from oracle import gat_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


# ----------------------------------------------------------------------------- SCE
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_sce_oracle_matches_reference(tag):
    z = load(f"sce_{tag}.npz")
    x = torch.from_numpy(z["x"]).requires_grad_(True)
    loss = O.sce_loss(x, torch.from_numpy(z["y"]), alpha=int(z["alpha"]))
    loss.backward()
    assert abs(float(loss) - float(z["loss"])) < 1e-7
    assert np.abs(x.grad.numpy() - z["grad_x"]).max() < 1e-7


# ----------------------------------------------------------------------------- KNN
@pytest.mark.parametrize("tag", ["d64k10", "d256k8", "d2048k2"])
def test_knn_oracle_matches_sklearn_search(tag):
    z = load(f"knn_raw_{tag}.npz")
    assert str(z["algorithm"]) == "brute"       # d >= 16: the reference's search is brute force
    tr, q = z["train"].astype(np.float32), z["query"].astype(np.float32)
    d, i = knn_oracle.exact_topk(tr, q, int(z["k"]))
    assert np.array_equal(i, z["idx"])          # tie-free data: identical neighbour indices
    # sklearn expands |x|^2-2xy+|y|^2 (float64 GEMM on float32 chunks): its distances carry
    # float32-level relative error, ~1e-7 * dist, plus ~1e-7 absolute near zero
    assert (np.abs(d - z["dist"]) < 2e-6 + 2e-7 * z["dist"]).all()


def test_knn_oracle_on_duplicates_matches_distances():
    z = load("knn_raw_dups.npz")
    d, i = knn_oracle.exact_topk(z["train"], z["query"], int(z["k"]))
    # exact duplicates tie: sklearn returns tied neighbours in heap order, so only the
    # distance lists (and the index sets up to ties) are comparable
    assert np.abs(d - z["dist"]).max() < 2e-6


@pytest.mark.parametrize("ds,k", [("trace", 10), ("cadets", 200)])
def test_entity_level_scores_match_reference(ds, k):
    z = load("knn_entity.npz")
    random.seed(1234)
    mean_distance, dists, score = knn_oracle.entity_scores(z[f"{ds}_x_train"], z[f"{ds}_x_test"], k)
    assert abs(mean_distance - float(z[f"{ds}_mean_distance"])) < 1e-6
    assert np.abs(dists - z[f"{ds}_dists"]).max() < 2e-6
    from sklearn.metrics import roc_auc_score
    assert abs(roc_auc_score(z[f"{ds}_y_test"], score) - float(z[f"{ds}_auc"])) < 1e-9


@pytest.mark.parametrize("ds,train_count", [("streamspot", 400), ("wget", 100)])
def test_batch_level_scores_match_reference(ds, train_count):
    from sklearn.metrics import roc_auc_score
    z = load("knn_batch.npz")
    x, y = z[f"{ds}_emb"].astype(np.float32), z[f"{ds}_labels"]
    k = min(int(train_count * 0.02), 10)
    benign, attack = np.where(y == 0)[0], np.where(y == 1)[0]
    aucs = []
    for s in range(3):     # model/eval.py:130-162
        random.seed(s)
        np.random.seed(s)
        torch.manual_seed(s)
        np.random.shuffle(benign)
        np.random.shuffle(attack)
        x_train = x[benign[:train_count]]
        x_test = np.concatenate([x[benign[train_count:]], x[attack]], axis=0)
        y_test = np.concatenate([y[benign[train_count:]], y[attack]], axis=0)
        _, score = knn_oracle.batch_scores(x_train, x_test, k)
        aucs.append(roc_auc_score(y_test, score))
    assert abs(np.mean(aucs) - float(z[f"{ds}_auc_rep3"])) < 1e-9
    assert abs(np.std(aucs) - float(z[f"{ds}_std_rep3"])) < 1e-9


# ----------------------------------------------------------------------------- pooling / factories
def test_pooling_fixture_is_self_consistent_and_product_has_no_cpu_path():
    """pool.npz (outputs of the reference's utils/poolers.py) against a plain numpy statement of
    utils/poolers.py:49-73; the product read-out is a HIP kernel and refuses CPU tensors (its
    comparison with the fixture is in the GPU suite)."""
    from magic_amd import _lib
    from magic_amd.graph import Graph
    from magic_amd.utils.poolers import Pooling
    z = load("pool.npz")
    feat, ty = z["feat"].astype(np.float64), z["type"]
    assert np.allclose(feat.mean(0, keepdims=True), z["mean"], atol=1e-6)
    for t in range(8):
        rows = feat[ty == t]
        want = rows.mean(0) if rows.shape[0] else np.zeros(16)
        assert np.allclose(z["mean_typed"][0, t * 16:(t + 1) * 16], want, atol=1e-6)
    g = Graph(torch.zeros(1, dtype=torch.long), torch.zeros(1, dtype=torch.long), 500)
    g.ndata["type"] = torch.from_numpy(ty)
    with pytest.raises(_lib.MagicHipError):
        Pooling("mean")(g, torch.from_numpy(z["feat"]), n_types=8)


def test_create_norm_and_activation_follow_reference():
    from magic_amd.utils.utils import create_activation, create_norm
    assert create_norm("BatchNorm") is None              # the model's spelling never matches
    assert create_norm("batchnorm") is torch.nn.BatchNorm1d
    assert create_norm("layernorm") is torch.nn.LayerNorm
    assert isinstance(create_activation("prelu"), torch.nn.PReLU)
    assert isinstance(create_activation(None), torch.nn.Identity)
    with pytest.raises(NotImplementedError):
        create_activation("swish")
    assert O.create_norm("BatchNorm") is None


# ----------------------------------------------------------------------------- checkpoints
DIMS = {"theia": (5, 17), "trace": (11, 23), "cadets": (5, 13)}


def _dims(manifest):
    return manifest["enc_mask_token"][1], manifest["encoder.gats.0.edge_fc.weight"][1]


@pytest.mark.parametrize("ds", ["theia", "trace", "cadets", "fivedirections"])
def test_state_dict_layout_equals_reference_checkpoints(ds):
    from magic_amd.model.autoencoder import GMAEModel
    man = json.load(open(os.path.join(G, "checkpoint_manifest.json")))[ds]
    n_dim, e_dim = _dims(man)
    mine = GMAEModel(n_dim, e_dim, 64, 3, 4, "prelu", 0.1, 0.2, True, "BatchNorm", 0.5, "sce", 3)
    ref = O.build_model(n_dim, e_dim, 64, 3)
    for model in (mine, ref):
        sd = model.state_dict()
        assert list(sd.keys()) == list(man.keys())
        assert {k: list(v.shape) for k, v in sd.items()} == man
    assert len(man) == 38


def test_reference_checkpoint_loads_into_both_models():
    from magic_amd.model.autoencoder import GMAEModel
    z = load("checkpoint_theia.npz")
    sd = {k: torch.from_numpy(z[k]) for k in z.files}
    mine = GMAEModel(5, 17, 64, 3, 4, "prelu", 0.1, 0.2, True, "BatchNorm", 0.5, "sce", 3)
    ref = O.build_model(5, 17, 64, 3)
    mine.load_state_dict(sd, strict=True)
    ref.load_state_dict(sd, strict=True)
    assert sum(v.numel() for v in sd.values()) == 59122


def test_same_seed_gives_same_initial_weights():
    from magic_amd.model.autoencoder import GMAEModel
    torch.manual_seed(0)
    ref = O.build_model(8, 20, 256, 4)
    torch.manual_seed(0)
    mine = GMAEModel(8, 20, 256, 4, 4, "prelu", 0.1, 0.2, True, "BatchNorm", 0.5, "sce", 3)
    for (k1, v1), (k2, v2) in zip(ref.state_dict().items(), mine.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2), k1
    assert sum(p.numel() for p in mine.parameters()) == 1215181


def test_mask_indices_follow_cpu_generator():
    from magic_amd.model.autoencoder import GMAEModel
    gold = json.load(open(os.path.join(G, "randperm.json")))
    m = GMAEModel(5, 17, 64, 3, 4, "prelu", 0.1, 0.2, True, "BatchNorm", 0.5, "sce", 3)
    for key, want in gold.items():
        seed, n = (int(t) for t in key.split("_"))
        torch.manual_seed(seed)
        p = m._randperm(n, torch.device("cpu"))
        assert p[:16].tolist() == want["head"] and p[-4:].tolist() == want["tail"]
        assert int(p[: n // 2].sum()) == want["sum_head_half"]
    torch.manual_seed(0)
    assert torch.randperm(10).tolist() == gold["0_10"]["head"][:10]


# ----------------------------------------------------------------------------- CSR / plan / batch
def _rand_graph(n, e, seed):
    from magic_amd import synth
    rng = np.random.default_rng(seed)
    return synth.power_law_edges(n, e, rng)


def test_csr_oracle_matches_scipy():
    import scipy.sparse as sp
    n = 400
    src, dst = _rand_graph(n, 3000, 1)
    # scipy sorts the minor index inside every row / column; an edge list ordered by (src, dst)
    # makes "ascending original edge id" the same order in both directions
    order = np.lexsort((dst, src))
    src, dst = src[order], dst[order]
    in_ptr, in_src, in_eid, out_ptr, out_dst, out_eid = csr_oracle.build(src, dst, n)
    data = np.arange(src.size) + 1
    csc = sp.coo_matrix((data, (src, dst)), shape=(n, n)).tocsc()     # column = destination
    assert np.array_equal(csc.indptr, in_ptr) and np.array_equal(csc.indices, in_src)
    assert np.array_equal(csc.data - 1, in_eid)
    csr = sp.coo_matrix((data, (src, dst)), shape=(n, n)).tocsr()     # row = source
    assert np.array_equal(csr.indptr, out_ptr) and np.array_equal(csr.indices, out_dst)
    assert np.array_equal(csr.data - 1, out_eid)


def test_segment_plan_matches_oracle():
    from magic_amd.graph import CHUNK, LIGHT_MAX, build_segment_plan
    n = 300
    src, dst = _rand_graph(n, 6000, 2)
    dst[:700] = 7          # one very long segment
    ptr = csr_oracle.csr_by_key(dst, src, n)[0]
    plan = build_segment_plan(torch.from_numpy(ptr))
    hv, cp, cn, cb, ce = csr_oracle.segment_plan(ptr, LIGHT_MAX, CHUNK)
    assert np.array_equal(plan.heavy_node.numpy(), hv) and np.array_equal(plan.heavy_cptr.numpy(), cp)
    assert np.array_equal(plan.chunk_node.numpy(), cn)
    assert np.array_equal(plan.chunk_beg.numpy(), cb) and np.array_equal(plan.chunk_end.numpy(), ce)
    assert hv.size >= 1 and (ce - cb).max() == CHUNK
    # walk order: the node ids cut into 8 ranges of equal work (edges of light segments + 2 per node), inside
    # a range by ascending length class (0 | 1-4 | (4,8] | (8,16] | ...), ties by ascending id
    deg = np.diff(ptr).astype(np.int64)
    cls = np.where(deg <= 0, 0, np.where(deg <= 4, 1, np.ceil(np.log2(np.maximum(deg, 1))))).astype(np.int64)
    cost = np.where(deg <= LIGHT_MAX, deg, 0) + 2
    xcd = np.minimum((np.cumsum(cost) - cost) * 8 // cost.sum(), 7)
    assert np.array_equal(plan.order.numpy(), np.argsort(xcd * 64 + cls, kind="stable").astype(np.int32))
    xs = plan.xcd_start.numpy()
    assert xs[0] == 0 and xs[8] == n and np.all(np.diff(xs) >= 0)
    for k in range(8):
        assert np.all(xcd[xs[k]:xs[k + 1]] == k)
        assert np.array_equal(np.sort(plan.order.numpy()[xs[k]:xs[k + 1]]), np.arange(xs[k], xs[k + 1]))
    empty = build_segment_plan(torch.zeros(5, dtype=torch.int32))
    assert empty.n_heavy == 0 and empty.n_chunks == 0


def test_batch_concatenates_csr_bit_exact():
    from magic_amd.graph import Graph, batch
    graphs, srcs, dsts, off = [], [], [], 0
    for i in range(4):
        n = 100 + 13 * i
        src, dst = _rand_graph(n, 500 + 50 * i, 10 + i)
        g = Graph(torch.from_numpy(src), torch.from_numpy(dst), n)
        g.structure.set_csr(*(torch.from_numpy(a) for a in csr_oracle.build(src, dst, n)))
        g.ndata["type"] = torch.full((n,), i)
        graphs.append(g)
        srcs.append(src + off)
        dsts.append(dst + off)
        off += n
    bg = batch(graphs)
    want = csr_oracle.build(np.concatenate(srcs), np.concatenate(dsts), off)
    s = bg.structure
    for w, got in zip(want, (s.in_ptr, s.in_src, s.in_eid, s.out_ptr, s.out_dst, s.out_eid)):
        assert got.dtype == torch.int32 and np.array_equal(w, got.numpy())
    assert bg.num_nodes() == off and bg.batch_num_nodes() == [g.num_nodes() for g in graphs]
    assert torch.equal(bg.ndata["type"], torch.cat([g.ndata["type"] for g in graphs]))


def test_graph_clone_shares_feature_storage_like_dgl():
    from magic_amd.graph import Graph
    g = Graph(torch.tensor([0, 1]), torch.tensor([1, 2]), 3)
    g.ndata["attr"] = torch.eye(3)
    c = g.clone()
    c.ndata["attr"][1] = 7.0              # in-place write is visible through the original
    assert float(g.ndata["attr"][1, 0]) == 7.0
    c.ndata["new"] = torch.zeros(3)       # but the dictionaries are separate
    assert "new" not in g.ndata
    with g.local_scope():
        g.ndata["tmp"] = torch.ones(3)
    assert "tmp" not in g.ndata


def test_oracle_alias_switch_changes_only_the_sce_target():
    from tests.util import small_graph, to_ograph
    n = 200
    src, dst, nt, et = small_graph(n, 800, 5, 17, 3)
    torch.manual_seed(0)
    m = O.build_model(5, 17, 64, 3).eval()
    with torch.no_grad():
        m.enc_mask_token.fill_(0.25)
    perm = torch.randperm(n)
    pos = torch.arange(n)
    neg = (torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long))
    outs = {}
    for alias in (True, False):
        m.alias_clone = alias
        g = to_ograph(src, dst, n, nt, et, 5, 17)
        _, parts = m.compute_loss(g, perm=perm, pos_idx=pos, neg_pairs=neg, return_parts=True)
        outs[alias] = parts
        touched = not torch.equal(g.ndata["attr"], torch.nn.functional.one_hot(torch.from_numpy(nt), 5).float())
        assert touched == alias
    assert torch.equal(outs[True]["recon"], outs[False]["recon"])
    assert float(outs[True]["sce"]) != float(outs[False]["sce"])
    assert float(outs[True]["bce"]) == float(outs[False]["bce"])


# ----------------------------------------------------------------------------- C ABI
def test_library_exports_every_declared_symbol():
    from magic_amd import _lib
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "magic_hip.h")).read()
    declared = set(re.findall(r"\b(magic_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no prototypes found"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.magic_abi_version() >= 1
    assert lib.magic_status_string(0) == b"ok"
    assert lib.magic_status_string(-3) == b"workspace missing or too small"
    # size queries are host-only arithmetic and must work without a GPU
    assert lib.magic_gat_fwd_workspace_bytes(1000, 10, 4, 16) > 1000 * 4 * 4
    assert lib.magic_knn_workspace_bytes(1000, 10, 64, 5) >= 8000


def test_product_path_has_no_cpu_fallback():
    from magic_amd import _lib
    from magic_amd.graph import Graph
    from magic_amd.model.gat import GATConv
    from magic_amd.model.loss_func import sce_loss
    g = Graph(torch.tensor([0, 1]), torch.tensor([1, 2]), 3)
    g.edata["attr"] = torch.eye(2)
    conv = GATConv(4, 2, 4, 4)
    with pytest.raises(_lib.MagicHipError):
        conv(g, torch.randn(3, 4))
    with pytest.raises(_lib.MagicHipError):
        sce_loss(torch.randn(3, 4), torch.randn(3, 4))
    with pytest.raises(RuntimeError):
        g.structure.ensure_csr()
    with pytest.raises(_lib.MagicHipError):
        _lib.knn_topk(torch.randn(8, 64), torch.randn(2, 64), 2)


def test_product_does_not_import_the_oracle():
    root = os.path.join(os.path.dirname(__file__), "..", "magic_amd")
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f


def test_sample_range_equals_random_sample():
    """The array form of random.sample(range(n), k): same numbers, same generator state afterwards."""
    import random

    from magic_amd.utils.utils import sample_range
    for n, k, seed in [(2040000, 10000, 0), (170000, 10000, 1), (70000, 10000, 5), (100, 10, 3), (65558, 10000, 2),
                       (65557, 10000, 2), (3000000, 10000, 9), (2 ** 31 + 5, 1000, 4), (12715, 8407, 6), (40000, 10000, 8)]:
        random.seed(seed)
        want = random.sample(range(n), k)
        after = [random.random(), random.getrandbits(64)]
        random.seed(seed)
        got = sample_range(n, k)
        assert got.tolist() == want
        assert [random.random(), random.getrandbits(64)] == after
    with pytest.raises(ValueError):
        sample_range(5, 6)


def test_node_link_reader_equals_networkx_route():
    """graph_from_node_link against the route the reference takes (utils/loaddata.py:62-66):
    nx.node_link_graph, then DGL's relabelling (convert_node_labels_to_integers, ordering='sorted')
    and its edge / attribute order (iteration order of the relabelled DiGraph)."""
    nx = pytest.importorskip("networkx")
    import json
    import warnings
    from magic_amd.utils.loaddata import graph_from_node_link
    rng = np.random.default_rng(3)
    for trial in range(4):
        n = 40 + 10 * trial
        labels = rng.permutation(np.arange(100, 100 + 3 * n, 3))[:n].tolist()      # shuffled, gapped ids
        nodes = [{"type": int(rng.integers(0, 8)), "id": int(lab)} for lab in labels]
        links = []
        for _ in range(6 * n):
            u, v = (int(x) for x in rng.choice(labels[: n - 3], 2))                # three nodes stay isolated
            links.append({"type": int(rng.integers(0, 20)), "source": u, "target": v})
        links += [{"type": int(rng.integers(0, 20)), "source": e["source"], "target": e["target"]}
                  for e in links[:15]]                                             # repeated pairs, new attributes
        data = {"directed": True, "multigraph": False, "graph": {}, "nodes": nodes, "links": links}
        data = json.loads(json.dumps(data))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                gx = nx.node_link_graph(data, edges="links")
            except TypeError:
                gx = nx.node_link_graph(data)
        hx = nx.convert_node_labels_to_integers(gx, ordering="sorted")
        want_edges = [(u, v, d["type"]) for u, v, d in hx.edges(data=True)]
        want_ntype = [hx.nodes[i]["type"] for i in range(hx.number_of_nodes())]
        g = graph_from_node_link(data)
        src, dst = g.edges()
        got_edges = list(zip(src.tolist(), dst.tolist(), g.edata["type"].tolist()))
        assert g.num_nodes() == hx.number_of_nodes()
        assert got_edges == want_edges
        assert g.ndata["type"].tolist() == want_ntype


def test_json_dataset_files_and_labels(tmp_path):
    """Wget layout of utils/loaddata.py:133-150: <root>/wget/final/{0..149}.json, graphs 0-24 attacks."""
    import json
    from magic_amd.utils.loaddata import load_batch_level_dataset, load_json_dataset
    d = tmp_path / "wget" / "final"
    d.mkdir(parents=True)
    for i in range(150):
        data = {"directed": True, "multigraph": False, "graph": {},
                "nodes": [{"type": (i + j) % 5, "id": j} for j in range(4)],
                "links": [{"type": (i + j) % 3, "source": j, "target": (j + 1) % 4} for j in range(4)]}
        (d / f"{i}.json").write_text(json.dumps(data))
    ds = load_json_dataset("wget", root=str(tmp_path))
    assert len(ds) == 150 and [y for _, y in ds] == [1] * 25 + [0] * 125
    info = load_batch_level_dataset("wget")
    assert info["n_feat"] == 5 and info["e_feat"] == 3 and info["train_index"] == list(range(25, 150))
    g = ds[7][0]
    assert g.num_nodes() == 4 and g.edges()[0].tolist() == [0, 1, 2, 3] and g.edges()[1].tolist() == [1, 2, 3, 0]


def test_custom_ops_are_registered_with_schemas_and_fake_kernels():
    """torch.ops.magic.*: every operator resolves, has a schema, and its fake kernel infers output
    shapes / dtypes without a GPU (FakeTensorMode); a real CPU call has no kernel."""
    import magic_amd  # noqa: F401
    from torch._subclasses.fake_tensor import FakeTensorMode
    names = ["csr_build", "proj_fwd", "proj_bwd_input", "proj_bwd_input_", "proj_bwd_weight", "gat_fwd_", "gat_bwd_",
             "sce_fwd_bwd", "sce_loss", "linear_nobias", "knn_topk", "knn_merge", "readout", "negative_filter", "edge_head_fwd", "edge_head_bwd",
             "coded_table", "node_code", "gat_fwd_coded_", "gat_bwd_coded_", "coded_weight_grad"]
    for n in names:
        op = getattr(torch.ops.magic, n)
        assert "magic::" + n in str(op.default._schema)
    assert "Tensor(a!) out" in str(torch.ops.magic.gat_fwd_.default._schema)
    assert "Tensor? dst_active" in str(torch.ops.magic.gat_fwd_.default._schema)
    assert "Tensor? dst_active" in str(torch.ops.magic.gat_bwd_.default._schema)
    with FakeTensorMode():
        dev = "cuda"
        n, e, h, d = 100, 400, 4, 16
        i32 = dict(dtype=torch.int32, device=dev)
        plan = [torch.empty(n + 1, **i32)] + [torch.empty(0, **i32) for _ in range(5)] + \
            [torch.empty(n, **i32), torch.empty(9, **i32)]
        f = lambda *s: torch.empty(*s, device=dev)   # noqa: E731
        out = f(n, h * d)
        pre, attn, stat = torch.ops.magic.gat_fwd_(*plan, torch.empty(e, **i32), torch.empty(e, **i32), h, d,
                                                   f(n, h * d), f(n, h), f(n, h), f(17, h), torch.empty(e, **i32),
                                                   None, 0.2, f(h * d), None, f(1), True, None, 1.0, True, True, None, out)
        assert pre.shape == (n, h * d) and attn.shape == (e, h) and stat.shape == (n, h, 2)
        d_table, d_logit, d_bias, d_prelu = torch.ops.magic.gat_bwd_(
            *plan, *plan, torch.empty(e, **i32), torch.empty(e, **i32), torch.empty(e, **i32), h, d, f(n, h * d),
            f(n, h), stat, f(17, h), torch.empty(e, **i32), None, 17, 0.2, None, None, None, f(n, h * d), f(n, h * d),
            f(n, h * d + 2 * h), True, False, None)
        assert d_table.shape == (17, h) and d_logit.numel() == 0 and d_bias.shape == (h * d,) and d_prelu.numel() == 0
        ptrs = torch.ops.magic.csr_build(torch.empty(e, dtype=torch.int64, device=dev),
                                         torch.empty(e, dtype=torch.int64, device=dev), n)
        assert ptrs[0].shape == (n + 1,) and ptrs[1].dtype == torch.int32 and ptrs[5].shape == (e,)
        dist, idx = torch.ops.magic.knn_topk(f(1000, 64), f(37, 64), 10, 0)
        assert dist.shape == (37, 10) and dist.dtype == torch.float64 and idx.dtype == torch.int64
        assert torch.ops.magic.readout(f(n, 256), torch.empty(n, dtype=torch.int64, device=dev), 8, True).shape == (1, 2048)
        hs, score, res = torch.ops.magic.proj_fwd(f(n, 8), f(64 + 8 + 64, 8), 64, 8, 64, None, 1.0, None, None)
        assert hs.shape == (n, 64) and score.shape == (n, 8) and res.shape == (n, 64)
        loss, d_x, d_t = torch.ops.magic.sce_fwd_bwd(f(n, 8), 5, f(1, 5), True, torch.empty(50, dtype=torch.int64, device=dev), 3.0)
        assert loss.shape == (1,) and d_x.shape == (n, 8) and d_t.shape == (5,)
        i64 = dict(dtype=torch.int64, device=dev)
        ok = torch.ops.magic.negative_filter(plan[0], torch.empty(e, **i32), plan[0], torch.empty(e, **i32),
                                             torch.empty(125, **i64), torch.empty(125, **i64), 100)
        assert ok.shape == (125,) and ok.dtype == torch.bool
        loss, d_logit, wsum = torch.ops.magic.edge_head_fwd(f(225, 64), f(64), f(1, 64), f(1), None, 100, 0.2)
        assert loss.shape == (1,) and d_logit.shape == (225,) and wsum.shape == (1,)
        d_h, d_b1, d_w2, d_b2 = torch.ops.magic.edge_head_bwd(f(225, 64), f(64), f(1, 64), d_logit, wsum, f(1), 0.2)
        assert d_h.shape == (225, 64) and d_b1.shape == (64,) and d_w2.shape == (64,) and d_b2.shape == (1,)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.magic.knn_topk(torch.zeros(10, 4), torch.zeros(2, 4), 1, 0)


# ----------------------------------------------------------------------------- detection tail, loaders
def test_device_metric_tail_equals_sklearn_and_the_reference_loop():
    """model/metrics.py (torch sort + scan) against scikit-learn's roc_auc_score / precision_recall_curve
    and the threshold walk + confusion loop of model/eval.py:327-357, with and without tied scores."""
    from sklearn.metrics import precision_recall_curve, roc_auc_score
    from magic_amd.model import metrics as M
    rng = np.random.default_rng(0)
    for trial, ds in enumerate(["trace", "theia", "cadets", "fivedirections"]):
        n = 6000
        y = (rng.random(n) < 0.05).astype(np.float64)
        s = rng.standard_normal(n) + 4.0 * y
        if trial % 2:
            s = np.round(s, 1)
        ys, ss = torch.from_numpy(y), torch.from_numpy(s)
        assert abs(float(M.roc_auc(ys, ss)) - roc_auc_score(y, s)) < 1e-12
        p, r, t = precision_recall_curve(y, s)
        p2, r2, t2 = M.precision_recall_curve(ys, ss)
        assert np.array_equal(t, t2.numpy()) and np.allclose(p, p2.numpy(), atol=1e-15) and np.allclose(r, r2.numpy(), atol=1e-15)
        f1 = 2 * p * r / (r + p + 1e-9)
        best = -1
        for i in range(len(f1)):                      # model/eval.py:331-341
            if ds in M.RECALL_FLOOR and r[i] < M.RECALL_FLOOR[ds]:
                best = i - 1
                break
        thr = t[best]
        want = dict(tp=int(((y == 1) & (s >= thr)).sum()), fn=int(((y == 1) & (s < thr)).sum()),
                    tn=int(((y == 0) & (s < thr)).sum()), fp=int(((y == 0) & (s >= thr)).sum()))
        got = M.entity_level_detection_metrics(ds, ys, ss)
        assert {k: got[k] for k in want} == want and got["threshold"] == thr
        assert abs(got["f1"] - f1[best]) < 1e-12 and abs(got["recall"] - r[best]) < 1e-15
    # eval.py:99-114
    x = torch.arange(40.).view(10, 4)
    xs, ys = M.entity_level_test_set(x, [1, 7], 6)
    assert xs[:, 0].tolist() == [4.0, 24.0, 28.0, 32.0, 36.0] and ys.tolist() == [1.0, 0.0, 1.0, 0.0, 0.0]


def test_entity_level_loaders_read_the_parser_outputs(tmp_path):
    """train.pkl / test.pkl / malicious.pkl as utils/trace_parser.py:235-237 writes them (networkx
    node-link dictionaries) -> metadata.json with the reference's keys + per-graph npz caches; a pickle
    that references any global is refused."""
    import json
    import pickle
    import networkx as nx
    from magic_amd.utils import loaddata as L
    rng = np.random.default_rng(3)

    def make(n, e):
        g = nx.DiGraph()
        for i in rng.permutation(n):
            g.add_node(int(i), type=int(rng.integers(0, 6)))
        for _ in range(e):
            u, v = int(rng.integers(0, n)), int(rng.integers(0, n))
            g.add_edge(u, v, type=int(rng.integers(0, 9)))
        return g
    base = tmp_path / "data" / "toy"
    base.mkdir(parents=True)
    train, test = [make(30, 80), make(25, 60)], [make(40, 90)]
    pickle.dump([nx.node_link_data(g) for g in train], open(base / "train.pkl", "wb"))
    pickle.dump([nx.node_link_data(g) for g in test], open(base / "test.pkl", "wb"))
    pickle.dump(([3, 17], ["a", "b"]), open(base / "malicious.pkl", "wb"))
    root = str(tmp_path / "data")
    meta = L.load_metadata("toy", root=root)
    assert set(meta) == {"node_feature_dim", "edge_feature_dim", "malicious", "n_train", "n_test"}
    assert meta["n_train"] == 2 and meta["n_test"] == 1 and meta["malicious"] == [[3, 17], ["a", "b"]]
    assert json.load(open(base / "metadata.json"))["node_feature_dim"] == meta["node_feature_dim"]
    for kind, gs in (("train", train), ("test", test)):
        for i, gx in enumerate(gs):
            g = L.load_entity_level_dataset("toy", kind, i, root=root)
            want = L.graph_from_node_link(nx.node_link_data(gx))
            assert torch.equal(g.edges()[0], want.edges()[0]) and torch.equal(g.edges()[1], want.edges()[1])
            assert torch.equal(g.ndata["type"], want.ndata["type"]) and torch.equal(g.edata["type"], want.edata["type"])
            assert g.ndata["attr"].shape == (g.num_nodes(), meta["node_feature_dim"])
            assert g.edata["attr"].shape == (g.num_edges(), meta["edge_feature_dim"])
            # same node numbering / edge order as dgl.from_networkx: sorted node labels, DiGraph edge order
            nodes = sorted(gx.nodes())
            assert [(nodes.index(u), nodes.index(v)) for u, v in gx.edges()] == \
                list(zip(g.edges()[0].tolist(), g.edges()[1].tolist()))
    pickle.dump([np.arange(3)], open(base / "evil.pkl", "wb"))          # needs numpy globals to load
    with pytest.raises(pickle.UnpicklingError):
        L.load_plain_pickle(str(base / "evil.pkl"))


def test_oracle_sparse_ops_match_scipy_sparse_products():
    """Independent restatement of the two DGL operators the GAT layer rests on, in scipy.sparse: with A the
    [N, E] destination incidence matrix, the per-destination softmax is exp(e - rowmax) / (A exp(e - rowmax))
    and the aggregation is the sparse product (dst, src, a) @ hs per head.  Also the softmax backward identity
    DGL's EdgeSoftmax uses (g_e = a * (g - sum_seg a * g)) against autograd through the restatement."""
    import scipy.sparse as sp
    rng = np.random.default_rng(11)
    for n, e, heads, width in ((50, 400, 4, 16), (300, 900, 1, 8), (64, 64, 4, 4)):
        src = rng.integers(0, n, e)
        dst = rng.integers(0, max(1, n - n // 4), e)             # the last quarter of the nodes has no in-edges
        dst[: e // 5] = 0                                        # a hub
        logits = rng.standard_normal((e, heads, 1)) * 3.0
        hs = rng.standard_normal((n, heads, width))
        ts, td = torch.from_numpy(src), torch.from_numpy(dst)
        a = O.edge_softmax(td, torch.from_numpy(logits), n).numpy()
        inc = sp.csr_matrix((np.ones(e), (dst, np.arange(e))), shape=(n, e))
        for h in range(heads):
            x = logits[:, h, 0]
            rowmax = np.full(n, -np.inf)
            np.maximum.at(rowmax, dst, x)
            ex = np.exp(x - rowmax[dst])
            want = ex / (inc @ ex)[dst]
            assert np.abs(a[:, h, 0] - want).max() < 1e-14
            assert np.abs(inc @ a[:, h, 0] - (np.bincount(dst, minlength=n) > 0)).max() < 1e-12   # rows sum to 1
            adj = sp.csr_matrix((a[:, h, 0], (dst, src)), shape=(n, n))      # duplicate (dst, src) pairs add up
            got = O.u_mul_e_sum(ts, td, torch.from_numpy(hs), torch.from_numpy(a), n).numpy()[:, h]
            assert np.abs(got - adj @ hs[:, h]).max() < 1e-12
        assert not got[n - n // 4:].any() if n // 4 else True                # no in-edges -> zeros
        # softmax backward: autograd through the restatement vs the closed form on segment sums
        lt = torch.from_numpy(logits).requires_grad_(True)
        g = torch.from_numpy(rng.standard_normal((e, heads, 1)))
        at = O.edge_softmax(td, lt, n)
        (at * g).sum().backward()
        ag = (at.detach() * g).numpy()[:, :, 0]
        seg = np.stack([inc @ ag[:, h] for h in range(heads)], 1)
        want_grad = ag - at.detach().numpy()[:, :, 0] * seg[dst]
        assert np.abs(lt.grad.numpy()[:, :, 0] - want_grad).max() < 1e-12


def test_oracle_gat_layer_matches_a_dense_adjacency_restatement():
    """Second, independently shaped statement of the layer ``oracle.GATConv`` restates (``model/gat.py:284-335``): a
    dense [N, N, H] logit array with -inf where there is no edge, a row softmax over sources, and an einsum for the
    aggregation -- no edge lists, no segment operators.  Duplicate-free graphs only (a dense array holds one edge per
    pair); nodes without in-edges, self-loops, a hub, both residual kinds and the activation are covered."""
    torch.manual_seed(3)
    rng = np.random.default_rng(9)
    n, n_dim, e_dim, H, D = 37, 6, 5, 4, 8
    pairs = {(int(a), int(b)) for a, b in zip(rng.integers(0, n, 260), rng.integers(0, n, 260))}
    pairs |= {(u, 3) for u in range(0, n, 2)} | {(5, 5), (9, 9)}           # a hub and two self-loops
    pairs = sorted(p for p in pairs if p[1] not in (0, 1, 2))               # nodes 0..2 keep no in-edge
    src = torch.tensor([p[0] for p in pairs])
    dst = torch.tensor([p[1] for p in pairs])
    for in_dim in (n_dim, H * D):                                           # linear / identity residual
        g = O.OGraph(src, dst, n)
        etype = torch.from_numpy(rng.integers(0, e_dim, len(pairs)))
        g.edata["attr"] = torch.nn.functional.one_hot(etype, e_dim).double()
        layer = O.GATConv(in_dim, e_dim, D, H, residual=True, activation=torch.nn.PReLU(), negative_slope=0.2).double()
        with torch.no_grad():
            layer.bias.normal_()
        x = torch.randn(n, in_dim, dtype=torch.float64)
        got, attn = layer(g, x, get_attention=True)
        # dense restatement
        feat = (x @ layer.fc.weight.t()).view(n, H, D)
        eh = (feat * layer.attn_h[0]).sum(-1)                               # [N, H] score as a source
        et = (feat * layer.attn_t[0]).sum(-1)                               # [N, H] score as a destination
        ee_type = ((layer.edge_fc.weight.t().view(e_dim, H, D)) * layer.attn_e[0]).sum(-1)   # [e_dim, H]
        logit = torch.full((n, n, H), -float("inf"), dtype=torch.float64)   # [dst, src, head]
        logit[dst, src] = torch.nn.functional.leaky_relu(eh[src] + ee_type[etype] + et[dst], 0.2)
        w = torch.softmax(logit, dim=1)
        w = torch.where(torch.isnan(w), torch.zeros_like(w), w)            # rows without any edge
        out = torch.einsum("vuh,uhd->vhd", w, feat) + layer.bias.view(1, H, D)
        res = x.view(n, H, D) if in_dim == H * D else (x @ layer.res_fc.weight.t()).view(n, H, D)
        want = layer.activation((out + res).flatten(1))
        assert float((got - want).abs().max()) < 1e-12
        assert float((attn.squeeze(-1) - w[dst, src]).abs().max()) < 1e-14
        assert float(out[:3].sub(layer.bias.view(1, H, D)).abs().max()) == 0.0      # no in-edges: bias + residual only
