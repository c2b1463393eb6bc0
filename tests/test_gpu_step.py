"""The step loops (train.py:107-118 and model/train.py:18-48 of the reference) against the float64
oracle over a three-step Adam trajectory with injected randomness, and the HIP-graph form of the
step against the eager step (bitwise)."""
import copy

import numpy as np
import pytest
import torch

from oracle import gat_oracle as O
from tests.util import small_graph, to_graph, to_ograph

pytestmark = pytest.mark.gpu
N_DIM, E_DIM, HIDDEN, LAYERS = 5, 17, 64, 3


def _twins(dev, seed=0, dims=(N_DIM, E_DIM, HIDDEN, LAYERS)):
    from magic_amd.model.autoencoder import GMAEModel
    torch.manual_seed(seed)
    ref = O.build_model(*dims).double()
    torch.manual_seed(seed)
    mine = GMAEModel(dims[0], dims[1], dims[2], dims[3], 4, "prelu", 0.1, 0.2, True, "BatchNorm", 0.5, "sce", 3).to(dev)
    with torch.no_grad():
        ref.enc_mask_token.normal_(0, 0.3)
        mine.enc_mask_token.copy_(ref.enc_mask_token.float())
    return ref, mine


def _inject(n, e, hidden, layers, n_dim, seed):
    """Permutation, positive edge ids, negative pairs and dropout keep masks of one step."""
    gen = torch.Generator().manual_seed(seed)
    t = min(10000, n)
    widths = [n_dim] + [hidden] * (layers - 1) + [hidden]
    return dict(perm=torch.randperm(n, generator=gen), pos_idx=torch.randperm(e, generator=gen)[:t],
                neg_pairs=(torch.randint(0, n, (t,), generator=gen), torch.randint(0, n, (t,), generator=gen)),
                drop_masks=[(torch.rand(n, w, generator=gen) < 0.9) for w in widths])


def _compare_params(ref, mine, lr, steps):
    """Adam's first updates have magnitude ~lr whatever the gradient's size, so an element whose
    gradient is rounding noise may move in the other direction; such elements are rare and bounded
    by steps * 2 * lr.  Everything else must agree to fp32 accuracy."""
    for (k, po), (_, pm) in zip(ref.named_parameters(), mine.named_parameters()):
        d = (pm.detach().double().cpu() - po.detach()).abs().flatten()
        assert float(d.max()) <= steps * 2 * lr * 1.01, k
        assert float(torch.quantile(d, 0.995)) < 2e-5, (k, float(torch.quantile(d, 0.995)))


def test_entity_level_loop_three_adam_steps_equal_oracle(cuda):
    from magic_amd.model.train import entity_level_train
    ref, mine = _twins(cuda)
    graphs = []
    for i in range(3):
        n, e = 500 + 60 * i, 2400 + 100 * i
        graphs.append((n,) + small_graph(n, e, N_DIM, E_DIM, seed=40 + i, hub_deg=150, hub_out=100))
    inj = [_inject(n, src.size, HIDDEN, LAYERS, N_DIM, 200 + i) for i, (n, src, *_r) in enumerate(graphs)]
    lr, wd = 1e-3, 5e-4
    # oracle: the reference's loop (train.py:107-118) in float64
    opt_o = torch.optim.Adam(ref.parameters(), lr=lr, weight_decay=wd)
    ref.train()
    losses_o = []
    for i, (n, src, dst, nt, et) in enumerate(graphs):
        og = to_ograph(src, dst, n, nt, et, N_DIM, E_DIM, torch.float64)
        kw = dict(inj[i], drop_masks=[m.double() for m in inj[i]["drop_masks"]])
        loss = ref(og, **kw) / len(graphs)
        opt_o.zero_grad()
        loss.backward()
        opt_o.step()
        losses_o.append(float(loss))
    # product loop
    from magic_amd.utils.utils import create_optimizer
    opt_m = create_optimizer("adam", mine, lr, wd)
    losses_m = []
    h = mine.register_forward_hook(lambda m, a, out: losses_m.append(out.detach() / len(graphs)))

    def load(i):
        n, src, dst, nt, et = graphs[i]
        return to_graph(src, dst, n, nt, et, N_DIM, E_DIM, "cpu")

    def hook(epoch, i, g):
        return dict(inj[i], perm=inj[i]["perm"].to(cuda))
    entity_level_train(mine, load, len(graphs), opt_m, 1, cuda, N_DIM, E_DIM, progress=False, step_hook=hook)
    h.remove()
    for lo, lm in zip(losses_o, losses_m):
        assert abs(float(lm) - lo) / abs(lo) < 1e-4, (float(lm), lo)
    _compare_params(ref, mine, lr, len(graphs))


def test_batch_level_loop_three_adam_steps_equal_oracle(cuda):
    from magic_amd.graph import Graph
    from magic_amd.model.train import batch_level_train
    dims = (8, 20, 64, 2)
    ref, mine = _twins(cuda, seed=2, dims=dims)
    n_dim, e_dim, hidden, layers = dims
    raw = []
    for i in range(4):
        n, e = 300 + 30 * i, 1500 + 200 * i
        raw.append((n,) + small_graph(n, e, n_dim, e_dim, seed=60 + i, hub_deg=120))
    loader = [[0, 1], [2, 3], [1, 3]]
    sizes = [(sum(raw[j][0] for j in b), sum(raw[j][1].size for j in b)) for b in loader]
    inj = [_inject(n, e, hidden, layers, n_dim, 300 + k) for k, (n, e) in enumerate(sizes)]
    lr, wd = 1e-3, 5e-4
    opt_o = torch.optim.Adam(ref.parameters(), lr=lr, weight_decay=wd)
    ref.train()
    losses_o = []
    ographs = []
    for n, src, dst, nt, et in raw:
        og = O.OGraph(torch.from_numpy(src), torch.from_numpy(dst), n)
        og.ndata["type"], og.edata["type"] = torch.from_numpy(nt), torch.from_numpy(et)
        ographs.append(og)
    for k, b in enumerate(loader):                      # model/train.py:36-46
        bg = O.obatch([O.transform_graph(ographs[j], n_dim, e_dim) for j in b])
        bg.ndata["attr"], bg.edata["attr"] = bg.ndata["attr"].double(), bg.edata["attr"].double()
        kw = dict(inj[k], drop_masks=[m.double() for m in inj[k]["drop_masks"]])
        loss = ref(bg, **kw)
        opt_o.zero_grad()
        loss.backward()
        opt_o.step()
        losses_o.append(float(loss))
    graphs = []
    for n, src, dst, nt, et in raw:
        g = Graph(torch.from_numpy(src), torch.from_numpy(dst), n)
        g.ndata["type"], g.edata["type"] = torch.from_numpy(nt), torch.from_numpy(et)
        graphs.append((g, 0))
    opt_m = torch.optim.Adam(mine.parameters(), lr=lr, weight_decay=wd)
    losses_m = []
    h = mine.register_forward_hook(lambda m, a, out: losses_m.append(out.detach()))
    batch_level_train(mine, graphs, loader, opt_m, 1, cuda, n_dim, e_dim, progress=False,
                      step_hook=lambda ep, k, bg: dict(inj[k], perm=inj[k]["perm"].to(cuda)))
    h.remove()
    for lo, lm in zip(losses_o, losses_m):
        assert abs(float(lm) - lo) / abs(lo) < 1e-4, (float(lm), lo)
    _compare_params(ref, mine, lr, len(loader))


@pytest.mark.parametrize("inject_neg", [True, False])
def test_recorded_step_equals_eager_step_bitwise(cuda, inject_neg):
    """Five steps through the HIP graph (two eager warm-up calls, one recording, replays) against
    five eager steps of a twin model with the same permutation / samples / dropout seeds."""
    from magic_amd.model.step import CapturedTrainStep, make_adam
    _, a = _twins(cuda, seed=5)
    b = copy.deepcopy(a)
    n, e = 900, 4000
    src, dst, nt, et = small_graph(n, e, N_DIM, E_DIM, seed=91, hub_deg=300, hub_out=150)
    ga = to_graph(src, dst, n, nt, et, N_DIM, E_DIM, cuda)
    gb = to_graph(src, dst, n, nt, et, N_DIM, E_DIM, cuda)
    sa = CapturedTrainStep(a, make_adam(a, 1e-3, 5e-4), ga, N_DIM, loss_scale=0.25)
    sb = CapturedTrainStep(b, make_adam(b, 1e-3, 5e-4), gb, N_DIM, loss_scale=0.25, capture=True, warmup=2)
    gen = torch.Generator().manual_seed(8)
    t = min(10000, n)
    losses = []
    for step in range(5):
        kw = dict(perm=torch.randperm(n, generator=gen), pos_idx=torch.randperm(src.size, generator=gen)[:t],
                  seeds=torch.randint(0, 2 ** 62, (LAYERS + 1,), generator=gen))
        if inject_neg:
            kw["neg_pairs"] = (torch.randint(0, n, (t,), generator=gen), torch.randint(0, n, (t,), generator=gen))
            kw["neg_valid"] = torch.rand(t, generator=gen) < 0.9
            la, lb = sa(**kw).clone(), sb(**kw).clone()
            assert torch.equal(la, lb), step
            for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
                assert torch.equal(pa, pb), (step, k)
        else:
            lb = sb(**kw).clone()           # negative pairs drawn inside the recorded step
        losses.append(float(lb))
    assert sb._graphs is not None and all(np.isfinite(losses))
    if not inject_neg:
        # the dropout masks and negative pairs differ from replay to replay: the loss is not a constant
        assert len(set(losses)) == len(losses)


def test_entity_level_loop_with_recorded_steps_trains(cuda):
    from magic_amd.model.step import make_adam
    from magic_amd.model.train import entity_level_train
    _, mine = _twins(cuda, seed=6)
    graphs = []
    for i in range(2):
        n, e = 600 + 50 * i, 3000
        src, dst, nt, et = small_graph(n, e, N_DIM, E_DIM, seed=80 + i, hub_deg=100)
        graphs.append(to_graph(src, dst, n, nt, et, N_DIM, E_DIM, cuda))
    w0 = mine.encoder.gats[0].fc.weight.detach().clone()
    torch.manual_seed(0)
    entity_level_train(mine, lambda i: graphs[i], 2, make_adam(mine, 5e-3, 5e-4), 6, cuda, N_DIM, E_DIM,
                       progress=False, capture=True)
    assert not torch.equal(w0, mine.encoder.gats[0].fc.weight)
    assert all(bool(torch.isfinite(p).all()) for p in mine.parameters())


def test_fused_adam_equals_torch_adam(cuda):
    """magic_adam_step (one multi-tensor launch, step count on the device) against torch.optim.Adam in float64
    on the same gradients: 70 tensors (more than one launch holds) of ragged sizes from 1 element to several
    2048-element chunks, weight decay on, six steps; then the state the optimizer keeps and a second parameter
    group with its own hyper-parameters."""
    from magic_amd.optim import FusedAdam
    from magic_amd.utils.utils import create_optimizer
    gen = torch.Generator().manual_seed(5)
    sizes = [(1,), (7,), (2048,), (2049,), (64, 33), (300, 41), (5000,)] * 10
    base = [torch.randn(*s, generator=gen) for s in sizes]
    mine = [torch.nn.Parameter(b.clone().to(cuda)) for b in base]
    ref = [torch.nn.Parameter(b.clone().double()) for b in base]
    om = FusedAdam([dict(params=mine[:60]), dict(params=mine[60:], lr=3e-3, weight_decay=0.0, betas=(0.8, 0.99))],
                   lr=1e-3, weight_decay=5e-4)
    orf = torch.optim.Adam([dict(params=ref[:60]), dict(params=ref[60:], lr=3e-3, weight_decay=0.0, betas=(0.8, 0.99))],
                           lr=1e-3, weight_decay=5e-4)
    for step in range(6):
        for i, (pm, pr) in enumerate(zip(mine, ref)):
            if step == 2 and i == 5:
                pm.grad = pr.grad = None      # a parameter without a gradient is left alone for that step
                continue
            g = torch.randn(pm.shape, generator=gen) * (10.0 ** (i % 5 - 2))
            pm.grad, pr.grad = g.to(cuda), g.double()
        om.step()
        orf.step()
    for i, (pm, pr) in enumerate(zip(mine, ref)):
        # (parameter 5 skipped one step: it is counted on its own from then on, as torch counts per parameter)
        err = (pm.detach().cpu().double() - pr.detach()).abs().max() / pr.detach().abs().max().clamp_min(1e-3)
        assert float(err) < 2e-6, (i, float(err))
    st = om.state[mine[0]]
    assert int(st["step"]) == 6 and st["exp_avg"].shape == mine[0].shape and st["exp_avg_sq"].dtype == torch.float32
    assert om.state[mine[0]]["step"] is om.state[mine[1]]["step"] and om.state[mine[65]]["step"] is not st["step"]
    assert int(om.state[mine[5]]["step"]) == 5 and om.state[mine[5]]["step"] is not st["step"]
    sd = om.state_dict()
    om2 = FusedAdam([dict(params=mine[:60]), dict(params=mine[60:])], lr=1e-3)
    om2.load_state_dict(sd)
    assert om2.param_groups[1]["lr"] == 3e-3 and int(om2.state[mine[3]]["step"]) == 6
    assert om2.state[mine[3]]["step"] is om2.state[mine[4]]["step"] and int(om2.state[mine[5]]["step"]) == 5
    # stepping goes on after a load: from this class's own state dict (moved to the host, as torch.load with
    # map_location='cpu' leaves it) and from the state dict torch.optim.Adam wrote (float counts on the host)
    import copy
    host_sd = copy.deepcopy(sd)
    for v in host_sd["state"].values():
        for k2 in v:
            v[k2] = v[k2].cpu()
    om3 = FusedAdam([dict(params=mine[:60]), dict(params=mine[60:])], lr=1e-3)
    om3.load_state_dict(host_sd)
    mine_t = [torch.nn.Parameter(p_.detach().double().cpu().float().to(cuda)) for p_ in ref]
    om4 = FusedAdam([dict(params=mine_t[:60]), dict(params=mine_t[60:])], lr=1e-3)
    om4.load_state_dict(orf.state_dict())
    assert om4.state[mine_t[0]]["step"].dtype == torch.int64 and om4.state[mine_t[0]]["step"].is_cuda
    assert om4.state[mine_t[0]]["exp_avg"].dtype == torch.float32 and int(om4.state[mine_t[5]]["step"]) == 5
    for step in range(2):
        for pm, pt, pr in zip(mine, mine_t, ref):
            g = torch.randn(pm.shape, generator=gen)
            pm.grad, pt.grad, pr.grad = g.to(cuda), g.to(cuda), g.double()
        om3.step()
        om4.step()
        orf.step()
    for i, (pm, pt, pr) in enumerate(zip(mine, mine_t, ref)):
        scale = pr.detach().abs().max().clamp_min(1e-3)
        assert float((pm.detach().cpu().double() - pr.detach()).abs().max() / scale) < 3e-6, i
        assert float((pt.detach().cpu().double() - pr.detach()).abs().max() / scale) < 3e-6, i
    # a parameter whose first gradient arrives later starts its own count at 1
    late = torch.nn.Parameter(torch.ones(10, device=cuda))
    early = torch.nn.Parameter(torch.ones(10, device=cuda))
    ol = FusedAdam([late, early], lr=1e-2)
    for k in range(3):
        early.grad = torch.ones(10, device=cuda)
        late.grad = torch.ones(10, device=cuda) if k == 2 else None
        ol.step()
    assert int(ol.state[early]["step"]) == 3 and int(ol.state[late]["step"]) == 1
    assert abs(float(late[0]) - (1.0 - 1e-2)) < 1e-6          # first Adam step: lr * sign(g)
    # the reference's factory hands out this optimizer for a model on the GPU and torch's own on the CPU
    lin = torch.nn.Linear(4, 3)
    assert type(create_optimizer("adam", lin, 1e-3, 5e-4)) is torch.optim.Adam
    assert isinstance(create_optimizer("adam", lin.to(cuda), 1e-3, 5e-4), FusedAdam)
    cpu_p = torch.nn.Parameter(torch.zeros(3))
    cpu_p.grad = torch.ones(3)
    with pytest.raises(Exception, match="GPU"):
        FusedAdam([cpu_p], lr=1e-3).step()
