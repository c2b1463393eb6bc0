#!/usr/bin/env python3
"""Benchmark of the MAGIC hot path on MI355X.

One "step" = one training step of the masked-GAT auto-encoder (``compute_loss`` forward,
backward, Adam update; dropout on) on one batch, inputs resident in HBM.  The default workload
is StreamSpot batch-level training: a batch of 12 StreamSpot-sized graphs per GPU (union graph
N=96,000 nodes, E=2,040,000 edges, n_dim 8, e_dim 20, hidden 256, 4 layers), every rank on its
own batch, the flat gradient buffer all-reduced once per step (weak scaling).
``value`` = edges of all ranks / step time.

Further legs in the same JSON line: ``"entity_level"`` -- the same step on one E3-THEIA-shaped
full graph (N=700,000, E=3,000,000, hidden 64, 3 layers; single GPU runs only) -- and ``"knn"``
-- the KNN outlier scorer (k=10, d=64, E3-Trace-sized train set of 2,604,530 rows sharded over
the ranks, a batch of queries per step).

Usage: python bench.py [--gpus N] [--steps K] [--warmup W]
       (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec
F64_MFMA_PEAK_TF = 78.6    # MI355X fp64 matrix spec
F32_MFMA_PEAK_TF = 157.3   # MI355X fp32 matrix spec


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--workload", default="streamspot_batch",
                   help="streamspot_batch (12 graphs per step) or a synthetic shape name (magic_amd.synth.SHAPES)")
    p.add_argument("--no-entity", action="store_true", help="skip the E3-THEIA-shaped full-graph leg")
    p.add_argument("--scale", type=float, default=1.0, help="shrink the graph (debugging only)")
    p.add_argument("--knn-train", type=int, default=2604530)
    p.add_argument("--knn-queries", type=int, default=65536)
    p.add_argument("--knn-steps", type=int, default=3)
    p.add_argument("--no-knn", action="store_true")
    p.add_argument("--no-cpu-baseline", action="store_true")
    return p.parse_args()


def timed(fn, steps, world):
    """barrier + synchronize on both sides, max over ranks (seconds for `steps` calls)."""
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


# --------------------------------------------------------------------------- algorithmic bytes
def kernel_bytes(kind, N, E_seg, H, C, res):
    """Algorithmic HBM bytes of one launch of a sparse sweep over the light segments of one layer
    (fp32 values, int32 ids, every tensor counted once).  ``res``: None | "linear" | "identity"."""
    if kind == "fwd":
        # per edge: src id, logit in, attention weight out, gathered hs row
        # per node: segment pointer, output row, residual row (+ 1-byte keep mask when it is the dropped input)
        per_node = 4 + 4 * C + (0 if res is None else 4 * C) + (C if res == "identity" else 0)
        return E_seg * (4 + 8 * H + 4 * C) + N * per_node
    if kind == "bwd_src":
        # per edge: dst id, CSC position, gathered gradient row, gathered a, t out
        # per node: segment pointer, own hs row, d_hs row out
        return E_seg * (8 + 8 * H + 4 * C) + N * (4 + 8 * C)
    if kind == "bwd_dst":
        # per edge: a, t in, d_logit out ; per node: segment pointer, d_et out
        return E_seg * 12 * H + N * (4 + 4 * H)
    raise ValueError(kind)


def _stage_collectives_through_host():
    """Rehearsal mode only: gloo on device tensors of two ranks sharing one GPU does not progress
    here, so the collectives this file and the package use take a host copy, exchange it, copy back."""
    real = {k: getattr(dist, k) for k in ("all_reduce", "broadcast", "all_gather_into_tensor")}

    def all_reduce(t, op=dist.ReduceOp.SUM, group=None):
        h = t.cpu()
        real["all_reduce"](h, op=op, group=group)
        t.copy_(h)

    def broadcast(t, src=0, group=None):
        h = t.cpu()
        real["broadcast"](h, src=src, group=group)
        t.copy_(h)

    def all_gather_into_tensor(out, t, group=None):
        ho, h = out.cpu(), t.cpu()
        real["all_gather_into_tensor"](ho, h, group=group)
        out.copy_(ho)
    dist.all_reduce, dist.broadcast, dist.all_gather_into_tensor = all_reduce, broadcast, all_gather_into_tensor


def dense_work(n, n_dim, hidden, layers, n_pairs):
    """Algorithmic flops (2 per multiply-add) and HBM bytes (fp32 operands and results once, epilogue
    operands not counted) of one step's projection launches, per kernel name."""
    from magic_amd.model.gat import _pad_head_dim
    H, C, L = 4, hidden, layers
    dp = _pad_head_dim(n_dim)
    w = {k: [0.0, 0.0] for k in ("proj_fwd_mfma_f32", "proj_bwd_input_mfma_f32", "proj_bwd_weight_mfma_f32")}

    def add(name, rows, k, m):
        w[name][0] += 2.0 * rows * k * m
        w[name][1] += 4.0 * rows * (k + m)
    for l in range(L):
        f = n_dim if l == 0 else C
        m = C + 2 * H + (C if f != C else 0)       # hs | eh, et | linear residual
        add("proj_fwd_mfma_f32", n, f, m)
        add("proj_bwd_weight_mfma_f32", n, 2 * f if l == 0 else f, m)  # layer 0 carries the mask-token indicator columns
        if l > 0:
            add("proj_bwd_input_mfma_f32", n, m, f)
    add("proj_fwd_mfma_f32", n, L * C, C)          # encoder_to_decoder
    add("proj_fwd_mfma_f32", n, C, dp + 2)         # decoder
    add("proj_bwd_input_mfma_f32", n, C, L * C)
    add("proj_bwd_input_mfma_f32", n, dp + 2, C)
    add("proj_bwd_weight_mfma_f32", n, L * C, C)
    add("proj_bwd_weight_mfma_f32", n, C, dp + 2)
    add("proj_bwd_weight_mfma_f32", n_pairs, 2 * L * C, C)   # edge-reconstruction MLP
    add("proj_bwd_weight_mfma_f32", n_pairs, C, 1)
    return w


def gat_leg(args, workload, steps, warmup, world, rank, dev):
    """Times `steps` training steps of `workload`; returns the numbers of the JSON line."""
    from magic_amd import _lib, synth
    from magic_amd.model.autoencoder import GMAEModel, global_uniform_negative_sampling
    from magic_amd.parallel import FlatGradAllReduce, broadcast_parameters
    from magic_amd.utils.utils import set_random_seed

    batch_graphs = 1
    shape_name = workload
    if workload == "streamspot_batch":   # one batch-level training step: 12 StreamSpot-sized graphs
        shape_name, batch_graphs = "streamspot", 12
    n, e, n_dim, e_dim, hidden, layers = synth.SHAPES[shape_name]
    n, e = max(64, int(n * args.scale)), max(64, int(e * args.scale))
    set_random_seed(0)
    model = GMAEModel(n_dim, e_dim, hidden, layers, 4, "prelu", 0.1, 0.2, True, "BatchNorm", 0.5, "sce", 3).to(dev)
    broadcast_parameters(model)
    model.train()
    flat = FlatGradAllReduce(model.parameters())
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-4, fused=True)

    assembly_ms = e2e_ms = None
    if batch_graphs == 1:
        g = synth.make_graph(n, e, n_dim, e_dim, seed=1004 + rank, attrs=False).to(dev)
        g.structure.ensure_csr()
    else:
        import magic_amd
        parts = [synth.make_graph(n, e, n_dim, e_dim, seed=1002 + 1000 * rank + i, attrs=False).to(dev)
                 for i in range(batch_graphs)]
        for p_ in parts:
            p_.structure.ensure_csr()
        g = magic_amd.batch(parts)          # model/train.py:39
        # reported beside the step time: assembling the batch from device-resident graphs with
        # pre-built CSR (offset-add concatenation, segment plan)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            g2 = magic_amd.batch(parts)
            g2.structure.ensure_csr()
        torch.cuda.synchronize()
        assembly_ms = (time.perf_counter() - t0) / 3 * 1e3
        del g2
        n = g.num_nodes()
        if world == 1:
            # the reference's loop itself (model/train.py:18-48): per step transform_graph on the 12
            # graphs, batch, masking / sampling inside the model, backward, Adam -- nothing pre-generated
            from magic_amd.model.train import batch_level_train
            e2e_model = GMAEModel(n_dim, e_dim, hidden, layers, 4, "prelu", 0.1, 0.2, True, "BatchNorm", 0.5,
                                  "sce", 3).to(dev)
            e2e_opt = torch.optim.Adam(e2e_model.parameters(), lr=1e-3, weight_decay=5e-4, fused=True)
            pool = [(p_, 0) for p_ in parts]
            loader = [list(range(batch_graphs))]
            batch_level_train(e2e_model, pool, loader, e2e_opt, 2, dev, n_dim, e_dim, progress=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            batch_level_train(e2e_model, pool, loader * 8, e2e_opt, 1, dev, n_dim, e_dim, progress=False)
            torch.cuda.synchronize()
            e2e_ms = (time.perf_counter() - t0) / 8 * 1e3
            del e2e_model, e2e_opt
    E = g.num_edges()
    ntype, etype = g.ndata["type"], g.edata["type"]
    g.edata["attr"] = torch.nn.functional.one_hot(etype, e_dim).float()
    g.edge_types_csc()
    total = warmup + 2 * steps + 2
    gen = torch.Generator(device=dev).manual_seed(100 + rank)
    thr = min(10000, n)
    perms = [torch.randperm(n, device=dev, generator=gen) for _ in range(total)]
    pos = [torch.randint(0, E, (thr,), device=dev, generator=gen) for _ in range(total)]
    negs = [global_uniform_negative_sampling(g, thr, generator=gen) for _ in range(total)]
    counter = [0]

    def step(_i):
        k = counter[0] % total
        counter[0] += 1
        # fresh one-hot node attributes every step, as the reference's loaders produce them
        g.ndata["attr"] = torch.nn.functional.one_hot(ntype, n_dim).float()
        loss = model.compute_loss(g, perm=perms[k], pos_idx=pos[k], neg_pairs=negs[k])
        flat.zero()
        loss.backward()
        flat.allreduce()
        opt.step()
        return loss

    for i in range(warmup):
        step(i)
    dt = timed(step, steps, world)
    ms_per_step = dt / steps * 1e3
    edges_per_s = E * world * steps / dt

    # ---- per-kernel timing over a second timed region (HIP events on the launch stream)
    _lib.prof_reset()
    _lib.prof_enable(True)
    dt_prof = timed(step, steps, world)
    _lib.prof_enable(False)
    prof = _lib.prof_read(64)
    _lib.prof_reset()

    roofline = None
    others = []
    kernels = {}
    if rank == 0:
        s = g.structure
        deg_in = (s.in_ptr[1:] - s.in_ptr[:-1]).long()
        deg_out = (s.out_ptr[1:] - s.out_ptr[:-1]).long()
        from magic_amd.graph import LIGHT_MAX
        e_light_in = int(deg_in[deg_in <= LIGHT_MAX].sum())
        e_light_out = int(deg_out[deg_out <= LIGHT_MAX].sum())

        from magic_amd.model.gat import _pad_head_dim
        # (H, C, residual kind) of every GAT layer: encoder layers, then the decoder (head padded)
        lay = [(4, hidden, "linear" if (n_dim if l == 0 else hidden) != hidden else "identity")
               for l in range(layers)] + [(1, _pad_head_dim(n_dim), None)]

        def step_bytes(kind, e_seg):   # all launches of one step
            return sum(kernel_bytes(kind, n, e_seg, H, C, r) for H, C, r in lay)
        algo = {"gat_fwd_light": ("hbm", step_bytes("fwd", e_light_in)),
                "gat_bwd_src_light": ("hbm", step_bytes("bwd_src", e_light_out)),
                "gat_bwd_dst_light": ("hbm", step_bytes("bwd_dst", e_light_in))}
        # a projection is matrix-bound when its flops per byte exceed the ridge of the fp32 matrix peak
        for name, (fl, by) in dense_work(n, n_dim, hidden, layers, 2 * thr).items():
            algo[name] = ("mfma", fl) if fl / by > F32_MFMA_PEAK_TF * 1e3 / HBM_PEAK_GBS else ("hbm", by)
        for name, (cnt, tot_ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
            kernels[name] = {"launches": cnt, "avg_us": round(tot_ms / cnt * 1e3, 2),
                             "total_ms_per_step": round(tot_ms / steps, 4)}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        try:
            traffic_tab = json.load(open(tpath)).get(workload, {}) if os.path.exists(tpath) else {}
        except Exception:
            traffic_tab = {}
        ranked = [k for k in sorted(prof, key=lambda k: -prof[k][1]) if k in algo]
        for name in ranked[:4]:
            cnt, tot_ms = prof[name]
            per_step = cnt // steps
            bound, work = algo[name]            # algorithmic bytes / flops of one step's launches of this kernel
            sec = tot_ms / steps * 1e-3         # their summed duration
            if bound == "hbm":
                ach, peak, unit = work / sec / 1e9, HBM_PEAK_GBS, "GB/s"
            else:
                ach, peak, unit = work / sec / 1e12, F32_MFMA_PEAK_TF, "TFLOP/s"
            r = {"kernel": name, "bound": bound, "achieved": round(ach, 1), "peak": peak, "unit": unit,
                 "frac": round(ach / peak, 4), "traffic": traffic_tab.get(name),
                 ("algorithmic_bytes_per_launch" if bound == "hbm" else "algorithmic_flops_per_launch"):
                     int(work / max(per_step, 1)),
                 "avg_launch_us": round(tot_ms / cnt * 1e3, 2), "launches_per_step": per_step}
            if roofline is None:
                r["profiled_ms_per_step"] = round(dt_prof / steps * 1e3, 3)
                roofline = r
            else:
                others.append(r)
    desc = {"theia": "E3-THEIA-shaped full graph", "streamspot_batch": "StreamSpot batch-level training, batch of "
            "12 graphs"}.get(workload, workload)
    return {"ms_per_step": ms_per_step, "edges_per_s": edges_per_s, "roofline": roofline, "roofline_next": others,
            "kernels": kernels, "batch_assembly_ms": assembly_ms, "end_to_end_ms": e2e_ms,
            "workload": f"{desc} per GPU ({workload}): N={n} E={E} n_dim={n_dim} e_dim={e_dim} hidden={hidden} "
                        f"layers={layers}; compute_loss fwd + bwd + Adam, dropout on, mask perms and sampled pairs "
                        "pre-generated",
            "dims": (n_dim, e_dim, hidden, layers)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # rehearsal of the multi-rank path on a one-GPU box: MAGIC_BENCH_REHEARSAL=1 puts every rank on
    # device 0 and exchanges over gloo (RCCL refuses two ranks on one device); numbers are meaningless
    rehearsal = os.environ.get("MAGIC_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
        import faulthandler
        faulthandler.dump_traceback_later(75, exit=True)   # a stuck rehearsal says where
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
            _stage_collectives_through_host()
        else:
            dist.init_process_group("nccl", device_id=dev)

    leg = gat_leg(args, args.workload, args.steps, args.warmup, world, rank, dev)

    # ---- entity-level full graph (one GPU per graph in the reference's recipe: single-GPU runs only)
    entity = None
    if world == 1 and not args.no_entity and args.workload != "theia":
        torch.cuda.empty_cache()
        el = gat_leg(args, "theia", args.steps, args.warmup, world, rank, dev)
        entity = {"metric": "masked-GAT fwd+bwd edges/sec", "value": round(el["edges_per_s"], 1), "unit": "edges/s",
                  "ms_per_step": round(el["ms_per_step"], 4), "config": {"workload": el["workload"]},
                  "roofline": el["roofline"], "roofline_next": el["roofline_next"], "kernels": el["kernels"]}
        torch.cuda.empty_cache()

    # ---- KNN leg
    knn = None
    if not args.no_knn:
        knn = knn_leg(args, world, rank, dev)

    # ---- CPU baselines (rank 0, N=1 only)
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = cpu_gat_baseline(args, *leg["dims"])
        if knn is not None:
            knn["cpu_baseline"] = cpu_knn_baseline(args)

    if rank == 0:
        out = {
            "metric": "masked-GAT fwd+bwd edges/sec", "value": round(leg["edges_per_s"], 1), "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(leg["ms_per_step"], 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": leg["workload"],
                       "parallelism": f"dp{world} (one batch per rank, flat-buffer gradient all-reduce)"
                                      if world > 1 else "single GPU",
                       "batch_assembly_ms": None if leg["batch_assembly_ms"] is None
                       else round(leg["batch_assembly_ms"], 3),
                       "end_to_end_ms_per_step": None if leg["end_to_end_ms"] is None
                       else round(leg["end_to_end_ms"], 3)},
            "roofline": leg["roofline"], "roofline_next": leg["roofline_next"], "cpu_baseline": cpu_baseline,
            "entity_level": entity, "knn": knn, "kernels": leg["kernels"],
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def knn_leg(args, world, rank, dev):
    """KNN scoring: train rows sharded over ranks, the same query batch on every rank."""
    from magic_amd import _lib
    from magic_amd.knn import ShardedNearestNeighbors, NearestNeighbors
    nt, nq, d, k = args.knn_train, args.knn_queries, 64, 10
    if args.scale != 1.0:
        nt, nq = max(4096, int(nt * args.scale)), max(512, int(nq * args.scale))
    per = (nt + world - 1) // world
    lo, hi = min(rank * per, nt), min((rank + 1) * per, nt)
    gen = torch.Generator(device=dev).manual_seed(1005)
    centers = torch.randn(200, d, device=dev, generator=gen)
    gen_r = torch.Generator(device=dev).manual_seed(2000 + rank)
    train = centers[torch.randint(0, 200, (hi - lo,), device=dev, generator=gen_r)] + \
        0.05 * torch.randn(hi - lo, d, device=dev, generator=gen_r)
    gen_q = torch.Generator(device=dev).manual_seed(3000)
    query = centers[torch.randint(0, 200, (nq,), device=dev, generator=gen_q)] + \
        0.05 * torch.randn(nq, d, device=dev, generator=gen_q)
    if world > 1:
        nn_ = ShardedNearestNeighbors(k, device=dev)
        nn_.fit(x_local=train, offset=lo)

        def step(_i):
            dd, _ = nn_.kneighbors_device(query, k)
            return dd.mean(dim=1)
    else:
        nn_ = NearestNeighbors(k, device=dev).fit(train)

        def step(_i):
            dd, _ = nn_.kneighbors_device(query, k)
            return dd.mean(dim=1)
    step(0)
    _lib.prof_reset()
    _lib.prof_enable(True)
    dt = timed(step, args.knn_steps, world)
    _lib.prof_enable(False)
    prof = _lib.prof_read(64)
    _lib.prof_reset()
    res = {"metric": "KNN-score queries/sec", "value": round(nq * args.knn_steps / dt, 1), "unit": "queries/s",
           "ms_per_step": round(dt / args.knn_steps * 1e3, 3), "dtype": "f64",
           "config": {"workload": f"E3-Trace-shaped: train {nt} x {d} f32 sharded over {world} rank(s), "
                                  f"{nq} queries per step, k={k}"}}
    name = "knn_topk_f64_mfma"
    if name in prof:
        cnt, tot_ms = prof[name]
        flops = 2.0 * nq * (hi - lo) * d
        ach = flops / (tot_ms / cnt * 1e-3) / 1e12
        res["roofline"] = {"kernel": name, "bound": "mfma", "achieved": round(ach, 2), "peak": F64_MFMA_PEAK_TF,
                           "unit": "TFLOP/s", "frac": round(ach / F64_MFMA_PEAK_TF, 4), "traffic": None,
                           "avg_launch_ms": round(tot_ms / cnt, 3)}
    return res


def cpu_gat_baseline(args, n_dim, e_dim, hidden, layers):
    """Torch-CPU restatement of the reference's DGL path (the oracle), same recipe, on a bounded
    sample of the workload: a batch of 2 (of the 12) StreamSpot-sized graphs, or a 1/10 scale graph
    of the same shape family for the full-graph workloads."""
    from magic_amd import synth
    from oracle import gat_oracle as O
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(cores)
    if args.workload == "streamspot_batch":
        n0, e0 = synth.SHAPES["streamspot"][:2]
        n1, e1 = max(64, int(n0 * args.scale)), max(64, int(e0 * args.scale))
        parts = [synth.make_graph(n1, e1, n_dim, e_dim, seed=1002 + i) for i in range(2)]
        src = torch.cat([p_.edges()[0] + i * n1 for i, p_ in enumerate(parts)])
        dst = torch.cat([p_.edges()[1] + i * n1 for i, p_ in enumerate(parts)])
        ntype = torch.cat([p_.ndata["type"] for p_ in parts])
        e_attr = torch.cat([p_.edata["attr"] for p_ in parts])
        n, e = 2 * n1, int(src.numel())
        what = f"a batch of 2 StreamSpot-sized graphs (N={n}, E={e}; the GPU step takes 12)"
    else:
        n0, e0 = synth.SHAPES[args.workload][:2]
        scale = 0.1 * args.scale
        n, e = max(64, int(n0 * scale)), max(64, int(e0 * scale))
        g = synth.make_graph(n, e, n_dim, e_dim, seed=1004)
        src, dst = g.edges()
        ntype, e_attr = g.ndata["type"], g.edata["attr"]
        what = f"a {args.workload}-shaped graph at 1/10 scale (N={n}, E={e})"
    og = O.OGraph(src, dst, n)
    og.edata["attr"] = e_attr
    torch.manual_seed(0)
    model = O.build_model(n_dim, e_dim, hidden, layers)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-4)
    thr = min(10000, n)

    def step():
        og.ndata["attr"] = torch.nn.functional.one_hot(ntype, n_dim).float()
        neg = O.uniform_negative_pairs(og, thr, seed=1)
        loss = model.compute_loss(og, neg_pairs=neg, pos_idx=random.sample(range(e), thr))
        opt.zero_grad()
        loss.backward()
        opt.step()
    step()
    t0 = time.perf_counter()
    reps = 0
    while reps < 3 or (time.perf_counter() - t0 < 10 and reps < 50):
        step()
        reps += 1
    dt = time.perf_counter() - t0
    return {"value": round(e * reps / dt, 1), "unit": "edges/s", "cores": cores, "kind": "port",
            "sample": f"torch-CPU restatement of the DGL path (oracle/gat_oracle.py), same recipe on {what}, "
                      f"{reps} steps"}


def cpu_knn_baseline(args):
    """The reference's actual CPU path: sklearn NearestNeighbors(n_jobs=-1) (model/eval.py:294-311)
    on a query sub-sample against the full train set (brute force is linear in Q)."""
    nt, d, k = args.knn_train, 64, 10
    if args.scale != 1.0:
        nt = max(4096, int(nt * args.scale))
    rng = np.random.default_rng(1005)
    centers = rng.standard_normal((200, d)).astype(np.float32)
    train = centers[rng.integers(0, 200, nt)] + 0.05 * rng.standard_normal((nt, d)).astype(np.float32)
    nq = 2000
    query = centers[rng.integers(0, 200, nq)] + 0.05 * rng.standard_normal((nq, d)).astype(np.float32)
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    try:
        from sklearn.neighbors import NearestNeighbors
        nbrs = NearestNeighbors(n_neighbors=k, n_jobs=-1).fit(train)
        t0 = time.perf_counter()
        nbrs.kneighbors(query[:200], n_neighbors=k)
        t1 = time.perf_counter()
        if (t1 - t0) * 10 > 40:
            nq = 400
        t0 = time.perf_counter()
        nbrs.kneighbors(query[:nq], n_neighbors=k)
        dt = time.perf_counter() - t0
        kind, what = "reference", "sklearn.neighbors.NearestNeighbors(n_jobs=-1).kneighbors"
    except ImportError:
        from oracle import knn_oracle
        nq = 200
        t0 = time.perf_counter()
        knn_oracle.exact_topk(train, query[:nq], k)
        dt = time.perf_counter() - t0
        kind, what = "port", "numpy float64 brute force (oracle/knn_oracle.py)"
    return {"value": round(nq / dt, 1), "unit": "queries/s", "cores": cores, "kind": kind,
            "sample": f"{what}: {nq} queries against the full {nt} x {d} train set, k={k}"}


if __name__ == "__main__":
    main()
