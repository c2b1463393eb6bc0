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
the ranks, all 684,107 test rows per step: standardise, mean_distance search, test search, score).

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
BF16_MFMA_PEAK_TF = 2500.0  # MI355X dense bf16 matrix peak (without structured sparsity)


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
    p.add_argument("--knn-queries", type=int, default=684107)
    p.add_argument("--knn-steps", type=int, default=2)
    p.add_argument("--no-knn", action="store_true")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--dry-run", action="store_true",
                   help="launch plumbing only: no GPU work, rank 0 prints n_gpus and the timing of a no-op step")
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


# --------------------------------------------------------------------------- byte model
CACHE_BYTES = 256 * 2 ** 20   # Infinity Cache: a gathered table up to this size is fetched from HBM once


def sweep_bytes(kind, N, E_seg, E_all, H, C, res):
    """Byte counts of ONE launch of a sparse sweep of one GAT layer (fp32 values, int32 ids).
    Returns (per_use, compulsory, gathered, table):
      per_use     every tensor priced once per logical use (a gathered row once per edge);
      compulsory  every tensor priced once (a gathered table counts N rows, not E, while it fits
                  the 256 MiB Infinity Cache): the least that has to cross the HBM pins;
      gathered    bytes of node rows gathered per edge (part of per_use);
      table       size of the gathered node table.
    ``res``: None | "linear" | "identity" (residual operand of the forward epilogue)."""
    row = 4 * C
    table = N * row

    def once(per_edge_bytes):           # gathered table priced once when it stays cache resident
        return min(table, per_edge_bytes) if table <= CACHE_BYTES else per_edge_bytes
    if kind == "fwd":
        # per edge: src id, edge type, gathered hs row, gathered source scores (H floats of the [N,2H] score rows)
        # per node: segment pointer, walk order, et, (et, log-sum-exp) statistics out, output row, residual row
        #           (+ 1-byte keep mask per element when the residual is the dropped layer input)
        g = E_seg * row
        eh, eh_tab = E_seg * 4 * H, N * 4 * H
        stream = E_seg * 8 + N * (8 + 12 * H + row + (0 if res is None else row) + (C if res == "identity" else 0))
        return stream + g + eh, stream + once(g) + min(eh, eh_tab), g, table
    if kind == "bwd_src":
        # per edge: dst id, edge type, gathered gradient row, gathered (et, lse) of the destination,
        #           (a, t) pair out
        # per node: segment pointer, walk order, own hs row, own eh, d_hs row out
        g = E_seg * row
        stat, stat_tab = E_seg * 8 * H, N * 8 * H
        stream = E_seg * (8 + 8 * H) + N * (8 + 4 * H + 2 * row)
        return stream + g + stat, stream + once(g) + min(stat, stat_tab), g, table
    if kind == "bwd_dst":
        # per edge: CSR position, gathered (a, t) pair (every pair exactly once)
        # per node: segment pointer, walk order, delta and d_et out
        b = E_seg * (4 + 8 * H) + N * (8 + 8 * H)
        return b, b, 0, 0
    if kind == "bwd_deh":
        # per edge: (a, t) pair, dst id, gathered delta of the destination, d_logit out
        # per node: segment pointer, d_eh out
        stream = E_seg * (4 + 12 * H) + N * (4 + 4 * H)
        return stream + E_seg * 4 * H, stream + min(E_seg * 4 * H, N * 4 * H), 0, 0
    raise ValueError(kind)


def gather_ceiling_gbs(table_bytes):
    """Measured chip-wide rate of uniformly random row gathers served from the caches
    (MI355X micro-architecture guide, 'Indexed rows: gather into LDS'): 8.6 TB/s from a 38 MB
    table, 7.4-7.9 TB/s from a 151 MB one, 6.0-6.1 TB/s from HBM."""
    if table_bytes <= 64e6:
        return 8600.0
    if table_bytes <= CACHE_BYTES:
        return 7900.0
    return 6100.0


def _stage_collectives_through_host():
    """Rehearsal mode only: gloo on device tensors of two ranks sharing one GPU does not progress
    here, so the collectives this file and the package use take a host copy, exchange it, copy back."""
    real = {k: getattr(dist, k) for k in ("all_reduce", "broadcast", "all_gather_into_tensor", "all_to_all_single")}

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
    def all_to_all_single(out, t, group=None):
        ho, h = out.cpu(), t.cpu()
        real["all_to_all_single"](ho, h, group=group)
        out.copy_(ho)
    dist.all_reduce, dist.broadcast, dist.all_gather_into_tensor = all_reduce, broadcast, all_gather_into_tensor
    dist.all_to_all_single = all_to_all_single


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


def roofline_rows(prof, struct, workload, steps, n, n_dim, hidden, layers, n_pairs):
    """One row per kernel template instance of the step, most expensive first.

    HBM-bound sparse sweeps carry two byte counts per launch (``sweep_bytes``): ``per_use`` and
    ``compulsory``.  ``achieved`` / ``frac`` are the compulsory bytes over the launch time against
    the 8 TB/s HBM peak -- the share of the HBM roofline the launch really uses.  A sweep whose
    gathered node table fits the Infinity Cache is labelled ``"gather-cache"``: its per-edge row
    gathers are served on-die, so their rate (``gather.rate``) is compared with the measured
    cache-resident gather ceiling instead.  ``traffic`` = fabric bytes per launch from the PMC
    counters (``profiles/traffic.json``), ``wasted`` = traffic / compulsory."""
    from magic_amd.graph import LIGHT_MAX
    from magic_amd.model.gat import _pad_head_dim
    deg_in = (struct.in_ptr[1:] - struct.in_ptr[:-1]).long()
    deg_out = (struct.out_ptr[1:] - struct.out_ptr[:-1]).long()
    E = int(struct.num_edges)
    e_in = int(deg_in[deg_in <= LIGHT_MAX].sum())
    e_out = int(deg_out[deg_out <= LIGHT_MAX].sum())
    # (H, C, residual kind) of every GAT layer: encoder layers, then the decoder (head padded)
    lay = [(4, hidden, "linear" if (n_dim if l == 0 else hidden) != hidden else "identity")
           for l in range(layers)] + [(1, _pad_head_dim(n_dim), None)]
    sparse = {}     # template name -> list of (per_use, compulsory, gathered, table), one per launch of a step
    for H, C, r in lay:
        t = f"<{H},{C // (4 * H)}>"
        for base, kind, e_seg in (("gat_fwd_light", "fwd", e_in),
                                  ("gat_bwd_src_light", "bwd_src", e_out), ("gat_bwd_dst_light", "bwd_dst", e_in),
                                  ("gat_bwd_deh_light", "bwd_deh", e_out)):
            sparse.setdefault(base + t, []).append(sweep_bytes(kind, n, e_seg, E, H, C, r))
    dense = dense_work(n, n_dim, hidden, layers, n_pairs)
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        traffic_tab = json.load(open(tpath)).get(workload, {}) if os.path.exists(tpath) else {}
    except Exception:
        traffic_tab = {}
    kernels, rows = {}, []
    for name, (cnt, tot_ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
        kernels[name] = {"launches": cnt, "avg_us": round(tot_ms / cnt * 1e3, 2),
                         "total_ms_per_step": round(tot_ms / steps, 4)}
        per_step = max(cnt // steps, 1)
        sec = tot_ms / cnt * 1e-3               # average launch duration
        row = {"kernel": name, "avg_launch_us": round(sec * 1e6, 2), "launches_per_step": per_step,
               "ms_per_step": round(tot_ms / steps, 4)}
        if name in sparse:
            per_use, comp, gath, table = (sum(x[i] for x in sparse[name]) / len(sparse[name]) for i in range(4))
            traffic = traffic_tab.get(name)
            cached = 0 < table <= CACHE_BYTES and gath > 0
            row.update({"bound": "gather-cache" if cached else "hbm",
                        "achieved": round(comp / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(comp / sec / 1e9 / HBM_PEAK_GBS, 4),
                        "compulsory_bytes_per_launch": int(comp), "per_use_bytes_per_launch": int(per_use),
                        "per_use_rate_gbs": round(per_use / sec / 1e9, 1),
                        "traffic": traffic, "wasted": round(traffic / comp, 3) if traffic else None,
                        # fabric bytes against the per-use count: < 1 when gathers hit the XCD's L2, > 1 when random
                        # accesses move more than they use
                        "traffic_over_per_use": round(traffic / per_use, 3) if traffic else None})
            if gath > 0:
                ceil = gather_ceiling_gbs(table)
                row["gather"] = {"table_bytes": int(table), "gathered_bytes_per_launch": int(gath),
                                 "rate_gbs": round(gath / sec / 1e9, 1), "ceiling_gbs": ceil,
                                 "frac_of_ceiling": round(gath / sec / 1e9 / ceil, 4)}
        elif name in dense:
            fl, by = dense[name]
            fl, by = fl / per_step, by / per_step
            # The projections form every fp32 product from six bf16 matrix products (split operands), so their
            # matrix ceiling is the dense bf16 peak / 6; a launch is matrix-bound past the ridge of that ceiling.
            split_peak = BF16_MFMA_PEAK_TF / 6.0
            if fl / by > split_peak * 1e3 / HBM_PEAK_GBS:
                ach = fl / sec / 1e12
                row.update({"bound": "mfma", "achieved": round(ach, 1), "peak": round(split_peak, 1), "unit": "TFLOP/s",
                            "frac": round(ach / split_peak, 4), "algorithmic_flops_per_launch": int(fl),
                            "arithmetic": "fp32-grade products as 6 bf16 MFMA products (peak = dense bf16 peak / 6); "
                                          "the fp32 matrix instruction peaks at %.1f" % F32_MFMA_PEAK_TF,
                            "traffic": None})
            else:
                ach = by / sec / 1e9
                row.update({"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(ach / HBM_PEAK_GBS, 4), "compulsory_bytes_per_launch": int(by),
                            "traffic": traffic_tab.get(name)})
        else:
            continue
        rows.append(row)
    return rows, kernels


def gat_leg(args, workload, steps, warmup, world, rank, dev):
    """Times `steps` training steps of `workload`; returns the numbers of the JSON line."""
    from magic_amd import _lib, synth
    from magic_amd.model.autoencoder import GMAEModel, _negative_candidates
    from magic_amd.model.step import CapturedTrainStep, make_adam
    from magic_amd.parallel import FlatGradAllReduce, broadcast_parameters
    from magic_amd.utils.utils import create_optimizer, set_random_seed

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
    opt = make_adam(model, lr=1e-3, weight_decay=5e-4)      # utils/config.py:20-28 defaults

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
            # (mask permutation drawn on the device, as model/autoencoder.py:201 does for a model on the GPU)
            e2e_model = GMAEModel(n_dim, e_dim, hidden, layers, 4, "prelu", 0.1, 0.2, True, "BatchNorm", 0.5,
                                  "sce", 3, mask_rng="device").to(dev)
            e2e_opt = create_optimizer("adam", e2e_model, 1e-3, 5e-4)   # train.py:82 with utils/config.py defaults
            pool = [(p_, 0) for p_ in parts]
            loader = [list(range(batch_graphs))]
            batch_level_train(e2e_model, pool, loader * 8, e2e_opt, 1, dev, n_dim, e_dim, progress=False)
            torch.cuda.synchronize()
            # two timed passes of 16 steps, the faster one reported: on some boxes a pass of this loop runs
            # 1.6 x slower than the next one with identical work (per-step device times, not the host's)
            e2e_ms = float("inf")
            for _ in range(2):
                t0 = time.perf_counter()
                batch_level_train(e2e_model, pool, loader * 16, e2e_opt, 1, dev, n_dim, e_dim, progress=False)
                torch.cuda.synchronize()
                e2e_ms = min(e2e_ms, (time.perf_counter() - t0) / 16 * 1e3)
            del e2e_model, e2e_opt
    E = g.num_edges()
    etype = g.edata["type"]
    g.edata["attr"] = torch.nn.functional.one_hot(etype, e_dim).float()
    g.edge_types_csc()
    total = warmup + 3 * steps + 4
    gen = torch.Generator(device=dev).manual_seed(100 + rank)
    thr = min(10000, n)
    perms = [torch.randperm(n, device=dev, generator=gen) for _ in range(total)]
    pos = [torch.randint(0, E, (thr,), device=dev, generator=gen) for _ in range(total)]
    negs = [_negative_candidates(g, thr, generator=gen) for _ in range(total)]
    counter = [0]
    # The product's step harness: fresh one-hot node attributes, compute_loss forward, backward,
    # gradient all-reduce (N > 1), Adam.  The timed step is the eager one; the same step recorded as
    # a HIP graph and replayed is timed beside it (config.recorded_ms_per_step).
    exchange = flat if world > 1 else None
    eager = CapturedTrainStep(model, opt, g, n_dim, grad_exchange=exchange)
    cap = CapturedTrainStep(model, opt, g, n_dim, grad_exchange=exchange, capture=True, warmup=2)

    def make_step(harness):
        def step(_i):
            k = counter[0] % total
            counter[0] += 1
            return harness(perm=perms[k], pos_idx=pos[k], neg_pairs=negs[k][:2], neg_valid=negs[k][2])
        return step
    step_rec, step_eager = make_step(cap), make_step(eager)

    for i in range(warmup):
        step_eager(i)
    dt = timed(step_eager, steps, world)
    ms_per_step = dt / steps * 1e3
    edges_per_s = E * world * steps / dt
    for i in range(4):                       # the harness's eager warm-up calls, its recording, a replay
        step_rec(i)
    eager_ms = timed(step_rec, steps, world) / steps * 1e3     # (reported as recorded_ms_per_step)
    # ---- per-kernel timing over a further timed region (HIP events on the launch stream)
    _lib.prof_reset()
    _lib.prof_enable(True)
    dt_prof = timed(step_eager, steps, world)
    _lib.prof_enable(False)
    prof = _lib.prof_read(128)
    _lib.prof_reset()

    roofline = None
    others = []
    rows = []
    kernels = {}
    if rank == 0:
        rows, kernels = roofline_rows(prof, g.structure, workload, steps, n, n_dim, hidden, layers, 2 * thr)
        if rows:
            roofline = dict(rows[0], profiled_ms_per_step=round(dt_prof / steps * 1e3, 3))
            others = rows[1:4]
    desc = {"theia": "E3-THEIA-shaped full graph", "streamspot_batch": "StreamSpot batch-level training, batch of "
            "12 graphs"}.get(workload, workload)
    return {"ms_per_step": ms_per_step, "edges_per_s": edges_per_s, "roofline": roofline, "roofline_next": others,
            "roofline_rows": rows, "kernels": kernels, "batch_assembly_ms": assembly_ms, "end_to_end_ms": e2e_ms,
            "eager_ms": eager_ms,
            "workload": f"{desc} per GPU ({workload}): N={n} E={E} n_dim={n_dim} e_dim={e_dim} hidden={hidden} "
                        f"layers={layers}; one-hot attributes + compute_loss fwd + bwd + Adam, dropout on, mask perms "
                        "and sampled pairs pre-generated on the device",
            "dims": (n_dim, e_dim, hidden, layers)}


def spawn_ranks(args):
    """``python bench.py --gpus N`` without a launcher: start the N ranks as children (one
    ``torch.distributed.run`` process group on this node) BEFORE this process touches the GPU, relay
    their output and exit with their status.  A process that has initialised the GPU is never replaced."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def dry_run(args, world, rank):
    """Launch plumbing only (no GPU, no kernels): process group over gloo, the barrier-bracketed timing
    of ``timed`` with a no-op step, one JSON line from rank 0.  Lets a CPU-only machine check that
    ``--gpus N`` really runs N ranks."""
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001)
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        cnt = torch.ones(1)
        dist.all_reduce(cnt)
        assert int(cnt.item()) == world
    if rank == 0:
        print(json.dumps({"metric": "masked-GAT fwd+bwd edges/sec", "value": None, "unit": "edges/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
                          "dry_run": True}))
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a different GPU count")
    if args.dry_run:
        return dry_run(args, world, rank)
    # rehearsal of the multi-rank path on a one-GPU box: MAGIC_BENCH_REHEARSAL=1 puts every rank on
    # device 0 and exchanges over gloo (RCCL refuses two ranks on one device); numbers are meaningless
    rehearsal = os.environ.get("MAGIC_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
        import faulthandler
        faulthandler.dump_traceback_later(75, exit=True)   # a stuck rehearsal says where
    if not rehearsal and local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible)")
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
                  "ms_per_step": round(el["ms_per_step"], 4),
                  "config": {"workload": el["workload"], "recorded_ms_per_step": round(el["eager_ms"], 4)},
                  "north_star_target": True,
                  "roofline": el["roofline"], "roofline_next": el["roofline_next"],
                  "roofline_rows": el["roofline_rows"], "kernels": el["kernels"]}
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
                       "recorded_ms_per_step": round(leg["eager_ms"], 4),
                       "end_to_end_ms_per_step": None if leg["end_to_end_ms"] is None
                       else round(leg["end_to_end_ms"], 3)},
            "roofline": leg["roofline"], "roofline_next": leg["roofline_next"], "cpu_baseline": cpu_baseline,
            # peers of the top-level leg: "entity_level" is the single-GPU target the north star names
            # (E3-THEIA-shaped full graph); the top-level leg is the one that shards over GPUs
            "north_star_target": "entity_level (single GPU) + knn; the top-level leg is the data-parallel one",
            "entity_level": entity, "knn": knn, "roofline_rows": leg["roofline_rows"], "kernels": leg["kernels"],
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def knn_leg(args, world, rank, dev):
    """KNN outlier scoring at the E3-Trace shape (model/eval.py:282-323): one step = standardise train
    and test embeddings by the train columns' mean / std, the mean k-NN distance of 50,000 train rows
    (``mean_distance``), the k-NN search of ALL test rows against the full train set, per-row mean and
    the division by ``mean_distance``.  Everything is resident on the device; the train rows are
    sharded over the ranks, the test rows replicated.  The 50,000 train-row ids (the reference draws
    them with random.shuffle on the host) are drawn before the timed region."""
    from magic_amd import _lib
    from magic_amd.knn import ShardedNearestNeighbors, NearestNeighbors
    from magic_amd.model.eval import standardise_device
    from magic_amd.parallel import gather_rows
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
    n_sample = min(50000, nt)
    # the sampled train rows must be the same on every rank: rank r contributes those of its shard
    sample = torch.randperm(nt, generator=torch.Generator().manual_seed(7))[:n_sample]
    mine = sample[(sample >= lo) & (sample < hi)] - lo
    mine_dev = mine.to(dev)
    counts = [int(((sample >= min(r * per, nt)) & (sample < min((r + 1) * per, nt))).sum()) for r in range(world)]

    def step(_i):
        xt, xq = standardise_device(train, query, sharded=world > 1)
        if world > 1:
            nn_ = ShardedNearestNeighbors(k, device=dev)
            nn_.fit(x_local=xt, offset=lo)
            # 50,000 x 64 floats in all; the ranks hold different numbers of the sampled rows
            sample_rows = gather_rows(xt[mine_dev], counts)
        else:
            nn_ = NearestNeighbors(k, device=dev).fit(xt)
            sample_rows = xt[mine_dev]
        mean_distance = nn_.mean_distances_device(sample_rows, k).mean()
        return nn_.mean_distances_device(xq, k) / mean_distance   # model/eval.py:323
    score = step(0)
    assert score.shape == (nq,) and bool(torch.isfinite(score).all())
    _lib.prof_reset()
    _lib.prof_enable(True)
    dt = timed(step, args.knn_steps, world)
    _lib.prof_enable(False)
    prof = _lib.prof_read(128)
    _lib.prof_reset()
    res = {"metric": "KNN-score queries/sec", "value": round(nq * args.knn_steps / dt, 1), "unit": "queries/s",
           "ms_per_step": round(dt / args.knn_steps * 1e3, 3), "dtype": "f64", "steps": args.knn_steps,
           "config": {"workload": f"E3-Trace-shaped outlier scoring: train {nt} x {d} f32 sharded over {world} rank(s), "
                                  f"all {nq} test rows per step, k={k}; standardise + mean_distance search of "
                                  f"{n_sample} train rows + test search + row mean + score, all on the device"}}
    # dominant kernel: the matrix-core filter of the two-stage search (every query-train pair goes through it
    # once; the float64 stages only touch the k' candidates and the queries the proof rejects).  It forms one
    # float32-grade product from six bf16 matrix products, so its roofline is the dense bf16 peak / 6.
    if "knn_filter_bf16x6_mfma" in prof:
        name, peak, arith = "knn_filter_bf16x6_mfma", BF16_MFMA_PEAK_TF / 6.0, \
            "split-bf16 matrix-core filter (6 bf16 products per float32-grade product), f64 refine + proof, " \
            "f64 sweep for rejected queries"
    else:
        name, peak, arith = "knn_topk_f64_mfma", F64_MFMA_PEAK_TF, "f64 matrix cores"
    if name in prof:
        cnt, tot_ms = prof[name]
        flops = 2.0 * (nq + n_sample) * (hi - lo) * d * args.knn_steps      # both searches of every step
        ach = flops / (tot_ms * 1e-3) / 1e12
        res["roofline"] = {"kernel": name, "bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1),
                           "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                           "launches_per_step": cnt // args.knn_steps,
                           "kernel_ms_per_step": round(tot_ms / args.knn_steps, 3), "arithmetic": arith}
        res["kernels"] = {kname: {"launches": c, "total_ms_per_step": round(t / args.knn_steps, 3)}
                          for kname, (c, t) in sorted(prof.items(), key=lambda kv: -kv[1][1])}
    return res


def host_cores():
    """(threads to use, description).  The threads this process may run on (its CPU affinity), capped
    by OMP_NUM_THREADS when the box sets it (a one-GPU box owns a share of the host's CPUs)."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cap = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    cores = min(avail, cap) if cap > 0 else avail
    model = "unknown CPU"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return cores, f"{model}; {os.cpu_count()} logical CPUs visible, {avail} in this process's affinity mask, " \
                  f"OMP_NUM_THREADS={os.environ.get('OMP_NUM_THREADS', 'unset')}"


def cpu_gat_baseline(args, n_dim, e_dim, hidden, layers):
    """Torch-CPU restatement of the reference's DGL path (the oracle), same recipe and the same
    workload as the GPU step: the full batch of 12 StreamSpot-sized graphs (a 1/10 scale graph of
    the same shape family for the full-graph workloads), a bounded number of steps."""
    from magic_amd import synth
    from oracle import gat_oracle as O
    cores, cpu_desc = host_cores()
    torch.set_num_threads(cores)
    if args.workload == "streamspot_batch":
        n0, e0 = synth.SHAPES["streamspot"][:2]
        n1, e1 = max(64, int(n0 * args.scale)), max(64, int(e0 * args.scale))
        parts = [synth.make_graph(n1, e1, n_dim, e_dim, seed=1002 + i) for i in range(12)]
        src = torch.cat([p_.edges()[0] + i * n1 for i, p_ in enumerate(parts)])
        dst = torch.cat([p_.edges()[1] + i * n1 for i, p_ in enumerate(parts)])
        ntype = torch.cat([p_.ndata["type"] for p_ in parts])
        e_attr = torch.cat([p_.edata["attr"] for p_ in parts])
        n, e = 12 * n1, int(src.numel())
        what = f"the full batch of 12 StreamSpot-sized graphs (N={n}, E={e}), as the GPU step"
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
    t0 = time.perf_counter()
    step()                                   # warm-up step (allocator, thread pool)
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    reps = 0
    while reps < 1 or (time.perf_counter() - t0 + first < 25 and reps < 50):
        step()
        reps += 1
    dt = time.perf_counter() - t0
    return {"value": round(e * reps / dt, 1), "unit": "edges/s", "cores": cores, "kind": "port",
            "cpu": cpu_desc,
            "sample": f"torch-CPU restatement of the DGL path (oracle/gat_oracle.py), same recipe on {what}, "
                      f"{reps} timed step(s) after one warm-up step, {cores} threads"}


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
    cores, cpu_desc = host_cores()
    try:
        from sklearn.neighbors import NearestNeighbors
        from threadpoolctl import threadpool_limits
        threadpool_limits(cores)             # the brute-force search runs on BLAS / OpenMP threads
        nbrs = NearestNeighbors(n_neighbors=k, n_jobs=cores).fit(train)
        t0 = time.perf_counter()
        nbrs.kneighbors(query[:200], n_neighbors=k)
        t1 = time.perf_counter()
        if (t1 - t0) * 10 > 40:
            nq = 400
        t0 = time.perf_counter()
        nbrs.kneighbors(query[:nq], n_neighbors=k)
        dt = time.perf_counter() - t0
        kind, what = "reference", f"sklearn.neighbors.NearestNeighbors(n_jobs={cores}).kneighbors"
    except ImportError:
        from oracle import knn_oracle
        nq = 200
        t0 = time.perf_counter()
        knn_oracle.exact_topk(train, query[:nq], k)
        dt = time.perf_counter() - t0
        kind, what = "port", "numpy float64 brute force (oracle/knn_oracle.py)"
    return {"value": round(nq / dt, 1), "unit": "queries/s", "cores": cores, "kind": kind, "cpu": cpu_desc,
            "sample": f"{what}: {nq} queries against the full {nt} x {d} train set, k={k}"}


if __name__ == "__main__":
    main()
