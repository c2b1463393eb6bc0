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
    p.add_argument("--knn-only", action="store_true", help="only the KNN leg (prints its dictionary)")
    p.add_argument("--knn-k", type=int, default=10)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-configs", action="store_true", help="skip the small side configurations (c1, c3)")
    p.add_argument("--rows-file", default=None, help="where the per-kernel tables go (default: gpurun_out/ or here)")
    p.add_argument("--dry-run", action="store_true",
                   help="launch plumbing only: no GPU work, rank 0 prints n_gpus and the timing of a no-op step")
    return p.parse_args()


# MAGIC_BENCH_FORCE_DIST=1 makes a one-rank run take every branch of the multi-rank run (RCCL process group,
# barriers, gradient exchange, sharded KNN): the distributed code path on a one-GPU machine.
FORCE_DIST = os.environ.get("MAGIC_BENCH_FORCE_DIST") == "1"


def multi(world):
    return world > 1 or FORCE_DIST


def timed(fn, steps, world):
    """barrier + synchronize on both sides, max over ranks (seconds for `steps` calls)."""
    if multi(world):
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    torch.cuda.synchronize()
    if multi(world):
        dist.barrier()
    dt = time.perf_counter() - t0
    if multi(world):
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
    if kind == "fwd_coded":
        # first layer on code words: the projected rows come out of a table of a few dozen rows that stays on the
        # die, so per edge only the src id, the edge type and the source's 4-byte code word move; per node the
        # segment pointer, walk order, own code word, statistics out and the output row (the residual is a table row)
        b = E_seg * 12 + N * (12 + 8 * H + row)
        return b, E_seg * 8 + min(E_seg, N) * 4 + N * (12 + 8 * H + row), 0, 0
    if kind == "bwd_src_coded":
        # as bwd_src with the node's own row and scores replaced by its code word
        g = E_seg * row
        stat, stat_tab = E_seg * 8 * H, N * 8 * H
        stream = E_seg * (8 + 8 * H) + N * (12 + row)
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


def dense_launch(name, n, hidden, layers, n_pairs):
    """Algorithmic flops (2 per multiply-add) and HBM bytes (fp32 operands and results once, epilogue operands
    not counted) of ONE projection launch, from its profile name ``proj_fwd<K,M>`` / ``proj_bwd_input<K,M>`` /
    ``proj_bwd_weight<M,F>``: a ``[rows, K] x [K, M]`` product.  The launches of the edge-reconstruction head are
    the ones with the pair matrix's width (2 x layers x hidden) as one dimension: their row count is the number of
    sampled pairs, every other launch runs over the N nodes."""
    base, _, rest = name.partition("<")
    if base not in ("proj_fwd", "proj_bwd_input", "proj_bwd_weight") or not rest:
        return None
    a, b = (int(v) for v in rest.rstrip(">").split(","))
    rows = n_pairs if 2 * layers * hidden in (a, b) else n
    return 2.0 * rows * a * b, 4.0 * rows * (a + b), rows


def step_algorithmic(n, e, n_dim, hidden, layers, heads=4):
    """Algorithmic HBM bytes and dense flops of one whole fwd+bwd step (fp32 values, int32 ids, every tensor moved
    once per logical use): per GAT layer with input width F, H heads, C = H x D outputs
      edge phase  fwd E(8 + 8H + 4C) + N(4 + 4H + 4C),  bwd E(20 + 16H + 8C) + N(8 + 8H + 8C)
      node phase  fwd 4N(F + C + 2H),                   bwd 4N(F + 2C + 2H)
    over the encoder layers and the decoder layer (H = 1, C = n_dim), plus encoder_to_decoder 8N(hL + h), the
    gather of the sampled pairs 16T x 2hL (T = min(10000, N)) and the SCE loss 16 floor(N/2) n_dim;
    flops 6NFC per layer + 6N hL h + 6 (2T) (2hL) h."""
    t = min(10000, n)
    by = fl = 0.0
    lay = [(n_dim if l == 0 else hidden, heads, hidden) for l in range(layers)] + [(hidden, 1, n_dim)]
    for f, h, c in lay:
        by += e * (8 + 8 * h + 4 * c) + n * (4 + 4 * h + 4 * c) + e * (20 + 16 * h + 8 * c) + n * (8 + 8 * h + 8 * c)
        by += 4 * n * (f + c + 2 * h) + 4 * n * (f + 2 * c + 2 * h)
        fl += 6.0 * n * f * c
    hl = hidden * layers
    by += 8 * n * (hl + hidden) + 16 * t * 2 * hl + 16 * (n // 2) * n_dim
    fl += 6.0 * n * hl * hidden + 6.0 * 2 * t * 2 * hl * hidden
    return by, fl


L2_GATHER_PEAK_GBS = 16800.0   # rows re-gathered from an XCD's L2: 16.8-18.8 TB/s chip-wide (micro-architecture guide)


def classify_sweep(per_use, compulsory, gathered, table, traffic, sec):
    """Roofline fields of one sparse-sweep launch.  ``achieved`` is always the algorithmic (per-use) byte count
    over the launch time.  The bound follows the fabric counters: when the measured traffic is at least ~0.9 of
    the per-use bytes the gathers are NOT served on-die and the launch is priced against the 8 TB/s HBM peak;
    when the fabric moves clearly less than the per-use count (rows re-read from L2 / Infinity Cache) the
    per-use rate is priced against the guide's L2 gather rate instead, so that no fraction exceeds 1."""
    rate = per_use / sec / 1e9
    ratio = traffic / per_use if traffic else None
    on_die = gathered > 0 and ((ratio is not None and ratio < 0.9) or (ratio is None and rate > HBM_PEAK_GBS))
    peak = L2_GATHER_PEAK_GBS if on_die else HBM_PEAK_GBS
    return {"bound": "l2-gather" if on_die else "hbm", "achieved": round(rate, 1), "peak": peak, "unit": "GB/s",
            "frac": round(rate / peak, 4), "per_use_bytes_per_launch": int(per_use),
            "compulsory_bytes_per_launch": int(compulsory),
            "compulsory_frac_of_hbm_peak": round(compulsory / sec / 1e9 / HBM_PEAK_GBS, 4),
            "traffic": traffic, "traffic_over_per_use": round(ratio, 3) if ratio is not None else None,
            "traffic_rate_gbs": round(traffic / sec / 1e9, 1) if traffic else None}


def roofline_rows(prof, struct, workload, steps, n, n_dim, hidden, layers, n_pairs):
    """One row per kernel template instance / launch shape of the step, most expensive first.

    Sparse sweeps: ``achieved`` = per-use bytes (``sweep_bytes``: every tensor once per logical use, a gathered
    row once per edge) over the average launch time, see ``classify_sweep`` for the bound.  ``traffic`` = fabric
    bytes per launch from the PMC counters (``profiles/traffic.json``).  Projection launches are named by shape
    and priced by ``dense_launch``: against the matrix ceiling of their arithmetic past its ridge, against HBM
    below it."""
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
    coded0 = any(k.startswith("gat_fwd_light_coded<") for k in prof)    # first layer ran on code words
    for l, (H, C, r) in enumerate(lay):
        t = f"<{H},{C // (4 * H)}>"
        c = "_coded" if (l == 0 and coded0) else ""
        for base, kind, e_seg in (("gat_fwd_light" + c, "fwd" + c, e_in),
                                  ("gat_bwd_src_light" + c, "bwd_src" + c, e_out),
                                  ("gat_bwd_dst_light", "bwd_dst", e_in), ("gat_bwd_deh_light", "bwd_deh", e_out)):
            sparse.setdefault(base + t, []).append(sweep_bytes(kind, n, e_seg, E, H, C, r))
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
        dense = dense_launch(name, n, hidden, layers, n_pairs)
        if name in sparse:
            per_use, comp, gath, table = (sum(x[i] for x in sparse[name]) / len(sparse[name]) for i in range(4))
            row.update(classify_sweep(per_use, comp, gath, table, traffic_tab.get(name), sec))
        elif dense is not None:
            fl, by, nrows = dense
            # The projections form every fp32 product from six bf16 matrix products (split operands), so their
            # matrix ceiling is the dense bf16 peak / 6; a launch is matrix-bound past the ridge of that ceiling.
            split_peak = BF16_MFMA_PEAK_TF / 6.0
            row["rows"] = nrows
            if fl / by > split_peak * 1e3 / HBM_PEAK_GBS:
                ach = fl / sec / 1e12
                row.update({"bound": "mfma", "achieved": round(ach, 1), "peak": round(split_peak, 1), "unit": "TFLOP/s",
                            "frac": round(ach / split_peak, 4), "frac_of_f32_mfma_peak": round(ach / F32_MFMA_PEAK_TF, 4),
                            "algorithmic_flops_per_launch": int(fl), "traffic": traffic_tab.get(name)})
            else:
                ach = by / sec / 1e9
                row.update({"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(ach / HBM_PEAK_GBS, 4), "compulsory_bytes_per_launch": int(by),
                            "traffic": traffic_tab.get(name)})
        else:
            continue
        rows.append(row)
    return rows, kernels


def gat_leg(args, workload, steps, warmup, world, rank, dev, detail=True):
    """Times `steps` training steps of `workload`; returns the numbers of the JSON line.  ``detail=False``
    (the small side configurations): eager and recorded step time only -- no per-kernel pass, no
    end-to-end loop."""
    from magic_amd import _lib, synth
    from magic_amd.model.autoencoder import GMAEModel, _negative_candidates, negative_candidate_count
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
        if world == 1 and detail:
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
    exchange = flat if multi(world) else None
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
    if not detail:
        return {"ms_per_step": ms_per_step, "edges_per_s": edges_per_s, "eager_ms": eager_ms, "n": n, "e": E}
    # ---- per-kernel timing over a further timed region (HIP events on the launch stream)
    _lib.prof_reset()
    _lib.prof_enable(True)
    dt_prof = timed(step_eager, steps, world)
    _lib.prof_enable(False)
    prof = _lib.prof_read(160)
    seq_path = os.environ.get("MAGIC_BENCH_SEQUENCE")
    if seq_path and rank == 0:
        # the launch names of ONE step, in order (every step launches the same sequence): lets a counter pass over
        # this program attribute its dispatches to the launch shapes named here
        seq = _lib.prof_sequence(1 << 16)
        assert len(seq) % steps == 0 and seq[:len(seq) // steps] == seq[len(seq) // steps:2 * (len(seq) // steps)]
        with open(seq_path, "w") as f:
            json.dump({"workload": workload, "step": seq[:len(seq) // steps]}, f)
    _lib.prof_reset()

    roofline = None
    others = []
    rows = []
    kernels = {}
    if rank == 0:
        rows, kernels = roofline_rows(prof, g.structure, workload, steps, n, n_dim, hidden, layers,
                                       thr + negative_candidate_count(thr))
        if rows:
            roofline = dict(rows[0], profiled_ms_per_step=round(dt_prof / steps * 1e3, 3))
            others = rows[1:4]
    desc = {"theia": "E3-THEIA-shaped full graph", "streamspot_batch": "StreamSpot batch-level training, batch of "
            "12 graphs"}.get(workload, workload)
    adam_ms = sum(v["total_ms_per_step"] for k, v in kernels.items() if k.startswith("adam_"))
    return {"ms_per_step": ms_per_step, "edges_per_s": edges_per_s, "roofline": roofline, "roofline_next": others,
            "roofline_rows": rows, "kernels": kernels, "batch_assembly_ms": assembly_ms, "end_to_end_ms": e2e_ms,
            "eager_ms": eager_ms, "adam_ms": adam_ms, "n": n, "e": E, "shape": (n_dim, e_dim, hidden, layers),
            "kernel_sum_ms": sum(v["total_ms_per_step"] for v in kernels.values()),
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
    if FORCE_DIST:
        os.environ["MAGIC_FORCE_COLLECTIVES"] = "1"
        for key, val in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_PORT", "29533")):
            os.environ.setdefault(key, val)
    if multi(world):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
            _stage_collectives_through_host()
        else:
            dist.init_process_group("nccl", device_id=dev)

    if args.knn_only:
        knn = knn_leg(args, world, rank, dev, k=args.knn_k)
        if rank == 0:
            print(json.dumps(knn))
        if multi(world):
            dist.destroy_process_group()
        return
    leg = gat_leg(args, args.workload, args.steps, args.warmup, world, rank, dev)

    # ---- entity-level full graph (one GPU per graph in the reference's recipe: single-GPU runs only)
    entity = None
    if world == 1 and not args.no_entity and args.workload != "theia":
        torch.cuda.empty_cache()
        entity = gat_leg(args, "theia", args.steps, args.warmup, world, rank, dev)
        torch.cuda.empty_cache()

    # ---- the small configurations (single StreamSpot graph, Wget average / largest): step times only
    configs = {}
    if world == 1 and not args.no_configs and args.workload == "streamspot_batch":
        for key, wl in (("c1", "streamspot"), ("c3_avg", "wget"), ("c3_max", "wget_max")):
            configs[key] = gat_leg(args, wl, max(args.steps, 10), max(args.warmup, 3), world, rank, dev, detail=False)
            torch.cuda.empty_cache()

    # ---- KNN leg (and, on one GPU, the embedding extraction that feeds it in the reference's eval.py)
    knn = None
    if not args.no_knn:
        knn = knn_leg(args, world, rank, dev)
        if world == 1 and rank == 0:
            torch.cuda.empty_cache()
            knn["embed"] = embed_leg(args, dev)
            torch.cuda.empty_cache()
            knn["cadets"] = cadets_leg(args, dev)

    # ---- CPU baselines (rank 0, N=1 only)
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = cpu_gat_baseline(args, *leg["dims"])
        if knn is not None:
            knn["cpu_baseline"] = cpu_knn_baseline(args)

    if rank == 0:
        rows_path = write_rows_file(args, leg, entity, knn)
        print(json.dumps(compact_line(args, world, leg, entity, knn, configs, cpu_baseline, rows_path)))
    if multi(world):
        dist.destroy_process_group()


def write_rows_file(args, leg, entity, knn):
    """Everything too long for the one-line result (one roofline row and one timing entry per kernel of every
    leg) goes to a side file: ``--rows-file``, else ``gpurun_out/bench_rows.json`` when that directory exists,
    else ``bench_rows.json`` beside this script."""
    path = args.rows_file
    if not path:
        d = os.path.join(ROOT, "gpurun_out")
        path = os.path.join(d if os.path.isdir(d) else ROOT, "bench_rows.json")
    doc = {"note": "sparse sweeps: achieved = per-use bytes / average launch time; projections: every fp32-grade "
                   "product is six bf16 MFMA products, matrix ceiling = dense bf16 peak / 6 = "
                   f"{BF16_MFMA_PEAK_TF / 6:.1f} TFLOP/s (the fp32 matrix instruction peaks at {F32_MFMA_PEAK_TF})",
           "top": {"workload": leg["workload"], "roofline_rows": leg["roofline_rows"], "kernels": leg["kernels"]}}
    if entity:
        doc["entity_level"] = {"workload": entity["workload"], "roofline_rows": entity["roofline_rows"],
                               "kernels": entity["kernels"]}
    if knn and "kernels" in knn:
        doc["knn"] = {"workload": knn["config"]["workload"], "kernels": knn["kernels"], "roofline": knn.get("roofline")}
    try:
        with open(path, "w") as f:
            json.dump(doc, f, indent=1)
    except OSError:
        return None
    return os.path.relpath(path, ROOT)


ROOF_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "launches_per_step")


def _roof(r):
    return None if not r else {k: r.get(k) for k in ROOF_KEYS}


def _step_roof(leg):
    """Whole-step figure beside the dominant kernel's: algorithmic bytes of the step at 8 TB/s over the step."""
    by, fl = step_algorithmic(leg["n"], leg["e"], leg["shape"][0], leg["shape"][2], leg["shape"][3])
    t_hbm = by / (HBM_PEAK_GBS * 1e9) * 1e3
    return {"algorithmic_gb": round(by / 1e9, 2), "dense_gflop": round(fl / 1e9, 1),
            "ms_at_hbm_peak": round(t_hbm, 3), "frac": round(t_hbm / leg["ms_per_step"], 4)}


def compact_line(args, world, leg, entity, knn, configs, cpu_baseline, rows_path):
    """The ONE result line (kept under 4 KB: the per-kernel tables live in the rows file)."""
    def r4(x):
        return None if x is None else round(x, 4)
    out = {
        "metric": "masked-GAT fwd+bwd edges/sec", "value": round(leg["edges_per_s"], 1), "unit": "edges/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r4(leg["ms_per_step"]),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": leg["workload"],
                   "parallelism": f"dp{world}: one batch per rank, one flat gradient all-reduce"
                   if multi(world) else "single GPU",
                   "batch_assembly_ms": r4(leg["batch_assembly_ms"]), "recorded_ms_per_step": r4(leg["eager_ms"]),
                   "end_to_end_ms_per_step": r4(leg["end_to_end_ms"]), "adam_ms": r4(leg.get("adam_ms"))},
        "roofline": _roof(leg["roofline"]), "step_roofline": _step_roof(leg) if leg.get("shape") else None,
        "cpu_baseline": None if cpu_baseline is None else {k: cpu_baseline[k] for k in
                                                           ("value", "unit", "cores", "kind", "sample")},
    }
    if entity:
        out["entity_level"] = {"north_star_target": True, "value": round(entity["edges_per_s"], 1), "unit": "edges/s",
                               "ms_per_step": r4(entity["ms_per_step"]), "recorded_ms_per_step": r4(entity["eager_ms"]),
                               "adam_ms": r4(entity.get("adam_ms")),
                               "workload": "E3-THEIA-shaped full graph N=%d E=%d hidden=%d layers=%d, same step" %
                                           (entity["n"], entity["e"], entity["shape"][2], entity["shape"][3]),
                               "roofline": _roof(entity["roofline"]), "step_roofline": _step_roof(entity)}
    if knn:
        k = {key: knn[key] for key in ("metric", "value", "unit", "ms_per_step", "dtype", "steps") if key in knn}
        k["workload"] = knn["config"]["short"]
        k["roofline"] = _roof(knn.get("roofline"))
        for extra in ("cadets", "embed"):
            if knn.get(extra):
                k[extra] = knn[extra]
        cb = knn.get("cpu_baseline")
        k["cpu_baseline"] = None if cb is None else {x: cb[x] for x in ("value", "unit", "cores", "kind", "sample")}
        out["knn"] = k
    if configs:
        out["configs"] = {key: {"n": c["n"], "e": c["e"], "ms_per_step": r4(c["ms_per_step"]),
                                "recorded_ms_per_step": r4(c["eager_ms"]),
                                "edges_per_s": round(c["e"] / min(c["ms_per_step"], c["eager_ms"]) * 1e3, 1)}
                          for key, c in configs.items()}
    out["rows_file"] = rows_path
    return out


def embed_leg(args, dev):
    """First half of the E3-Trace configuration: embedding extraction (``model/autoencoder.py:269-282`` under
    ``eval.py:80-97``): the encoder alone, eval mode, on one E3-Trace-sized train graph with one-hot node types."""
    from magic_amd import synth
    from magic_amd.model.autoencoder import GMAEModel
    from magic_amd.utils.utils import set_random_seed
    n, e, n_dim, e_dim, hidden, layers = synth.SHAPES["trace"]
    n, e = max(64, int(n * args.scale)), max(64, int(e * args.scale))
    set_random_seed(0)
    model = GMAEModel(n_dim, e_dim, hidden, layers, 4, "prelu", 0.1, 0.2, True, "BatchNorm", 0.5, "sce", 3).to(dev).eval()
    g = synth.make_graph(n, e, n_dim, e_dim, seed=1005).to(dev)
    g.structure.ensure_csr()
    with torch.no_grad():
        for _ in range(3):
            emb = model.embed(g)
        assert emb.shape == (n, hidden)
        reps = max(args.steps, 10)
        dt = timed(lambda _i: model.embed(g), reps, 1)
    return {"ms_per_graph": round(dt / reps * 1e3, 4), "nodes_per_s": round(n * reps / dt, 1),
            "workload": f"embed() of one E3-Trace-sized graph: N={n} E={g.num_edges()} n_dim={n_dim} hidden={hidden} layers={layers}"}


def cadets_leg(args, dev):
    """The k = 200 search of the cadets evaluation (``model/eval.py:289``) on a bounded sample: lists of that length
    do not fit the matrix-core filter's lane-per-entry candidate lists, so the whole search runs on the float64
    matrix-core kernel (buffered candidate sets).  Reported next to the k = 10 leg; not part of its value."""
    from magic_amd.knn import NearestNeighbors
    k, d = 200, 64
    nt, nq = max(4096, int(350_000 * args.scale)), max(512, int(65_536 * args.scale))
    gen = torch.Generator(device=dev).manual_seed(1006)
    centers = torch.randn(200, d, device=dev, generator=gen)
    train = centers[torch.randint(0, 200, (nt,), device=dev, generator=gen)] + 0.05 * torch.randn(nt, d, device=dev, generator=gen)
    query = centers[torch.randint(0, 200, (nq,), device=dev, generator=gen)] + 0.05 * torch.randn(nq, d, device=dev, generator=gen)
    nn_ = NearestNeighbors(k, device=dev).fit(train)
    nn_.mean_distances_device(query[:4096], k)
    reps = 2
    dt = timed(lambda _i: nn_.mean_distances_device(query, k), reps, 1)
    tf = 2.0 * nq * nt * d * reps / dt / 1e12
    return {"k": k, "ms_per_step": round(dt / reps * 1e3, 3), "queries_per_s": round(nq * reps / dt, 1),
            "achieved_tflops": round(tf, 2), "frac_of_f64_mfma_peak": round(tf / F64_MFMA_PEAK_TF, 4),
            "workload": f"k=200 search of {nq} queries against {nt} x {d} train rows, float64 matrix-core kernel"}


def knn_leg(args, world, rank, dev, k=10):
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
    nt, nq, d = args.knn_train, args.knn_queries, 64
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
        xt, xq = standardise_device(train, query, sharded=multi(world))
        if multi(world):
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
           "config": {"short": f"E3-Trace-shaped scoring: train {nt}x{d} over {world} rank(s), {nq} test rows + "
                               f"{n_sample}-row mean_distance search per step, k={k}",
                      "workload": f"E3-Trace-shaped outlier scoring: train {nt} x {d} f32 sharded over {world} rank(s), "
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
