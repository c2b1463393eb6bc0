#!/usr/bin/env python3
"""Build profiles/traffic.json and profiles/<round>_pmc_summary_<workload>.txt from two rocprofv3 --pmc
passes (FETCH_SIZE in one run, WRITE_SIZE in another) over `bench.py --steps 2 --warmup 1`.

HBM bytes per launch = 2 * FETCH_SIZE KiB + WRITE_SIZE KiB.  The factor 2 on the read side is the
gfx950 correction for 16-byte-per-lane loads (FETCH_SIZE tallies 128-byte requests at 64 bytes);
it is re-checked on every run against a kernel of this library whose byte count is known exactly:
gat_bwd_node_adjoint<4,LPH> streams two [N, 16*LPH] fp32 operands with float4 loads and writes one.
usage: make_traffic.py <pmc_rd_dir> <pmc_wr_dir> <workload> <n_nodes> <n_encoder_layers> [<step sequence json>]
"""
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import summarise  # noqa: E402

BENCH_NAME = {"k_fwd_light": "gat_fwd_light", "k_bwd_src_light": "gat_bwd_src_light",
              "k_bwd_dst_light": "gat_bwd_dst_light", "k_bwd_deh_light": "gat_bwd_deh_light"}
ROUND = os.environ.get("MAGIC_PROFILE_ROUND", "r03")


PROJ_MAIN = ("k_proj_rows<", "k_proj_strip<", "k_proj_gemm<", "k_proj_stream<", "k_proj<", "k_proj_dw<", "k_proj_dw_wide")


def _clean(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("magic::", "").strip()


def dispatches(directory, counter):
    """[(dispatch id, kernel name, value)] of one counter, in dispatch order."""
    import csv
    rows = []
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), _clean(r["Kernel_Name"]), float(r["Counter_Value"])))
    rows.sort()
    return rows


def projection_traffic(rd_dir, wr_dir, seq_path):
    """HBM bytes per launch of every projection launch SHAPE of the step.  Every step launches the same
    sequence of kernels; ``seq_path`` (written by bench.py under MAGIC_BENCH_SEQUENCE) names the launches of one
    step in order, so the j-th main projection kernel dispatch of the counter pass is launch j mod p of that
    sequence.  Returns ({shape name: bytes}, report lines)."""
    step = [n for n in json.load(open(seq_path))["step"]
            if n.startswith(("proj_fwd<", "proj_bwd_input<", "proj_bwd_weight<"))]
    p = len(step)
    out, lines = {}, []
    per = {}
    for ctr, directory, scale in (("FETCH_SIZE", rd_dir, 2.0), ("WRITE_SIZE", wr_dir, 1.0)):
        d = [x for x in dispatches(directory, ctr)
             if x[1].startswith(PROJ_MAIN) and not x[1].startswith("k_proj_dw_wide_reduce")]
        if not d or len(d) % p:
            lines.append(f"# projection launches: {len(d)} dispatches do not tile the {p}-launch step sequence ({ctr})")
            return {}, lines
        kern = {}
        for j, (_, k, v) in enumerate(d):
            name = step[j % p]
            kern.setdefault(name, set()).add(k)
            per.setdefault(name, {}).setdefault(ctr, []).append(v * scale * 1024)
        bad = {n: ks for n, ks in kern.items() if len(ks) > 1 and step.count(n) == 1}
        if bad:
            lines.append(f"# projection launches: sequence and dispatches disagree: {bad}")
            return {}, lines
    lines.append(f"{'projection launch (kernel)':44s} {'read MB':>12s} {'write MB':>12s} {'HBM MB':>16s}")
    for name in sorted(per):
        rd = sum(per[name].get("FETCH_SIZE", [0])) / max(1, len(per[name].get("FETCH_SIZE", [0])))
        wr = sum(per[name].get("WRITE_SIZE", [0])) / max(1, len(per[name].get("WRITE_SIZE", [0])))
        out[name] = int(rd + wr)
        lines.append(f"{name:44s} {rd / 1e6:12.1f} {wr / 1e6:12.1f} {(rd + wr) / 1e6:16.1f}")
    return out, lines


def main():
    rd_dir, wr_dir, workload, n_nodes = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    n_enc = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    seq_path = sys.argv[6] if len(sys.argv) > 6 else None
    rd = {}
    for f in glob.glob(os.path.join(rd_dir, "**", "*counter_collection.csv"), recursive=True):
        for k, v in summarise(f).items():
            if "FETCH_SIZE" in v:
                rd[k] = v["FETCH_SIZE"]
    wr = {}
    for f in glob.glob(os.path.join(wr_dir, "**", "*counter_collection.csv"), recursive=True):
        for k, v in summarise(f).items():
            if "WRITE_SIZE" in v:
                wr[k] = v["WRITE_SIZE"]
    lines = []
    cal_key = next((k for k in rd if k.startswith("k_bwd_node_adjoint<4,")), None)
    cal = rd.get(cal_key) if cal_key else None
    if cal:
        lph = int(cal_key.split(",")[1].strip(" >"))
        exact = 2 * n_nodes * 16 * lph * 4 / 1024
        lines.append(f"# calibration: gat_{cal_key[2:]} reads exactly {exact:.0f} KiB, FETCH_SIZE reports "
                     f"{cal:.0f} KiB -> factor {exact / cal:.3f} (the gfx950 factor 2 for 16-byte-per-lane loads)")
    lines.append(f"{'kernel':44s} {'FETCH KiB':>12s} {'WRITE KiB':>12s} {'HBM MB (2*F+W)':>16s}")
    traffic = {}
    for k in sorted(set(rd) | set(wr)):
        f, w = rd.get(k, 0.0), wr.get(k, 0.0)
        total = (2 * f + w) * 1024
        traffic[k] = total
        if k.startswith("k_"):      # the library's kernels (the tensor library's own launches are left out)
            lines.append(f"{k[:44]:44s} {f:12.0f} {w:12.0f} {total / 1e6:16.1f}")
    # bench.py reports one row per template instance ("gat_bwd_src_light<4,4>")
    out = {}
    lph_of = {}     # head count -> lanes per head of this workload's layers (from the two-argument templates)
    for k in traffic:
        args = k.split("<", 1)[1].rstrip(">").replace(" ", "").split(",") if "<" in k else []
        if k.startswith("k_fwd_light<") and len(args) >= 2:
            lph_of[args[0]] = args[1]
    for k, v in traffic.items():
        base = k.split("<")[0]
        if base == "k_bwd_src_wave":    # the wavefront-per-node source sweep exists for 4 heads x 16 lanes only
            coded = k.replace(" ", "").endswith(",true>")
            out["gat_bwd_src_light" + ("_coded" if coded else "") + "<4,16>"] = int(v)
        if base in BENCH_NAME and "<" in k:
            args = k.split("<", 1)[1].rstrip(">").replace(" ", "").split(",")
            if len(args) == 1:                                  # templated on the head count only
                args = [args[0], lph_of.get(args[0], "?")]
            # (heads, lanes per head); of the further flags only "coded layer input" (the last one of the forward
            # and source sweeps) is part of the bench name
            coded = base in ("k_fwd_light", "k_bwd_src_light") and len(args) == 4 and args[3] == "true"
            out[BENCH_NAME[base] + ("_coded" if coded else "") + "<" + ",".join(args[:2]) + ">"] = int(v)
    if workload == "knn":               # the KNN search: its kernels by their bench names
        out = {}
        for k, v in traffic.items():
            if k.startswith("k_knn_filter_bf16"):
                out["knn_filter_bf16x6_mfma"] = int(v)
            elif k.startswith("k_knn_mfma"):
                out["knn_topk_f64_mfma"] = int(v)
            elif k.startswith("k_knn_refine"):
                out["knn_refine"] = int(v)
    if seq_path and os.path.exists(seq_path):
        proj, plines = projection_traffic(rd_dir, wr_dir, seq_path)
        out.update(proj)
        lines += plines
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
    path = os.path.join(root, "traffic.json")
    data = {}
    if os.path.exists(path):
        data = json.load(open(path))
    data[workload] = out
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    open(os.path.join(root, f"{ROUND}_pmc_summary_{workload}.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:3]))
    print(out)


if __name__ == "__main__":
    main()
