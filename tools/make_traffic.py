#!/usr/bin/env python3
"""Build profiles/traffic.json and profiles/<round>_pmc_summary_<workload>.txt from two rocprofv3 --pmc
passes (FETCH_SIZE in one run, WRITE_SIZE in another) over `bench.py --steps 2 --warmup 1`.

HBM bytes per launch = 2 * FETCH_SIZE KiB + WRITE_SIZE KiB.  The factor 2 on the read side is the
gfx950 correction for 16-byte-per-lane loads (FETCH_SIZE tallies 128-byte requests at 64 bytes);
it is re-checked on every run against a kernel of this library whose byte count is known exactly:
gat_bwd_node_adjoint<4,LPH> streams two [N, 16*LPH] fp32 operands with float4 loads and writes one.
usage: make_traffic.py <pmc_rd_dir> <pmc_wr_dir> <workload> <n_nodes> <n_encoder_layers>
"""
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import summarise  # noqa: E402

BENCH_NAME = {"k_fwd_light": "gat_fwd_light", "k_bwd_src_light": "gat_bwd_src_light",
              "k_bwd_dst_light": "gat_bwd_dst_light", "k_bwd_deh_light": "gat_bwd_deh_light"}
ROUND = os.environ.get("MAGIC_PROFILE_ROUND", "r02")


def main():
    rd_dir, wr_dir, workload, n_nodes = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    n_enc = int(sys.argv[5]) if len(sys.argv) > 5 else 3
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
        lines.append(f"{k:44s} {f:12.0f} {w:12.0f} {total / 1e6:16.1f}")
        traffic[k] = total
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
            out["gat_bwd_src_light<4,16>"] = int(v)
        if base in BENCH_NAME and "<" in k:
            args = k.split("<", 1)[1].rstrip(">").replace(" ", "").split(",")
            if len(args) == 1:                                  # templated on the head count only
                args = [args[0], lph_of.get(args[0], "?")]
            args = args[:2]                                     # (heads, lanes per head); drop further flags
            out[BENCH_NAME[base] + "<" + ",".join(args) + ">"] = int(v)
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
