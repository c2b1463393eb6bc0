#!/usr/bin/env python3
"""Print the per-kernel rows of one or more bench.py JSON lines side by side (development aid)."""
import json
import sys


def load(path):
    for line in open(path):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"{path}: no JSON line")


def main():
    recs = [load(p) for p in sys.argv[1:]]
    for leg_name in ("main", "entity_level"):
        legs = [r if leg_name == "main" else r.get("entity_level") for r in recs]
        if not any(legs):
            continue
        print(f"== {leg_name}: ms/step " + "  ".join(f"{l['ms_per_step']:.3f}" if l else "-" for l in legs))
        names = []
        for l in legs:
            for k in (l or {}).get("kernels", {}):
                if k not in names:
                    names.append(k)
        for k in names:
            cells = []
            for l in legs:
                v = (l or {}).get("kernels", {}).get(k)
                cells.append(f"{v['avg_us']:8.1f}us x{v['launches'] // max(recs[0]['steps'], 1):<2d} {v['total_ms_per_step']:.3f}" if v else " " * 20)
            print(f"  {k:32s} " + " | ".join(cells))


if __name__ == "__main__":
    main()
