#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel name, mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict


def summarise(pattern):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("magic::", "")
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, cs in acc.items():
        out[k] = {c: sum(v) / len(v) for c, v in cs.items()}
        out[k]["_count"] = max(len(v) for v in cs.values())
    return out


if __name__ == "__main__":
    out = {}
    for pat in sys.argv[1:]:
        for k, cs in summarise(pat).items():
            out.setdefault(k, {}).update(cs)
    keys = [k for k in out if k.startswith("k_")]
    for k in sorted(keys):
        print(k)
        for c, v in sorted(out[k].items()):
            if c == "_count":
                continue
            print(f"    {c:32s} {v:16.1f}")
