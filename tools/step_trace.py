#!/usr/bin/env python3
"""Print the kernel sequence of the last training step from a rocprofv3 --kernel-trace CSV.

usage: step_trace.py <kernel_trace.csv> [anchor-substring]
The anchor (default: the first kernel-name substring 'proj_fwd' ... of a step) splits the trace
into steps: a step starts at the first anchor kernel that follows an 'adam'/'multi_tensor' kernel.
"""
import csv
import re
import sys


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(at::native::)?([\w:]+)(<.*)?", name)
    base = m.group(2) if m else name
    if base.endswith("elementwise_kernel") or base.endswith("elementwise_kernel_manual_unroll"):
        f = re.search(r"at::native::(\w+?)(_kernel|_functor|Functor|<)", name[20:])
        base = "ew:" + (f.group(1) if f else "?")
    tmpl = re.match(r"[\w:]+<([^>]{0,24})", name)
    if tmpl and ("k_proj" in name or "k_fwd" in name or "k_bwd" in name):
        base += "<" + tmpl.group(1) + ">"
    return base[:60]


def main():
    path = sys.argv[1]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # steps end with the optimizer's multi-tensor kernels
    ends = [i for i, r in enumerate(rows) if "multi_tensor" in r[2] or "adam" in r[2].lower()]
    if not ends:
        first, last = 0, len(rows)
    else:
        last = ends[-1] + 1
        # previous step's end: last optimizer kernel separated from the final group by other kernels
        j = len(ends) - 1
        while j > 0 and ends[j - 1] >= ends[j] - 8:
            j -= 1
        first = ends[j - 1] + 1 if j > 0 else 0
    step = rows[first:last]
    t0 = step[0][0]
    busy = 0
    prev_end = t0
    print(f"{'start_us':>9} {'dur_us':>8} {'gap_us':>7}  kernel")
    for s, e, n in step:
        busy += e - s
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f}  {short(n)}")
        prev_end = max(prev_end, e)
    print(f"# {len(step)} kernels, span {(step[-1][1] - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms")


if __name__ == "__main__":
    main()
