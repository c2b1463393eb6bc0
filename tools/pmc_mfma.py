#!/usr/bin/env python3
"""Matrix-pipe utilisation per kernel from one rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES).

  busy share = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)   matrix pipe busy while its CU is busy
(GRBM_GUI_ACTIVE is printed as collected: the sum over the chip's XCDs, so it is no per-launch cycle count.)
usage: pmc_mfma.py <pmc dir> [<pmc dir> ...] > profiles/rNN_pmc_mfma.txt
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import summarise  # noqa: E402

WANT = ("k_knn_filter_bf16", "k_knn_mfma", "k_proj_gemm", "k_proj_dw_wide", "k_proj_rows", "k_proj_strip", "k_proj_dw<",
        "k_proj<", "k_proj_stream")


def main():
    print("# one pass: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -- python3 bench.py ...")
    print("# mean per dispatch of each kernel template instance; busy share = MFMA_BUSY / (4 SIMDs x BUSY_CU): the share of")
    print("# its CU's busy time in which a SIMD's matrix pipe was busy (the split-bf16 kernels issue six bf16 matrix")
    print("# instructions per fp32-grade product)")
    print(f"{'kernel':46s} {'dispatches':>10s} {'MFMA_BUSY':>14s} {'BUSY_CU':>14s} {'GUI_ACTIVE':>12s} {'busy share':>11s}")
    for d in sys.argv[1:]:
        print(f"# {os.path.basename(os.path.normpath(d))}")
        acc = summarise(os.path.join(d, "**", "*counter_collection.csv"))
        for k in sorted(acc):
            if not k.startswith(WANT) or "reduce" in k or "sums" in k:
                continue
            c = acc[k]
            mf, cu, ga = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("SQ_BUSY_CU_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
            share = mf / (4.0 * cu) if cu else float("nan")
            print(f"{k[:46]:46s} {int(c.get('_count', 0)):10d} {mf:14.0f} {cu:14.0f} {ga:12.0f} {share:11.3f}")


if __name__ == "__main__":
    main()
