#!/usr/bin/env python3
"""Groups a rocprofv3 kernel_trace.csv by (kernel, workgroup count): launches, min / median / max duration in us.
    python tools/trace_by_grid.py gpurun_out/prof/x_kernel_trace.csv > profiles/r01_kernel_trace_by_grid.csv
    python tools/trace_by_grid.py TRACE.csv N SKIP
Also prints (stderr) the mean duration of the N GEMM launches of bench.py's roofline replay (the N before the last SKIP
GEMM launches, which are the by_layer legs), the number to hold against `roofline.avg_launch_us`."""
import csv
import statistics
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    last = int(sys.argv[2]) if len(sys.argv) > 2 else 576
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # GEMM launches after the roofline replay (bench.py's by_layer legs)
    groups, gemm = defaultdict(list), []
    for r in csv.DictReader(open(path)):
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        wgs = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
        name = r["Kernel_Name"]
        groups[(name, wgs)].append(us)
        if any(k in name for k in ("gemm_kernel", "gemm16_kernel", "sgemm2_kernel", "gemm_dma_kernel", "grouped_rowln_kernel", "hgemm_kernel", "hfqa_kernel", "hmlp_kernel", "xfqa_kernel", "xgemm_kernel")) and "tn_gemm" not in name:
            gemm.append((int(r["Start_Timestamp"]), us))
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "workgroups", "calls", "min_us", "median_us", "max_us"])
    for (name, wgs), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([name, wgs, len(v), f"{min(v):.2f}", f"{statistics.median(v):.2f}", f"{max(v):.2f}"])
    gemm.sort()
    tail = [us for _, us in (gemm[-last - skip:-skip] if skip else gemm[-last:])]
    if tail:
        print(f"the {len(tail)} GEMM launches of the roofline replay: mean {sum(tail) / len(tail):.2f} us", file=sys.stderr)


if __name__ == "__main__":
    main()
