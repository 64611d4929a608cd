#!/usr/bin/env python3
"""MFMA-pipe utilisation of the roofline leg's GEMM launches from a rocprofv3 PMC pass:

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d out -o m -- \
        python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
    python tools/pmc_mfma_util.py out/m_counter_collection.csv [N [SKIP]] > profiles/r02_pmc_mfma_util.json
(N launches of the roofline replay, before the last SKIP GEMM launches = the by_layer legs)

SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs (checked: = 64 cycles x the number of wave-level
v_mfma_f32_32x32x2_f32 the launch issues), GRBM_GUI_ACTIVE over its 8 XCDs, so
utilisation = busy / (1024 * active / 8)."""
import collections
import csv
import json
import sys


def main():
    last = int(sys.argv[2]) if len(sys.argv) > 2 else 576
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # GEMM launches after the roofline replay (bench.py's by_layer legs)
    by = collections.defaultdict(dict)
    for r in csv.DictReader(open(sys.argv[1])):
        d = by[int(r["Dispatch_Id"])]
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        d["name"], d["grid"], d["wg"] = r["Kernel_Name"], int(r["Grid_Size"]), int(r["Workgroup_Size"])
    gemm = [v for _, v in sorted(by.items()) if any(k in v["name"] for k in ("gemm_kernel", "gemm16_kernel", "sgemm2_kernel", "gemm_dma_kernel", "grouped_rowln_kernel", "hfqa_kernel", "hmlp_kernel", "xfqa_kernel", "xgemm_kernel"))]
    gemm = gemm[-last - skip:-skip] if skip else gemm[-last:]
    groups = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for v in gemm:
        g = groups[(v["name"].split("(")[0].replace("void ", ""), v["grid"] // v["wg"])]
        g[0] += v["SQ_VALU_MFMA_BUSY_CYCLES"]
        g[1] += v["GRBM_GUI_ACTIVE"]
        g[2] += 1
    util = lambda busy, active: busy / (1024.0 * active / 8.0)
    out = {"counters": ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"], "launches": len(gemm),
           "formula": "busy / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)",
           "mfma_utilisation_all": round(util(sum(g[0] for g in groups.values()), sum(g[1] for g in groups.values())), 4),
           "per_kernel": [{"kernel": k[0], "workgroups": k[1], "launches": g[2], "mfma_utilisation": round(util(g[0], g[1]), 4)}
                          for k, g in sorted(groups.items(), key=lambda kv: -kv[1][1])]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
