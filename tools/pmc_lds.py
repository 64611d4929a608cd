#!/usr/bin/env python3
"""LDS bank-conflict share of every kernel of the DDIM loop from a rocprofv3 PMC pass:

    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE [GRBM_GUI_ACTIVE] --kernel-trace --output-format csv -d out -o lds -- \
        python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-train-leg --no-secondary --no-roofline
    python tools/pmc_lds.py out/lds_counter_collection.csv > profiles/rNN_pmc_lds_conflicts.json

SQ_LDS_BANK_CONFLICT = cycles the LDS is stalled by bank conflicts, SQ_LDS_IDX_ACTIVE = cycles it is busy with indexed
operations (both summed over the chip); conflict_share = the former over the latter, per kernel over all its launches.
With GRBM_GUI_ACTIVE (summed over the 8 XCDs) in the same pass: lds_busy = SQ_LDS_IDX_ACTIVE / (256 CUs x GRBM_GUI_ACTIVE / 8) - the share of
a launch's cycles in which a CU's LDS is busy with ds_read / ds_write (the LDS-DMA's writes are not indexed operations and not in it)."""
import collections
import csv
import json
import sys


def main():
    by = collections.defaultdict(dict)
    for r in csv.DictReader(open(sys.argv[1])):
        d = by[int(r["Dispatch_Id"])]
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        d["name"] = r["Kernel_Name"]
    groups = collections.defaultdict(lambda: [0.0, 0.0, 0, 0.0])
    for v in by.values():
        if "pafuse" not in v["name"]:
            continue
        g = groups[v["name"].split("(")[0].replace("void ", "")]
        g[0] += v.get("SQ_LDS_BANK_CONFLICT", 0.0)
        g[1] += v.get("SQ_LDS_IDX_ACTIVE", 0.0)
        g[2] += 1
        g[3] += v.get("GRBM_GUI_ACTIVE", 0.0)
    rows = [{"kernel": k, "launches": g[2], "lds_active_cycles": int(g[1]), "bank_conflict_cycles": int(g[0]),
             "conflict_share": round(g[0] / g[1], 4) if g[1] else None,
             "lds_busy": round(g[1] / (256.0 * g[3] / 8.0), 4) if g[3] else None} for k, g in sorted(groups.items(), key=lambda kv: -kv[1][1])]
    print(json.dumps({"counters": ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"],
                      "conflict_share_all": round(sum(g[0] for g in groups.values()) / max(1.0, sum(g[1] for g in groups.values())), 4),
                      "per_kernel": rows}, indent=1))


if __name__ == "__main__":
    main()
