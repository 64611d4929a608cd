#!/usr/bin/env python3
"""Where the waves of each kernel of the DDIM loop spend their cycles, from rocprofv3 PMC passes (one counter group per pass, each

    rocprofv3 --pmc <group> --kernel-trace --output-format csv -d out -o <tag> -- \
        python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-train-leg --no-secondary --no-roofline

    python tools/pmc_issue.py out/a_counter_collection.csv out/b_counter_collection.csv ... > profiles/rNN_pmc_issue_breakdown.json

Counters are summed per kernel over all its launches (the passes run the same launches); every SQ counter is reported per wave-cycle
(SQ_WAVE_CYCLES: the cycles waves are resident, summed over waves; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, and
WAIT_ANY (parked at s_waitcnt / a barrier) + WAIT_INST_ANY (issue stall: MFMA read-after-write, a busy pipe) + ACTIVE_INST_ANY ~ WAVE_CYCLES;
SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD, so its ratio to wave quad-cycles is not a fraction), instruction counts per wave."""
import collections
import csv
import json
import sys


def main():
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(int)
    passes = collections.defaultdict(set)   # a counter collected in several passes (SQ_WAVE_CYCLES, SQ_WAVES) is averaged over them
    for i, path in enumerate(sys.argv[1:]):
        seen = set()
        for r in csv.DictReader(open(path)):
            if "pafuse" not in r["Kernel_Name"]:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            passes[r["Counter_Name"]].add(i)
            if i == 0 and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                launches[k] += 1
    for c in tot.values():
        for name in c:
            c[name] /= len(passes[name])
    rows = []
    for k, c in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0.0)):
        wc, waves = c.get("SQ_WAVE_CYCLES", 0.0), c.get("SQ_WAVES", 0.0)
        row = {"kernel": k, "launches": launches.get(k)}
        for name, v in sorted(c.items()):
            if name in ("SQ_WAVE_CYCLES", "SQ_WAVES", "GRBM_GUI_ACTIVE"):
                continue
            if name.startswith("SQ_INSTS"):
                row[name + "_per_wave"] = round(v / waves, 1) if waves else None
            else:
                row[name + "_per_wave_cycle"] = round(v / wc, 4) if wc else None
        row["wave_cycles_per_wave"] = round(wc / waves, 0) if waves else None
        rows.append(row)
    print(json.dumps({"per_kernel": rows}, indent=1))


if __name__ == "__main__":
    main()
