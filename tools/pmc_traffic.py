#!/usr/bin/env python3
"""HBM-side traffic per kernel family from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -o fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d OUT -o write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
    python tools/pmc_traffic.py OUT/fetch_counter_collection.csv OUT/write_counter_collection.csv OUT/fetch_kernel_trace.csv \
        [--sha256 DIGEST] [--git GITSHA] [--dtype bf16x3] > profiles/r02_pmc_traffic.json

Units and corrections exactly as MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes: both counters are in KiB of
fabric requests; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced streams at 64 bytes, so it is
DOUBLED; WRITE_SIZE is exact for 16-byte-per-lane streaming stores.  Durations come from the same (counter-collecting,
hence serialised) run, so GB/s per family = bytes / summed kernel time is a per-kernel figure, not a whole-loop one.
`--sha256` is the kernel-source digest the profiled bench.py printed (`kernel_source_sha256` of its JSON line); bench.py
only quotes this file when it equals the digest of the tree it runs from and `--dtype` (the product mode of the
profiled run, default bf16x3 = bench.py's default) is the mode it is timing.
"""
import collections
import csv
import json
import sys


def family(name):
    for key, fam in (("gemm16_kernel", "gemm"), ("sgemm2_kernel", "gemm"), ("grouped_rowln_kernel", "gemm"), ("gemm_dma_kernel", "gemm"), ("hfqa_kernel", "gemm"), ("hmlp_kernel", "gemm"), ("xfqa_kernel", "gemm"), ("xgemm_kernel", "gemm"), ("hgemm_kernel", "gemm"), ("tn_gemm_kernel", "gemm_dw"), ("gemm_kernel", "gemm"),
                     ("attn_backward", "attention_bwd"), ("attn_kernel", "attention"), ("embed_kernel", "embed"),
                     ("finalize_kernel", "finalize"), ("time_embed", "time_embed"), ("split_weights", "split_weights"),
                     ("ln_backward", "ln_bwd")):
        if key in name:
            return fam
    return "other"


def per_dispatch(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            out[int(r["Dispatch_Id"])] = (r["Kernel_Name"], float(r["Counter_Value"]))
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opts = dict(zip(sys.argv[1:], sys.argv[2:]))
    fetch, write = per_dispatch(args[0], "FETCH_SIZE"), per_dispatch(args[1], "WRITE_SIZE")
    dur = {}
    for r in csv.DictReader(open(args[2])):
        dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    fam = collections.defaultdict(lambda: {"launches": 0, "fetch_KiB_raw": 0.0, "seconds": 0.0})
    by_kernel = collections.defaultdict(lambda: {"launches": 0, "fetch_KiB_raw": 0.0, "write_KiB": 0.0})
    for d, (name, v) in fetch.items():
        f = fam[family(name)]
        f["launches"] += 1
        f["fetch_KiB_raw"] += v
        f["seconds"] += dur.get(d, 0.0)
        k = by_kernel[name.split("(")[0].replace("void ", "")]
        k["launches"] += 1
        k["fetch_KiB_raw"] += v
    wfam = collections.defaultdict(float)
    for d, (name, v) in write.items():
        wfam[family(name)] += v
        by_kernel[name.split("(")[0].replace("void ", "")]["write_KiB"] += v
    families = {}
    for k, f in sorted(fam.items(), key=lambda kv: -kv[1]["seconds"]):
        bytes_ = (2.0 * f["fetch_KiB_raw"] + wfam[k]) * 1024.0
        families[k] = {"launches": f["launches"], "fetch_bytes_corrected_x2": 2.0 * f["fetch_KiB_raw"] * 1024.0,
                       "write_bytes": wfam[k] * 1024.0, "seconds_serialised": f["seconds"],
                       "hbm_GBps": round(bytes_ / f["seconds"] / 1e9, 1) if f["seconds"] > 0 else None}
    g = families.get("gemm", {"launches": 0})
    doc = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, each with --kernel-trace) over "
                  "python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline (P=20, T=10); all dispatches of the process",
        "kernel_source_sha256": opts.get("--sha256"), "git_sha": opts.get("--git"), "dtype": opts.get("--dtype", "bf16x3"),
        "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 64 B per 128-B request on 16-B/lane streams); "
                "WRITE_SIZE exact for dwordx4 stores; counters are fabric-side: Infinity-Cache hits are included",
        "gemm_launches": g["launches"],
        "traffic_bytes_per_launch": (g["fetch_bytes_corrected_x2"] + g["write_bytes"]) / g["launches"] if g["launches"] else None,
        "hbm_GBps_by_kernel_family": {k: v["hbm_GBps"] for k, v in families.items()},
        "families": families,
        "per_kernel": {k: {"launches": v["launches"],
                           "avg_fetch_MB_corrected": round(2.0 * v["fetch_KiB_raw"] * 1024 / v["launches"] / 1e6, 2),
                           "avg_write_MB": round(v["write_KiB"] * 1024 / v["launches"] / 1e6, 2)}
                       for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1]["fetch_KiB_raw"])},
    }
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
