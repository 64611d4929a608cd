#!/bin/bash
# rocprofv3 --kernel-trace --stats of the training bench:   bash tools/prof_train.sh [tag]   -> gpurun_out/prof_train_<tag>/
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-a}
L=${2:-}
O=$R/gpurun_out/prof_train_$T
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as ge; ge.build()" || exit 1
if [ -n "$L" ]; then PROG="$R/tools/bench_with_lib.py $R/$L"; else PROG="$R/bench.py"; fi
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o train -- python3 $PROG --train --steps 3 --warmup 1 --no-cpu-baseline > "$O/train_bench_line.json" 2> "$O/train.err" || exit 1
python3 - "$O" <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms")
for r in rows[:28]:
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f}% {r['Calls']:>6} {float(r['AverageNs'])/1e3:8.1f}us  {r['Name'][:100]}")
# per (kernel, grid) for the GEMM-like kernels: which shape takes what
import collections
t = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(t)):
    n = r["Kernel_Name"]
    if any(k in n for k in ("tn_", "gemm_kernel", "gemm_dma", "attn_backward", "ln_backward")):
        agg[(n[:70], r["Grid_Size_X"], r["Grid_Size_Y"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
q = collections.defaultdict(float); span = {}
for r in csv.DictReader(open(t)):
    k = (r["Queue_Id"], r["Stream_Id"]); b, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q[k] += (e - b) / 1e6
    lo, hi = span.get(k, (b, e)); span[k] = (min(lo, b), max(hi, e))
print("per (queue, stream): kernel ms, first-to-last span ms")
for k, v in sorted(q.items(), key=lambda kv: -kv[1])[:8]:
    print(f"  queue {k[0]} stream {k[1]}: {v:8.1f} {(span[k][1] - span[k][0]) / 1e6:8.1f}")
print("per (kernel, grid x, grid y): calls, mean us, min us")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {len(v):5d} {sum(v)/len(v):8.1f} {min(v):8.1f}  {k[0]}  grid {k[1]} x {k[2]}")
PY
rm -f "$O"/*kernel_trace.csv
