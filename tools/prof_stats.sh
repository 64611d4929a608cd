#!/bin/bash
# rocprofv3 --kernel-trace --stats of a bench run:   bash tools/prof_stats.sh <dtype> [tag] [streams]   -> gpurun_out/prof_<tag>/
# streams 0 (default): the single-stream schedule - every launch alone on the chip, the averages are the kernels' own;
# streams 2: the timed loop's schedule (the three parts on three queues): what the kernels take beside each other.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=${1:-bf16x3}
T=${2:-$D}
S=${3:-0}
O=$R/gpurun_out/prof_$T
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as ge; ge.build()" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o stats1 -- python3 "$R/bench.py" --dtype "$D" --streams "$S" --steps 2 --warmup 1 --no-cpu-baseline --no-train-leg --no-roofline > "$O/stats1_bench_line.json" 2> "$O/stats1.err" || exit 1
find "$O" -name "*kernel_stats.csv" | head -3
