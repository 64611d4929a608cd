#!/bin/bash
# the --streams 0 rocprofv3 pass of tools/collect_profiles.sh alone (gpurun_out/prof/stats1_*)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as ge; ge.build()" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o stats1 -- python3 "$R/bench.py" --streams 0 --steps 2 --warmup 1 --no-cpu-baseline --no-train-leg > "$O/stats1_bench_line.json" 2> "$O/stats1.err"
