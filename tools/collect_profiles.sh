#!/bin/bash
# Collects the rocprofv3 evidence behind profiles/rNN_* on the GPU box (run through gpurun from the repo root), ONCE per round on the
# tree that ships (A/Bs in between use `bench.py --no-secondary --no-cpu-baseline --no-train-leg`):
#   bash tools/collect_profiles.sh            -> gpurun_out/prof/*
# One pass per counter group (FETCH_SIZE and WRITE_SIZE cannot share a pass; counters never together with --stats),
# program directly after `--`, from /tmp with TMPDIR=/tmp as the pool requires.  Summaries are made afterwards in the
# build container with tools/pmc_traffic.py, tools/pmc_mfma_util.py, tools/trace_by_grid.py.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
# heartbeat: gpurun kills a call that writes nothing for 7 minutes (the CPU oracle of the parity report can take that long)
( while sleep 60; do date >> "$O/heartbeat.log"; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
run() { echo "== $*" >&2; "$@"; }
# build BEFORE anything runs under the profiler: a stale or missing library would otherwise be compiled (hipcc and its
# children exec'd) inside a profiled, GPU-initialised process, and pollute the first pass
run python3 -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as ge; ge.build()" || exit 1
run rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o stats -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-train-leg --no-secondary > "$O/stats_bench_line.json" 2> "$O/stats.err" || exit 1
# the single-stream schedule (grouped grids): every launch alone on the chip, so the --stats averages are the kernels' own
run rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o stats1 -- python3 "$R/bench.py" --streams 0 --steps 2 --warmup 1 --no-cpu-baseline --no-train-leg --no-secondary > "$O/stats1_bench_line.json" 2> "$O/stats1.err" || exit 1
run rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O" -o fetch -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-train-leg --no-secondary > "$O/fetch_bench_line.json" 2> "$O/fetch.err" || exit 1
run rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O" -o write -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-train-leg --no-secondary > "$O/write_bench_line.json" 2> "$O/write.err" || exit 1
run rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O" -o mfma -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-train-leg --no-secondary > "$O/mfma_bench_line.json" 2> "$O/mfma.err" || exit 1
run rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o train -- python3 "$R/bench.py" --train --steps 3 --warmup 1 --no-cpu-baseline > "$O/train_bench_line.json" 2> "$O/train.err" || exit 1
run python3 "$R/bench.py" > "$O/bench_line.json" 2> "$O/bench.err" || exit 1
run python3 "$R/bench.py" --dtype f32 --no-cpu-baseline --no-train-leg > "$O/bench_line_f32.json" 2>> "$O/bench.err" || exit 1
run python3 "$R/bench.py" --dtype f16x2 --no-cpu-baseline --no-train-leg > "$O/bench_line_f16x2.json" 2>> "$O/bench.err" || exit 1
run rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o stats1h -- python3 "$R/bench.py" --dtype f16x2 --streams 0 --steps 2 --warmup 1 --no-cpu-baseline --no-train-leg --no-roofline > "$O/stats1h_bench_line.json" 2> "$O/stats1h.err" || exit 1
run python3 "$R/bench.py" --train --no-cpu-baseline > "$O/train_bench_line_unprofiled.json" 2>> "$O/bench.err" || exit 1
run python3 "$R/bench.py" --train --train-dtype f32 --no-cpu-baseline > "$O/train_bench_line_f32_unprofiled.json" 2>> "$O/bench.err" || exit 1
run python3 "$R/tools/soak_determinism.py" 200 bf16x3 > "$O/soak_determinism.json" 2>> "$O/bench.err" || exit 1
run python3 "$R/tests/reports/parity_report.py" --out "$O/parity_report.json" > "$O/parity_report.log" 2>&1 || exit 1
ls -la "$O" | head -40
