#!/bin/bash
# Turns the raw outputs of tools/collect_profiles.sh (gpurun_out/prof/) into the committed summaries under profiles/.
#   bash tools/summarise_profiles.sh [GITSHA of the collected tree, default HEAD] [round tag, default r06]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/prof
GIT=${1:-$(git -C "$R" rev-parse HEAD)}
RND=${2:-r06}
SHA=$(python3 -c "import json;print(json.load(open('$O/fetch_bench_line.json'))['kernel_source_sha256'])")
N=$(python3 -c "import json;print(json.load(open('$O/stats_bench_line.json'))['roofline']['launches'])")
# GEMM launches bench.py issues after the roofline replay: every by_layer leg once as warm-up + its timed repetitions
SKIP=$(python3 -c "import json;b=json.load(open('$O/stats_bench_line.json'))['roofline'].get('by_layer',{});print(sum(v['launches'] for v in b.values())*4//3)")
python3 "$R/tools/pmc_traffic.py" "$O/fetch_counter_collection.csv" "$O/write_counter_collection.csv" "$O/fetch_kernel_trace.csv" \
    --sha256 "$SHA" --git "$GIT" --dtype bf16x3 > "$R/profiles/${RND}_pmc_traffic.json"
cp "$O/stats_kernel_stats.csv" "$R/profiles/${RND}_kernel_stats_bench_P20_T10.csv"
python3 "$R/tools/trace_by_grid.py" "$O/stats_kernel_trace.csv" "$N" "$SKIP" > "$R/profiles/${RND}_kernel_trace_by_grid.csv"
python3 "$R/tools/pmc_mfma_util.py" "$O/mfma_counter_collection.csv" "$N" "$SKIP" > "$R/profiles/${RND}_pmc_mfma_util.json"
cp "$O/stats1_kernel_stats.csv" "$R/profiles/${RND}_kernel_stats_bench_one_stream.csv"
cp "$O/stats1_bench_line.json" "$R/profiles/${RND}_bench_line_one_stream_under_rocprof.json"
cp "$O/train_kernel_stats.csv" "$R/profiles/${RND}_train_kernel_stats_B37.csv"
cp "$O/bench_line.json" "$R/profiles/${RND}_bench_line.json"
cp "$O/bench_line_f32.json" "$R/profiles/${RND}_bench_line_f32.json"
cp "$O/bench_line_f16x2.json" "$R/profiles/${RND}_bench_line_f16x2.json"
cp "$O/stats1h_kernel_stats.csv" "$R/profiles/${RND}_kernel_stats_f16x2_one_stream.csv"
cp "$O/stats_bench_line.json" "$R/profiles/${RND}_bench_line_under_rocprof.json"
cp "$O/train_bench_line.json" "$R/profiles/${RND}_train_bench_line_under_rocprof.json"
cp "$O/train_bench_line_unprofiled.json" "$R/profiles/${RND}_train_bench_line_unprofiled.json"
cp "$O/train_bench_line_f32_unprofiled.json" "$R/profiles/${RND}_train_bench_line_f32_unprofiled.json"
cp "$O/soak_determinism.json" "$R/profiles/${RND}_soak_determinism.json"
cp "$O/parity_report.json" "$R/profiles/${RND}_parity_report.json"
python3 "$R/tools/kernel_resources.py" > "$R/profiles/${RND}_kernel_resources.txt"
echo "kernel sources $SHA (tree now: $(python3 -c "import sys; sys.path.insert(0,'$R'); from pafuse_amd._lib import kernel_source_digest as k; print(k())"))"
