#!/bin/bash
# One GPU call: library builds with / without packed-fp32 VALU instructions x 1 / 3 / 6 streams (profiles/r03_slp_streams_ab.json).
# The variant libraries are NOT kept in the tree; build them first from the product flags of pafuse_amd/build_flags.py:
#   noslp        product flags minus the -Xclang feature switch, plus -fno-slp-vectorize
#   noslp_lanes  the same plus -DPAFUSE_ALLOW_BF16_LANES          slp_lanes  neither switch, plus -DPAFUSE_ALLOW_BF16_LANES
#   (hipcc <flags> -shared pafuse_amd/csrc/pafuse_hip.hip -o tools/bin/<variant>/libpafuse_hip.so)
# "product" in this script was the round-2 build (packed fp32, bf16 modes on one stream); today's product build is the fixed one.
set -x
mkdir -p gpurun_out/r3u
O=gpurun_out/r3u
timeout -k 10 200 tools/bin/mfma_queue_isolate 40 > $O/isolate_pk.log 2>&1; tail -40 $O/isolate_pk.log
for v in product noslp; do
  L=tools/bin/$v/libpafuse_hip.so; [ $v = product ] && L=pafuse_amd/libpafuse_hip.so
  timeout -k 10 200 python tools/bench_with_lib.py $L --steps 20 --warmup 5 --streams 0 --no-cpu-baseline > $O/bench_${v}_s0.json 2> $O/bench_${v}_s0.err || tail -5 $O/bench_${v}_s0.err
done
for v in noslp_lanes slp_lanes; do
  timeout -k 10 200 python tools/bench_with_lib.py tools/bin/$v/libpafuse_hip.so --steps 20 --warmup 5 --streams 2 --no-cpu-baseline > $O/bench_${v}_s2.json 2> $O/bench_${v}_s2.err || tail -5 $O/bench_${v}_s2.err
  timeout -k 10 200 python tools/bench_with_lib.py tools/bin/$v/libpafuse_hip.so --steps 20 --warmup 5 --streams 5 --no-cpu-baseline > $O/bench_${v}_s5.json 2> $O/bench_${v}_s5.err || tail -5 $O/bench_${v}_s5.err
done
timeout -k 10 300 python tools/soak_determinism.py 150 bf16x3 2 tools/bin/noslp_lanes/libpafuse_hip.so > $O/soak_noslp_lanes.json 2> $O/soak_noslp_lanes.err
timeout -k 10 300 python tools/soak_determinism.py 60 bf16x3 2 tools/bin/slp_lanes/libpafuse_hip.so > $O/soak_slp_lanes.json 2> $O/soak_slp_lanes.err
cat $O/*.json
